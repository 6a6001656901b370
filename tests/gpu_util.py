"""Helpers for the -m gpu tests: move lists of numpy buffers through the batched C ABI."""
import numpy as np
import torch

from vbz_compression_amd import _lib, batch

_codec = None
SRC_ALIGN = 64   # tests may lower these to exercise unaligned arena offsets
DST_ALIGN = 64
SRC_SKEW = 0     # extra bytes in front of the first buffer


def codec():
    global _codec
    if _codec is None:
        _codec = batch.GpuCodec(0)
    return _codec


def _pack(bufs, align=None):
    align = SRC_ALIGN if align is None else align
    sizes = [int(b.nbytes) for b in bufs]
    off, total = batch.layout(sizes, align)
    off = off + SRC_SKEW
    total += SRC_SKEW
    arena = np.zeros(total + 64, np.uint8)
    for b, o in zip(bufs, off.tolist()):
        arena[o : o + b.nbytes] = np.frombuffer(np.ascontiguousarray(b).tobytes(), np.uint8)
    return arena, off, sizes


def run_stage(fn, bufs, caps, **kw):
    """fn(codec, src, src_off, src_size, dst, dst_off, dst_cap, result, **kw); returns list of bytes or error ints."""
    c = codec()
    dev = c.device
    arena, off, sizes = _pack(bufs)
    src = torch.from_numpy(arena).to(dev)
    src_off = off.to(dev)
    src_size = torch.tensor(sizes, dtype=torch.int64).to(torch.int32).to(dev)
    doff, dtotal = batch.layout([int(x) + 32 for x in caps], DST_ALIGN)
    dst = torch.zeros(dtotal + 64, dtype=torch.uint8, device=dev)
    dst_off = doff.to(dev)
    caps64 = torch.tensor([int(x) for x in caps], dtype=torch.int64)
    dst_cap = torch.where(caps64 >= 2**31, caps64 - 2**32, caps64).to(torch.int32).to(dev)
    result = torch.full((len(bufs),), -8, dtype=torch.int32, device=dev)
    fn(c, src, src_off, src_size, dst, dst_off, dst_cap, result, **kw)
    torch.cuda.synchronize()
    res = [int(x) & 0xFFFFFFFF for x in result.cpu().tolist()]
    host = dst.cpu().numpy()
    out = []
    for r, o, cap in zip(res, doff.tolist(), caps):
        if _lib.is_error(r):
            out.append(r)
        else:
            assert r <= cap, (r, cap)
            out.append(host[o : o + r].copy())
    return out


def svb_compress(bufs, size, zigzag, version=0):
    caps = [(b.nbytes // size + 3) // 4 + 4 * (b.nbytes // size) for b in bufs]
    return run_stage(lambda c, *a: c.svb_compress(*a, size=size, zigzag=zigzag, version=version), bufs, caps)


def svb_decompress(streams, nbytes, size, zigzag, version=0):
    return run_stage(lambda c, *a: c.svb_decompress(*a, size=size, zigzag=zigzag, version=version), streams, nbytes)


def zstd_compress(streams, key_bytes=None):
    caps = [s.nbytes + (s.nbytes >> 8) + 64 + 16 for s in streams]
    if key_bytes is None:
        return run_stage(lambda c, *a: c.zstd_compress(*a), streams, caps)
    kb = torch.tensor(key_bytes, dtype=torch.int32, device=codec().device)
    return run_stage(lambda c, *a: c.zstd_compress(*a, key_bytes=kb), streams, caps)


def zstd_decompress(frames, caps):
    return run_stage(lambda c, *a: c.zstd_decompress(*a), frames, caps)


def compress(bufs, opts, sized=False):
    L = _lib.load()
    import ctypes

    caps = [L.vbz_max_compressed_size(b.nbytes, ctypes.byref(opts)) for b in bufs]
    caps = [c if not _lib.is_error(c) else 64 for c in caps]
    return run_stage(lambda c, *a: c.compress(*a, opts, sized=sized), bufs, caps)


def decompress(frames, nbytes, opts, sized=False):
    return run_stage(lambda c, *a: c.decompress(*a, opts, sized=sized), frames, nbytes)
