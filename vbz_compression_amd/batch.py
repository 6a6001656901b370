"""Batched, device-resident codec: torch tensors in HBM -> vbz_gpu_*_batch (include/vbz_gpu.h).

PyTorch is plumbing here (device memory, streams, torch.distributed); the codec itself is the HIP
library.  All tensors must live on the codec's device.  Offsets are int64, sizes/results int32
tensors whose bits are read as uint64 / uint32 by the C ABI.
"""
import ctypes

import torch

from . import _lib


def _u32(t):
    """view an int32 result tensor as python ints in [0, 2**32)"""
    return [int(x) & 0xFFFFFFFF for x in t.tolist()]


class GpuCodec:
    """One context on one GPU.  The codec owns a torch stream (`self.stream`) and the C library
    launches every kernel on it; each call first makes that stream wait for torch's current stream
    and afterwards makes the current stream wait for the codec, so tensors produced or consumed by
    ordinary torch code need no extra synchronisation.  Under `with torch.cuda.stream(codec.stream)`
    both waits are no-ops."""

    def __init__(self, device=None):
        self.L = _lib.load()
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(self.device)
            self.ctx = self.L.vbz_gpu_create(self.device.index, ctypes.c_void_p(self.stream.cuda_stream))
        if not self.ctx:
            raise RuntimeError("vbz_gpu_create failed: no usable gfx950 device (the codec has no CPU path)")

    def _enter(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
        return cur

    def _exit(self, cur):
        if cur != self.stream:
            cur.wait_stream(self.stream)

    def set_trailers(self, enable):
        """Decoder hints (checkpoints, span index) in skippable frames behind the zstd frame: on by default (include/vbz_gpu.h)."""
        self.L.vbz_gpu_set_trailers(self.ctx, int(bool(enable)))

    def set_canonical(self, enable):
        """A read's compressed bytes depend on the read, the options and the library version only -- not on the batch it arrives in
        (include/vbz_gpu.h: vbz_gpu_set_canonical)."""
        self.L.vbz_gpu_set_canonical(self.ctx, int(bool(enable)))

    def close(self):
        if getattr(self, "ctx", None):
            self.L.vbz_gpu_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------------
    def _batch(self, src, src_off, src_size, dst, dst_off, dst_cap, result):
        n = int(src_off.numel())
        for t, dt in ((src_off, torch.int64), (dst_off, torch.int64), (src_size, torch.int32), (dst_cap, torch.int32), (result, torch.int32)):
            assert t.dtype == dt and t.is_contiguous() and t.device == self.device, (t.dtype, dt, t.device)
        assert src.dtype == torch.uint8 and dst.dtype == torch.uint8
        b = _lib.GpuBatch()
        b.n_reads = n
        b.src = src.data_ptr()
        b.src_off = src_off.data_ptr()
        b.src_size = src_size.data_ptr()
        b.src_bytes = src.numel()
        b.dst = dst.data_ptr()
        b.dst_off = dst_off.data_ptr()
        b.dst_cap = dst_cap.data_ptr()
        b.dst_bytes = dst.numel()
        b.result = result.data_ptr()
        return b

    def _rc(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.vbz_gpu_last_error(self.ctx).decode()))

    @staticmethod
    def options(zigzag=True, size=2, level=1, version=1):
        return _lib.CompressionOptions(bool(zigzag), int(size), int(level), int(version))

    # -- full path -------------------------------------------------------------------------------
    def compress(self, src, src_off, src_size, dst, dst_off, dst_cap, result, opts, sized=False):
        b = self._batch(src, src_off, src_size, dst, dst_off, dst_cap, result)
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_compress_batch(self.ctx, ctypes.byref(b), ctypes.byref(opts), int(sized)), "compress_batch")
        finally:
            self._exit(cur)

    def decompress(self, src, src_off, src_size, dst, dst_off, dst_cap, result, opts, sized=False):
        b = self._batch(src, src_off, src_size, dst, dst_off, dst_cap, result)
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_decompress_batch(self.ctx, ctypes.byref(b), ctypes.byref(opts), int(sized)), "decompress_batch")
        finally:
            self._exit(cur)

    # -- stages ----------------------------------------------------------------------------------
    def svb_compress(self, src, src_off, src_size, dst, dst_off, dst_cap, result, size=2, zigzag=True, version=0):
        b = self._batch(src, src_off, src_size, dst, dst_off, dst_cap, result)
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_svb_compress_batch(self.ctx, ctypes.byref(b), size, int(zigzag), version), "svb_compress_batch")
        finally:
            self._exit(cur)

    def svb_decompress(self, src, src_off, src_size, dst, dst_off, dst_cap, result, size=2, zigzag=True, version=0):
        b = self._batch(src, src_off, src_size, dst, dst_off, dst_cap, result)
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_svb_decompress_batch(self.ctx, ctypes.byref(b), size, int(zigzag), version), "svb_decompress_batch")
        finally:
            self._exit(cur)

    def zstd_compress(self, src, src_off, src_size, dst, dst_off, dst_cap, result, key_bytes=None):
        b = self._batch(src, src_off, src_size, dst, dst_off, dst_cap, result)
        kb = key_bytes.data_ptr() if key_bytes is not None else None
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_zstd_compress_batch(self.ctx, ctypes.byref(b), kb), "zstd_compress_batch")
        finally:
            self._exit(cur)

    def zstd_decompress(self, src, src_off, src_size, dst, dst_off, dst_cap, result):
        b = self._batch(src, src_off, src_size, dst, dst_off, dst_cap, result)
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_zstd_decompress_batch(self.ctx, ctypes.byref(b)), "zstd_decompress_batch")
        finally:
            self._exit(cur)

    # -- synthetic workload ----------------------------------------------------------
    def synth_lengths(self, seed, first_read, n_reads):
        out = torch.empty(n_reads, dtype=torch.int32, device=self.device)
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_synth_lengths(self.ctx, seed, first_read, n_reads, out.data_ptr()), "synth_lengths")
        finally:
            self._exit(cur)
        return out

    def synth_signal(self, seed, first_read, dst, off, length):
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_synth_signal(self.ctx, seed, first_read, int(off.numel()), dst.data_ptr(), off.data_ptr(), length.data_ptr()), "synth_signal")
        finally:
            self._exit(cur)

    def synth_u32(self, seed, first_read, dst, off, length):
        cur = self._enter()
        try:
            self._rc(self.L.vbz_gpu_synth_u32(self.ctx, seed, first_read, int(off.numel()), dst.data_ptr(), off.data_ptr(), length.data_ptr()), "synth_u32")
        finally:
            self._exit(cur)

    # -- profiling -------------------------------------------------------------------------------
    def profile(self, enable=True):
        self.L.vbz_gpu_profile_enable(self.ctx, int(enable))

    def profile_reset(self):
        self.L.vbz_gpu_profile_reset(self.ctx)

    def profile_read(self):
        cap = 32
        names = (ctypes.c_char_p * cap)()
        launches = (ctypes.c_uint32 * cap)()
        ms = (ctypes.c_double * cap)()
        k = self.L.vbz_gpu_profile_read(self.ctx, names, launches, ms, cap)
        return {names[i].decode(): (int(launches[i]), float(ms[i])) for i in range(min(k, cap))}

    def decode_paths(self):
        """(frames, batched, walked) of the last decompress launch group: include/vbz_gpu.h, vbz_gpu_decode_paths."""
        b, w = ctypes.c_uint32(0), ctypes.c_uint32(0)
        n = self.L.vbz_gpu_decode_paths(self.ctx, ctypes.byref(b), ctypes.byref(w))
        if n < 0:
            raise RuntimeError("vbz_gpu_decode_paths failed")
        return n, int(b.value), int(w.value)

    def decode_literals_ahead(self):
        """walked frames of the last decompress launch group whose first block's literals were decoded beside the walk: vbz_gpu_decode_literals_ahead."""
        n = self.L.vbz_gpu_decode_literals_ahead(self.ctx)
        if n < 0:
            raise RuntimeError("vbz_gpu_decode_literals_ahead failed")
        return n

    def decode_span_paths(self):
        """(frames, by_spans) of the last decompress launch group on the large-read path: include/vbz_gpu.h, vbz_gpu_decode_span_paths."""
        b = ctypes.c_uint32(0)
        n = self.L.vbz_gpu_decode_span_paths(self.ctx, ctypes.byref(b))
        if n < 0:
            raise RuntimeError("vbz_gpu_decode_span_paths failed")
        return n, int(b.value)

    def synchronize(self):
        self._rc(self.L.vbz_gpu_synchronize(self.ctx), "synchronize")


def layout(sizes, align=64, device="cpu"):
    """Offsets (int64) for slots of the given byte sizes, each aligned to `align`; returns (off, total)."""
    sizes = torch.as_tensor(sizes, dtype=torch.int64)
    padded = (sizes + (align - 1)) // align * align
    off = torch.zeros_like(padded)
    if padded.numel() > 1:
        off[1:] = torch.cumsum(padded, 0)[:-1]
    total = int(padded.sum().item()) + 64
    return off.to(device), total
