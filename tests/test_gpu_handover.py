"""The svb encoder's hand-over to the entropy stage (svb_kernels.hip CNT): for int16 zig-zag reads on their way to the entropy stage
the encoder counts, while the data bytes pass through LDS, exactly the sample the planning launch's region_histogram would take (one
kilobyte in four and the unaligned ends; the other bytes too, apart, for reads so short that the region may be counted exactly) and
leaves the histogram in the read's plan.  Held here, read by read, to a numpy statement of that sample (zstd_encode.hip; the bytes
are the data bytes of the svb stream of vbz/v0/vbz_streamvbyte_impl_sse3.h:406-466).  Runs against the experiments build of the
library (lib/libvbz_hip_x.so), which exports the svb half of vbz_gpu_compress_batch on its own."""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu
PLAN_WORDS = 932   # sizeof(EncPlan) / 4 (vbz_kernels.h): reg[2] x (8 + 256 + 34), 4, cp[64], tok_done / tok_nrec / tok_lit / hist_mode, state, pad[7], histB[256]
REG1_CTABLE = 298 + 8
TOK_DONE = 2 * 298 + 4 + 64
HISTB = TOK_DONE + 4 + 8


def sample_statement(data, K):
    """(histogram of the bytes region_histogram's sample counts, histogram of the others) for the data bytes of a stream whose scratch
    slot starts 16-byte aligned: the unaligned head and tail, and of the aligned 16-byte chunks between them every fourth stripe of 64."""
    S = len(data)
    h = min((16 - K % 16) % 16, S)
    nvec = (S - h) >> 4
    tail0 = h + 16 * nvec
    p = np.arange(S)
    c = (p - h) >> 4
    sampled = (p < h) | (p >= tail0) | (((c >> 6) & 3) == 0)
    return np.bincount(data[sampled], minlength=256), np.bincount(data[~sampled], minlength=256)


def test_svb_encoder_handover_matches_the_statement():
    from vbz_compression_amd import _lib, batch

    keep = _lib._lib, _lib.LIB_PATH
    try:
        _lib._lib, _lib.LIB_PATH = None, _lib.EXPERIMENTS_LIB_PATH
        c = batch.GpuCodec(0)
        L = c.L
        assert b"+experiments" in L.vbz_gpu_version()
        L.vbz_gpu_x_plan_bytes.restype = ctypes.c_size_t
        assert L.vbz_gpu_x_plan_bytes() == 4 * PLAN_WORDS
        L.vbz_gpu_x_svb_handover.restype = ctypes.c_int
        L.vbz_gpu_x_svb_handover.argtypes = [ctypes.c_void_p, ctypes.POINTER(_lib.GpuBatch), ctypes.c_void_p]
        rng = np.random.default_rng(3)
        t = O.synth_signal(5, 99, 15643)
        reads = [O.synth_signal(5, 9000 + i, n) for i, n in enumerate([0, 5, 1639, 1640, 1641, 2048, 2049, 4095, 4096, 6000, 8191, 8192, 8193, 12000, 32767, 32768, 32769,
                                                                        50000, 65536, 100000, 100003, 110000, 250000, 524284, 524292])]
        reads += [np.resize(t, 100000), rng.integers(-32768, 32767, 60000, dtype=np.int16), np.full(40000, 7, np.int16),
                  np.repeat(rng.integers(-2000, 2000, 400).astype(np.int16), 250), np.arange(0, 30000, dtype=np.int16)]
        blocks = []
        for run in (10, 11, 12, 13, 40, 511, 512, 513, 959, 960, 961, 2047, 2048, 2049):
            blocks += [np.zeros(4 * run, np.int64), rng.integers(-3000, 3000, 8)]
        reads.append(np.cumsum(np.concatenate(blocks * 3)).astype(np.int16))
        opts = _lib.CompressionOptions(True, 2, 1, 1)
        sizes = [a.nbytes for a in reads]
        off, total = batch.layout(sizes, 64)
        arena = np.zeros(total + 64, np.uint8)
        for a, o in zip(reads, off.tolist()):
            arena[o:o + a.nbytes] = a.view(np.uint8)
        caps = [L.vbz_max_compressed_size(s, ctypes.byref(opts)) for s in sizes]
        doff, dtotal = batch.layout([x + 32 for x in caps], 64)
        dev = c.device
        src = torch.from_numpy(arena).to(dev)
        dst = torch.zeros(dtotal + 64, dtype=torch.uint8, device=dev)
        plans = torch.zeros(len(reads) * PLAN_WORDS, dtype=torch.int32, device=dev)
        res = torch.full((len(reads),), -8, dtype=torch.int32, device=dev)
        b = c._batch(src, off.to(dev), torch.tensor(sizes, dtype=torch.int32, device=dev), dst, doff.to(dev), torch.tensor(caps, dtype=torch.int32, device=dev), res)
        cur = c._enter()
        try:
            assert L.vbz_gpu_x_svb_handover(c.ctx, ctypes.byref(b), plans.data_ptr()) == 0
        finally:
            c._exit(cur)
        torch.cuda.synchronize()
        host = dst.cpu().numpy()
        P = plans.cpu().numpy().view(np.uint32).reshape(len(reads), PLAN_WORDS)
        N = [int(x) & 0xFFFFFFFF for x in res.cpu().tolist()]
        base = dst.data_ptr()
        tokenised = 0
        for i, a in enumerate(reads):
            n = len(a)
            want = O.svb_compress(a, 2, True, 0)
            K = (n + 3) // 4
            assert N[i] == len(want), (n, N[i], len(want))
            o = int(doff[i])
            got = host[o:o + N[i]]
            assert got[K:].tobytes() == want[K:].tobytes(), ("data bytes", n)
            hmode = int(P[i, TOK_DONE + 3])
            assert got[:K].tobytes() == want[:K].tobytes(), ("control bytes", n)
            assert hmode == (0 if n < 1640 else (1 if n >= 32768 else 2)), (n, hmode)
            if hmode == 0:
                continue
            tokenised += 1
            hs, hr = sample_statement(want[K:], K)
            assert (P[i, REG1_CTABLE:REG1_CTABLE + 256] == hs).all(), ("sampled histogram", n)
            if hmode == 2:
                assert (P[i, HISTB:HISTB + 256] == hr).all(), ("the other bytes' histogram", n)
        assert tokenised >= 25
        c.close()
    finally:
        _lib._lib, _lib.LIB_PATH = keep
