"""A large batch is coded as two halves on two streams inside ONE call (vbz_api.hip: split_batch).  A read's bytes do not depend on its
neighbours, so the split call must write exactly what the unsplit call writes -- frames, sizes, error verdicts -- and decode the same.
The reference is a pure function of (input, options) per buffer (vbz/vbz.cpp:116-208); whatever the library does with a batch's shape
must not show in a read's output."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _codec(split_min, stagger=1):
    from vbz_compression_amd import batch

    old = {k: os.environ.get(k) for k in ("VBZ_HIP_SPLIT_MIN", "VBZ_HIP_SPLIT_STAGGER")}
    os.environ["VBZ_HIP_SPLIT_MIN"] = str(split_min)
    os.environ["VBZ_HIP_SPLIT_STAGGER"] = str(stagger)
    try:
        return batch.GpuCodec(0)   # (the knobs are read when the context is created)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _make(c, first, n, opts, sized, long_reads=False, bad=()):
    import torch
    from vbz_compression_amd import batch

    dev = c.device
    L = c.L
    with torch.cuda.stream(c.stream):
        lens = (c.synth_lengths(5, first, n) // 8) & ~1  # ~12 k samples per read (even: the 4-byte options want whole values)
        lens[3] = 0                                      # an empty read in the lower half, one in the upper
        lens[n - 2] = 0
        lens[5] = 2
        lens[n // 2] = 6
        if long_reads:                                   # per-read routing beside the split: one long read in each half
            lens[11] = 700000
            lens[n - 5] = 900002
        sizes = lens.to(torch.int64) * 2
        for i in bad:                                    # an odd byte count: VBZ_INPUT_SIZE_ERROR for that read only
            sizes[i] += 1
        off, total = batch.layout(sizes.cpu(), 64)
        caps = torch.tensor([L.vbz_max_compressed_size(int(s) & ~1, ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
        coff, ctotal = batch.layout(caps, 64)
        raw = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
        off = off.to(dev)
        c.synth_signal(5, first, raw, off, lens)
    torch.cuda.synchronize()
    return dict(n=n, raw=raw, off=off, size=sizes.to(torch.int32).to(dev), coff=coff.to(dev), cap=caps.to(torch.int32).to(dev), ctotal=ctotal, total=total,
                sized=sized)


def _roundtrip(c, B, opts):
    import torch

    dev = c.device
    n = B["n"]
    comp = torch.zeros(B["ctotal"] + 64, dtype=torch.uint8, device=dev)
    csize = torch.full((n,), -8, dtype=torch.int32, device=dev)
    back = torch.zeros(B["total"] + 64, dtype=torch.uint8, device=dev)
    res = torch.full((n,), -8, dtype=torch.int32, device=dev)
    with torch.cuda.stream(c.stream):
        c.compress(B["raw"], B["off"], B["size"], comp, B["coff"], B["cap"], csize, opts, sized=B["sized"])
        # (a read that failed to compress has an error code for a size: give the decoder an empty source for it)
        ok = (csize >= 0) | (csize < -8)
        dsize = torch.where(ok, csize, torch.zeros_like(csize))
        c.decompress(comp, B["coff"], dsize, back, B["off"], B["size"], res, opts, sized=B["sized"])
    torch.cuda.synchronize()
    cs = [int(x) & 0xFFFFFFFF for x in csize.cpu().tolist()]
    host = comp.cpu().numpy()
    digests = []
    for z, o in zip(cs, B["coff"].cpu().tolist()):
        digests.append(z if z >= 0xFFFFFFF8 else hashlib.sha256(host[o : o + z].tobytes()).hexdigest())
    return cs, digests, res.cpu(), back, ok.cpu()


@pytest.mark.parametrize("sized", [False, True])
@pytest.mark.parametrize("stagger", [0, 1])
def test_split_call_writes_what_the_unsplit_call_writes(sized, stagger):
    import torch

    plain = _codec(0)
    split = _codec(64, stagger)
    opts = plain.options(True, 2, 1, 1)
    n = 301                                              # an odd count: the halves differ
    B = _make(plain, 1000, n, opts, sized, bad=(17, n - 9))
    cs0, dg0, res0, back0, ok0 = _roundtrip(plain, B, opts)
    cs1, dg1, res1, back1, ok1 = _roundtrip(split, B, opts)
    assert cs0 == cs1                                    # sizes and error codes, read by read
    assert dg0 == dg1                                    # sha256 of every frame
    assert torch.equal(res0, res1)
    good = [i for i in range(n) if i not in (17, n - 9)]
    assert all(int(res1[i]) == int(B["size"][i]) for i in good)
    assert all((cs1[i] & 0xFFFFFFFF) >= 0xFFFFFFF8 for i in (17, n - 9))
    raw = B["raw"].cpu().numpy()
    b1 = back1.cpu().numpy()
    for i in good:
        o, z = int(B["off"][i]), int(B["size"][i])
        assert raw[o : o + z].tobytes() == b1[o : o + z].tobytes()
    assert split.decode_paths()[0] == n                  # both halves' frames are accounted for
    plain.close()
    split.close()


def test_split_call_beside_routed_long_reads_and_other_options():
    """Per-read routing (long reads among short ones take the large-read path on a third stream) and the split in one call; and the
    generic instantiations (uint32 without zig-zag at level 3, int8 with zig-zag) through split calls."""
    import torch

    plain = _codec(0)
    split = _codec(64)
    for (zz, size, level, ver), long_reads in (((True, 2, 1, 1), True), ((False, 4, 3, 0), False), ((True, 1, 1, 0), False), ((True, 2, 1, 0), False)):
        opts = plain.options(zz, size, level, ver)
        B = _make(plain, 7000, 200, opts, True, long_reads=long_reads)
        cs0, dg0, res0, back0, _ = _roundtrip(plain, B, opts)
        cs1, dg1, res1, back1, _ = _roundtrip(split, B, opts)
        assert cs0 == cs1 and dg0 == dg1, (zz, size, level, ver)
        assert torch.equal(res0, res1) and bool((res1 == B["size"].cpu()).all())
        assert torch.equal(back1[: B["total"]], B["raw"][: B["total"]])
    plain.close()
    split.close()


def test_split_is_the_default_for_large_batches_and_profile_labels_add_up():
    """16 384 reads and more are split by default; the library's per-label launch counts then show two launches per call."""
    import torch
    from vbz_compression_amd import batch

    c = batch.GpuCodec(0)
    opts = c.options(True, 2, 1, 1)
    n = 16384
    with torch.cuda.stream(c.stream):
        lens = c.synth_lengths(5, 0, n) // 32          # ~3 k samples per read
        sizes = lens.to(torch.int64) * 2
        off, total = batch.layout(sizes.cpu(), 64)
        caps = torch.tensor(np.array([c.L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()]), dtype=torch.int64)
        coff, ctotal = batch.layout(caps, 64)
        raw = torch.zeros(total, dtype=torch.uint8, device=c.device)
        off = off.to(c.device)
        c.synth_signal(5, 0, raw, off, lens)
        comp = torch.zeros(ctotal, dtype=torch.uint8, device=c.device)
        csize = torch.zeros(n, dtype=torch.int32, device=c.device)
        back = torch.zeros(total, dtype=torch.uint8, device=c.device)
        res = torch.zeros(n, dtype=torch.int32, device=c.device)
        size32, cap32, coffd = sizes.to(torch.int32).to(c.device), caps.to(torch.int32).to(c.device), coff.to(c.device)
        c.profile_reset()
        c.profile(True)
        c.compress(raw, off, size32, comp, coffd, cap32, csize, opts)
        c.decompress(comp, coffd, csize, back, off, size32, res, opts)
    torch.cuda.synchronize()
    c.profile(False)
    prof = c.profile_read()
    assert bool((res == size32).all()) and torch.equal(raw, back)
    for label in ("svb_encode", "zstd_encode", "zstd_decode", "svb_decode"):
        assert prof[label][0] == 2, (label, prof[label])
    assert prof["plan_scratch"][0] == 2                  # ... and ONE scratch plan per direction
    assert c.decode_paths()[0] == n
    c.close()
