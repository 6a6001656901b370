"""CPU tests: pin the oracle (oracle/) against every known answer the reference's tests hold
for the VBZ path (tests/golden/kat.json, tests/golden/fast5_chunks.*) -- SURVEY.md section 8(c)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))
DT = {"int16": np.int16, "uint16": np.uint16, "int32": np.int32, "int8": np.int8, "uint32": np.uint32}
SZ = {"int16": 2, "uint16": 2, "int32": 4, "int8": 1, "uint32": 4}


def _hex(s):
    return bytes.fromhex(s.replace(" ", ""))


@pytest.mark.parametrize("k", KAT["l1"], ids=lambda k: k["cite"][:48])
def test_l1_known_answers(k):
    a = np.array(k["input"], dtype=DT[k["dtype"]])
    want = np.array(k["svb_i8"], np.int8).view(np.uint8)
    got = O.svb_compress(a, SZ[k["dtype"]], k["zigzag"], k["version"])
    assert got.tobytes() == want.tobytes()
    back = O.svb_decompress(got, a.nbytes, SZ[k["dtype"]], k["zigzag"], k["version"])
    assert back.tobytes() == a.tobytes()


@pytest.mark.parametrize("k", KAT["l2"], ids=lambda k: k["cite"][:48])
def test_l2_known_answers(k):
    a = np.array(k["input"], dtype=DT[k["dtype"]])
    o = k["opts"]
    opts = O.options(o["zigzag"], o["size"], o["level"], o["version"])
    want = np.array(k["out_i8"], np.int8).view(np.uint8).tobytes() if "out_i8" in k else _hex(k["out_hex"])
    got = O.compress(a, opts, sized=k["sized"])
    assert got.tobytes() == want
    back = O.decompress(got, a.nbytes, opts, sized=k["sized"])
    assert back.tobytes() == a.tobytes()


@pytest.mark.parametrize("k", KAT["zstd_encoder_pins"], ids=lambda k: k["what"][:40])
def test_zstd_encoder_pins(k):
    if O.lib().vbo_zstd_version() != b"1.4.8":
        pytest.skip("encoder bytes are pinned for libzstd 1.4.8 only")
    a = np.arange(k["arange"][0], k["arange"][1], dtype=DT[k["dtype"]])
    o = k["opts"]
    got = O.compress(a, O.options(o["zigzag"], o["size"], o["level"], o["version"]), sized=k["sized"])
    assert len(got) == k["nbytes"]
    assert got.tobytes() == _hex(k["out_hex"])


@pytest.mark.parametrize("k", KAT["size_pins"], ids=lambda k: "%d-%d" % (k["samples"], k["level"]))
def test_max_size_pins(k):
    assert O.max_compressed_size(k["samples"] * 2, O.options(True, 2, k["level"], 1)) == k["max"]


def _test_read():
    return np.fromfile(os.path.join(GOLDEN, "test_data_read.i16"), dtype="<i2")


@pytest.mark.parametrize("k", KAT["hash_pins"], ids=lambda k: "level%d" % k["level"])
def test_real_read_hash_pins(k):
    a = _test_read()
    assert len(a) == 15643
    if k["level"] and O.lib().vbo_zstd_version() not in (b"1.4.8", b"1.4.9", b"1.5.7"):
        pytest.skip("zstd encoder output is version dependent")
    got = O.compress(a, O.options(True, 2, k["level"], 1))
    assert len(got) == k["nbytes"]
    assert O.decompress(got, a.nbytes, O.options(True, 2, k["level"], 1)).tobytes() == a.tobytes()


def test_real_read_roundtrips_like_reference():
    # vbz/test/vbz_test.cpp:248-288: zigzag only; zigzag+zstd1; "no options" (int_size 1, level 0)
    a = _test_read()
    for opts in (O.options(True, 2, 0, 0), O.options(True, 2, 1, 0), O.options(False, 1, 0, 0)):
        c = O.compress(a, opts)
        assert not isinstance(c, int)
        assert O.decompress(c, a.nbytes, opts).tobytes() == a.tobytes()


def _chunks():
    idx = json.load(open(os.path.join(GOLDEN, "fast5_chunks.json")))
    blob = np.fromfile(os.path.join(GOLDEN, "fast5_chunks.bin"), np.uint8)
    for e in idx:
        yield e, blob[e["chunk_offset"] : e["chunk_offset"] + e["chunk_size"]]


def test_shipped_fast5_decode_pins():
    # python/test/test_vbz_filter.py:57-73: every stored vbz chunk decodes to the gzip file's samples
    n = 0
    for e, chunk in _chunks():
        assert e["v1_identical"]  # v0 and v1 files hold byte-identical chunks for int16
        assert e["filter_v0"][0] == 32020
        cd = e["filter_v0"][1]
        opts = O.options(cd[2] != 0, cd[1], cd[3] if len(cd) > 3 else 1, cd[0])
        out = O.decompress(chunk, e["samples"] * 2, opts, sized=True)
        assert not isinstance(out, int), O.ERRORS.get(out)
        assert hashlib.sha256(out.tobytes()).hexdigest() == e["raw_sha256"]
        # the restated zstd decoder agrees with libzstd on the shipped (older-zstd) frames
        frame = chunk[4:]
        svb = O.zstd_restate_decompress(frame, int(O.zstd_content_size(frame)))
        assert svb is not None
        assert svb.tobytes() == O.zstd_decompress(frame, len(svb)).tobytes()
        n += 1
    assert n == 10


def test_shipped_fast5_svb_encode_pins():
    """Encoder pin on real data: the svb stream the REFERENCE LIBRARY wrote into the shipped files
    (recovered by un-zstd-ing each chunk) must equal the oracle's svb encoding of the decoded samples,
    and re-encoding with libzstd 1.4.8 must give the chunk sizes the survey measured for the reference."""
    sizes = KAT["chunk_sizes_libzstd_1_4_8"]["sizes"]
    for i, (e, chunk) in enumerate(_chunks()):
        frame = chunk[4:]
        svb_ref = O.zstd_decompress(frame, int(O.zstd_content_size(frame)))
        samples = O.svb_decompress(svb_ref, e["samples"] * 2, 2, True, 0)
        assert hashlib.sha256(samples.tobytes()).hexdigest() == e["raw_sha256"]
        svb = O.svb_compress(samples.view(np.int16), 2, True, 0)
        assert svb.tobytes() == svb_ref.tobytes()
        if O.lib().vbo_zstd_version() == b"1.4.8":
            z = O.compress(samples.view(np.int16), O.options(True, 2, 1, 0), sized=True)
            assert len(z) == sizes[i]


def test_reference_style_roundtrips():
    # streamvbyte_test.cpp:98-135 and vbz_test.cpp:13-142: iota(100) +/- zigzag, random values
    rng = np.random.default_rng(7)
    for dt, size in ((np.int8, 1), (np.int16, 2), (np.int32, 4)):
        info = np.iinfo(dt)
        iota = np.arange(100).astype(dt)
        rnd = rng.integers(info.min // 2, info.max // 2, 100000).astype(dt)
        full = rng.integers(info.min, info.max, 10000, endpoint=True).astype(dt)
        for a in (iota, rnd, full):
            for zz in (False, True):
                for ver in (0, 1):
                    c = O.svb_compress(a, size, zz, ver)
                    assert O.svb_decompress(c, a.nbytes, size, zz, ver).tobytes() == a.tobytes()
                    for level in (0, 1):
                        opts = O.options(zz, size, level, ver)
                        z = O.compress(a, opts, sized=True)
                        assert O.decompress(z, a.nbytes, opts, sized=True).tobytes() == a.tobytes()


def test_error_behaviour():
    a = np.arange(10, dtype=np.int16)
    assert O.compress(a, O.options(True, 3, 1, 0)) == 0xFFFFFFFD  # INTEGER_SIZE
    assert O.compress(a, O.options(True, 2, 1, 2)) == 0xFFFFFFFA  # VERSION
    assert O.compress(np.zeros(3, np.uint8), O.options(True, 2, 0, 0)) == 0xFFFFFFFE  # INPUT_SIZE
    c = O.compress(a, O.options(True, 2, 0, 0))
    assert O.decompress(c[:-1], 20, O.options(True, 2, 0, 0)) == 0xFFFFFFFB  # STREAM
    assert O.decompress(c, 22, O.options(True, 2, 0, 0)) == 0xFFFFFFFB
    assert O.decompress(c[:2], 20, O.options(True, 2, 0, 0)) == 0xFFFFFFFE  # shorter than the keys
    assert O.decompress(c, 19, O.options(True, 2, 0, 0)) == 0xFFFFFFFC  # DESTINATION_SIZE
    z = O.compress(a, O.options(True, 2, 1, 0))
    bad = z.copy()
    bad[0] ^= 1
    assert O.decompress(bad, 20, O.options(True, 2, 1, 0)) == 0xFFFFFFFF  # ZSTD


def test_int16_decoder_body_tail_split():
    # sse3.h:498-540 keeps the low 16 bits in the SIMD body, :542-572 un-zigzags in 32 bits in the tail
    n = 64
    keys = np.full(n // 4, 0xAA, np.uint8)  # every code = 2 (three data bytes)
    data = np.zeros(3 * n, np.uint8)
    data[2::3] = 1  # value = 0x10000 : low 16 bits are 0
    out = O.svb_decompress(np.concatenate([keys, data]), 2 * n, 2, True, 0)
    v = out.view(np.int16)
    # groups whose start leaves >= 32 data bytes go through the body: delta 0
    body = sum(1 for g in range(n // 8) if 3 * n - 24 * g >= 32) * 8
    assert (v[:body] == 0).all()
    # tail: (0x10000 >> 1) = 0x8000 added per value, truncated to int16
    want = (np.arange(1, n - body + 1) * 0x8000) & 0xFFFF
    assert (v[body:].view(np.uint16) == want).all()


def test_synthetic_generator_statistics():
    # seed 5, n = 100000: one-byte fraction ~0.989, svb ~1.261 B/sample, ratio ~2.40
    a = O.synth_signal(5, 0, 100000)
    svb = O.svb_compress(a, 2, True, 0)
    assert abs(len(svb) / 100000 - 1.261) < 0.01
    z = O.compress(a, O.options(True, 2, 1, 1))
    assert abs(200000 / len(z) - 2.397) < 0.05
    assert 90000 <= O.synth_read_length(5, 0) <= 110000
    u = O.synth_u32(5, 0, 100000)
    codes = (u > 0xFF).astype(int) + (u > 0xFFFF) + (u > 0xFFFFFF)
    frac = np.bincount(codes, minlength=4) / len(u)
    assert abs(frac[0] - 0.70) < 0.03 and abs(frac[1] - 0.195) < 0.03 and frac[3] > 0.005


def test_hdf5_filter_convention():
    import ctypes

    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    a = O.synth_signal(5, 1, 5000)
    buf = libc.malloc(a.nbytes)
    ctypes.memmove(buf, a.ctypes.data, a.nbytes)
    pbuf = ctypes.c_void_p(buf)
    size = ctypes.c_size_t(a.nbytes)
    cd = (ctypes.c_uint * 4)(1, 2, 1, 1)
    used = O.lib().vbo_filter(0, 4, cd, a.nbytes, ctypes.byref(size), ctypes.byref(pbuf))
    assert used > 0 and used < a.nbytes
    chunk = np.ctypeslib.as_array(ctypes.cast(pbuf, ctypes.POINTER(ctypes.c_uint8)), (used,)).copy()
    assert int(chunk[:4].view("<u4")[0]) == a.nbytes
    size = ctypes.c_size_t(used)
    back = O.lib().vbo_filter(0x100, 4, cd, used, ctypes.byref(size), ctypes.byref(pbuf))
    assert back == a.nbytes
    got = np.ctypeslib.as_array(ctypes.cast(pbuf, ctypes.POINTER(ctypes.c_int16)), (len(a),)).copy()
    assert (got == a).all()
    libc.free(pbuf)
    # cd_nelmts < 3 fails and leaves the buffer alone (vbz_plugin.cpp:109-112)
    p2 = ctypes.c_void_p(libc.malloc(16))
    s2 = ctypes.c_size_t(16)
    assert O.lib().vbo_filter(0, 2, cd, 16, ctypes.byref(s2), ctypes.byref(p2)) == 0
    libc.free(p2)


def test_fuzz_corpus_replay_on_the_oracle():
    """The reference's corpus runner (vbz/fuzzing/vbz_fuzz_runner.cpp, a ctest) replays LLVMFuzzerTestOneInput over the 238
    corpus files and only requires "no crash".  The same replay on the oracle (oracle/vbz_oracle_fuzz.c): every file is
    the committed data (sha256 in the index), every option set x every guessed size runs, and the tally of verdicts is
    pinned so that a change of the oracle's error behaviour shows up here before it shows up in the GPU comparison."""
    import collections
    import hashlib

    index = json.load(open(os.path.join(GOLDEN, "fuzz_corpus.json")))
    blob = open(os.path.join(GOLDEN, "fuzz_corpus.bin"), "rb").read()
    assert len(index) == 238 and sum(e["size"] for e in index) == len(blob)
    tally = collections.Counter()
    calls = 0
    for e in index:
        data = blob[e["offset"] : e["offset"] + e["size"]]
        assert hashlib.sha256(data).hexdigest() == e["sha256"]
        for zz in (True, False):
            for isz in (0, 1, 2, 4):
                for lvl in (0, 1):
                    for ver in (0, 1):
                        G, res = O.fuzz_sweep(data, O.options(zz, isz, lvl, ver))
                        calls += res.size
                        ok = res < O.FIRST_ERROR
                        tally["ok"] += int(ok.sum())
                        for v, c in zip(*np.unique(res[~ok], return_counts=True)):
                            tally[O.ERRORS[int(v)]] += int(c)
    assert calls == 949216
    assert dict(tally) == {"ok": 87750, "VBZ_DESTINATION_SIZE_ERROR": 539540, "VBZ_INPUT_SIZE_ERROR": 512, "VBZ_ZSTD_ERROR": 252044,
                           "VBZ_STREAMVBYTE_STREAM_ERROR": 69322, "VBZ_OUT_OF_MEMORY_ERROR": 48}, dict(tally)


def test_ssse3_svb_of_the_baseline_leg_is_the_scalar_restatement():
    """oracle/vbz_oracle_simd.c (own SSSE3 code, used by bench.py's CPU baseline so that the baseline is the reference's class
    of CPU path) against the scalar restatement the known answers pin: byte-identical streams, identical samples and errors."""
    if not O.simd_available():
        pytest.skip("no SSSE3 on this host")
    rng = np.random.default_rng(31)
    reads = [O.synth_signal(5, i, n) for i, n in enumerate([1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 1000, 4097, 100003])]
    reads.append(rng.integers(-32768, 32767, 30001, endpoint=True).astype(np.int16))   # full range, wraps: mostly two-byte codes
    reads.append(np.array([32767, -32768] * 500, np.int16))
    reads.append(np.zeros(5000, np.int16))
    reads.append(np.fromfile(os.path.join(GOLDEN, "test_data_read.i16"), dtype="<i2"))
    for a in reads:
        want = O.svb_compress(a, 2, True, 0)
        got = O.i16zz_compress_simd(a)
        assert got.tobytes() == want.tobytes(), len(a)
        back = O.i16zz_decompress_simd(want, a.nbytes)
        assert not isinstance(back, int) and back.tobytes() == a.tobytes(), len(a)
        for bad, nb in ((want[:-1], a.nbytes), (np.concatenate([want, np.zeros(1, np.uint8)]), a.nbytes), (want, a.nbytes + 2)):
            w = O.svb_decompress(bad, nb, 2, True, 0)
            g = O.i16zz_decompress_simd(bad, nb)
            if g == O.SIMD_DECLINED:
                continue   # (a code above 1 appeared where the stream was cut: the scalar function decides)
            assert (isinstance(w, int) and g == w) or (not isinstance(w, int) and g.tobytes() == w.tobytes()), (len(a), nb)
    # a stream with a code above 1 is declined, not decoded
    st = O.svb_compress(reads[15], 2, True, 0).copy()
    st[3] |= 0x80
    assert O.i16zz_decompress_simd(st, reads[15].nbytes) in (O.SIMD_DECLINED, 0xFFFFFFFB)
    # and the whole path through vbo_compress / vbo_decompress with the switch on gives the same frames
    a = reads[17]
    oo = O.options(True, 2, 1, 1)
    f0 = O.compress(a, oo)
    O.lib().vbo_use_simd_svb(1)
    try:
        f1 = O.compress(a, oo)
        b1 = O.decompress(f0, a.nbytes, oo)
    finally:
        O.lib().vbo_use_simd_svb(0)
    assert f1.tobytes() == f0.tobytes() and b1.tobytes() == a.tobytes()
