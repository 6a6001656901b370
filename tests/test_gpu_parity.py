"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of libvbz_hip.so, against the
CPU oracle (oracle/) on the same seeded inputs, against the committed golden fixtures, and through
size-independent properties at full size.  Bit-exact everywhere (integer / byte work)."""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))
DT = {"int16": np.int16, "uint16": np.uint16, "int32": np.int32, "int8": np.int8, "uint32": np.uint32}
SZ = {"int16": 2, "uint16": 2, "int32": 4, "int8": 1, "uint32": 4}
LENGTHS = [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 63, 64, 65, 255, 256, 257, 511, 2047, 2048, 2049, 4095, 4096, 4097, 6000, 20000, 100003]


def _same(got, want):
    if isinstance(want, int) or isinstance(got, int):
        return isinstance(got, int) and isinstance(want, int) and got == want
    return got.tobytes() == want.tobytes()


def _signals(rng):
    reads = [O.synth_signal(5, i, n) for i, n in enumerate(LENGTHS)]
    reads.append(rng.integers(-32768, 32767, 30001, endpoint=True).astype(np.int16))   # full range, wraps
    reads.append(np.array([32767, -32768] * 500, np.int16))                              # wrap-around quirk
    reads.append(np.zeros(5000, np.int16))
    reads.append(np.arange(0, 1000, dtype=np.int16))
    reads.append(np.fromfile(os.path.join(GOLDEN, "test_data_read.i16"), dtype="<i2"))
    return reads


# ------------------------------------------------------------------------------------------------
# stage 1: delta zig-zag streamvbyte
# ------------------------------------------------------------------------------------------------
def test_svb_int16_zigzag_encode_bit_exact():
    import gpu_util as G

    reads = _signals(np.random.default_rng(11))
    got = G.svb_compress(reads, 2, True)
    for a, g in zip(reads, got):
        want = O.svb_compress(a, 2, True, 0)
        assert _same(g, want), len(a)


def test_svb_int16_zigzag_whole_tile_pairs_and_their_fallback():
    """The int16 zig-zag encoder codes pairs of whole tiles (4096 samples) in a loop of its own (svb_kernels.hip: I16Pairs) whose stage
    buffers hold 3/2 bytes per value; a pair that needs more ends that loop and the tile-by-tile loop codes the rest.  Quiet signal
    (one byte per value), noise (two), wrap-around deltas, and reads that change from one to the other inside and at the edges of a
    pair -- through the svb stage alone (level 0 of the product: byte-identical to the reference, sse3.h:406-466) and through the
    whole compress path with its histogram hand-over and probe, unaligned destinations included."""
    import gpu_util as G
    from vbz_compression_amd import _lib

    rng = np.random.default_rng(606)
    T = 2048

    def quiet(n):
        return (300 + np.cumsum(rng.integers(-40, 41, n))).astype(np.int16)

    def noise(n):
        return rng.integers(-32768, 32768, n).astype(np.int16)

    def border(n):   # values whose zig-zag sits at the one-byte / two-byte border: 127, -128 | 128, -129
        return np.cumsum(rng.choice(np.array([127, -128, 128, -129, 0, 1, -1]), n)).astype(np.int16)

    reads = [quiet(20 * T + 77), noise(9 * T + 5), border(13 * T), quiet(2 * T), quiet(3 * T - 1), quiet(4 * T + 1),
             np.concatenate([quiet(5 * T), noise(3 * T), quiet(6 * T + 9)]),          # the loop ends on the noisy pair, the rest goes tile by tile
             np.concatenate([quiet(T), noise(T // 2), quiet(7 * T + T // 2)]),        # a pair that is half noise still fits
             np.concatenate([noise(2 * T), quiet(8 * T)]),
             np.concatenate([quiet(6 * T), border(6 * T + 3), noise(100)]),
             np.full(10 * T, -32768, np.int16), np.tile(np.array([32767, -32768], np.int16), 5 * T),
             np.arange(0, 12 * T, dtype=np.int64).astype(np.int16)]
    got = G.svb_compress(reads, 2, True)
    for a, g in zip(reads, got):
        assert _same(g, O.svb_compress(a, 2, True, 0)), len(a)
    for align in (64, 1):   # (1: data sections that start anywhere -- the partly owned first 16 bytes)
        old = G.DST_ALIGN
        G.DST_ALIGN = align
        try:
            for level in (0, 1):
                opts = _lib.CompressionOptions(True, 2, level, 1)
                frames = G.compress(reads, opts, sized=True)
                for a, f in zip(reads, frames):
                    assert not isinstance(f, int), f
                    if level == 0:
                        assert _same(f, O.compress(a, O.options(True, 2, 0, 1), sized=True)), len(a)
                    else:
                        assert O.decompress(f, a.nbytes, O.options(True, 2, 1, 1), sized=True).tobytes() == a.tobytes(), len(a)
        finally:
            G.DST_ALIGN = old


def test_svb_int16_zigzag_decoder_pipeline_and_its_hand_over():
    """The int16 zig-zag decoder decodes pairs of whole tiles as a pipeline (svb_kernels.hip: I16DecPairs: the pair after the one being
    decoded is planned and its bytes requested a trip ahead) and hands over to the tile loop -- which decides every verdict -- where a
    pair is not of its kind: codes of 3 / 4 bytes (the reference's SIMD body keeps their low 16 bits, sse3.h:510-514), more bytes than a
    stage buffer holds (noise), a stream shorter or longer than its control bytes announce.  Samples or the oracle's error, either way."""
    import gpu_util as G

    rng = np.random.default_rng(607)
    T = 2048

    def quiet(n):
        return (300 + np.cumsum(rng.integers(-40, 41, n))).astype(np.int16)

    def noise(n):
        return rng.integers(-32768, 32768, n).astype(np.int16)

    cases = []
    for a in (quiet(20 * T + 77), noise(9 * T + 5), quiet(2 * T), quiet(4 * T), quiet(4 * T + 1), quiet(6 * T - 1),
              np.concatenate([quiet(5 * T), noise(3 * T), quiet(6 * T + 9)]), np.concatenate([noise(2 * T), quiet(8 * T)]),
              np.concatenate([quiet(T), noise(T // 2), quiet(7 * T + T // 2)])):
        st = O.svb_compress(a, 2, True, 0)
        cases.append((st, a.nbytes))
        cases += [(st[:-1], a.nbytes), (st[:-3000], a.nbytes), (np.concatenate([st, np.zeros(2, np.uint8)]), a.nbytes), (st, a.nbytes - 2), (st, a.nbytes + 2)]
    a = quiet(16 * T)
    st = O.svb_compress(a, 2, True, 0)
    K = (len(a) + 3) // 4
    for value in (0, 7, 2 * T - 1, 2 * T, 4 * T + 3, 9 * T + 1000, 16 * T - 1):   # one wide code somewhere: first pair, a boundary, the last pair
        for code in (2, 3):
            w = st.copy()
            w[value >> 2] |= code << (2 * (value & 3))
            grown = np.concatenate([w, rng.integers(0, 256, 3, dtype=np.uint8)])   # (with the bytes the wider code announces)
            cases += [(w, a.nbytes), (grown[: len(w) + code - (1 if (st[value >> 2] >> (2 * (value & 3))) & 1 else 0)], a.nbytes)]
    got = G.svb_decompress([c[0] for c in cases], [c[1] for c in cases], 2, True)
    for (stream, nbytes), g in zip(cases, got):
        want = O.svb_decompress(stream, nbytes, 2, True, 0)
        assert _same(g, want), (len(stream), nbytes)


def test_svb_int16_zigzag_decode_bit_exact():
    import gpu_util as G

    reads = _signals(np.random.default_rng(12))
    streams = [O.svb_compress(a, 2, True, 0) for a in reads]
    got = G.svb_decompress(streams, [a.nbytes for a in reads], 2, True)
    for a, g in zip(reads, got):
        assert _same(g, np.frombuffer(a.tobytes(), np.uint8)), len(a)


@pytest.mark.parametrize("k", KAT["l1"], ids=lambda k: k["cite"][:40])
def test_svb_known_answers(k):
    import gpu_util as G

    a = np.array(k["input"], dtype=DT[k["dtype"]])
    want = np.array(k["svb_i8"], np.int8).view(np.uint8)
    got = G.svb_compress([a], SZ[k["dtype"]], k["zigzag"], 0)[0]
    assert _same(got, want)
    back = G.svb_decompress([want], [a.nbytes], SZ[k["dtype"]], k["zigzag"], 0)[0]
    assert back.tobytes() == a.tobytes()


@pytest.mark.parametrize("size,zigzag", [(4, False), (4, True), (2, False), (1, False), (1, True)])
def test_svb_generic_bit_exact(size, zigzag):
    import gpu_util as G

    rng = np.random.default_rng(100 + size)
    dt = {1: np.int8, 2: np.int16, 4: np.int32}[size]
    info = np.iinfo(dt)
    bufs = []
    for i, n in enumerate(LENGTHS):
        if size == 4 and not zigzag:
            bufs.append(O.synth_u32(5, i, n).view(np.int32))   # config 4 values: all four code lengths
        else:
            bufs.append(rng.integers(info.min // 2, info.max // 2, n).astype(dt))
    bufs.append(rng.integers(info.min, info.max, 50001, endpoint=True).astype(dt))
    got = G.svb_compress(bufs, size, zigzag)
    streams = []
    for a, g in zip(bufs, got):
        want = O.svb_compress(a, size, zigzag, 0)
        assert _same(g, want), (size, zigzag, len(a))
        streams.append(want)
    back = G.svb_decompress(streams, [a.nbytes for a in bufs], size, zigzag)
    for a, g in zip(bufs, back):
        assert _same(g, np.frombuffer(a.tobytes(), np.uint8)), (size, zigzag, len(a))


@pytest.mark.parametrize("zigzag", [False, True])
def test_svb_v1_nibble_codec_bit_exact(zigzag):
    """v1 codes 1-byte integers with the nibble codec (vbz/v1/vbz_streamvbyte_impl.h:20-216): bit-exact both ways,
    all four codes, every ragged length, and the oracle's verdict on malformed streams."""
    import gpu_util as G

    rng = np.random.default_rng(7)
    bufs = []
    for n in LENGTHS:
        bufs.append(rng.integers(-128, 127, n, endpoint=True).astype(np.int8))      # mostly two and four nibbles
    bufs.append(np.zeros(5000, np.int8))                                            # code 0 only: no data bytes at all
    bufs.append(rng.integers(-3, 3, 70001, endpoint=True).astype(np.int8))          # zeros and single nibbles, odd totals
    bufs.append(np.cumsum(rng.integers(-1, 1, 30000, endpoint=True)).astype(np.int8))
    got = G.svb_compress(bufs, 1, zigzag, 1)
    streams = []
    for a, g in zip(bufs, got):
        want = O.svb_compress(a, 1, zigzag, 1)
        assert _same(g, want), (zigzag, len(a))
        streams.append(want if not isinstance(want, int) else np.zeros(0, np.uint8))
    back = G.svb_decompress(streams, [a.nbytes for a in bufs], 1, zigzag, 1)
    for a, g in zip(bufs, back):
        assert _same(g, np.frombuffer(a.tobytes(), np.uint8)), (zigzag, len(a))
    # malformed: truncated, padded, wrong count
    s0 = streams[12]
    n0 = bufs[12].nbytes
    bad = [s0[:-1], np.concatenate([s0, np.zeros(1, np.uint8)]), s0, s0, s0[: len(s0) // 2]]
    sizes = [n0, n0, n0 + 1, n0 - 1, n0]
    got = G.svb_decompress(bad, sizes, 1, zigzag, 1)
    for s_, n_, g in zip(bad, sizes, got):
        want = O.svb_decompress(s_, n_, 1, zigzag, 1)
        assert _same(g, want), (len(s_), n_, g if isinstance(g, int) else "data", want if isinstance(want, int) else "data")


def test_svb_decode_errors_match_oracle():
    import gpu_util as G

    a = O.synth_signal(5, 3, 5000)
    s = O.svb_compress(a, 2, True, 0)
    u = O.synth_u32(5, 3, 3000)
    su = O.svb_compress(u, 4, False, 0)
    cases = [
        (s[:-1], a.nbytes, 2, True), (np.concatenate([s, np.zeros(1, np.uint8)]), a.nbytes, 2, True),
        (s[:100], a.nbytes, 2, True), (s, a.nbytes - 2, 2, True), (s, a.nbytes + 2, 2, True), (s, a.nbytes + 1, 2, True),
        (s[:0], 0, 2, True), (s, 0, 2, True),
        (su[:-1], u.nbytes, 4, False), (su, u.nbytes - 4, 4, False), (su[:10], u.nbytes, 4, False), (su, 0, 4, False),
        (su[:0], 0, 4, False), (su, u.nbytes + 3, 4, False),
    ]
    for stream, nbytes, size, zz in cases:
        want = O.svb_decompress(stream, nbytes, size, zz, 0)
        got = G.svb_decompress([stream], [nbytes], size, zz)[0]
        assert _same(got, want), (len(stream), nbytes, size, zz, got, want)


def test_svb_int16_decoder_body_tail_split():
    # codes 2/3 in an int16 stream: the reference's SIMD body truncates to 16 bits, its tail does not
    import gpu_util as G

    rng = np.random.default_rng(5)
    for n in (64, 8, 40, 333, 5000):
        keys = rng.integers(0, 256, (n + 3) // 4, dtype=np.uint8)
        if n % 4:
            keys[-1] &= (1 << (2 * (n % 4))) - 1
        codes = np.array([(keys[i >> 2] >> (2 * (i & 3))) & 3 for i in range(n)])
        data = rng.integers(0, 256, int((codes + 1).sum()), dtype=np.uint8)
        stream = np.concatenate([keys, data])
        want = O.svb_decompress(stream, 2 * n, 2, True, 0)
        got = G.svb_decompress([stream], [2 * n], 2, True)[0]
        assert _same(got, want), n


def test_wave_svb_decoder_behind_the_entropy_stage():
    """The int16 zig-zag stream behind the entropy stage has two decoders: svb_decode_kernel with its block path (lanes own
    32 consecutive values, packed 16-bit arithmetic), and -- VBZ_HIP_FUSE_SVB=1 -- the wavefront that decoded the frame
    (svb_wave.h).  Same verdicts from both: foreign and malformed svb streams (codes 2 / 3, wrong lengths, short and long
    streams, every length class of the block paths) packed into frames by libzstd -- what a foreign writer could store --
    must give the oracle's samples or the oracle's error."""
    import gpu_util as G
    from vbz_compression_amd import _lib, batch

    rng = np.random.default_rng(77)
    cases = []   # (svb stream, claimed output bytes)
    for n in (0, 1, 7, 8, 511, 512, 513, 4095, 4096, 4097, 8191, 8192 + 512, 12288, 40000, 100003):
        a = O.synth_signal(5, n, n)
        st = O.svb_compress(a, 2, True, 0)
        cases.append((st, a.nbytes))
        if n:
            cases += [(st[:-1], a.nbytes), (np.concatenate([st, np.zeros(1, np.uint8)]), a.nbytes), (st, a.nbytes - 2), (st, a.nbytes + 2), (st, a.nbytes + 1)]
    for n in (64, 8, 40, 333, 5000, 4096, 9000, 30000):   # random control bytes: codes 2 / 3, the reference's body / tail split
        keys = rng.integers(0, 256, (n + 3) // 4, dtype=np.uint8)
        if n % 4:
            keys[-1] &= (1 << (2 * (n % 4))) - 1
        codes = np.array([(keys[i >> 2] >> (2 * (i & 3))) & 3 for i in range(n)])
        data = rng.integers(0, 256, int((codes + 1).sum()), dtype=np.uint8)
        cases.append((np.concatenate([keys, data]), 2 * n))
    a = O.synth_signal(5, 99, 30000)   # wide codes in the middle of an ordinary stream: the pipeline hands over to the tile loop
    st = O.svb_compress(a, 2, True, 0).copy()
    K = (len(a) + 3) // 4
    wide = st.copy()
    wide[5000] |= 0x80      # value 20003 gets code 2 or 3: the stream is then one or two bytes short
    cases.append((wide, a.nbytes))
    cases.append((np.concatenate([wide, np.zeros(2, np.uint8)]), a.nbytes))
    cases.append((np.concatenate([wide, np.zeros(1, np.uint8)]), a.nbytes))
    oo = O.options(True, 2, 1, 1)
    frames = [O.zstd_compress(st, 1) for st, _ in cases]
    want = [O.decompress(f, nb, oo) for f, (_, nb) in zip(frames, cases)]
    assert sum(1 for w in want if isinstance(w, int)) >= 40 and sum(1 for w in want if not isinstance(w, int)) >= 20
    keep = G._codec, _lib._lib, _lib.LIB_PATH
    try:
        for fuse in ("0", "1"):   # the separate svb_decode launch (the product), and the frame's own wavefront (experiments build)
            if fuse == "1":
                _lib._lib, _lib.LIB_PATH = None, _lib.EXPERIMENTS_LIB_PATH
            os.environ["VBZ_HIP_FUSE_SVB"] = fuse
            try:
                G._codec = batch.GpuCodec(0)
                assert (b"+experiments" in G._codec.L.vbz_gpu_version()) == (fuse == "1")
            finally:
                del os.environ["VBZ_HIP_FUSE_SVB"]
            got = G.decompress(frames, [nb for _, nb in cases], _lib.CompressionOptions(True, 2, 1, 1))
            for i, (w, g) in enumerate(zip(want, got)):
                assert _same(g, w), (fuse, i, len(cases[i][0]), cases[i][1], g if isinstance(g, int) else "samples", w if isinstance(w, int) else "samples")
    finally:
        G._codec, _lib._lib, _lib.LIB_PATH = keep


# ------------------------------------------------------------------------------------------------
# stage 2: zstd-format entropy stage
# ------------------------------------------------------------------------------------------------
def _lib_opts(zz, size, level, version):
    from vbz_compression_amd import _lib

    return _lib.CompressionOptions(zz, size, level, version)


def _svb_streams():
    out = []
    for i, n in enumerate([0, 1, 5, 40, 200, 700, 1000, 3000, 9000, 30000, 100000, 110000, 400000]):
        out.append(O.svb_compress(O.synth_signal(5, i, n), 2, True, 0))
    out.append(O.svb_compress(np.arange(0, 1000, dtype=np.int16), 2, True, 0))
    out.append(O.svb_compress(O.synth_u32(5, 1, 200000), 4, False, 0))
    return out


def test_zstd_decode_libzstd_frames():
    import gpu_util as G

    if O.lib().vbo_zstd_version() is None:
        pytest.skip("no libzstd on this box")
    rng = np.random.default_rng(21)
    contents = _svb_streams()
    contents.append(rng.integers(0, 256, 70000, dtype=np.uint8))                     # incompressible -> raw blocks
    contents.append(np.zeros(300000, np.uint8))                                       # RLE blocks
    text = np.frombuffer((b"the quick brown fox jumps over the lazy dog " * 4000), np.uint8)
    contents.append(text.copy())                                                      # long matches, repeat offsets
    contents.append(np.minimum(rng.geometric(0.3, 150000), 255).astype(np.uint8))
    # matches that overlap their own output, at every kind of distance the byte movers tell apart
    for period in (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 63, 64, 79, 80, 81, 100, 255, 256, 257, 1000, 5000):
        unit = rng.integers(0, 256, period, dtype=np.uint8)
        rep = np.tile(unit, 40000 // period + 2)[: 40000 + period % 7]
        contents.append(np.concatenate([rng.integers(0, 256, 37 + period % 11, dtype=np.uint8), rep, unit[: period // 2], rep[: 300 + period]]))
    frames, want = [], []
    for c in contents:
        for level in (1, 3, 9):
            frames.append(O.zstd_compress(c, level))
            want.append(c)
    got = G.zstd_decompress(frames, [len(w) for w in want])
    for f, w, g in zip(frames, want, got):
        assert _same(g, np.ascontiguousarray(w)), (len(w), len(f))


def test_zstd_decode_foreign_zero_run_frames():
    """Frames libzstd writes for data that looks like a control-byte region (long zero runs between short bursts):
    their sequences are mostly 'repeat offset 1' matches, which is what the decoder's out-of-order zero-run path
    keys on -- with libzstd's own FSE tables, literal lengths of zero, other offsets in between.  Whatever path a
    block takes (parallel placement, serial chain, restart in order), the bytes must be libzstd's."""
    import gpu_util as G

    if O.lib().vbo_zstd_version() is None:
        pytest.skip("no libzstd on this box")
    rng = np.random.default_rng(77)
    contents = []
    for n, p_burst in ((30000, 0.01), (120000, 0.004), (250000, 0.02), (5000, 0.05)):
        a = np.zeros(n, np.uint8)
        starts = np.nonzero(rng.random(n) < p_burst)[0]
        for s0 in starts:
            ln = min(int(rng.integers(1, 12)), n - int(s0))
            a[int(s0) : int(s0) + ln] = rng.integers(1, 86, ln, dtype=np.uint8)
        contents.append(a)
        b = a.copy()
        m = min(4000, n - n // 3)
        b[n // 3 : n // 3 + m] = np.tile(rng.integers(0, 256, 40, dtype=np.uint8), 100)[:m]   # real matches in the middle
        contents.append(b)
    c = np.full(60000, 7, np.uint8)                                                              # runs of a non-zero byte
    c[rng.integers(0, 60000, 300)] = rng.integers(0, 256, 300, dtype=np.uint8)
    contents.append(c)
    frames, want = [], []
    for a in contents:
        for level in (1, 2, 3, 5):
            frames.append(O.zstd_compress(a, level))
            want.append(a)
    got = G.zstd_decompress(frames, [len(w) for w in want])
    for f, w, g in zip(frames, want, got):
        assert _same(g, w), (len(w), len(f))


def test_zstd_decode_shipped_fast5_chunks():
    # decode pins: python/test/test_vbz_filter.py:57-73 (frames written by the reference itself)
    import gpu_util as G

    idx = json.load(open(os.path.join(GOLDEN, "fast5_chunks.json")))
    blob = np.fromfile(os.path.join(GOLDEN, "fast5_chunks.bin"), np.uint8)
    chunks = [blob[e["chunk_offset"] : e["chunk_offset"] + e["chunk_size"]] for e in idx]
    opts = G.codec().options(True, 2, 1, 0)
    got = G.decompress(chunks, [2 * e["samples"] for e in idx], opts, sized=True)
    for e, g in zip(idx, got):
        assert not isinstance(g, int), g
        assert hashlib.sha256(g.tobytes()).hexdigest() == e["raw_sha256"]


def test_zstd_encode_cross_decodes_with_libzstd():
    import gpu_util as G

    streams = _svb_streams()
    keys = [0, 1, 2, 10, 50, 175, 250, 750, 2250, 7500, 25000, 27500, 100000, 250, 50000]
    frames = G.zstd_compress(streams, key_bytes=keys)
    frames_nosplit = G.zstd_compress(streams)
    for s, f, f2 in zip(streams, frames, frames_nosplit):
        for fr in (f, f2):
            assert not isinstance(fr, int), fr
            assert O.zstd_content_size(fr) == len(s)
            back = O.zstd_decompress(fr, len(s))
            assert back is not None and back.tobytes() == s.tobytes(), len(s)
            mine = O.zstd_restate_decompress(fr, len(s))
            assert mine is not None and mine.tobytes() == s.tobytes()
    # the GPU decoder reads its own frames
    got = G.zstd_decompress(frames, [len(s) for s in streams])
    for s, g in zip(streams, got):
        assert _same(g, s)


def test_sequences_section_round_and_lane_boundaries():
    """The encoder's sequences section runs its two FSE state chains on all lanes (four-state warm-up walks; a lane whose
    walks have not met takes its neighbour's state; rounds of 256 sequences).  Control-byte regions with exactly N runs, N
    around every boundary of that scheme, with literal-length and match-length codes of one, two, three and four table
    cells, must decode with libzstd and the restated decoder to the same bytes."""
    import gpu_util as G

    rng = np.random.default_rng(11)
    streams, keys = [], []
    for nruns in [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 259, 260, 261, 511, 512, 513, 767, 768, 769, 1500, 4000]:
        for style in range(3):
            parts = []
            for _ in range(nruns):
                if style == 0:      # long literal gaps (many-cell literal-length codes), runs of every length
                    gap, run = int(rng.integers(1, 40)), int(rng.integers(12, 300))
                elif style == 1:    # shortest runs (match-length code with two cells), no or one literal between them
                    gap, run = int(rng.integers(1, 3)), int(rng.integers(12, 15))
                else:               # literal lengths 0..3 and the match lengths around 12: the codes with several cells
                    gap, run = int(rng.integers(1, 5)), 12 + int(rng.integers(0, 2)) * int(rng.integers(0, 40))
                parts.append(rng.integers(1, 256, gap, dtype=np.uint8))
                parts.append(np.zeros(run, np.uint8))
            k = np.concatenate(parts)
            data = rng.integers(0, 256, 3000, dtype=np.uint8) // rng.integers(1, 9)
            streams.append(np.concatenate([k, data.astype(np.uint8)]))
            keys.append(len(k))
    frames = G.zstd_compress(streams, key_bytes=keys)
    for s, f in zip(streams, frames):
        assert not isinstance(f, int), f
        back = O.zstd_decompress(f, len(s))
        assert back is not None and back.tobytes() == s.tobytes(), len(s)
        mine = O.zstd_restate_decompress(f, len(s))
        assert mine is not None and mine.tobytes() == s.tobytes()
    got = G.zstd_decompress(frames, [len(s) for s in streams])
    for s, g in zip(streams, got):
        assert _same(g, s)


def test_sampled_histogram_that_misleads():
    """Regions of 32 KB and more are coded from a histogram of every fourth kilobyte when that sample shows nearly all byte
    values (zstd_encode.hip: region_histogram).  Data whose sampled kilobytes look nothing like the rest -- here: almost
    constant where the sample looks, uniform noise elsewhere -- would be coded 37 % LONGER than it is; the encoder has to
    notice, start the region over with the exact histogram and end up no larger than the stored form."""
    import gpu_util as G

    rng = np.random.default_rng(21)
    streams = []
    for n in (40000, 131072, 200000, 1 << 20):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        for s0 in range(0, n, 4096):                       # the sampled stripe of every four: zeros, every value sprinkled in
            stripe = np.zeros(min(1024, n - s0), np.uint8)
            k = min(len(stripe), 256)
            stripe[rng.permutation(len(stripe))[:k]] = np.arange(k, dtype=np.uint8)
            a[s0 : s0 + len(stripe)] = stripe
        streams.append(a)
        b = a.copy()                                        # the other way round: noise where the sample looks, zeros elsewhere
        b[:] = 0
        for s0 in range(0, n, 4096):
            m = min(1024, n - s0)
            b[s0 : s0 + m] = rng.integers(0, 256, m, dtype=np.uint8)
        streams.append(b)
    frames = G.zstd_compress(streams)
    for s, f in zip(streams, frames):
        assert not isinstance(f, int), (len(s), f)
        assert len(f) <= len(s) + (len(s) >> 7) + 64, (len(s), len(f))
        back = O.zstd_decompress(f, len(s))
        assert back is not None and back.tobytes() == s.tobytes(), len(s)
    got = G.zstd_decompress(frames, [len(s) for s in streams])
    for s, g in zip(streams, got):
        assert _same(g, s)
    # the second kind is highly compressible once counted properly (3/4 zeros): the exact histogram must have been used
    for s, f in zip(streams[1::2], frames[1::2]):
        assert len(f) < 0.45 * len(s), (len(s), len(f))


def test_huffman_blocks_stay_below_the_block_maximum():
    """A compressed block above Block_Maximum_Size (128 KB) is one no decoder accepts.  On the one-wavefront path blocks of a
    long stream used to be cut at 128 KB of CONTENT, and a Huffman block can come out longer than its content (11-bit codes
    from a region-wide or sampled table on a block that looks nothing like the rest): content is now cut at 92 KB.  A 1.5 M-sample
    read next to small ones (the batch stays on the one-wavefront path) whose data bytes change character block by block."""
    import gpu_util as G
    from vbz_compression_amd import _lib

    rng = np.random.default_rng(41)
    parts = []
    for k in range(12):   # quiet stretches (one byte value dominates) next to noise (all byte values): one table serves neither well
        if k % 2:
            parts.append(rng.integers(-127, 128, 125000).astype(np.int16))
        else:
            parts.append((rng.integers(0, 2, 125000) * 3).astype(np.int16))
    big = np.cumsum(np.concatenate(parts)).astype(np.int16)
    reads = [big] + [O.synth_signal(5, i, 20000) for i in range(7)]
    go, oo = _lib.CompressionOptions(True, 2, 1, 1), O.options(True, 2, 1, 1)
    frames = G.compress(reads, go)
    back = G.decompress(frames, [a.nbytes for a in reads], go)
    for a, f, b in zip(reads, frames, back):
        assert not isinstance(f, int) and not isinstance(b, int)
        assert b.tobytes() == a.tobytes()
        assert O.decompress(f, a.nbytes, oo).tobytes() == a.tobytes()          # libzstd refuses oversized blocks
    # every block header of the big frame announces at most 128 KB
    f = frames[0]
    p = 4 + 1 + 4   # magic, frame header descriptor (0xA0), 4-byte content size
    assert f[4] == 0xA0
    while True:
        bh = int(f[p]) | (int(f[p + 1]) << 8) | (int(f[p + 2]) << 16)
        last, bt, bs = bh & 1, (bh >> 1) & 3, bh >> 3
        assert bs <= 128 * 1024
        p += 3 + (1 if bt == 1 else bs)
        if last:
            break


def test_region_larger_than_the_sort_key_counts():
    """One wavefront on a 40 MB stream (ordinary path forced): the histogram of the data region exceeds the 24 bits the
    table construction's sort keys hold and is scaled down; the code stays valid and the frame decodes everywhere."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import gpu_util as G, oracle_lib as O\n"
        "rng = np.random.default_rng(5)\n"
        "a = (rng.normal(0, 30, 20_000_000)).astype(np.int16)\n"
        "o = G.codec().options(True, 2, 1, 1)\n"
        "f = G.compress([a], o)[0]\n"
        "assert not isinstance(f, int), f\n"
        "assert O.decompress(f, a.nbytes, O.options(True, 2, 1, 1)).tobytes() == a.tobytes()\n"
        "b = G.decompress([f], [a.nbytes], o)[0]\n"
        "assert b.tobytes() == a.tobytes()\n"
        "print('ok', len(f))\n" % os.path.dirname(os.path.abspath(__file__))
    )
    env = dict(os.environ, VBZ_HIP_SEGMENTED="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout[-400:], r.stderr[-1200:])


def test_zstd_ratio_close_to_libzstd_on_signal():
    import gpu_util as G

    reads = [O.synth_signal(5, i, O.synth_read_length(5, i)) for i in range(8)]
    opts = G.codec().options(True, 2, 1, 1)
    frames = G.compress(reads, opts)
    gpu = sum(len(f) for f in frames)
    ref = sum(len(O.compress(a, O.options(True, 2, 1, 1))) for a in reads)
    assert abs(gpu / ref - 1.0) < 0.01, (gpu, ref)  # ratio within 1 % of the reference


def test_zstd_levels_above_one_write_the_level_one_frames():
    """The reference hands zstd_compression_level to libzstd (vbz/vbz.cpp:194-207) and its own HDF5 test writes at level 5
    (vbz_plugin/test/vbz_hdf_plugin_test.cpp:34).  This library has ONE encoder (include/vbz.h): any level but 0 gives the bytes of
    level 1 -- and on nanopore signal that is where libzstd's levels are anyway: the ten real reads of the shipped fast5 file and eight
    synthetic ones at level 5 (the plugin's case: UD=32020,0,2,1,1,5) come out within 1 % of what libzstd writes at level 1 (libzstd's
    own level 5 gains 0.1 % on the real reads and 0.5 % on the synthetic ones), sized and decodable by the reference path at either level."""
    import gpu_util as G

    idx = json.load(open(os.path.join(GOLDEN, "fast5_chunks.json")))
    blob = np.fromfile(os.path.join(GOLDEN, "fast5_chunks.bin"), np.uint8)
    real = [O.decompress(blob[e["chunk_offset"] : e["chunk_offset"] + e["chunk_size"]], 2 * e["samples"], O.options(True, 2, 1, 0), sized=True).view(np.int16) for e in idx]
    reads = real + [O.synth_signal(5, 100 + i, O.synth_read_length(5, 100 + i)) for i in range(8)]
    by_level = {lv: G.compress(reads, _lib_opts(True, 2, lv, 0), sized=True) for lv in (1, 5, 19)}
    for f1, f5, f19, a in zip(by_level[1], by_level[5], by_level[19], reads):
        assert not isinstance(f5, int) and f1.tobytes() == f5.tobytes() == f19.tobytes()
        assert O.decompress(f5, a.nbytes, O.options(True, 2, 5, 0), sized=True).tobytes() == a.tobytes()
    for group in (slice(0, len(real)), slice(len(real), None)):
        gpu = sum(len(f) for f in by_level[5][group])
        ref1 = sum(len(O.compress(a, O.options(True, 2, 1, 0), sized=True)) for a in reads[group])
        ref5 = sum(len(O.compress(a, O.options(True, 2, 5, 0), sized=True)) for a in reads[group])
        assert abs(gpu / ref1 - 1.0) < 0.01, (gpu, ref1, ref5)
        assert gpu < 1.02 * ref5, (gpu, ref1, ref5)    # ... and within 2 % of what libzstd's level 5 finds


def test_checkpoint_trailer_is_optional_and_untrusted():
    """The encoder appends a skippable frame with decoder checkpoints (zstd_encode.hip, CP_MAGIC).  It must be
    invisible to the reference's decoder, optional for ours, and never trusted: a wrong trailer costs speed only."""
    import gpu_util as G

    opts = G.codec().options(True, 2, 1, 1)
    reads = [O.synth_signal(5, i, O.synth_read_length(5, i)) for i in range(3)]
    frames = G.compress(reads, opts)
    rng = np.random.default_rng(3)
    variants, want = [], []
    for a, f in zip(reads, frames):
        assert not isinstance(f, int), f
        tb = int(f[-4:].view("<u4")[0])
        head = f[len(f) - tb : len(f) - tb + 12].view("<u4")
        if head[0] == 0x184D2A5C:   # the span index of the large-read path (VBZ_HIP_SEGMENTED=1 forces it): not this test's subject
            f = f[: len(f) - tb].copy()
            tb = int(f[-4:].view("<u4")[0])
            head = f[len(f) - tb : len(f) - tb + 12].view("<u4")
        assert head[0] == 0x184D2A5B and head[1] == tb - 8 and tb == 16 + 4 * (int(head[2]) >> 16)
        back = O.decompress(f, a.nbytes, O.options(True, 2, 1, 1))           # the reference path (libzstd) skips it
        assert not isinstance(back, int) and back.tobytes() == a.tobytes()
        body = f[: len(f) - tb]
        cands = [f, body]                                                       # as written; without the trailer
        for _ in range(6):                                                      # damaged checkpoints, spacing, count
            g = f.copy()
            at = len(f) - tb + 8 + int(rng.integers(0, tb - 12))
            g[at] ^= 1 << int(rng.integers(0, 8))
            cands.append(g)
        junk = np.concatenate([np.array([0x5B, 0x2A, 0x4D, 0x18, 20, 0, 0, 0], np.uint8), rng.integers(0, 256, 20, dtype=np.uint8)])
        cands.append(np.concatenate([body, junk]))                             # some other skippable frame behind
        cands.append(np.concatenate([f, junk]))                                # ... and behind the trailer
        for c in cands:
            variants.append(np.ascontiguousarray(c))
            want.append(a)
    got = G.decompress(variants, [w.nbytes for w in want], opts)
    for v, w, g in zip(variants, want, got):
        lz = O.decompress(v, w.nbytes, O.options(True, 2, 1, 1))
        if isinstance(lz, int):      # damage that breaks the skippable frame's own header: both must refuse
            assert isinstance(g, int), (g, lz)
        else:
            assert not isinstance(g, int), g
            assert g.tobytes() == w.tobytes()
    # frames libzstd wrote, with skippable frames behind them; and a second data frame, which vbz never writes
    svb = O.svb_compress(reads[0], 2, True, 1)
    zf = O.zstd_compress(svb, 1)
    junk = np.concatenate([np.array([0x50, 0x2A, 0x4D, 0x18, 5, 0, 0, 0], np.uint8), rng.integers(0, 256, 5, dtype=np.uint8)])
    got = G.zstd_decompress([np.concatenate([zf, junk]), np.concatenate([zf, junk, junk]), np.concatenate([zf, zf])], [len(svb)] * 3)
    assert got[0].tobytes() == svb.tobytes() and got[1].tobytes() == svb.tobytes()
    assert isinstance(got[2], int)


def test_zstd_decode_rejects_corruption():
    import gpu_util as G

    rng = np.random.default_rng(33)
    s = O.svb_compress(O.synth_signal(5, 2, 30000), 2, True, 0)
    frame = O.zstd_compress(s, 1)
    frames = [frame[:cut].copy() for cut in (0, 1, 4, 5, 8, 9, 12, len(frame) // 2, len(frame) - 1)]
    for _ in range(60):
        bad = frame.copy()
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        frames.append(bad)
    got = G.zstd_decompress(frames, [len(s)] * len(frames))
    for f, g in zip(frames, got):
        mine = O.zstd_restate_decompress(f, len(s))
        if len(f) == 0:
            continue
        if mine is None:
            assert isinstance(g, int) and g == 0xFFFFFFFF
        else:
            assert _same(g, mine)


def test_zstd_decode_rejects_corruption_of_long_libzstd_frames():
    """A read-sized frame of the reference: its four literal streams are long enough to be decoded in 64 pieces and its
    sequences are executed out of order.  Damage must be handled exactly as the strict restatement handles it."""
    import gpu_util as G

    if O.lib().vbo_zstd_version() is None:
        pytest.skip("no libzstd on this box")
    rng = np.random.default_rng(35)
    s = O.svb_compress(O.synth_signal(5, 7, 120000), 2, True, 0)
    frames, sizes = [], []
    for level in (1, 3):
        frame = O.zstd_compress(s, level)
        frames += [frame[:cut].copy() for cut in (len(frame) // 3, len(frame) - 2)]
        for k in range(80):
            bad = frame.copy()
            for _ in range(int(rng.integers(1, 3))):
                # every region gets its share: headers, trees, the streams, the sequence section
                lo = 0 if k % 4 else int(len(bad) * 0.9)
                bad[int(rng.integers(lo, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            frames.append(bad)
    got = G.zstd_decompress(frames, [len(s)] * len(frames))
    refused = 0
    for f, g in zip(frames, got):
        mine = O.zstd_restate_decompress(f, len(s))
        if mine is None:
            refused += 1
            assert isinstance(g, int) and g == 0xFFFFFFFF
        else:
            assert _same(g, mine)
    assert refused > 20


def test_zstd_decode_rejects_corruption_of_own_frames():
    """Bit flips in frames of this library's encoder (zero-run sequences, treeless blocks, checkpoint trailer): the
    decoder must agree with the strict restatement -- same bytes or both refuse -- whichever of its paths a damaged
    frame ends up on."""
    import gpu_util as G

    rng = np.random.default_rng(91)
    a = O.synth_signal(5, 11, 60000)
    svb = O.svb_compress(a, 2, True, 1)
    frame = G.compress([a], G.codec().options(True, 2, 1, 1))[0]
    assert O.zstd_restate_decompress(frame, len(svb)).tobytes() == svb.tobytes()
    frames = []
    for _ in range(150):
        bad = frame.copy()
        for _ in range(int(rng.integers(1, 3))):
            # half of the damage goes to the first 3 KB (headers, tree, sequences), the rest anywhere
            hi = 3000 if rng.random() < 0.5 else len(bad)
            bad[int(rng.integers(0, hi))] ^= 1 << int(rng.integers(0, 8))
        frames.append(bad)
    got = G.zstd_decompress(frames, [len(svb)] * len(frames))
    agree = 0
    for f, g in zip(frames, got):
        mine = O.zstd_restate_decompress(f, len(svb))
        if mine is None:
            assert isinstance(g, int), "accepted a frame the strict decoder refuses"
        else:
            assert _same(g, mine)
            agree += 1
    assert agree >= 1   # some flips only hit bytes that do not matter (trailer, padding)


# ------------------------------------------------------------------------------------------------
# the whole path through the drop-in C ABI (host pointers)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k", KAT["l2"], ids=lambda k: k["cite"][:40])
def test_c_abi_known_answers(k):
    from vbz_compression_amd import _lib, vbz

    a = np.array(k["input"], dtype=DT[k["dtype"]])
    o = k["opts"]
    opts = _lib.CompressionOptions(o["zigzag"], o["size"], o["level"], o["version"])
    want = np.array(k["out_i8"], np.int8).view(np.uint8).tobytes() if "out_i8" in k else bytes.fromhex(k["out_hex"])
    got = vbz.compress_raw(a, opts, sized=k["sized"])
    assert not isinstance(got, int), got
    assert got.tobytes() == want
    back = vbz.decompress_raw(got, a.nbytes, opts, sized=k["sized"])
    assert back.tobytes() == a.tobytes()


def test_c_abi_cross_codec_round_trips():
    """GPU-compressed buffers decode with the oracle (= the reference's decoder) and vice versa."""
    from vbz_compression_amd import _lib, vbz

    rng = np.random.default_rng(44)
    cases = []
    for dt, size in ((np.int16, 2), (np.int32, 4), (np.int8, 1)):
        info = np.iinfo(dt)
        for n in (0, 1, 100, 4099, 70001):
            cases.append((rng.integers(info.min // 2, info.max // 2, n).astype(dt), size))
    cases.append((O.synth_signal(5, 0, 400000), 2))      # config 1: one 400k-sample read
    for a, size in cases:
        for zz in (True, False):
            for level in (0, 1):
                for ver in (0, 1):
                    for sized in (False, True):
                        go = _lib.CompressionOptions(zz, size, level, ver)
                        oo = O.options(zz, size, level, ver)
                        g = vbz.compress_raw(a, go, sized=sized)
                        assert not isinstance(g, int), (g, len(a), size, zz, level, ver, sized)
                        assert O.decompress(g, a.nbytes, oo, sized=sized).tobytes() == a.tobytes()
                        r = O.compress(a, oo, sized=sized)
                        back = vbz.decompress_raw(r, a.nbytes, go, sized=sized)
                        assert not isinstance(back, int), (back, len(a), size, zz, level, ver, sized)
                        assert back.tobytes() == a.tobytes()
                        if level == 0:
                            assert g.tobytes() == r.tobytes()   # no entropy stage: byte-identical output


def test_single_buffer_calls_across_the_pinned_staging_boundaries():
    """vbz_compress / vbz_decompress stage reads and results of up to 1 MB through pinned host memory and take the result back through
    a flag the calling thread polls (vbz_api.hip run_one, hand_back_kernel); larger ones use plain copies.  Sizes on both sides of the
    boundary (for the input, for the result, for the worst-case slot), empty and tiny reads, an incompressible read whose result is
    larger than its input, error verdicts -- from four threads at once, several hundred calls each: every result byte for byte what
    the oracle says, and what the batched entry points write."""
    import threading

    import gpu_util as G
    from vbz_compression_amd import _lib, vbz

    rng = np.random.default_rng(77)
    go, oo = _lib.CompressionOptions(True, 2, 1, 1), O.options(True, 2, 1, 1)
    sizes = [0, 1, 7, 300, 5000, 100000, 262000, 262144, 262200, 466000, 466100, 524288, 524300, 700000]   # samples: 1 MB = 524 288
    reads = [O.synth_signal(5, 900 + i, n) for i, n in enumerate(sizes)]
    reads.append(rng.integers(-32768, 32767, 300000, dtype=np.int16))       # noise: the frame is larger than half the input
    reads.append(rng.integers(-32768, 32767, 520000, dtype=np.int16))       # ... and larger than 1 MB from an input below it
    batched = [G.compress([a], go)[0] for a in reads]     # (a call of its own each: the launch paths go by the shape of the call)
    errors = []

    def work(k):
        try:
            for rep in range(3):
                for i in range(k, len(reads), 4):
                    a = reads[i]
                    g = vbz.compress_raw(a, go)
                    if isinstance(g, int) or g.tobytes() != batched[i].tobytes():
                        errors.append((i, "compress differs from the batched call"))
                        continue
                    back = vbz.decompress_raw(g, a.nbytes, go)
                    if isinstance(back, int) or back.tobytes() != a.tobytes():
                        errors.append((i, "round trip"))
                    ref = O.compress(a, oo)
                    back = vbz.decompress_raw(ref, a.nbytes, go)
                    if isinstance(back, int) or back.tobytes() != a.tobytes():
                        errors.append((i, "reference frame"))
                    if a.nbytes >= 16:   # error verdicts come back through the same flag
                        bad = vbz.decompress_raw(g[: len(g) // 2], a.nbytes, go)
                        want = O.decompress(g[: len(g) // 2], a.nbytes, oo)
                        if not isinstance(bad, int) and (isinstance(want, int) or want.tobytes() != bad.tobytes()):
                            errors.append((i, "truncated frame accepted"))
                        small = vbz.decompress_raw(g, a.nbytes - 2, go)
                        if not isinstance(small, int):
                            errors.append((i, "a destination that is too small"))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_c_abi_calls_from_many_threads_overlap():
    """The reference's calls are re-entrant and stateless; its callers parallelise by calling from many threads
    (SURVEY 8b "Threading").  Here a call borrows one of a pool of contexts, so concurrent calls must (a) stay correct
    and (b) overlap on the GPU instead of queueing behind one another."""
    import threading
    import time

    from vbz_compression_amd import _lib, vbz

    go = _lib.CompressionOptions(True, 2, 1, 1)
    oo = O.options(True, 2, 1, 1)
    reads = [O.synth_signal(5, 100 + i, 60000 + 997 * i) for i in range(32)]
    want = [O.compress(a, oo, sized=True) for a in reads]
    vbz.compress_raw(reads[0], go, sized=True)  # context creation is not what is timed
    errors = []

    def work(ids, rounds):
        try:
            for _ in range(rounds):
                for i in ids:
                    g = vbz.compress_raw(reads[i], go, sized=True)
                    back = vbz.decompress_raw(want[i], reads[i].nbytes, go, sized=True)    # a frame of the reference
                    mine = vbz.decompress_raw(g, reads[i].nbytes, go, sized=True)
                    if isinstance(g, int) or isinstance(back, int) or isinstance(mine, int):
                        errors.append((i, "error code"))
                    elif back.tobytes() != reads[i].tobytes() or mine.tobytes() != reads[i].tobytes():
                        errors.append((i, "bytes differ"))
                    elif O.decompress(g, reads[i].nbytes, oo, sized=True).tobytes() != reads[i].tobytes():
                        errors.append((i, "oracle cannot read it"))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", repr(e)))

    def timed(nthreads, rounds):
        ids = list(range(len(reads)))
        ts = [threading.Thread(target=work, args=(ids[k::nthreads], rounds)) for k in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    timed(8, 1)          # warm the pool
    one = timed(1, 2)
    eight = timed(8, 2)
    assert not errors, errors[:3]
    # most of a call is latency (copies, a handful of one-workgroup kernels): eight callers must not take eight turns
    # (measured 0.3 - 0.65 of the single-thread time, Python's own locking included; behind one mutex it would be >= 1)
    assert eight < 0.9 * one, (one, eight)


def test_config3_uint32_ten_million_elements():
    """BASELINE.json configs[3]: uint32, no zig-zag (UD=32020,5,0,0,4,0,3), one 10M-element buffer through the
    single-buffer C ABI: many passes of blocks in one frame, every code length; cross-decoded both ways."""
    from vbz_compression_amd import _lib, vbz

    a = O.synth_u32(5, 3, 10_000_000)
    go = _lib.CompressionOptions(False, 4, 3, 0)
    oo = O.options(False, 4, 3, 0)
    g = vbz.compress_raw(a, go, sized=True)
    assert not isinstance(g, int), g
    back = O.decompress(g, a.nbytes, oo, sized=True)
    assert not isinstance(back, int) and back.tobytes() == a.tobytes()
    r = O.compress(a, oo, sized=True)
    mine = vbz.decompress_raw(r, a.nbytes, go, sized=True)
    assert not isinstance(mine, int) and mine.tobytes() == a.tobytes()
    assert abs(len(g) / len(r) - 1.0) < 0.01, (len(g), len(r))   # T2: within 1 % of the oracle (libzstd level 3) on config 4


def test_cpp_caller_relinked_against_libvbz_hip(tmp_path):
    """a C++ translation unit written against the reference's interface (here: a small
    caller in the idiom of vbz/test/vbz_test.cpp) builds against include/vbz.h and runs after linking -lvbz_hip."""
    import subprocess

    from vbz_compression_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "c_caller")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "host", "c_caller.cpp"),
                           "-L", libdir, "-lvbz_hip", "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)


def test_c_abi_error_behaviour():
    from vbz_compression_amd import _lib, vbz

    a = np.arange(10, dtype=np.int16)
    C = _lib.CompressionOptions
    assert vbz.compress_raw(a, C(True, 3, 1, 0)) == _lib.VBZ_INTEGER_SIZE_ERROR
    assert vbz.compress_raw(a, C(True, 2, 1, 2)) == _lib.VBZ_VERSION_ERROR
    assert vbz.compress_raw(np.zeros(3, np.uint8), C(True, 2, 0, 0)) == _lib.VBZ_INPUT_SIZE_ERROR
    c = vbz.compress_raw(a, C(True, 2, 0, 0))
    assert vbz.decompress_raw(c[:-1], 20, C(True, 2, 0, 0)) == _lib.VBZ_STREAMVBYTE_STREAM_ERROR
    assert vbz.decompress_raw(c, 22, C(True, 2, 0, 0)) == _lib.VBZ_STREAMVBYTE_STREAM_ERROR
    assert vbz.decompress_raw(c[:2], 20, C(True, 2, 0, 0)) == _lib.VBZ_INPUT_SIZE_ERROR
    assert vbz.decompress_raw(c, 19, C(True, 2, 0, 0)) == _lib.VBZ_DESTINATION_SIZE_ERROR
    z = vbz.compress_raw(a, C(True, 2, 1, 0))
    bad = z.copy()
    bad[0] ^= 1
    assert vbz.decompress_raw(bad, 20, C(True, 2, 1, 0)) == _lib.VBZ_ZSTD_ERROR
    L = _lib.load()
    for k in KAT["size_pins"]:
        o = C(True, 2, k["level"], 1)
        assert L.vbz_max_compressed_size(k["samples"] * 2, ctypes.byref(o)) == k["max"]
    assert L.vbz_error_string(_lib.VBZ_ZSTD_ERROR) == b"VBZ_ZSTD_ERROR"
    assert L.vbz_error_string(12345) == b"VBZ_UNKNOWN_ERROR"
    assert L.vbz_is_error(_lib.VBZ_OUT_OF_MEMORY_ERROR) and not L.vbz_is_error(1000)


def test_pyvbz_interface():
    # python/pyvbz/tests/unit: [1..10] and 200 000 random values, 6 dtypes, default + v1
    from vbz_compression_amd import vbz

    rng = np.random.default_rng(55)
    for dt in (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32):
        info = np.iinfo(dt)
        for a in (np.arange(1, 11).astype(dt), rng.integers(info.min, info.max, 200000, endpoint=True).astype(dt)):
            for ver in (0, 1):
                c = vbz.compress(a, version=ver)
                assert vbz.decompressed_size(c, dt, version=ver) == a.nbytes
                assert (vbz.decompress(c, dt, version=ver) == a).all()
                # the reference's positional form: compress(data, options), decompress(data, dtype, options)
                o = vbz.compression_options(np.issubdtype(dt, np.signedinteger), np.dtype(dt).itemsize, 1, ver)
                assert vbz.compress(a, o).tobytes() == c.tobytes() and (vbz.decompress(c, dt, o) == a).all()
    with pytest.raises(TypeError):
        vbz.compress(np.arange(4, dtype=np.int16), True)   # a bare zig-zag flag in the options slot is refused, not ignored
    sig = np.arange(0, 1000, dtype=np.int16)    # python/pyvbz/README.md:18-23
    c = vbz.compress(sig)
    # the reference's answer is 27 bytes (libzstd finds the two runs: kat.json "pyvbz_readme"); this encoder's run sequences -- the 251
    # zero bytes and the 999 twos of the 1 250-byte stream are two sequences behind two literals -- need 25, on either launch path
    assert c[:4].view("<u4")[0] == 2000 and len(c) <= 27, len(c)
    assert (vbz.decompress(c, np.int16) == sig).all()


# ------------------------------------------------------------------------------------------------
# batched device-resident path
# ------------------------------------------------------------------------------------------------
def test_batch_ragged_reads_and_per_read_errors():
    import gpu_util as G

    reads = [O.synth_signal(5, i, n) for i, n in enumerate([0, 1, 17, 4096, 100000, 0, 33333, 90001])]
    for level in (0, 1):
        for sized in (False, True):
            opts = G.codec().options(True, 2, level, 1)
            frames = G.compress(reads, opts, sized=sized)
            for a, f in zip(reads, frames):
                assert not isinstance(f, int), f
                assert O.decompress(f, a.nbytes, O.options(True, 2, level, 1), sized=sized).tobytes() == a.tobytes()
            caps = [a.nbytes for a in reads]
            back = G.decompress(frames, caps, opts, sized=sized)
            for a, g in zip(reads, back):
                assert _same(g, np.frombuffer(a.tobytes(), np.uint8))
    # one bad read does not disturb its neighbours
    opts = G.codec().options(True, 2, 1, 1)
    frames = G.compress(reads, opts)
    frames[3] = frames[3][: len(frames[3]) // 2].copy()
    back = G.decompress(frames, [a.nbytes for a in reads], opts)
    for i, (a, g) in enumerate(zip(reads, back)):
        if i == 3:
            assert g == 0xFFFFFFFF
        else:
            assert _same(g, np.frombuffer(a.tobytes(), np.uint8))
    odd = G.compress([np.zeros(3, np.uint8), reads[2]], G.codec().options(True, 2, 1, 1))
    assert odd[0] == 0xFFFFFFFE and not isinstance(odd[1], int)


def test_unaligned_arena_offsets():
    """vbz_gpu.h asks for 16-byte aligned reads but promises a (slower) path for anything else: odd byte offsets on
    both arenas, every stage and the whole path, same bytes as with aligned offsets."""
    import gpu_util as G

    opts = G.codec().options(True, 2, 1, 1)
    reads = [O.synth_signal(5, i, n) for i, n in enumerate([100003, 4097, 50000, 1, 0, 7, 90001])]
    G.SRC_ALIGN, G.DST_ALIGN, G.SRC_SKEW = 1, 1, 3
    try:
        svb = G.svb_compress(reads, 2, True, 1)
        for a, g in zip(reads, svb):
            assert _same(g, O.svb_compress(a, 2, True, 1)), len(a)
        streams = [O.svb_compress(a, 2, True, 1) for a in reads if len(a)]
        back = G.svb_decompress(streams, [a.nbytes for a in reads if len(a)], 2, True, 1)
        for a, g in zip([a for a in reads if len(a)], back):
            assert g.tobytes() == a.tobytes()
        frames = G.compress(reads, opts, sized=True)
        for a, f in zip(reads, frames):
            assert not isinstance(f, int), f
            d = O.decompress(f, a.nbytes, O.options(True, 2, 1, 1), sized=True)
            assert not isinstance(d, int) and d.tobytes() == a.tobytes(), len(a)
        got = G.decompress(frames, [a.nbytes for a in reads], opts, sized=True)
        for a, g in zip(reads, got):
            assert (not isinstance(g, int)) and g.tobytes() == a.tobytes(), len(a)
    finally:
        G.SRC_ALIGN, G.DST_ALIGN, G.SRC_SKEW = 64, 64, 0


def test_full_size_device_round_trip_properties():
    """Config-2-shaped batch generated on the device: encode -> decode is the identity (checked on the
    device), every frame header carries the svb size, and a sample of reads matches the oracle."""
    import torch
    from vbz_compression_amd import batch
    import gpu_util as G

    c = G.codec()
    dev = c.device
    n = 16384
    lens = c.synth_lengths(5, 0, n)
    sizes = (lens.to(torch.int64) * 2)
    off, total = batch.layout(sizes.cpu(), 64)
    raw = torch.zeros(total, dtype=torch.uint8, device=dev)
    off = off.to(dev)
    c.synth_signal(5, 0, raw, off, lens)
    size32 = sizes.to(torch.int32)
    opts = c.options(True, 2, 1, 1)
    L = c.L
    caps = torch.tensor([L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
    coff, ctotal = batch.layout(caps, 64)
    comp = torch.zeros(ctotal, dtype=torch.uint8, device=dev)
    coff = coff.to(dev)
    cap32 = caps.to(torch.int32).to(dev)
    csize = torch.zeros(n, dtype=torch.int32, device=dev)
    c.compress(raw, off, size32, comp, coff, cap32, csize, opts)
    back = torch.zeros_like(raw)
    res = torch.zeros(n, dtype=torch.int32, device=dev)
    c.decompress(comp, coff, csize, back, off, size32, res, opts)
    torch.cuda.synchronize()
    assert (csize > 0).all() and (res == size32).all()
    assert torch.equal(raw, back)
    ratio = float(sizes.sum()) / float(csize.to(torch.int64).sum())
    assert 2.2 < ratio < 2.6, ratio
    offh, coffh, csz = off.cpu().tolist(), coff.cpu().tolist(), csize.cpu().tolist()
    oo = O.options(True, 2, 1, 1)
    osum = gsum = 0
    for i in [0, 1, 777, n - 1] + list(range(5, n, n // 28)):   # 32+ reads spread over the batch, each against the oracle
        o, s = offh[i], int(sizes[i])
        a = raw[o : o + s].cpu().numpy().view(np.int16)
        assert (a == O.synth_signal(5, i, s // 2)).all()                 # device generator == oracle generator
        f = comp[coffh[i] : coffh[i] + csz[i]].cpu().numpy()
        assert O.decompress(f, s, oo).tobytes() == a.tobytes()           # the reference's decoder reads the device's frame
        ref = O.compress(a, oo)
        osum += len(ref)
        gsum += len(f)
    assert abs(gsum / osum - 1.0) < 0.01, (gsum, osum)                   # T2: size within 1 % of the oracle's on the sample


def test_declared_extents_that_no_device_holds():
    """src_bytes / dst_bytes size the library's scratch.  A declared extent beyond 2^46 bytes is refused before any arithmetic is done with
    it (-2, a message); one that is merely more than the device has is an allocation failure (-1) that leaves neither the context nor
    the calling thread's HIP state damaged: the next call -- and the caller's own next HIP call -- work."""
    import torch
    from vbz_compression_amd import batch

    c = batch.GpuCodec(0)
    dev = c.device
    opts = c.options(True, 2, 1, 1)
    a = O.synth_signal(5, 1, 50000)
    n = 4
    with torch.cuda.stream(c.stream):
        raw = torch.from_numpy(np.tile(a.view(np.uint8), n).copy()).to(dev)
        off = torch.arange(n, dtype=torch.int64, device=dev) * a.nbytes
        size32 = torch.full((n,), a.nbytes, dtype=torch.int32, device=dev)
        cap = c.L.vbz_max_compressed_size(a.nbytes, ctypes.byref(opts))
        coff = torch.arange(n, dtype=torch.int64, device=dev) * ((cap + 63) // 64 * 64)
        comp = torch.zeros(int(coff[-1]) + cap + 64, dtype=torch.uint8, device=dev)
        cap32 = torch.full((n,), cap, dtype=torch.int32, device=dev)
        csize = torch.zeros(n, dtype=torch.int32, device=dev)

        def call(src_bytes, dst_bytes, decompress=False):
            b = c._batch(raw, off, size32, comp, coff, cap32, csize)
            b.src_bytes, b.dst_bytes = src_bytes, dst_bytes
            fn = c.L.vbz_gpu_decompress_batch if decompress else c.L.vbz_gpu_compress_batch
            rc = fn(c.ctx, ctypes.byref(b), ctypes.byref(opts), 0)
            torch.cuda.synchronize()
            return rc

        assert call(raw.numel(), comp.numel()) == 0
        good = csize.clone()
        assert int(good[0]) > 0
        for sb, db in ((1 << 62, comp.numel()), (raw.numel(), 1 << 62), ((1 << 64) - 1, (1 << 64) - 1), ((1 << 46) + 1, comp.numel())):
            for dec in (False, True):
                csize.zero_()
                assert call(sb, db, dec) == -2
                assert b"not plausible" in c.L.vbz_gpu_last_error(c.ctx)
                assert int(csize.abs().sum()) == 0          # nothing ran
        for field in ("src", "src_off", "src_size", "dst", "dst_off", "dst_cap", "result"):   # a NULL table or arena: refused, not dereferenced
            b = c._batch(raw, off, size32, comp, coff, cap32, csize)
            setattr(b, field, None)
            assert c.L.vbz_gpu_compress_batch(c.ctx, ctypes.byref(b), ctypes.byref(opts), 0) == -2
            assert b"NULL" in c.L.vbz_gpu_last_error(c.ctx)
        csize.zero_()                                      # (a torch call of the same thread: the runtime's last error must be clean)
        assert call(1 << 45, 1 << 45) == -1                 # more than any device has: the allocation fails, nothing else
        assert b"hipMalloc" in c.L.vbz_gpu_last_error(c.ctx)
        probe = torch.zeros(16, device=dev) + 1            # the caller's own HIP work goes on
        assert float(probe.sum()) == 16.0
        csize.zero_()
        assert call(raw.numel(), comp.numel()) == 0 and torch.equal(csize, good)
    c.close()


def test_two_contexts_in_flight_do_not_disturb_each_other():
    """INTEGRATION.md section 4 tells a caller with a queue of batches to keep two contexts busy at once, each on a stream of its own.
    Two contexts code different batches (one of ordinary reads, one with large reads among them) back to back without any
    synchronisation in between, three rounds: each must write exactly the bytes it writes when it has the device to itself, and decode
    its own input back -- nothing of a context (scratch, plans, tables, the hand-over words) may be shared with another."""
    import torch
    from vbz_compression_amd import batch

    dev = torch.device("cuda", 0)
    ctxs = [batch.GpuCodec(0), batch.GpuCodec(0)]
    opts = ctxs[0].options(True, 2, 1, 1)
    L = ctxs[0].L

    def make(c, first, n, stretch):
        with torch.cuda.stream(c.stream):
            lens = c.synth_lengths(5, first, n)
            if stretch:   # a few long reads among the others: per-read routing, its second stream and its join
                lens[7] = 700000
                lens[n // 2] = 1300001
            sizes = lens.to(torch.int64) * 2
            off, total = batch.layout(sizes.cpu(), 64)
            caps = torch.tensor([L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
            coff, ctotal = batch.layout(caps, 64)
            raw = torch.zeros(total, dtype=torch.uint8, device=dev)
            off = off.to(dev)
            c.synth_signal(5, first, raw, off, lens)
            return dict(n=n, raw=raw, off=off, size=sizes.to(torch.int32).to(dev), coff=coff.to(dev), cap=caps.to(torch.int32).to(dev),
                        csize=torch.zeros(n, dtype=torch.int32, device=dev), comp=torch.zeros(ctotal, dtype=torch.uint8, device=dev),
                        back=torch.zeros(total, dtype=torch.uint8, device=dev), res=torch.zeros(n, dtype=torch.int32, device=dev))

    work = [make(ctxs[0], 0, 3000, False), make(ctxs[1], 50000, 2500, True)]

    def step(c, B):
        with torch.cuda.stream(c.stream):
            c.compress(B["raw"], B["off"], B["size"], B["comp"], B["coff"], B["cap"], B["csize"], opts)
            c.decompress(B["comp"], B["coff"], B["csize"], B["back"], B["off"], B["size"], B["res"], opts)

    alone = []
    for c, B in zip(ctxs, work):   # each context with the device to itself
        step(c, B)
        torch.cuda.synchronize()
        assert bool((B["res"] == B["size"]).all()) and torch.equal(B["raw"], B["back"])
        alone.append((B["comp"].clone(), B["csize"].clone()))
    for _ in range(3):             # both in flight, nothing waits for anything
        for B in work:
            B["comp"].zero_()
            B["back"].zero_()
        torch.cuda.synchronize()
        for k in range(2):
            for c, B in zip(ctxs, work):
                step(c, B)
        torch.cuda.synchronize()
        for B, (comp, csize) in zip(work, alone):
            assert torch.equal(B["csize"], csize)
            assert bool((B["res"] == B["size"]).all()) and torch.equal(B["raw"], B["back"])
            for i in (0, 7, B["n"] // 2, B["n"] - 1):   # the frames themselves (the slots' unused tails are scratch)
                o, z = int(B["coff"][i]), int(csize[i])
                assert torch.equal(B["comp"][o : o + z], comp[o : o + z])
    for c in ctxs:
        c.close()


def test_batched_contexts_on_threads_of_their_own():
    """Four host threads, each with a context of its own created ON that thread, run batched compress / decompress calls at the same
    time (ctypes releases the interpreter lock inside a call): per-process state of the library -- the sequence tables it builds
    once, the kernel images, the pool the single-buffer API keeps -- must not care.  Every thread checks its own round trips and a
    frame of each round against the oracle."""
    import threading

    import torch
    from vbz_compression_amd import batch

    errors = []

    # (the helper functions of gpu_util drive ONE shared context; this test needs one per thread: the plain tensor interface)
    def worker(k):
        try:
            c = batch.GpuCodec(0)
            o = [(True, 2, 1, 1), (True, 2, 1, 0), (False, 4, 3, 0), (True, 2, 3, 1)][k]
            opts = c.options(*o)
            size = o[1]
            dev = c.device
            with torch.cuda.stream(c.stream):
                for rnd in range(5):
                    n = 300 + 50 * k
                    lens = c.synth_lengths(5 + k, 1000 * rnd, n)
                    if rnd == 2:
                        lens[5] = 400000
                    sizes = lens.to(torch.int64) * size
                    off, total = batch.layout(sizes.cpu(), 64)
                    caps = torch.tensor([c.L.vbz_max_compressed_size(int(z), ctypes.byref(opts)) for z in sizes.cpu().tolist()], dtype=torch.int64)
                    coff, ctotal = batch.layout(caps, 64)
                    raw = torch.zeros(total, dtype=torch.uint8, device=dev)
                    off = off.to(dev)
                    (c.synth_u32 if size == 4 else c.synth_signal)(5 + k, 1000 * rnd, raw, off, lens)
                    comp = torch.zeros(ctotal, dtype=torch.uint8, device=dev)
                    coff = coff.to(dev)
                    csize = torch.zeros(n, dtype=torch.int32, device=dev)
                    back = torch.zeros_like(raw)
                    res = torch.zeros(n, dtype=torch.int32, device=dev)
                    size32 = sizes.to(torch.int32).to(dev)
                    c.compress(raw, off, size32, comp, coff, caps.to(torch.int32).to(dev), csize, opts)
                    c.decompress(comp, coff, csize, back, off, size32, res, opts)
                    c.stream.synchronize()
                    assert bool((res == size32).all()) and torch.equal(raw, back), "round trip in thread %d round %d" % (k, rnd)
                    i = 5
                    o0, z0 = int(off[i]), int(sizes[i])
                    f = comp[int(coff[i]) : int(coff[i]) + int(csize[i])].cpu().numpy()
                    want = O.decompress(f, z0, O.options(*o))
                    assert not isinstance(want, int) and want.view(np.uint8).tobytes() == raw[o0 : o0 + z0].cpu().numpy().tobytes()
            c.close()
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_zstd_encoder_tables_match_host_statement():
    """The wave-parallel table construction on the device (bitonic sort, package-merge, FSE-coded weights) must give the
    tree description that its serial host statement gives for the same bytes (zstd_entropy.h: huf_build_pm, checked for
    optimality on the CPU; huf_write_tree, checked byte-for-byte against libzstd).  Large regions: see below."""
    import entropy_host as E
    import gpu_util as G

    rng = np.random.default_rng(77)
    regions = []
    for i, n in enumerate([300, 5000, 40000, 100000, 131072, 7777, 65536, 2000]):
        a = O.synth_signal(5, i, n)
        regions.append(O.svb_compress(a, 2, True, 0)[(n + 3) // 4 :].copy())           # data bytes of a signal
    regions.append(np.minimum(rng.geometric(0.05, 90000), 255).astype(np.uint8))
    regions.append(np.clip(rng.normal(128, 4, 50000), 0, 255).astype(np.uint8))
    regions.append((rng.integers(0, 100, 30000) < 4).astype(np.uint8) * 16)               # two symbols
    frames = G.zstd_compress(regions)
    checked = sampled = 0
    for data, f in zip(regions, frames):
        assert not isinstance(f, int)
        lit = E.parse_first_block_literals(f)
        if lit is None or lit[0] != 2:
            continue   # the encoder stored this region raw (too small to pay for a table)
        log, nb, tree = E.tree_description(data, package_merge=True)
        if len(data) < (32 << 10):
            assert lit[3] == tree, len(data)
        else:
            # regions of 32 KB and more may be coded from a histogram of a quarter of their bytes (zstd_encode.hip:
            # region_histogram): the code must still have a word for every byte that occurs and cost at most 0.2 % more
            mine = E.weights_from_tree(lit[3]).astype(np.int64)
            cnt = np.bincount(data, minlength=256).astype(np.int64)
            assert (mine[cnt > 0] > 0).all() and mine.max() <= 11
            assert sum(2.0 ** -int(x) for x in mine if x) == 1.0
            assert (cnt * mine).sum() <= 1.002 * (cnt * nb.astype(np.int64)).sum(), len(data)
            sampled += lit[3] != tree
        checked += 1
    assert checked >= 8 and sampled >= 2


def test_hdf5_filter_32020_calling_convention():
    """The plugin's H5Z filter function with libhdf5's calling convention (vbz_plugin.cpp:97-229):
    chunks it writes decode with the oracle's filter, chunks the reference wrote decode with it."""
    from vbz_compression_amd import _lib

    P = ctypes.CDLL(_lib.PLUGIN_PATH)
    sz, vp = ctypes.c_size_t, ctypes.c_void_p
    P.vbz_filter.restype = sz
    P.vbz_filter.argtypes = [ctypes.c_uint, sz, ctypes.POINTER(ctypes.c_uint), sz, ctypes.POINTER(sz), ctypes.POINTER(vp)]
    libc = ctypes.CDLL(None)
    libc.malloc.restype = vp
    libc.malloc.argtypes = [sz]
    libc.free.argtypes = [vp]

    def run(fn, flags, cd, data):
        buf = libc.malloc(max(len(data), 1))
        ctypes.memmove(buf, data.ctypes.data, len(data))
        pbuf, size = vp(buf), sz(len(data))
        cdv = (ctypes.c_uint * len(cd))(*cd)
        used = fn(flags, len(cd), cdv, len(data), ctypes.byref(size), ctypes.byref(pbuf))
        out = None
        if used:
            out = np.ctypeslib.as_array(ctypes.cast(pbuf, ctypes.POINTER(ctypes.c_uint8)), (used,)).copy()
        libc.free(pbuf)
        return used, out

    a = O.synth_signal(5, 4, 123627)
    raw = np.frombuffer(a.tobytes(), np.uint8)
    for cd in ([1, 2, 1, 1], [0, 2, 1, 1], [0, 2, 1], [1, 2, 1, 0], [0, 2, 1, 1, 1]):
        used, chunk = run(P.vbz_filter, 0, cd, raw)
        assert used > 0 and int(chunk[:4].view("<u4")[0]) == len(raw)
        u2, back = run(O.lib().vbo_filter, 0x100, cd, chunk)       # the reference's decoder reads it
        assert u2 == len(raw) and back.tobytes() == raw.tobytes()
        u3, chunk_ref = run(O.lib().vbo_filter, 0, cd, raw)          # and the plugin reads the reference's chunk
        u4, back2 = run(P.vbz_filter, 0x100, cd, chunk_ref)
        assert u4 == len(raw) and back2.tobytes() == raw.tobytes()
    assert run(P.vbz_filter, 0, [1, 2], raw)[0] == 0                  # cd_nelmts < 3
    assert run(P.vbz_filter, 0, [1, 2, 1, 1], raw[:-1])[0] == 0        # size not a multiple of integer_size
    assert run(P.vbz_filter, 0x100, [1, 2, 1, 1], raw[:1000])[0] == 0  # not a vbz chunk
    # config 3 stand-in: the 10 real reads of the shipped fast5 file through the filter, both directions
    idx = json.load(open(os.path.join(GOLDEN, "fast5_chunks.json")))
    blob = np.fromfile(os.path.join(GOLDEN, "fast5_chunks.bin"), np.uint8)
    total_ref = total_gpu = 0
    for e in idx:
        chunk = blob[e["chunk_offset"] : e["chunk_offset"] + e["chunk_size"]]
        cd = e["filter_v0"][1]
        u, sig = run(P.vbz_filter, 0x100, cd, chunk)
        assert u == 2 * e["samples"] and hashlib.sha256(sig.tobytes()).hexdigest() == e["raw_sha256"]
        u, mine = run(P.vbz_filter, 0, cd[:4], sig)
        u2, back = run(O.lib().vbo_filter, 0x100, cd[:4], mine)
        assert back.tobytes() == sig.tobytes()
        # per-chunk svb payload byte-identical to the reference's (recovered by un-zstd-ing both)
        svb_mine = O.zstd_decompress(mine[4:], int(O.zstd_content_size(mine[4:])))
        svb_ref = O.zstd_decompress(chunk[4:], int(O.zstd_content_size(chunk[4:])))
        assert svb_mine.tobytes() == svb_ref.tobytes()
        total_ref += len(chunk)
        total_gpu += len(mine)
    print("real reads: gpu %d B, reference %d B" % (total_gpu, total_ref))
    assert abs(total_gpu / total_ref - 1.0) < 0.01, (total_gpu, total_ref)


@pytest.mark.gpu
def test_fast5_bulk_repacker(tmp_path):
    """the reference's fast5 conversion (python/fast5compress/fast5vbz.py:17-55) done in bulk -- every
    read_*/Raw/Signal of a multi-read file coded in one batched GPU call and stored with H5Dwrite_chunk.  The file is
    the reference's own test file (python/test/test_vbz_filter.py:57-73 reads the same reads).  What must hold: the
    samples survive (sha256 pinned by the golden index), every stored chunk is a sized VBZ buffer the reference path
    decodes, sizes are within 1 % of the reference's, and -d brings the file back to gzip."""
    import hashlib
    import shutil

    from vbz_compression_amd import fast5

    src = str(tmp_path / "reads.fast5")
    shutil.copy(os.path.join(GOLDEN, "multi_fast5_zip.fast5"), src)
    idx = {e["read"]: e for e in json.load(open(os.path.join(GOLDEN, "fast5_chunks.json")))}
    try:
        before = fast5.list_fast5(src, export_signal=str(tmp_path / "sig0"))
    except fast5.Hdf5NotFound:
        pytest.skip("no libhdf5 >= 1.10.3 on this box")
    assert len(before) == 10 and all(r["filters"] == [1] and r["integer_size"] == 2 for r in before)
    sig = np.fromfile(str(tmp_path / "sig0"), np.int16)
    pos, signals = 0, {}
    for r in before:
        signals[r["name"]] = sig[pos : pos + r["samples"]]
        pos += r["samples"]
        assert hashlib.sha256(signals[r["name"]].tobytes()).hexdigest() == idx[r["name"]]["raw_sha256"]
    for version in (1, 0):
        out = fast5.compress_fast5(src, ".v%d" % version, vbz_version=version)
        assert out == src + ".v%d" % version
        after = fast5.list_fast5(out, export_chunks=str(tmp_path / "chunks"))
        assert [r["name"] for r in after] == [r["name"] for r in before]
        chunks = np.fromfile(str(tmp_path / "chunks"), np.uint8)
        pos = 0
        oo = O.options(True, 2, 1, version)
        for r, r0 in zip(after, before):
            assert r["filters"] == [32020] and r["samples"] == r0["samples"] and r["fnv1a64"] == r0["fnv1a64"]
            assert r["chunk_bytes"] == r["stored_bytes"] > 0
            chunk = chunks[pos : pos + r["chunk_bytes"]]
            pos += r["chunk_bytes"]
            want = signals[r["name"]]
            back = O.decompress(chunk, want.nbytes, oo, sized=True)        # the reference path reads what was written
            assert not isinstance(back, int) and back.tobytes() == want.tobytes()
            ref = O.compress(want, oo, sized=True)
            assert abs(len(chunk) - len(ref)) <= 0.01 * len(ref) + 16, (r["name"], len(chunk), len(ref))
            assert abs(len(chunk) - idx[r["name"]]["chunk_size"]) <= 0.01 * idx[r["name"]]["chunk_size"] + 16  # the shipped vbz file
    # and back to gzip (fast5vbz.py -d), reading the vbz chunks directly
    back = fast5.compress_fast5(src + ".v1", ".gz", decompress=True)
    again = fast5.list_fast5(back)
    assert [(r["name"], r["filters"], r["fnv1a64"]) for r in again] == [(r["name"], [1], r["fnv1a64"]) for r in before]


@pytest.mark.gpu
def test_fast5_repacker_pipelines_many_files(tmp_path):
    """The reference's users convert many files side by side (README.md:36-40 `xargs -P 10 ... h5repack`, fast5vbz.py:72-75 one file
    after the other).  `vbz_fast5_repack a b c` is a pipeline: file k + 1 is copied, read and inflated and file k - 1 stored while the
    GPU codes file k.  Three copies of the reference's test file in one call: every read's samples survive (sha256 pinned by the golden
    index), every chunk is a sized VBZ buffer the reference path decodes -- the same bytes whichever position the file had --, the
    names come back in input order, and three files take less than twice the time of one."""
    import hashlib
    import shutil
    import time

    from vbz_compression_amd import fast5

    idx = {e["read"]: e for e in json.load(open(os.path.join(GOLDEN, "fast5_chunks.json")))}
    files = []
    for i in range(4):
        f = str(tmp_path / ("reads%d.fast5" % i))
        shutil.copy(os.path.join(GOLDEN, "multi_fast5_zip.fast5"), f)
        files.append(f)
    try:
        fast5.file_samples(files[:1])
    except fast5.Hdf5NotFound:
        pytest.skip("no libhdf5 >= 1.10.3 on this box")
    fast5.compress_many(files[3:], ".warm", vbz_version=1)      # (the first call of a process pays for the page-in of the libraries)
    t0 = time.perf_counter()
    one = fast5.compress_many(files[:1], ".one", vbz_version=1)
    t1 = time.perf_counter()
    three = fast5.compress_many(files[:3], ".vbz", vbz_version=1)
    t2 = time.perf_counter()
    assert one == [files[0] + ".one"] and three == [f + ".vbz" for f in files[:3]]
    print("one file %.0f ms, three files %.0f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    assert (t2 - t1) < 2.0 * (t1 - t0), (t1 - t0, t2 - t1)
    oo = O.options(True, 2, 1, 1)
    first = None
    for out in three:
        listing = fast5.list_fast5(out, export_chunks=str(tmp_path / "chunks"), export_signal=str(tmp_path / "sig"))
        assert len(listing) == 10 and all(r["filters"] == [32020] for r in listing)
        chunks = np.fromfile(str(tmp_path / "chunks"), np.uint8)
        sig = np.fromfile(str(tmp_path / "sig"), np.int16)
        cp = sp = 0
        digest = hashlib.sha256(chunks.tobytes()).hexdigest()
        first = first or digest
        assert digest == first                                    # the same chunks in every copy
        for r in listing:
            a = sig[sp : sp + r["samples"]]
            sp += r["samples"]
            assert hashlib.sha256(a.tobytes()).hexdigest() == idx[r["name"]]["raw_sha256"]
            chunk = chunks[cp : cp + r["chunk_bytes"]]
            cp += r["chunk_bytes"]
            back = O.decompress(chunk, a.nbytes, oo, sized=True)   # the reference path reads what was written
            assert not isinstance(back, int) and back.tobytes() == a.tobytes()
    # ... and back to gzip, three files in one call
    back = fast5.compress_many(three, ".gz", decompress=True)
    for b, out in zip(back, three):
        assert [(r["name"], r["filters"], r["fnv1a64"]) for r in fast5.list_fast5(b)] == [(r["name"], [1], r["fnv1a64"]) for r in fast5.list_fast5(out)]


@pytest.mark.gpu
def test_h5repack_through_filter_32020(tmp_path):
    """BASELINE.json configs[2]: test_data/multi_fast5_zip.fast5 re-packed by libhdf5's own h5repack with this
    library's plugin on HDF5_PLUGIN_PATH (UD=32020,0,4,0,2,1,1: the 1.10.6 syntax) -- libhdf5 loads
    libvbz_hdf_plugin.so and calls vbz_filter once per chunk.  h5diff must find no difference, every chunk must be a sized
    VBZ buffer the reference path decodes, of the reference's size within 1 %, as is what the bulk
    re-packer writes for the same read (one encoder behind both boundaries)."""
    import shutil
    import subprocess

    from vbz_compression_amd import _lib, fast5

    h5repack = shutil.which("h5repack") or "/opt/conda/bin/h5repack"
    h5diff = shutil.which("h5diff") or "/opt/conda/bin/h5diff"
    if not (os.path.exists(h5repack) and os.path.exists(h5diff)):
        pytest.skip("no h5repack / h5diff on this box")
    src = str(tmp_path / "reads.fast5")
    out = str(tmp_path / "repacked.fast5")
    shutil.copy(os.path.join(GOLDEN, "multi_fast5_zip.fast5"), src)
    env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.dirname(_lib.LIB_PATH))
    r = subprocess.run([h5repack, "-f", "UD=32020,0,4,0,2,1,1", src, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True)
    assert r.returncode == 0 and "error" not in r.stdout.lower(), r.stdout[-2000:]
    d = subprocess.run([h5diff, src, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert d.returncode == 0, d.stdout[-2000:]
    try:
        before = fast5.list_fast5(src, export_signal=str(tmp_path / "sig"))
        after = fast5.list_fast5(out, export_chunks=str(tmp_path / "chunks"))
        bulk = fast5.list_fast5(fast5.compress_fast5(src, ".bulk", vbz_version=0), export_chunks=str(tmp_path / "bulk_chunks"))
    except fast5.Hdf5NotFound:
        pytest.skip("no libhdf5 >= 1.10.3 for the listing tool")
    sig = np.fromfile(str(tmp_path / "sig"), np.int16)
    chunks = np.fromfile(str(tmp_path / "chunks"), np.uint8)
    bulk_chunks = np.fromfile(str(tmp_path / "bulk_chunks"), np.uint8)
    assert [(r["name"], r["fnv1a64"]) for r in after] == [(r["name"], r["fnv1a64"]) for r in before]
    oo = O.options(True, 2, 1, 0)
    spos = cpos = bpos = 0
    for r, rb in zip(after, bulk):
        assert r["filters"] == [32020] and r["chunk_bytes"] == r["stored_bytes"] > 0
        want = sig[spos : spos + r["samples"]]
        chunk = chunks[cpos : cpos + r["chunk_bytes"]]
        bchunk = bulk_chunks[bpos : bpos + rb["chunk_bytes"]]
        spos += r["samples"]
        cpos += r["chunk_bytes"]
        bpos += rb["chunk_bytes"]
        ref = O.compress(want, oo, sized=True)
        for ch in (chunk, bchunk):
            back = O.decompress(ch, want.nbytes, oo, sized=True)
            assert not isinstance(back, int) and back.tobytes() == want.tobytes()
            assert abs(len(ch) - len(ref)) <= 0.01 * len(ref) + 16, (r["name"], len(ch), len(ref))
        # (the bytes of the two need not be the same: which kernels code a read -- one wavefront, or spans -- follows the shape
        # of the call it arrives in: one chunk per filter call here, a batch of ten in the bulk tool)


@pytest.mark.gpu
def test_hdf5_write_read_through_registered_filter(tmp_path):
    """The reference's C++ HDF5 integration test (vbz_plugin/test/vbz_hdf_plugin_test.cpp:15-136) against this plugin and a
    real libhdf5: H5Zregister(vbz_plugin_info()), then H5Dwrite / H5Dread of {int,uint}{8,16,32} datasets chunked count / 8
    with filter 32020 (iota at level 5, random values at level 1, version 1)."""
    import subprocess

    from vbz_compression_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "h5_filter_roundtrip")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "host", "h5_filter_roundtrip.cpp"),
                           "-L", libdir, "-lvbz_hdf_plugin", "-lvbz_hip", "-ldl", "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe, str(tmp_path / "test_file.h5")], capture_output=True, text=True, timeout=600)
    if r.returncode == 3:
        pytest.skip("no libhdf5 on this box")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.gpu
def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver and the judge read (metric, value, unit, n_gpus, steps,
    warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline{bound, achieved,
    peak, unit, frac, traffic}); a small batch keeps this a test, not a measurement."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "512", "--steps", "3", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "ratio"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "MB/s" and d["dtype"] == "int16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["reads_per_step"] == 512 and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert d["value"] > 0 and 2.3 < d["ratio"] < 2.5
    # (2 + c) bytes per sample per direction, an end-to-end entry, the svb-only stage line, the PCIe-inclusive rate
    assert abs(rf["algorithmic_bytes_per_sample"] - (2 + 2 / d["ratio"])) < 1e-3
    for k in ("encode", "decode"):
        assert 0 < rf["per_direction"][k]["frac"] < 1
    assert 0 < rf["end_to_end"]["frac"] < 1
    assert d["stages"]["svb_only"]["encode_MBps"] > d["encode_MBps"] and 1.2 < d["stages"]["svb_only"]["svb_bytes_per_sample"] < 1.3
    assert d["host_resident"]["round_trip_ok"] and 0 < d["host_resident"]["encode_decode_MBps"] < d["value"]
    assert d["config"]["distinct_reads"] >= 512


def test_many_tiny_reads_in_one_batch():
    """150 000 reads of 0 ... 300 samples in ONE call (more workgroups than a 16-bit grid index holds, 147 workgroups of the
    slot plan): every read round-trips on the device, a sample of 600 of them is byte-checked against the oracle in both
    directions (level 0: identical bytes; level 1: each side decodes the other's frames)."""
    import gpu_util as G
    from vbz_compression_amd import _lib

    rng = np.random.default_rng(150)
    n = 150_000
    base = [O.synth_signal(5, i, int(k)) for i, k in enumerate(rng.integers(0, 301, 512))]
    base[0] = np.zeros(0, np.int16)
    reads = [base[int(j)] for j in rng.integers(0, len(base), n)]
    pick = rng.choice(n, 600, replace=False)
    for level in (0, 1):
        go, oo = _lib.CompressionOptions(True, 2, level, 1), O.options(True, 2, level, 1)
        comp = G.compress(reads, go)
        assert all(not isinstance(f, int) for f in comp), "a read was refused"
        back = G.decompress(comp, [a.nbytes for a in reads], go)
        for a, b in zip(reads, back):
            assert not isinstance(b, int) and b.tobytes() == a.tobytes()
        for i in pick:
            a = reads[int(i)]
            want = O.compress(a, oo)
            if level == 0:
                assert comp[int(i)].tobytes() == want.tobytes()
            else:
                assert O.decompress(comp[int(i)], a.nbytes, oo).tobytes() == a.tobytes()
        mine = G.decompress([O.compress(reads[int(i)], oo) for i in pick], [reads[int(i)].nbytes for i in pick], go)
        for i, b in zip(pick, mine):
            assert not isinstance(b, int) and b.tobytes() == reads[int(i)].tobytes()


def test_weights_alphabet_beyond_eleven_is_refused_on_every_decoder_path():
    """tests/test_oracle_zstd.py::test_weights_alphabet_ends_at_eleven on the device: the frame whose damaged tree description lists
    weights 12 and 13 (and still decodes consistently) is refused like libzstd refuses it -- by the batched own-frame decoder's
    weights kernel, by the one-wavefront decoder's wave-parallel reader and by the span path's (a handful of reads per call) --, the
    undamaged frame decodes, and bit flips in the description never make the device accept what libzstd refuses."""
    import gpu_util as G
    from vbz_compression_amd import _lib

    z = np.load(os.path.join(GOLDEN, "weights_alphabet_beyond_11.npz"))
    good, bad = z["original"], z["damaged"]
    opts = _lib.CompressionOptions(True, 2, 1, 0)
    oo = O.options(True, 2, 1, 0)
    want = O.decompress(good, 40000, oo)
    assert not isinstance(want, int) and isinstance(O.decompress(bad, 40000, oo), int)
    rng = np.random.default_rng(6)
    flips = []
    for _ in range(200):
        b = good.copy()
        for _ in range(int(rng.integers(1, 3))):
            b[int(rng.integers(23, 33))] ^= 1 << int(rng.integers(0, 8))
        flips.append(b)
    for env in ({}, {"VBZ_HIP_FAST_DECODE": "0"}, {"VBZ_HIP_SEGMENTED": "1"}):
        for batch_of in (1, 40, 3000):      # the span path for a handful of reads, the batched decoder, calls that walk chains
            frames = ([good, bad] + flips)[:max(2, min(batch_of, 202))] * (1 if batch_of <= 202 else 15)
            got = _run_decode_in_env(frames, 40000, env)
            for f, g in zip(frames, got):
                lz = O.decompress(f, 40000, oo)
                if isinstance(g, int):
                    continue                                     # (stricter than libzstd is allowed)
                assert not isinstance(lz, int) and lz.tobytes() == g.tobytes(), (env, batch_of)
            assert not isinstance(got[0], int) and got[0].tobytes() == want.tobytes()
            assert isinstance(got[1], int)


def _run_decode_in_env(frames, nbytes, env):
    """decompress `frames` in a process of its own with `env` added to the environment; returns arrays / error codes"""
    import pickle
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "in.pkl"), "wb") as fh:
            pickle.dump(([np.asarray(f) for f in frames], nbytes), fh)
        code = ("import pickle, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "import gpu_util as G\nfrom vbz_compression_amd import _lib\n"
                "frames, nb = pickle.load(open(%r, 'rb'))\n"
                "got = G.decompress(frames, [nb] * len(frames), _lib.CompressionOptions(True, 2, 1, 0))\n"
                "pickle.dump(got, open(%r, 'wb'))\n") % (root, os.path.join(root, "tests"), os.path.join(td, "in.pkl"), os.path.join(td, "out.pkl"))
        subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), check=True, capture_output=True)
        with open(os.path.join(td, "out.pkl"), "rb") as fh:
            return pickle.load(fh)


def test_zstd_decode_libzstd_frames_of_several_blocks():
    """Frames the reference writes for 4-byte integers: svb streams of 150 ... 400 KB, i.e. several 128 KB blocks per frame,
    whose sequence chains hand their repeat offsets (and, in Repeat_Mode, their tables) from block to block -- the state
    the decoder keeps in lane 0 only.  Every data shape of the soak tool, both integer types, each frame in a call of its own
    and all of them in one batch: the device must decode what the reference path (oracle + libzstd 1.4.8) wrote, bit for bit.
    (A soak run found a decoder variant that started later blocks from stale lanes' repeat offsets; one-block int16 frames,
    which is what the other tests send, do not notice.)"""
    import sys

    import gpu_util as G
    from vbz_compression_amd import _lib

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak

    rng = np.random.default_rng(141)
    cases = []
    for dt in (np.uint32, np.int32):
        for kind in (0, 1, 3, 4, 5, 6):
            for n in (100003, 70001):
                cases.append(soak.make_read(rng, dt, kind, n))
    for level in (1, 3):
        go, oo = _lib.CompressionOptions(True, 4, level, 1), O.options(True, 4, level, 1)
        frames = [O.compress(a, oo) for a in cases]
        for a, f in zip(cases, frames):
            b = G.decompress([f], [a.nbytes], go)[0]
            assert not isinstance(b, int) and b.tobytes() == a.tobytes(), (a.dtype, len(a), level)
        back = G.decompress(frames, [a.nbytes for a in cases], go)
        for a, b in zip(cases, back):
            assert not isinstance(b, int) and b.tobytes() == a.tobytes(), (a.dtype, len(a), level, "batch")


def test_descriptor_tables_are_checked_against_the_declared_arenas():
    """include/vbz_gpu.h: the descriptor table is untrusted like the data.  A read whose source slot does not lie inside
    [0, src_bytes) gets VBZ_INPUT_SIZE_ERROR, one whose destination slot does not lie inside [0, dst_bytes) gets
    VBZ_DESTINATION_SIZE_ERROR (64-bit arithmetic: offsets near 2^64 do not wrap), the other reads of the batch are coded as
    usual and nothing outside the arenas is touched (canaries behind both arenas)."""
    import torch

    import gpu_util as G
    from vbz_compression_amd import _lib, batch

    c = G.codec()
    dev = c.device
    L = _lib.load()
    reads = [O.synth_signal(5, 40 + i, 3000 + 500 * i) for i in range(8)]
    for sized in (False, True):
        opts = c.options(True, 2, 1, 1)
        arena, off, sizes = G._pack(reads)
        src = torch.from_numpy(arena).to(dev)
        caps = [L.vbz_max_compressed_size(a.nbytes, ctypes.byref(opts)) for a in reads]
        doff, dtotal = batch.layout([x + 32 for x in caps], 64)
        canary = 4096
        dst_all = torch.full((dtotal + 64 + canary,), 0xA5, dtype=torch.uint8, device=dev)
        dst = dst_all[: dtotal + 64]
        src_off = off.clone()
        src_size = torch.tensor(sizes, dtype=torch.int64)
        dst_off = doff.clone()
        dst_cap = torch.tensor(caps, dtype=torch.int64)
        src_off[1] = src.numel() - 10                 # sticks out of the source arena
        src_off[2] = (1 << 62)                        # nowhere near it
        src_size[3] = 0xFFFFFFF0                      # a size that is no size
        dst_off[4] = dst.numel() - 8                  # slot sticks out of the destination arena
        dst_cap[5] = 0xFFFFFF00                       # capacity beyond the arena
        dst_off[6] = -16                              # 2^64 - 16: offset + capacity would wrap
        want = {1: _lib.VBZ_INPUT_SIZE_ERROR, 2: _lib.VBZ_INPUT_SIZE_ERROR, 3: _lib.VBZ_INPUT_SIZE_ERROR,
                4: _lib.VBZ_DESTINATION_SIZE_ERROR, 5: _lib.VBZ_DESTINATION_SIZE_ERROR, 6: _lib.VBZ_DESTINATION_SIZE_ERROR}

        def i32(t):
            return torch.where(t >= 2**31, t - 2**32, t).to(torch.int32).to(dev)

        result = torch.full((len(reads),), -8, dtype=torch.int32, device=dev)
        c.compress(src, src_off.to(dev), i32(src_size), dst, dst_off.to(dev), i32(dst_cap), result, opts, sized=sized)
        torch.cuda.synchronize()
        res = [int(x) & 0xFFFFFFFF for x in result.cpu().tolist()]
        host = dst_all.cpu().numpy()
        assert (host[dtotal + 64:] == 0xA5).all()
        frames = {}
        for i, a in enumerate(reads):
            if i in want:
                assert res[i] == want[i], (sized, i, hex(res[i]))
            else:
                assert not _lib.is_error(res[i])
                f = host[int(doff[i]) : int(doff[i]) + res[i]].copy()
                assert O.decompress(f, a.nbytes, O.options(True, 2, 1, 1), sized=sized).tobytes() == a.tobytes()
                frames[i] = f
        # the same table checks on the way back
        good = sorted(frames)
        fr = [frames[i] for i in good]
        farena, foff, fsizes = G._pack(fr)
        fsrc = torch.from_numpy(farena).to(dev)
        ocaps = [reads[i].nbytes for i in good]
        ooff, ototal = batch.layout([x + 32 for x in ocaps], 64)
        out_all = torch.full((ototal + 64 + canary,), 0x5A, dtype=torch.uint8, device=dev)
        out = out_all[: ototal + 64]
        fo = foff.clone()
        oo = ooff.clone()
        fo[0] = fsrc.numel() + 5
        oo[1] = out.numel() - 3
        result = torch.full((len(good),), -8, dtype=torch.int32, device=dev)
        c.decompress(fsrc, fo.to(dev), torch.tensor(fsizes, dtype=torch.int32, device=dev), out, oo.to(dev),
                     torch.tensor(ocaps, dtype=torch.int32, device=dev), result, opts, sized=sized)
        torch.cuda.synchronize()
        res = [int(x) & 0xFFFFFFFF for x in result.cpu().tolist()]
        host = out_all.cpu().numpy()
        assert (host[ototal + 64:] == 0x5A).all()
        assert res[0] == _lib.VBZ_INPUT_SIZE_ERROR and res[1] == _lib.VBZ_DESTINATION_SIZE_ERROR
        for k in range(2, len(good)):
            a = reads[good[k]]
            assert res[k] == a.nbytes and host[int(ooff[k]) : int(ooff[k]) + a.nbytes].tobytes() == a.tobytes()


def test_batched_own_frame_decoder_and_its_fallbacks():
    """zstd_decode_fast.hip decodes frames of the shape this library writes with one lane per frame for the headers and one
    wavefront per frame for the streams; everything else -- frames libzstd wrote, frames with a window descriptor or raw blocks,
    damaged frames -- must come out of the same call exactly as the one-wavefront decoder (VBZ_HIP_FAST_DECODE=0) gives it:
    same bytes, same verdicts, in one batch, with both kinds of frame next to each other."""
    import subprocess
    import sys

    import gpu_util as G

    rng = np.random.default_rng(77)
    reads = [O.synth_signal(5, 300 + i, n) for i, n in enumerate([100000, 90001, 33333, 4096, 17, 0, 1, 250000])]
    reads.append(np.zeros(120000, np.int16))                                     # RLE / raw blocks: not the fast shape
    reads.append(rng.integers(-32768, 32767, 60000, endpoint=True).astype(np.int16))   # incompressible: raw blocks
    reads.append(np.tile(O.synth_signal(5, 1, 7000), 20))                        # repeated template: general sequences
    opts = G.codec().options(True, 2, 1, 1)
    os.environ["VBZ_HIP_SEGMENTED"] = "0"   # (a handful of reads would take the large-read path otherwise)
    code = r"""
import os, sys, pickle
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gpu_util as G, oracle_lib as O
reads, = pickle.load(open(sys.argv[1], 'rb'))
opts = G.codec().options(True, 2, 1, 1)
own = G.compress(reads, opts)
ref = [O.compress(a, O.options(True, 2, 1, 1)) for a in reads]
frames, caps = [], []
for a, f, g in zip(reads, own, ref):
    frames += [f, g]; caps += [a.nbytes, a.nbytes]
# damaged copies of own frames: a flipped byte in a stream, a truncated frame, a flipped byte in the tree description
for k in (0, 1, 2):
    f = own[k].copy(); f[len(f) // 2] ^= 0x40; frames.append(f); caps.append(reads[k].nbytes)
    f = own[k][: len(own[k]) - 7].copy(); frames.append(f); caps.append(reads[k].nbytes)
    f = own[k].copy(); f[40] ^= 0x08; frames.append(f); caps.append(reads[k].nbytes)
    frames.append(own[k].copy()); caps.append(reads[k].nbytes - 2)      # destination too small
out = G.decompress(frames, caps, opts)
pickle.dump([o if isinstance(o, int) else o.tobytes() for o in out], open(sys.argv[2], 'wb'))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    import pickle
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        pickle.dump((reads,), open(os.path.join(td, "in.pkl"), "wb"))
        outs = {}
        for fast in ("1", "0"):
            env = dict(os.environ, VBZ_HIP_FAST_DECODE=fast, VBZ_HIP_SEGMENTED="0")
            subprocess.run([sys.executable, "-c", code, os.path.join(td, "in.pkl"), os.path.join(td, "out%s.pkl" % fast)], check=True, env=env)
            outs[fast] = pickle.load(open(os.path.join(td, "out%s.pkl" % fast), "rb"))
    os.environ.pop("VBZ_HIP_SEGMENTED", None)
    assert len(outs["1"]) == len(outs["0"]) == 2 * len(reads) + 12
    for i, (a, b) in enumerate(zip(outs["1"], outs["0"])):
        assert a == b, i
    for i, a in enumerate(reads):   # and both are right
        assert outs["1"][2 * i] == a.tobytes() and outs["1"][2 * i + 1] == a.tobytes()
    assert sum(isinstance(o, int) for o in outs["1"][2 * len(reads):]) >= 9


def test_walked_chains_of_reference_frames():
    """zstd_decode_ref.hip walks the sequence chains of frames the reference wrote (libzstd: FSE-coded sequences, Repeat_Mode tables,
    several blocks) one lane per frame and hands the records to the one-wavefront decoder.  The call must give exactly what it gives
    with VBZ_HIP_REF_CHAINS=0 (2 = walk whatever the size of the call; by default calls of 2 560 reads and more do) -- same bytes, same verdicts, damaged frames included -- the walked frames must be the ones expected
    (vbz_gpu_decode_paths), and the bytes must be the reference's."""
    import pickle
    import subprocess
    import sys
    import tempfile

    if O.lib().vbo_zstd_version() is None:
        pytest.skip("no libzstd on this box")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak

    rng = np.random.default_rng(4242)
    reads = [O.synth_signal(5, 900 + i, n) for i, n in enumerate([100000, 110000, 104000, 250000, 400000, 450000, 33333, 700, 17, 1, 0])]
    reads.append(np.tile(O.synth_signal(5, 1, 7000), 20))                                  # repeated template: long matches
    reads.append(np.zeros(120000, np.int16))                                               # RLE blocks, no sequences
    reads.append(rng.integers(-32768, 32767, 60000, endpoint=True).astype(np.int16))       # raw blocks
    for kind in (0, 1, 3, 4, 5, 6):
        reads.append(soak.make_read(rng, np.int16, kind, 90001))
    wide = [soak.make_read(rng, dt, kind, 100003) for dt in (np.uint32, np.int32) for kind in (0, 3, 5)]   # several blocks per frame
    code = r"""
import os, sys, pickle
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gpu_util as G, oracle_lib as O
from vbz_compression_amd import _lib
reads, wide, frames16, frames32, zoo, zoo_n, bad, bad_n = pickle.load(open(sys.argv[1], 'rb'))
out = {}
caps16 = [a.nbytes for a in reads] * (len(frames16) // len(reads))
out['i16'] = G.decompress(frames16, caps16, _lib.CompressionOptions(True, 2, 1, 1))
out['i16_paths'] = G.codec().decode_paths()
out['i16_ahead'] = G.codec().decode_literals_ahead()
# both kinds of frame in one call: the walk runs beside the batched decoder's launches for this library's frames
own = G.compress(reads, _lib.CompressionOptions(True, 2, 1, 1))
mix, mixcaps = [], []
for a, f, g in zip(reads, own, frames16[: len(reads)]):
    mix += [f, g]; mixcaps += [a.nbytes, a.nbytes]
out['mix'] = G.decompress(mix, mixcaps, _lib.CompressionOptions(True, 2, 1, 1))
out['mix_paths'] = G.codec().decode_paths()
caps32 = [a.nbytes for a in wide] * (len(frames32) // len(wide))
out['i32'] = G.decompress(frames32, caps32, _lib.CompressionOptions(True, 4, 1, 1))
out['i32_paths'] = G.codec().decode_paths()
out['zoo'] = G.zstd_decompress(zoo, zoo_n)
out['zoo_paths'] = G.codec().decode_paths()
out['bad'] = G.zstd_decompress(bad, bad_n)
out['bad_paths'] = G.codec().decode_paths()
for k in ('i16', 'mix', 'i32', 'zoo', 'bad'):
    out[k] = [o if isinstance(o, int) else o.tobytes() for o in out[k]]
pickle.dump(out, open(sys.argv[2], 'wb'))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    frames16 = [O.compress(a, O.options(True, 2, level, 1)) for level in (1, 3, 9) for a in reads]
    frames32 = [O.compress(a, O.options(True, 4, level, 1)) for level in (1, 3) for a in wide]
    # the entropy stage alone on other content: text, periodic data (matches that overlap their output), control-byte look-alikes
    zoo_src = [np.frombuffer(b"the quick brown fox jumps over the lazy dog " * 4000, np.uint8).copy(),
               np.minimum(rng.geometric(0.3, 150000), 255).astype(np.uint8)]
    for period in (1, 3, 8, 17, 80, 257, 5000):
        unit = rng.integers(0, 256, period, dtype=np.uint8)
        zoo_src.append(np.concatenate([rng.integers(0, 256, 41, dtype=np.uint8), np.tile(unit, 40000 // period + 2), unit[: period // 2]]))
    z = np.zeros(120000, np.uint8)
    for s0 in np.nonzero(rng.random(120000) < 0.004)[0]:
        z[int(s0) : int(s0) + 5] = rng.integers(1, 86, len(z[int(s0) : int(s0) + 5]), dtype=np.uint8)
    zoo_src.append(z)
    zoo = [O.zstd_compress(c, level) for c in zoo_src for level in (1, 3, 9)]
    zoo_n = [len(c) for c in zoo_src for _ in (1, 3, 9)]
    # damage, mostly in the sequences sections (the last tenth of a block's bytes) of one- and two-block frames
    bad, bad_n = [], []
    for a in (reads[0], reads[1]):
        s = O.svb_compress(a, 2, True, 0)
        for level in (1, 3):
            frame = O.zstd_compress(s, level)
            bad += [frame[: len(frame) - 2].copy(), frame[: len(frame) // 2].copy()]
            bad_n += [len(s)] * 2
            for k in range(60):
                f = frame.copy()
                for _ in range(int(rng.integers(1, 3))):
                    lo = int(len(f) * (0.9 if k % 3 else (0.40 if k % 2 else 0.0)))
                    hi = len(f) if k % 3 else int(len(f) * (0.55 if k % 2 else 1.0))
                    f[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
                bad.append(f)
                bad_n.append(len(s))
    with tempfile.TemporaryDirectory() as td:
        pickle.dump((reads, wide, frames16, frames32, zoo, zoo_n, bad, bad_n), open(os.path.join(td, "in.pkl"), "wb"))
        outs = {}
        # walked with the tables in LDS (what a batch of this size gets), walked with the tables in memory (what a batch of more
        # than 9216 frames gets), not walked
        # ... and walked without the literals of the first block decoded beside the walk (round 6: ref_pieces_kernel; on by default)
        for walk, tables, lits in (("2", "lds", "1"), ("2", "mem", "1"), ("0", "lds", "1"), ("2", "mem", "0")):
            env = dict(os.environ, VBZ_HIP_REF_CHAINS=walk, VBZ_HIP_REF_TABLES=tables, VBZ_HIP_SEGMENTED="0", VBZ_HIP_ROUTING="0", VBZ_HIP_REF_LITERALS=lits)
            subprocess.run([sys.executable, "-c", code, os.path.join(td, "in.pkl"), os.path.join(td, "out.pkl")], check=True, env=env)
            outs[walk + tables + lits] = pickle.load(open(os.path.join(td, "out.pkl"), "rb"))
    on, mem, off, nolits = outs["2lds1"], outs["2mem1"], outs["0lds1"], outs["2mem0"]
    for k in ("i16", "mix", "i32", "zoo", "bad"):
        assert len(on[k]) == len(off[k]) == len(mem[k]) == len(nolits[k])
        for i, (a, b, c, d) in enumerate(zip(on[k], off[k], mem[k], nolits[k])):
            assert a == b and a == c and a == d, (k, i)
        assert off[k + "_paths"][2] == 0 and mem[k + "_paths"] == on[k + "_paths"] == nolits[k + "_paths"]
    # the literals were decoded ahead for the reads that have four streams of a kilobyte and more in their first block (not: the short
    # reads, zeros, noise); never without the walk or when switched off
    assert on["i16_ahead"] == mem["i16_ahead"] and on["i16_ahead"] >= 3 * 5 and off["i16_ahead"] == 0 and nolits["i16_ahead"] == 0, \
        (on["i16_ahead"], mem["i16_ahead"], off["i16_ahead"], nolits["i16_ahead"])
    for i, f in enumerate(frames16):
        assert on["i16"][i] == reads[i % len(reads)].tobytes(), i
    for i in range(2 * len(reads)):
        assert on["mix"][i] == reads[i // 2].tobytes(), i
    nmix, bmix, wmix = on["mix_paths"]
    assert nmix == 2 * len(reads) and bmix >= 5 and wmix >= 13, on["mix_paths"]   # (own frames batched, reference frames walked)
    for i, f in enumerate(frames32):
        assert on["i32"][i] == wide[i % len(wide)].tobytes(), i
    for i, (c, lv) in enumerate((c, lv) for c in zoo_src for lv in (1, 3, 9)):
        assert on["zoo"][i] == c.tobytes(), i
    refused = 0
    for f, n, g in zip(bad, bad_n, on["bad"]):
        mine = O.zstd_restate_decompress(f, n)
        if mine is None:
            refused += 1
            assert isinstance(g, int) and g == 0xFFFFFFFF
        else:
            assert g == mine.tobytes()
    assert refused > 40
    # which frames were walked: every level-1 frame of a read with sequences and at most four blocks of them
    n16, _, w16 = on["i16_paths"]
    assert n16 == len(frames16) and w16 >= 3 * 13, on["i16_paths"]      # (not: the empty read, zeros, noise, 1 sample, > 4 blocks)
    assert on["i32_paths"][2] >= len(frames32) - 2, on["i32_paths"]
    assert on["zoo_paths"][2] >= on["zoo_paths"][0] // 2, on["zoo_paths"]
    assert 0 < on["bad_paths"][2] < len(bad), on["bad_paths"]


def test_walked_chains_in_calls_large_enough_to_walk_by_default():
    """Calls of 2 560 reads and more walk the chains of reference-written frames without being told to: tables in LDS up to 9 216
    reads, in memory beyond, the walk on the context's second stream beside the batched decoder's launches for this library's own
    frames.  Short reads keep this a test: 3 000 reference frames; 5 000 + 5 000 frames of both writers interleaved (> 9 216)."""
    import gpu_util as G

    if O.lib().vbo_zstd_version() is None:
        pytest.skip("no libzstd on this box")
    rng = np.random.default_rng(99)
    opts, oo = G.codec().options(True, 2, 1, 1), O.options(True, 2, 1, 1)
    reads = [O.synth_signal(5, 5000 + i, int(rng.integers(1500, 4000))) for i in range(5000)]
    ref = [O.compress(a, oo) for a in reads]
    # (a) reference frames only, tables in LDS
    back = G.decompress(ref[:3000], [a.nbytes for a in reads[:3000]], opts)
    n, batched, walked = G.codec().decode_paths()
    forced = n == 0   # (the whole suite is also run with every call forced onto the large-read path: no report from there)
    assert forced or (n == 3000 and batched == 0 and walked >= 2900), (n, batched, walked)
    for a, b in zip(reads, back):
        assert not isinstance(b, int) and b.tobytes() == a.tobytes()
    # (b) both writers in one call of 10 000 frames, tables in memory
    own = G.compress(reads, opts)
    mix, caps = [], []
    for a, f, g in zip(reads, own, ref):
        mix += [f, g]
        caps += [a.nbytes, a.nbytes]
    back = G.decompress(mix, caps, opts)
    n, batched, walked = G.codec().decode_paths()
    assert forced or (n == 10000 and walked >= 4800), (n, batched, walked)
    for i, b in enumerate(back):
        assert not isinstance(b, int) and b.tobytes() == reads[i // 2].tobytes(), i



def test_reference_frames_literals_decoded_beside_the_walk():
    """Round 6: for frames the reference wrote, the four Huffman streams of the first block are cut into 64 pieces that are walked once
    (ref_pieces_kernel, beside the chain walk) and left in stripes the general decoder reads in place (zstd_decode_fast.hip / RefLits).
    A call large enough to walk by default, reads chosen for what the decoder does with stripes: first blocks too short to be cut (below
    ~ 8 000 samples), cut into 4, 8 and 16 pieces a stream (from ~ 8 000, 15 000, 30 000 samples), up to full 128 KB ones, second blocks, long runs of literals (moved by the whole wavefront across
    stripes), long zero runs (patterns taken from the literals), matches that copy their own literals, flat and noisy stretches."""
    import gpu_util as G
    from multiprocessing.pool import ThreadPool

    if O.lib().vbo_zstd_version() is None:
        pytest.skip("no libzstd on this box")
    rng = np.random.default_rng(606)
    opts, oo = G.codec().options(True, 2, 1, 1), O.options(True, 2, 1, 1)
    reads = []
    for i in range(2700):
        n = int(rng.integers(5000, 60000)) if i % 50 else int(rng.integers(120000, 300000))
        a = O.synth_signal(5, 7000 + i, n).copy()
        k = i % 7
        if k == 1:      # flat stretches: zero runs of hundreds of control bytes between the literals
            for _ in range(3):
                s0 = int(rng.integers(0, max(1, n - 6000)))
                a[s0 : s0 + int(rng.integers(1500, 6000))] = a[s0]
        elif k == 2:    # a noisy stretch: two-byte codes, no matches -- a long run of literals
            s0 = int(rng.integers(0, max(1, n - 9000)))
            a[s0 : s0 + 8000] = rng.integers(-20000, 20000, len(a[s0 : s0 + 8000])).astype(np.int16)
        elif k == 3:    # a stretch that repeats itself
            s0 = int(rng.integers(4000, max(4001, n - 9000)))
            k4 = len(a[s0 : s0 + 4000])
            a[s0 : s0 + k4] = a[s0 - 4000 : s0 - 4000 + k4]
        elif k == 4:    # steps: runs of equal control bytes that are not zero
            a[: n // 3] = (np.arange(n // 3) * 300 % 30000).astype(np.int16)
        reads.append(a)
    with ThreadPool(min(16, os.cpu_count() or 1)) as pool:
        ref = pool.map(lambda a: O.compress(a, oo), reads)
    back = G.decompress(ref, [a.nbytes for a in reads], opts)
    n, batched, walked = G.codec().decode_paths()
    ahead = G.codec().decode_literals_ahead()
    for i, (a, b) in enumerate(zip(reads, back)):
        assert not isinstance(b, int) and b.tobytes() == a.tobytes(), i
    forced = n == 0   # (the suite is also run with every call forced onto the large-read path: no report from there)
    assert forced or (n == len(reads) and walked >= 2600 and 1900 <= ahead <= 2650), (n, batched, walked, ahead)
