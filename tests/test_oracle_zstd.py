"""CPU tests: the restated zstd decoder (oracle/zstd_restate.c, the serial mirror of the HIP decoder)
against the pinned dependency itself (libzstd, dlopen'd) on frames libzstd produced."""
import numpy as np
import pytest

import oracle_lib as O


def _gen(rng, kind, n):
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return np.clip(rng.normal(128, 20, n), 0, 255).astype(np.uint8)
    if kind == 2:
        return (rng.integers(0, 100, n) < 4).astype(np.uint8)
    if kind == 3:
        words = [bytes(rng.integers(97, 123, rng.integers(2, 9), dtype=np.uint8)) for _ in range(200)]
        s = b" ".join(words[i] for i in rng.integers(0, 200, n // 4 + 1))
        return np.frombuffer(s[:n].ljust(n, b"x"), np.uint8).copy()
    if kind == 4:
        return np.zeros(n, np.uint8)
    if kind == 5:
        return (np.arange(n) % 251).astype(np.uint8)
    if kind == 6:
        a = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(20):
            if n < 200:
                break
            i = int(rng.integers(0, n - 100))
            l = int(rng.integers(4, min(5000, n - i)))
            j = int(rng.integers(0, n - l))
            a[j : j + l] = a[i : i + l].copy()
        return a
    return np.minimum(rng.geometric(0.3, n), 255).astype(np.uint8)


SIZES = [0, 1, 2, 5, 17, 100, 255, 256, 257, 1000, 5000, 20000, 70000, 131072, 131073, 200000, 400000]


def test_restated_decoder_matches_libzstd():
    if O.lib().vbo_zstd_version() is None:
        pytest.skip("no libzstd.so.1 on this box")
    rng = np.random.default_rng(1)
    frames = 0
    for it in range(240):
        n = SIZES[it % len(SIZES)] if it % 3 else int(rng.integers(1, 300000))
        data = _gen(rng, it % 8, n)
        for level in (1, 3, int(rng.integers(-5, 20))):
            frame = O.zstd_compress(data, level)
            out = O.zstd_restate_decompress(frame, n)
            assert out is not None, (it, n, level)
            assert out.tobytes() == data.tobytes(), (it, n, level)
            frames += 1
    assert frames == 720


def test_restated_decoder_on_svb_streams():
    # the frames the VBZ path actually produces: svb streams of synthetic signal, levels 1 and 3
    for r in range(4):
        a = O.synth_signal(5, r, 60000 + 50000 * r)
        svb = O.svb_compress(a, 2, True, 0)
        for level in (1, 3):
            frame = O.zstd_compress(svb, level)
            out = O.zstd_restate_decompress(frame, len(svb))
            assert out is not None and out.tobytes() == svb.tobytes()
    u = O.synth_u32(5, 0, 300000)
    svb = O.svb_compress(u, 4, False, 0)
    frame = O.zstd_compress(svb, 3)
    out = O.zstd_restate_decompress(frame, len(svb))
    assert out is not None and out.tobytes() == svb.tobytes()


def test_restated_decoder_rejects_corruption_without_crashing():
    rng = np.random.default_rng(3)
    a = O.synth_signal(5, 9, 30000)
    svb = O.svb_compress(a, 2, True, 0)
    frame = O.zstd_compress(svb, 1)
    accepted = 0
    for _ in range(300):
        bad = frame.copy()
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        mine = O.zstd_restate_decompress(bad, len(svb))
        ref = O.zstd_decompress(bad, len(svb))
        # The restatement follows RFC 8878 strictly (every Huffman/FSE bitstream must be consumed
        # exactly).  libzstd 1.4.8 is lenient in a few corner cases (double-symbol Huffman decoder's
        # last-symbol clamp, the superfluous final FSE state update), so it may accept a corrupted
        # frame the restatement rejects -- never the other way round, and never with different bytes.
        if mine is not None:
            accepted += 1
            assert ref is not None and ref.tobytes() == mine.tobytes()
    assert accepted < 300
    for cut in (0, 1, 4, 5, 8, 9, 12, len(frame) // 2, len(frame) - 1):
        assert O.zstd_restate_decompress(frame[:cut], len(svb)) is None or cut == 0


def test_weights_alphabet_ends_at_eleven():
    """A Huffman tree description whose FSE-coded weights LIST a symbol beyond 11 -- one no weight ever takes -- can still be a complete
    tree, and a frame built on it can decode consistently (every stream ends on its first bit): tools/soak_corrupt.py found one in
    round 5 (seed 73: one bit of the accuracy log flipped in the frame of a read with four distinct data bytes).  libzstd >= 1.4.7 refuses
    such a description (its workspace is sized for weights 0 .. HUF_TABLELOG_MAX - 1), so the restatement -- and the device decoders,
    which mirror it -- must too: they may be stricter than libzstd, never more lenient.  The fixture holds the frame before and after."""
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "weights_alphabet_beyond_11.npz"))
    good, bad = z["original"], z["damaged"]
    assert np.count_nonzero(good != bad) == 1
    ref = O.zstd_decompress(good, 60000)
    assert ref is not None and len(ref) == 45000
    assert O.zstd_restate_decompress(good, 60000).tobytes() == ref.tobytes()
    assert O.zstd_decompress(bad, 60000) is None                     # libzstd 1.4.8: "Corrupted block detected"
    assert O.zstd_restate_decompress(bad, 60000) is None
    # ... and the hole is closed in general: bit flips in the description never make the restatement accept what libzstd refuses
    rng = np.random.default_rng(5)
    for _ in range(600):
        b = good.copy()
        for _ in range(int(rng.integers(1, 3))):
            b[int(rng.integers(23, 33))] ^= 1 << int(rng.integers(0, 8))
        mine = O.zstd_restate_decompress(b, 60000)
        if mine is not None:
            lz = O.zstd_decompress(b, 60000)
            assert lz is not None and lz.tobytes() == mine.tobytes()
