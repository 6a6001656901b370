"""GPU tests (-m gpu) of the long-repeat matcher.

The reference hands any level to libzstd (vbz/vbz.cpp:194-207), whose match finder is on at every level, and its own perf
generator fills reads by cycling a 15 643-sample template (vbz/perf/test_data_generator.h:61-67), on which libzstd level 1
reaches ratios of 15-30.  The encoder here looks for ONE repeat distance in the data bytes of every read, at every level
(a cheap probe in the ordinary kernel; reads that have one are coded by a second launch with the matcher), and codes every
period after the first as matches with that explicit offset: same options, same input -> about the same ratio."""
import os

import numpy as np
import pytest

import oracle_lib as O
from vbz_compression_amd import _lib

import gpu_util as G

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cycled(n, phase=0):
    t = np.fromfile(os.path.join(GOLDEN, "test_data_read.i16"), dtype="<i2")     # the reference's real read (data fixture)
    idx = (np.arange(n) + phase) % len(t)
    return t[idx].astype(np.int16)


def test_template_cycling_reads_compress_like_libzstd():
    rng = np.random.default_rng(12)
    lengths = [30000, 47011, 65536, 100000, 123457, 200000, 300000, 400000, 500000, 1000000]   # (the last four: the large-read path)
    reads = [_cycled(n, int(rng.integers(0, 15643))) for n in lengths]
    reads.append(O.synth_signal(5, 3, 100000))                       # nothing periodic: must not get worse
    reads.append(np.tile(O.synth_signal(5, 4, 5000), 20)[:99999])    # a short period (5 000 samples)
    oo1, oo4 = O.options(True, 2, 1, 1), O.options(True, 2, 4, 1)
    g1 = G.compress(reads, _lib.CompressionOptions(True, 2, 1, 1))
    g4 = G.compress(reads, _lib.CompressionOptions(True, 2, 4, 1))
    for level, frames, oo in ((1, g1, oo1), (4, g4, oo4)):
        back = G.decompress(frames, [a.nbytes for a in reads], _lib.CompressionOptions(True, 2, level, 1))
        for i, (a, f, b) in enumerate(zip(reads, frames, back)):
            assert not isinstance(f, int) and not isinstance(b, int)
            assert b.tobytes() == a.tobytes()                                           # device decodes its own frame
            assert O.decompress(f, a.nbytes, oo).tobytes() == a.tobytes()                # ... and so does the reference's decoder
            ref = O.compress(a, oo)                                                      # libzstd at the SAME level
            r, rr = a.nbytes / len(f), a.nbytes / len(ref)
            if i < len(lengths) and len(a) > 524288:
                # more control bytes than one block holds: they repeat at a distance of their own and are coded as matches, chunk by
                # chunk (round 5; round 4 Huffman coded them, 44 x against libzstd's 148 x; round 3 coded such a read as spans, 2.4 x)
                assert r > 0.88 * rr, (level, i, r, rr)
                print("cycled read of %d samples, level %d: %.2f, libzstd %.2f" % (len(a), level, r, rr))
            elif i < len(lengths):
                # same options, same input: about what libzstd gets (T2; round 2 asserted r < 3 here at level 1; round 4 0.75 - 0.8 x:
                # the control bytes' own period was not used)
                assert r > 0.88 * rr, (level, i, r, rr)
                print("cycled read of %d samples, level %d: %.2f, libzstd %.2f" % (len(a), level, r, rr))
            elif i == len(lengths):
                assert abs(r / rr - 1) < (0.01 if level == 1 else 0.02), (level, r, rr)  # plain signal: the +-1 % contract (level 1)
            else:
                assert r > 0.5 * rr, (level, r, rr)
    for f1, f4 in zip(g1, g4):   # the matcher does not depend on the level
        assert len(f1) == len(f4)


def test_matcher_on_every_dtype_and_shape():
    """The matcher must never cost correctness: every integer size, zig-zag or not, periodic or not, sizes around the tile
    and block boundaries, sized and unsized, levels 1 and 4 -- decoded by the device and by the reference's decoder."""
    rng = np.random.default_rng(13)
    cases = []
    for dt, size in ((np.int16, 2), (np.int32, 4), (np.int8, 1), (np.uint32, 4), (np.uint16, 2)):
        info = np.iinfo(dt)
        base = rng.integers(max(info.min, -5000), min(info.max, 5000), 7001).astype(dt)
        for n in (0, 100, 9000, 32768, 70001, 150000, 300001):
            cases.append((np.tile(base, n // len(base) + 1)[:n].copy(), size))                    # period 7001 values
        cases.append((rng.integers(info.min // 2, info.max // 2, 50000).astype(dt), size))        # nothing to find
        wavy = np.tile(base, 20)[:120000].copy()
        wavy[::977] = wavy[::977] + 1                                                              # the period, with mismatches
        cases.append((wavy.astype(dt), size))
    for zz in (True, False):
        for sized in (False, True):
            bufs = [a for a, _ in cases]
            for size, level in ((1, 4), (2, 1), (2, 4), (4, 1)):
                sel = [a for a, sz in cases if sz == size]
                go, oo = _lib.CompressionOptions(zz, size, level, 0), O.options(zz, size, level, 0)
                frames = G.compress(sel, go, sized=sized)
                back = G.decompress(frames, [a.nbytes for a in sel], go, sized=sized)
                for a, f, b in zip(sel, frames, back):
                    assert not isinstance(f, int) and not isinstance(b, int), (len(a), size, zz, sized)
                    assert b.tobytes() == a.tobytes()
                    assert O.decompress(f, a.nbytes, oo, sized=sized).tobytes() == a.tobytes()
                    if len(a) >= 32768:   # (half a megabyte and more take the large-read path: the matcher runs in front of it)
                        ref = O.compress(a, oo, sized=sized)      # libzstd at the same level
                        assert len(f) <= 1.6 * len(ref) + 64 or len(f) <= 0.45 * a.nbytes, (len(a), size, zz, len(f), len(ref))
