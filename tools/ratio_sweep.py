#!/usr/bin/env python3
"""Compression ratio of this library against the reference path (svb + the pinned libzstd at the level each caller asks for) on THE
REFERENCE'S OWN inputs -- the reference hands `zstd_compression_level` to libzstd (vbz/vbz.cpp:194-207); here every level above 0 writes
level-1-shaped frames (include/vbz.h), so the question is what that costs where the reference's tests and benchmarks go:

  perf/sequence, perf/signal   vbz/perf/test_data_generator.h:12-23 (iota, 1 MB) and :28-74 (reads of 30 000 - 200 000 values that cycle
                               the 15 643-sample read of vbz/test/test_data.h), int8 / int16 / int32, zig-zag, level 1 (vbz_perf.cpp:113-119)
  plugin/linear, plugin/random vbz_plugin/test/vbz_hdf_plugin_test.cpp:15-136: iota of 100 values in chunks of 12 at LEVEL 5; uniform random
                               values over the whole type in chunks of count / 8, level 1; six integer types, zig-zag
  benchmark/randint            python/benchmark/benchmark.py:86-89: numpy.random.randint(-50, 50) as i1 / i2 / i4, one chunk of 1 - 19 MB,
                               zig-zag, level 1 (create_vbz_zstd)
  pyvbz/basic, pyvbz/rand      python/pyvbz/tests/unit: [1 .. 10] and 200 000 random values, six types, default options and version 1

    python tools/ratio_sweep.py [--quick] [--md profiles/r06_ratio_sweep.md]

Every case is also decoded: this library's frames by the reference path (the oracle), the oracle's frames by this library.
`--quick` (what tests/test_gpu_ratio.py runs) takes fewer and smaller buffers of each kind."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def template():
    """the read of vbz/test/test_data.h: the 15 643-sample read of the reference's test file (tests/golden, decoded by the oracle)"""
    import oracle_lib as O

    G = os.path.join(ROOT, "tests", "golden")
    idx = json.load(open(os.path.join(G, "fast5_chunks.json")))
    blob = np.fromfile(os.path.join(G, "fast5_chunks.bin"), np.uint8)
    e = [x for x in idx if x["samples"] == 15643][0]
    a = O.decompress(blob[e["chunk_offset"] : e["chunk_offset"] + e["chunk_size"]], 2 * e["samples"], O.options(True, 2, 1, 0), sized=True)
    return np.frombuffer(a.tobytes(), np.int16)


def cases(quick):
    rng = np.random.default_rng(5)
    t = template()
    out = []   # (name, [arrays], (zigzag, size, level, version))
    for dt in (np.int8, np.int16, np.int32):
        n = 1000 * 1000 // np.dtype(dt).itemsize
        out.append(("perf/sequence %s" % np.dtype(dt).name, [np.arange(n, dtype=np.int64).astype(dt)], (True, np.dtype(dt).itemsize, 1, 0)))
        lens = rng.integers(30000, 200001, 12 if quick else 24)   # (the generator's own spread of lengths: what it reports is the aggregate)
        out.append(("perf/signal %s" % np.dtype(dt).name, [np.resize(t, int(k)).astype(dt) for k in lens], (True, np.dtype(dt).itemsize, 1, 0)))
    for dt in (np.int8, np.int16, np.int32, np.uint8, np.uint16, np.uint32):
        isz = np.dtype(dt).itemsize
        lin = np.arange(100, dtype=np.int64).astype(dt)
        out.append(("plugin/linear %s, level 5" % np.dtype(dt).name, [lin[i : i + 12] for i in range(0, 96, 12)], (True, isz, 5, 0)))
        info = np.iinfo(dt)
        count = (1000 * 1000 if quick else 10 * 1000 * 1000) // 8
        out.append(("plugin/random %s" % np.dtype(dt).name, [rng.integers(info.min, info.max, count, dtype=dt, endpoint=True) for _ in range(1 if quick else 2)],
                    (True, isz, 1, 0)))
    for dt in (np.int8, np.int16, np.int32):
        isz = np.dtype(dt).itemsize
        for mb in ((1,) if quick else (1, 5, 19)):
            out.append(("benchmark/randint[-50,50) %s, %d MB" % (np.dtype(dt).name, mb), [rng.integers(-50, 50, mb * 1000000 // isz).astype(dt)], (True, isz, 1, 0)))
    for dt, rmin, rmax in ((np.int8, -2**7, 2**7 - 1), (np.uint8, 0, 2**7 - 1), (np.int16, -2**15, 2**15 - 1), (np.uint16, 0, 2**15 - 1),
                           (np.int32, -2**31, 2**31 - 1), (np.uint32, 0, 2**31 - 1)):
        isz = np.dtype(dt).itemsize
        signed = np.issubdtype(dt, np.signedinteger)
        for ver in (0, 1):
            if ver == 1 and isz == 4:
                continue   # (the unit tests run version 1 for the 8- and 16-bit types)
            out.append(("pyvbz/basic %s v%d" % (np.dtype(dt).name, ver), [np.arange(1, 11).astype(dt)], (signed, isz, 1, ver)))
            out.append(("pyvbz/rand %s v%d" % (np.dtype(dt).name, ver), [rng.integers(rmin, rmax, 200000).astype(dt)], (signed, isz, 1, ver)))
    return out


def run(quick=False):
    import oracle_lib as O
    import gpu_util as G
    from vbz_compression_amd import _lib

    rows = []
    for name, arrays, (zz, size, level, ver) in cases(quick):
        opts = _lib.CompressionOptions(zz, size, level, ver)
        oo = O.options(zz, size, level, ver)
        mine = G.compress(arrays, opts, sized=True)
        ref = [O.compress(a, oo, sized=True) for a in arrays]
        assert not any(isinstance(f, int) for f in mine), (name, mine)
        for a, f, r in zip(arrays, mine, ref):   # both ways
            assert O.decompress(f, a.nbytes, oo, sized=True).tobytes() == a.tobytes(), name
        back = G.decompress(ref, [a.nbytes for a in arrays], opts, sized=True)
        for a, b in zip(arrays, back):
            assert not isinstance(b, int) and b.tobytes() == a.tobytes(), name
        raw = sum(a.nbytes for a in arrays)
        m, r = sum(len(f) for f in mine), sum(len(f) for f in ref)
        rows.append({"case": name, "buffers": len(arrays), "raw": raw, "this": m, "reference": r, "ratio_this": raw / m, "ratio_reference": raw / r,
                     "relative": r / m, "slack_bytes": m - r})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--md", default="")
    args = ap.parse_args()
    import oracle_lib as O

    rows = run(args.quick)
    lines = ["# Compression ratio on the reference's own inputs (tools/ratio_sweep.py%s)" % (" --quick" if args.quick else ""), "",
             "Reference = the oracle: svb restated + libzstd %s at the level the caller asks for.  `relative` = reference bytes / this library's bytes" % (O.lib().vbo_zstd_version() or b"?").decode(),
             "(1.0 = the same size, above 1 = smaller than the reference).  Sized format (4-byte header included), decoder hints (skippable trailers) included.", "",
             "| case | buffers | raw bytes | this library | reference | ratio here | ratio reference | relative |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | %d | %d | %d | %d | %.3f | %.3f | %.3f |" % (r["case"], r["buffers"], r["raw"], r["this"], r["reference"], r["ratio_this"], r["ratio_reference"], r["relative"]))
    text = "\n".join(lines) + "\n"
    print(text)
    if args.md:
        with open(args.md, "w") as f:
            f.write(text)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
