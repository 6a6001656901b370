"""Audit tool for tests/test_gpu_fuzz_corpus.py: list every call of the fuzz-corpus replay where the device says VBZ_ZSTD_ERROR
and the oracle (the reference path with libzstd 1.4.8) says something else, grouped by (file, option set, sized, oracle
verdict) with the guessed destination sizes as inclusive ranges.

    python tools/fuzz_divergences.py [out.json]          (needs the GPU and the oracle; default gpurun_out/fuzz_divergences.json)

Round 3 used it to replace the test's blanket "device says ZSTD_ERROR -> allowed" rule: all 11 856 such calls turned out to
be frames in zstd's legacy v0.7 format (magic 0xFD2FB527, 32 corpus files) plus one frame with a Dictionary_ID field of 0
(file 210, which the decoder now accepts like libzstd does)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
import test_gpu_fuzz_corpus as T  # noqa: E402
from vbz_compression_amd import _lib  # noqa: E402


def ranges(values):
    out = []
    for v in sorted(values):
        if out and out[-1][1] + 1 == v:
            out[-1][1] = v
        else:
            out.append([v, v])
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fuzz_divergences.json")
    entries = []
    calls = 0
    for (zz, isz, lvl, ver) in T.OPTION_SETS:
        if lvl == 0:
            continue
        oo = O.options(zz, isz, lvl, ver)
        go = _lib.CompressionOptions(zz, isz, lvl, ver)
        sweeps = [O.fuzz_sweep(f, oo) for f in T.FILES]
        guesses = [list(range(G + 1)) for G, _ in sweeps]
        owner = np.repeat(np.arange(len(T.FILES)), [len(g) for g in guesses])
        flat_guess = np.concatenate([np.array(g) for g in guesses])
        for sized in (False, True):
            want = np.concatenate([r[:, 1 if sized else 0] for _, r in sweeps]).astype(np.int64)
            got, _, _ = T._decompress_batch(T.FILES, guesses, go, sized)
            calls += len(want)
            sel = np.nonzero((got == T.E_ZSTD) & (want != T.E_ZSTD))[0]
            groups = {}
            for k in sel:
                groups.setdefault((int(owner[k]), int(want[k]) if want[k] >= T.E_OOM else -1), []).append(int(flat_guess[k]))
            for (fi, w), gs in sorted(groups.items()):
                entries.append({"file": fi, "name": T.INDEX[fi].get("name", ""), "zigzag": bool(zz), "integer_size": isz, "level": lvl,
                                "version": ver, "sized": bool(sized), "oracle": "success" if w < 0 else O.ERRORS.get(w, hex(w)),
                                "guesses": ranges(gs)})
    n = sum(b - a + 1 for e in entries for a, b in e["guesses"])
    doc = {"about": "calls of the fuzz-corpus replay where the device reports VBZ_ZSTD_ERROR and the reference path (libzstd 1.4.8) "
                    "reports something else",
           "calls_replayed": calls, "calls_listed": n, "entries": entries}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
        f.write("\n")
    by = {}
    for e in entries:
        by[e["oracle"]] = by.get(e["oracle"], 0) + sum(b - a + 1 for a, b in e["guesses"])
    print("listed %d of %d calls in %d entries: %s" % (n, calls, len(entries), by))


if __name__ == "__main__":
    main()
