#!/usr/bin/env python3
"""Time the decode of frames the reference wrote (libzstd, through the oracle) on a GPU box.

    python tools/time_foreign_decode.py [--reads 2048] [--samples 400000]

Prints the number of reads that did not decode to their input and the average duration of the decode kernels."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gpu_util as G  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--samples", type=int, default=0, help="samples per read (default: the generator's ~100 k; larger reads are several blocks)")
    args = ap.parse_args()
    opts = G.codec().options(True, 2, 1, 1)
    oo = O.options(True, 2, 1, 1)
    reads = [O.synth_signal(5, i, args.samples or O.synth_read_length(5, i)) for i in range(args.reads)]
    from multiprocessing.pool import ThreadPool

    with ThreadPool(min(16, os.cpu_count() or 1)) as pool:   # (the oracle's calls release the GIL)
        frames = pool.map(lambda a: O.compress(a, oo, sized=True), reads)
    sizes = [a.nbytes for a in reads]
    got = G.decompress(frames, sizes, opts, sized=True)
    bad = sum(1 for a, g in zip(reads, got) if isinstance(g, int) or g.tobytes() != a.tobytes())
    G.decompress(frames, sizes, opts, sized=True)   # (a second untimed call: the first runs as two halves and sizes the halves' buffers only)
    c = G.codec()
    c.profile_reset()
    c.profile(True)
    for _ in range(3):
        G.decompress(frames, sizes, opts, sized=True)
    c.profile(False)
    p = c.profile_read()
    print("libzstd frames: reads %d, bad %d, ms per launch %s, (frames, batched, walked) of the last call %s" % (
        args.reads, bad, {k: round(v[1] / max(v[0], 1), 3) for k, v in p.items() if "decode" in k}, c.decode_paths()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
