#!/usr/bin/env python3
"""Randomised soak of the whole path on a GPU box (not part of the test suite: run it for as long as you like).

    python tools/soak.py [--seconds 120] [--seed 1]

Every round draws a batch of reads of random lengths and shapes (nanopore-like signal, noise, constants, ramps,
extreme values, sparse spikes, repeated templates; every fifth round few large reads, which take the large-read path, every
fifth many short reads with a few long ones among them, which per-read routing splits into two launch groups)
and a random option set (levels 0, 1, 3, 4), then checks, read by read:
  * GPU compress -> oracle (reference path + libzstd) decompress == input
  * oracle compress -> GPU decompress == input
  * GPU compress -> GPU decompress == input
  * with zstd off, GPU bytes == oracle bytes
Any mismatch prints the reproducer (seed, round, read) and exits 1."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gpu_util as G  # noqa: E402
import oracle_lib as O  # noqa: E402


def make_read(rng, dt, kind, n):
    info = np.iinfo(dt)
    if kind == 0:  # nanopore-like: slowly moving level + noise
        a = O.synth_signal(int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20)), n).astype(np.int64)
        a = np.clip(a, info.min, info.max)
    elif kind == 1:  # uniform noise over the full range
        a = rng.integers(info.min, info.max, n, endpoint=True)
    elif kind == 2:  # constant
        a = np.full(n, int(rng.integers(info.min, info.max, endpoint=True)))
    elif kind == 3:  # ramp with wrap-around
        a = (np.arange(n) * int(rng.integers(1, 1000)) + int(rng.integers(0, 1000))) % (int(info.max) - int(info.min) + 1) + int(info.min)
    elif kind == 4:  # alternating extremes
        a = np.where(np.arange(n) & 1, info.max, info.min)
    elif kind == 6:  # a template repeated, with a few changed values (what the level >= 4 matcher is for)
        per = int(rng.integers(50, 9000))
        t = make_read(rng, dt, int(rng.integers(0, 2)), per).astype(np.int64)
        a = np.tile(t, n // per + 1)[:n].copy()
        if n and rng.random() < 0.5:
            k = max(1, n // 3001)
            a[rng.integers(0, n, k)] = rng.integers(info.min, info.max, k, endpoint=True)
    else:  # sparse spikes on a flat line
        a = np.full(n, int(rng.integers(-100, 100)) if info.min < 0 else 7)
        k = max(1, n // 97)
        if n:
            a[rng.integers(0, n, k)] = rng.integers(info.min, info.max, k, endpoint=True)
    return a.astype(dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--start-round", type=int, default=0, help="draw the earlier rounds' data (same random stream) but do not run them")
    ap.add_argument("--verbose", action="store_true", help="print every round's parameters before it runs (to find a crashing one)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    rounds = reads = 0
    while time.time() - t0 < args.seconds:
        size = int(rng.choice([1, 2, 2, 2, 4]))
        dt = {1: np.int8, 2: np.int16, 4: np.int32}[size] if rng.random() < 0.7 else {1: np.uint8, 2: np.uint16, 4: np.uint32}[size]
        zz = bool(rng.integers(0, 2))
        level = int(rng.choice([0, 1, 1, 3, 4, 4]))
        ver = int(rng.integers(0, 2))
        sized = bool(rng.integers(0, 2))
        shape = rng.random()
        if shape < 0.2:  # few, large reads: the large-read path (segments and spans)
            lens = [int(x) for x in rng.integers(300000, 1500000, int(rng.integers(1, 4)))]
            lens += [int(x) for x in rng.choice([0, 5, 4097, 700001, 1048576], 2)]
        elif shape < 0.4:  # many short reads and a few long ones among them: per-read routing (two launch groups)
            lens = [int(x) for x in rng.integers(0, 60000, 40)]
            for _ in range(int(rng.integers(1, 4))):
                lens.insert(int(rng.integers(0, len(lens) + 1)), int(rng.integers(600000 // size, 2500000 // size)))
        else:
            lens = [int(x) for x in rng.choice([0, 1, 2, 3, 5, 63, 64, 65, 255, 257, 1000, 4095, 4097, 20000, 100003, 300001], 24)]
            lens += [int(x) for x in rng.integers(0, 150000, 8)]
        bufs = [make_read(rng, dt, int(rng.integers(0, 7)), n) for n in lens]
        if rounds < args.start_round:
            rounds += 1
            continue
        if args.verbose:
            print("round %d: dtype %s zigzag %d level %d version %d sized %d lens %s" % (rounds, np.dtype(dt).name, zz, level, ver, sized, lens), flush=True)
        go = G.codec().options(zz, size, level, ver)
        oo = O.options(zz, size, level, ver)
        gc = G.compress(bufs, go, sized=sized)
        oc = [O.compress(b, oo, sized=sized) for b in bufs]
        for i, (b, g, o) in enumerate(zip(bufs, gc, oc)):
            where = "seed %d round %d read %d (dtype %s, n %d, zigzag %d, level %d, version %d, sized %d)" % (
                args.seed, rounds, i, np.dtype(dt).name, len(b), zz, level, ver, sized)
            if isinstance(g, int) or isinstance(o, int):
                if g != o:
                    print("MISMATCH (error codes)", where, g if isinstance(g, int) else "data", o if isinstance(o, int) else "data")
                    return 1
                continue
            d = O.decompress(g, b.nbytes, oo, sized=sized)
            if isinstance(d, int) or d.tobytes() != b.tobytes():
                print("MISMATCH (gpu -> oracle)", where)
                return 1
            if level == 0 and g.tobytes() != o.tobytes():
                print("MISMATCH (bytes, zstd off)", where)
                return 1
        ok_idx = [i for i, (g, o) in enumerate(zip(gc, oc)) if not isinstance(g, int) and not isinstance(o, int)]
        back1 = G.decompress([oc[i] for i in ok_idx], [bufs[i].nbytes for i in ok_idx], go, sized=sized)
        back2 = G.decompress([gc[i] for i in ok_idx], [bufs[i].nbytes for i in ok_idx], go, sized=sized)
        for i, d1, d2 in zip(ok_idx, back1, back2):
            for tag, d in (("oracle -> gpu", d1), ("gpu -> gpu", d2)):
                if isinstance(d, int) or d.tobytes() != bufs[i].tobytes():
                    print("MISMATCH (%s) seed %d round %d read %d (dtype %s, n %d, zigzag %d, level %d, version %d, sized %d)" % (
                        tag, args.seed, rounds, i, np.dtype(dt).name, len(bufs[i]), zz, level, ver, sized), d if isinstance(d, int) else "")
                    return 1
        rounds += 1
        reads += len(bufs)
    print("soak ok: %d rounds, %d reads, %.0f s, seed %d" % (rounds, reads, time.time() - t0, args.seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
