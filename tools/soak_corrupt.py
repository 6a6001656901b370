#!/usr/bin/env python3
"""Randomised corruption soak of the decoder on a GPU box (not part of the test suite).

    python tools/soak_corrupt.py [--seconds 120] [--seed 1]

Every round compresses a few reads on the device (random dtype / shape / level 1 or 4; sometimes few large reads, whose
frames carry a span index), damages the compressed buffers -- bit flips, byte overwrites, truncation, garbage appended,
a damaged trailer -- and decodes them on the device.  Required: no fault, and whenever the device returns samples the
reference path (oracle + libzstd) must return the same samples (the device may refuse what libzstd's leniency lets
through, never the other way round)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gpu_util as G  # noqa: E402
import oracle_lib as O  # noqa: E402
import soak  # noqa: E402


def damage(rng, f):
    g = f.copy()
    kind = int(rng.integers(0, 6))
    n = len(g)
    if n == 0:
        return g
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))):
            g[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        a = int(rng.integers(0, n))
        k = len(g[a : a + int(rng.integers(1, 9))])
        g[a : a + k] = rng.integers(0, 256, k, dtype=np.uint8)
    elif kind == 2:
        g = g[: int(rng.integers(0, n))].copy()
    elif kind == 3:
        g = np.concatenate([g, rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8)])
    elif kind == 4:  # the tail (where the trailers live)
        a = max(0, n - int(rng.integers(1, 300)))
        g[a] ^= 1 << int(rng.integers(0, 8))
    else:            # the head (frame and block headers)
        g[int(rng.integers(0, min(n, 24)))] ^= 1 << int(rng.integers(0, 8))
    return np.ascontiguousarray(g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--writer", choices=["device", "reference"], default="device",
                    help="reference: the frames are libzstd's (the oracle's compressor, levels 1 and 3), int16 reads long enough for the chain walk and the "
                         "literals beside it -- run with VBZ_HIP_REF_CHAINS=2, which walks in calls of any size")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    rounds = decoded = refused = stricter = 0
    while time.time() - t0 < args.seconds:
        size = int(rng.choice([1, 2, 2, 2, 4]))
        dt = {1: np.int8, 2: np.int16, 4: np.int32}[size]
        zz = bool(rng.integers(0, 2))
        level = int(rng.choice([1, 1, 4]))
        sized = bool(rng.integers(0, 2))
        if rng.random() < 0.3:
            lens = [int(x) for x in rng.integers(300000, 900000, 2)]
        else:
            lens = [int(x) for x in rng.choice([0, 5, 64, 257, 4097, 20000, 100003], 6)] + [int(x) for x in rng.integers(0, 120000, 2)]
        if args.writer == "reference":
            size, dt, sized = 2, np.int16, bool(rng.integers(0, 2))
            level = int(rng.choice([1, 1, 3]))
            lens = [int(x) for x in rng.integers(25000, 160000, 5)] + [int(x) for x in rng.integers(0, 30000, 2)] + [int(rng.integers(160000, 520000))]
        bufs = [soak.make_read(rng, dt, int(rng.choice([0, 0, 0, 5, 6, 3])) if args.writer == "reference" else int(rng.integers(0, 7)), n) for n in lens]
        if args.verbose:
            print("round %d: %s zz %d level %d sized %d lens %s" % (rounds, np.dtype(dt).name, zz, level, sized, lens), flush=True)
        go = G.codec().options(zz, size, level, 0)
        oo = O.options(zz, size, level, 0)
        frames = [O.compress(b, oo, sized=sized) for b in bufs] if args.writer == "reference" else G.compress(bufs, go, sized=sized)
        bad, want = [], []
        for b, f in zip(bufs, frames):
            if isinstance(f, int):
                continue
            for _ in range(4):
                bad.append(damage(rng, f))
                want.append(b.nbytes)
        got = G.decompress(bad, want, go, sized=sized)
        for v, nb, g in zip(bad, want, got):
            ref = O.decompress(v, nb, oo, sized=sized)
            if isinstance(g, int):
                refused += 1
                stricter += 0 if isinstance(ref, int) else 1
            else:
                decoded += 1
                if isinstance(ref, int) or ref.tobytes() != g.tobytes():
                    print("MISMATCH seed %d round %d: the device returned samples the reference path does not (%s)" % (
                        args.seed, rounds, hex(ref) if isinstance(ref, int) else "different samples"))
                    # the reproducer: the damaged buffer, the undamaged frames of the round and the options
                    out = os.path.join(ROOT, "gpurun_out", "soak_mismatch_seed%d_round%d.npz" % (args.seed, rounds))
                    os.makedirs(os.path.dirname(out), exist_ok=True)
                    np.savez(out, damaged=v, nbytes=nb, size=size, zz=zz, level=level, sized=sized, lens=np.array(lens),
                             device=g, index=np.array([i for i, x in enumerate(bad) if x is v][:1]),
                             **{"frame%d" % i: f for i, f in enumerate(frames) if not isinstance(f, int)})
                    print("reproducer written to", out)
                    return 1
        rounds += 1
    print("corruption soak ok: %d rounds, %d damaged buffers decoded like the reference, %d refused (%d of them accepted by libzstd), %.0f s, seed %d"
          % (rounds, decoded, refused, stricter, time.time() - t0, args.seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
