import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, oracle_lib as O
from vbz_compression_amd import _lib, vbz
rng = np.random.default_rng(44)
cases = []
for dt, size in ((np.int16, 2), (np.int32, 4), (np.int8, 1)):
    info = np.iinfo(dt)
    for n in (0, 1, 100, 4099, 70001):
        cases.append((rng.integers(info.min // 2, info.max // 2, n).astype(dt), size))
cases.append((O.synth_signal(5, 0, 400000), 2))
for ci,(a, size) in enumerate(cases):
    for zz in (True, False):
        for ver in (0,1):
            if ver==1 and size==1: continue
            go = _lib.CompressionOptions(zz, size, 1, ver); oo = O.options(zz, size, 1, ver)
            g = vbz.compress_raw(a, go)
            d = O.decompress(g, a.nbytes, oo)
            if isinstance(d,int) or d.tobytes()!=a.tobytes():
                svb = O.svb_compress(a, size, zz, ver)
                mine = O.zstd_decompress(g, len(svb)+1000)
                K=(len(a)+3)//4
                bad = np.nonzero(mine[:len(svb)]!=svb)[0] if mine is not None and len(mine)==len(svb) else None
                print("FAIL case",ci,"n",len(a),"size",size,"zz",zz,"ver",ver,"K",K,"N",len(svb),"declen",None if mine is None else len(mine), "firstbad", None if bad is None else bad[:10], "nbad", None if bad is None else len(bad))
                if bad is not None and len(bad):
                    b=bad[0]; print(" svb", svb[b-4:b+12], "mine", mine[b-4:b+12])
print("done")
