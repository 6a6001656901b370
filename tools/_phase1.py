import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
import gpu_util as G
from vbz_compression_amd import _lib
opts = _lib.CompressionOptions(True, 2, 1, 1)
for n in (50000, 50000, 100000, 100000):
    a = O.synth_signal(5, 1, n)
    f = G.compress([a], opts)
    b = G.decompress(f, [a.nbytes], opts)
    assert b[0].tobytes() == a.tobytes()
