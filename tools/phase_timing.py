#!/usr/bin/env python3
"""Per-phase shader-clock cycles of the two entropy kernels for ONE read on an otherwise idle GPU (a lone wavefront):

    VBZ_HIP_PHASE_TIMING=1 VBZ_HIP_SEGMENTED=0 python tools/phase_timing.py [--reference] [samples ...]

The timed kernel instantiations live in the experiments build of the library (lib/libvbz_hip_x.so, -DVBZ_EXPERIMENTS), which
this tool selects itself (VBZ_HIP_LIB).

--reference: the frame is written by the oracle (the reference path + libzstd) instead of the device encoder.

The library prints the phase lines on stderr (see dbg_end in vbz_api.hip).  The kernels are latency-bound per wavefront
(a lone wavefront issues about one instruction every ten cycles), so these numbers track what a change does to a frame's
critical path; what it does to throughput under load is bench.py's business."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VBZ_HIP_LIB", os.path.join(ROOT, "vbz_compression_amd", "lib", "libvbz_hip_x.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402  (test infrastructure: only generates the signal and checks the round trip here)
import gpu_util as G  # noqa: E402
from vbz_compression_amd import _lib  # noqa: E402

opts = _lib.CompressionOptions(True, 2, 1, 1)
args = sys.argv[1:]
a_seen = []   # (a different read of the generator each time)
reference = "--reference" in args
args = [x for x in args if x != "--reference"]
for n in [int(x) for x in args] or [100000, 100000]:
    a = O.synth_signal(5, 1 + len(a_seen), n)
    a_seen.append(n)
    f = [O.compress(a, O.options(True, 2, 1, 1))] if reference else G.compress([a], opts)
    b = G.decompress(f, [a.nbytes], opts)
    assert b[0].tobytes() == a.tobytes()
