"""A bounded slice of the soak tools inside the suite the driver runs: tools/soak.py and
tools/soak_corrupt.py draw random batches / damage device-written buffers and cross-check GPU <-> oracle in all directions; the
out-of-suite runs take minutes to hours, these take about a minute in all.  Fixed seeds, three launch paths (by batch shape, the
one-wavefront / batched kernels forced, the large-read path forced)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, seconds, seed, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--seconds", str(seconds), "--seed", str(seed)], env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


# (the second slice: the one-wavefront / batched kernels forced, and every reference-written frame's sequence chain walked ahead)
@pytest.mark.parametrize("seed,env", [(3, {}), (20260902, {"VBZ_HIP_SEGMENTED": "0", "VBZ_HIP_REF_CHAINS": "2"}), (7, {"VBZ_HIP_SEGMENTED": "1"})])
def test_soak_slice(seed, env):
    out = _run("soak.py", 13, seed, env)
    assert "reads" in out


def test_soak_corrupt_slice():
    _run("soak_corrupt.py", 10, 5, {"VBZ_HIP_SEGMENTED": "0"})


def test_config5_workload_at_one_gpu():
    """BASELINE configs[4] (a fixed job sharded over the ranks through the work queue) at N = 1, reduced: strong scaling, the
    rank's range covers the whole job, every batch verified."""
    import json

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config5", "--total-reads", "20000", "--reads", "4096",
                        "--no-cpu", "--no-pcie", "--no-stages"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 0
    assert d["rank_imbalance"] == pytest.approx(1.0)
    assert d["config"]["distinct_reads"] == 20000 and d["steps"] == 5   # 20 000 reads in batches of 4 096
    assert d["config"]["workload"].startswith("configs[4]: a FIXED job of 20000 reads")
    assert 2.3 < d["ratio"] < 2.5


def test_phase_timing_builds_write_the_same_frames():
    """VBZ_HIP_PHASE_TIMING selects timed instantiations of the entropy kernels (2 / 3: the encoder's planning / packing launch
    under load) in the experiments build of the library (lib/libvbz_hip_x.so, -DVBZ_EXPERIMENTS; the shipped library has neither
    the instantiations nor the knob): measurement aids, but they must code and decode what the product kernels do."""
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gpu_util as G, oracle_lib as O
from vbz_compression_amd import _lib
reads = [O.synth_signal(5, 7000 + i, n) for i, n in enumerate([100000, 65536, 33333, 4096, 250000, 17, 0] * 2)]
opts = _lib.CompressionOptions(True, 2, 1, 1)
frames = G.compress(reads, opts)
back = G.decompress(frames, [a.nbytes for a in reads], opts)
for a, f, b in zip(reads, frames, back):
    assert not isinstance(f, int) and not isinstance(b, int) and b.tobytes() == a.tobytes()
    assert O.decompress(f, a.nbytes, O.options(True, 2, 1, 1)).tobytes() == a.tobytes()
import hashlib
print("frames", sum(len(f) for f in frames), hashlib.sha256(b"".join(f.tobytes() for f in frames)).hexdigest(), G.codec().L.vbz_gpu_version().decode())
""" % (ROOT, os.path.join(ROOT, "tests"))
    from vbz_compression_amd import _lib
    sizes = {}
    for lv in ("0", "2", "3"):
        env = dict(os.environ, VBZ_HIP_PHASE_TIMING=lv, VBZ_HIP_SEGMENTED="0")
        if lv != "0":
            env["VBZ_HIP_LIB"] = _lib.EXPERIMENTS_LIB_PATH
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("frames")][-1].split()
        assert ("+experiments" in line) == (lv != "0"), line
        sizes[lv] = line[1:3]   # total bytes and their sha256: byte for byte the product's frames
    assert sizes["0"] == sizes["2"] == sizes["3"]
