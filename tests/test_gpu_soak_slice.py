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
    """VBZ_HIP_PHASE_TIMING selects timed instantiations of the entropy kernels (1: one launch per frame, 3: the encoder's packing
    launch under load) in the experiments build of the library (lib/libvbz_hip_x.so, -DVBZ_EXPERIMENTS; the shipped library has neither
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
    for lv in ("0", "1", "3"):
        env = dict(os.environ, VBZ_HIP_PHASE_TIMING=lv, VBZ_HIP_SEGMENTED="0")
        if lv != "0":
            env["VBZ_HIP_LIB"] = _lib.EXPERIMENTS_LIB_PATH
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("frames")][-1].split()
        assert ("+experiments" in line) == (lv != "0"), line
        sizes[lv] = line[1:3]   # total bytes and their sha256: byte for byte the product's frames
    assert sizes["0"] == sizes["1"] == sizes["3"]


def test_staged_and_fused_encoder_write_the_same_frames():
    """The staged encoder (svb_encode's hand-over, the planning launch, zstd_pack_kernel: the default) and the one-launch encoder
    (VBZ_HIP_STAGED_ENCODE=0: zstd_encode_kernel<.., 0> for every read) claim the same frames byte for byte (vbz_kernels.h;
    ADVICE round 4).  Held to it frame by frame on the shapes that take the side exits of the staged form: empty and tiny reads (no
    control-byte region), reads at the edges of the svb encoder's tokeniser (1 639 / 1 640 values, 32 767 / 32 768), more than 16 blocks
    in a region, sampled histograms that mislead (both ways), template-cycling reads (the matcher's launch), noise and constants --
    and destination slots so tight that the packing launch gives a read back (or the frame does not fit at all: same verdict)."""
    code = r"""
import hashlib, json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import gpu_util as G, oracle_lib as O
from vbz_compression_amd import _lib
rng = np.random.default_rng(5)
def from_data_bytes(pattern):   # int16 samples whose svb data bytes are exactly `pattern` (one byte per value, control bytes all zero)
    u = pattern.astype(np.int64)
    d = (u >> 1) ^ -(u & 1)
    return np.cumsum(d).astype(np.int16)
def misleading(n, noise_in_sample):
    K = (n + 3) // 4
    h = (16 - K %% 16) %% 16        # the scratch slot starts 16-byte aligned: data byte p is sampled iff ((p - h) >> 10) & 3 == 0
    p = np.arange(n)
    sampled = (p < h) | ((((p - h) >> 10) & 3) == 0)
    a = np.where(sampled == noise_in_sample, rng.integers(0, 256, n), 0).astype(np.uint8)
    if not noise_in_sample:          # the sample must show (nearly) every byte value or it is not used at all
        idx = np.flatnonzero(sampled)
        a[idx[rng.permutation(len(idx))[:256 * (len(idx) // 1024)]]] = np.tile(np.arange(256, dtype=np.uint8), len(idx) // 1024)
    return from_data_bytes(a)
t = O.synth_signal(5, 99, 15643)
reads = [O.synth_signal(5, 8000 + i, n) for i, n in enumerate([0, 1, 17, 400, 1638, 1639, 1640, 1641, 2048, 8192, 32767, 32768, 32769, 65536, 100000, 110000, 250000, 1500000])]
reads += [misleading(n, k) for n in (40000, 131072, 200000) for k in (False, True)]
reads += [np.resize(t, n) for n in (30000, 100000, 200000)]
reads += [rng.integers(-32768, 32767, n, dtype=np.int16) for n in (5000, 100000)]
reads += [np.full(40000, 7, np.int16), np.arange(0, 1000, dtype=np.int16), np.repeat(rng.integers(-2000, 2000, 400).astype(np.int16), 250)]
out = []
for lvl, sized in ((1, False), (1, True)):
    opts = _lib.CompressionOptions(True, 2, lvl, 1)
    frames = G.compress(reads, opts, sized=sized)
    for a, f in zip(reads, frames):
        assert not isinstance(f, int), (len(a), f)
        assert O.decompress(f, a.nbytes, O.options(True, 2, lvl, 1), sized=sized).tobytes() == a.tobytes(), len(a)
    out.append([hashlib.sha256(f.tobytes()).hexdigest() for f in frames])
    if not sized:   # the same reads into slots that are just large enough, a little too small, far too small
        for slack in (600, 64, 0, -7, -1000):
            caps = [max(len(f) + slack, 16) for f in frames]
            tight = G.run_stage(lambda c, *x: c.compress(*x, opts, sized=False), reads, caps)
            out.append([f if isinstance(f, int) else hashlib.sha256(f.tobytes()).hexdigest() for f in tight])
print("RESULT", json.dumps(out))
""" % (ROOT, os.path.join(ROOT, "tests"))
    import json

    got = {}
    for staged in ("1", "0"):
        env = dict(os.environ, VBZ_HIP_STAGED_ENCODE=staged, VBZ_HIP_SEGMENTED="0", VBZ_HIP_ROUTING="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        got[staged] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1][7:])
    for k, (a, b) in enumerate(zip(got["1"], got["0"])):
        differ = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
        assert not differ, (k, differ)
    # the tight slots did make some reads fail (same reads, same codes either way: compared above) and let others through
    # (lists: 0 ordinary slots; 1 .. 5 slack 600, 64, 0, -7, -1000; 6 the sized variant)
    assert len(got["1"]) == 7 and not any(isinstance(x, int) for x in got["1"][0]), [x for x in got["1"][0] if isinstance(x, int)]
    assert sum(isinstance(x, int) for x in got["1"][5]) >= 10, got["1"][5]   # a thousand bytes too few: no ordinary frame fits
