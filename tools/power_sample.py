#!/usr/bin/env python3
"""Socket power and shader clock while ONE direction of the path runs back to back for a few seconds (65 536 reads per launch
group), sampled from rocm-smi in a side thread:

    python tools/power_sample.py [reads] [seconds]

Says whether a kernel runs into the board's power management (clock below the 2.4 GHz boost under sustained load) -- in which
case its time follows the work it does, not the latency of its dependent chains."""
import ctypes
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vbz_compression_amd import batch

codec = batch.GpuCodec(0)
torch.cuda.set_stream(codec.stream)
opts = codec.options(True, 2, 1, 1)
L = codec.L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
lens = codec.synth_lengths(5, 0, n)
sizes = lens.to(torch.int64) * 2
off, total = batch.layout(sizes.cpu(), 64)
raw = torch.empty(total, dtype=torch.uint8, device="cuda")
off = off.cuda()
codec.synth_signal(5, 0, raw, off, lens)
s32 = sizes.to(torch.int32)
caps = torch.tensor([L.vbz_max_compressed_size(int(s), ctypes.byref(opts)) for s in sizes.cpu().tolist()], dtype=torch.int64)
coff, ctotal = batch.layout(caps, 64)
comp = torch.empty(ctotal, dtype=torch.uint8, device="cuda")
coff = coff.cuda()
cap32 = caps.to(torch.int32).cuda()
cs = torch.zeros(n, dtype=torch.int32, device="cuda")
back = torch.empty_like(raw)
res = torch.zeros(n, dtype=torch.int32, device="cuda")
codec.compress(raw, off, s32, comp, coff, cap32, cs, opts)
codec.decompress(comp, coff, cs, back, off, s32, res, opts)
torch.cuda.synchronize()
assert torch.equal(raw, back)

samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        except Exception:
            break
        row = {}
        for ln in out.splitlines():
            if "sclk" in ln:
                row["sclk"] = ln.split("(")[-1].split("M")[0]
            elif "Power (W)" in ln:
                row["W"] = ln.split(":")[-1].strip()
            elif "Sensor memory" in ln:
                row["Tmem"] = ln.split(":")[-1].strip()
            elif "Sensor junction" in ln:
                row["Tj"] = ln.split(":")[-1].strip()
        samples.append((time.perf_counter(), row))
        time.sleep(0.3)


def run(name, fn):
    global samples
    samples = []
    torch.cuda.synchronize()
    time.sleep(3.0)   # cool down a little between the legs
    codec.profile_reset()
    codec.profile(True)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < secs:
        fn()
        k += 1
        if k % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    codec.profile(False)
    per = {kk: round(v[1] / max(v[0], 1), 3) for kk, v in codec.profile_read().items() if "zstd" in kk or "svb" in kk}
    rows = [r for (t, r) in samples if t0 + 1.0 <= t <= t1]
    print(name, "calls", k, "ms/call", round(1000 * (t1 - t0) / k, 3), per)
    print("   ", " | ".join("%s W %s MHz Tm %s" % (r.get("W"), r.get("sclk"), r.get("Tmem")) for r in rows))


th = threading.Thread(target=sampler, daemon=True)
th.start()
run("idle", lambda: time.sleep(0.05))
run("encode", lambda: codec.compress(raw, off, s32, comp, coff, cap32, cs, opts))
run("decode", lambda: codec.decompress(comp, coff, cs, back, off, s32, res, opts))
o0 = codec.options(True, 2, 0, 1)
run("svb only (level 0) encode", lambda: codec.compress(raw, off, s32, comp, coff, cap32, cs, o0))
stop = True
