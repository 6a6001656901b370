#!/usr/bin/env python3
"""The launch chain of the large-read path from a rocprofv3 kernel trace: for the LAST call of each direction in the trace, every
kernel with its start (relative to the chain's first kernel), duration and the gap to the kernel before it.

    rocprofv3 --kernel-trace -f csv -d gpurun_out/chain -- python3 bench.py --workload config4 --buffers 1 --steps 5
    python3 tools/launch_chain.py gpurun_out/chain

A chain starts at a call's first planning launch; encode chains contain zstd_encode_kernel, decode chains zstd_decode_kernel."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
chains = []
cur = None
for s, e, name in rows:
    short = name.replace("vbzhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "vbzhip" not in name:
        continue
    # a call's chain begins with its planning launches (descriptor check, segment plan, scratch plan: whichever comes first)
    starter = any(k in name for k in ("validate_batch_kernel", "seg_plan_kernel", "plan_scratch_kernel"))
    if cur is None or (starter and any(("zstd_" in k or "svb_" in k or "hand_back" in k) for _, _, k in cur)):
        cur = []
        chains.append(cur)
    cur.append((s, e, short))
for kind in ("zstd_encode_kernel", "zstd_decode_kernel"):
    sel = [c for c in chains if any(kind in k for _, _, k in c) and not any("synth" in k for _, _, k in c)]
    if not sel:
        continue
    c = sel[-1]
    t0 = c[0][0]
    busy = sum(e - s for s, e, _ in c)
    print("%s chain: %d launches, %.1f us from first start to last end, %.1f us inside kernels" % (kind.split("_")[1], len(c), (c[-1][1] - t0) / 1e3, busy / 1e3))
    prev = None
    for s, e, k in c:
        print("   +%7.1f us  %6.1f us  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, k[:90]))
        prev = e
