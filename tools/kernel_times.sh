#!/bin/bash
# Per-kernel durations of any command on the GPU box (rocprofv3 kernel trace), library kernels only:
#   bash tools/kernel_times.sh <tag> python3 tools/time_large.py
set -u
tag=${1:?usage: kernel_times.sh <tag> <command...>}; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -- "$@" > "$out/kt.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/*/*kernel_stats.csv")
if not f:
    print("no kernel stats; log tail:"); print(open(sys.argv[1] + "/kt.log").read()[-1500:]); sys.exit(1)
for r in csv.DictReader(open(f[0])):
    if "vbzhip" in r["Name"]:
        print(r["Name"][:120].replace("vbzhip::(anonymous namespace)::", "").replace("void ", ""), r["Calls"],
              "avg_us=%.1f min=%.1f max=%.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
