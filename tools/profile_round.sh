#!/bin/bash
# Profile recipe of one round, run on the GPU box from the repo root (via gpurun):
#   bash tools/profile_round.sh <tag>
# Leaves under gpurun_out/<tag>/: the default bench line, the rocprofv3 kernel-trace stats of the same
# command, and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other traces).
# Afterwards, in the development container:
#   python tools/summarize_profile.py <tag> gpurun_out/<tag>/kt/*/*kernel_stats.csv \
#          gpurun_out/<tag>/fetch/*/*counter_collection.csv gpurun_out/<tag>/write/*/*counter_collection.csv
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -- python3 bench.py --no-cpu > "$out/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d "$out/fetch" -- python3 bench.py --no-cpu --steps 3 --warmup 1 > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d "$out/write" -- python3 bench.py --no-cpu --steps 3 --warmup 1 > "$out/write.log" 2>&1
ls -R "$out" | head -40
cat "$out/bench.json"
