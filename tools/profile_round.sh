#!/bin/bash
# Profile recipe of one round, run on the GPU box from the repo root (via gpurun):
#   bash tools/profile_round.sh <tag>
# Leaves under gpurun_out/<tag>/: the default bench line, the rocprofv3 kernel-trace stats of the same
# command, and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other traces).
# Afterwards, in the development container:
#   python tools/summarize_profile.py <tag> gpurun_out/<tag>/kt/*/*kernel_stats.csv \
#          gpurun_out/<tag>/fetch/*/*counter_collection.csv gpurun_out/<tag>/write/*/*counter_collection.csv \
#          [gpurun_out/<tag>/sq1/*/*counter_collection.csv gpurun_out/<tag>/sq2/*/*counter_collection.csv]
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
# The profiler passes run with the in-call split OFF (VBZ_HIP_SPLIT_MIN=0): one launch per kernel and call over the whole batch, nothing
# beside it -- per-kernel durations and counters that mean what bench.py's per_kernel table says they mean.  (With the split on, every
# kernel runs twice per call over half the reads, beside its twin: set PROFILE_SPLIT=1 to profile that.)
if [ "${PROFILE_SPLIT:-0}" != "1" ]; then export VBZ_HIP_SPLIT_MIN=0; fi
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -- python3 bench.py --no-cpu --no-pcie --no-stages --no-configs > "$out/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d "$out/fetch" -- python3 bench.py --no-cpu --no-pcie --no-stages --no-configs --steps 3 --warmup 1 > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d "$out/write" -- python3 bench.py --no-cpu --no-pcie --no-stages --no-configs --steps 3 --warmup 1 > "$out/write.log" 2>&1
# SQ counters (instruction mix, stalls, LDS conflicts), two more separate passes
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d "$out/sq1" -- python3 bench.py --no-cpu --no-pcie --no-stages --no-configs --steps 3 --warmup 1 > "$out/sq1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC -f csv -d "$out/sq2" -- python3 bench.py --no-cpu --no-pcie --no-stages --no-configs --steps 3 --warmup 1 > "$out/sq2.log" 2>&1
# the large-read path (BASELINE configs[3]: uint32 buffers of 10 M elements): bench lines, kernel stats, traffic, SQ counters
python3 bench.py --workload config4 --no-cpu > "$out/bench_config4.json" 2> "$out/bench_config4.err"
python3 bench.py --workload config4 --buffers 1 --no-cpu > "$out/bench_config4_one_buffer.json" 2>> "$out/bench_config4.err"
python3 bench.py --workload config1 --no-cpu > "$out/bench_config1.json" 2>> "$out/bench_config4.err"
rocprofv3 --kernel-trace --stats -f csv -d "$out/c4kt" -- python3 bench.py --workload config4 --no-cpu --steps 10 --warmup 2 > "$out/c4kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d "$out/c4fetch" -- python3 bench.py --workload config4 --no-cpu --steps 3 --warmup 1 > "$out/c4fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d "$out/c4write" -- python3 bench.py --workload config4 --no-cpu --steps 3 --warmup 1 > "$out/c4write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d "$out/c4sq1" -- python3 bench.py --workload config4 --no-cpu --steps 3 --warmup 1 > "$out/c4sq1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC -f csv -d "$out/c4sq2" -- python3 bench.py --workload config4 --no-cpu --steps 3 --warmup 1 > "$out/c4sq2.log" 2>&1
unset VBZ_HIP_SPLIT_MIN
# per-call latencies of the large-read path and the aggregate rate with two contexts in flight
python3 tools/time_one_read.py 10000 100000 400000 2>&1 | grep samples > "$out/one_read.txt"
python3 tools/time_large.py 2>&1 | grep case > "$out/time_large.txt"
python3 tools/two_in_flight.py --reads 32768 2>&1 | grep "reads per batch" > "$out/two_in_flight.txt"
ls -R "$out" | head -60
cat "$out/bench.json"
