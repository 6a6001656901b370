#!/bin/bash
# baseline of the large-read path: timings and a kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/large
export TMPDIR=/tmp
python tools/time_one_read.py 100000 400000 > gpurun_out/large/one_read.txt 2>&1
python tools/time_large.py > gpurun_out/large/time_large.txt 2>&1
python bench.py --workload config4 --steps 20 --warmup 5 > gpurun_out/large/config4.json 2> gpurun_out/large/config4.err
python bench.py --workload config4 --buffers 1 --steps 20 --warmup 5 > gpurun_out/large/config4_one.json 2>> gpurun_out/large/config4.err
python bench.py --workload config1 --steps 20 --warmup 5 > gpurun_out/large/config1.json 2>> gpurun_out/large/config4.err
rocprofv3 --kernel-trace --stats -d gpurun_out/large/prof8 -o c4 -- python3 bench.py --workload config4 --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/large/prof1 -o c41 -- python3 bench.py --workload config4 --buffers 1 --steps 10 --warmup 3 > /dev/null 2>&1
ls -R gpurun_out/large | head -30
