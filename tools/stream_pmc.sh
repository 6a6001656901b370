#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the microbenchmark's kernels (separate PMC passes, as MI355X_MICROARCH.md prescribes); run on the GPU box:
#   bash tools/stream_pmc.sh       -> gpurun_out/r04_stream/pmc_{fetch,write}_{1720,1792}/p_counter_collection.csv
mkdir -p gpurun_out/r04_stream; R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
for sym in 1720 1792; do
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r04_stream/pmc_fetch_$sym -o p -- $R/tools/stream_phase_bench --mini --symbols $sym > $R/gpurun_out/r04_stream/pmc_fetch_$sym.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r04_stream/pmc_write_$sym -o p -- $R/tools/stream_phase_bench --mini --symbols $sym > $R/gpurun_out/r04_stream/pmc_write_$sym.log 2>&1
done
