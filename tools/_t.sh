mkdir -p gpurun_out/r04_soak
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_soak/tests.log; tail -15 gpurun_out/r04_soak/tests.log
VBZ_HIP_SEGMENTED=0 timeout 300 python3 tools/soak.py --seconds 200 --seed 11 > gpurun_out/r04_soak/soak_onewave_fast.log 2>&1; echo "rc=$?" >> gpurun_out/r04_soak/soak_onewave_fast.log
timeout 250 python3 tools/soak.py --seconds 150 --seed 12 > gpurun_out/r04_soak/soak_default.log 2>&1; echo "rc=$?" >> gpurun_out/r04_soak/soak_default.log
VBZ_HIP_SEGMENTED=0 timeout 250 python3 tools/soak_corrupt.py --seconds 150 --seed 13 > gpurun_out/r04_soak/corrupt_onewave_fast.log 2>&1; echo "rc=$?" >> gpurun_out/r04_soak/corrupt_onewave_fast.log
tail -3 gpurun_out/r04_soak/soak*.log gpurun_out/r04_soak/corrupt*.log
