mkdir -p gpurun_out/r04c
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python3 bench.py > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04c/bench.json'))
print(d['value'], d['ms_per_step'], d['kernels_ms_per_launch'], d['ratio'])
print(json.dumps(d.get('decode_reference_frames'), indent=1))
print(d['cpu_baseline']['value'], d.get('vs_single_socket_cpu'))
PY
for m in 0 3; do VBZ_HIP_FAST_RUNS_MODE=$m python3 bench.py --no-cpu --no-pcie --no-stages --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('runs mode $m', d['kernels_ms_per_launch'])"; done
./tools/pack_phase_bench > gpurun_out/r04c/pack_sweep.log 2>&1; head -12 gpurun_out/r04c/pack_sweep.log
