#!/bin/bash
# quick A/B on the GPU box: parity tests of the plan path, then the two bench regimes with per-kernel times
set -o pipefail
mkdir -p gpurun_out
T=${1:-"tests/test_gpu_plan.py tests/test_gpu_robots.py tests/test_gpu_planner_api.py"}
timeout -k 10 600 python3 -m pytest $T -m gpu -x -q > gpurun_out/ab_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/ab_tests.log
for B in 64 1024; do
  timeout -k 10 300 python3 bench.py --batch $B --steps 10 --no-cpu-baseline --no-variants > gpurun_out/ab_b$B.json 2> gpurun_out/ab_b$B.err || { tail -5 gpurun_out/ab_b$B.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab_b$B.json')); print('B=$B', round(d['value']), 'traj/s', round(d['ms_per_step'],3), 'ms', {k: round(v['avg_ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
done
