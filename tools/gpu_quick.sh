#!/bin/bash
# Quick GPU check while iterating on a kernel: parity tests (optionally filtered), phase timing, a short bench.
# usage: tools/gpu_quick.sh ["pytest -k expression"]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 900 python -m pytest tests -m gpu -x -q "${K[@]}" 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python tools/phase_timing.py 2>&1 | tail -14 | tee gpurun_out/phase_timing.log
timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench_quick.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'evals/s', d['ms_per_step'], 'ms/step', d['config']['us_per_learn_step_per_chain'], 'us/learn step')"
