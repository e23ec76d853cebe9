#!/bin/bash
# Quick GPU check while iterating on a kernel: parity tests (optionally filtered), phase timing, a short bench (twice:
# boxes and clocks differ by several per cent between calls).
# usage: tools/gpu_quick.sh ["pytest -k expression"] [extra hipcc flags for the phase-timing build]
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
timeout 900 python -m pytest tests -m gpu -x -q "${K[@]}" 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python tools/phase_timing.py $2 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" | tail -24 | tee gpurun_out/phase_timing.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench_quick.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'evals/s', d['ms_per_step'], 'ms/step', d['config']['us_per_learn_step_per_chain'], 'us/learn step')"
done
