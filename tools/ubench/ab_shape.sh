#!/bin/bash
# Diagnostic A/B on ONE box: bench.py with the shape-specialised DDQN kernel vs the generic instantiation (LENV_NO_FIXED_SHAPE=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "inner_loop or ddqn or step_budget or master_run_ddqn or full_size" 2>&1 | tail -4
for i in 1 2; do
  for m in 0 1; do
    LENV_NO_FIXED_SHAPE=$m timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('generic' if $m else 'fixed  ', d['value'], 'evals/s', d['ms_per_step'], 'ms/step', d['config']['us_per_learn_step_per_chain'], 'us/learn step')"
  done
done
timeout 300 python tools/phase_timing.py 2>/dev/null | grep "wave\|phase totals"
