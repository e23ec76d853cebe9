#!/bin/bash
# A/B of two builds of liblenv_hip.so on ONE box for one of the other BASELINE configurations: tools/ab_config.sh N A.so B.so [rounds]
# (N = 2, 3, 4 as bench.py --only-config) alternates the two libraries under the product path and prints ms per generation.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp learning_environments_amd/liblenv_hip.so /tmp/liblenv_hip_orig.so
trap 'cp /tmp/liblenv_hip_orig.so learning_environments_amd/liblenv_hip.so' EXIT      # also when the run is interrupted
for i in $(seq 1 ${4:-3}); do
  for v in $2 $3; do
    cp $v learning_environments_amd/liblenv_hip.so
    python bench.py --only-config $1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print('$v', round(d['ms_per_step'],1), 'ms', round(d['us_per_learn_step_per_chain'],2), 'us/learn')"
  done
done
