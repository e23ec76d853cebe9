#!/bin/bash
# A/B of two builds of liblenv_hip.so on ONE box (boxes differ by several per cent): tools/ab_bench.sh A.so B.so [rounds]
# alternates the two libraries under the product path and prints the headline bench value of each run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp learning_environments_amd/liblenv_hip.so /tmp/liblenv_hip_orig.so
trap 'cp /tmp/liblenv_hip_orig.so learning_environments_amd/liblenv_hip.so' EXIT      # also when the run is interrupted
for i in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    cp $v learning_environments_amd/liblenv_hip.so
    python bench.py --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), d['ms_per_step'])"
  done
done
