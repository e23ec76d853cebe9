#!/bin/bash
# round 5, GPU batch G: A/B of DDQN team-path variants at the strong-scaling shards (tools/team_ab_ddqn.py), alternating on ONE box
# usage: tools/gpu_r05_g.sh A.so B.so [more.so ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
ORIG=/tmp/liblenv_hip_orig.so
cp learning_environments_amd/liblenv_hip.so $ORIG
trap 'cp $ORIG learning_environments_amd/liblenv_hip.so' EXIT
for round in 1 2; do
  for v in "$@"; do
    cp $v learning_environments_amd/liblenv_hip.so
    echo "== $v (round $round)"
    timeout 300 python tools/team_ab_ddqn.py 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['chains'], 'chains: G1', d['G1']['ms_per_generation'], 'ms | G', d['Gauto']['workgroups_per_chain'], d['Gauto']['ms_per_generation'], 'ms | x', d['speedup'])"
  done
done 2>&1 | tee gpurun_out/r05g_ab.log
cp ${@: -1} learning_environments_amd/liblenv_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team or ddqn_chain" 2>&1 | tail -3 | tee gpurun_out/r05g_pytest.log
