#!/bin/bash
# round 5, GPU batch A: wave priorities by remaining work in the DDQN kernel's forward / gradient intervals (LENV_DDQN_PRIO / LENV_DDQN_GPRIO builds
# made by tools/build_variant.sh), each at the headline population and at the strong-scaling shards, alternating on ONE box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
ORIG=/tmp/liblenv_hip_orig.so
cp learning_environments_amd/liblenv_hip.so $ORIG
trap 'cp $ORIG learning_environments_amd/liblenv_hip.so' EXIT
for round in 1 2; do
  for v in base p1 p1b p1g g1; do
    cp gpurun_out_lib_$v.so learning_environments_amd/liblenv_hip.so
    echo "== $v (round $round)"
    timeout 300 python tools/team_ab_ddqn.py 6 2>/dev/null
  done
done 2>&1 | tee gpurun_out/r05a_ab.log
cp gpurun_out_lib_p1g.so learning_environments_amd/liblenv_hip.so
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ddqn" 2>&1 | tail -5 | tee gpurun_out/r05a_pytest.log
cp $ORIG learning_environments_amd/liblenv_hip.so
for v in tbase tp1g; do
  echo "== phase timing $v"
  LENV_TIMING_LIB=gpurun_out_lib_$v.so timeout 300 python tools/phase_timing.py 2>&1 | grep -v "warning\|amdgpu.ids" | tail -24
done 2>&1 | tee gpurun_out/r05a_phase.log
