cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
python tools/uneven_team_ab.py shipped 8
LENV_TIMING_LIB=gpurun_out_lib_uneven9.so python tools/uneven_team_ab.py uneven_9_3 8
LENV_TIMING_LIB=gpurun_out_lib_uneven10.so python tools/uneven_team_ab.py uneven_10_2 8
done 2>&1 | grep -v Warning | tee gpurun_out/r06_ddqn_uneven_team_ab.log
