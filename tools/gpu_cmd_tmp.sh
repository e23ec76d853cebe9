LENV_TIMING_LIB=gpurun_out_lib_wc_timing.so timeout 300 python tools/phase_timing_wc.py 2>&1 | grep -v "warning" | tail -44
timeout 300 python bench.py --only-config 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print('cfg3', d['ms_per_step'], 'ms', d['us_per_learn_step_per_chain'], 'us/learn')"
