python tools/wc_debug_td3.py 3 20 24 2>&1 | tail -4
python tools/wc_debug_td3.py 4 70 6 2>&1 | tail -4
LENV_TIMING_POP=8 LENV_TIMING_LIB=gpurun_out_lib_t3w_timing_sub.so timeout 300 python tools/phase_timing_t3w.py 2>&1 | grep "test step"
timeout 300 python bench.py --only-config 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print('cfg5', d['ms_per_step'], 'ms', d['us_per_learn_step_per_chain'], 'us/learn')"
