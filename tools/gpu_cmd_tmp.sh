python tools/wc_debug_td3.py 3 20 24 2>&1 | tail -4
python tools/wc_debug_td3.py 3 12 45 2>&1 | tail -4
bash tools/gpu_t3w_timing.sh 2>&1 | grep -v "^fwd \|^bwd \|        0 cycles\|        0          0\|detail\|wgrad "
