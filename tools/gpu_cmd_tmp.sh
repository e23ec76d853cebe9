#!/bin/bash
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavechain or bench_launch or team or foreign" 2>&1 | tail -3
bash tools/ab_config.sh 4 gpurun_out_lib_base.so gpurun_out_lib_new.so 2
bash tools/ab_config.sh 2 gpurun_out_lib_base.so gpurun_out_lib_new.so 3
