#!/bin/bash
cp learning_environments_amd/liblenv_hip.so /tmp/orig.so; cp gpurun_out_lib_new.so learning_environments_amd/liblenv_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wavechain_dueling or dueling_bench_launch or dueling_team or wavechain_plain" 2>&1 | tail -3
cp /tmp/orig.so learning_environments_amd/liblenv_hip.so
bash tools/ab_config.sh 2 gpurun_out_lib_base.so gpurun_out_lib_new.so 3
