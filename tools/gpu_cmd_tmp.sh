#!/bin/bash
bash tools/ab_config.sh 2 gpurun_out_lib_v1.so gpurun_out_lib_v2.so 3
