#!/bin/bash
# round 5, GPU batch M: the rocprofv3 kernel-trace run of the configs[1] strong-scaling shard alone (bench.py --only-config 1 times the 24-chain
# shard only; the 2- / 4-GPU shards of the projected curve belong to the default run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_cfg1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg1 -- python3 $R/bench.py --only-config 1 > $R/gpurun_out/prof_cfg1.json 2> $R/gpurun_out/prof_cfg1.log
tail -c 600 $R/gpurun_out/prof_cfg1.json
