#!/bin/bash
# One GPU-box round (run via gpurun from the repo root): parity tests, smoke, bench, rocprofv3 kernel trace, and the
# HBM-traffic PMC passes (separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes).  Outputs: gpurun_out/.
# usage: tools/gpu_round.sh [tag]   (tag names the files written under profiles/, default r02)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
python bench.py 2>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench.json
python tools/phase_timing.py 2>&1 | tail -14 | tee gpurun_out/phase_timing.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_run.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_configs -- python3 $R/tools/bench_configs.py 2 2full 3 4 5 3full 5full > $R/gpurun_out/bench_configs.jsonl 2> $R/gpurun_out/prof_configs.log
cd $R
python3 tools/summarize_profiles.py $TAG
