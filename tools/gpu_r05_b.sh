#!/bin/bash
# round 5, GPU batch B: the whole GPU suite + the driver's bench command on the rebuilt library
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r05b_pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r05b_bench.err | tail -1 > gpurun_out/r05b_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05b_bench.json"))
print("bench", d["value"], d["ms_per_step"], d["ranks"])
for c in d.get("configs", []):
    print(c["config"], c["ms_per_step"], c.get("us_per_learn_step_per_chain"), c.get("projected_8gpu_value"), c.get("workgroups_per_chain"))
PY
