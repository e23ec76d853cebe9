#!/bin/bash
# Diagnostic: bench.py with liblenv_hip.so whose DDQN kernel file is rebuilt with extra compiler flags ($1), next to the in-tree build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rm -rf /tmp/flagtree && mkdir -p /tmp/flagtree/learning_environments_amd && cp -r include /tmp/flagtree/ && cp -r learning_environments_amd/csrc /tmp/flagtree/learning_environments_amd/
cd /tmp/flagtree/learning_environments_amd/csrc && rm -f _build/ddqn_se_inner_loop.o ../liblenv_hip.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fno-slp-vectorize $1 -c ddqn_se_inner_loop.hip -o _build/ddqn_se_inner_loop.o 2>&1 | grep "error" | head -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/liblenv_flags.so _build/*.o
cd $R
for i in 1 2; do
  echo "== in-tree"; python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['us_per_learn_step_per_chain'])"
  echo "== flags $1"; python - <<'PY' 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['us_per_learn_step_per_chain'])"
import sys, runpy
sys.path.insert(0, ".")
from learning_environments_amd import _lib
_lib.LIB_PATH = "/tmp/liblenv_flags.so"
sys.argv = ["bench.py", "--no-cpu-baseline"]
runpy.run_path("bench.py", run_name="__main__")
PY
done
