#!/bin/bash
# Diagnostic A/B on ONE box: bench_configs 3 with the in-tree library vs one whose dueling kernel is built from a scratch source
# (tools/ubench/_old_duel_fix.hip, git-ignored) with extra flags ($1).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
[ -f tools/ubench/_old_duel_fix.hip ] || { echo "no scratch source"; exit 1; }
rm -rf /tmp/oldtree && mkdir -p /tmp/oldtree/learning_environments_amd && cp -r include /tmp/oldtree/ && cp -r learning_environments_amd/csrc /tmp/oldtree/learning_environments_amd/
rm -f /tmp/oldtree/learning_environments_amd/csrc/*.o /tmp/oldtree/learning_environments_amd/csrc/*.so
cp tools/ubench/_old_duel_fix.hip /tmp/oldtree/learning_environments_amd/csrc/dueling_se_inner_loop.hip
(cd /tmp/oldtree/learning_environments_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared $1 -o /tmp/liblenv_old.so *.hip 2>&1 | grep "error:" | head)
for i in 1 2; do
  echo "== in-tree"; python tools/bench_configs.py 3 2>/dev/null | grep -o '"us_per_learn_step_per_chain": [0-9.]*'
  echo "== scratch $1"; python - <<'PY' 2>/dev/null | grep -o '"us_per_learn_step_per_chain": [0-9.]*'
import sys, runpy
sys.path.insert(0, ".")
from learning_environments_amd import _lib
_lib.LIB_PATH = "/tmp/liblenv_old.so"
sys.argv = ["bench_configs.py", "3"]
runpy.run_path("tools/bench_configs.py", run_name="__main__")
PY
done
