#!/bin/bash
# Diagnostic A/B on ONE box: bench_configs 5 with the in-tree library vs one whose TD3 kernel is an older source (scratch copies
# tools/ubench/_old_*, git-ignored).
# The old sources are produced locally first, e.g.:  git show <commit>:learning_environments_amd/csrc/td3_rn_inner_loop.hip >
# tools/ubench/_old_td3_rn_inner_loop.hip  (likewise _old_lenv_device.cuh, _old_mlp_forward.hip)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
[ -f tools/ubench/_old_td3_rn_inner_loop.hip ] || { echo "no tools/ubench/_old_* sources: see the header of this script"; exit 1; }
rm -rf /tmp/oldtree && mkdir -p /tmp/oldtree/learning_environments_amd && cp -r include /tmp/oldtree/ && cp -r learning_environments_amd/csrc /tmp/oldtree/learning_environments_amd/ && ln -sfn /tmp/oldtree/learning_environments_amd/csrc /tmp/oldcsrc && rm -f /tmp/oldcsrc/*.o /tmp/oldcsrc/*.so
cp tools/ubench/_old_td3_rn_inner_loop.hip /tmp/oldcsrc/td3_rn_inner_loop.hip
cp tools/ubench/_old_lenv_device.cuh /tmp/oldcsrc/lenv_device.cuh
cp tools/ubench/_old_mlp_forward.hip /tmp/oldcsrc/mlp_forward.hip
(cd /tmp/oldtree/learning_environments_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -o /tmp/liblenv_old.so *.hip 2>&1 | grep "error:" | head)
for i in 1 2; do
  echo "== new"; python tools/bench_configs.py 5 2>/dev/null | grep -o '"us_per_learn_step_per_chain": [0-9.]*'
  echo "== old"; python - <<'PY' 2>/dev/null | grep -o '"us_per_learn_step_per_chain": [0-9.]*'
import sys, runpy
sys.path.insert(0, ".")
from learning_environments_amd import _lib
_lib.LIB_PATH = "/tmp/liblenv_old.so"
sys.argv = ["bench_configs.py", "5"]
runpy.run_path("tools/bench_configs.py", run_name="__main__")
PY
done
