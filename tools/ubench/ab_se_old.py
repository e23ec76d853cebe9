import sys, runpy, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from learning_environments_amd import _lib
_lib.LIB_PATH = os.path.join(R, "tools/ubench/_old_liblenv_before.so")
runpy.run_path(os.path.join(R, "tools/ubench/ab_se.py"), run_name="__main__")
