#!/usr/bin/env python3
"""Diagnostic: build liblenv_hip_timing.so with -DLENV_PHASE_TIMING and print per-phase shader-clock shares of chain 0
for the bench workload.  Never used by the product path or by bench.py."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "learning_environments_amd", "csrc")
OUT = "/tmp/liblenv_hip_timing.so"
srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
if os.environ.get("LENV_TIMING_LIB"):                    # a diagnostic build made elsewhere (same flags): just load it
    OUT = os.path.abspath(os.environ["LENV_TIMING_LIB"])
else:
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                         "-fPIC", "-shared", "-DLENV_PHASE_TIMING", "-o", OUT] + [a for a in sys.argv[1:] if a.startswith("-")]
                        + [os.path.join(CSRC, s) for s in srcs])
from learning_environments_amd import _lib
_lib.LIB_PATH = OUT
import torch
import bench
master, cfgd = bench.build_master(int(os.environ.get("LENV_TIMING_POP", bench.POP)))      # e.g. 32 / 16 / 8: the team launches
extra = dict(a.split("=") for a in sys.argv[1:] if not a.startswith("-"))
for k, v in extra.items():
    setattr(master.cfg, k, int(v))
if extra:
    master.inner = master.engine.make_inner(master.cfg, master.cpw * master.n_local)
master.step(0)
torch.cuda.synchronize()
import time
t0 = time.time(); master.step(1); torch.cuda.synchronize(); dt = time.time() - t0
buf = (C.c_ulonglong * 64)()
_lib.lib().lenv_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong)]
assert _lib.lib().lenv_debug_phase_cycles(buf) == 0
names = ["(unused)", "(unused)", "phaseA+wait(B1)", "forward(B2)", "td-error(B3)", "grad-reduce(B4)", "adam(B5)", "test", "-", "loop-overhead",
         "envwave:act", "envwave:se+append"]
tot = sum(buf[i] for i in (2, 3, 4, 5, 6, 7, 9))
print("generation wall %.1f ms; stats %s" % (dt * 1e3, master.inner.stats[0].tolist()))
for i, n in enumerate(names):
    if buf[i]:
        print("%-22s %12d cycles  %5.1f%%" % (n, buf[i], 100.0 * buf[i] / tot))
print("own-work cycles before each barrier (forward | split layout: rows written (waves 10, 11) / all rows seen (8, 9) | gradient | Adam), per learn step:")
steps = max(1, int(master.inner.stats[0][2]))
for slot, w in enumerate(range(12)):
    print("  wave %2d: " % w + " | ".join("%7.0f" % (buf[12 + 4 * slot + i] / steps) for i in range(4)))
print("phase totals per learn step: " + ", ".join("%s %.0f" % (names[i], buf[i] / steps) for i in (2, 3, 4, 5, 6)))
