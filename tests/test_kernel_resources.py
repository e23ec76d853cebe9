"""Compile-time guard on scratch memory in the wave-chain kernels (docs/notebook_r05.md sections 3-4).

The out-of-line phase routines of `dueling_wavechain.hip` / `td3_wavechain.hip` must keep LLVM's no-callee-saved-registers treatment
(TargetFrameLowering::isSafeForNoCSROpt): without `-fno-optimize-sibling-calls` TailCallElim marks their calls `tail`, the routines lose
it, and every call saves / restores up to 108 VGPRs per lane through scratch (configs[2] shard: +6 % time, 1.6x the fabric traffic).
The test compiles each file to assembly with the Makefile's own flags (hipcc cross-compiles without a GPU; ~1 min per file) and counts
`scratch_store` / `scratch_load` per function.  The kernels' own frames are bounded by what the round measured, so that a regression shows."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "learning_environments_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _makefile_flags(obj):
    """FLAGS of the Makefile + the target-specific EXTRA of `obj` (so the test compiles what the product compiles)."""
    text = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^FLAGS = (.*)$", text, re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "").split()
    for m in re.finditer(r"^(.*?): EXTRA \+= (.*)$", text, re.M):
        if obj in m.group(1).split():
            flags += m.group(2).split()
    return [f for f in flags if f not in ("-fPIC", "-Wall", "-Wno-unused-parameter")]


def _scratch_by_function(src, tmp_path):
    out = str(tmp_path / (src + ".s"))
    subprocess.check_call([HIPCC] + _makefile_flags("_build/%s.o" % src) + ["-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                                                                             os.path.join(CSRC, src + ".hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    funcs, cur = {}, None
    for line in open(out):
        m = re.match(r"^(_ZN[A-Za-z0-9_]*):", line)
        if m:
            cur = funcs.setdefault(m.group(1), {"st": 0, "ld": 0, "scratch": None})
        if cur is not None:
            cur["st"] += "scratch_store" in line
            cur["ld"] += "scratch_load" in line
            m2 = re.search(r"; ScratchSize: (\d+)", line)
            if m2:
                cur["scratch"] = int(m2.group(1))
    return funcs


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.timeout(900)
@pytest.mark.parametrize("src,kernel,routines,frame_cap", [
    ("dueling_wavechain", "dueling_wavechain_kernelILi1E", ("wct_forward", "wc_forward_big", "wct_backward_chain", "wct_wgrad_layer", "wct_wgrad_ends",
                                                            "wc_backward_big", "wc_test_steps", "wc_forward_thin_layers"), 300),
    ("td3_wavechain", "td3_wavechain_kernelILi1E", ("t3v_forward", "t3v_backward", "t3v_wgrad", "t3v_learn_step", "t3w_forward", "t3w_test_steps"), 760),
])
def test_wavechain_routines_keep_no_callee_saved_registers(tmp_path, src, kernel, routines, frame_cap):
    assert "-fno-optimize-sibling-calls" in _makefile_flags("_build/%s.o" % src)
    funcs = _scratch_by_function(src, tmp_path)
    seen = 0
    for name, f in funcs.items():
        if "ILi1E" not in name and "Li1ELi" not in name:
            continue                                   # the BASELINE shape's instantiations
        if any(("%d%s" % (len(r), r)) in name for r in routines):
            seen += 1
            # a callee-saved prologue / epilogue is dozens of symmetric stores and loads; what is allowed is a handful of real spills
            assert f["st"] <= 8 and f["ld"] <= 8, "%s: %d scratch stores / %d loads (callee-saved registers are being saved again?)" % (name, f["st"], f["ld"])
        if kernel in name:
            assert f["scratch"] is not None and f["scratch"] <= frame_cap, "%s: ScratchSize %s B/lane (round 5: dueling 268, td3 720)" % (name, f["scratch"])
            seen += 100
    assert seen >= 100 + 4, "instantiations not found in the assembly (name mangling changed?): %d" % seen
