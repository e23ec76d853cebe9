#!/usr/bin/env python3
"""Diagnostic (container, no GPU): scan a gfx950 assembly listing (hipcc -save-temps: *-hip-amdgcn-amd-amdhsa-gfx950.s) for the pattern that
broke a first build of dueling_wavechain.hip in round 4 -- a VGPR-to-VGPR copy (a live-range split copy of the register allocator) that
sits at the top of a basic block IN FRONT of the block's `s_or_b64 exec, exec, ...` (the end of a divergent region): the copy then runs
with the region's partial exec mask, and a value that is live in all lanes (e.g. a kernel-wide zero register parked in a callee-saved VGPR
around a call) loses the inactive lanes.  Prints every such block head; a hit is a reason to look, not a proof of a bug (copies of values
that are only live in the active lanes are fine).
usage: tools/diag/check_exec_restore.py FILE.s [more.s]"""
import re
import sys

label = re.compile(r"^\.LBB\d+_\d+:")
vmov = re.compile(r"^\s+v_mov_b32_e32 (v\d+), (v\d+)\s*$")
vmov64 = re.compile(r"^\s+v_mov_b64_e32 (v\[\d+:\d+\]), (v\[\d+:\d+\])\s*$")
exec_or = re.compile(r"^\s+s_or_b64 exec, exec, ")
skip = re.compile(r"^\s*(;|s_nop|v_writelane_b32|v_readlane_b32|s_mov_b32|s_mov_b64|s_waitcnt|\.)")
for path in sys.argv[1:]:
    func = "?"
    lines = open(path).read().split("\n")
    hits = 0
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.rstrip().endswith(":"):
            func = l.split(":")[0]
        if not label.match(l):
            continue
        copies = []
        for j in range(i + 1, min(i + 40, len(lines))):
            t = lines[j]
            if exec_or.match(t):
                if copies:
                    hits += 1
                    print("%s:%d %s %s: %d VGPR copies in front of the exec restore: %s" % (path.split("/")[-1], i + 1, func[:60], l.split(":")[0], len(copies), ", ".join(copies[:6])))
                break
            m = vmov.match(t) or vmov64.match(t)
            if m:
                copies.append("%s<-%s" % (m.group(1), m.group(2)))
                continue
            if label.match(t):
                break                                      # the copy belongs to an edge into the NEXT block (a phi copy): normal
            if skip.match(t) or not t.strip():
                continue
            break                                          # any other instruction: not a block-prologue pattern
    print("%s: %d block head(s) flagged" % (path.split("/")[-1], hits))
