#!/bin/bash
LENV_TIMING_POP=8 timeout 900 python3 tools/phase_timing_t3w.py 2>&1 | grep -v " 0          0 per" | tail -30
LENV_TIMING_POP=8 timeout 900 python3 tools/phase_timing_t3w.py -DLENV_PHASE_TIMING_SUB 2>&1 | grep -v " 0          0 per" | tail -40
