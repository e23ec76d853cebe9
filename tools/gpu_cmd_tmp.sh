#!/bin/bash
timeout 900 python3 tools/phase_timing_wc.py 2>&1 | grep -E "team:|waits|wgrad layer|backward  " | tail -26
