#!/bin/bash
timeout 1500 python3 tools/nes_learning_demo.py 30 2>/dev/null | tee gpurun_out/nes_learning_demo_r04.jsonl | tail -8
