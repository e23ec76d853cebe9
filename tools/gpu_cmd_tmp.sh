#!/bin/bash
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py 2>/dev/null | tail -1 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_launch" 2>&1 | tail -2
