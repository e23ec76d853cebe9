python -m pytest tests -m gpu -q -x -k "ql_" 2>&1 | tail -6
timeout 300 python bench.py --only-config 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print('cfg4', d['ms_per_step'], 'ms kernel', d['kernel_ms'])"
