LENV_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench2.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}, d['ranks'], d['config'].get('graphs_per_generation'), d['config'].get('graph_capture_error'), 'strong', d['strong']['value'], d['strong']['ms_per_step'])"
tail -3 gpurun_out/bench2.err
python -m pytest tests -m gpu -q -x -k "two_rank or gtn_master or graph" 2>&1 | tail -3
