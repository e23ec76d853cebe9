for v in base vA vB new base vA vB new; do
cp gpurun_out_lib_$v.so learning_environments_amd/liblenv_hip.so
python bench.py --only-config 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())[0]; print('$v', round(d['ms_per_step'],1), 'ms')"
done
