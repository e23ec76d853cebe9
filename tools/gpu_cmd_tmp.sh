#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','vs_baseline','dtype')})
print('roofline', d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['traffic'])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for c in d['configs']: print(c['config'], round(c['ms_per_step'],1), 'ms', round(c.get('us_per_learn_step_per_chain',0),2), 'us/learn', c.get('workgroups_per_chain'), c.get('mfma_f32_frac_of_busy_cus'))
"
