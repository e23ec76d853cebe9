#!/bin/bash
# Diagnostic (run via gpurun from the repo root): SQ-side PMC counters of the DuelingDDQN kernel at config 3's shapes
# (tools/bench_configs.py 3), own rocprofv3 runs with --kernel-trace only.  Prints per-launch sums.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmcbig_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcbig_$tag -- python3 $R/tools/bench_configs.py 3 > $R/gpurun_out/pmcbig_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcbig_*/*/*counter_collection.csv')):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if 'dueling_se_inner' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in acc: print(f.split('/')[1], k, "%.6g per launch (n=%d)" % (acc[k] / max(n[k], 1), n[k]))
PY
