#!/bin/bash
# SQ-side PMC counters for the fused DDQN kernel (own runs, kernel-trace only; never together with other trace domains).
# usage: tools/pmc_round.sh [tag]  -> gpurun_out/pmc_sq_<tag>.txt (copy into profiles/ to keep)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
python3 - $TAG <<'PY'
import csv, glob, collections, sys
out = open('gpurun_out/pmc_sq_%s.txt' % sys.argv[1], 'w')
for f in sorted(glob.glob('gpurun_out/pmc_SQ_*/*/*counter_collection.csv')):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if 'ddqn_se_inner' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in acc:
        line = "%s %s %.6g per launch (n=%d dispatch rows)" % (f.split('/')[1], k, acc[k] / max(n[k], 1), n[k])
        print(line); out.write(line + "\n")
PY
