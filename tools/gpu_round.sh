#!/bin/bash
# One GPU-box round (run via gpurun from the repo root): bench with the CPU baseline, rocprofv3 kernel trace of the headline and of
# every other BASELINE configuration (one run and one CSV per configuration), the HBM-traffic PMC passes (separate runs, kernel-trace
# only, as MI355X_MICROARCH.md prescribes) for the three big kernels, and the SQ counter sets.  Outputs: gpurun_out/ -> profiles/.
# usage: tools/gpu_round.sh [tag]   (before the gpurun call, `rm -rf gpurun_out/pmc* gpurun_out/prof*` in the container: gpurun MERGES
# the box's files into the local scratch, and counter files of earlier calls would be summarised along with the new ones)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
mkdir -p $R/gpurun_out
cd $R
# the driver's own command line: the bench line and the profiled headline run are THIS command
DRIVER_ARGS="--gpus 1 --steps 20 --warmup 5"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc_* $R/gpurun_out/prof_cfg* $R/gpurun_out/pmccfg_*
# one-rank RCCL run of the same bench command under a launcher (a communicator exists: ranks.backend rccl, collective_ran true)
(cd $R && HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py $DRIVER_ARGS --no-cpu-baseline --no-configs 2>gpurun_out/bench_rccl1.err | tail -1 > gpurun_out/bench_rccl1.json)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py $DRIVER_ARGS --no-cpu-baseline --no-configs > $R/gpurun_out/prof_run.log 2>&1      # (the CPU-only baseline leg and the other configurations' shards have their own runs below)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $R/gpurun_out/pmc_$c.log 2>&1
done
for n in 1 2 3 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg$n -- python3 $R/bench.py --only-config $n > $R/gpurun_out/prof_cfg$n.json 2> $R/gpurun_out/prof_cfg$n.log
done
for n in 2 4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmccfg_${n}_$c -- python3 $R/bench.py --only-config $n > $R/gpurun_out/pmccfg_${n}_$c.log 2>&1
  done
done
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq_$t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $R/gpurun_out/pmc_sq_$t.log 2>&1
done
for n in 2 4; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
    t=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmccfg_${n}_sq_$t -- python3 $R/bench.py --only-config $n > $R/gpurun_out/pmccfg_${n}_sq_$t.log 2>&1
  done
done
cd $R
# the bench line's roofline.traffic is read from profiles/<tag>_summary.json -- so THIS round's PMC passes are condensed
# first, the bench line is taken afterwards (it then carries this round's FETCH_SIZE / WRITE_SIZE figure and names the file), and the
# summary is written once more with the line attached
rm -f gpurun_out/bench.json
python3 tools/summarize_profiles.py $TAG > /dev/null
cp profiles/${TAG}_summary.json gpurun_out/
python bench.py $DRIVER_ARGS 2>gpurun_out/bench.err | tail -1 | tee gpurun_out/bench.json | cut -c1-400
python3 tools/summarize_profiles.py $TAG
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
