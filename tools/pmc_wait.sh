#!/bin/bash
# where do the waves of the dominant kernels wait?  SQ wait/active counters (one pass each), text summary only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/pmcw
KERN="main_bwd_kernel,main_fwd_kernel,bin_kernel,grid_encode_kernel,accumulate_kernel"
: > gpurun_out/pmc_wait.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pmcw/p$i
  timeout -k 5 240 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmcw/p$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmcw/p$i.log 2>&1
  echo "== pass $i: --pmc $grp (rc=$?)" >> gpurun_out/pmc_wait.txt
  python3 tools/rocpd_pmc.py $KERN /tmp/pmcw/p$i/pmc_results.db >> gpurun_out/pmc_wait.txt 2>&1
done
cat gpurun_out/pmc_wait.txt | cut -c1-120
