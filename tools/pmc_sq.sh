#!/bin/bash
# the wave-state PMC passes of tools/pmc_run.sh only (matrix-pipe busy, instruction mix, wait states) for the MLP kernels
TAG=${1:-dev}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/pmc
KERN="main_bwd,main_fwd,prop_bwd_kernel,prop_fwd_kernel"
OUT=gpurun_out/pmc_sq_$TAG.txt
: > $OUT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pmc/q$i
  timeout -k 5 240 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc/q$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /tmp/pmc/q$i.log 2>&1
  echo "== pass $i: --pmc $grp (rc=$?)" >> $OUT
  python3 tools/rocpd_pmc.py $KERN /tmp/pmc/q$i/pmc_results.db >> $OUT 2>&1
  tail -n 2 /tmp/pmc/q$i.log | cut -c1-200 >> $OUT
done
cut -c1-220 $OUT
