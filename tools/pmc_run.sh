#!/bin/bash
# PMC passes over a short bench run (one counter group per pass), each guarded by a timeout; only text summaries are kept.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/pmc
KERN="main_bwd_kernel,main_fwd_kernel,accumulate_kernel,bin_kernel,grid_encode_kernel,prop_bwd_kernel,composite"
: > gpurun_out/pmc_summary.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc/p$i
  timeout -k 5 240 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc/p$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc/p$i.log 2>&1
  echo "== pass $i: --pmc $grp (rc=$?)" >> gpurun_out/pmc_summary.txt
  python3 tools/rocpd_pmc.py $KERN /tmp/pmc/p$i/pmc_results.db >> gpurun_out/pmc_summary.txt 2>&1
done
python3 tools/rocpd_stats.py /tmp/pmc/p1/pmc_results.db 30 > gpurun_out/pmc_pass1_kernel_stats.txt 2>&1
tail -60 gpurun_out/pmc_summary.txt
