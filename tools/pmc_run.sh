#!/bin/bash
# PMC passes over a short bench run (one counter group per pass), each guarded by a timeout; only text summaries are kept.
#   gpurun -- 'bash tools/pmc_run.sh [tag] [bench args]'  ->  gpurun_out/pmc_summary_<tag>.txt
TAG=${1:-dev}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/pmc
KERN="main_bwd,main_fwd,accumulate_kernel,bin_kernel,grid_encode,prop_bwd_kernel,prop_fwd_kernel,composite,absmax,adam,grid4,flow_,blend_,route,voxel,points,interlevel,per_ray"
OUT=gpurun_out/pmc_summary_$TAG.txt
: > $OUT
# the hash of the kernel sources THESE counters are collected on (tools/pmc_traffic.py stamps profiles/traffic.json with it)
echo "src_sha16=$(python3 -c 'import bench; print(bench.kernel_sources_sha())')" >> $OUT
# PMC_BASIC=1: the four groups that price a kernel against its roofline (HBM bytes, matrix-pipe busy, L2 hit rate) -- used for the
# secondary configurations (cfg 3 / cfg 4), whose steps are long
# PMC_GATHER=1: the basic groups + the L1 / L2 request counters that price a gather-bound pass (prior extraction, cfg 5)
if [ "${PMC_GATHER:-0}" = "1" ]; then
  GROUPS_LIST=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum")
elif [ "${PMC_BASIC:-0}" = "1" ]; then
  GROUPS_LIST=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum")
else
  GROUPS_LIST=("FETCH_SIZE" "WRITE_SIZE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM")
fi
i=0
for grp in "${GROUPS_LIST[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc/p$i
  timeout -k 5 240 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc/p$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --psnr-steps 0 "$@" > /tmp/pmc/p$i.log 2>&1
  echo "== pass $i: --pmc $grp (rc=$?)" >> $OUT
  python3 tools/rocpd_pmc.py $KERN /tmp/pmc/p$i/pmc_results.db >> $OUT 2>&1
  tail -n 3 /tmp/pmc/p$i.log | cut -c1-300 >> $OUT
done
python3 tools/rocpd_stats.py /tmp/pmc/p1/pmc_results.db 30 > gpurun_out/pmc_pass1_kernel_stats_$TAG.txt 2>&1
tail -n 40 $OUT | cut -c1-160
