#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/trace_cfg3
timeout -k 5 420 rocprofv3 --kernel-trace -d /tmp/trace_cfg3 -o trace -- python3 tools/run_cfg3.py 16 8192 > gpurun_out/cfg3_run.txt 2>&1
python3 tools/rocpd_stats.py $(ls /tmp/trace_cfg3/*.db | head -n 1) 25 > gpurun_out/kernel_stats_cfg3.txt 2>&1
tail -n 2 gpurun_out/cfg3_run.txt; head -n 22 gpurun_out/kernel_stats_cfg3.txt | cut -c1-140; tail -n 1 gpurun_out/kernel_stats_cfg3.txt
