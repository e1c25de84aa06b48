#!/bin/bash
# rocprofv3 kernel trace of the default bench command, summarised to text (the rocpd .db is too large to keep).
#   gpurun -- 'bash tools/trace_run.sh <tag> [bench args]'   ->  gpurun_out/kernel_stats_<tag>.txt, bench_<tag>.json
TAG=${1:-dev}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf /tmp/trace_$TAG
timeout -k 5 420 rocprofv3 --kernel-trace -d /tmp/trace_$TAG -o trace -- python3 bench.py --no-cpu-baseline --psnr-steps 0 "$@" > gpurun_out/bench_$TAG.json 2> /tmp/trace_$TAG.err
echo "rocprofv3 rc=$?"
python3 tools/rocpd_stats.py $(ls /tmp/trace_$TAG/*.db | head -n 1) 60 > gpurun_out/kernel_stats_$TAG.txt 2>&1
python3 tools/rocpd_gaps.py $(ls /tmp/trace_$TAG/*.db | head -n 1) 45 > gpurun_out/kernel_gaps_$TAG.txt 2>&1
head -n 45 gpurun_out/kernel_stats_$TAG.txt | cut -c1-150
cat gpurun_out/bench_$TAG.json | cut -c1-400
