#!/bin/bash
# kernel trace of a short bench run -> every dispatch of one step + dispatches per step
#   gpurun -- 'bash tools/step_trace.sh <tag> [bench args]'   ->  gpurun_out/timeline_<tag>.txt, kernel_stats_<tag>.txt
TAG=${1:-dev}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf /tmp/trace_$TAG
timeout -k 5 420 rocprofv3 --kernel-trace -d /tmp/trace_$TAG -o trace -- python3 bench.py --no-cpu-baseline --psnr-steps 0 --no-secondary --steps 10 --warmup 3 "$@" > gpurun_out/bench_$TAG.json 2> /tmp/trace_$TAG.err
echo "rocprofv3 rc=$?"
DB=$(ls /tmp/trace_$TAG/*.db | head -n 1)
python3 tools/rocpd_timeline.py $DB 0 gather_batch 6 > gpurun_out/timeline_$TAG.txt 2>&1
python3 tools/rocpd_stats.py $DB 80 > gpurun_out/kernel_stats_$TAG.txt 2>&1
echo "dispatches in the step: $(($(wc -l < gpurun_out/timeline_$TAG.txt) - 1))"
head -1 gpurun_out/timeline_$TAG.txt | cut -c1-30
tail -1 gpurun_out/bench_$TAG.json | cut -c1-300
