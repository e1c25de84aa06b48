#!/bin/bash
# round-2 opening measurement: counter list, GPU tests, kernel trace, PMC passes of the state inherited from round 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -o "TCP_[A-Z0-9_a-z]*\|TA_[A-Z0-9_a-z]*\|TD_[A-Z0-9_a-z]*" gpurun_out/counters_list.txt | sort -u > gpurun_out/counters_tcp_ta.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/gputest_r02a.txt 2>&1
tail -n 5 gpurun_out/gputest_r02a.txt
bash tools/trace_run.sh r02a --steps 10 --warmup 3
bash tools/pmc_run.sh r02a
