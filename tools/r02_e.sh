#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_hip_ms.py -x -q -s ) > gpurun_out/gputest_r02e.txt 2>&1
tail -n 40 gpurun_out/gputest_r02e.txt
bash tools/trace_run.sh r02e_cfg3 --config cfg3 --steps 5 --warmup 2 > /dev/null 2>&1
head -n 60 gpurun_out/kernel_stats_r02e_cfg3.txt | cut -c1-160
