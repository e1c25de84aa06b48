#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_hip_dist.py -x -q ) > gpurun_out/gputest_r02b.txt 2>&1
tail -n 25 gpurun_out/gputest_r02b.txt
( time timeout 600 python bench.py ) > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err
tail -n 3 gpurun_out/bench_r02b.err; cut -c1-1500 gpurun_out/bench_r02b.json
( time timeout 600 python bench.py --config cfg3 --steps 5 --warmup 2 ) > gpurun_out/bench_r02b_cfg3.json 2> gpurun_out/bench_r02b_cfg3.err
tail -n 3 gpurun_out/bench_r02b_cfg3.err; cut -c1-600 gpurun_out/bench_r02b_cfg3.json
