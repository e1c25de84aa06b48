#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_hip_ms.py -x -q -s ) > gpurun_out/gputest_r02d.txt 2>&1
tail -n 50 gpurun_out/gputest_r02d.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_hip_ms.py ) > gpurun_out/gputest_r02d_all.txt 2>&1
tail -n 15 gpurun_out/gputest_r02d_all.txt
( timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/bench_r02d.json 2> gpurun_out/bench_r02d.err
tail -n 3 gpurun_out/bench_r02d.err; cut -c1-300 gpurun_out/bench_r02d.json
( timeout 600 python bench.py --config cfg3 --steps 5 --warmup 2 ) > gpurun_out/bench_r02d_cfg3.json 2> gpurun_out/bench_r02d_cfg3.err
tail -n 3 gpurun_out/bench_r02d_cfg3.err; cut -c1-300 gpurun_out/bench_r02d_cfg3.json
( timeout 600 python bench.py --config cfg3 --steps 5 --warmup 2 --rays 8192 ) > gpurun_out/bench_r02d_cfg3_8k.json 2> gpurun_out/bench_r02d_cfg3_8k.err
tail -n 3 gpurun_out/bench_r02d_cfg3_8k.err; cut -c1-300 gpurun_out/bench_r02d_cfg3_8k.json
