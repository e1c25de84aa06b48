#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -n 4
( time timeout 900 python bench.py --config extract --steps 2 --warmup 1 ) > gpurun_out/bench_r02i_extract.json 2> gpurun_out/bench_r02i_extract.err
tail -n 4 gpurun_out/bench_r02i_extract.err; cut -c1-1800 gpurun_out/bench_r02i_extract.json
