#!/bin/bash
# round-2 closing measurement: GPU tests, bench lines (cfg 2, cfg 3 at 65536 and 8192 rays, extraction), kernel traces, PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02}
( time timeout 1800 python -m pytest tests -m gpu -q -s ) > gpurun_out/gputest_$TAG.txt 2>&1
tail -n 4 gpurun_out/gputest_$TAG.txt
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -n 2 > gpurun_out/smoke_$TAG.txt; cat gpurun_out/smoke_$TAG.txt
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cut -c1-200 gpurun_out/bench_$TAG.json
timeout 600 python bench.py --config cfg3 --steps 8 --warmup 3 > gpurun_out/bench_${TAG}_cfg3.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_cfg3.json
timeout 600 python bench.py --config cfg3 --steps 8 --warmup 3 --rays 8192 > gpurun_out/bench_${TAG}_cfg3_8192.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_cfg3_8192.json
timeout 600 python bench.py --config extract --steps 2 --warmup 1 > gpurun_out/bench_${TAG}_extract.json 2>/dev/null; cut -c1-200 gpurun_out/bench_${TAG}_extract.json
bash tools/trace_run.sh ${TAG}_cfg2 --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/trace_run.sh ${TAG}_cfg3 --config cfg3 --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/pmc_run.sh $TAG > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_summary_$TAG.txt gpurun_out/traffic_$TAG.json
