#!/bin/bash
# Everything a round's profiles/ needs, in one GPU lease:  gpurun -- 'bash tools/profiles.sh <tag>'
#   un-profiled default bench line (with its secondary block), rocprofv3 kernel traces of cfg 2 / cfg 3 / cfg 4, PMC passes of cfg 2
T=${1:-r03}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_${T}.json 2> gpurun_out/bench_${T}.err
bash tools/trace_run.sh ${T}_cfg2 --steps 10 --warmup 3 --no-secondary > /dev/null 2>&1
bash tools/trace_run.sh ${T}_cfg3 --config cfg3 --steps 4 --warmup 2 > /dev/null 2>&1
bash tools/trace_run.sh ${T}_cfg3_8192 --config cfg3 --rays 8192 --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/trace_run.sh ${T}_cfg4 --config cfg4 --steps 4 --warmup 2 > /dev/null 2>&1
python3 bench.py --config extract --no-cpu-baseline > gpurun_out/bench_${T}_extract.json 2>/dev/null
bash tools/pmc_run.sh $T --no-secondary > /dev/null 2>&1
ls -la gpurun_out | grep $T
