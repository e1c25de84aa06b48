#!/bin/bash
# Everything a round's profiles/ needs, in one GPU lease:  gpurun -- 'bash tools/profiles.sh <tag>'
#   un-profiled default bench line (with its secondary block, psnr_after_k_steps and the dry run of the overlapped exchange), rocprofv3
#   kernel traces of cfg 2 / cfg 3 (65 536 and 8192 rays) / cfg 4, the cfg-5 lines (production tile and cfg-2 fields), PMC passes of
#   cfg 2 (all counter groups) and of cfg 3 / cfg 4 (HBM bytes, matrix-pipe busy, L2 hit rate)
T=${1:-r05}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PRESIGHT_NO_DRY_OVERLAP=1   # (the traced / counted runs measure the plain step; the default line below carries the dry run)
PRESIGHT_NO_DRY_OVERLAP=0 python3 bench.py > gpurun_out/bench_${T}.json 2> gpurun_out/bench_${T}.err
cp bench_detail.json gpurun_out/bench_${T}_detail.json
python3 tools/soak.py 1000 > gpurun_out/soak_${T}.txt 2>&1
bash tools/trace_run.sh ${T}_cfg2 --steps 10 --warmup 3 --no-secondary > /dev/null 2>&1
bash tools/trace_run.sh ${T}_cfg3 --config cfg3 --steps 4 --warmup 2 > /dev/null 2>&1
bash tools/trace_run.sh ${T}_cfg3_8192 --config cfg3 --rays 8192 --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/trace_run.sh ${T}_cfg4 --config cfg4 --steps 4 --warmup 2 > /dev/null 2>&1
python3 bench.py --config extract > gpurun_out/bench_${T}_extract.json 2>/dev/null
cp bench_detail.json gpurun_out/bench_${T}_extract_detail.json
python3 bench.py --config extract --extract-model cfg2 --no-cpu-baseline > gpurun_out/bench_${T}_extract_cfg2.json 2>/dev/null
bash tools/pmc_run.sh $T --no-secondary > /dev/null 2>&1
PMC_BASIC=1 bash tools/pmc_run.sh ${T}_cfg3 --config cfg3 > /dev/null 2>&1
PMC_BASIC=1 bash tools/pmc_run.sh ${T}_cfg4 --config cfg4 > /dev/null 2>&1
ls -la gpurun_out | grep $T
