#!/bin/bash
# tools/dbg/abl.sh <variant names...>: kernel-trace the bench under each variant library, print the matching kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PAT=${PAT:-"bin_kernel|accumulate"}
for v in "$@"; do
  if [ "$v" == "base" ]; then unset PRESIGHT_HIP_LIB; else export PRESIGHT_HIP_LIB=$PWD/presight_amd/_variants/lib_$v.so; fi
  rm -rf /tmp/tr_$v
  timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/tr_$v -o trace -- python3 bench.py --no-cpu-baseline --steps ${STEPS:-4} --warmup 2 ${BARGS} > /tmp/bench_$v.json 2> /tmp/tr_$v.err
  echo "== $v rc=$? $(python3 -c "import json;print(round(json.load(open('/tmp/bench_$v.json'))['ms_per_step'],2))" 2>/dev/null)"
  python3 tools/rocpd_stats.py $(ls /tmp/tr_$v/*.db | head -n 1) 80 | grep -E "$PAT" | cut -c1-60,90-150
done
