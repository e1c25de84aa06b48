#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of the main-field kernels for the in-tree library and named variants
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -z "$v" ]; then unset PRESIGHT_HIP_LIB; name=base; else export PRESIGHT_HIP_LIB=$PWD/presight_amd/_variants/lib_$v.so; name=$v; fi
  bash tools/trace_run.sh tv_$name --steps 8 --warmup 3 > /dev/null 2>&1
  echo "== $name  $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_tv_$name.json)"
  grep "main_bwd\|main_fwd" gpurun_out/kernel_stats_tv_$name.txt | sed 's/(anonymous namespace):://g' | cut -c1-40,90-160
done
