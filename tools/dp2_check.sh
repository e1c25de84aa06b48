#!/bin/bash
# two ranks time-slicing GPU 0 over gloo (functional check of the multi-rank path on a one-GPU box); a hang dumps the stacks
cd $GRAFT_REPO_ROOT
export PRESIGHT_SINGLE_DEVICE=1 PRESIGHT_DIST_BACKEND=gloo PRESIGHT_HANG_DUMP=${HANG:-90}
for extra in "--fixed-batches" ""; do
  echo "=== bench.py --gpus 2 $extra $@"
  timeout 200 python bench.py --gpus 2 --steps 3 --warmup 1 --rays 4096 --no-cpu-baseline $extra "$@" 2>&1 | tail -n 40 | cut -c1-300
done
