#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -z "$v" ]; then unset PRESIGHT_HIP_LIB; name=baseline; else export PRESIGHT_HIP_LIB=$PWD/presight_amd/_variants/lib_$v.so; name=$v; fi
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$name', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
done
