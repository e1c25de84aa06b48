#!/bin/bash
# compare library variants (tools/build_variant.sh) on a bench configuration:  bash tools/run_bin_variants.sh "<bench args>" <variant> ...
cd $GRAFT_REPO_ROOT
args=$1; shift
for v in "" "$@"; do
  if [ -z "$v" ]; then unset PRESIGHT_HIP_LIB; name=baseline; else export PRESIGHT_HIP_LIB=$PWD/presight_amd/_variants/lib_$v.so; name=$v; fi
  PRESIGHT_NO_DRY_OVERLAP=1 timeout 400 python bench.py $args --no-cpu-baseline --psnr-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$name', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'scatter' in a or a=='adam'})"
done
