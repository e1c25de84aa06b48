cd $GRAFT_REPO_ROOT
for cfgargs in "--config cfg3 --steps 5 --warmup 2" "--steps 8 --warmup 3 --no-secondary --psnr-steps 0"; do
for v in "" ppt1 ppt4; do
  if [ -z "$v" ]; then unset PRESIGHT_HIP_LIB; name=baseline; else export PRESIGHT_HIP_LIB=$PWD/presight_amd/_variants/lib_$v.so; name=$v; fi
  PRESIGHT_NO_DRY_OVERLAP=1 timeout 400 python bench.py $cfgargs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$name', '$cfgargs'[:14], round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items() if 'scatter' in a or a=='adam'})"
done; done
