#!/bin/bash
# A/B of library variants copied to tools/_vlib (tools/build_variant.sh builds them under presight_amd/_variants, which does not travel):
#   gpurun -- 'bash tools/ab_variants.sh "<bench args>" <rounds> <variant> ...'   (the product library always runs first in a round)
cd $GRAFT_REPO_ROOT
args=$1; rounds=$2; shift; shift
cat > /tmp/ab_fmt.py <<'PY'
import json, os, sys
sys.stdin.read()
if sys.argv[2] != "0" or not os.path.exists("bench_detail.json"):  # (a failed / timed-out run must not report the previous variant's file)
    print(sys.argv[1], "FAILED rc=" + sys.argv[2], flush=True)
    sys.exit(0)
d = json.load(open("bench_detail.json"))  # (the stdout line is the compact line of record; per-kernel times live in the detail file)
k = d["kernels_ms"]
keep = {a: round(b, 3) for a, b in k.items() if any(t in a for t in ("scatter", "adam", "prop", "encode", "main_field", "main_bwd"))}
print(sys.argv[1], round(d["ms_per_step"], 2), keep)
PY
for r in $(seq $rounds); do
  for v in "" "$@"; do
    if [ -z "$v" ]; then unset PRESIGHT_HIP_LIB; name=product; else export PRESIGHT_HIP_LIB=$PWD/tools/_vlib/lib_$v.so; name=$v; fi
    rm -f bench_detail.json
    PRESIGHT_NO_DRY_OVERLAP=1 timeout 400 python bench.py $args --no-cpu-baseline --psnr-steps 0 --no-secondary > /tmp/ab_out.txt 2>/dev/null; rc=$?
    tail -1 /tmp/ab_out.txt | python /tmp/ab_fmt.py $name $rc
  done
done
