#!/bin/bash
# Same-box A/B of two source TREES (e.g. a git worktree of the previous round under _wt_r05/ against the working tree), alternating:
#   git worktree add _wt_r05 <commit> && (cd _wt_r05 && python -c 'import __graft_entry__ as g; g.build()')   # (_wt_*/ is git-ignored, travels with gpurun)
#   gpurun -- 'bash tools/ab_trees.sh "<bench args>" <rounds> <dir A> <dir B> ...'      ("." = the working tree)
# prints ms_per_step of every run; a failed run prints FAILED
root=$GRAFT_REPO_ROOT
args=$1; rounds=$2; shift; shift
for r in $(seq $rounds); do
  for d in "$@"; do
    cd $root/$d
    rm -f bench_detail.json
    PRESIGHT_NO_DRY_OVERLAP=1 timeout 400 python bench.py $args --no-cpu-baseline --psnr-steps 0 --no-secondary > /tmp/ab_out.txt 2>/dev/null; rc=$?
    if [ $rc -ne 0 ]; then echo "$d FAILED rc=$rc"; continue; fi
    tail -1 /tmp/ab_out.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', round(d['ms_per_step'],3), round(d['value']/1e6,3), 'M rays/s')"
  done
done
