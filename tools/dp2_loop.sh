#!/bin/bash
# tools/dp2_loop.sh [N]: the two-rank bench (both exchange modes) N times in one lease; on a failure / timeout the per-rank stack
# dumps (dp2_hang_rank*.txt) and collective sequence logs (dp2_comm_rank*.log) stay in gpurun_out/
cd $GRAFT_REPO_ROOT
N=${1:-25}
ok=0; bad=0
for i in $(seq 1 $N); do
  if timeout 400 python -m pytest tests/test_hip_dist.py -x -q -k "two_ranks" > /tmp/dp2_$i.log 2>&1; then
    if grep -q "skipped" /tmp/dp2_$i.log; then bad=$((bad+1)); echo "run $i: SKIPPED (hang)"; tail -n 5 /tmp/dp2_$i.log; cp gpurun_out/dp2_comm_rank0.log gpurun_out/dp2_hang${i}_comm_rank0.log 2>/dev/null; cp gpurun_out/dp2_comm_rank1.log gpurun_out/dp2_hang${i}_comm_rank1.log 2>/dev/null; else ok=$((ok+1)); fi
  else bad=$((bad+1)); echo "run $i: FAILED"; tail -n 15 /tmp/dp2_$i.log; fi
done
echo "dp2 loop: $ok clean, $bad not clean of $N"
