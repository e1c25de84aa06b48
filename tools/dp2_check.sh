#!/bin/bash
# functional 2-rank run on ONE GPU (gloo transport, both ranks on device 0) in directory $1 (default .)
cd ${1:-.}
PRESIGHT_DIST_BACKEND=gloo PRESIGHT_SINGLE_DEVICE=1 timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 > /tmp/dp2.log 2>&1
echo "$1 rc=$? $(grep -c metric /tmp/dp2.log) metric line(s) $(grep -o '"replicas_max_abs_diff": [0-9.e-]*' /tmp/dp2.log)"
