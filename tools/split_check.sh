#!/bin/bash
# A/B of the three-kernel main backward (PRESIGHT_MAIN_BWD_SPLIT=0: the single fused kernel): parity tests with the split on, then the bench with the split off / on
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fields.py tests/test_hip_model.py tests/test_hip_ms.py -q -x 2>&1 | tail -5
for sp in 0 1 1; do
  PRESIGHT_MAIN_BWD_SPLIT=$sp timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('split=$sp', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
done
PRESIGHT_MAIN_BWD_SPLIT=1 timeout 300 python bench.py --config cfg3 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('cfg3 split=1', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
PRESIGHT_MAIN_BWD_SPLIT=0 timeout 300 python bench.py --config cfg3 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('cfg3 split=0', round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
