"""aten operator counts of one cfg-3-shaped (K = 16) training step"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["run_cfg3.py", "16", "8192"]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "run_cfg3.py"))
tr, batches = ns["tr"], ns["batches"]
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.step(batches[0])
torch.cuda.synchronize()
rows = [(e.count, e.key, str(e.input_shapes)[:100]) for e in prof.key_averages(group_by_input_shape=False) if e.key.startswith("aten::")]
rows.sort(reverse=True)
skip = ("empty", "as_strided", "view", "reshape", "select", "slice", "detach", "unsqueeze", "squeeze", "narrow", "expand", "alias", "result_type", "_unsafe_view", "t", "transpose", "permute", "is_")
for c, k, sh in rows:
    if not any(k == "aten::" + s or k.startswith("aten::" + s + "_") for s in skip):
        print(f"{c:5d} {k}")
