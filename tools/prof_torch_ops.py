import sys, os, json, collections
sys.path.insert(0, "/root/repo")
import torch
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-exact-f32"]
from torch.profiler import profile, ProfilerActivity
import runpy
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    try:
        runpy.run_path("/root/repo/bench.py", run_name="__main__")
    except SystemExit:
        pass
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 and ("aten::" in e.key)]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:40s} calls {e.count:5d} cuda_us {e.device_time_total:10.0f} shapes {str(e.input_shapes)[:90]}")
