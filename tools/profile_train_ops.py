#!/usr/bin/env python3
"""Developer tool: which torch ops launch the small kernels of a training step (torch.profiler over tools/time_train.py's
bf16 loop): ops by number of calls per step with their device time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "bf16"]
import torch
from torch.profiler import profile, ProfilerActivity
src = open(os.path.join(ROOT, "tools", "time_train.py")).read()
src = src.replace("    n = 20\n", "    n = 20\n    prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]); prof.__enter__()\n")
src = src.replace("    torch.cuda.synchronize(); dt =", "    torch.cuda.synchronize(); prof.__exit__(None, None, None); dt =")
g = {"__name__": "__main__", "__file__": os.path.join(ROOT, "tools", "time_train.py"), "profile": profile, "ProfilerActivity": ProfilerActivity}
exec(compile(src, "time_train.py", "exec"), g)
ka = g["prof"].key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print(f"{'op':60s} calls/step  cpu us/step  device us/step")
for e in rows[:45]:
    dev = getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0))
    print(f"{e.key[:60]:60s} {e.count / 20:9.1f} {e.cpu_time_total / 20:11.1f} {dev / 20:11.1f}")
