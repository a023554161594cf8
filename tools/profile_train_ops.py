#!/usr/bin/env python3
"""Developer tool: which torch ops launch the small kernels of a training step - torch.profiler over the TIMED steps of
bench.py --workload c4 (switched on / off at the synchronize() calls that bracket the timed region): ops by number of
calls per step with their host and device time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = 20
sys.argv = ["bench.py", "--workload", "c4", "--steps", str(STEPS), "--warmup", "5", "--no-cpu-baseline", "--no-extra",
            "--sustain-seconds", "0"]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
real_sync, state = torch.cuda.synchronize, {"n": 0}
def sync(*a, **k):
    r = real_sync(*a, **k)
    state["n"] += 1
    if state["n"] == 2: prof.__enter__()
    if state["n"] == 3: prof.__exit__(None, None, None)
    return r
torch.cuda.synchronize = sync
bench.main()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print(f"{'op':70s} calls/step  cpu us/step  device us/step")
for e in rows[:60]:
    dev = getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0))
    print(f"{e.key[:70]:70s} {e.count / STEPS:9.1f} {e.cpu_time_total / STEPS:11.1f} {dev / STEPS:11.1f}")

# who calls the fills / copies: innermost Python frame of this repo per aten::fill_ / aten::zero_ / aten::copy_ event
from collections import Counter
who = Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::ones_like"):
        st = [f for f in (ev.stack or []) if "dfanerf" in f or "bench.py" in f or "autograd" in f]
        who[(ev.name, st[0] if st else "?")] += 1
for (name, frame), c in who.most_common(40):
    print(f"{c / STEPS:6.1f}/step  {name:18s} {frame[:110]}")
