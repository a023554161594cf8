#!/usr/bin/env python3
"""Developer tool: host-side (Python) profile of the TIMED training steps of bench.py c4 (warm-up excluded: the
profiler is switched on at the synchronize() that opens the timed region and off at the one that closes it)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--workload", "c4", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-extra",
            "--sustain-seconds", "0"]
import torch
import bench
pr = cProfile.Profile()
real_sync = torch.cuda.synchronize
state = {"n": 0}
def sync(*a, **k):
    r = real_sync(*a, **k)
    state["n"] += 1
    if state["n"] == 2: pr.enable()
    if state["n"] == 3: pr.disable()
    return r
torch.cuda.synchronize = sync
bench.main()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
txt = s.getvalue()
print(txt[:7000])
# the input stage must be device-side: no image decode and no host-to-device copy inside the timed steps
for needle in ("imread", "Image.open", "_imread", "pin_memory", "PinnedUpload"):
    print(f"calls matching {needle!r} in the timed region: {sum(needle in ln for ln in txt.splitlines())}")
