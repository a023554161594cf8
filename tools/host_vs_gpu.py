#!/usr/bin/env python3
"""Developer tool (GPU box): is the training loop of bench.py c4 host-bound?  Runs the timed steps with a wall clock around
the enqueue loop alone (no synchronize inside) and another one after the final synchronize: if the first is close to the
second, the Python side (autograd, ctypes calls, stream bookkeeping) is what paces the step, not the GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--workload", "c4", "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-extra",
            "--sustain-seconds", "0"]
import torch
import bench
real_sync = torch.cuda.synchronize
marks = []
def sync(*a, **k):
    marks.append(("before", time.perf_counter()))
    r = real_sync(*a, **k)
    marks.append(("after", time.perf_counter()))
    return r
torch.cuda.synchronize = sync
bench.main()
# timed(n): sync, [barrier], sync, t0, loop, sync ... : the LAST 'before' preceded by an 'after' brackets the enqueue loop
pairs = [(marks[i][1], marks[i + 1][1], marks[i + 2][1]) for i in range(len(marks) - 2)
         if marks[i][0] == "after" and marks[i + 1][0] == "before" and marks[i + 2][0] == "after"]
t_open, t_enq, t_done = max(pairs, key=lambda p: p[1] - p[0])
print(f"enqueue loop {1e3 * (t_enq - t_open) / 300:.3f} ms/step (host alone), until the GPU is done {1e3 * (t_done - t_open) / 300:.3f} ms/step")
