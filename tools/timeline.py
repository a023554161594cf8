#!/usr/bin/env python3
"""Developer tool: per-step kernel timeline of the training step from a rocprofv3 --kernel-trace csv.

   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $REPO/tools/time_train.py bf16
   python tools/timeline.py $(find /tmp/tl -name '*kernel_trace.csv') [step]

Prints the kernels of one steady-state step (delimited by the launches of the fused forward, render_kernel) with their
queue, start offset and duration, so that what really overlaps with what can be read off (the kernel-trace serialises
nothing: the streams run as they do un-profiled)."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
k = lambda r, *names: next(r[n] for n in names if n in r)
ev = sorted(((int(k(r, "Start_Timestamp")), int(k(r, "End_Timestamp")), k(r, "Kernel_Name"), k(r, "Queue_Id", "Stream_Id"))
             for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "render_kernel" in e[2]]
a, b = starts[which], starts[which + 1]
# the step's front end (signal encoders, fold, pack) precedes the forward: walk back to the previous step's last Adam
while a > 0 and "adam_multi" not in ev[a - 1][2]:
    a -= 1
while b > 0 and "adam_multi" not in ev[b - 1][2]:
    b -= 1
t0 = ev[a][0]
qs = {}
print(f"{'start us':>9} {'dur us':>8}  q  kernel")
for s, e, name, q in ev[a:b]:
    qi = qs.setdefault(q, len(qs))
    short = name.split("(")[0].replace("void ", "").replace("dfn::", "")
    short = short if len(short) < 70 else short[:67] + "..."
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {qi}  {'    ' * qi}{short}")
print(f"step: {(ev[b][0] - t0) / 1e3:.1f} us from the first kernel of the step to the first kernel of the next; "
      f"{b - a} launches, {len(qs)} queues")
