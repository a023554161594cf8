#!/usr/bin/env python3
"""Developer probe (GPU box): what does an event record BETWEEN two kernels of one stream cost?  The training step's schedule joins
streams through events (torch.cuda.Event / wait_stream); the kernel timeline shows ~13 us between two dX kernels that have one
record between them and ~0 between kernels that have none.  Variants: nothing, a torch event (hipEventDisableTiming), a raw HIP
event with hipEventReleaseToDevice (no system-scope release), a timing event; plus a cross-stream hand-over (record on A, wait on B)."""
import ctypes as C, sys, time
import torch
hip = C.CDLL("libamdhip64.so")
hipEventDisableTiming, hipEventReleaseToDevice = 0x2, 0x40000000
dev = torch.device("cuda")
x = torch.zeros(1 << 22, device=dev)


def kern():
    x.add_(1.0)          # a ~10-us kernel


def raw_event(flags):
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), flags) == 0
    return e


st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
side = torch.cuda.Stream()
sps = C.c_void_p(side.cuda_stream)


def run(label, between, n=2000):
    for _ in range(50):
        kern(); between(); kern()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        kern(); between(); kern()
    e1.record(); torch.cuda.synchronize()
    return label, e0.elapsed_time(e1) / n * 1e3


ev_t = torch.cuda.Event()
ev_tt = torch.cuda.Event(enable_timing=True)
ev_dev = raw_event(hipEventDisableTiming | hipEventReleaseToDevice)
ev_sys = raw_event(hipEventDisableTiming)
res = [run("nothing between", lambda: None),
       run("torch.cuda.Event().record()", lambda: ev_t.record()),
       run("torch timing event record", lambda: ev_tt.record()),
       run("raw hipEvent, DisableTiming", lambda: hip.hipEventRecord(ev_sys, sp)),
       run("raw hipEvent, DisableTiming | ReleaseToDevice", lambda: hip.hipEventRecord(ev_dev, sp))]


def handover(ev, rec, wait):
    def f():
        rec(ev)
        wait(ev)
    return f


def side_kernel():
    with torch.cuda.stream(side):
        x.mul_(1.0)


res.append(run("record (torch) + side stream waits + side kernel", lambda: (ev_t.record(), side.wait_event(ev_t), side_kernel())))
res.append(run("record (ReleaseToDevice) + side waits + side kernel",
               lambda: (hip.hipEventRecord(ev_dev, sp), hip.hipStreamWaitEvent(sps, ev_dev, 0), side_kernel())))
base = res[0][1]
for label, us in res:
    print(f"{label:56s}: {us:8.2f} us per (kernel, between, kernel)   (+{us - base:6.2f})")
