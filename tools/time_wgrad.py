#!/usr/bin/env python3
"""Developer tool (GPU box): the weight-gradient call of one training step (both fields, 2048 rays x 64 samples) in a
loop, wall clock and GPU events.   python tools/time_wgrad.py [bf16|f32]   (DFN_LIB / DFN_WGRAD_KSPLIT select variants)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import training
from dfanerf._lib import lib, check

tier = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
buf = training.TrainBuffers(tier, 2048, dev)
for f in (0, 1):
    for arr in (buf.act[f], buf.dy[f]):
        if arr.dtype == torch.uint8:      # 16-bit tier: MX-fp8 [tile][rows x 32 e4m3 | 128 scale bytes]
            arr.copy_((torch.randn(arr.shape, device=dev) * 8).to(torch.float8_e4m3fn).view(torch.uint8))
            arr[:, -128:] = 120
        else:
            arr.copy_(torch.randn_like(arr) * 0.1)
g_flat = torch.zeros(955242, device=dev)
gb = [torch.zeros(buf.nb[f], device=dev) for f in (0, 1)]
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def call():
    for f in (0, 1):
        check(lib.dfn_weight_bias_grad(buf.tier, f, p(buf.dy[f]), p(buf.act[f]), buf.NP, p(buf.ws[f]), p(g_flat), p(gb[f]), st), "wgrad")
for _ in range(5): call()
torch.cuda.synchronize()
n = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(n): call()
e1.record(); torch.cuda.synchronize()
nbytes = sum(t.numel() * t.element_size() for f in (0, 1) for t in (buf.act[f], buf.dy[f]))
ms = e0.elapsed_time(e1) / n
print(f"{tier}: wall {(time.perf_counter() - t0) / n * 1e3:.3f} ms  events {ms:.3f} ms per step   operands {nbytes / 1e9:.2f} GB -> {nbytes / ms / 1e9:.2f} TB/s if read once")
