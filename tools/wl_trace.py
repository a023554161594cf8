#!/usr/bin/env python3
"""Developer tool (GPU box, a -DDFN_WL_TRACE variant library): per-workgroup timeline of the 16-bit tier's weight-gradient launch.
   DFN_LIB=exp_libs/<name>.so python tools/wl_trace.py [field]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import numpy as np, torch
from dfanerf import training
from dfanerf._lib import lib, check, LIB_PATH
raw = C.CDLL(LIB_PATH)
dev = torch.device("cuda")
n_fine = int(os.environ.get("N_FINE", "0"))
buf = training.TrainBuffers("bf16", 2048, dev, n_fine=n_fine)
for f in (0, 1):
    for arr in (buf.act[f], buf.dy[f]):
        arr.copy_((torch.randn(arr.shape, device=dev) * 8).to(torch.float8_e4m3fn).view(torch.uint8))
        arr[:, -128:] = 120
g_flat = torch.zeros(955242, device=dev)
gb = [torch.zeros(buf.nb[f], device=dev) for f in (0, 1)]
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
trace = torch.zeros(2048, 4, dtype=torch.int64, device=dev)
for f in ([int(sys.argv[1])] if len(sys.argv) > 1 else [0, 1]):
    for _ in range(3):
        check(lib.dfn_weight_bias_grad(buf.tier, f, p(buf.dy[f]), p(buf.act[f]), buf.NP, p(buf.ws[f]), p(g_flat), p(gb[f]), st), "wgrad")
    torch.cuda.synchronize()
    trace.zero_()
    assert raw.dfn_debug_wl_trace(p(trace)) == 0
    check(lib.dfn_weight_bias_grad(buf.tier, f, p(buf.dy[f]), p(buf.act[f]), buf.NP, p(buf.ws[f]), p(g_flat), p(gb[f]), st), "wgrad")
    torch.cuda.synchronize()
    raw.dfn_debug_wl_trace(None)
    t = trace.cpu().numpy()
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    s, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # us
    print(f"field {f}: {len(t)} workgroups, span {e.max():.1f} us; starts: min {s.min():.1f} median {np.median(s):.1f} max {s.max():.1f}")
    for op in sorted(set(t[:, 2].tolist())):
        m = t[:, 2] == op
        d = e[m] - s[m]
        print(f"  op {op:2d}: {m.sum():3d} slices  start {s[m].min():6.1f}..{s[m].max():6.1f}  duration {d.min():6.1f} / {d.mean():6.1f} / {d.max():6.1f}  end max {e[m].max():6.1f}")
