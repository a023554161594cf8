#!/usr/bin/env python3
"""Developer tool (GPU box): the dX chain (dfn_mlp_bwd) of one training step alone, per field, in a loop.
   python tools/time_dx.py [bf16|f32]      (DFN_LIB selects a variant library, e.g. a -DDFN_NOPUT build)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import training, synth
from dfanerf.decoder import Decoder
from dfanerf._lib import lib, check

tier = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
buf = training.TrainBuffers(tier, 2048, dev)
dec = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_all_states(0)["decoder"].items()})
dec.to(dev)
flat = buf.bind(dec)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for f in (0, 1):
    check(lib.dfn_pack_weights_bwd(buf.tier, f, p(flat), p(buf.packed_T[f]), st), "pack")
    buf.masks[f].copy_(torch.randint(-2**31, 2**31 - 1, buf.masks[f].shape, device=dev, dtype=torch.int32))
buf.samples.copy_(torch.rand_like(buf.samples))
buf.dsamples.copy_(torch.randn_like(buf.dsamples) * 1e-3)
if os.environ.get("DX_ZERO"):         # power probe: the same instruction stream on all-zero gradients (operands that do not toggle)
    buf.dsamples.zero_()
for f in (0, 1):
    call = lambda: check(lib.dfn_mlp_bwd(buf.tier, f, p(buf.packed_T[f]), p(buf.samples), p(buf.dsamples), p(buf.masks[f]),
                                         buf.NP, p(buf.dy[f]), st), "dfn_mlp_bwd")
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    gb = buf.dy[f].numel() * buf.dy[f].element_size() / 1e9
    print(f"{tier} field {f}: {us:.1f} us per launch, dy_T {gb:.2f} GB -> {gb / us * 1e3:.2f} TB/s written")
    if hasattr(lib, "dfn_debug_bwd_timing"):
        import numpy as np
        nw = buf.NP // 32
        out = np.zeros(6 * 8192, dtype=np.uint64)
        lib.dfn_debug_bwd_timing(C.c_void_p(out.ctypes.data), C.c_long(out.size))
        t = out.reshape(-1, 6)[:min(nw, 8192)].astype(np.float64)
        tot, wait, bar, real, gemm, epi = t.T
        print(f"   per wave: total {tot.mean():.0f} cycles ({real.mean() / 100:.1f} us, {tot.mean() / real.mean() * 100:.0f} MHz), hand-over waitcnt "
              f"{wait.mean():.0f} ({wait.mean() / tot.mean():.1%}), barrier {bar.mean():.0f} ({bar.mean() / tot.mean():.1%}); "
              f"slowest wave {tot.max():.0f}, fastest {tot.min():.0f}; in bwd_layer: MFMA phases {gemm.mean():.0f} ({gemm.mean() / tot.mean():.1%}), "
              f"mask + convert epilogues {epi.mean():.0f} ({epi.mean() / tot.mean():.1%})")
