#!/usr/bin/env python3
"""Developer tool: read the in-kernel cycle counters of a -DDFN_TIMING build (tools/build_variant.sh timing -DDFN_TIMING).
  DFN_LIB=exp_libs/timing.so python tools/kernel_timing.py [c2|c3]
Per ray (= per wave): total shader cycles, 100 MHz ticks, cycles inside the MLP passes, inside sample_pdf/merge, and
at the slab hand-over (waitcnt / barrier / DMA issue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import numpy as np
import torch
from dfanerf import engine, synth
from dfanerf._lib import FIELD_HEAD, FIELD_TORSO

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
fields = 2 if wl == "c3" else 1
dev = torch.device("cuda:0")
sc = synth.bench_scene(0, n_frames=2)
st = synth.synth_all_states(0)
flat = engine.flatten_state(st["decoder"], dev)
pk = engine.PackedDecoder(flat, sys.argv[2] if len(sys.argv) > 2 else "f16")
zs, za = [torch.from_numpy(v).to(dev) for v in synth.synth_latents(0)]
sig_h = torch.randn(96, device=dev) * 0.1
sig_t = torch.randn(42, device=dev) * 0.1
bias = pk.fold(sig_h, sig_t if fields == 2 else None, zs, za)
H, W = sc["H"], sc["W"]
bg = (torch.from_numpy(sc["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], sc["near"], sc["far"],
                       n_fine=128, fields=fields)
for _ in range(2):
    out = engine.render(pk, bias, fr, bg, want_z=True)
torch.cuda.synchronize()
z = out[-1].cpu().numpy()[:, 64:72].astype(np.float64)
tot, real, mlp, pdf, wait, bar, issue, wave = z.T
print(f"rays {len(tot)}  clock = {np.mean(tot / real) * 100:.0f} MHz (shader cycles per 100 MHz tick)")
npass = 6 * fields        # 2 coarse + 4 fine tiles per field
print(f"per ray: total {tot.mean():.0f} cyc | mlp {mlp.mean():.0f} ({mlp.mean()/tot.mean():.1%}) = {mlp.mean()/npass:.0f} per pass"
      f" (ideal MFMA-only, 2 waves/SIMD: {2*32*(1138 if fields==1 else (1138+1302)/2):.0f}) | pdf+merge {pdf.mean():.0f} ({pdf.mean()/tot.mean():.1%})")
print(f"slab hand-over per ray: waitcnt {wait.mean():.0f} ({wait.mean()/tot.mean():.1%})  barrier {bar.mean():.0f} ({bar.mean()/tot.mean():.1%})"
      f"  dma issue {issue.mean():.0f} ({issue.mean()/tot.mean():.1%})")
for w in range(8):
    m = wave == w
    print(f"  wave {w}: total {tot[m].mean():.0f} wait {wait[m].mean():.0f} barrier {bar[m].mean():.0f} issue {issue[m].mean():.0f}")
