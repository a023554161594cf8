#!/usr/bin/env python3
"""Developer tool: is render_kernel limited by issue/stalls or by the power budget (DVFS)?  Times the same launch with
the real (random) weights and with all-zero weights (same instruction stream, far fewer toggling bits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import numpy as np
import torch
from dfanerf import engine, synth
dev = torch.device("cuda:0")
sc = synth.bench_scene(0, n_frames=2)
st = synth.synth_all_states(0)
flat = engine.flatten_state(st["decoder"], dev)
zs, za = [torch.from_numpy(v).to(dev) for v in synth.synth_latents(0)]
sig_h = torch.randn(96, device=dev) * 0.1
H, W = sc["H"], sc["W"]
bg = (torch.from_numpy(sc["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], sc["near"], sc["far"],
                       n_fine=128, fields=1)
for name, fl in (("random weights", flat), ("zero weights", torch.zeros_like(flat)), ("random weights", flat)):
    pk = engine.PackedDecoder(fl, "bf16")
    bias = pk.fold(sig_h, None, zs, za)
    for _ in range(3):
        engine.render(pk, bias, fr, bg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        engine.render(pk, bias, fr, bg)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 8:.2f} ms per frame")
