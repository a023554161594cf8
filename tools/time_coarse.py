#!/usr/bin/env python3
"""Developer tool: the inference render kernel on the training step's geometry (2048 rays, 64 coarse samples, both
fields, no fine pass) - what the training forward would cost without the recorder."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import engine, synth
dev = torch.device("cuda:0")
sc = synth.bench_scene(0, n_frames=2)
st = synth.synth_all_states(0)
flat = engine.flatten_state(st["decoder"], dev)
zs, za = [torch.from_numpy(v).to(dev) for v in synth.synth_latents(0)]
pk = engine.PackedDecoder(flat, "bf16")
bias = pk.fold(torch.randn(96, device=dev) * 0.1, torch.randn(42, device=dev) * 0.1, zs, za)
H, W = sc["H"], sc["W"]
bg = (torch.from_numpy(sc["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
for n in (2048, 4096, 8192, 16384):
    fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], sc["near"], sc["far"],
                           n_fine=0, fields=2, ray_begin=50000, ray_count=n)
    for _ in range(3): engine.render(pk, bias, fr, bg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): engine.render(pk, bias, fr, bg)
    e1.record(); torch.cuda.synchronize()
    print(f"{n} rays, coarse only, two fields: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
