#!/usr/bin/env python3
"""Developer tool: wall time of the C2 render launch for the library DFN_LIB points at (no result checks: usable with
deliberately wrong experiment builds).  python tools/time_render.py [c2|c3] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import engine, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
fields = 2 if wl == "c3" else 1
dev = torch.device("cuda:0")
sc = synth.bench_scene(0, n_frames=2)
st = synth.synth_all_states(0)
flat = engine.flatten_state(st["decoder"], dev)
zs, za = [torch.from_numpy(v).to(dev) for v in synth.synth_latents(0)]
pk = engine.PackedDecoder(flat, sys.argv[2] if len(sys.argv) > 2 else "f16")
bias = pk.fold(torch.randn(96, device=dev) * 0.1, torch.randn(42, device=dev) * 0.1 if fields == 2 else None, zs, za)
H, W = sc["H"], sc["W"]
bg = (torch.from_numpy(sc["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], sc["near"], sc["far"],
                       n_fine=128, fields=fields)
for _ in range(3):
    engine.render(pk, bias, fr, bg)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    engine.render(pk, bias, fr, bg)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('DFN_LIB', 'intree')}: {e0.elapsed_time(e1) / reps:.3f} ms per frame ({wl}, {sys.argv[2] if len(sys.argv) > 2 else 'f16'})")
