#!/usr/bin/env python3
"""Developer tool: repeatability soak of the render launch (the asm fragment fetch must never pick up stale data): the
same full frame N times, every output compared bit for bit with the first.  python tools/soak.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import engine, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
sc = synth.bench_scene(0, n_frames=2)
st = synth.synth_all_states(0)
flat = engine.flatten_state(st["decoder"], dev)
zs, za = [torch.from_numpy(v).to(dev) for v in synth.synth_latents(0)]
pk = engine.PackedDecoder(flat, "bf16")
H, W = sc["H"], sc["W"]
bg = (torch.from_numpy(sc["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
bad = 0
for fields, nf in ((1, 128), (2, 128), (1, 0), (2, 64)):
    bias = pk.fold(torch.full((96,), 0.1, device=dev), torch.full((42,), -0.1, device=dev) if fields == 2 else None, zs, za)
    fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], sc["near"], sc["far"],
                           n_fine=nf, fields=fields)
    ref = [o.clone() for o in engine.render(pk, bias, fr, bg) if o is not None]
    n_bad = 0
    for i in range(reps):
        out = [o for o in engine.render(pk, bias, fr, bg) if o is not None]
        if not all(torch.equal(a, b) for a, b in zip(out, ref)):
            n_bad += 1
    torch.cuda.synchronize()
    print(f"fields={fields} n_fine={nf}: {reps} launches, {n_bad} differ from the first")
    bad += n_bad
sys.exit(1 if bad else 0)
