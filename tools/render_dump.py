#!/usr/bin/env python3
"""Developer tool: render fixed synthetic frames with the library DFN_LIB points at and dump every output, so that two
builds can be compared bit for bit:   DFN_LIB=a.so python tools/render_dump.py a.npz; DFN_LIB=b.so ... b.npz;
python tools/render_dump.py --compare a.npz b.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
        d = float(np.max(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)))) if a[k].shape == b[k].shape else -1
        print(f"{k:28s} {'bit-identical' if same else 'DIFFERENT'}  max|d|={d:.3e}")
        bad += not same
    sys.exit(1 if bad else 0)

import torch
from dfanerf import engine, synth
dev = torch.device("cuda:0")
sc = synth.bench_scene(0, n_frames=2)
st = synth.synth_all_states(0)
flat = engine.flatten_state(st["decoder"], dev)
zs, za = [torch.from_numpy(v).to(dev) for v in synth.synth_latents(0)]
g = torch.Generator().manual_seed(3)
sig_h = (torch.randn(96, generator=g) * 0.1).to(dev)
sig_t = (torch.randn(42, generator=g) * 0.1).to(dev)
H, W = sc["H"], sc["W"]
bg = (torch.from_numpy(sc["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
out = {}
for tier in ("bf16", "f32"):
    pk = engine.PackedDecoder(flat, tier)
    for fields in (1, 2):
        bias = pk.fold(sig_h, sig_t if fields == 2 else None, zs, za)
        for nf in (0, 64, 128):
            for cbg in (True, False):
                n = 202500 if (tier == "bf16" and nf == 128 and cbg) else 3001
                fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], sc["near"],
                                       sc["far"], n_fine=nf, fields=fields, ray_begin=0 if n == 202500 else 90000,
                                       ray_count=n, concate_bg=cbg)
                r = engine.render(pk, bias, fr, bg, want_weights=True, want_z=True)
                torch.cuda.synchronize()
                for name, t in zip(("rgb_h", "rgb_c", "w_h", "w_c", "z"), r):
                    if t is not None:
                        out[f"{tier}_f{fields}_nf{nf}_bg{int(cbg)}_{name}"] = t.cpu().numpy()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], len(out), "arrays")
