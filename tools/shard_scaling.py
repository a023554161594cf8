#!/usr/bin/env python3
"""Developer tool: what one rank of an N-GPU run does per frame (conditioning signals + fold + render of its ray shard),
measured on ONE GPU for N = 1, 2, 4, 8 -> the strong-scaling efficiency the partition allows (no collective)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import engine, nets, synth
dev = torch.device("cuda:0")
F = 8
sc = synth.bench_scene(0, n_frames=F)
st = synth.synth_all_states(0)
flat = engine.flatten_state(st["decoder"], dev)
pk = engine.PackedDecoder(flat, "bf16", fields=(0,))
zs, za = [torch.from_numpy(v[0]).to(dev) for v in synth.synth_latents(0)]
mods = [nets.AudioNet_W2L(), nets.ExpressionEnc(), nets.AudioAttNet(96, 4), nets.AudioAttNet(42, 8)]
for m, k in zip(mods, ("AudNet", "ExpNet", "AudAttNet", "PoseAttNet")):
    m.load_state_dict({kk: torch.from_numpy(v) for kk, v in st[k].items()})
    m.to(dev)
enc = engine.SignalEncoder(*mods, torch.from_numpy(sc["aud"]).to(dev), torch.from_numpy(sc["exp"]).to(dev),
                           torch.from_numpy(sc["poses"]).to(dev))
fid = [torch.tensor([f], dtype=torch.int32, device=dev) for f in range(F)]
bg = torch.from_numpy(sc["bg"]).reshape(-1, 3).to(dev)
H, W = sc["H"], sc["W"]
R = H * W
base = None
# --two-streams: consecutive frames alternate between two streams (each with its own blob and output): the workgroups of
# frame k + 1 start on the compute units that frame k's last, partial round of workgroups leaves idle
TWO = "--two-streams" in sys.argv
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)] if TWO else [torch.cuda.current_stream(dev)] * 2
for n in (1, 2, 4, 8):
    per = (R + n - 1) // n
    outs = [torch.empty(per, 3, device=dev) for _ in range(2)]
    biases = [None, None]
    def step(i):
        f = i % F
        with torch.cuda.stream(streams[i & 1]):
            s2, _ = enc.encode(fid[f], 4, 8)
            biases[i & 1] = pk.fold(s2[0], None, zs, za, out=biases[i & 1])
            fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][f], sc["pose_body"], sc["near"],
                                   sc["far"], ray_begin=0, ray_count=per, n_fine=128, fields=1)
            engine.render(pk, biases[i & 1], fr, bg, out_head=outs[i & 1])
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 24
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    base = base or ms
    print(f"N={n}: {per} rays/rank  {ms:.3f} ms/frame/rank  -> speed-up {base / ms:.2f}x, efficiency {base / ms / n:.1%}")
