#!/usr/bin/env python3
"""Developer experiment (GPU box): the 200-step loss-curve gate (tests/test_gpu_train.py::test_bf16_training_tracks_f32_over_200_steps)
with the recorded ACTIVATIONS re-rounded to a narrower MX format in front of the weight-gradient GEMMs (tools/diag_mx_narrow.py's
exact re-rounding, every step): does the 16-bit tier's loss curve still end on the f32 tier's?   python tools/diag_mx_narrow_curve.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "dfa-nerf_amd", "oracle", "tools"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
os.environ["DIAG_ONLY"] = "__none__"
import diag_mx_narrow as D            # (runs the shipping single step once; reuses requant / Proxy)
from dfanerf import frames, nets, run_nerf, training, synth
import test_gpu_train as T
t = T.t
scene, states, latents = D.scene, D.states, D.latents
dev = torch.device("cuda")
n, n_steps = 1024, 200
H, W = scene["H"], scene["W"]
args = run_nerf.config_parser().parse_args(
    f"--expname t --concate_bg --N_rand={n} --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
    "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000".split())
ds = [{"auds": t(scene["aud"]).to(dev), "exp": t(scene["exp"]).to(dev), "poses": t(scene["poses"]).to(dev),
       "bc_img": (t(scene["bg"]).float() / 255.0).to(dev), "hwfcxy": [H, W, scene["focal"], scene["cx"], scene["cy"]],
       "near": 0.3, "far": 0.9}]
zs, za = [t(v).to(dev) for v in latents]
embed_fn, _ = nets.get_embedder(3, 0)
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
img = torch.stack([0.5 + 0.4 * torch.sin(6 * xx + k) * torch.cos(5 * yy - k) for k in range(3)], -1)
gt = [((img.roll(7 * f, 1) * 255).to(torch.uint8).reshape(-1, 3).to(dev),
       ((1 - img).roll(5 * f, 0) * 255).to(torch.uint8).reshape(-1, 3).to(dev)) for f in range(4)]


def curve(tier, mode):
    mods = T._modules(states, dev)
    opts = {k: run_nerf.make_adam(m.parameters(), 5e-4) for k, m in mods.items()}
    buf = training.TrainBuffers(tier, n, dev)
    buf.signal_trainer = training.SignalTrainer(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                                ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    sampler = frames.PixelSampler(H, W, n, 0, dev, seed=77)
    losses = []
    keep = training.lib
    try:
        for k in range(n_steps):
            if mode:
                training.lib = D.Proxy(keep, buf, mode)          # (fresh: its `done` set is per step)
            f = k % 4
            loss, *_ = run_nerf.train_step_loss_hip(mods, ds, 0, f, sampler.draw(), gt[f][0], gt[f][1], zs, za, 300000, args,
                                                    scene["aud"].shape[0], embed_fn, ds[0]["poses"][0], buf)
            for o in opts.values():
                o.zero_grad()
            loss.backward()
            run_nerf.optimizer_steps(opts, 300000, args)
            losses.append(loss.detach())
    finally:
        training.lib = keep
    torch.cuda.synchronize()
    D.QERR.clear()
    return torch.stack(losses).cpu().numpy()


a = curve("f32", None)
for label, mode in (("MX-fp8 act (shipping)", None), ("act e2m3", [("act", "e2m3", 64)]), ("act e2m1", [("act", "e2m1", 64)])):
    b = curve("bf16", mode)
    final = abs(b[-20:].mean() - a[-20:].mean()) / a[-20:].mean()
    worst = float(np.max((np.abs(b - a) / a)[n_steps // 2:]))
    print(f"{label:24s}: final-loss difference {final:.3%}, worst step of the second half {worst:.3%}   [gates 2 % / 5 %]   "
          f"loss {b[0]:.4f} -> {b[-1]:.4f} (f32: {a[-1]:.4f})", flush=True)
