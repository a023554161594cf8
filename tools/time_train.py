#!/usr/bin/env python3
"""Developer tool: the same training step (2048 rays, both fields, coarse, five gated Adams) three ways on one GPU:
HIP bf16 tier, HIP f32 tier (both with the HIP conditioning networks), and the reference-style ATen autograd path
(run_nerf.train_step_loss: torch ops for everything, what the reference does on a GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import numpy as np, torch
from dfanerf import synth, nets, run_nerf, training
from dfanerf.decoder import Decoder

def t(x): return torch.from_numpy(np.asarray(x))
dev = torch.device("cuda")
st = synth.synth_all_states(0); sc = synth.bench_scene(0)
args = run_nerf.config_parser().parse_args("--expname t --concate_bg --N_rand=2048 --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed --dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000".split())
H, W = 450, 450
ds = [{"auds": t(sc["aud"]).to(dev), "exp": t(sc["exp"]).to(dev), "poses": t(sc["poses"]).to(dev),
       "bc_img": (t(sc["bg"]).float() / 255).to(dev), "hwfcxy": [H, W, 1200., 225., 225.], "near": 0.3, "far": 0.9}]
zs, za = [t(v).to(dev) for v in synth.synth_latents(0)]
embed_fn, _ = nets.get_embedder(3, 0)
tgt = torch.rand(2048, 3, device=dev)
for tier in (sys.argv[1:] or ("bf16", "f32", "aten")):
    mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
            "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in st[k].items()}); m.to(dev)
    opts = {k: run_nerf.make_adam(m.parameters(), 5e-4) for k, m in mods.items()}
    buf = None
    if tier != "aten":
        buf = training.TrainBuffers(tier, 2048, dev)
        buf.signal_trainer = training.SignalTrainer(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                                    ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    rng = np.random.RandomState(0)
    def step():
        sel = run_nerf.select_coords(H, W, 2048, 0, None, rng)
        if tier == "aten":
            loss, *_ = run_nerf.train_step_loss(mods, ds, 0, 3, sel, tgt, tgt, zs, za, 300000, args, 8, embed_fn, ds[0]["poses"][0, :3, :4])
        else:
            loss, *_ = run_nerf.train_step_loss_hip(mods, ds, 0, 3, sel, tgt, tgt, zs, za, 300000, args, 8, embed_fn, ds[0]["poses"][0], buf)
        for o in opts.values(): o.zero_grad()
        loss.backward()
        run_nerf.optimizer_steps(opts, 300000, args)
        return loss
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tier}: {dt*1e3:.2f} ms/step  ({2048/dt:.0f} rays/s)  loss {l.item():.4f}")
