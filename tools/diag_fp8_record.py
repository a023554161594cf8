#!/usr/bin/env python3
"""Developer experiment (GPU box): what would 8-bit storage of the recorded arrays cost in gradient accuracy?
The training step (2048 rays, bf16 tier, the full-size test's setting) with the recorded pre-activation gradients dy_T and /
or activations act_T rounded to MX-style fp8 (one power-of-two scale per row and 32-point tile = the block the
mfma_scale_f32_32x32x64_f8f6f4 instruction scales) between the dX chain and the weight-gradient GEMMs, emulated in torch;
gradient norms / whole-tensor errors against torch CPU autograd through the oracle, next to the plain bf16 step."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "dfa-nerf_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
import dfa_oracle as O
from dfanerf import synth, nets, run_nerf, training
import test_gpu_train as T
t = T.t
scene = synth.bench_scene(0, n_frames=8); states = synth.synth_all_states(0); latents = synth.synth_latents(0)
dev = torch.device("cuda")
step, n = 300000, 2048
H, W = scene["H"], scene["W"]
flat_px = np.random.RandomState(11).permutation(H * W)[:n]
sel = np.stack([flat_px // W, flat_px % W], axis=1).astype(np.int64)
tgt_h = t(synth.synth_tensor(0, "g8/th", (H, W, 3), 0.5)) + 0.5
tgt_c = t(synth.synth_tensor(0, "g8/tc", (H, W, 3), 0.5)) + 0.5
ref_loss, ref_g = T._oracle_full_step(states, scene, latents, sel, tgt_h, tgt_c, step)
training._OVERLAP = False          # one stream: the quantisation below sits between the dX chain and the GEMMs


QERR = []
BASE = {}


def quant(x, rows, fmt, rb=1):
    """x: bf16 [rows, NP] stored tile-major [NP/32][rows][32]; fp8 with a power-of-two scale per (tile, block of rb rows)"""
    v = x.view(-1, rows, 32).float()
    if rb == 1:
        amax = v.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    else:
        pad = (-rows) % rb
        vp = torch.nn.functional.pad(v.abs(), (0, 0, 0, pad)).view(v.shape[0], -1, rb, 32)
        amax = vp.amax((-1, -2), keepdim=True).expand(-1, -1, rb, 1).reshape(v.shape[0], -1, 1)[:, :rows].clamp_min(1e-30)
    emax = 7 if fmt == torch.float8_e4m3fn else 14          # amax / scale in [2^emax, 2^(emax+1)): below the format maximum
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    q = (v / scale).to(fmt).float() * scale
    QERR.append(float((q - v).norm() / v.norm()))
    x.view(-1, rows, 32).copy_(q.to(x.dtype))


class Proxy:
    def __init__(self, lib, buf, mode):
        self._lib, self._buf, self._mode = lib, buf, mode
    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if k != "dfn_weight_bias_grad" or not self._mode:
            return f
        def wrapped(tier, field, *a):
            b = self._buf
            for what, fmt, *rb in self._mode:
                arr = b.dy[field] if what == "dy" else b.act[field]
                quant(arr, arr.shape[0], fmt, *(rb or [1]))
            return f(tier, field, *a)
        return wrapped


def run(mode, label):
    mods = T._modules(states, dev)
    args = run_nerf.config_parser().parse_args(
        "--expname t --concate_bg --N_rand=2048 --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
        "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000".split())
    ds = [{"auds": t(scene["aud"]).to(dev), "exp": t(scene["exp"]).to(dev), "poses": t(scene["poses"]).to(dev),
           "bc_img": (t(scene["bg"]).float() / 255.0).to(dev), "hwfcxy": [H, W, scene["focal"], scene["cx"], scene["cy"]],
           "near": 0.3, "far": 0.9}]
    zs, za = [t(v).to(dev) for v in latents]
    embed_fn, _ = nets.get_embedder(3, 0)
    buf = training.TrainBuffers("bf16", n, dev)
    buf.signal_trainer = training.SignalTrainer(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                                ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    ys, xs = t(sel[:, 0]).to(dev), t(sel[:, 1]).to(dev)
    keep = training.lib
    training.lib = Proxy(keep, buf, mode)
    try:
        loss, lh, lc, _, _ = run_nerf.train_step_loss_hip(mods, ds, 0, 3, sel, tgt_h.to(dev)[ys, xs], tgt_c.to(dev)[ys, xs], zs,
                                                          za, step, args, scene["aud"].shape[0], embed_fn, ds[0]["poses"][0], buf)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        training.lib = keep
    wn = wd = 0.0
    kn = kd = None
    errs, vs_base = [], []
    for tag, m in mods.items():
        for k, p in m.named_parameters():
            ref = ref_g[f"{tag}/{k}"]
            rn = 0.0 if ref is None else float(ref.double().norm())
            if rn == 0.0 or p.grad is None:
                continue
            g = p.grad.detach().cpu()
            en = abs(float(g.double().norm()) - rn) / rn
            ed = float((g - ref).double().norm()) / rn
            errs.append(ed)
            if mode is None and (ed > 0.03 or os.environ.get("DIAG_ALL")):
                print(f"    {tag}/{k}: norm {float(g.double().norm()):.3e} vs {rn:.3e}  whole-tensor err {ed:.3f}")
            if mode is None:
                BASE[f"{tag}/{k}"] = g.clone()
            else:
                b = BASE[f"{tag}/{k}"]
                vs_base.append((float((g - b).double().norm() / b.double().norm()), f"{tag}/{k}"))
            if en > wn: wn, kn = en, f"{tag}/{k}"
            if ed > wd: wd, kd = ed, f"{tag}/{k}"
    print(f"{label:28s}: vs oracle: worst norm err {wn:.4f}, worst whole-tensor err {wd:.4f} ({kd}), median {np.median(errs):.4f}", flush=True)
    if vs_base:
        vs_base.sort()
        print(f"{'':28s}  vs the bf16 step's own gradient: median {vs_base[len(vs_base) // 2][0]:.4f}, worst {vs_base[-1][0]:.4f} ({vs_base[-1][1]}); "
              f"storage rounding error per array: {np.mean(QERR):.4f}", flush=True)
    QERR.clear()


E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2
run(None, "bf16 (shipping)")
if os.environ.get("DIAG_ONLY_SHIPPING"):
    sys.exit(0)
run([("dy", E4), ("act", E4)], "both e4m3, scale per row x 32 pts")
run([("dy", E4, 32), ("act", E4, 32)], "both e4m3, per 32 rows x 32 pts")
run([("dy", E4, 256), ("act", E4, 256)], "both e4m3, per 256 rows x 32 pts")
run([("dy", E5, 32), ("act", E4, 32)], "dy e5m2 / act e4m3, 32 x 32")
