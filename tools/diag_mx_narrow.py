#!/usr/bin/env python3
"""Developer experiment (GPU box): what would NARROWER storage of the recorded arrays (MX-fp6 e2m3 / e3m2, MX-fp4 e2m1 - the other
operand formats of v_mfma_scale_f32_32x32x64_f8f6f4, at twice the fp8 rate and 3/4 resp. 1/2 of its bytes) cost in gradient
accuracy?  VERDICT r3 next #1(b): emulate first, build only what holds the gates.

The shipping 16-bit training step (2048 rays, full-size test's setting) with the MX-fp8 arrays re-rounded IN PLACE between the dX
chain and their consumers (dfn_signal_grad, dfn_weight_bias_grad): every 32-row x 32-point block is dequantised, rounded to the
narrow format's grid under a fresh power-of-two block scale (no saturation: scale = 2^ceil(log2(amax / format max))), and written
back as e4m3 bytes under the block's ORIGINAL scale byte - exact, because a narrow-format value times a power of two is an e4m3
value (<= 3 mantissa bits, inside its range).  The kernels then run unchanged on data that carries exactly the narrow format's
information.  Gradients against torch CPU autograd through the oracle and against the shipping step's own gradients."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "dfa-nerf_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
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
training._OVERLAP = False          # one stream: the re-rounding below sits between the dX chain and its consumers

# format: (largest value, exponent of the smallest normal binade, mantissa bits)
FORMATS = {"e2m3": (7.5, 0, 3), "e3m2": (28.0, -2, 2), "e2m1": (6.0, 0, 1), "e4m3": None}
QERR = {}
SCALE_BYTES = 128


def grid_round(q, fmt):
    """q (|q| <= format max) -> nearest value of the format's grid (round half to even)"""
    fmax, emin, mbits = FORMATS[fmt]
    a = q.abs()
    e = torch.floor(torch.log2(a.clamp_min(2.0 ** (emin - 20)))).clamp_min(float(emin))     # binade (subnormals share emin's step)
    step = torch.exp2(e - mbits)
    return (torch.round(a / step) * step).clamp_max(fmax) * torch.sign(q)


def requant(arr, rows, fmt, rb):
    """arr uint8 [tiles, rows * 32 + 128] (dfn_mlp.h "MX-fp8 recording") -> the same array carrying only `fmt` information, the block
    scale taken over rb consecutive rows (32 = one stored block, 64 = the tile pair the producers share a scale over)"""
    if fmt == "e4m3":
        return
    nb = rows // 32
    T_ = arr.shape[0]
    CH = 1024                                                 # tiles per chunk (memory)
    num = den = 0.0
    for c0 in range(0, T_, CH):
        a = arr[c0:c0 + CH]
        data = a[:, :rows * 32].reshape(-1, nb, 1024)
        sc = torch.exp2(a[:, rows * 32:rows * 32 + nb].float() - 127.0).unsqueeze(-1)            # [t, nb, 1]
        v8 = data.view(torch.float8_e4m3fn).float()
        v = v8 * sc
        g = rb // 32
        if g > 1 and nb % g == 0:
            amax = v.abs().reshape(-1, nb // g, g * 1024).amax(-1, keepdim=True).repeat_interleave(g, 1)
        else:
            amax = v.abs().amax(-1, keepdim=True)
        fmax = FORMATS[fmt][0]
        s2 = torch.exp2(torch.ceil(torch.log2((amax / fmax).clamp_min(1e-38))))
        q = grid_round(v / s2, fmt) * s2
        num += float(((q - v) ** 2).sum()); den += float((v ** 2).sum())
        back = (q / sc).to(torch.float8_e4m3fn)
        assert torch.equal(back.float() * sc, q), "not exactly representable under the original scale"
        data.copy_(back.view(torch.uint8))
    QERR.setdefault(fmt, []).append((num / max(den, 1e-300)) ** 0.5)


class Proxy:
    def __init__(self, lib, buf, mode):
        self._lib, self._buf, self._mode, self._done = lib, buf, mode, set()

    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if not self._mode or k not in ("dfn_signal_grad", "dfn_weight_bias_grad"):
            return f

        def wrapped(tier, field, *a):
            b = self._buf
            from dfanerf._lib import lib as L
            for what, fmt, rb in self._mode:
                if (what, field) in self._done or (what == "act" and k == "dfn_signal_grad"):
                    continue
                self._done.add((what, field))
                arr = b.dy[field] if what == "dy" else b.act[field]
                rows = (arr.shape[1] - SCALE_BYTES) // 32
                requant(arr, rows, fmt, rb)
            return f(tier, field, *a)
        return wrapped


BASE = {}


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
    kd = kn = None
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
            if mode is None:
                BASE[f"{tag}/{k}"] = g.clone()
            else:
                b = BASE[f"{tag}/{k}"]
                vs_base.append((float((g - b).double().norm() / b.double().norm()), f"{tag}/{k}"))
            if en > wn: wn, kn = en, f"{tag}/{k}"
            if ed > wd: wd, kd = ed, f"{tag}/{k}"
    print(f"{label:44s}: vs oracle: worst norm err {wn:.4f} ({kn}), worst whole-tensor err {wd:.4f} ({kd}), median {np.median(errs):.4f}"
          f"   [gates: 0.06 / 0.15]", flush=True)
    if vs_base:
        vs_base.sort()
        q = "; ".join(f"{k} storage rounding {np.mean(v):.4f}" for k, v in QERR.items())
        print(f"{'':44s}  vs the shipping step's own gradient: median {vs_base[len(vs_base) // 2][0]:.4f}, worst {vs_base[-1][0]:.4f} "
              f"({vs_base[-1][1]}); {q}", flush=True)
    QERR.clear()


run(None, "MX-fp8 e4m3 (shipping)")
for label, mode in (
        ("act e2m3 / dy e4m3, scale per 32 rows", [("act", "e2m3", 32)]),
        ("act e2m3 / dy e4m3, scale per 64 rows", [("act", "e2m3", 64)]),
        ("act e4m3 / dy e2m3, per 64", [("dy", "e2m3", 64)]),
        ("act e4m3 / dy e3m2, per 64", [("dy", "e3m2", 64)]),
        ("both e2m3, per 64", [("act", "e2m3", 64), ("dy", "e2m3", 64)]),
        ("act e2m3 / dy e3m2, per 64", [("act", "e2m3", 64), ("dy", "e3m2", 64)]),
        ("act e2m1 / dy e4m3, per 64", [("act", "e2m1", 64)]),
        ("act e2m1 / dy e4m3, per 32", [("act", "e2m1", 32)]),
        ("act e4m3 / dy e2m1, per 64", [("dy", "e2m1", 64)]),
        ("act e2m1 / dy e2m3, per 64", [("act", "e2m1", 64), ("dy", "e2m3", 64)]),
        ("both e2m1, per 64", [("act", "e2m1", 64), ("dy", "e2m1", 64)]),
        ("both e2m1, per 32", [("act", "e2m1", 32), ("dy", "e2m1", 32)])):
    if os.environ.get("DIAG_ONLY") and os.environ["DIAG_ONLY"] not in label:
        continue
    run(mode, label)
