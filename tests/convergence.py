"""Train-to-convergence harness of the training tiers (test infrastructure: imported by
tests/test_gpu_convergence.py and tools/convergence.py only).

north_star: "PSNR within 0.05 dB of reference" - the reference trains in fp32 (run_nerf_com_trainExpLater.py:916-931,
1250), the 16-bit training tier computes its weight gradients on MX-fp8 x MX-fp4 operands.  One step against the oracle
(tests/test_gpu_train.py) bounds the per-step gradient error; THIS harness asks the question a user asks: does a model
trained in the 16-bit tier come out as good as one trained in the exact tier?

  teacher   a fixed synthetic scene: the five networks of synth.synth_all_states(0), F frames of poses / audio /
            expression features (synth.bench_scene), rendered by the f32 tier (64 coarse samples, both fields: exactly
            what the training step differentiates, MAIN:855-899) -> uint8 ground-truth frames (head, composite), as a
            dataset on disk would hold them;
  students  the teacher's networks with EVERY tensor of all five moved by `perturb` x rms(tensor) x N(0, 1) (a fixed seed): the
            rendered frames start ~15-20 dB from the ground truth and training has to find the way back - all five networks
            receive gradients, and the end of the run is where a quantised gradient matters most: close to an optimum, where
            the true gradient is small against the rounding of its operands.  (A student initialised from scratch - torch's
            default initialisation - is the other possible start; on THIS synthetic scene its head field's density dies in
            the first steps - relu(sigma) = 0 on every ray, in every tier, the exact one included: the head image stays the
            background, 10.94 dB, for 12,000 steps - so it measures nothing about the tiers: profiles/r05a_convergence_fresh_init.txt.)
            Trained with the PRODUCTION step (run_nerf.train_step_loss_hip -> training.backward -> run_nerf.optimizer_steps:
            device pixel sampler, uint8 targets gathered in the forward's epilogue, HipAdam, all five optimizers live) on the
            first F_train frames, same start, same frame and pixel sequence in every tier;
  score     PSNR of the student's f32-tier renders against the ground truth, on the training frames and on the HELD-OUT
            frames (poses and audio the student never saw), head and composite images.
Training trajectories are chaotic: two runs that differ in rounding only end at slightly different models.  The harness
therefore also trains the exact tier a second time with ANOTHER pixel-sampling seed: the spread between the two f32 runs
is the noise floor a tier difference has to be read against.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))

from dfanerf import engine, frames, nets, run_nerf, synth, training        # noqa: E402
from dfanerf.decoder import Decoder                                         # noqa: E402

F_TRAIN, F_HELD = 8, 4


def _t(x):
    return torch.from_numpy(np.asarray(x))


def make_modules(dev, states=None, seed=None, perturb=None):
    """the five networks: from a state dict set (the teacher), freshly initialised under `seed` (torch's default
    initialisation), or - perturb = a - the state dicts with every tensor moved by a x rms(tensor) x N(0, 1) under `seed`"""
    if seed is not None:
        torch.manual_seed(seed)
    mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
            "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    gen = torch.Generator().manual_seed(0 if seed is None else seed)
    for k, m in mods.items():
        if states is not None:
            sd = {kk: _t(v).clone() for kk, v in states[k].items()}
            if perturb:
                for kk, v in sd.items():
                    if v.dtype.is_floating_point:
                        rms = float(v.double().pow(2).mean().sqrt())
                        sd[kk] = v + perturb * rms * torch.randn(v.shape, generator=gen)
            m.load_state_dict(sd)
        m.to(dev)
    return mods


class Scene:
    """Geometry + features of F_TRAIN + F_HELD frames, latent codes, background; everything resident on the device."""

    def __init__(self, dev, size=450):
        F = F_TRAIN + F_HELD
        sc = synth.bench_scene(0, n_frames=F, H=size, W=size)
        sc["focal"] = 1200.0 * size / 450
        self.sc, self.dev, self.F, self.H, self.W = sc, dev, F, size, size
        self.zs, self.za = [_t(v).to(dev) for v in synth.synth_latents(0)]
        self.bg8 = _t(sc["bg"]).reshape(-1, 3).to(dev)
        self.aud, self.exp, self.poses = [_t(sc[k]).to(dev) for k in ("aud", "exp", "poses")]

    def split(self, name):
        """frame range of a split: the training sequence and the held-out sequence are two datasets of their own (like the
        reference's transforms_train / transforms_val files): a smoothing window never reaches across them"""
        return (0, F_TRAIN) if name == "train" else (F_TRAIN, self.F)

    def render(self, mods, tier, split, n_fine=0, fields=2, only=None):
        """-> [(rgb_head [H*W,3], rgb_com or None)] f32 for the frames of `split` (only = indices within it): the networks `mods`
        through the inference path in `tier` - HIP signal encoders over the split's own sequence (window padding at its ends
        as MAIN:36-57), fold, fused render."""
        sc, dev = self.sc, self.dev
        a, b = self.split(split)
        flat = engine.flatten_state(mods["decoder"].state_dict(), dev)
        pk = engine.PackedDecoder(flat, tier, fields=(0, 1) if fields == 2 else (0,))
        enc = engine.SignalEncoder(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                   self.aud[a:b].contiguous(), self.exp[a:b].contiguous(), self.poses[a:b].contiguous())
        out = []
        for k in (range(b - a) if only is None else only):
            s2, t2 = enc.encode([int(k)], 4, 8)
            bias = pk.fold(s2[0], t2[0] if fields == 2 else None, self.zs[0], self.za[0])
            fr = engine.make_frame(self.H, self.W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][a + k], sc["pose_body"], sc["near"],
                                   sc["far"], n_fine=n_fine, fields=fields)
            r = engine.render(pk, bias, fr, self.bg8)
            out.append((r[0].clone(), r[1].clone() if fields == 2 else None))
        torch.cuda.synchronize()
        return out


def to8b(x):
    """HELP:17 on the device: (255 * clip(x, 0, 1)) truncated"""
    return (255.0 * x.clamp(0.0, 1.0)).to(torch.uint8)


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10.0 * np.log10(1.0 / mse) if mse > 0 else float("inf")


def score(scene, mods, gt8, split, tier="f32"):
    """PSNR of the student's renders (tier f32: the exact arithmetic) against the uint8 ground truth / 255 over the frames of
    `split`: {head, com} = 10 log10(1 / mean squared error over all pixels of all its frames)"""
    imgs = scene.render(mods, tier, split)
    a, _ = scene.split(split)
    se = {"head": 0.0, "com": 0.0}
    for k, (rh, rc) in enumerate(imgs):
        se["head"] += float(((rh.double() - gt8[a + k][0].double() / 255.0) ** 2).mean())
        se["com"] += float(((rc.double() - gt8[a + k][1].double() / 255.0) ** 2).mean())
    return {k: float(10.0 * np.log10(len(imgs) / v)) for k, v in se.items()}


PERTURB = 0.2


def student_start(dev, init_seed=1234, perturb=PERTURB):
    """the students' common start: the perturbed teacher (perturb > 0) or torch's default initialisation (perturb = None)"""
    if perturb:
        return make_modules(dev, states=synth.synth_all_states(0), seed=init_seed, perturb=perturb)
    return make_modules(dev, seed=init_seed)


def train_student(scene, gt8, tier, steps, act_format=None, init_seed=1234, pixel_seed=100, n_rand=2048, curve_every=0,
                  log=None, perturb=PERTURB):
    """`steps` production steps of a student on the F_TRAIN training frames.  -> (modules, info)"""
    dev, sc = scene.dev, scene.sc
    mods = student_start(dev, init_seed, perturb)
    a = run_nerf.config_parser().parse_args(
        (f"--expname conv --concate_bg --N_rand={n_rand} --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
         "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --nosmo_iters 0 --noexp_iters 0 "
         "--lrate 5e-4 --lrate_decay 500").split())
    ds = [{"auds": scene.aud[:F_TRAIN].contiguous(), "exp": scene.exp[:F_TRAIN].contiguous(),
           "poses": scene.poses[:F_TRAIN].contiguous(), "bc_img": (scene.bg8.float() / 255.0),
           "hwfcxy": [scene.H, scene.W, sc["focal"], sc["cx"], sc["cy"]], "near": sc["near"], "far": sc["far"]}]
    embed_fn, _ = nets.get_embedder(3, 0)
    opts = {k: run_nerf.make_adam(m.parameters(), a.lrate) for k, m in mods.items()}
    buf = training.TrainBuffers(tier, n_rand, dev, act_format=act_format)
    buf.signal_trainer = training.SignalTrainer(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                                ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    buf.signal_trainer.adopt_optimizers(opts)
    sampler = frames.PixelSampler(scene.H, scene.W, n_rand, 0, dev, seed=pixel_seed, pipeline=True,
                                  stream=run_nerf.draw_stream(buf))
    rng = np.random.RandomState(pixel_seed)
    pose_torso = ds[0]["poses"][0]
    curve, losses = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(steps):
        img_i = int(rng.randint(0, F_TRAIN))
        loss, *_ = run_nerf.train_step_loss_hip(mods, ds, 0, img_i, sampler.draw(), gt8[img_i][0], gt8[img_i][1], scene.zs,
                                                scene.za, step, a, F_TRAIN, embed_fn, pose_torso, buf)
        for o in opts.values():
            o.zero_grad()
        training.backward(loss, buf)
        run_nerf.optimizer_steps(opts, step, a)
        run_nerf.update_lrate(opts, step, a)
        if step % 50 == 0:
            losses.append(loss.detach())
        if curve_every and (step + 1) % curve_every == 0 and step + 1 < steps:
            buf.signal_trainer.join()
            torch.cuda.synchronize()
            t_pause = time.perf_counter()
            curve.append((step + 1, score(scene, mods, gt8, "held")))
            if log:
                log(f"    {tier}{'/' + act_format if act_format else ''} step {step + 1}: held-out PSNR {curve[-1][1]}")
            t0 += time.perf_counter() - t_pause
    buf.signal_trainer.join()
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    ls = [float(x) for x in losses]
    info = {"tier": tier, "act_format": ("e2m1" if buf.act_format == 1 else "e4m3") if tier == "bf16" else "f32", "steps": steps,
            "seconds": secs, "ms_per_step": secs / max(steps, 1) * 1e3, "first_loss": ls[0] if ls else None,
            "last_loss": float(np.mean(ls[-4:])) if ls else None, "finite": bool(np.all(np.isfinite(ls))), "curve": curve}
    return mods, info


def teacher_ground_truth(scene):
    """uint8 (head, com) frames [H*W,3] of the teacher for every frame (both splits), rendered in the exact tier"""
    teacher = make_modules(scene.dev, states=synth.synth_all_states(0))
    imgs = scene.render(teacher, "f32", "train") + scene.render(teacher, "f32", "held")
    return [(to8b(rh), to8b(rc)) for rh, rc in imgs]


def run(steps, variants, size=450, curve_every=0, log=None, with_inference_check=True, perturb=PERTURB):
    """variants: list of (name, tier, act_format, pixel_seed).  -> dict of per-variant scores and the pairwise differences"""
    dev = torch.device("cuda")
    scene = Scene(dev, size)
    gt8 = teacher_ground_truth(scene)
    res = {"steps": steps, "size": size, "frames_train": F_TRAIN, "frames_held_out": F_HELD, "perturb": perturb, "variants": {}}
    # what the students' start scores (the scale of what training buys)
    res["untrained"] = score(scene, student_start(dev, 1234, perturb), gt8, "held")
    keep = {}
    for name, tier, fmt, pseed in variants:
        mods, info = train_student(scene, gt8, tier, steps, act_format=fmt, pixel_seed=pseed, curve_every=curve_every, log=log,
                                   perturb=perturb)
        info["psnr_held_out"] = score(scene, mods, gt8, "held")
        info["psnr_train_frames"] = score(scene, mods, gt8, "train")
        res["variants"][name] = info
        keep[name] = mods
        if log:
            log(f"  {name}: {info['ms_per_step']:.3f} ms/step, loss {info['first_loss']:.5f} -> {info['last_loss']:.5f}, held-out "
                f"head {info['psnr_held_out']['head']:.3f} dB com {info['psnr_held_out']['com']:.3f} dB, training frames head "
                f"{info['psnr_train_frames']['head']:.3f} com {info['psnr_train_frames']['com']:.3f}")
    if with_inference_check:
        # the models the 16-bit tier trained, through the f16 INFERENCE tier: the full-frame accuracy clause (>= 49.4 dB against
        # the exact tier) on TRAINED weights - configs[1] (head, 64 + 128) and configs[2] (two fields)
        for name, mods in keep.items():
            if res["variants"][name]["tier"] != "bf16":
                continue
            chk = {}
            for tag, n_fine, fields in (("c2", 128, 1), ("c3", 128, 2), ("coarse", 0, 2)):
                ref = scene.render(mods, "f32", "held", n_fine=n_fine, fields=fields, only=[0])[0]
                got = scene.render(mods, "f16", "held", n_fine=n_fine, fields=fields, only=[0])[0]
                k = 1 if fields == 2 else 0
                chk[tag] = {"psnr_db": psnr(got[k], ref[k]),
                            "worst_block_db": min(psnr(got[k][i:i + 2500], ref[k][i:i + 2500])
                                                  for i in range(0, got[k].shape[0], 2500)),
                            "finite": bool(torch.isfinite(got[k]).all())}
            res["variants"][name]["f16_inference_vs_f32"] = chk
    return res
