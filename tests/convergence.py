"""Train-to-convergence harness of the training tiers (test infrastructure: imported by
tests/test_gpu_convergence.py and tools/convergence.py only).

north_star: "PSNR within 0.05 dB of reference" - the reference trains in fp32 (run_nerf_com_trainExpLater.py:916-931,
1250), the 16-bit training tier computes its weight gradients on MX-fp8 x MX-fp4 operands.  One step against the oracle
(tests/test_gpu_train.py) bounds the per-step gradient error; THIS harness asks the question a user asks: does a model
trained in the 16-bit tier come out as good as one trained in the exact tier?

  teacher   a fixed synthetic scene: the five networks in torch's default initialisation (seed 7) with the density and colour
            layers calibrated so that there is structure to learn (make_teacher), F frames of poses / audio / expression
            features (synth.bench_scene), rendered by the f32 tier (64 coarse samples, both fields: exactly what the training
            step differentiates, MAIN:855-899) -> uint8 ground-truth frames (head, composite), as a dataset on disk holds them;
  students  fresh networks (torch's default initialisation, another seed), trained with the PRODUCTION step
            (run_nerf.train_step_loss_hip -> training.backward -> run_nerf.optimizer_steps: device pixel sampler, uint8
            targets gathered in the forward's epilogue, HipAdam, all five optimizers live) on the first F_train frames,
            same start, same frame and pixel sequence in every tier;
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


def make_modules(dev, states=None, seed=None):
    """the five networks: from a state dict set, or freshly initialised (torch's default initialisation) under `seed`"""
    if seed is not None:
        torch.manual_seed(seed)
    mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
            "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        if states is not None:
            m.load_state_dict({kk: _t(v) for kk, v in states[k].items()})
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


STUDENT_SIGMA_BIAS = 0.05
# Learning rate of the harness: 1e-4.  scripts/train_obama.sh trains at 5e-4; on this synthetic scene the first few hundred Adam steps
# at 5e-4 decide by chance whether the head field's faint starting density survives (profiles/r05d_convergence_scan.txt: after 2,000
# steps the exact tier's head image is the bare background, 17.96 dB, and the 16-bit tier's is at 35.7 dB - and with another seed it
# is the other way round): a coin toss is no yardstick.  At 1e-4 every tier's student learns both fields.
LRATE = 1e-4
# ... decayed by the reference's own schedule (MAIN:1081-1094: lrate x 0.1^(step / (lrate_decay x 1500))) with --lrate_decay =
# steps / 3000, i.e. to 1 % over the run: at a CONSTANT rate the head field's density keeps collapsing and recovering (held-out head
# PSNR of one and the same run: 34.4 dB at step 6,000, 17.96 dB = empty at 9,000, 34.6 dB at 12,000; two exact-tier runs that differ in
# the pixel seed end 1.7 dB apart, profiles/r05f_convergence_constant_lr.txt) and the final score measures where in that cycle a run
# was stopped.  A decayed rate lets every run settle.


def student_start(dev, init_seed=1234):
    """the students' common start: torch's default initialisation under a fixed seed (not the teacher's seed), with ONE value
    set by hand: sigma_out.bias = +0.05.  The default draw puts the raw density of a fresh decoder at -0.02 +- 0.0075 over the
    whole frustum: relu(sigma) = 0 at every sample of every ray, no gradient reaches the density layer, and the field stays
    empty for good - in every tier, the exact one included (profiles/r05c_convergence_scan.txt: the head image of all students
    equals the bare background to four digits after 1,500 steps).  A small positive bias is a faint fog the loss can shape."""
    mods = make_modules(dev, seed=init_seed)
    with torch.no_grad():
        mods["decoder"].sigma_out.bias.fill_(STUDENT_SIGMA_BIAS)
    return mods


def train_student(scene, gt8, tier, steps, act_format=None, init_seed=1234, pixel_seed=100, n_rand=2048, curve_every=0,
                  log=None, lrate=LRATE, start_states=None, constant_lr=False):
    """`steps` production steps of a fresh student on the F_TRAIN training frames.  -> (modules, info).
    start_states (name -> state_dict): continue from these parameters instead (fresh Adam moments); constant_lr: no decay."""
    dev, sc = scene.dev, scene.sc
    mods = student_start(dev, init_seed) if start_states is None else make_modules(dev, states=start_states)
    decay = 10 ** 6 if constant_lr else max(1, steps // 3000)
    a = run_nerf.config_parser().parse_args(
        (f"--expname conv --concate_bg --N_rand={n_rand} --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
         "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --nosmo_iters 0 --noexp_iters 0 "
         f"--lrate {lrate} --lrate_decay {decay}").split())
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


def raw_outputs(scene, mods, frame=0, n_rays=2048, seed=5):
    """raw decoder outputs (sigma_head, rgb_head[3], sigma_torso, rgb_torso[3]) [n_rays * 64, 8] of `mods` on random pixels of one
    training frame: the exact tier's training forward (dfn_train_fwd), whose recorder leaves them per sample"""
    import ctypes as C
    from dfanerf._lib import check as chk, lib
    from dfanerf.engine import _ptr, _stream
    dev, sc = scene.dev, scene.sc
    pk = engine.PackedDecoder(engine.flatten_state(mods["decoder"].state_dict(), dev), "f32", fields=(0, 1))
    enc = engine.SignalEncoder(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                               scene.aud[:F_TRAIN].contiguous(), scene.exp[:F_TRAIN].contiguous(), scene.poses[:F_TRAIN].contiguous())
    s2, t2 = enc.encode([frame], 4, 8)
    bias = pk.fold(s2[0], t2[0], scene.zs[0], scene.za[0])
    rows = [chk(lib.dfn_train_rows(f, 0), "rows") for f in (0, 1)]
    mrows = [chk(lib.dfn_train_rows(f, 2), "rows") for f in (0, 1)]
    NP = n_rays * 64
    act = [torch.empty(NP // 32, rows[f], 32, dtype=torch.float32, device=dev) for f in (0, 1)]
    masks = [torch.empty(NP // 32, mrows[f], 64, dtype=torch.int32, device=dev) for f in (0, 1)]
    samples = torch.empty(NP, 8, dtype=torch.float32, device=dev)
    rgb = torch.empty(2, n_rays, 3, dtype=torch.float32, device=dev)
    pix = torch.randperm(scene.H * scene.W, generator=torch.Generator().manual_seed(seed))[:n_rays].to(torch.int32).to(dev)
    fr = engine.make_frame(scene.H, scene.W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][frame], sc["pose_body"], sc["near"],
                           sc["far"], ray_count=n_rays, n_fine=0, fields=2)
    nh = pk.bias_floats(0)
    chk(lib.dfn_train_fwd(0, C.byref(fr), _ptr(pk.packed[0]), _ptr(pk.packed[1]), _ptr(bias), C.c_void_p(bias.data_ptr() + 4 * nh),
                          None, _ptr(scene.bg8), _ptr(pix), _ptr(rgb[0]), _ptr(rgb[1]), _ptr(samples), _ptr(act[0]), _ptr(masks[0]),
                          _ptr(act[1]), _ptr(masks[1]), _stream()), "dfn_train_fwd(raw outputs)")
    torch.cuda.synchronize()
    return samples


TEACHER_SIGMA_STD, TEACHER_SIGMA_MEAN, TEACHER_LOGIT_STD, TEACHER_PE_DAMP = 6.0, -1.0, 1.5, 0.75


def make_teacher(scene, seed=7):
    """The teacher: the five networks in torch's default initialisation (seed 7) with the two output layers re-scaled so that the
    scene is worth learning AND learnable - a freshly initialised decoder renders a uniform grey fog.  Calibrated on 2048 rays of
    training frame 0: sigma_out's gain and bias such that the raw density over the sampled points has mean -1 and standard
    deviation 6 (about 40 % of space carries density, up to ~15: translucent to opaque structures, smooth in space; head and
    torso fields pooled: they share the layer), feat_out's gain such that the colour logits have standard deviation 1.5
    (saturated and pale regions).  synth.synth_all_states(0) - the parity tests' network - is NOT used here: its density layer
    (weights x 1.7, bias -27) is so steep that 1,500 Adam steps at the reference's learning rate push a student that STARTS
    25.9 dB from the ground truth to an empty head field (loss 0.011 -> 0.105, in every tier: profiles/r05b_convergence_scan.txt)."""
    mods = make_modules(scene.dev, seed=seed)
    dec = mods["decoder"]
    with torch.no_grad():
        # a SMOOTH scene: the weights that read octave i of the positional encoding (columns 6 i .. 6 i + 5 of every layer fed
        # with PE(p), decoder.py:257-275) are damped by 2^(-0.75 i) - with the default initialisation all ten octaves weigh the
        # same and the teacher's colour is noise at the pixel scale, which no student fits (the composite of the first long runs
        # stayed at 19 dB in every tier, profiles/r05g_convergence.txt)
        damp = torch.tensor([2.0 ** (-TEACHER_PE_DAMP * (c // 6)) for c in range(60)], device=scene.dev)
        for lin in (dec.fc_in, dec.fc_p_skips[0], dec.fc_in_torso, dec.fc_p_skips_torso[0], dec.deform_net.blocks_embed[0],
                    dec.deform_net.blocks_signal[0], dec.deform_net.fc_embed_skips[0]):
            lin.weight[:, :60].mul_(damp)
        s = raw_outputs(scene, mods)
        sig = torch.cat([s[:, 0], s[:, 4]]).double()
        b_old = float(dec.sigma_out.bias[0])
        mu, sd = float((sig - b_old).mean()), float((sig - b_old).std())
        g = TEACHER_SIGMA_STD / max(sd, 1e-6)
        dec.sigma_out.weight.mul_(g)
        dec.sigma_out.bias.fill_(TEACHER_SIGMA_MEAN - g * mu)
        y = torch.cat([s[:, 1:4], s[:, 5:8]]).double().clamp(1e-6, 1 - 1e-6)
        logit = torch.log(y / (1 - y))
        gc = TEACHER_LOGIT_STD / max(float((logit - logit.mean(0, keepdim=True)).std()), 1e-6)
        dec.feat_out.weight.mul_(gc)
    info = {"sigma_gain": g, "sigma_bias": float(dec.sigma_out.bias[0]), "rgb_gain": gc, "raw_sigma_mean_before": mu + b_old,
            "raw_sigma_std_before": sd}
    return mods, info


def teacher_ground_truth(scene, teacher):
    """uint8 (head, com) frames [H*W,3] of the teacher for every frame (both splits), rendered in the exact tier"""
    imgs = scene.render(teacher, "f32", "train") + scene.render(teacher, "f32", "held")
    return [(to8b(rh), to8b(rc)) for rh, rc in imgs]


def run(steps, variants, size=450, curve_every=0, log=None, with_inference_check=True, lrate=LRATE):
    """variants: list of (name, tier, act_format, pixel_seed).  -> dict of per-variant scores and the pairwise differences"""
    dev = torch.device("cuda")
    scene = Scene(dev, size)
    teacher, tinfo = make_teacher(scene)
    gt8 = teacher_ground_truth(scene, teacher)
    res = {"steps": steps, "size": size, "frames_train": F_TRAIN, "frames_held_out": F_HELD, "lrate": lrate, "teacher": tinfo,
           "variants": {}}
    # how far the teacher's frames are from the bare background (what there is to learn), and what the students' start scores
    a, b = scene.split("held")
    bgf = scene.bg8.double() / 255.0
    res["teacher"]["held_out_psnr_of_the_bare_background"] = {
        "head": float(10 * np.log10(1.0 / np.mean([float(((gt8[k][0].double() / 255.0 - bgf) ** 2).mean()) for k in range(a, b)]))),
        "com": float(10 * np.log10(1.0 / np.mean([float(((gt8[k][1].double() / 255.0 - bgf) ** 2).mean()) for k in range(a, b)])))}
    res["untrained"] = score(scene, student_start(dev, 1234), gt8, "held")
    if log:
        log(f"  teacher: {tinfo}; students' start scores {res['untrained']} on the held-out frames")
    keep = {}
    for name, tier, fmt, pseed in variants:
        mods, info = train_student(scene, gt8, tier, steps, act_format=fmt, pixel_seed=pseed, curve_every=curve_every, log=log,
                                   lrate=lrate)
        info["psnr_held_out"] = score(scene, mods, gt8, "held")
        info["psnr_train_frames"] = score(scene, mods, gt8, "train")
        # the same held-out frames rendered in the ARITHMETIC THE 16-BIT TIER TRAINS IN (bf16 operands): a model is fitted to what
        # its own forward renders, so scoring it in another arithmetic adds that arithmetic's distance (bf16 vs f32: 47 dB on a full
        # frame, LABNOTES.md 3) to its error - 10 log10(1 + 10^((P - 47) / 10)) dB at a P-dB model, 0.3 dB at 35.6 dB
        info["psnr_held_out_bf16_render"] = score(scene, mods, gt8, "held", tier="bf16")
        res["variants"][name] = info
        keep[name] = mods
        if log:
            log(f"  {name}: {info['ms_per_step']:.3f} ms/step, loss {info['first_loss']:.5f} -> {info['last_loss']:.5f}, held-out "
                f"head {info['psnr_held_out']['head']:.3f} dB com {info['psnr_held_out']['com']:.3f} dB (rendered in bf16: head "
                f"{info['psnr_held_out_bf16_render']['head']:.3f} com {info['psnr_held_out_bf16_render']['com']:.3f}), training frames head "
                f"{info['psnr_train_frames']['head']:.3f} com {info['psnr_train_frames']['com']:.3f}")
    if with_inference_check:
        # the models the 16-bit tier trained, through the f16 INFERENCE tier: the full-frame accuracy clause (>= 49.4 dB against
        # the exact tier, LABNOTES.md 3) on TRAINED weights - configs[1] (head, 64 + 128) and configs[2] (two fields)
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
            # ... and what the f16 tier's ACCURACY GUARD (dfanerf/f16guard.py, round 6) says about these trained weights: every held-out
            # frame in f16 and f32 (64 + 128 samples, both fields), the model's own PSNR against the scene's ground truth measured
            # on the same frames -> the gate the 0.05-dB clause implies for THIS model, and the margin to it
            from dfanerf import f16guard
            lo = scene.render(mods, "f16", "held", n_fine=128, fields=2)
            hi = scene.render(mods, "f32", "held", n_fine=128, fields=2)
            a0, _ = scene.split("held")
            blocks = [{"head": (lo[k][0], hi[k][0], gt8[a0 + k][0].float() / 255.0), "com": (lo[k][1], hi[k][1], gt8[a0 + k][1].float() / 255.0)}
                      for k in range(len(lo))]
            st = f16guard.accuracy_stats(blocks)
            for im, q in st.items():
                q["gate_db"] = f16guard.psnr_gate(q["model_psnr_db"])
                q["margin_db"] = q["psnr_db"] - q["gate_db"]
                q["worst_frame_margin_db"] = q["worst_block_db"] - (q["gate_db"] - f16guard.BLOCK_SLACK_DB)
            try:
                f16guard.check_accuracy(st)
                verdict = "accepted"
            except f16guard.F16AccuracyError as e:
                verdict = "refused: " + str(e)[:200]
            res["variants"][name]["f16_accuracy_guard"] = {"verdict": verdict, **st}
    return res


def run_continuation(base_steps, cont_steps, variants, cont_lrate=1e-5, size=450, log=None):
    """The PAIRED form of the comparison: ONE student trained to convergence in the exact tier (base_steps), then continued for
    cont_steps at a small constant rate from those very parameters (fresh Adam moments) by every variant (name, tier, act_format,
    pixel_seed).  All continuations sit in the same basin, so what separates two fresh runs (which basin, where on the way: +-0.5 dB
    between two exact-tier seeds, LABNOTES.md 9.2) is gone and a format that biased the weight gradients would show as a drift of its
    continuation away from the exact tier's.  -> base scores + per-variant scores"""
    dev = torch.device("cuda")
    scene = Scene(dev, size)
    teacher, tinfo = make_teacher(scene)
    gt8 = teacher_ground_truth(scene, teacher)
    base, binfo = train_student(scene, gt8, "f32", base_steps, log=log)
    states = {k: {kk: v.detach().cpu().clone() for kk, v in m.state_dict().items()} for k, m in base.items()}
    res = {"base_steps": base_steps, "cont_steps": cont_steps, "cont_lrate": cont_lrate,
           "base": {"psnr_held_out": score(scene, base, gt8, "held"), "psnr_train_frames": score(scene, base, gt8, "train"),
                    "last_loss": binfo["last_loss"]}, "variants": {}}
    if log:
        log(f"  base (f32, {base_steps} steps): {res['base']}")
    for name, tier, fmt, pseed in variants:
        mods, info = train_student(scene, gt8, tier, cont_steps, act_format=fmt, pixel_seed=pseed, lrate=cont_lrate,
                                   start_states=states, constant_lr=True)
        info["psnr_held_out"] = score(scene, mods, gt8, "held")
        info["psnr_train_frames"] = score(scene, mods, gt8, "train")
        res["variants"][name] = info
        if log:
            log(f"  {name}: held-out {info['psnr_held_out']}, training frames {info['psnr_train_frames']}, loss -> {info['last_loss']:.6f}")
    return res
