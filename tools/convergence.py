#!/usr/bin/env python3
"""Developer tool (GPU box): the long form of tests/test_gpu_convergence.py - train fresh students to convergence on the
synthetic teacher scene in the exact tier and in the 16-bit training tier (both recorded-activation formats), same start, same
frame / pixel sequence; score them on held-out frames in the exact tier; render the 16-bit-trained students in the f16 inference
tier.  Harness: tests/convergence.py.

    python tools/convergence.py [steps=12000] [curve_every=2000] > profiles/r05_convergence.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import convergence as CV      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
# third argument: learning rate (default convergence.LRATE = 1e-4: see there) or "scan" = a quick look at the scene at two rates
mode = sys.argv[3] if len(sys.argv) > 3 else str(CV.LRATE)
log = lambda s: print(s, flush=True)
if mode == "scan":
    for lr in (5e-4, 1e-4):
        r = CV.run(steps, [("f32", "f32", None, 100), ("bf16_fp4", "bf16", "fp4", 100)], curve_every=every, log=log, with_inference_check=False, lrate=lr)
        for k, i in r["variants"].items():
            log(f"lr {lr} {k}: bare background {r['teacher']['held_out_psnr_of_the_bare_background']}, start {r['untrained']} -> after {steps} steps "
                f"held-out head {i['psnr_held_out']['head']:.2f} com {i['psnr_held_out']['com']:.2f} dB (loss {i['first_loss']:.5f} -> {i['last_loss']:.5f})")
    sys.exit(0)
lrate = float(mode)
log(f"# convergence of the training tiers: {steps} production steps of 2048 rays (lr {lrate}) on {CV.F_TRAIN} training frames (450 x 450), scored on "
    f"{CV.F_HELD} held-out frames; teacher = convergence.make_teacher (calibrated default-init networks) rendered in the f32 tier")
variants = [("f32", "f32", None, 100), ("f32_other_pixels", "f32", None, 101), ("bf16_fp4", "bf16", "fp4", 100),
            ("bf16_e4m3", "bf16", "e4m3", 100)]
# the 16-bit tier is cheap (12 s per run): more pixel seeds of it, for a spread of its own
variants += [("bf16_fp4_s101", "bf16", "fp4", 101), ("bf16_fp4_s102", "bf16", "fp4", 102), ("bf16_e4m3_s101", "bf16", "e4m3", 101),
             ("bf16_e4m3_s102", "bf16", "e4m3", 102)]
res = CV.run(steps, variants, curve_every=every, log=log, lrate=lrate)
log(f"the bare background against the teacher's frames (held-out): {res['teacher']['held_out_psnr_of_the_bare_background']}")
v = res["variants"]
log("")
log(f"the students' start (held-out): head {res['untrained']['head']:.3f} dB, com {res['untrained']['com']:.3f} dB")
log(f"{'variant':<18}{'ms/step':>9}{'held head':>11}{'held com':>10}{'train head':>12}{'train com':>11}{'held head, bf16 render':>24}{'com':>8}")
for k, i in v.items():
    log(f"{k:<18}{i['ms_per_step']:>9.3f}{i['psnr_held_out']['head']:>11.3f}{i['psnr_held_out']['com']:>10.3f}"
        f"{i['psnr_train_frames']['head']:>12.3f}{i['psnr_train_frames']['com']:>11.3f}"
        f"{i['psnr_held_out_bf16_render']['head']:>24.3f}{i['psnr_held_out_bf16_render']['com']:>8.3f}")
d = lambda a, b, s, im: v[a][s][im] - v[b][s][im]
log("")
log("differences against the exact tier (dB; + = better than f32):")
for k in [k for k in v if k != "f32"]:
    log(f"  {k:<18} held-out head {d(k, 'f32', 'psnr_held_out', 'head'):+.3f} com {d(k, 'f32', 'psnr_held_out', 'com'):+.3f}   "
        f"training frames head {d(k, 'f32', 'psnr_train_frames', 'head'):+.3f} com {d(k, 'f32', 'psnr_train_frames', 'com'):+.3f}")
log("  (f32_other_pixels = the exact tier again with another pixel-sampling seed: the noise floor of a training trajectory)")
import numpy as np
for grp, keys in (("f32 (2 seeds)", ["f32", "f32_other_pixels"]), ("bf16 / fp4 (3 seeds)", ["bf16_fp4", "bf16_fp4_s101", "bf16_fp4_s102"]),
                  ("bf16 / e4m3 (3 seeds)", ["bf16_e4m3", "bf16_e4m3_s101", "bf16_e4m3_s102"])):
    for im in ("head", "com"):
        x = [v[k]["psnr_held_out"][im] for k in keys]
        log(f"  {grp:<22} held-out {im:<4}: mean {np.mean(x):.3f} dB, min {min(x):.3f}, max {max(x):.3f}")
log("")
for k in ("bf16_fp4", "bf16_e4m3"):
    log(f"{k} rendered in the f16 inference tier vs the f32 tier (held-out frame 0, full frame): " +
        ", ".join(f"{t} {c['psnr_db']:.2f} dB (worst 2500-ray block {c['worst_block_db']:.2f})" for t, c in v[k]["f16_inference_vs_f32"].items()))
log("")
log("JSON " + json.dumps(res))
