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
if mode == "continue":
    # the paired form (convergence.run_continuation): python tools/convergence.py 12000 0 continue [cont_steps=3000] [lr=1e-5] [pixel seeds=2]
    cont = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
    clr = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-5
    log(f"# paired continuation: one exact-tier student trained {steps} steps, then continued {cont} steps at a constant {clr} from the "
        f"same parameters by every variant; held-out / training-frame PSNR in the exact tier")
    n_seeds = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    seeds = [200 + k for k in range(n_seeds)]
    variants = []
    for sd in seeds:
        variants += [(f"f32_s{sd}", "f32", None, sd), (f"bf16_fp4_s{sd}", "bf16", "fp4", sd), (f"bf16_e4m3_s{sd}", "bf16", "e4m3", sd)]
    r = CV.run_continuation(steps, cont, variants, cont_lrate=clr, log=log)
    b, v = r["base"], r["variants"]
    log("")
    log(f"{'':<26}{'held head':>11}{'held com':>10}{'train head':>12}{'train com':>11}")
    log(f"{'base (before)':<26}{b['psnr_held_out']['head']:>11.3f}{b['psnr_held_out']['com']:>10.3f}{b['psnr_train_frames']['head']:>12.3f}{b['psnr_train_frames']['com']:>11.3f}")
    for k, i in v.items():
        log(f"{k:<26}{i['psnr_held_out']['head']:>11.3f}{i['psnr_held_out']['com']:>10.3f}{i['psnr_train_frames']['head']:>12.3f}{i['psnr_train_frames']['com']:>11.3f}")
    import numpy as np
    log("")
    log(f"paired differences against the exact tier's continuation on the SAME pixel sequence, over {n_seeds} pixel seeds (dB; + = better): mean +- standard error [min, max]")
    cols = (("psnr_held_out", "head"), ("psnr_held_out", "com"), ("psnr_train_frames", "head"), ("psnr_train_frames", "com"))
    for fmt in ("bf16_fp4", "bf16_e4m3"):
        parts = []
        for s_, im in cols:
            d = np.array([v[f"{fmt}_s{sd}"][s_][im] - v[f"f32_s{sd}"][s_][im] for sd in seeds])
            se = d.std(ddof=1) / np.sqrt(len(d)) if len(d) > 1 else float("nan")
            parts.append(f"{s_[5:]} {im} {d.mean():+.3f} +- {se:.3f} [{d.min():+.3f}, {d.max():+.3f}]")
        log(f"  {fmt:<10} " + "   ".join(parts))
    parts = []
    for s_, im in cols:
        x = np.array([v[f"f32_s{sd}"][s_][im] for sd in seeds])
        parts.append(f"{s_[5:]} {im} std {x.std(ddof=1) if len(x) > 1 else float('nan'):.3f} [{x.min():.3f}, {x.max():.3f}]")
    log("  the exact tier's own continuations across the seeds: " + "   ".join(parts))
    log("")
    log(json.dumps(r, default=str))
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
