#!/usr/bin/env python3
"""Developer tool (GPU box): the training tiers at the REFERENCE'S horizon - 100,000 steps (scripts/train_obama.sh trains 10^5-class
step counts in fp32) - on the teacher scene of tests/convergence.py: one exact-tier run, the 16-bit tier in both recorded-activation
formats on the same pixel sequence, and each format once more on another sequence.  Learning rate 1e-4 decayed to 1 % over the run.

    python tools/convergence_long.py [steps=100000] [name:tier:format:seed ...] > profiles/r05_convergence_100k.txt
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import convergence as CV      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
log = lambda s: print(s, flush=True)
log(f"# {steps} production steps of 2048 rays per variant, lr {CV.LRATE} decayed to 1 % (--lrate_decay {max(1, steps // 3000)}), "
    f"{CV.F_TRAIN} training frames, {CV.F_HELD} held-out frames; PSNR in the exact tier")
variants = [("f32", "f32", None, 100), ("bf16_fp4", "bf16", "fp4", 100), ("bf16_e4m3", "bf16", "e4m3", 100),
            ("bf16_fp4_s101", "bf16", "fp4", 101), ("bf16_e4m3_s101", "bf16", "e4m3", 101)]
if len(sys.argv) > 2:      # explicit variants: name:tier:format:pixel_seed ... (format "-" for the exact tier), compared against the first
    variants = [(n, t, None if f == "-" else f, int(sd)) for n, t, f, sd in (a.split(":") for a in sys.argv[2:])]
res = CV.run(steps, variants, curve_every=max(1, steps // 10), log=log,
             with_inference_check=len(sys.argv) <= 2 or os.environ.get("DFN_CONV_INFERENCE_CHECK") == "1")
v = res["variants"]
log("")
log(f"{'variant':<18}{'ms/step':>9}{'finite':>8}{'last loss':>12}{'held head':>11}{'held com':>10}{'train head':>12}{'train com':>11}")
for k, i in v.items():
    log(f"{k:<18}{i['ms_per_step']:>9.3f}{str(i['finite']):>8}{i['last_loss']:>12.2e}{i['psnr_held_out']['head']:>11.3f}{i['psnr_held_out']['com']:>10.3f}"
        f"{i['psnr_train_frames']['head']:>12.3f}{i['psnr_train_frames']['com']:>11.3f}")
log("")
ref = "f32" if "f32" in v else next(iter(v))
for k in [k for k in v if k != ref]:
    d = lambda s_, im: v[k][s_][im] - v[ref][s_][im]
    log(f"  {k:<18} minus {ref}: held-out head {d('psnr_held_out', 'head'):+.3f} com {d('psnr_held_out', 'com'):+.3f}   "
        f"training frames head {d('psnr_train_frames', 'head'):+.3f} com {d('psnr_train_frames', 'com'):+.3f}")
log("")
for k, i in v.items():
    if "f16_inference_vs_f32" in i:
        log(f"  {k} through the f16 inference tier vs the exact tier: {i['f16_inference_vs_f32']}")
    if "f16_accuracy_guard" in i:
        log(f"  {k}: the f16 tier's accuracy guard on these weights: {json.dumps(i['f16_accuracy_guard'])}")
log("")
log(json.dumps(res, default=str))
