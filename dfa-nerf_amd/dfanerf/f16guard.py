"""Range guard and accuracy guard of the f16 inference tier.

The throughput tier carries the decoder's activations and weights as IEEE half precision (v_mfma_f32_32x32x16_f16:
10 mantissa bits, 5 exponent bits).  The conversion at the end of every layer (v_cvt_pk_f16_f32) does not saturate: an
activation above 65504 becomes inf, the next layer turns it into NaN, and the frame comes out black or NaN - silently.
The synthetic networks of the tests peak at |h| = 13; a trained checkpoint is whatever training made it.  So before the
f16 tier renders a sequence it is CALIBRATED:

  weights      max |w| over the decoder's parameters (what dfn_pack_weights rounds to f16), at pack time;
  activations  one launch of the EXACT tier's training forward (dfn_train_fwd, f32: its recorder writes every GEMM input -
               i.e. exactly the values the 16-bit tiers convert - feature-major) over a few hundred rays of up to eight
               frames of the sequence, with the sequence's own poses and conditioning signals: max |h_l| per layer.

check() refuses (F16RangeError, naming the layer) when a bound times MARGIN exceeds f16's largest value; the bounds stay
with the packed decoder (PackedDecoder.f16_bounds) so that a caller can read them.  bf16 and f32 have f32's exponent range
and need no guard.  Reference semantics of the path: decoder.py:277-349 (fp32 throughout upstream).

ACCURACY (round 6).  A checkpoint inside f16's RANGE can still lose the north star's clause - "PSNR within 0.05 dB of
reference" - to f16's 10 mantissa bits: a sharper density field (larger sigma gain) amplifies the rounding of the last
layers.  The clause is about the model's PSNR against ground truth; what can be measured without ground truth is the f16
image against the exact tier's.  An error uncorrelated with the model's own adds in mean square: the model's PSNR moves by
10 log10(1 + mse_f16 / mse_model) <= 0.05 dB  <=>  PSNR(f16 vs f32) >= PSNR(model) + 19.36 dB (psnr_gate: 49.4 dB for a
30-dB model, 54.4 for a 35-dB one).  accuracy_stats() takes the SAME calibration sample - a few hundred rays of up to eight
frames, rendered in both tiers with the production settings (two fields, the run's n_fine) - and reports the whole-sample
and the worst-frame PSNR per image; check_accuracy() refuses (F16AccuracyError) under the gate.  The model's own PSNR is
measured on the sample where ground truth exists (training / validation frames), a flag otherwise (--hip_f16_model_psnr,
default 30 dB: the level the reference's lineage reports).  `--hip_tier auto` renders in f16 when both guards pass and in
the exact tier when one refuses."""
import ctypes as C
import math

import numpy as np
import torch

from . import engine
from ._lib import FIELD_HEAD, FIELD_TORSO, check as _chk, lib
from .engine import _ptr, _stream

F16_MAX = 65504.0
MARGIN = 4.0          # an unseen frame may exceed the calibrated maximum; 4x still leaves f16 13 binades above a |h| of 13


CLAUSE_DB = 0.05      # BASELINE.json north_star: "PSNR within 0.05 dB of reference"
DEFAULT_MODEL_PSNR = 30.0
BLOCK_SLACK_DB = 3.0  # a single frame's few hundred rays are 1/8 of the sample: its PSNR may sit this far under the gate


class F16RangeError(RuntimeError):
    pass


class F16AccuracyError(RuntimeError):
    pass


def psnr_gate(model_psnr_db, clause_db=CLAUSE_DB):
    """PSNR (dB) of the f16 image against the exact tier's from which on the model's PSNR against ground truth cannot move by
    more than clause_db (errors uncorrelated with the model's own add in mean square)."""
    return float(model_psnr_db) + 10.0 * math.log10(1.0 / (10.0 ** (clause_db / 10.0) - 1.0))


def _psnr(mse):
    return float("inf") if mse <= 0.0 else -10.0 * math.log10(mse)


def accuracy_stats(blocks):
    """blocks: one dict per calibration frame, image name ("head" / "com") -> (rgb_f16 [n,3], rgb_f32 [n,3], gt [n,3] or None),
    float tensors in [0, 1].  -> {"head": {...}, "com": {...}} with psnr_db (whole sample, f16 against f32), worst_block_db,
    max_abs, n_rays, and model_psnr_db (f32 against ground truth, None without one)."""
    out = {}
    for name in sorted({k for b in blocks for k in b}):
        mses, gts, top, n = [], [], 0.0, 0
        for b in blocks:
            if name not in b:
                continue
            lo, hi, gt = b[name]
            d = lo.double() - hi.double()
            mses.append(float((d * d).mean()))
            top = max(top, float(d.abs().max()))
            n += lo.shape[0]
            if gt is not None:
                e = hi.double() - gt.double()
                gts.append(float((e * e).mean()))
        if not mses:
            continue
        out[name] = {"psnr_db": _psnr(sum(mses) / len(mses)), "worst_block_db": _psnr(max(mses)), "max_abs": top, "n_rays": n,
                     "blocks": len(mses), "model_psnr_db": _psnr(sum(gts) / len(gts)) if gts else None}
    return out


def check_accuracy(stats, model_psnr_db=None, what="the decoder"):
    """raise F16AccuracyError if an image of the calibration sample is under the clause's gate; -> the gates used, per image.
    model_psnr_db: the model's PSNR against ground truth where the sample had none (default DEFAULT_MODEL_PSNR); an image
    whose sample came with ground truth is gated on its own measured PSNR."""
    bad, gates = [], {}
    for name, st in stats.items():
        m = st.get("model_psnr_db")
        if m is None:
            m = DEFAULT_MODEL_PSNR if model_psnr_db is None else float(model_psnr_db)
        g = gates[name] = psnr_gate(m)
        if not (st["psnr_db"] >= g):
            bad.append(f"{name} image: {st['psnr_db']:.1f} dB against the exact tier over {st['n_rays']} rays, the clause needs "
                       f"{g:.1f} dB (model at {m:.1f} dB)")
        elif not (st["worst_block_db"] >= g - BLOCK_SLACK_DB):
            bad.append(f"{name} image: worst frame of the sample {st['worst_block_db']:.1f} dB against the exact tier, "
                       f"{BLOCK_SLACK_DB:g} dB under the clause's {g:.1f} dB (model at {m:.1f} dB)")
    if bad:
        raise F16AccuracyError(
            f"--hip_tier f16: {what} loses the accuracy clause (PSNR within {CLAUSE_DB} dB of the reference) in half precision: "
            + "; ".join(bad) + ".  Use --hip_tier f32 (exact) or --hip_tier auto (f16 only where both guards pass)")
    return gates


# groups of act_T rows (csrc/dfn_mlp.h: RecMap): every GEMM input of a field, in recorder order
def _groups(field):
    g, r = [], 0

    def add(name, n):
        nonlocal r
        g.append((name, r, r + n))
        r += n
    add("positional encoding", 64)
    if field == FIELD_TORSO:
        for k in range(5):
            add(f"deform_net.blocks_embed.{k} output", 64)
            add(f"deform_net.blocks_signal.{k} output", 64)
        add("deformed point / pose signal (fc_in_torso input)", 128)
    add("fc_in output", 256)
    for k in range(7):
        add(f"blocks.{k} output" + (" + skip" if k == 3 else ""), 256)
    add("feat_view output", 256)
    add("view encoding", 32)
    return g


def weight_bound(flat_params):
    """max |parameter| of the decoder (one device reduction + one host read)"""
    return float(flat_params.detach().abs().max())


def activation_bounds(flat_params, frames, sig_heads, sig_torsos, z_shape, z_app, bg, n_rays=256, seed=0, n_fine=0, z_dim=256):
    """max |GEMM input| per layer and field over n_rays random pixels of every frame in `frames` (engine.make_frame objects;
    ray_count / n_fine / fields are overwritten), with that frame's conditioning signals sig_heads[k] [96] / sig_torsos[k] [42].
    n_fine = 64 / 128: the hierarchical forward (dfn_train_fwd_hier) - the fine samples cluster at the surfaces, where the
    activations are largest, and the production render evaluates them.  -> {"head": {layer: max}, "torso": {...}}.
    Runs in the EXACT tier (f32 MFMAs, f32 recording)."""
    dev = flat_params.device
    pk = engine.PackedDecoder(flat_params, "f32", fields=(FIELD_HEAD, FIELD_TORSO), z_dim=z_dim)
    rows = [_chk(lib.dfn_train_rows(f, 0), "dfn_train_rows") for f in (0, 1)]
    mrows = [_chk(lib.dfn_train_rows(f, 2), "dfn_train_rows") for f in (0, 1)]
    n_coarse = 64 if n_fine else int(frames[0].n_coarse)      # (--N_samples 32 / 128 render coarse only)
    S = n_coarse + int(n_fine)
    NP = n_rays * S
    act = [torch.empty(NP // 32, rows[f], 32, dtype=torch.float32, device=dev) for f in (0, 1)]
    masks = [torch.empty(NP // 32, mrows[f], 64, dtype=torch.int32, device=dev) for f in (0, 1)]
    samples = torch.empty(NP, 8, dtype=torch.float32, device=dev)
    rgb = torch.empty(2, n_rays, 3, dtype=torch.float32, device=dev)
    z_all = torch.empty(n_rays, S, dtype=torch.float32, device=dev) if n_fine else None
    ranks = torch.empty(n_rays, S, dtype=torch.uint8, device=dev) if n_fine else None
    gen = torch.Generator(device="cpu").manual_seed(seed)
    top = [torch.zeros(rows[f], dtype=torch.float32, device=dev) for f in (0, 1)]
    nh = pk.bias_floats(FIELD_HEAD)
    bg_f32 = bg if bg.dtype == torch.float32 else None
    bg_u8 = bg if bg.dtype == torch.uint8 else None
    for k, fr in enumerate(frames):
        pix = torch.randperm(fr.H * fr.W, generator=gen)[:n_rays].to(torch.int32).to(dev)
        bias = pk.fold(sig_heads[k], sig_torsos[k], z_shape, z_app)
        fr.ray_begin, fr.ray_count, fr.n_coarse, fr.n_fine, fr.fields = 0, n_rays, n_coarse, int(n_fine), 2
        if n_fine:
            _chk(lib.dfn_train_fwd_hier(0, C.byref(fr), _ptr(pk.packed[FIELD_HEAD]), _ptr(pk.packed[FIELD_TORSO]), _ptr(bias),
                                        C.c_void_p(bias.data_ptr() + 4 * nh), _ptr(bg_f32), _ptr(bg_u8), _ptr(pix), _ptr(rgb[0]),
                                        _ptr(rgb[1]), _ptr(samples), _ptr(act[0]), _ptr(masks[0]), _ptr(act[1]), _ptr(masks[1]),
                                        _ptr(z_all), _ptr(ranks), _stream()), "dfn_train_fwd_hier(calibration)")
        else:
            _chk(lib.dfn_train_fwd(0, C.byref(fr), _ptr(pk.packed[FIELD_HEAD]), _ptr(pk.packed[FIELD_TORSO]), _ptr(bias),
                                   C.c_void_p(bias.data_ptr() + 4 * nh), _ptr(bg_f32), _ptr(bg_u8), _ptr(pix), _ptr(rgb[0]),
                                   _ptr(rgb[1]), _ptr(samples), _ptr(act[0]), _ptr(masks[0]), _ptr(act[1]), _ptr(masks[1]),
                                   _stream()), "dfn_train_fwd(calibration)")
        for f in (0, 1):
            top[f] = torch.maximum(top[f], act[f].abs().amax(dim=(0, 2)))
    out = {}
    for f, name in ((FIELD_HEAD, "head"), (FIELD_TORSO, "torso")):
        t = top[f].cpu().numpy()
        out[name] = {g: float(np.nanmax(t[a:b])) if np.isfinite(t[a:b]).all() else float("inf") for g, a, b in _groups(f)}
    return out


def check(bounds, weight_max=None, margin=MARGIN, what="the decoder"):
    """raise F16RangeError if a calibrated bound x margin does not fit f16; -> the largest activation bound otherwise"""
    bad, top = [], 0.0
    if weight_max is not None and not (weight_max * margin < F16_MAX):
        bad.append(f"max |parameter| = {weight_max:.4g}")
    for field, layers in (bounds or {}).items():
        for layer, v in layers.items():
            top = max(top, v)
            if not (v * margin < F16_MAX):
                bad.append(f"{field} field, {layer}: max |activation| = {v:.4g}")
    if bad:
        raise F16RangeError(
            f"--hip_tier f16: {what} does not fit half precision's range (largest value {F16_MAX:.0f}, calibration margin "
            f"x{margin:g}): " + "; ".join(bad) + ".  The f16 tier's conversions do not saturate - it would render inf / NaN.  "
            "Use --hip_tier bf16 (same speed, f32's exponent range, 7 mantissa bits) or --hip_tier f32 (exact)")
    return top
