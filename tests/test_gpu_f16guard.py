"""The f16 inference tier's range guard (dfanerf/f16guard.py): v_cvt_pk_f16_f32 does not saturate, so an
activation beyond 65504 renders inf / NaN silently.  The guard calibrates max |h_l| per layer in the exact tier and max |w| at
pack time and refuses loudly.  Reference semantics: decoder.py:277-349 is fp32 throughout - any checkpoint renders there."""
import numpy as np
import pytest
import torch

from dfanerf import engine, f16guard, run_nerf, synth
from dfanerf.decoder import Decoder

pytestmark = pytest.mark.gpu


def _renderer(dec_state, scene, latents, tier):
    dev = torch.device("cuda")
    dec = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in dec_state.items()})
    dec.to(dev)
    args = run_nerf.config_parser().parse_args(
        f"--expname t --concate_bg --dim_signal=96 --n_object=1 --use_deformation_field --hierarchical --N_importance 128 "
        f"--hip_tier {tier}".split())
    zs, za = [torch.from_numpy(v).to(dev) for v in latents]
    bg = (torch.from_numpy(scene["bg"]).float() / 255.0).to(dev)
    return run_nerf.FrameRenderer(dec, zs, za, bg, [scene["H"], scene["W"], scene["focal"], scene["cx"], scene["cy"]],
                                  scene["near"], scene["far"], args)


def _signals(dev):
    sh = torch.from_numpy(synth.synth_tensor(0, "g3/sig", (96,), 0.8)).to(dev)
    st = torch.from_numpy(synth.synth_tensor(0, "g3/sigt", (42,), 0.8)).to(dev)
    return lambda k: (sh, st)


def test_calibration_of_the_synthetic_network_passes(states, scene, latents):
    dev = torch.device("cuda")
    R = _renderer(states["decoder"], scene, latents, "f16")
    b = R.check_f16_range(list(scene["poses"]), scene["pose_body"], _signals(dev))
    pk = R.decoder.packed("f16")
    assert b is pk.f16_bounds and set(b) == {"head", "torso"} and pk.f16_weight_max < 100.0
    assert len(b["head"]) == 11 and len(b["torso"]) == 22
    top = max(v for d in b.values() for v in d.values())
    assert 2.0 < top < 200.0, top                                  # (LABNOTES.md 8.1: the synthetic network peaks at ~13)
    for f in ("head", "torso"):
        assert b[f]["positional encoding"] <= 1.0 + 1e-6 and b[f]["view encoding"] <= 1.0 + 1e-6      # sin / cos
        assert all(np.isfinite(v) and v >= 0 for v in b[f].values())
    # the bounds really bound what the f16 kernel converts: a frame renders finite, and no other tier is ever calibrated
    assert _renderer(states["decoder"], scene, latents, "bf16").check_f16_range(list(scene["poses"]), scene["pose_body"],
                                                                               _signals(dev)) is None


def test_out_of_range_activations_are_refused_not_rendered(states, scene, latents):
    """the head's activations scaled by 1e4 (max ~1.3e5 > 65504): the guard refuses and names the layer; without it the f16
    tier renders non-finite pixels where bf16 and f32 render the image"""
    dev = torch.device("cuda")
    big = synth.scale_head_activations(states["decoder"], 1.0e4)
    R = _renderer(big, scene, latents, "f16")
    with pytest.raises(f16guard.F16RangeError, match="max .activation. = .*hip_tier bf16"):
        R.check_f16_range(list(scene["poses"]), scene["pose_body"], _signals(dev))
    sh, st = _signals(dev)(0)
    n = 4096
    imgs = {}
    for tier in ("f16", "bf16", "f32"):
        Rt = _renderer(big, scene, latents, tier)
        rh, _ = Rt.render(scene["poses"][0], scene["pose_body"], [sh[None], None], st, ray_begin=100000, ray_count=n, fields=1)
        imgs[tier] = rh.cpu()
    assert not torch.isfinite(imgs["f16"]).all()                   # what the guard is there to prevent
    assert torch.isfinite(imgs["bf16"]).all() and torch.isfinite(imgs["f32"]).all()
    assert float((imgs["bf16"] - imgs["f32"]).abs().max()) < 0.2


def test_out_of_range_weight_is_refused_at_pack_time(states):
    dev = torch.device("cuda")
    st = dict(states["decoder"])
    w = st["blocks.2.weight"].copy()
    w[5, 7] = 1.0e5
    st["blocks.2.weight"] = w
    flat = engine.flatten_state(st, dev)
    with pytest.raises(f16guard.F16RangeError, match="max .parameter."):
        engine.PackedDecoder(flat, "f16")
    engine.PackedDecoder(flat, "bf16")                             # f32's exponent range: packs
    pk = engine.PackedDecoder(engine.flatten_state(states["decoder"], dev), "f16")
    assert 0.0 < pk.f16_weight_max < 100.0


# ---- the accuracy guard (round 6) ----------------------------------------------------------------------
def _sharpen(dec_state, gain):
    """a density-sharpened network: raw sigma and the colour logits scaled by `gain` (harder surfaces, saturated colours) - every
    activation stays where it was (the range guard sees the same network), the f16 rounding of the last hidden layers is
    amplified `gain`-fold in front of exp() and the sigmoid"""
    st = {k: v.copy() for k, v in dec_state.items()}
    for k in ("sigma_out.weight", "sigma_out.bias", "feat_out.weight", "feat_out.bias"):
        st[k] = st[k] * np.float32(gain)
    return st


def test_accuracy_guard_passes_the_synthetic_network_and_reports(states, scene, latents):
    dev = torch.device("cuda")
    R = _renderer(states["decoder"], scene, latents, "f16")
    st = R.check_f16_accuracy(list(scene["poses"]), scene["pose_body"], _signals(dev))
    assert set(st) == {"head", "com"} and st is R.decoder.packed("f16").f16_accuracy
    gate = f16guard.psnr_gate(30.0)
    assert abs(gate - 49.36) < 0.02                                          # 30 dB + 10 log10(1 / (10^0.005 - 1))
    for name, s in st.items():
        assert s["n_rays"] == 256 * min(8, len(scene["poses"])) and s["model_psnr_db"] is None
        assert s["psnr_db"] >= gate and s["worst_block_db"] >= gate - f16guard.BLOCK_SLACK_DB, (name, s)
        assert s["worst_block_db"] <= s["psnr_db"] + 1e-9 and 0.0 < s["max_abs"] < 0.1
    # ground truth on the sample: the model's own PSNR is measured and decides the gate (a perfect model cannot afford f16)
    sh, stt = _signals(dev)(0)
    R32 = _renderer(states["decoder"], scene, latents, "f32")
    exact = [R32.render(p, scene["pose_body"], [sh[None], None], stt)[::-1] for p in scene["poses"][:2]]       # (com, head) per frame
    with pytest.raises(f16guard.F16AccuracyError, match="model at"):
        R.check_f16_accuracy(list(scene["poses"][:2]), scene["pose_body"], _signals(dev),
                             targets=lambda k: (exact[k][1], exact[k][0]))
    assert R.decoder.packed("f16").f16_accuracy["head"]["model_psnr_db"] > 100.0
    # any other tier is never calibrated
    assert _renderer(states["decoder"], scene, latents, "bf16").check_f16_accuracy(list(scene["poses"]), scene["pose_body"],
                                                                                  _signals(dev)) is None


def test_sharpened_network_passes_the_range_guard_and_is_refused_for_accuracy(states, scene, latents, capsys):
    """the case the range guard cannot see: same activations, sharper density / colours.  The first gain whose f16 images fall
    under the clause's 49.4 dB is refused by name; --hip_tier auto renders that checkpoint in the exact tier, bit for bit."""
    dev = torch.device("cuda")
    poses, body, sig = list(scene["poses"]), scene["pose_body"], _signals(dev)
    refused = None
    for gain in (4.0, 16.0, 64.0, 256.0):
        R = _renderer(_sharpen(states["decoder"], gain), scene, latents, "f16")
        b = R.check_f16_range(poses, body, sig)                              # in range: the hidden activations did not move
        assert max(v for d in b.values() for v in d.values()) < 200.0
        try:
            st = R.check_f16_accuracy(poses, body, sig)
            print(f"gain {gain}: accepted, " + ", ".join(f"{n} {s['psnr_db']:.1f} dB" for n, s in st.items()))
        except f16guard.F16AccuracyError as e:
            refused = (gain, str(e), R.decoder.packed("f16").f16_accuracy)
            break
    assert refused is not None, "no gain up to 256 broke the clause"
    gain, msg, st = refused
    print(f"gain {gain}: refused: {msg}")
    assert "loses the accuracy clause" in msg and "hip_tier auto" in msg
    assert min(s["psnr_db"] for s in st.values()) < f16guard.psnr_gate(30.0) + 1e-9 or \
        min(s["worst_block_db"] for s in st.values()) < f16guard.psnr_gate(30.0) - f16guard.BLOCK_SLACK_DB
    # the same checkpoint with a better model claimed: refused as well; with a worse one (gate 39.4 dB): judged on its numbers
    Ra = _renderer(_sharpen(states["decoder"], gain), scene, latents, "auto")
    assert Ra.tier == "f16" and Ra.auto
    assert Ra.check_f16(poses, body, sig) == "f32" and Ra.tier == "f32"
    assert "rendering in the exact tier" in capsys.readouterr().out
    sh, stt = sig(0)
    n = 2048
    a = Ra.render(poses[0], body, [sh[None], None], stt, ray_begin=90000, ray_count=n)
    e = _renderer(_sharpen(states["decoder"], gain), scene, latents, "f32").render(poses[0], body, [sh[None], None], stt,
                                                                                  ray_begin=90000, ray_count=n)
    assert torch.equal(a[0], e[0]) and torch.equal(a[1], e[1])
    # auto on the unsharpened network stays in f16
    Rb = _renderer(states["decoder"], scene, latents, "auto")
    assert Rb.check_f16(poses, body, sig) == "f16"
