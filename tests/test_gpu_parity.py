"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed goldens.
Run on a real MI355X with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import dfa_oracle as O
from dfanerf import synth

pytestmark = pytest.mark.gpu

torch.set_num_threads(max(1, min(32, torch.get_num_threads())))


def t(x):
    return torch.from_numpy(np.asarray(x))


@pytest.fixture(scope="module")
def eng():
    from dfanerf import engine
    engine.require_gpu()
    return engine


@pytest.fixture(scope="module")
def flat(eng, states):
    return eng.flatten_state(states["decoder"], "cuda")


@pytest.fixture(scope="module")
def packed(eng, flat):
    return {tier: eng.PackedDecoder(flat, tier, fields=(0, 1, 2)) for tier in ("f32", "bf16", "f16")}


def psnr(a, b):
    mse = float(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean())
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)


# ---------------------------------------------------------------------------------------------------------
def test_mfma_fragment_maps(eng):
    """Pins the C/D map and the (half, slot) pairing of both MFMA flavours the kernels rely on."""
    d = eng.mfma_layout_probe().cpu().numpy()
    i = np.arange(32)
    want = np.outer(i + 1, 2 * i + 1).astype(np.float32)       # asymmetric: catches a transposed map
    assert np.array_equal(d[0], want), "v_mfma_f32_32x32x16_bf16 fragment map"
    assert np.array_equal(d[1], want), "v_mfma_f32_32x32x2_f32 fragment map"
    assert np.array_equal(d[2], want), "v_mfma_f32_32x32x16_f16 fragment map"


def test_get_rays_bitwise_full_frame(eng, scene, golden):
    g = golden("g1_rays")
    H, W = scene["H"], scene["W"]
    for tag in ("a", "b"):
        pose = g["pose_" + tag]
        ro, rd = eng.get_rays(H, W, scene["focal"], pose[:3, :4], scene["cx"], scene["cy"])
        oro, ord_ = O.get_rays(H, W, scene["focal"], pose[:3, :4], scene["cx"], scene["cy"])
        assert np.array_equal(rd.cpu().numpy(), ord_.numpy())           # all 202,500 rays, bit-for-bit
        assert np.array_equal(ro.cpu().numpy(), oro.numpy())
        assert np.array_equal(rd.reshape(-1, 3)[t(g["idx"]).cuda()].cpu().numpy(), g["rays_d_" + tag])
        no, nd = eng.ndc_rays(H, W, scene["focal"], 1.0, ro.reshape(-1, 3), rd.reshape(-1, 3))
        idx = g["idx"]
        np.testing.assert_allclose(no.cpu().numpy()[idx], g["ndc_o_" + tag], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(nd.cpu().numpy()[idx], g["ndc_d_" + tag], rtol=2e-6, atol=1e-6)
    _, rd = eng.get_rays(8, 6, 100.0, scene["poses"][1][:3, :4])
    assert np.array_equal(rd.cpu().numpy(), g["small_rays_d"])
    # the `stride` argument (HELP:449-451; round 5 raised NotImplementedError): every ray bitwise against the oracle, the golden's
    # rays bitwise against the reference - through the drop-in helper
    from dfanerf import helpers
    gc = golden("g1c_rays_stride")
    for s in (2, 3, 7):
        ro, rd = helpers.get_rays(H, W, scene["focal"], t(gc["pose"][:3, :4]).cuda(), scene["cx"], scene["cy"], stride=s)
        oro, ord_ = O.get_rays(H, W, scene["focal"], gc["pose"][:3, :4], scene["cx"], scene["cy"], stride=s)
        assert tuple(rd.shape) == (H // s, W // s, 3) and np.array_equal(rd.cpu().numpy(), ord_.numpy())
        assert np.array_equal(ro.cpu().numpy(), oro.numpy())
        assert np.array_equal(rd.reshape(-1, 3)[t(gc[f"idx_{s}"]).cuda()].cpu().numpy(), gc[f"rays_d_{s}"])


def _assert_samples_match(got, bins, w, n, u=None):
    """sample_pdf against the oracle.  Conditioning of the reference algorithm sets the tolerance:
    - t = (u - cdf_below) / denom amplifies a CDF rounding difference (a few 1e-7 after a 60-term f32 cumsum
      whose normaliser torch.sum()s in an order that is not even stable across CPU vector widths) by
      1/denom: tolerance = 3e-7 + bin_width * 8e-7 / denom;
    - it is discontinuous where u coincides with a CDF knot (u = 1.0 against cdf[-1] ~ 1 is the common case)
      and where denom sits at its 1e-5 switch: there the sample may land anywhere in the neighbouring bin."""
    ref = O.sample_pdf(bins, w, n, det=u is None, u=u).numpy()
    wp = w + 1e-5
    cdf = torch.cat([torch.zeros_like(wp[:, :1]), torch.cumsum(wp / wp.sum(-1, keepdim=True), -1)], -1).numpy()
    uu = (O.linspace01(n)[None].expand(w.shape[0], n) if u is None else u).numpy()
    nb = cdf.shape[1]
    inds = (cdf[:, None, :] <= uu[:, :, None]).sum(-1)
    below, above = np.clip(inds - 1, 0, None), np.clip(inds, None, nb - 1)
    denom = np.take_along_axis(cdf, above, 1) - np.take_along_axis(cdf, below, 1)
    at_switch = np.abs(denom - 1e-5) < 1e-8
    denom = np.where(denom < 1e-5, 1.0, denom)
    width = float(np.diff(bins.numpy(), axis=1).max()) if bins.shape[1] > 1 else 0.0
    tol = 3e-7 + width * 8e-7 / denom
    bad = np.abs(got - ref) > tol
    near_knot = (np.abs(uu[:, :, None] - cdf[:, None, :]) < 1e-6).any(-1)
    assert not (bad & ~(near_knot | at_switch)).any(), np.argwhere(bad & ~(near_knot | at_switch))[:5]
    assert bad.mean() < 0.05
    assert np.abs(got - ref).max() <= width * 1.0001 + 1e-7


def test_sample_pdf(eng, golden):
    """HIP sample_pdf against (a) the oracle with the normaliser summed in the kernels' documented order: BIT FOR BIT,
    every mode, ragged shapes; (b) the reference's own output (golden G5 == oracle.sample_pdf bitwise) within the
    conditioning-aware tolerance - the two differ only by the rounding of torch.sum's ISA-dependent order."""
    g = golden("g5_sample_pdf")
    bins, w = t(g["bins"]), t(g["weights"])
    assert np.array_equal(O.sample_pdf(bins, w, 128, det=True).numpy(), g["det128"])   # oracle == reference here too
    u = t(g["u_pytest"])
    for n, uu in ((128, None), (16, None), (128, u), (64, None)):
        got = eng.sample_pdf(bins.cuda(), w.cuda(), n, det=uu is None, u=None if uu is None else uu.cuda()).cpu().numpy()
        ref = O.sample_pdf(bins, w, n, det=uu is None, u=uu, fixed_order=True).numpy()
        assert np.array_equal(got, ref), (n, np.abs(got - ref).max())                   # (a) bitwise
        _assert_samples_match(got, bins, w, n, u=uu)                                    # (b) vs the reference's sum order
    # the two normalisers differ by at most one ulp
    s_fix, s_torch = O.wave_sum64(w + 1e-5), torch.sum(w + 1e-5, -1, keepdim=True)
    assert float(((s_fix - s_torch).abs() / s_torch).max()) <= 4e-7          # a few ulp
    # wider rows than one wave: the per-lane sequential part of the documented order
    rs = np.random.RandomState(3)
    wb = t(rs.rand(37, 200).astype(np.float32) ** 4)
    bb = t(np.sort(rs.rand(37, 201).astype(np.float32), 1))
    got = eng.sample_pdf(bb.cuda(), wb.cuda(), 96, det=True).cpu().numpy()
    assert np.array_equal(got, O.sample_pdf(bb, wb, 96, det=True, fixed_order=True).numpy())
    # random (non-det) mode draws its own u: only the range is checkable
    r = eng.sample_pdf(bins.cuda(), w.cuda(), 64).cpu().numpy()
    assert (r >= g["bins"].min() - 1e-6).all() and (r <= g["bins"].max() + 1e-6).all()
    # ragged / empty shapes
    assert eng.sample_pdf(bins[:0].cuda(), w[:0].cuda(), 8, det=True).shape == (0, 8)
    b2, w1 = bins[:1, :2].contiguous(), w[:1, :1].contiguous()
    got = eng.sample_pdf(b2.cuda(), w1.cuda(), 5, det=True).cpu().numpy()
    assert np.array_equal(got, O.sample_pdf(b2, w1, 5, det=True, fixed_order=True).numpy())


def test_composite_and_weights(eng, golden):
    g = golden("g4_composite")
    sig, feat = t(g["sigma"]).cuda(), t(g["feat"]).cuda()
    s2, f2 = eng.composite(sig, feat)
    assert np.array_equal(s2.cpu().numpy(), g["sigma_sum2"])
    np.testing.assert_allclose(f2.cpu().numpy(), g["feat2"], atol=1e-7, rtol=0)
    s1, f1 = eng.composite(sig[:1], feat[:1])
    assert np.array_equal(s1.cpu().numpy(), g["sigma_sum1"]) and np.array_equal(f1.cpu().numpy(), g["feat1"])
    z, ray = t(g["z"]).cuda(), t(g["ray"]).cuda()
    for ss, key, ld in ((s2, "w2", 1e10), (s1, "w1", 1e10), (s1, "w_lastdist005", 0.05)):
        w = eng.volume_weights(z, ray, ss, last_dist=ld).cpu().numpy()
        np.testing.assert_allclose(w, g[key], atol=1e-6, rtol=0)


def test_to8b(eng, golden):
    g = golden("g10_to8b")
    assert np.array_equal(eng.to8b(t(g["x"]).cuda()).cpu().numpy(), g["y"])
    x = torch.rand(100003, device="cuda") * 1.2 - 0.1
    assert np.array_equal(eng.to8b(x).cpu().numpy(), O.to8b(x.cpu().numpy()))


# ---------------------------------------------------------------------------------------------------------
def _decoder_case(eng, packed, golden, latents, tier, field, S):
    g = golden("g3_decoder")
    zs, za = latents
    fi = 1 if field == 1 else 0
    sig = {0: g["sig_aud"][0], 1: g["sig_torso"][0], 2: None}[field]
    pk = packed[tier]
    bias = pk.fold_single(field, sig, zs[0, fi], za[0, fi])
    feat, sigma = eng.decoder_forward(pk, field, bias, t(g[f"p_{S}"][0]).cuda(), t(g[f"r_{S}"][0]).cuda())
    name = {0: "head", 1: "torso", 2: "listener"}[field]
    return feat.cpu().numpy(), sigma.cpu().numpy(), g[f"feat_{name}_{S}"][0], g[f"sigma_{name}_{S}"][0]


@pytest.mark.parametrize("field", [0, 1, 2])
@pytest.mark.parametrize("S", [64, 192])
def test_decoder_f32_tier_vs_reference_golden(eng, packed, golden, latents, field, S):
    feat, sigma, rf, rs = _decoder_case(eng, packed, golden, latents, "f32", field, S)
    np.testing.assert_allclose(feat, rf, atol=1e-5, rtol=0)              # 1e-5 abs
    np.testing.assert_allclose(sigma, rs, atol=2e-4, rtol=1e-5)          # |sigma| ~ 30: 1e-5 relative


@pytest.mark.parametrize("field", [0, 1, 2])
def test_decoder_bf16_tier_vs_reference_golden(eng, packed, golden, latents, field):
    feat, sigma, rf, rs = _decoder_case(eng, packed, golden, latents, "bf16", field, 192)
    print(f"bf16 field {field}: max|dfeat| {np.abs(feat - rf).max():.2e}  max|dsigma| {np.abs(sigma - rs).max():.2e}"
          f"  rms dsigma {np.sqrt(((sigma - rs) ** 2).mean()):.2e}")
    assert np.abs(feat - rf).max() < 2e-2
    assert np.sqrt(((sigma - rs) ** 2).mean()) < 0.5


@pytest.mark.parametrize("field", [0, 1, 2])
def test_decoder_f16_tier_vs_reference_golden(eng, packed, golden, latents, field):
    """f16 operands carry 10 mantissa bits (bf16: 7): 8x tighter gates than the bf16 tier's."""
    feat, sigma, rf, rs = _decoder_case(eng, packed, golden, latents, "f16", field, 192)
    print(f"f16 field {field}: max|dfeat| {np.abs(feat - rf).max():.2e}  max|dsigma| {np.abs(sigma - rs).max():.2e}"
          f"  rms dsigma {np.sqrt(((sigma - rs) ** 2).mean()):.2e}")
    assert np.abs(feat - rf).max() < 2.5e-3
    assert np.sqrt(((sigma - rs) ** 2).mean()) < 0.06


@pytest.mark.parametrize("field", [0, 1])
def test_decoder_f32_large_coordinates_vs_oracle(eng, packed, golden, latents, states, field):
    """Row A4 at the arguments where it is delicate: the 64 points of golden G3's `p_big` (|p| up to 1.6: the tenth octave's
    argument 2^9 pi p/2 reaches 1.3e3 rad, where a sin/cos with a sloppy range reduction is off in the third digit).  The
    kernel's positional encoding is only observable through the decoder, so the f32 tier's (feat, sigma) at those points are
    compared with the oracle's decoder (whose own encoding equals the reference's `pe_big` bit for bit,
    tests/test_oracle_golden.py).  A wrong octave anywhere in the 60 columns moves sigma by percents."""
    g = golden("g3_decoder")
    zs, za = latents
    P = O.params_to_torch(states["decoder"])
    p_big = t(g["p_big"])                                                        # [1,64,3]
    gen = torch.Generator().manual_seed(17)
    r = torch.nn.functional.normalize(torch.randn(1, 64, 3, generator=gen), dim=-1)
    sig = t(g["sig_aud"]) if field == 0 else t(g["sig_torso"])
    with torch.no_grad():
        rf, rs = O.decoder_forward(P, p_big, r, t(zs)[:, field], t(za)[:, field], [sig, None] if field == 0 else sig,
                                   'head' if field == 0 else 'torso')
    pk = packed["f32"]
    bias = pk.fold_single(field, sig[0].numpy(), zs[0, field], za[0, field])
    feat, sigma = eng.decoder_forward(pk, field, bias, p_big[0].cuda(), r[0].cuda())
    np.testing.assert_allclose(feat.cpu().numpy(), rf[0].numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(sigma.cpu().numpy(), rs[0].numpy(), atol=5e-4, rtol=2e-5)


def test_decoder_ragged_sizes(eng, packed, golden, latents):
    g = golden("g3_decoder")
    zs, za = latents
    pk = packed["f32"]
    bias = pk.fold_single(0, g["sig_aud"][0], zs[0, 0], za[0, 0])
    p, r = t(g["p_192"][0]).cuda(), t(g["r_192"][0]).cuda()
    full_f, full_s = eng.decoder_forward(pk, 0, bias, p, r)
    for n in (1, 31, 33, 129, 700):
        f, s = eng.decoder_forward(pk, 0, bias, p[:n], r[:n])
        assert torch.equal(f, full_f[:n]) and torch.equal(s, full_s[:n])
    f0, s0 = eng.decoder_forward(pk, 0, bias, p[:0], r[:0])
    assert f0.shape == (0, 3) and s0.shape == (0,)


# ---------------------------------------------------------------------------------------------------------
def _render_subset(eng, packed, scene, latents, signal, signal_torso, ray_idx, tier, n_fine, fields, frame_i=2,
                   want_weights=False, want_z=False, n_coarse=64):
    zs, za = latents
    pk = packed[tier]
    bias = pk.fold(signal, signal_torso if fields == 2 else None, zs[0], za[0])
    fr = eng.make_frame(scene["H"], scene["W"], scene["focal"], scene["cx"], scene["cy"], scene["poses"][frame_i],
                        scene["pose_body"], scene["near"], scene["far"], ray_count=len(ray_idx), n_fine=n_fine,
                        fields=fields, n_coarse=n_coarse)
    bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3).cuda()
    return eng.render(pk, bias, fr, bg, pix_index=t(np.asarray(ray_idx, np.int32)).cuda(), want_weights=want_weights,
                      want_z=want_z)


def test_decoder_with_a_narrower_latent_code_vs_reference_golden(scene, golden):
    """--z_dim is free upstream (MAIN:372; Decoder(z_dim=args.z_dim), MAIN:518); round 5 refused everything but 256.  A decoder with
    z_dim = 64 RENDERS: the three layers the latent codes feed act on per-frame constants only, so they enter the library's 256-wide
    slots zero-padded (engine.flatten_state / PackedDecoder.pad_z).  The drop-in module against golden G16 - the reference's own
    Decoder(z_dim=64) - head, torso, listener at 1e-5 in the exact tier; a frame through FrameRenderer equals the one rendered from
    hand-padded 256-wide weights bit for bit (training: tests/test_gpu_train.py)."""
    from dfanerf import run_nerf
    from dfanerf.decoder import Decoder
    dev = torch.device("cuda")
    g, g3 = golden("g16_z_dim_64"), golden("g3_decoder")
    st = synth.synth_decoder_state(0, z_dim=64)
    dec = Decoder(z_dim=64, hidden_size=256, dim_signal=96, use_deformation_field=True)
    dec.load_state_dict({k: t(v) for k, v in st.items()})
    dec.to(dev)
    zs, za = [t(v).to(dev) for v in synth.synth_latents(0, z_dim=64)]
    p, r = t(g3["p_64"]).to(dev), t(g3["r_64"]).to(dev)
    sa, stt = t(g3["sig_aud"]).to(dev), t(g3["sig_torso"]).to(dev)
    with torch.no_grad():
        out = {"head": dec(p, r, zs[:, 0], za[:, 0], [sa, None], "head"), "torso": dec(p, r, zs[:, 1], za[:, 1], stt, "torso"),
               "listener": dec(p, r, zs[:, 0], za[:, 0], [None, None], "head")}
    for k, (f, s) in out.items():
        np.testing.assert_allclose(f.cpu().numpy(), g["feat_" + k], atol=1e-5, rtol=0)
        np.testing.assert_allclose(s.cpu().numpy(), g["sigma_" + k], rtol=1e-5, atol=2e-4)        # (the gates of golden G3's test)
    # the frame renderer: 64-wide codes and weights = the same network written out 256 wide by hand
    args = run_nerf.config_parser().parse_args("--expname t --concate_bg --dim_signal=96 --n_object=1 --use_deformation_field --z_dim 64 "
                                               "--render_person --hierarchical --N_importance 128".split())
    run_nerf.check_supported(args)
    bg = (t(scene["bg"]).float() / 255.0).to(dev)
    geo = [scene["H"], scene["W"], scene["focal"], scene["cx"], scene["cy"]]
    R = run_nerf.FrameRenderer(dec, zs, za, bg, geo, scene["near"], scene["far"], args)
    st256 = dict(st)
    for k in ("fc_z.weight", "fc_z_skips.0.weight", "fc_z_view.weight"):
        st256[k] = np.pad(st[k], ((0, 0), (0, 192)))
    dec256 = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
    dec256.load_state_dict({k: t(v) for k, v in st256.items()})
    dec256.to(dev)
    pad = lambda z: torch.nn.functional.pad(z, (0, 192))
    args256 = run_nerf.config_parser().parse_args("--expname t --concate_bg --dim_signal=96 --n_object=1 --use_deformation_field "
                                                  "--render_person --hierarchical --N_importance 128".split())
    R256 = run_nerf.FrameRenderer(dec256, pad(zs), pad(za), bg, geo, scene["near"], scene["far"], args256)
    a = R.render(scene["poses"][1], scene["pose_body"], [sa, None], stt[0], ray_begin=80000, ray_count=4096)
    b = R256.render(scene["poses"][1], scene["pose_body"], [sa, None], stt[0], ray_begin=80000, ray_count=4096)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and float(a[1].std()) > 0.01


def test_narrower_decoder_vs_reference_golden(golden):
    """--n_feat is free upstream (MAIN:374; 128 is the Decoder class's own default); round 5 refused everything but 256.  A decoder of
    hidden width 128 (and latent width 64) RENDERS: written out 256 wide with zero rows / columns it is the same function exactly
    (engine.flatten_state).  Against golden G18 - the reference's own Decoder(hidden_size=128, z_dim=64) - head, torso, listener at
    G3's gates in the exact tier (training: tests/test_gpu_train.py)."""
    from dfanerf.decoder import Decoder
    dev = torch.device("cuda")
    g, g3 = golden("g18_n_feat_128"), golden("g3_decoder")
    dec = Decoder(z_dim=64, hidden_size=128, dim_signal=96, use_deformation_field=True)
    dec.load_state_dict({k: t(v) for k, v in synth.synth_decoder_state(0, z_dim=64, hidden=128).items()})
    dec.to(dev)
    zs, za = [t(v).to(dev) for v in synth.synth_latents(0, z_dim=64)]
    p, r = t(g3["p_64"]).to(dev), t(g3["r_64"]).to(dev)
    sa, stt = t(g3["sig_aud"]).to(dev), t(g3["sig_torso"]).to(dev)
    with torch.no_grad():
        out = {"head": dec(p, r, zs[:, 0], za[:, 0], [sa, None], "head"), "torso": dec(p, r, zs[:, 1], za[:, 1], stt, "torso"),
               "listener": dec(p, r, zs[:, 0], za[:, 0], [None, None], "head")}
    for k, (f, s) in out.items():
        np.testing.assert_allclose(f.cpu().numpy(), g["feat_" + k], atol=1e-5, rtol=0)
        np.testing.assert_allclose(s.cpu().numpy(), g["sigma_" + k], rtol=1e-5, atol=2e-4)
    with torch.no_grad():
        f16, _ = dec(p, r, zs[:, 1], za[:, 1], stt, "torso", tier="f16")
    assert float((f16.cpu() - t(g["feat_torso"])).abs().max()) < 3e-3


def test_decoder_without_deformation_field_vs_reference_golden(states, latents, golden):
    """--use_deformation_field is a store_true flag upstream (MAIN:411); without it the torso skips decoder.py:297-299.  Round 5
    required it.  Rendering now takes such a decoder: the fused torso program evaluates `deform(p) + p` with an all-zero deformation
    network (engine.flatten_state), which returns p exactly.  Against golden G17 (the reference's own Decoder without the flag), all
    three tiers (training: tests/test_gpu_train.py)."""
    from dfanerf.decoder import Decoder
    dev = torch.device("cuda")
    g, g3 = golden("g17_no_deformation_field"), golden("g3_decoder")
    dec = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=False)
    dec.load_state_dict({k: t(v) for k, v in states["decoder"].items() if not k.startswith("deform_net.")})
    dec.to(dev)
    assert not any(k.startswith("deform_net.") for k in dec.state_dict())
    zs, za = [t(v).to(dev) for v in latents]
    p, r, stt = t(g3["p_64"]).to(dev), t(g3["r_64"]).to(dev), t(g3["sig_torso"]).to(dev)
    with torch.no_grad():
        f, s = dec(p, r, zs[:, 1], za[:, 1], stt, "torso", tier="f32")
        fh, _ = dec(p, r, zs[:, 0], za[:, 0], [t(g3["sig_aud"]).to(dev), None], "head", tier="f32")
    np.testing.assert_allclose(f.cpu().numpy(), g["feat_torso"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(s.cpu().numpy(), g["sigma_torso"], rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(fh.cpu().numpy(), g3["feat_head_64"], atol=1e-5, rtol=0)              # the head never saw the flag
    for tier, tol in (("f16", 3e-3), ("bf16", 3e-2)):
        with torch.no_grad():
            ft, _ = dec(p, r, zs[:, 1], za[:, 1], stt, "torso", tier=tier)
        assert float((ft.cpu() - t(g["feat_torso"])).abs().max()) < tol, tier


@pytest.mark.parametrize("n_coarse", [32, 128])
def test_render_coarse_other_sample_counts_vs_reference_golden(eng, packed, scene, latents, golden, n_coarse):
    """--N_samples 32 / 128 (MAIN:612-619; round 5 refused everything but 64): the coarse renderer against golden G15, the
    reference's own loop at those counts - depths bitwise, weights 2e-6, RGB 2e-5 in the exact tier; the f16 tier at its
    image gate; the hierarchical mode keeps 64 coarse samples and says so."""
    g = golden("g15_coarse_nsamples")
    S = n_coarse
    out = _render_subset(eng, packed, scene, latents, g["signal"][0], g["signal_torso"].reshape(-1), g["ray_idx"], "f32", 0, 2,
                         want_weights=True, want_z=True, n_coarse=S)
    rh, rc, wh, wc, z = [o.cpu().numpy() for o in out]
    assert z.shape == (len(g["ray_idx"]), S) and np.array_equal(z[0], g[f"z_{S}"]) and (z == z[0]).all()
    np.testing.assert_allclose(wh, g[f"w_head_{S}"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(wc, g[f"w_com_{S}"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(rh, g[f"rgb_head_{S}"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(rc, g[f"rgb_com_{S}"], atol=2e-5, rtol=0)
    rh1, rc1 = _render_subset(eng, packed, scene, latents, g["signal"][0], None, g["ray_idx"], "f32", 0, 1, n_coarse=S)
    assert rc1 is None
    np.testing.assert_allclose(rh1.cpu().numpy(), rh, atol=1e-6, rtol=0)
    for tier, gate in (("f16", 55.0), ("bf16", 42.0)):
        th, tc = _render_subset(eng, packed, scene, latents, g["signal"][0], g["signal_torso"].reshape(-1), g["ray_idx"], tier, 0, 2,
                                n_coarse=S)
        ph, pc = psnr(th.cpu().numpy(), g[f"rgb_head_{S}"]), psnr(tc.cpu().numpy(), g[f"rgb_com_{S}"])
        print(f"N_samples {S}, {tier}: head {ph:.1f} dB, com {pc:.1f} dB against the reference")
        assert min(ph, pc) >= gate
    with pytest.raises(Exception, match="n_coarse = 64"):
        _render_subset(eng, packed, scene, latents, g["signal"][0], None, g["ray_idx"][:8], "f32", 128, 1, n_coarse=S)


def test_render_coarse_f32_vs_reference_golden(eng, packed, scene, latents, golden):
    """MAIN:653-709 semantics (coarse only, both images) against the imported-reference golden G7."""
    g = golden("g7_frame_coarse")
    out = _render_subset(eng, packed, scene, latents, g["signal"][0], g["signal_torso"].reshape(-1), g["ray_idx"],
                         "f32", 0, 2, want_weights=True)
    rh, rc, wh, wc = [o.cpu().numpy() for o in out]
    np.testing.assert_allclose(rh, g["rgb_head"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(rc, g["rgb_com"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(wh[:8], g["w_head_first8"], atol=2e-6, rtol=0)      # 1e-6 class
    np.testing.assert_allclose(wc[:8], g["w_com_first8"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(wh.sum(1), 1.0, atol=1e-5)
    for got, ref in ((rh, "rgb8_head"), (rc, "rgb8_com")):
        d = np.abs(O.to8b(got).astype(int) - g[ref].astype(int))
        assert d.max() <= 1 and (d > 0).mean() <= 2e-3
    print("PSNR f32 coarse vs reference: head %.1f dB, com %.1f dB" % (psnr(rh, g["rgb_head"]), psnr(rc, g["rgb_com"])))
    # single-field call gives the same head image
    rh1, rc1 = _render_subset(eng, packed, scene, latents, g["signal"][0], None, g["ray_idx"], "f32", 0, 1)
    assert rc1 is None
    np.testing.assert_allclose(rh1.cpu().numpy(), rh, atol=1e-6, rtol=0)


@pytest.mark.parametrize("fields", [1, 2])
def test_render_hierarchical_f32_vs_reference_golden(eng, packed, scene, latents, golden, states, fields):
    """Row H (64+128) against golden G7-hier, composed in make_golden.py from the reference's own
    sample_pdf / Decoder / composite_function / calc_volume_weights.
    sample_pdf is discontinuous at its `denom < 1e-5` switch, and empty space sits exactly there
    (pdf = 1e-5 / sum(w + 1e-5) with sum ~ 1), so a rounding difference in the CDF moves a fine sample inside
    an empty bin.  Staged check: (1) rays whose 192 merged depths all match the reference must match its RGB
    tightly; (2) every depth is within one coarse bin of the reference's; (3) on ALL rays the decoder +
    compositing at the GPU's own depths must match the oracle evaluated at those same depths."""
    g = golden("g7_frame_hier")
    gc = golden("g7_frame_coarse")
    idx = g["ray_idx"]
    rh, rc, z = _render_subset(eng, packed, scene, latents, gc["signal"][0], gc["signal_torso"].reshape(-1),
                               idx, "f32", 128, fields, want_z=True)
    rh, z = rh.cpu().numpy(), z.cpu().numpy()
    zref = g[f"z_all_f{fields}"]
    assert (np.diff(z, axis=1) >= 0).all() and np.allclose(z[:, 0], 0.3) and np.allclose(z[:, -1], 0.9)
    dz = np.abs(z - zref)
    assert dz.max() <= (0.6 / 63) * 1.001                                   # (2)
    clean = (dz <= 2e-6).all(1)
    print(f"fields={fields}: {clean.mean() * 100:.1f}% of rays have all 192 depths equal to the reference's; "
          f"{(dz > 2e-6).mean() * 100:.3f}% of depths moved")
    # measured (round 2, after the kernel's cumsum accumulates in double like torch's CPU cumsum): 99.85 % of the depths
    # and 71-80 % of the rays identical; the sampler itself is bit-exact given the coarse weights
    # (test_hierarchical_sampler_is_bit_exact_given_the_coarse_weights), so what moves a depth is the ~1e-7 difference
    # of the coarse weights at sample_pdf's `denom < 1e-5` switch
    # (round 6: the gate follows the measurement - 0.70, was 0.65; the box-to-box spread of the share is the coarse weights' 1e-7)
    assert clean.mean() >= 0.70 and (dz > 2e-6).mean() < 0.003
    np.testing.assert_allclose(rh[clean], g[f"rgb_head_f{fields}"][clean], atol=5e-5, rtol=0)      # (1)
    if fields == 2:
        np.testing.assert_allclose(rc.cpu().numpy()[clean], g[f"rgb_com_f{fields}"][clean], atol=5e-5, rtol=0)
    # (1b) what the switch flips cost in the IMAGE, on ALL rays: the distance of the final image of the configuration to the
    # golden, printed and gated - max, 99.9th percentile, PSNR over every ray.  A moved fine depth stays inside an (almost)
    # empty coarse bin, so the colour moves by what that bin's few 1e-5 of weight can carry.
    final, ref = (rc.cpu().numpy(), g[f"rgb_com_f{fields}"]) if fields == 2 else (rh, g[f"rgb_head_f{fields}"])
    d_all = np.abs(final - ref).max(1)
    p999 = float(np.quantile(d_all, 0.999))
    print(f"fields={fields}: |RGB - golden| over ALL {len(d_all)} rays: max {d_all.max():.2e}, 99.9th pct {p999:.2e}, "
          f"on the rays with a moved depth: max {d_all[~clean].max() if (~clean).any() else 0.0:.2e}; PSNR {psnr(final, ref):.1f} dB")
    # measured: head image max 3.1e-4 / 94.8 dB, two-field composite 2.2e-6 / 136.9 dB
    assert d_all.max() <= 1e-3 and psnr(final, ref) >= 85.0
    # ... and as the uint8 image the reference writes (to8b, run_nerf_helpers.py:17: truncation): never more than one level
    # off, one level on at most 0.5 % of the values ("identical except +-1 LSB")
    to8 = lambda x: (255.0 * np.clip(x, 0.0, 1.0)).astype(np.uint8).astype(np.int32)
    d8 = np.abs(to8(final) - to8(ref))
    print(f"fields={fields}: uint8 image vs golden: {(d8 > 0).mean() * 100:.3f}% of the values differ, max {d8.max()} level(s)")
    assert d8.max() <= 1 and (d8 > 0).mean() <= 0.005
    # (3)
    P = O.params_to_torch(states["decoder"])
    zs, za = [t(v) for v in latents]
    H, W = scene["H"], scene["W"]
    o_h, d_h = O.get_rays(H, W, scene["focal"], scene["poses"][2][:3, :4], scene["cx"], scene["cy"])
    o_t, d_t = O.get_rays(H, W, scene["focal"], scene["pose_body"][:3, :4], scene["cx"], scene["cy"])
    rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h, o_t, d_t)]
    bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3)[idx]
    with torch.no_grad():
        oh, oc = O.render_fixed_samples(P, *rays, bg, t(z), zs, za, [t(gc["signal"]), None],
                                        t(gc["signal_torso"]), fields)
    np.testing.assert_allclose(rh, oh.numpy(), atol=5e-5, rtol=0)
    if fields == 2:
        np.testing.assert_allclose(rc.cpu().numpy(), oc.numpy(), atol=5e-5, rtol=0)


@pytest.mark.parametrize("tier", ["f32", "f16"])
@pytest.mark.parametrize("fields,n_fine", [(1, 128), (2, 128), (2, 64)])
def test_hierarchical_sampler_is_bit_exact_given_the_coarse_weights(eng, packed, scene, latents, golden, tier, fields,
                                                                     n_fine):
    """Row H index work, exact: the fused kernel's inverse-CDF sampler + rank merge against the oracle (sample_pdf with
    the documented sum order, torch.sort) fed with the SAME coarse weights (the kernel's own, from a coarse-only launch:
    identical arithmetic, so identical bits): all 64 + n_fine merged depths of every ray bit for bit.  What is left
    between the kernel and the reference golden in the hierarchical tests is therefore only the 1e-7-level difference
    of the coarse weights themselves, amplified by sample_pdf's `denom < 1e-5` switch in empty space."""
    gc = golden("g7_frame_coarse")
    idx = np.arange(3, scene["H"] * scene["W"], 211)[:960]
    sig, sigt = gc["signal"][0], gc["signal_torso"].reshape(-1)
    out = _render_subset(eng, packed, scene, latents, sig, sigt, idx, tier, 0, fields, want_weights=True, want_z=True)
    w = (out[3] if fields == 2 else out[2]).cpu()
    z = out[-1].cpu()
    assert np.array_equal(z.numpy(), O.coarse_z(scene["near"], scene["far"], 64)[None].expand(len(idx), 64).numpy())
    z_all = _render_subset(eng, packed, scene, latents, sig, sigt, idx, tier, n_fine, fields, want_z=True)[-1].cpu()
    z_mid = .5 * (z[..., 1:] + z[..., :-1])
    z_f = O.sample_pdf(z_mid, w[..., 1:-1], n_fine, det=True, fixed_order=True)
    want, _ = torch.sort(torch.cat([z, z_f], -1), -1)
    assert np.array_equal(z_all.numpy(), want.numpy()), float((z_all - want).abs().max())


# "PSNR within 0.05 dB of reference" (BASELINE.json north_star) for a model that itself scores 30 dB against ground truth:
# an independent error of x dB costs 10 log10(1 + 10^((30 - x) / 10)) dB  ->  x >= 49.4 dB keeps the loss <= 0.05 dB.
PSNR_CLAUSE_DB = 49.4
# the bf16 tier (7 mantissa bits) is the 16-bit TRAINING tier; its rendered RGB is gated at what it measures
PSNR_GATE = {"f16": PSNR_CLAUSE_DB, "bf16": 42.0}


@pytest.mark.parametrize("tier", ["f16", "bf16"])
@pytest.mark.parametrize("n_fine,fields", [(0, 1), (128, 1), (128, 2), (64, 2)])
def test_render_16bit_psnr_vs_oracle(eng, packed, scene, latents, golden, states, tier, n_fine, fields):
    """16-bit tiers: PSNR of the rendered RGB against the fp32 ORACLE on a strided subset of the frame.  The f16 tier
    (the throughput tier bench.py reports) must hold the north star's accuracy clause (>= 49.4 dB, see above)."""
    gc = golden("g7_frame_coarse")
    idx = np.arange(0, scene["H"] * scene["W"], 397)[:256]
    sig, sigt = gc["signal"][0], gc["signal_torso"].reshape(-1)
    rh, rc = _render_subset(eng, packed, scene, latents, sig, sigt, idx, tier, n_fine, fields)
    P = O.params_to_torch(states["decoder"])
    zs, za = [t(v) for v in latents]
    H, W = scene["H"], scene["W"]
    o_h, d_h = O.get_rays(H, W, scene["focal"], scene["poses"][2][:3, :4], scene["cx"], scene["cy"])
    o_t, d_t = O.get_rays(H, W, scene["focal"], scene["pose_body"][:3, :4], scene["cx"], scene["cy"])
    rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h, o_t, d_t)]
    bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3)[idx]
    with torch.no_grad():
        oh, oc = O.render_rays_chunk(P, *rays, bg, scene["near"], scene["far"], zs, za, [t(sig)[None], None],
                                     t(sigt)[None], 64, n_fine, fields)
    p_h = psnr(rh.cpu().numpy(), oh.numpy())
    msg = f"{tier} n_fine={n_fine} fields={fields}: PSNR(head) {p_h:.1f} dB"
    if fields == 2:
        p_c = psnr(rc.cpu().numpy(), oc.numpy())
        msg += f", PSNR(com) {p_c:.1f} dB"
    print(msg)
    assert p_h >= PSNR_GATE[tier], msg
    if fields == 2:
        assert p_c >= PSNR_GATE[tier], msg


@pytest.mark.parametrize("fields", [1, 2])
def test_full_frame_psnr_16bit_tiers_vs_f32_tier(eng, packed, scene, latents, golden, fields):
    """BASELINE configs[1] / configs[2] at FULL size (all 202,500 rays, 64+128 samples): the 16-bit tiers against the f32
    tier on the device.  The f32 tier is pinned to the reference's own output (test_render_coarse_f32_vs_reference_golden,
    139 dB), so it is a valid on-device proxy for the reference at a size the CPU oracle cannot render in a test.
    f16 (the tier bench.py's headline is measured on): >= 49.4 dB on the whole frame AND on the worst 2,500-ray block."""
    gc = golden("g7_frame_coarse")
    zs, za = latents
    H, W = scene["H"], scene["W"]
    R = H * W
    bg8 = t(scene["bg"]).reshape(-1, 3).cuda()
    fr = eng.make_frame(H, W, scene["focal"], scene["cx"], scene["cy"], scene["poses"][2], scene["pose_body"],
                        scene["near"], scene["far"], ray_count=R, n_fine=128, fields=fields)
    out = {}
    for tier in ("f32", "f16", "bf16"):
        pk = packed[tier]
        bias = pk.fold(gc["signal"][0], gc["signal_torso"].reshape(-1) if fields == 2 else None, zs[0], za[0])
        rh, rc = eng.render(pk, bias, fr, bg8)
        out[tier] = [x.cpu().numpy().astype(np.float64) for x in ((rh, rc) if fields == 2 else (rh,))]
    for tier in ("f16", "bf16"):
        for img, ref, name in zip(out[tier], out["f32"], ("head", "com")):
            whole = psnr(img, ref)
            blk = ((img - ref) ** 2).reshape(-1, 2500, 3).mean((1, 2))          # 81 blocks of 2,500 rays
            worst = -10.0 * np.log10(blk.max())
            print(f"full frame fields={fields} {tier} {name}: PSNR {whole:.1f} dB, worst 2500-ray block {worst:.1f} dB, "
                  f"max |err| {np.abs(img - ref).max():.2e}")
            assert whole >= PSNR_GATE[tier], (tier, name, whole)
            if tier == "f16":
                assert worst >= PSNR_CLAUSE_DB, (tier, name, worst)
                # the clause itself, MEASURED instead of derived: "PSNR within 0.05 dB of reference" is a statement about the
                # PSNR a user computes against ground truth.  Ground truth here = the exact tier's frame plus what a trained
                # model still misses (independent noise at 30 dB, three seeds); the f16 frame's PSNR against it may be at most
                # 0.05 dB below the f32 frame's - on the whole frame and on every 2,500-ray block
                for seed in range(3):
                    rs = np.random.RandomState(seed)
                    gt = ref + rs.randn(*ref.shape) * 10 ** (-30 / 20)
                    loss = psnr(ref, gt) - psnr(img, gt)
                    e_ref = ((ref - gt) ** 2).reshape(-1, 2500, 3).mean((1, 2))
                    e_img = ((img - gt) ** 2).reshape(-1, 2500, 3).mean((1, 2))
                    loss_blk = float((10.0 * np.log10(e_img / e_ref)).max())
                    if seed == 0:
                        print(f"    measured PSNR loss of the f16 frame against a 30-dB ground truth: whole frame {loss:.4f} dB, "
                              f"worst block {loss_blk:.4f} dB (clause: 0.05)")
                    # (a block's 7,500 values leave a +-0.01 dB sampling term of the noise / error cross product)
                    assert loss <= 0.05 and loss_blk <= 0.05 + 0.015, (name, seed, loss, loss_blk)
        # the u8 image a user sees: identical up to +-1 LSB except where the sampler's switch moved a depth
        d = np.abs(O.to8b(out[tier][0]).astype(int) - O.to8b(out["f32"][0]).astype(int))
        print(f"  u8 head image {tier}: {(d > 0).mean() * 100:.2f} % of values differ, max {d.max()} LSB")


def test_render_full_frame_properties(eng, packed, scene, latents, golden):
    """Full 450x450 frame, bf16 tier, 64+128: size-independent properties.
    - run-to-run determinism (bitwise);  - ray sharding invariance (two half-frame calls == one call, bitwise:
      this is the multi-GPU partition);  - explicit pixel list == contiguous range;  - ragged tail (ray counts
      that are not a multiple of the 8 rays per workgroup);  - weights sum to one;  - rgb in [0,1]."""
    gc = golden("g7_frame_coarse")
    zs, za = latents
    pk = packed["bf16"]
    bias = pk.fold(gc["signal"][0], gc["signal_torso"].reshape(-1), zs[0], za[0])
    H, W = scene["H"], scene["W"]
    R = H * W
    bg8 = t(scene["bg"]).reshape(-1, 3).cuda()                  # uint8 background path
    mk = lambda b, n, nf=128, fl=1: eng.make_frame(H, W, scene["focal"], scene["cx"], scene["cy"], scene["poses"][2],
                                                   scene["pose_body"], scene["near"], scene["far"], ray_begin=b,
                                                   ray_count=n, n_fine=nf, fields=fl)
    full, _ = eng.render(pk, bias, mk(0, R), bg8)
    again, _ = eng.render(pk, bias, mk(0, R), bg8)
    assert torch.equal(full, again)
    cut = 101251                                               # not a multiple of 8
    a, _ = eng.render(pk, bias, mk(0, cut), bg8)
    b, _ = eng.render(pk, bias, mk(cut, R - cut), bg8)
    assert torch.equal(torch.cat([a, b]), full)
    idx = torch.arange(5000, 5000 + 1003, dtype=torch.int32, device="cuda")
    c, _ = eng.render(pk, bias, mk(0, 1003), bg8, pix_index=idx)
    assert torch.equal(c, full[5000:6003])
    f = full.cpu().numpy()
    assert np.isfinite(f).all() and f.min() >= -1e-6 and f.max() <= 1 + 1e-5
    bgf = (t(scene["bg"]).float() / 255.0).reshape(-1, 3).cuda()
    g32, _ = eng.render(pk, bias, mk(0, 4096), bgf)
    assert torch.equal(g32, full[:4096])                        # u8 and f32 background paths agree
    _, _, wh, wc = eng.render(pk, bias, mk(0, 2048, 128, 2), bg8, want_weights=True)
    np.testing.assert_allclose(wh.sum(1).cpu().numpy(), 1.0, atol=2e-5)
    np.testing.assert_allclose(wc.sum(1).cpu().numpy(), 1.0, atol=2e-5)


def test_fold_bias_matches_numpy(eng, packed, golden, latents, states):
    """Per-frame bias folding (replaces signal.expand+cat, fc_z, fc_z_skips, fc_z_view) vs float64 numpy."""
    g = golden("g3_decoder")
    zs, za = latents
    P = {k: np.asarray(v, np.float64) for k, v in states["decoder"].items()}
    sig = g["sig_aud"][0].astype(np.float64)
    bias = packed["f32"].fold_single(0, g["sig_aud"][0], zs[0, 0], za[0, 0]).cpu().numpy()

    def unperm(vec):     # blob order [tile][half][16] -> natural feature order
        out = np.zeros_like(vec)
        for e in range(vec.size):
            tt, h, r = e >> 5, (e >> 4) & 1, e & 15
            out[32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h] = vec[e]
        return out
    b_in = P["fc_in.bias"] + P["fc_in.weight"][:, 60:] @ sig + P["fc_z.weight"] @ zs[0, 0] + P["fc_z.bias"]
    np.testing.assert_allclose(unperm(bias[:256]), b_in, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(unperm(bias[256:512]), P["blocks.0.bias"], atol=0, rtol=0)
    b_sk = (P["fc_z_skips.0.weight"] @ zs[0, 0] + P["fc_z_skips.0.bias"] + P["fc_p_skips.0.bias"] +
            P["fc_p_skips.0.weight"][:, 60:] @ sig)
    np.testing.assert_allclose(unperm(bias[5 * 256:6 * 256]), b_sk, atol=2e-5, rtol=1e-5)
    view = unperm(bias[9 * 256:9 * 256 + 288])
    b_v = P["feat_view.bias"] + P["fc_z_view.weight"] @ za[0, 0] + P["fc_z_view.bias"] + P["fc_view.bias"]
    np.testing.assert_allclose(view[:256], b_v, atol=2e-5, rtol=1e-5)
    assert view[256] == np.float32(P["sigma_out.bias"][0]) and (view[257:] == 0).all()


@pytest.mark.parametrize("tier,n_fine,fields", [("f32", 0, 2), ("bf16", 128, 2), ("bf16", 64, 1)])
def test_render_u8_epilogue_equals_to8b_of_float_render(eng, packed, scene, latents, golden, tier, n_fine, fields):
    """dfn_render_fwd_u8 (to8b fused into the kernel epilogue) == to8b(dfn_render_fwd), byte for
    byte, on ragged ray counts; and against the reference's uint8 golden image (G7) for the coarse f32 case."""
    g = golden("g7_frame_coarse")
    zs, za = latents
    pk = packed[tier]
    bias = pk.fold(g["signal"][0], g["signal_torso"].reshape(-1) if fields == 2 else None, zs[0], za[0])
    bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3).cuda()
    for begin, count in ((0, 1003), (101250, 517)):
        fr = eng.make_frame(scene["H"], scene["W"], scene["focal"], scene["cx"], scene["cy"], scene["poses"][2],
                            scene["pose_body"], scene["near"], scene["far"], ray_begin=begin, ray_count=count,
                            n_fine=n_fine, fields=fields)
        f_h, f_c = eng.render(pk, bias, fr, bg)[:2]
        u_h, u_c = eng.render_u8(pk, bias, fr, bg)
        assert u_h.dtype == torch.uint8 and tuple(u_h.shape) == (count, 3)
        assert torch.equal(u_h, eng.to8b(f_h))
        if fields == 2:
            assert torch.equal(u_c, eng.to8b(f_c))
        else:
            assert u_c is None
    if tier == "f32" and n_fine == 0:
        idx = t(np.asarray(g["ray_idx"], np.int32)).cuda()
        fr = eng.make_frame(scene["H"], scene["W"], scene["focal"], scene["cx"], scene["cy"], scene["poses"][2],
                            scene["pose_body"], scene["near"], scene["far"], ray_count=len(g["ray_idx"]), n_fine=0,
                            fields=2)
        u_h, u_c = eng.render_u8(pk, bias, fr, bg, pix_index=idx)
        for got, ref in ((u_h, g["rgb8_head"]), (u_c, g["rgb8_com"])):
            d = np.abs(got.cpu().numpy().astype(np.int32) - ref.astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() <= 2e-3      # identical up to +-1 LSB on <= 0.1-0.2 %


def test_signal_encoders_hip_vs_reference_golden(eng, states, scene, golden):
    """dfn_encode_signal / dfn_encode_signal_torso against golden G6 (the reference's
    encode_signal / encode_signal_torso: both branches, zero-padded windows at both ends, the shortened sequence) and
    against the torch twins on every frame."""
    from dfanerf import nets
    g = golden("g6_signals")
    dev = torch.device("cuda")
    mods = {"AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in states[k].items()})
        m.to(dev)
    auds, exps, poses = [t(scene[k]).to(dev) for k in ("aud", "exp", "poses")]
    n = auds.shape[0]
    enc = eng.SignalEncoder(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"], auds, exps, poses)
    ids = [0, 1, 4, n - 1]
    for tag, smo, smo_t in (("raw", 0, 0), ("smo", 4, 8)):
        sig, sigt = enc.encode(ids, smo, smo_t)
        for b, i in enumerate(ids):
            np.testing.assert_allclose(sig[b].cpu().numpy(), g[f"aud_{tag}_{i}"].reshape(-1), rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(sigt[b].cpu().numpy(), g[f"torso_{tag}_{i}"].reshape(-1), rtol=2e-5, atol=2e-6)
    sig, _ = enc.encode([5], 4, 8, length=6)
    np.testing.assert_allclose(sig[0].cpu().numpy(), g["aud_smo_5_len6"].reshape(-1), rtol=2e-5, atol=2e-6)
    # every frame against the torch modules (batched twin)
    ds = [{"auds": auds, "exp": exps, "poses": poses}]
    embed_fn, _ = nets.get_embedder(3, 0)

    class A:
        nosmo_iters, smo_size, smo_torse_size = 300000, 4, 8
    for step, smo, smo_t in ((0, 0, 0), (300000, 4, 8)):
        with torch.no_grad():
            rs, rt = nets.encode_signals_batch(ds, 0, range(n), mods["AudNet"], mods["ExpNet"], mods["AudAttNet"],
                                               mods["PoseAttNet"], step, A, n, embed_fn)
        sig, sigt = enc.encode(range(n), smo, smo_t)
        torch.testing.assert_close(sig, rs, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sigt, rt, rtol=1e-4, atol=1e-5)
    # error paths
    with pytest.raises(Exception):
        enc.encode([0], 3, 8)


@pytest.mark.parametrize("fields", [1, 2])
def test_render_hierarchical_64_fine_vs_oracle(eng, packed, scene, latents, golden, states, fields):
    """Row H with n_fine = 64 (the other size the boundary accepts), f32 tier: the merged depths against the oracle's
    own sampler (within one coarse bin, sorted, end points), weights summing to one, and decoder + compositing at the
    kernel's depths against the oracle at those depths (stage (3) of the 64+128 test)."""
    gc = golden("g7_frame_coarse")
    idx = np.arange(5, scene["H"] * scene["W"], 1571)[:96]
    out = _render_subset(eng, packed, scene, latents, gc["signal"][0], gc["signal_torso"].reshape(-1), idx, "f32", 64,
                         fields, want_weights=True, want_z=True)
    rh, rc, wh, wc, z = [None if o is None else o.cpu().numpy() for o in out]
    assert z.shape == (96, 128) and (np.diff(z, axis=1) >= 0).all()
    assert np.allclose(z[:, 0], 0.3) and np.allclose(z[:, -1], 0.9)
    np.testing.assert_allclose(wh.sum(1), 1.0, atol=2e-6)
    P = O.params_to_torch(states["decoder"])
    zs, za = [t(v) for v in latents]
    H, W = scene["H"], scene["W"]
    o_h, d_h = O.get_rays(H, W, scene["focal"], scene["poses"][2][:3, :4], scene["cx"], scene["cy"])
    o_t, d_t = O.get_rays(H, W, scene["focal"], scene["pose_body"][:3, :4], scene["cx"], scene["cy"])
    rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h, o_t, d_t)]
    bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3)[idx]
    sig, sigt = [t(gc["signal"]), None], t(gc["signal_torso"])
    with torch.no_grad():
        _, _, aux = O.render_rays_chunk(P, *rays, bg, scene["near"], scene["far"], zs, za, sig, sigt, 64, 64, fields,
                                        return_aux=True)
        assert float((t(z) - aux["z_all"]).abs().max()) <= (0.6 / 63) * 1.001
        oh, oc = O.render_fixed_samples(P, *rays, bg, t(z), zs, za, sig, sigt, fields)
    np.testing.assert_allclose(rh, oh.numpy(), atol=5e-5, rtol=0)
    if fields == 2:
        np.testing.assert_allclose(wc.sum(1), 1.0, atol=2e-6)
        np.testing.assert_allclose(rc, oc.numpy(), atol=5e-5, rtol=0)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_render_coarse_f32_random_geometry_vs_oracle(eng, packed, states, latents, seed):
    """The live renderer (coarse, both fields) on random geometry the fixed scene does not cover: non-square images,
    other focal lengths / principal points, near / far like the dataset config files (0.4 / 1.0), random poses, a uint8
    background, last_dist and concate_bg variations - against the oracle on the same rays (f32 tier, 2e-5)."""
    rng = np.random.RandomState(seed)
    H, W = [(48, 64), (97, 51), (450, 450)][seed - 1]
    focal = float(rng.uniform(300, 1500))
    cx, cy = float(W / 2 + rng.uniform(-5, 5)), float(H / 2 + rng.uniform(-5, 5))
    near, far = [(0.4, 1.0), (0.25, 0.8), (0.3, 0.9)][seed - 1]

    def pose():
        a = rng.uniform(-0.2, 0.2, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
        P = np.eye(4, dtype=np.float32)
        P[:3, :3] = (Rz @ Ry @ Rx).astype(np.float32)
        P[:3, 3] = np.array([rng.uniform(-.05, .05), rng.uniform(-.05, .05), rng.uniform(0.5, 0.7)], np.float32)
        return P
    pose_h, pose_b = pose(), pose()
    bg_u8 = rng.randint(0, 256, size=(H * W, 3)).astype(np.uint8)
    sig = (rng.randn(96) * 0.5).astype(np.float32)
    sigt = (rng.randn(42) * 0.5).astype(np.float32)
    last_dist = [1e10, 0.05, 1e10][seed - 1]
    cbg = seed != 2
    NC = [64, 32, 128][seed - 1]              # --N_samples (round 6: 32 and 128 next to the scripts' 64)
    n = min(H * W, 257)
    idx = np.sort(rng.choice(H * W, n, replace=False)).astype(np.int32)
    zs, za = latents
    pk = packed["f32"]
    bias = pk.fold(sig, sigt, zs[0], za[0])
    fr = eng.make_frame(H, W, focal, cx, cy, pose_h, pose_b, near, far, last_dist=last_dist, ray_count=n, n_fine=0,
                        fields=2, concate_bg=cbg, n_coarse=NC)
    rh, rc, wh, wc = eng.render(pk, bias, fr, t(bg_u8).cuda(), pix_index=t(idx).cuda(), want_weights=True)
    P = O.params_to_torch(states["decoder"])
    o_h, d_h = O.get_rays(H, W, focal, pose_h[:3, :4], cx, cy)
    o_t, d_t = O.get_rays(H, W, focal, pose_b[:3, :4], cx, cy)
    rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h, o_t, d_t)]
    bg = t(bg_u8).float() / 255.0
    with torch.no_grad():
        z = O.coarse_z(near, far, NC)[None, :].expand(n, NC)
        s_h, f_h, s_t, f_t = O._eval_fields(P, *rays, z, t(zs), t(za), [t(sig)[None], None], t(sigt)[None], 2)
        oh, owh, oc, owc = O.integrate_fields(z, rays[1], rays[3], s_h, f_h, s_t, f_t, bg[idx], last_dist, cbg)
    np.testing.assert_allclose(rh.cpu().numpy(), oh.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(rc.cpu().numpy(), oc.numpy(), atol=2e-5, rtol=0)
    np.testing.assert_allclose(wh.cpu().numpy(), owh.numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(wc.cpu().numpy(), owc.numpy(), atol=2e-6, rtol=0)
