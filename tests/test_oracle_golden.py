"""Pin the CPU oracle (oracle/dfa_oracle.py) against vectors produced by the
imported reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import torch

import dfa_oracle as O
from dfanerf import synth

torch.set_num_threads(8)


def t(x):
    return torch.from_numpy(np.asarray(x))


def test_g1_get_rays_bitwise(golden):
    g = golden("g1_rays")
    H, W, focal, cx, cy = g["hwfcxy"]
    H, W = int(H), int(W)
    idx = g["idx"]
    for tag in ("a", "b"):
        ro, rd = O.get_rays(H, W, focal, g["pose_" + tag][:3, :4], cx, cy)
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        assert np.array_equal(rd[idx].numpy(), g["rays_d_" + tag])     # bit-for-bit
        assert np.array_equal(ro[idx].numpy(), g["rays_o_" + tag])
        assert np.array_equal(rd.double().sum(0).numpy(), g["sum_d_" + tag])   # all 202,500 rays
        assert np.array_equal((rd.double() ** 2).sum(0).numpy(), g["sumsq_d_" + tag])
        no, nd = O.ndc_rays(H, W, focal, 1.0, ro, rd)
        np.testing.assert_allclose(no[idx].numpy(), g["ndc_o_" + tag], rtol=0, atol=0)
        np.testing.assert_allclose(nd[idx].numpy(), g["ndc_d_" + tag], rtol=0, atol=0)
    sc = synth.bench_scene(0)
    _, rd = O.get_rays(8, 6, 100.0, sc["poses"][1][:3, :4])
    assert np.array_equal(rd.numpy(), g["small_rays_d"])


def test_g2_zvals_bitwise(golden):
    g = golden("g2_zvals")
    for n in (64, 128, 192):
        assert np.array_equal(O.linspace01(n).numpy(), g[f"t{n}"])
    for tag in "abc":
        near, far = g["nf_" + tag]
        assert np.array_equal(O.coarse_z(float(near), float(far), 64).numpy(), g["z_" + tag])


def test_g3_decoder(golden, states, latents):
    g = golden("g3_decoder")
    P = O.params_to_torch(states["decoder"])
    zs, za = [t(v) for v in latents]
    for S in (64, 192):
        p, r = t(g[f"p_{S}"]), t(g[f"r_{S}"])
        with torch.no_grad():
            fh, sh = O.decoder_forward(P, p, r, zs[:, 0], za[:, 0], [t(g["sig_aud"]), None], 'head')
            ft, st = O.decoder_forward(P, p, r, zs[:, 1], za[:, 1], t(g["sig_torso"]), 'torso')
            fl, sl = O.decoder_forward(P, p, r, zs[:, 0], za[:, 0], [None, None], 'head')
        for got, ref in ((fh, "feat_head"), (sh, "sigma_head"), (ft, "feat_torso"), (st, "sigma_torso"),
                         (fl, "feat_listener"), (sl, "sigma_listener")):
            np.testing.assert_allclose(got.numpy(), g[f"{ref}_{S}"], rtol=1e-5, atol=2e-5)
    # positional encoding is elementwise: bitwise
    assert np.array_equal(O.posenc(t(g["p_64"][:, :8]), 10).numpy(), g["pe_p"])
    assert np.array_equal(O.posenc(t(g["p_big"]), 10).numpy(), g["pe_big"])
    d = t(g["r_64"][:, :8])
    assert np.array_equal(O.posenc(d / torch.norm(d, dim=-1, keepdim=True), 4).numpy(), g["pe_v"])


def test_g1c_get_rays_stride_bitwise(golden):
    """get_rays' `stride` argument (HELP:449-451: non-integer pixel positions from torch.linspace; dead in the driver), all rays bitwise"""
    g = golden("g1c_rays_stride")
    H, W, focal, cx, cy = g["hwfcxy"]
    for s in (2, 3, 7):
        ro, rd = O.get_rays(int(H), int(W), focal, g["pose"][:3, :4], cx, cy, stride=s)
        assert tuple(rd.shape[:2]) == tuple(g[f"shape_{s}"]) == (int(H) // s, int(W) // s)
        idx = g[f"idx_{s}"]
        assert np.array_equal(rd.reshape(-1, 3)[idx].numpy(), g[f"rays_d_{s}"]) and np.array_equal(ro.reshape(-1, 3)[idx].numpy(), g[f"rays_o_{s}"])
        assert np.array_equal(rd.double().sum((0, 1)).numpy(), g[f"sum_d_{s}"])


def test_g16_z_dim_64(golden):
    """--z_dim is free upstream (MAIN:372, 518): the oracle's decoder with 64-wide latent codes against the reference's (G16)"""
    g, g3 = golden("g16_z_dim_64"), golden("g3_decoder")
    P = O.params_to_torch(synth.synth_decoder_state(0, z_dim=64))
    zs, za = [t(v) for v in synth.synth_latents(0, z_dim=64)]
    p, r = t(g3["p_64"]), t(g3["r_64"])
    with torch.no_grad():
        out = {"head": O.decoder_forward(P, p, r, zs[:, 0], za[:, 0], [t(g3["sig_aud"]), None], 'head'),
               "torso": O.decoder_forward(P, p, r, zs[:, 1], za[:, 1], t(g3["sig_torso"]), 'torso'),
               "listener": O.decoder_forward(P, p, r, zs[:, 0], za[:, 0], [None, None], 'head')}
    for k, (f, s) in out.items():
        np.testing.assert_allclose(f.numpy(), g["feat_" + k], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(s.numpy(), g["sigma_" + k], rtol=1e-5, atol=2e-5)


def test_g18_n_feat_128(golden):
    """--n_feat is free upstream (MAIN:374, 518): the oracle's decoder at hidden width 128, latent width 64 against the reference's (G18)"""
    g, g3 = golden("g18_n_feat_128"), golden("g3_decoder")
    P = O.params_to_torch(synth.synth_decoder_state(0, z_dim=64, hidden=128))
    zs, za = [t(v) for v in synth.synth_latents(0, z_dim=64)]
    p, r = t(g3["p_64"]), t(g3["r_64"])
    with torch.no_grad():
        out = {"head": O.decoder_forward(P, p, r, zs[:, 0], za[:, 0], [t(g3["sig_aud"]), None], 'head'),
               "torso": O.decoder_forward(P, p, r, zs[:, 1], za[:, 1], t(g3["sig_torso"]), 'torso'),
               "listener": O.decoder_forward(P, p, r, zs[:, 0], za[:, 0], [None, None], 'head')}
    for k, (f, s) in out.items():
        np.testing.assert_allclose(f.numpy(), g["feat_" + k], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(s.numpy(), g["sigma_" + k], rtol=1e-5, atol=2e-5)


def test_g17_no_deformation_field(golden, states, latents):
    """without --use_deformation_field (MAIN:411) the torso is the plain MLP on [PE, pose signal] (DEC:297 skipped): the oracle
    without the deform_net tensors against the reference's Decoder(use_deformation_field=False)"""
    g, g3 = golden("g17_no_deformation_field"), golden("g3_decoder")
    P = O.params_to_torch({k: v for k, v in states["decoder"].items() if not k.startswith("deform_net.")})
    zs, za = [t(v) for v in latents]
    with torch.no_grad():
        f, s = O.decoder_forward(P, t(g3["p_64"]), t(g3["r_64"]), zs[:, 1], za[:, 1], t(g3["sig_torso"]), 'torso')
    np.testing.assert_allclose(f.numpy(), g["feat_torso"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(s.numpy(), g["sigma_torso"], rtol=1e-5, atol=2e-5)


def test_g14_listener_backward(golden, states, latents):
    """the listener input layers (signal None: decoder.py:306-307, 322-323) under autograd: the oracle's gradients against the
    reference module's (G14) - which parameters get one, their norms, sampled entries, the two listener matrices in full"""
    g, g3 = golden("g14_listener_backward"), golden("g3_decoder")
    P = {k: v.clone().requires_grad_(True) for k, v in O.params_to_torch(states["decoder"]).items()}
    zs, za = [t(v) for v in latents]
    feat, sigma = O.decoder_forward(P, t(g3["p_64"]), t(g3["r_64"]), zs[:, 0], za[:, 0], [None, None], 'head')
    loss = (feat * t(g["w_f"])).sum() + (sigma * t(g["w_s"])).sum()
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    loss.backward()
    touched = open(os.path.join(os.path.dirname(__file__), "golden", "g14_listener_touched.txt")).read().split()
    assert sorted(k for k, v in P.items() if v.grad is not None and float(v.grad.abs().max()) > 0) == sorted(touched)
    for k in touched:
        gr = P[k].grad.reshape(-1)
        ref = float(g["gnorm/" + k])
        assert abs(float(gr.double().norm()) - ref) <= 1e-4 * ref, k
        np.testing.assert_allclose(gr[:: max(1, gr.numel() // 8)][:8].numpy(), g["gsamp/" + k], rtol=2e-3, atol=1e-5 * ref)
    for k in ("fc_in_listener.weight", "fc_p_skips_listener.0.weight"):
        d = (P[k].grad - t(g["gfull/" + k])).double().norm() / float(g["gnorm/" + k])
        assert float(d) < 1e-5, (k, float(d))


def test_g4_composite_weights(golden):
    g = golden("g4_composite")
    sig, feat = t(g["sigma"]), t(g["feat"])
    s2, f2 = O.composite_function(sig, feat)
    s1, f1 = O.composite_function(sig[:1], feat[:1])
    assert np.array_equal(s2.numpy(), g["sigma_sum2"]) and np.array_equal(f2.numpy(), g["feat2"])
    assert np.array_equal(s1.numpy(), g["sigma_sum1"]) and np.array_equal(f1.numpy(), g["feat1"])
    z, ray = t(g["z"]), t(g["ray"])
    assert np.array_equal(O.calc_volume_weights(z, ray, s2).numpy(), g["w2"])
    assert np.array_equal(O.calc_volume_weights(z, ray, s1).numpy(), g["w1"])
    assert np.array_equal(O.calc_volume_weights(z, ray, s1, last_dist=0.05).numpy(), g["w_lastdist005"])


def test_g5_sample_pdf(golden):
    g = golden("g5_sample_pdf")
    bins, w = t(g["bins"]), t(g["weights"])
    assert np.array_equal(O.sample_pdf(bins, w, 128, det=True).numpy(), g["det128"])
    assert np.array_equal(O.sample_pdf(bins, w, 16, det=True).numpy(), g["det16"])
    assert np.array_equal(O.sample_pdf(bins, w, 128, u=t(g["u_pytest"])).numpy(), g["pytest128"])


def test_g6_signals(golden, states, scene):
    g = golden("g6_signals")
    nets = {k: O.params_to_torch(v) for k, v in states.items() if k != "decoder"}
    auds, exps, poses = t(scene["aud"]), t(scene["exp"]), t(scene["poses"])
    n = auds.shape[0]
    with torch.no_grad():
        for i in (0, 1, 4, n - 1):
            for step, tag in ((0, "raw"), (300000, "smo")):
                s = O.encode_signal(nets, auds, exps, i, step, 300000, 4, n)
                assert s[1] is None and s[0].shape == (1, 96)
                np.testing.assert_allclose(s[0].numpy(), g[f"aud_{tag}_{i}"], rtol=1e-5, atol=1e-6)
                st = O.encode_signal_torso(nets, poses, i, step, 300000, 8, n)
                assert st.shape == g[f"torso_{tag}_{i}"].shape      # [1,42] vs [42] quirk
                np.testing.assert_allclose(st.numpy(), g[f"torso_{tag}_{i}"], rtol=1e-5, atol=1e-6)
        s = O.encode_signal(nets, auds, exps, 5, 300000, 300000, 4, 6)
        np.testing.assert_allclose(s[0].numpy(), g["aud_smo_5_len6"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(O.pose_to_euler_trans(poses).numpy(), g["euler_trans"])


def _frame_inputs(states, scene, latents, g):
    P = O.params_to_torch(states["decoder"])
    zs, za = [t(v) for v in latents]
    H, W = scene["H"], scene["W"]
    f = int(g["frame"][0])
    o_h, d_h = O.get_rays(H, W, scene["focal"], scene["poses"][f][:3, :4], scene["cx"], scene["cy"])
    o_t, d_t = O.get_rays(H, W, scene["focal"], scene["pose_body"][:3, :4], scene["cx"], scene["cy"])
    idx = g["ray_idx"]
    bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3)[idx]
    rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h, o_t, d_t)]
    return P, zs, za, rays, bg


def test_g7_frame_coarse(golden, states, scene, latents):
    g = golden("g7_frame_coarse")
    P, zs, za, rays, bg = _frame_inputs(states, scene, latents, g)
    with torch.no_grad():
        rh, rc, aux = O.render_rays_chunk(P, *rays, bg, 0.3, 0.9, zs, za, [t(g["signal"]), None],
                                          t(g["signal_torso"]), 64, 0, 2, return_aux=True)
    np.testing.assert_allclose(rh.numpy(), g["rgb_head"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(rc.numpy(), g["rgb_com"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(aux["w_head"][:8].numpy(), g["w_head_first8"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(aux["w_com"][:8].numpy(), g["w_com_first8"], atol=1e-6, rtol=0)
    for got, ref in ((rh, "rgb8_head"), (rc, "rgb8_com")):
        d = np.abs(O.to8b(got.numpy()).astype(int) - g[ref].astype(int))
        assert d.max() <= 1 and (d > 0).mean() <= 1e-3


def test_g15_coarse_other_sample_counts(golden, states, scene, latents):
    """--N_samples 32 and 128 (MAIN:612-619: the flag is free upstream), coarse only: the oracle's loop against the reference's"""
    g = golden("g15_coarse_nsamples")
    P, zs, za, rays, bg = _frame_inputs(states, scene, latents, g)
    for S in (32, 128):
        with torch.no_grad():
            rh, rc, aux = O.render_rays_chunk(P, *rays, bg, 0.3, 0.9, zs, za, [t(g["signal"]), None], t(g["signal_torso"]), S, 0, 2,
                                              return_aux=True)
        assert np.array_equal(O.coarse_z(0.3, 0.9, S).numpy(), g[f"z_{S}"])
        np.testing.assert_allclose(rh.numpy(), g[f"rgb_head_{S}"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(rc.numpy(), g[f"rgb_com_{S}"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(aux["w_head"].numpy(), g[f"w_head_{S}"], atol=1e-6, rtol=0)
        np.testing.assert_allclose(aux["w_com"].numpy(), g[f"w_com_{S}"], atol=1e-6, rtol=0)


def test_g7_frame_hier(golden, states, scene, latents):
    g = golden("g7_frame_hier")
    gc = golden("g7_frame_coarse")
    P, zs, za, rays, bg = _frame_inputs(states, scene, latents, g)
    for fields in (1, 2):
        with torch.no_grad():
            rh, rc, aux = O.render_rays_chunk(P, *rays, bg, 0.3, 0.9, zs, za, [t(gc["signal"]), None],
                                              t(gc["signal_torso"]), 64, 128, 2 if fields == 2 else 1,
                                              return_aux=True)
        np.testing.assert_allclose(aux["z_all"].numpy(), g[f"z_all_f{fields}"], atol=2e-6, rtol=0)
        np.testing.assert_allclose(rh.numpy(), g[f"rgb_head_f{fields}"], atol=1e-5, rtol=0)
        if fields == 2:
            np.testing.assert_allclose(rc.numpy(), g[f"rgb_com_f{fields}"], atol=1e-5, rtol=0)
        z = aux["z_all"].numpy()
        assert (np.diff(z, axis=1) >= 0).all() and np.allclose(z[:, -1], 0.9) and np.allclose(z[:, 0], 0.3)


def test_g8_train_step(golden, states, scene, latents):
    """Loss, gradient norms and sampled gradient entries of one training step for the three
    optimizer-gating regimes, then the Adam update with the reference's gating (MAIN:924-931)."""
    g = golden("g8_train_step")
    H, W = scene["H"], scene["W"]
    sel = t(g["sel_yx"])
    tgt_h = t(synth.synth_tensor(0, "g8/th", (H, W, 3), 0.5)) + 0.5
    tgt_c = t(synth.synth_tensor(0, "g8/tc", (H, W, 3), 0.5)) + 0.5
    bg = t(scene["bg"]).float() / 255.0
    zs, za = [t(v) for v in latents]
    auds, exps, poses = t(scene["aud"]), t(scene["exp"]), t(scene["poses"])
    for step in (0, 300000, 400000):
        allp = {tag: {k: t(v).clone().requires_grad_(True) for k, v in st.items()}
                for tag, st in states.items()}
        nets = {k: v for k, v in allp.items() if k != "decoder"}
        loss, lh, lc = O.train_loss(allp["decoder"], nets, sel, H, W, scene["focal"], scene["cx"],
                                    scene["cy"], poses[3], poses[0], bg, tgt_h, tgt_c, 0.3, 0.9, zs, za,
                                    auds, exps, poses, 3, step, 300000, 4, 8, auds.shape[0])
        np.testing.assert_allclose([loss.item(), lh.item(), lc.item()], g[f"loss_{step}"], rtol=2e-6)
        loss.backward()
        for tag, prm in allp.items():
            for k, v in prm.items():
                ref = float(g[f"gnorm_{step}/{tag}/{k}"])
                if ref < 0:
                    assert v.grad is None or float(v.grad.abs().sum()) == 0
                    continue
                got = 0.0 if v.grad is None else v.grad.double().norm().item()
                assert abs(got - ref) <= 2e-4 * ref + 1e-9, (step, tag, k, got, ref)
                if ref > 0:
                    gs = v.grad.reshape(-1)
                    samp = gs[:: max(1, gs.numel() // 8)][:8].numpy()
                    np.testing.assert_allclose(samp, g[f"gsamp_{step}/{tag}/{k}"], rtol=5e-3,
                                               atol=2e-4 * ref / np.sqrt(gs.numel()) + 1e-9)
        # Adam with the reference gating
        gate = {"decoder": True, "AudNet": True, "ExpNet": step >= 400000,
                "AudAttNet": step >= 300000, "PoseAttNet": step >= 300000}
        for tag, prm in allp.items():
            ps = [v for v in prm.values()]
            for v in ps:
                if v.grad is None:
                    v.grad = torch.zeros_like(v)
            opt = torch.optim.Adam(ps, lr=5e-4, betas=(0.9, 0.999))
            if gate[tag]:
                opt.step()
        for key in [k for k in g if k.startswith(f"after_{step}/")]:
            _, tag, name = key.split("/", 2)
            got = allp[tag][name].detach().reshape(-1)[: g[key].size].numpy()
            np.testing.assert_allclose(got, g[key].reshape(-1), rtol=0, atol=2e-6)


def test_g9_manifest(states):
    import os
    from conftest import GOLDEN
    lines = open(os.path.join(GOLDEN, "g9_manifest.txt")).read().strip().split("\n")
    want = {}
    for ln in lines:
        parts = ln.split(" ", 2)
        if parts[0] in states:
            want.setdefault(parts[0], []).append((parts[1], eval(parts[2])))
    for tag, items in want.items():
        got = [(k, tuple(v.shape)) for k, v in states[tag].items()]
        assert got == items, tag          # same keys, same order, same shapes as the reference modules
    n = sum(int(np.prod(v.shape)) for v in states["decoder"].values())
    assert f"n_params decoder {n}" in lines


def test_g10_to8b_psnr(golden):
    g = golden("g10_to8b")
    assert np.array_equal(O.to8b(g["x"]), g["y"])
    np.testing.assert_allclose(O.mse2psnr(t(g["mse"])).numpy(), g["psnr"], rtol=1e-6)
