#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's own Python modules on CPU.

Runs only in the build container (needs /root/reference, which does not exist on
the GPU box).  Outputs are data only: inputs + the reference's outputs.  Weights
are not stored; they are regenerated from dfanerf.synth (closed-form hash) and
loaded INTO the reference modules with load_state_dict.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Missing host-only modules (imageio, cv2, configargparse) are stubbed with empty
modules: they are touched only by file I/O / CLI code, never by the numerics.
rot_to_euler hard-codes .cuda() (run_nerf_com_trainExpLater.py:184), so
Tensor.cuda is patched to identity for the CPU import.
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/NeRFs/DFANeRF"
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
for m in ("imageio", "cv2", "configargparse"):
    if m not in sys.modules:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = types.ModuleType(m)
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self

import run_nerf_helpers as HELP          # noqa: E402  (reference)
import decoder as DEC                    # noqa: E402  (reference)
import run_nerf_com_trainExpLater as MAIN  # noqa: E402  (reference)
torch.autograd.set_detect_anomaly(False)

from dfanerf import synth                # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def t(x):
    return torch.from_numpy(np.asarray(x))


def build_ref_modules(seed=0):
    st = synth.synth_all_states(seed)
    dec = DEC.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True,
                      use_expression=False, use_aud_net=False)
    dec.load_state_dict({k: t(v) for k, v in st["decoder"].items()})
    aud = HELP.AudioNet_W2L()
    aud.load_state_dict({k: t(v) for k, v in st["AudNet"].items()})
    exp = HELP.ExpressionEnc()
    exp.load_state_dict({k: t(v) for k, v in st["ExpNet"].items()})
    att = HELP.AudioAttNet(dim_aud=96, seq_len=4)
    att.load_state_dict({k: t(v) for k, v in st["AudAttNet"].items()})
    patt = HELP.AudioAttNet(dim_aud=42, seq_len=8)
    patt.load_state_dict({k: t(v) for k, v in st["PoseAttNet"].items()})
    return dec, aud, exp, att, patt


ONLY = None      # name of the one fixture to (re)write; None: all of them


def save(name, **arrs):
    if ONLY is not None and name != ONLY:
        return
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB", {k: v.shape for k, v in out.items()})


class Args:
    pass


def main():
    sc = synth.bench_scene(0, n_frames=8)
    H, W, focal, cx, cy = sc["H"], sc["W"], sc["focal"], sc["cx"], sc["cy"]
    dec, audnet, expnet, attnet, pattnet = build_ref_modules(0)
    z_shape, z_app = [t(v) for v in synth.synth_latents(0)]

    # ---- G1: get_rays / ndc_rays -------------------------------------------------
    rng = np.random.RandomState(1)
    idx = np.unique(np.concatenate([
        [0, W - 1, (H - 1) * W, H * W - 1, 224 * W + 224, 225 * W + 225],
        np.arange(0, W, 37), np.arange(0, H, 41) * W, rng.randint(0, H * W, 200)]))[:256]
    g1 = {"idx": idx.astype(np.int64), "hwfcxy": np.array([H, W, focal, cx, cy], np.float64)}
    for name, pose in (("a", sc["poses"][0]), ("b", sc["poses"][5])):
        ro, rd = HELP.get_rays(H, W, focal, t(pose)[:3, :4], cx, cy)
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        g1["pose_" + name] = pose
        g1["rays_o_" + name] = ro[idx]
        g1["rays_d_" + name] = rd[idx]
        g1["sum_d_" + name] = rd.double().sum(0).numpy()
        g1["sumsq_d_" + name] = (rd.double() ** 2).sum(0).numpy()
        no, nd = HELP.ndc_rays(H, W, focal, 1.0, ro, rd)
        g1["ndc_o_" + name] = no[idx]
        g1["ndc_d_" + name] = nd[idx]
    # stride path and default cx/cy
    ro, rd = HELP.get_rays(8, 6, 100.0, t(sc["poses"][1])[:3, :4])
    g1["small_rays_d"] = rd
    save("g1_rays", **g1)
    # G1c (round 6): get_rays' `stride` argument (HELP:449-451; dead in the driver): 225 x 225 and 150 x 150 rays, 3 and 7 with sizes
    # that do not divide (450 // 7 = 64)
    g1c = {"hwfcxy": np.array([H, W, focal, cx, cy], np.float64), "pose": sc["poses"][5]}
    for stride in (2, 3, 7):
        ro, rd = HELP.get_rays(H, W, focal, t(sc["poses"][5])[:3, :4], cx, cy, stride=stride)
        n = rd.shape[0] * rd.shape[1]
        pick = np.unique(np.concatenate([[0, rd.shape[1] - 1, n - rd.shape[1], n - 1], np.random.RandomState(stride).randint(0, n, 250)]))
        g1c[f"idx_{stride}"] = pick.astype(np.int64)
        g1c[f"shape_{stride}"] = np.array(rd.shape[:2])
        g1c[f"rays_d_{stride}"] = rd.reshape(-1, 3)[pick]
        g1c[f"rays_o_{stride}"] = ro.reshape(-1, 3)[pick]
        g1c[f"sum_d_{stride}"] = rd.double().sum((0, 1))
    save("g1c_rays_stride", **g1c)

    # ---- G2: z_vals --------------------------------------------------------------
    g2 = {}
    for tag, (near, far) in (("a", (0.3, 0.9)), ("b", (0.4, 1.0)), ("c", (0.3127, 0.9127))):
        near_t, far_t = near * torch.ones((5, 1)), far * torch.ones((5, 1))
        tv = torch.linspace(0., 1., steps=64)
        z = near_t * (1. - tv) + far_t * tv
        g2["nf_" + tag] = np.array([near, far])
        g2["z_" + tag] = z[0]
    g2["t64"] = torch.linspace(0., 1., steps=64)
    g2["t128"] = torch.linspace(0., 1., steps=128)
    g2["t192"] = torch.linspace(0., 1., steps=192)
    save("g2_zvals", **g2)

    # ---- G3: decoder forward, head & torso ---------------------------------------
    sig_aud = t(synth.synth_tensor(0, "g3/sig", (1, 96), 0.8))
    sig_torso = t(synth.synth_tensor(0, "g3/sigt", (1, 42), 0.8))
    g3 = {"sig_aud": sig_aud, "sig_torso": sig_torso}
    ro, rd = HELP.get_rays(H, W, focal, t(sc["poses"][0])[:3, :4], cx, cy)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    rays = np.array([0, 101234, 150000, H * W - 1])
    for S in (64, 192):
        tv = torch.linspace(0., 1., steps=S)
        z = 0.3 * (1. - tv) + 0.9 * tv
        p = (ro[rays, None, :] + rd[rays, None, :] * z[None, :, None]).reshape(1, -1, 3)
        r = rd[rays, None, :].expand(len(rays), S, 3).reshape(1, -1, 3)
        with torch.no_grad():
            fh, sh = dec(p, r, z_shape[:, 0], z_app[:, 0], [sig_aud, None], 'head')
            ft, st_ = dec(p, r, z_shape[:, 1], z_app[:, 1], sig_torso, 'torso')
            ft1, st1 = dec(p, r, z_shape[:, 1], z_app[:, 1], sig_torso[0], 'torso')  # 1-D signal
            fl, sl = dec(p, r, z_shape[:, 0], z_app[:, 0], [None, None], 'head')     # listener
        assert torch.equal(ft, ft1) and torch.equal(st_, st1)
        g3.update({f"p_{S}": p, f"r_{S}": r, f"feat_head_{S}": fh, f"sigma_head_{S}": sh,
                   f"feat_torso_{S}": ft, f"sigma_torso_{S}": st_,
                   f"feat_listener_{S}": fl, f"sigma_listener_{S}": sl})
    g3["pe_p"] = dec.transform_points(g3["p_64"][:, :8])
    g3["pe_v"] = dec.transform_points(g3["r_64"][:, :8] / torch.norm(g3["r_64"][:, :8], dim=-1, keepdim=True),
                                      views=True)
    # large-argument PE case (|p| ~ 1.6 -> args up to ~800 rad)
    pbig = t(synth.synth_tensor(0, "g3/pbig", (1, 64, 3), 1.6))
    g3["p_big"] = pbig
    g3["pe_big"] = dec.transform_points(pbig)
    save("g3_decoder", **g3)

    # ---- G4: composite + weights -------------------------------------------------
    C, S = 6, 64
    sig = torch.relu(t(synth.synth_tensor(0, "g4/sig", (2, 1, C, S), 20.0)))
    sig[:, :, 0, 10:20] = 0.0       # rows where both fields are empty -> 1e-4 denominator
    sig[1, :, 1, :] = 0.0
    feat = t(synth.synth_tensor(0, "g4/feat", (2, 1, C, S, 3), 0.5)) + 0.5
    ss2, ff2 = MAIN.composite_function(sig.clone(), feat.clone())
    ss1, ff1 = MAIN.composite_function(sig[:1].clone(), feat[:1].clone())
    zv = (0.3 * (1 - torch.linspace(0, 1, S)) + 0.9 * torch.linspace(0, 1, S))[None, None, :].expand(1, C, S)
    rv = t(synth.synth_tensor(0, "g4/rv", (1, C, 3), 1.0))
    w2 = MAIN.calc_volume_weights(zv, rv, ss2, last_dist=1e10)
    w1 = MAIN.calc_volume_weights(zv, rv, ss1, last_dist=1e10)
    wld = MAIN.calc_volume_weights(zv, rv, ss1, last_dist=0.05)
    save("g4_composite", sigma=sig, feat=feat, sigma_sum2=ss2, feat2=ff2, sigma_sum1=ss1, feat1=ff1,
         z=zv, ray=rv, w2=w2, w1=w1, w_lastdist005=wld)

    # ---- G5: sample_pdf -----------------------------------------------------------
    R, nb = 16, 63
    bins = .5 * (zv[0, 0, 1:] + zv[0, 0, :-1])[None, :].expand(R, nb).contiguous()
    wts = torch.relu(t(synth.synth_tensor(0, "g5/w", (R, nb - 1), 1.0)))
    wts[0] = 0.0                      # flat (all mass from the 1e-5 floor)
    wts[1] = 0.0
    wts[1, 30] = 1.0                  # delta ray
    wts[2, :] = 0.0
    wts[2, 0] = 1.0                   # mass in first bin
    wts[3, :] = 0.0
    wts[3, -1] = 1.0                  # mass in last bin
    s_det = HELP.sample_pdf(bins, wts, 128, det=True)
    s_py = HELP.sample_pdf(bins, wts, 128, det=False, pytest=True)
    np.random.seed(0)
    u_py = np.random.rand(R, 128).astype(np.float32)
    s_det16 = HELP.sample_pdf(bins, wts, 16, det=True)
    save("g5_sample_pdf", bins=bins, weights=wts, det128=s_det, pytest128=s_py, u_pytest=u_py, det16=s_det16)

    # ---- G6: signals ---------------------------------------------------------------
    auds, exps = t(sc["aud"]), t(sc["exp"])
    poses = t(sc["poses"])
    ds = [{"auds": auds, "exp": exps, "poses": poses}]
    a = Args()
    a.nosmo_iters, a.smo_size, a.smo_torse_size = 300000, 4, 8
    embed_fn, in_ch = HELP.get_embedder(3, 0)
    assert in_ch == 21
    g6 = {}
    n = auds.shape[0]
    with torch.no_grad():
        for i in (0, 1, 4, n - 1):
            for step, tag in ((0, "raw"), (300000, "smo")):
                s = MAIN.encode_signal(ds, 0, i, 96, audnet, expnet, attnet, step, a, n, embed_fn=embed_fn)
                assert s[1] is None
                g6[f"aud_{tag}_{i}"] = s[0]
                st_ = MAIN.encode_signal_torso(ds, 0, i, pattnet, step, a, n, embed_fn=embed_fn)
                g6[f"torso_{tag}_{i}"] = st_
        # len_auds shorter than the array (training passes len(i_train), MAIN:779)
        s = MAIN.encode_signal(ds, 0, 5, 96, audnet, expnet, attnet, 300000, a, 6, embed_fn=embed_fn)
        g6["aud_smo_5_len6"] = s[0]
    g6["euler_trans"] = MAIN.pose_to_euler_trans(poses)
    save("g6_signals", **g6)

    # ---- G7: end-to-end frame loop (MAIN:633-715 restated with the imported functions) ----
    bc_img = t(sc["bg"]).float() / 255.0
    a.N_samples, a.chunk, a.concate_bg, a.last_dist = 64, 2048, True, 1e10
    with torch.no_grad():
        signal = MAIN.encode_signal(ds, 0, 2, 96, audnet, expnet, attnet, 300000, a, n, embed_fn=embed_fn)
        signal_torso = MAIN.encode_signal_torso(ds, 0, 2, pattnet, 300000, a, n, embed_fn=embed_fn)
    pose, pose_body = t(sc["poses"][2]), t(sc["pose_body"])

    def ref_chunk(ray_idx, z_vals_c):
        """MAIN:653-709 for an explicit list of rays and given z [C,S] (S may be 192)."""
        Sx = z_vals_c.shape[1]
        ro, rd = HELP.get_rays(H, W, focal, pose[:3, :4], cx, cy)
        rot, rdt = HELP.get_rays(H, W, focal, pose_body[:3, :4], cx, cy)
        ro, rd, rot, rdt = [x.reshape(-1, 3)[ray_idx] for x in (ro, rd, rot, rdt)]
        p_i = (ro[..., None, :] + rd[..., None, :] * z_vals_c[..., :, None]).reshape(1, -1, 3)
        r_i = rd.unsqueeze(1).expand([len(ray_idx), Sx, 3]).reshape(1, -1, 3)
        p_t = (rot[..., None, :] + rdt[..., None, :] * z_vals_c[..., :, None]).reshape(1, -1, 3)
        r_t = rdt.unsqueeze(1).expand([len(ray_idx), Sx, 3]).reshape(1, -1, 3)
        feat_i, sigma_i = dec(p_i, r_i, z_shape[:, 0], z_app[:, 0], signal, 'head')
        sigma_i = sigma_i.reshape(1, -1, Sx)
        feat_i = feat_i.reshape(1, -1, Sx, 3)
        bc_rgb = bc_img.reshape(1, H * W, 1, 3)[:, ray_idx]
        feat_i = torch.cat((feat_i[..., :-1, :], bc_rgb), dim=-2)
        feat_t, sigma_t = dec(p_t, r_t, z_shape[:, 1], z_app[:, 1], signal_torso, 'torso')
        sigma_t = sigma_t.reshape(1, -1, Sx)
        feat_t = feat_t.reshape(1, -1, Sx, 3)
        sigma_t[:, :, -1] = 0
        sigma = torch.relu(torch.stack([sigma_i], 0))
        feat = torch.stack([feat_i], 0)
        sigma_to = torch.relu(torch.stack([sigma_i, sigma_t], 0))
        feat_to = torch.stack([feat_i, feat_t], 0)
        sigma[-1, :, :, -1] = sigma[-1, :, :, -1] + 1e-6
        sigma_to[-1, :, :, -1] = sigma_to[-1, :, :, -1] + 1e-6
        ssum, fw = MAIN.composite_function(sigma, feat)
        ssum_t, fw_t = MAIN.composite_function(sigma_to, feat_to)
        wts_h = MAIN.calc_volume_weights(z_vals_c[None], rd[None], ssum, last_dist=1e10)
        wts_c = MAIN.calc_volume_weights(z_vals_c[None], rdt[None], ssum_t, last_dist=1e10)
        rgb_h = torch.sum(wts_h.unsqueeze(-1) * fw, dim=-2)[0]
        rgb_c = torch.sum(wts_c.unsqueeze(-1) * fw_t, dim=-2)[0]
        return rgb_h, rgb_c, wts_h[0], wts_c[0], sigma_i[0], sigma_t[0]

    tv = torch.linspace(0., 1., steps=64)
    sub = np.arange(0, H * W, 97)            # 2088 rays strided over the frame
    with torch.no_grad():
        zc = (0.3 * (1. - tv) + 0.9 * tv)[None].expand(len(sub), 64)
        rgb_h, rgb_c, w_h, w_c, sg_h, sg_t = ref_chunk(sub, zc)
    print("G7 coverage: sigma_head relu max %.2f mean %.2f ; alpha mass on bg (head) mean %.3f" % (
        torch.relu(sg_h).max(), torch.relu(sg_h).mean(), w_h[:, -1].mean()))
    save("g7_frame_coarse", ray_idx=sub, frame=np.array([2]), rgb_head=rgb_h, rgb_com=rgb_c,
         w_head_first8=w_h[:8], w_com_first8=w_c[:8],
         rgb8_head=HELP.to8b(rgb_h.numpy()), rgb8_com=HELP.to8b(rgb_c.numpy()),
         signal=signal[0], signal_torso=signal_torso)

    # ---- G15 (round 6): the same loop with --N_samples 32 and 128, coarse only (MAIN:612-619: the flag is free upstream)
    sub15 = np.arange(5, H * W, 811)
    g15 = {"ray_idx": sub15, "frame": np.array([2]), "signal": signal[0], "signal_torso": signal_torso}
    for S15 in (32, 128):
        tv15 = torch.linspace(0., 1., steps=S15)
        with torch.no_grad():
            z15 = (0.3 * (1. - tv15) + 0.9 * tv15)[None].expand(len(sub15), S15)
            rh15, rc15, wh15, wc15, _, _ = ref_chunk(sub15, z15)
        g15.update({f"rgb_head_{S15}": rh15, f"rgb_com_{S15}": rc15, f"w_head_{S15}": wh15, f"w_com_{S15}": wc15,
                    f"z_{S15}": z15[0]})
    save("g15_coarse_nsamples", **g15)

    # row H composed from the reference's own functions (sample_pdf, decoder, composite, weights)
    subh = np.arange(0, H * W, 397)[:512]
    with torch.no_grad():
        zc = (0.3 * (1. - tv) + 0.9 * tv)[None].expand(len(subh), 64)
        out = {}
        for fields in (1, 2):
            c_h, c_c, w_h, w_c, _, _ = ref_chunk(subh, zc)
            w = w_h if fields == 1 else w_c
            z_mid = .5 * (zc[..., 1:] + zc[..., :-1])
            z_f = HELP.sample_pdf(z_mid, w[..., 1:-1], 128, det=True)
            z_all, _ = torch.sort(torch.cat([zc, z_f], -1), -1)
            f_h, f_c, fw_h, fw_c, _, _ = ref_chunk(subh, z_all)
            out[f"z_all_f{fields}"] = z_all
            out[f"rgb_head_f{fields}"] = f_h
            out[f"rgb_com_f{fields}"] = f_c
            out[f"wsum_f{fields}"] = (fw_h.sum(-1) if fields == 1 else fw_c.sum(-1))
    save("g7_frame_hier", ray_idx=subh, frame=np.array([2]), **out)

    # ---- G8: one training step (MAIN:779-931 semantics), three gating regimes ------
    sel = np.random.RandomState(3).permutation(H * W)[:256]
    sel_yx = np.stack([sel // W, sel % W], 1)
    tgt_h = t(synth.synth_tensor(0, "g8/th", (H, W, 3), 0.5)) + 0.5
    tgt_c = t(synth.synth_tensor(0, "g8/tc", (H, W, 3), 0.5)) + 0.5
    g8 = {"sel_yx": sel_yx}
    for step in (0, 300000, 400000):
        dec2, aud2, exp2, att2, patt2 = build_ref_modules(0)
        opts = [torch.optim.Adam(m.parameters(), lr=5e-4, betas=(0.9, 0.999))
                for m in (dec2, aud2, exp2, att2, patt2)]
        img_i = 3
        sig = MAIN.encode_signal(ds, 0, img_i, 96, aud2, exp2, att2, step, a, n, embed_fn=embed_fn)
        sig_t = MAIN.encode_signal_torso(ds, 0, img_i, patt2, step, a, n, embed_fn=embed_fn)
        ro, rd = HELP.get_rays(H, W, focal, poses[img_i, :3, :4], cx, cy)
        rot, rdt = HELP.get_rays(H, W, focal, poses[0, :3, :4], cx, cy)
        ys, xs = t(sel_yx[:, 0]), t(sel_yx[:, 1])
        ro, rd, rot, rdt = ro[ys, xs], rd[ys, xs], rot[ys, xs], rdt[ys, xs]
        N = len(sel)
        zt = (0.3 * (1. - tv) + 0.9 * tv)[None].expand(N, 64)
        p_i = (ro[..., None, :] + rd[..., None, :] * zt[..., :, None]).reshape(1, -1, 3)
        r_i = rd.unsqueeze(1).expand([N, 64, 3]).reshape(1, -1, 3)
        p_t = (rot[..., None, :] + rdt[..., None, :] * zt[..., :, None]).reshape(1, -1, 3)
        r_t = rdt.unsqueeze(1).expand([N, 64, 3]).reshape(1, -1, 3)
        feat_i, sigma_i = dec2(p_i, r_i, z_shape[:, 0], z_app[:, 0], sig, 'head')
        sigma_i = sigma_i.reshape(1, N, 64)
        feat_i = feat_i.reshape(1, N, 64, -1)
        bc_rgb = bc_img[ys, xs]
        feat_i = torch.cat((feat_i[..., :-1, :], bc_rgb.reshape(1, N, 1, 3)), dim=-2)
        feat_t, sigma_t = dec2(p_t, r_t, z_shape[:, 1], z_app[:, 1], sig_t, 'torso')
        sigma_t = sigma_t.reshape(1, N, 64).clone()
        feat_t = feat_t.reshape(1, N, 64, -1)
        sigma_t[:, :, -1] = 0
        # out-of-place forms of MAIN:882-886 (same forward values; the in-place slice write on
        # a relu output is rejected by current autograd)
        bump = torch.zeros(1, 1, 64)
        bump[..., -1] = 1e-6
        sigma = torch.relu(torch.stack([sigma_i], 0))
        sigma = torch.cat([sigma[:-1], sigma[-1:] + bump], 0)
        sigma_to = torch.relu(torch.stack([sigma_i, sigma_t], 0))
        sigma_to = torch.cat([sigma_to[:-1], sigma_to[-1:] + bump], 0)
        feat = torch.stack([feat_i], 0)
        feat_to = torch.stack([feat_i, feat_t], 0)
        ssum, fw = MAIN.composite_function(sigma, feat)
        ssum_t, fw_t = MAIN.composite_function(sigma_to, feat_to)
        w_h = MAIN.calc_volume_weights(zt[None], rd[None], ssum, last_dist=1e10)
        w_c = MAIN.calc_volume_weights(zt[None], rdt[None], ssum_t, last_dist=1e10)
        rgb_com = torch.sum(w_h.unsqueeze(-1) * fw, dim=-2).squeeze(0)
        rgb_com_torso = torch.sum(w_c.unsqueeze(-1) * fw_t, dim=-2).squeeze(0)
        l_h = HELP.img2mse(rgb_com, tgt_h[ys, xs])
        l_c = HELP.img2mse(rgb_com_torso, tgt_c[ys, xs])
        loss = l_c + l_h
        for o in opts:
            o.zero_grad()
        loss.backward()
        opts[0].step()
        opts[1].step()
        if step >= 300000:
            opts[3].step()
            opts[4].step()
        if step >= 400000:
            opts[2].step()
        g8[f"loss_{step}"] = np.array([loss.item(), l_h.item(), l_c.item()])
        for mod, tag in ((dec2, "decoder"), (aud2, "AudNet"), (exp2, "ExpNet"), (att2, "AudAttNet"),
                         (patt2, "PoseAttNet")):
            for k, prm in mod.named_parameters():
                if prm.grad is None:
                    g8[f"gnorm_{step}/{tag}/{k}"] = np.array(-1.0)
                    continue
                g = prm.grad.reshape(-1)
                g8[f"gnorm_{step}/{tag}/{k}"] = np.array(g.double().norm().item())
                g8[f"gsamp_{step}/{tag}/{k}"] = g[:: max(1, g.numel() // 8)][:8].clone()
        for k in ("blocks.3.weight", "fc_in.weight", "deform_net.out_embed.bias"):
            g8[f"after_{step}/decoder/{k}"] = dict(dec2.named_parameters())[k].detach().reshape(-1)[:64].clone()
        g8[f"after_{step}/AudNet/encoder.4.bias"] = aud2.encoder[4].bias.detach().clone()
        g8[f"after_{step}/ExpNet/encoder.2.bias"] = exp2.encoder[2].bias.detach().clone()
        g8[f"after_{step}/AudAttNet/attentionNet.0.bias"] = att2.attentionNet[0].bias.detach().clone()
        g8[f"after_{step}/PoseAttNet/attentionNet.0.bias"] = patt2.attentionNet[0].bias.detach().clone()
        if step == 300000:
            # ---- G12: structure of the checkpoint the reference writes after this step (MAIN:1101-1115) --------------
            # built from the reference's own modules and torch.optim.Adam exactly as upstream does; z_shape / z_app are
            # the [1, 2 n_object, z_dim] latents of MAIN:549-550 (here n_object = 1)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import json
            from ckpt_manifest import checkpoint_manifest
            ck = {'global_step': step + 1, 'z_shape': z_shape, 'z_app': z_app,
                  'network_decoder_state_dict': dec2.state_dict(), 'network_AudNet_state_dict': aud2.state_dict(),
                  'network_ExpNet_state_dict': exp2.state_dict(), 'optimizer_decoder_state_dict': opts[0].state_dict(),
                  'optimizer_Aud_state_dict': opts[1].state_dict(), 'optimizer_Exp_state_dict': opts[2].state_dict(),
                  "network_AudAttNet_state_dict": att2.state_dict(), "optimizer_AudAtt_state_dict": opts[3].state_dict(),
                  "network_PoseAttNet_state_dict": patt2.state_dict(),
                  "optimizer_PoseAtt_state_dict": opts[4].state_dict()}
            with open(os.path.join(HERE, "g12_ckpt_manifest.json"), "w") as f:
                json.dump(checkpoint_manifest(ck), f, indent=0, sort_keys=True)
            print("g12_ckpt_manifest.json written")
    save("g8_train_step", **g8)

    # ---- G9: state_dict manifest (text) ---------------------------------------------
    with open(os.path.join(HERE, "g9_manifest.txt"), "w") as f:
        for mod, tag in ((dec, "decoder"), (audnet, "AudNet"), (expnet, "ExpNet"), (attnet, "AudAttNet"),
                         (pattnet, "PoseAttNet")):
            for k, v in mod.state_dict().items():
                f.write(f"{tag} {k} {tuple(v.shape)}\n")
        f.write("ckpt_keys global_step z_shape z_app network_decoder_state_dict network_AudNet_state_dict "
                "network_ExpNet_state_dict optimizer_decoder_state_dict optimizer_Aud_state_dict "
                "optimizer_Exp_state_dict network_AudAttNet_state_dict optimizer_AudAtt_state_dict "
                "network_PoseAttNet_state_dict optimizer_PoseAtt_state_dict\n")
        f.write("n_params decoder %d\n" % sum(p.numel() for p in dec.parameters()))

    # ---- G11: CLI surface (flag name, kind, default) parsed from the driver's config_parser ------------------
    import ast
    import json as _json
    tree = ast.parse(open(os.path.join(REF, "run_nerf_com_trainExpLater.py")).read())
    flags = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            name = node.args[0].value.lstrip("-")
            kw = {k.arg: k.value for k in node.keywords}
            kind = "value"
            if "action" in kw:
                kind = kw["action"].value
            default = None
            if "default" in kw:
                default = eval(compile(ast.Expression(kw["default"]), "<flag>", "eval"))
            typ = kw["type"].id if "type" in kw else None
            if kw.get("is_config_file") is not None:
                kind = "config_file"
            flags.append({"name": name, "kind": kind, "type": typ, "default": default})
    with open(os.path.join(HERE, "g11_cli_flags.json"), "w") as f:
        _json.dump(flags, f, indent=0)
    print("g11_cli_flags:", len(flags), "flags")

    # ---- G10: to8b / psnr ---------------------------------------------------------------
    x = np.array([-0.1, 0.0, 0.5 / 255, 0.999 / 255, 1.0 / 255, 0.5, 254.999 / 255, 1.0, 1.2,
                  0.99999994, 0.1, 0.2, 0.3], np.float32)
    mse = torch.tensor([1e-4, 0.01, 0.3])
    save("g10_to8b", x=x, y=HELP.to8b(x), mse=mse, psnr=HELP.mse2psnr(mse))

    g13_optional_branches()


def g13_optional_branches():
    """G13: the decoder built with use_expression / use_wav2lip (decoder.py:219-228): the two extra Linear layers in the
    state_dict (names, shapes, position), and - for the one person the scripts train (itr_obj 0: signal = [aud, None],
    MAIN:70) - outputs that do not depend on them.  Runs alone too: python make_golden.py g13"""
    import json as _json
    sc = synth.bench_scene(0, n_frames=8)
    H, W, focal, cx, cy = sc["H"], sc["W"], sc["focal"], sc["cx"], sc["cy"]
    st = synth.synth_all_states(0)
    z_shape, z_app = [t(v) for v in synth.synth_latents(0)]
    dec = DEC.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True, use_expression=True,
                      use_wav2lip=True, use_aud_net=False)
    extra = {"expnet.weight": t(synth.synth_tensor(0, "g13/expnet.weight", (256, 256), 0.1)),
             "expnet.bias": t(synth.synth_tensor(0, "g13/expnet.bias", (256,), 0.1)),
             "w2lnet.weight": t(synth.synth_tensor(0, "g13/w2lnet.weight", (256, 512), 0.1)),
             "w2lnet.bias": t(synth.synth_tensor(0, "g13/w2lnet.bias", (256,), 0.1))}
    sd = {k: t(v) for k, v in st["decoder"].items()}
    sd.update(extra)
    dec.load_state_dict(sd)
    manifest = [[k, list(v.shape)] for k, v in dec.state_dict().items()]
    with open(os.path.join(HERE, "g13_decoder_optional_keys.json"), "w") as f:
        _json.dump(manifest, f, indent=0)
    sig_aud = t(synth.synth_tensor(0, "g3/sig", (1, 96), 0.8))
    sig_torso = t(synth.synth_tensor(0, "g3/sigt", (1, 42), 0.8))
    ro, rd = HELP.get_rays(H, W, focal, t(sc["poses"][0])[:3, :4], cx, cy)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    rays = np.array([0, 101234, 150000, H * W - 1])
    tv = torch.linspace(0., 1., steps=64)
    z = 0.3 * (1. - tv) + 0.9 * tv
    p = (ro[rays, None, :] + rd[rays, None, :] * z[None, :, None]).reshape(1, -1, 3)
    r = rd[rays, None, :].expand(len(rays), 64, 3).reshape(1, -1, 3)
    with torch.no_grad():
        fh, sh = dec(p, r, z_shape[:, 0], z_app[:, 0], [sig_aud, None], 'head')
        ft, st_ = dec(p, r, z_shape[:, 1], z_app[:, 1], sig_torso, 'torso')
    save("g13_decoder_optional", p=p, r=r, sig_aud=sig_aud, sig_torso=sig_torso, feat_head=fh, sigma_head=sh, feat_torso=ft,
         sigma_torso=st_)          # (the extra layers' values: synth.synth_tensor(0, "g13/<key>", shape, 0.1), as above)
    print("g13: ", len(manifest), "state_dict entries")


def g16_z_dim_64():
    """G16 (round 6): --z_dim is free upstream (MAIN:372; Decoder(z_dim=args.z_dim), MAIN:518): the reference's Decoder with z_dim = 64
    - weights and 64-wide latent codes from the same closed-form generators - head, torso and listener outputs at G3's 4 x 64 points."""
    g3 = np.load(os.path.join(HERE, "g3_decoder.npz"))
    dec = DEC.Decoder(z_dim=64, hidden_size=256, dim_signal=96, use_deformation_field=True, use_expression=False, use_aud_net=False)
    dec.load_state_dict({k: t(v) for k, v in synth.synth_decoder_state(0, z_dim=64).items()})
    z_shape, z_app = [t(v) for v in synth.synth_latents(0, z_dim=64)]
    p, r = t(g3["p_64"]), t(g3["r_64"])
    with torch.no_grad():
        fh, sh = dec(p, r, z_shape[:, 0], z_app[:, 0], [t(g3["sig_aud"]), None], 'head')
        ft, st_ = dec(p, r, z_shape[:, 1], z_app[:, 1], t(g3["sig_torso"]), 'torso')
        fl, sl = dec(p, r, z_shape[:, 0], z_app[:, 0], [None, None], 'head')
    save("g16_z_dim_64", feat_head=fh, sigma_head=sh, feat_torso=ft, sigma_torso=st_, feat_listener=fl, sigma_listener=sl)


def g18_n_feat_128():
    """G18 (round 6): --n_feat is free upstream (MAIN:374; Decoder(hidden_size=args.n_feat), MAIN:518; 128 is the class's own default):
    the reference's Decoder(hidden_size=128, z_dim=64) - head, torso (deformation field on), listener at G3's 4 x 64 points."""
    g3 = np.load(os.path.join(HERE, "g3_decoder.npz"))
    dec = DEC.Decoder(z_dim=64, hidden_size=128, dim_signal=96, use_deformation_field=True, use_expression=False, use_aud_net=False)
    dec.load_state_dict({k: t(v) for k, v in synth.synth_decoder_state(0, z_dim=64, hidden=128).items()})
    z_shape, z_app = [t(v) for v in synth.synth_latents(0, z_dim=64)]
    p, r = t(g3["p_64"]), t(g3["r_64"])
    with torch.no_grad():
        fh, sh = dec(p, r, z_shape[:, 0], z_app[:, 0], [t(g3["sig_aud"]), None], 'head')
        ft, st_ = dec(p, r, z_shape[:, 1], z_app[:, 1], t(g3["sig_torso"]), 'torso')
        fl, sl = dec(p, r, z_shape[:, 0], z_app[:, 0], [None, None], 'head')
    save("g18_n_feat_128", feat_head=fh, sigma_head=sh, feat_torso=ft, sigma_torso=st_, feat_listener=fl, sigma_listener=sl)


def g17_no_deformation_field():
    """G17 (round 6): the reference's Decoder WITHOUT --use_deformation_field (a store_true flag, MAIN:411: off unless given) - the
    torso is then the plain 8-layer MLP on [PE, pose signal] (decoder.py:297-299 skipped): torso outputs at G3's 4 x 64 points, same
    weights minus the deform_net tensors."""
    g3 = np.load(os.path.join(HERE, "g3_decoder.npz"))
    dec = DEC.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=False, use_expression=False, use_aud_net=False)
    dec.load_state_dict({k: t(v) for k, v in synth.synth_decoder_state(0).items() if not k.startswith("deform_net.")})
    z_shape, z_app = [t(v) for v in synth.synth_latents(0)]
    with torch.no_grad():
        ft, st_ = dec(t(g3["p_64"]), t(g3["r_64"]), z_shape[:, 1], z_app[:, 1], t(g3["sig_torso"]), 'torso')
        fh, sh = dec(t(g3["p_64"]), t(g3["r_64"]), z_shape[:, 0], z_app[:, 0], [t(g3["sig_aud"]), None], 'head')
    assert np.array_equal(fh.numpy(), g3["feat_head_64"]) and not np.allclose(ft.numpy(), g3["feat_torso_64"], atol=1e-3)
    save("g17_no_deformation_field", feat_torso=ft, sigma_torso=st_)


def g14_listener_backward():
    """G14 (round 6): Decoder.forward with `signal is None` - the listener input layers fc_in_listener / fc_p_skips_listener
    (decoder.py:306-307, 322-323: what the reference's second person evaluates, MAIN:72-75) - UNDER AUTOGRAD in the reference's
    own module: a weighted sum of the outputs at 4 rays x 64 points (G3's points), the gradient of every parameter it reaches:
    norms for all, 8 sampled entries each, the two listener weight matrices in full."""
    g3 = np.load(os.path.join(HERE, "g3_decoder.npz"))
    dec = build_ref_modules(0)[0]
    z_shape, z_app = [t(v) for v in synth.synth_latents(0)]
    p, r = t(g3["p_64"]), t(g3["r_64"])
    n = p.shape[1]
    w_f = t(np.abs(synth.synth_tensor(0, "g14/wf", (1, n, 3), 1.0)))
    w_s = t(np.abs(synth.synth_tensor(0, "g14/ws", (1, n), 0.1)))
    feat, sigma = dec(p, r, z_shape[:, 0], z_app[:, 0], [None, None], 'head')
    assert np.array_equal(feat.detach().numpy(), g3["feat_listener_64"])
    loss = (feat * w_f).sum() + (sigma * w_s).sum()
    loss.backward()
    out = {"w_f": w_f, "w_s": w_s, "loss": loss.detach()}
    names = []
    for k, q in dec.named_parameters():
        if q.grad is None:
            continue
        names.append(k)
        g = q.grad.reshape(-1)
        out["gnorm/" + k] = g.double().norm()
        out["gsamp/" + k] = g[:: max(1, g.numel() // 8)][:8]
    for k in ("fc_in_listener.weight", "fc_p_skips_listener.0.weight"):
        out["gfull/" + k] = dict(dec.named_parameters())[k].grad
    assert "fc_in_listener.weight" in names and "fc_in.weight" not in names and "fc_in_torso.weight" not in names
    with open(os.path.join(HERE, "g14_listener_touched.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    save("g14_listener_backward", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        g14_listener_backward()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g16":
        g16_z_dim_64()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g18":
        g18_n_feat_128()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g17":
        g17_no_deformation_field()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "only":       # everything is computed, one fixture is written
        ONLY = sys.argv[2]
        main()
        sys.exit(0)
    if sys.argv[1:] == ["g13"]:
        g13_optional_branches()
    else:
        main()
