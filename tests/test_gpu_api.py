"""GPU tests of the names BASELINE.json's north star asks to keep - render_rays(), run_network(), create_nerf() - called
through the DROP-IN modules under NeRFs/DFANeRF/ (the files scripts/{train,test}_obama.sh run), against the oracle and
the reference goldens.  SURVEY.md 8(a) row A14: reference signature run_nerf_com_trainExpLater.py:114-143."""
import os
import sys

import numpy as np
import pytest
import torch

import dfa_oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.asarray(x))


@pytest.fixture(scope="module")
def dropin():
    d = os.path.join(ROOT, "NeRFs", "DFANeRF")
    sys.path.insert(0, d)
    import run_nerf_com_trainExpLater as M
    import decoder as D
    yield M, D
    sys.path.remove(d)
    for m in ("run_nerf_com_trainExpLater", "run_nerf_helpers", "decoder", "load_audface", "_bootstrap"):
        sys.modules.pop(m, None)


def _setup(dropin, states, scene, latents, golden, idx):
    M, D = dropin
    dev = torch.device("cuda")
    dec = D.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
    dec.load_state_dict({k: t(v) for k, v in states["decoder"].items()})
    dec.to(dev)
    gc = golden("g7_frame_coarse")
    H, W = scene["H"], scene["W"]
    o_h, d_h = O.get_rays(H, W, scene["focal"], scene["poses"][2][:3, :4], scene["cx"], scene["cy"])
    o_t, d_t = O.get_rays(H, W, scene["focal"], scene["pose_body"][:3, :4], scene["cx"], scene["cy"])
    rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h, o_t, d_t)]
    bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3)[idx]
    zs, za = [t(v) for v in latents]
    args = M.config_parser().parse_args(
        "--expname t --concate_bg --N_samples 64 --N_importance 128 --dim_signal=96 --dim_aud=96 --use_deformation_field "
        "--use_et_embed".split())
    return M, dec, gc, rays, bg, zs, za, args, dev


@pytest.mark.parametrize("mode", ["coarse", "fine"])
def test_render_rays_dropin_vs_oracle(dropin, states, scene, latents, golden, mode):
    """render_rays() through NeRFs/DFANeRF/run_nerf_com_trainExpLater.py: head field, coarse (S = 64, the reference's z)
    and coarse_or_fine='fine' (S = 192, the reference golden's merged depths) against the oracle's single-field
    integrate_fields at the f32 tolerances, and the coarse image against golden G7 itself."""
    g = golden("g7_frame_hier")
    idx = np.asarray(g["ray_idx"])[:192]
    M, dec, gc, rays, bg, zs, za, args, dev = _setup(dropin, states, scene, latents, golden, idx)
    o_h, d_h = rays[0], rays[1]
    R = len(idx)
    if mode == "coarse":
        z = O.coarse_z(scene["near"], scene["far"], 64)[None].expand(R, 64).contiguous()
    else:
        z = t(g["z_all_f1"])[:R].contiguous()
    S = z.shape[1]
    p_i = O.ray_points(o_h, d_h, z).reshape(1, R * S, 3)
    r_i = d_h[:, None, :].expand(R, S, 3).reshape(1, R * S, 3)
    sig = [t(gc["signal"]).to(dev), None]
    rgb, w = M.render_rays(dec, p_i.to(dev), r_i.to(dev), zs[:, 0].to(dev), za[:, 0].to(dev), sig, 'head', 1,
                           bg.reshape(1, R, 1, 3).to(dev), d_h[None].to(dev), z[None].to(dev), args, coarse_or_fine=mode)
    assert tuple(rgb.shape) == (R, 3) and tuple(w.shape) == (1, R, S)
    # grad mode is on and the decoder's parameters require grad: like the reference's, the result carries a graph (round 4:
    # Decoder.forward, composite_function and calc_volume_weights are autograd nodes over HIP kernels) - and under no_grad the
    # same launches give the same bits
    assert rgb.requires_grad and w.requires_grad
    with torch.no_grad():
        rgb0, w0 = M.render_rays(dec, p_i.to(dev), r_i.to(dev), zs[:, 0].to(dev), za[:, 0].to(dev), sig, 'head', 1,
                                 bg.reshape(1, R, 1, 3).to(dev), d_h[None].to(dev), z[None].to(dev), args, coarse_or_fine=mode)
    assert not rgb0.requires_grad and torch.equal(rgb0, rgb.detach()) and torch.equal(w0, w.detach())
    if mode == "coarse":
        (rgb * torch.linspace(0.5, 1.5, 3, device=dev)).sum().backward()
        gn = {k: float(p.grad.norm()) for k, p in dec.named_parameters() if p.grad is not None}
        assert gn["blocks.3.weight"] > 0 and gn["fc_in.weight"] > 0 and "fc_in_torso.weight" not in gn
    rgb, w = rgb.detach(), w.detach()
    P = O.params_to_torch(states["decoder"])
    with torch.no_grad():
        s_h, f_h, _, _ = O._eval_fields(P, *rays, z, zs, za, [t(gc["signal"]), None], None, 1)
        oh, ow, _, _ = O.integrate_fields(z, d_h, rays[3], s_h, f_h, None, None, bg, args.last_dist, True)
    np.testing.assert_allclose(rgb.cpu().numpy(), oh.numpy(), atol=2e-5 if mode == "coarse" else 5e-5, rtol=0)
    np.testing.assert_allclose(w[0].cpu().numpy(), ow.numpy(), atol=2e-6, rtol=0)
    if mode == "coarse":      # the reference's own coarse head image (G7), rays in the same order
        pos = {int(r): i for i, r in enumerate(np.asarray(gc["ray_idx"]))}
        both = [(k, pos[int(r)]) for k, r in enumerate(idx) if int(r) in pos]
        if both:
            a, b = zip(*both)
            np.testing.assert_allclose(rgb.cpu().numpy()[list(a)], gc["rgb_head"][list(b)], atol=2e-5, rtol=0)


def test_run_network_and_create_nerf_dropin(dropin, states, scene, latents, golden):
    """run_network() (decoder on points of any leading shape) against golden G3 (the reference's Decoder.forward) for
    the head and torso fields; create_nerf() builds the five networks with the reference's state_dict layout (G9) and
    HipAdam optimizers."""
    M, D = dropin
    dev = torch.device("cuda")
    g = golden("g3_decoder")
    dec = D.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
    dec.load_state_dict({k: t(v) for k, v in states["decoder"].items()})
    dec.to(dev)
    zs, za = [t(v).to(dev) for v in latents]
    p, r = t(g["p_192"])[0].to(dev), t(g["r_192"])[0].to(dev)
    n = p.shape[0] // 4 * 4
    pts, dirs = p[:n].reshape(4, n // 4, 3), r[:n].reshape(4, n // 4, 3)
    with torch.no_grad():
        feat, sigma = M.run_network(pts, dirs, dec, zs[:, 0], za[:, 0], [t(g["sig_aud"]).to(dev), None], 'head')
        ft, st = M.run_network(pts, dirs, dec, zs[:, 1], za[:, 1], t(g["sig_torso"]).to(dev), 'torso')
    assert tuple(feat.shape) == (4, n // 4, 3) and tuple(sigma.shape) == (4, n // 4)
    np.testing.assert_allclose(feat.reshape(-1, 3).cpu().numpy(), g["feat_head_192"][0][:n], atol=1e-5, rtol=0)
    np.testing.assert_allclose(sigma.reshape(-1).cpu().numpy(), g["sigma_head_192"][0][:n], atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(ft.reshape(-1, 3).cpu().numpy(), g["feat_torso_192"][0][:n], atol=1e-5, rtol=0)
    np.testing.assert_allclose(st.reshape(-1).cpu().numpy(), g["sigma_torso_192"][0][:n], atol=2e-4, rtol=1e-5)
    args = M.config_parser().parse_args(
        "--expname t --z_dim 256 --n_feat 256 --dim_signal=96 --dim_aud=96 --use_deformation_field --use_et_embed "
        "--smo_size 4 --smo_torse_size 8".split())
    nets, opts, embed_fn = M.create_nerf(args, dev)
    lines = open(os.path.join(ROOT, "tests", "golden", "g9_manifest.txt")).read().strip().split("\n")
    for tag, m in nets.items():
        want = [(ln.split(" ", 2)[1], eval(ln.split(" ", 2)[2])) for ln in lines if ln.startswith(tag + " ")]
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == want, tag
    assert set(opts) == set(nets) and all(type(o).__name__ == "HipAdam" for o in opts.values())
    assert embed_fn(torch.zeros(2, 3)).shape == (2, 21)


def test_decoder_with_the_optional_layers_vs_reference_golden(dropin, states, latents, golden):
    """use_expression / use_wav2lip (decoder.py:219-228): golden G13 = the REFERENCE decoder built with both flags (expnet, w2lnet
    registered and, for the one person the scripts train, evaluated by nothing: MAIN:70), head and torso outputs on G3's points.  The
    drop-in module built the same way: same outputs at the f32 tier's gates; in grad mode the two layers keep .grad = None (what
    autograd leaves a parameter no forward used) and every other gradient equals the plain decoder's bit for bit."""
    M, D = dropin
    from dfanerf import synth
    dev = torch.device("cuda")
    g = golden("g13_decoder_optional")
    sd = {k: t(v) for k, v in states["decoder"].items()}
    plain = D.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
    plain.load_state_dict(sd)
    plain.to(dev)
    for k, sh in (("expnet.weight", (256, 256)), ("expnet.bias", (256,)), ("w2lnet.weight", (256, 512)), ("w2lnet.bias", (256,))):
        sd[k] = t(synth.synth_tensor(0, "g13/" + k, sh, 0.1))
    dec = D.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True, use_expression=True, use_wav2lip=True)
    dec.load_state_dict(sd)
    dec.to(dev)
    zs, za = [t(v).to(dev) for v in latents]
    p, r = t(g["p"]).to(dev), t(g["r"]).to(dev)
    sig_aud, sig_torso = t(g["sig_aud"]).to(dev), t(g["sig_torso"]).to(dev)
    with torch.no_grad():
        fh, sh_ = dec(p, r, zs[:, 0], za[:, 0], [sig_aud, None], 'head')
        ft, st_ = dec(p, r, zs[:, 1], za[:, 1], sig_torso, 'torso')
    for got, key, atol, rtol in ((fh, "feat_head", 1e-5, 0), (ft, "feat_torso", 1e-5, 0), (sh_, "sigma_head", 2e-4, 1e-5),
                                 (st_, "sigma_torso", 2e-4, 1e-5)):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], atol=atol, rtol=rtol, err_msg=key)
    grads = []
    for m in (dec, plain):
        m.zero_grad(set_to_none=True)
        fh, sh_ = m(p, r, zs[:, 0], za[:, 0], [sig_aud, None], 'head')
        ft, st_ = m(p, r, zs[:, 1], za[:, 1], sig_torso, 'torso')
        (fh.sum() + 0.01 * sh_.sum() + ft.sum() + 0.01 * st_.sum()).backward()
        torch.cuda.synchronize()
        grads.append({k: (None if q.grad is None else q.grad.clone()) for k, q in m.named_parameters()})
    assert all(grads[0][k] is None for k in ("expnet.weight", "expnet.bias", "w2lnet.weight", "w2lnet.bias"))
    assert grads[1]["fc_in.weight"] is not None and float(grads[1]["fc_in.weight"].abs().max()) > 0
    for k, gv in grads[1].items():
        assert (gv is None) == (grads[0][k] is None), k
        if gv is not None:
            assert torch.equal(gv, grads[0][k]), k


def test_signal_encoders_single_frame_beyond_sequence_length(states, scene):
    """before --nosmo_iters the reference indexes auds[img_i] directly (MAIN:59-61) - a training frame whose
    index is >= len(i_train) (the count left after the speak_frames filter) must NOT get a zero signal.  The window
    clamp of the smoothed branch (MAIN:36-57) still applies."""
    from dfanerf import engine, nets
    dev = torch.device("cuda")
    mods = {"AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in states[k].items()})
        m.to(dev)
    auds, exps, poses = [t(scene[k]).to(dev) for k in ("aud", "exp", "poses")]
    n = auds.shape[0]
    enc = engine.SignalEncoder(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"], auds, exps, poses)
    full, fullt = enc.encode([n - 1], 0, 0)                         # whole sequence
    short, shortt = enc.encode([n - 1], 0, 0, length=n - 3)         # frame index beyond the "training length"
    assert torch.equal(full, short) and torch.equal(fullt, shortt) and float(short.abs().max()) > 0
    with torch.no_grad():
        ref = torch.cat([mods["AudNet"](auds[n - 1:n]), mods["ExpNet"](exps[n - 1:n])], 1)
    torch.testing.assert_close(short, ref, rtol=1e-4, atol=1e-5)
    # smoothed branch: rows beyond `length` are zero INPUT rows, so the result differs from the full-length window
    a, _ = enc.encode([n - 4], 4, 8, length=n - 3)
    b, _ = enc.encode([n - 4], 4, 8)
    assert not torch.equal(a, b)


def test_frame_prefetcher_equals_the_direct_front_end(dropin, states, scene, latents):
    """engine.FramePrefetcher (signal encoders + bias fold one frame ahead on a side stream, run_nerf.py's render loop and
    bench.py): the blob it hands out is bit for bit the blob of encode() + fold() on the render stream - with the hinted
    frame, with a hint that turns out wrong, without a hint, and when a blob's slot comes round again."""
    from dfanerf import engine, nets
    M, D = dropin
    dev = torch.device("cuda")
    mods = {"AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in states[k].items()})
        m.to(dev)
    dec = D.Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
    dec.load_state_dict({k: t(v) for k, v in states["decoder"].items()})
    dec.to(dev)
    auds, exps, poses = [t(scene[k]).to(dev) for k in ("aud", "exp", "poses")]
    enc = engine.SignalEncoder(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"], auds, exps, poses)
    zs, za = [t(v).reshape(-1, 256)[:2].to(dev).float().contiguous() for v in latents]
    pk = dec.packed("f16")
    n = auds.shape[0]
    order = [3, 0, n - 1, n - 1, 2, 5, 1]
    hints = [0, 4, n - 1, 2, None, 1, None]          # right, WRONG, right (same frame again), right, none, right, none
    pf = engine.FramePrefetcher(enc, pk, zs, za, 4, 8, fields=2)
    for f, nxt in zip(order, hints):
        got = pf.get(f, nxt).clone()
        pf.done()
        s2, t2 = enc.encode([f], 4, 8)
        want = pk.fold(s2[0], t2[0], zs, za)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f


def test_rccl_backend_world1_collectives(scene, states, latents, golden):
    """The collectives of the multi-GPU path on the RCCL backend ("nccl" on ROCm), on the one GPU a gpurun box has
    (world size 1; RCCL refuses two ranks on one device, so a real exchange needs the driver's multi-GPU node): process
    group initialisation as bench.py / parallel.init do it, the padded-shard all_gather_into_tensor of
    FrameRenderer.render_image (the rendered shard IS the frame), all_reduce of the flat gradient bucket, broadcast."""
    import torch.distributed as dist
    from dfanerf import engine, parallel
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == 1                                    # bench.py's `rccl_ranks`
        H, W = scene["H"], scene["W"]
        R = H * W
        begin, count, per = parallel.shard_range(R, 1, 0)
        assert (begin, count, per) == (0, R, R)
        flat = engine.flatten_state(states["decoder"], dev)
        pk = engine.PackedDecoder(flat, "f16", fields=(0,))
        gc = golden("g7_frame_coarse")
        zs, za = latents
        bias = pk.fold(gc["signal"][0], None, zs[0], za[0])
        bg8 = t(scene["bg"]).reshape(-1, 3).to(dev)
        shard = torch.zeros(1, per, 3, dtype=torch.uint8, device=dev)
        gathered = torch.empty(1, 1, per, 3, dtype=torch.uint8, device=dev)
        fr = engine.make_frame(H, W, scene["focal"], scene["cx"], scene["cy"], scene["poses"][2], scene["pose_body"],
                               scene["near"], scene["far"], ray_begin=begin, ray_count=count, n_fine=0, fields=1)
        engine.render_u8(pk, bias, fr, bg8, out_head=shard[0, :count])
        dist.all_gather_into_tensor(gathered.view(1 * 1, per, 3), shard)        # the concatenation form, as the product passes it
        assert torch.equal(gathered[0], shard) and int(gathered.sum()) > 0
        bucket = torch.arange(1138656, dtype=torch.float32, device=dev)      # the flat gradient bucket's size
        ref = bucket.clone()
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        dist.broadcast(bucket, src=0)
        assert torch.equal(bucket, ref)
    finally:
        dist.destroy_process_group()
