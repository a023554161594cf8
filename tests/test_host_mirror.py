"""CPU tests of the host-side mirror of the reference's Python surface (dfa-nerf_amd/dfanerf/*.py and the
drop-in modules under NeRFs/DFANeRF/): CLI flags, state_dict / checkpoint compatibility, the autograd
(training) path against golden G8, and the multi-process partitioning with gloo, world_size 2."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import twins
from conftest import GOLDEN, ROOT
from dfanerf import nets, parallel, run_nerf, synth
from dfanerf.decoder import Decoder

torch.set_num_threads(8)


def t(x):
    return torch.from_numpy(np.asarray(x))


def _modules(states):
    dec = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
    mods = {"decoder": dec, "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(),
            "AudAttNet": nets.AudioAttNet(96, 4), "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in states[k].items()})
    return mods


def test_cli_flags_match_reference():
    want = json.load(open(os.path.join(GOLDEN, "g11_cli_flags.json")))
    parser = run_nerf.config_parser()
    acts = {a.dest: a for a in parser._actions}
    for f in want:
        a = acts[f["name"]]
        if f["kind"] == "store_true":
            assert a.const is True and a.default is False, f
        elif f["kind"] == "store_false":
            assert a.const is False and a.default is True, f
        elif f["kind"] == "value":
            assert a.default == f["default"], f
            if f["type"]:
                assert a.type.__name__ == f["type"], f
    extra = set(acts) - {f["name"] for f in want} - {"help"}
    assert extra == {"hip_tier", "hierarchical", "image_ext", "hip_train_act", "hip_f16_model_psnr"}


def test_config_file_and_script_flags(tmp_path):
    cfg = tmp_path / "HeadNeRF_config_ba.txt"
    cfg.write_text("expname = obama_head\ndatadir = dataset/obama\nbasedir = dataset/obama/logs\n"
                   "near = 0.3127\nfar = 0.9127\ntestskip = 1\n")
    # the flag bundle of scripts/test_obama.sh
    argv = ("--config %s --last_dist=1e10 --datadir dataset/obama --concate_bg --N_rand=2048 --sample_rate=0 "
            "--i_print=100 --i_test_person=10000 --chunk=2048 --win_size=16 --smo_size=4 --smo_torse_size 8 "
            "--train_together --i_weights=100000 --all_speaker --sample_rate_mouth=0 --lrate_decay=500 --lrate=5e-4 "
            "--use_et_embed --nosmo_iters=300000 --dim_signal=96 --dim_aud=96 --n_object=1 --N_iters=600000 "
            "--expname=obama_TrainExpLater_smoMix --aud_file=obama_aud.pt --use_deformation_field "
            "--exp_file=obama_64_32.pt --use_ba --render_person --noexp_iters 400000 "
            "--resume x/280000.tar --test_file transforms_val_ba.json --render_video" % cfg).split()
    a = run_nerf.config_parser().parse_args(argv)
    assert a.near == 0.3127 and a.far == 0.9127 and a.testskip == 1         # from the config file
    assert a.expname == "obama_TrainExpLater_smoMix"                         # command line wins over the file
    assert a.chunk == 2048 and a.smo_size == 4 and a.render_person and a.concate_bg and a.use_ba
    assert run_nerf.parse_config_file(str(cfg)) == (0.3127, 0.9127)


def test_unsupported_configurations_are_refused_at_parse_time():
    ok = run_nerf.config_parser().parse_args(
        "--expname t --n_feat 256 --z_dim 256 --dim_signal 96 --n_object 1 --use_deformation_field".split())
    run_nerf.check_supported(ok)
    # --n_object 2 is the flag's DEFAULT upstream: accepted at parse time (round 6) - like upstream the run then stops in train()'s
    # per-person setup loop, whose `datadir` list has one entry (MAIN:449, 495-501)
    run_nerf.check_supported(run_nerf.config_parser().parse_args(
        "--expname t --n_feat 256 --z_dim 256 --dim_signal 96 --use_deformation_field".split()))
    # --z_dim / --n_feat 1 ... 256 and a missing --use_deformation_field (round 6): narrower / plainer decoders live in the library's
    # 256-wide layout with zero rows / columns / an all-zero deformation network - rendering and training
    for ok_extra in ("--use_deformation_field --z_dim 64 --render_person", "--use_deformation_field --n_feat 128 --z_dim 64",
                     "--render_person", ""):
        run_nerf.check_supported(run_nerf.config_parser().parse_args(("--expname t --dim_signal 96 --n_object 1 " + ok_extra).split()))
    for extra in ("--n_feat 512", "--z_dim 300 --render_person", "--N_samples 48", "--hierarchical --N_samples 128", "--dim_signal 128", "--hierarchical --N_importance 96", "--n_object 0",
                  "--hip_tier fp8", "--hip_train_act e2m3"):
        a = run_nerf.config_parser().parse_args(
            ("--expname t --z_dim 256 --dim_signal 96 --n_object 1 --use_deformation_field --n_feat 256 " + extra).split())
        with pytest.raises(SystemExit, match="unsupported configuration"):
            run_nerf.check_supported(a)


def test_decoder_optional_layers_are_registered_and_outside_the_flat_vector(states):
    """use_expression / use_wav2lip (decoder.py:219-228): the reference registers expnet / w2lnet and, for the one person its scripts
    train, evaluates neither (MAIN:70: signal = [aud, None]).  Golden G13 = the reference module's state_dict (names, shapes, order):
    the mirror's equals it, --use_expression is accepted at parse time, and the kernels' flat parameter vector is the same 955,242
    values with or without the two layers."""
    import json
    from dfanerf import engine, training
    from dfanerf._lib import N_DECODER_PARAMS
    want = [(k, tuple(sh)) for k, sh in json.load(open(os.path.join(GOLDEN, "g13_decoder_optional_keys.json")))]
    dec = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True, use_expression=True, use_wav2lip=True)
    assert [(k, tuple(v.shape)) for k, v in dec.state_dict().items()] == want
    assert dec.hip_supported()
    sd = {k: t(v) for k, v in states["decoder"].items()}
    dec.load_state_dict(sd, strict=False)
    base = _modules(states)["decoder"]
    assert torch.equal(engine.flatten_state(dec.state_dict(), "cpu"), engine.flatten_state(base.state_dict(), "cpu"))
    assert engine.flatten_state(dec.state_dict(), "cpu").numel() == N_DECODER_PARAMS
    fn = training._FlatNet(dec)
    assert fn.flat.numel() == N_DECODER_PARAMS and not any(n.startswith(("expnet.", "w2lnet.")) for n in fn.names)
    a = run_nerf.config_parser().parse_args(
        "--expname t --n_feat 256 --z_dim 256 --dim_signal 96 --n_object 1 --use_deformation_field --use_expression".split())
    run_nerf.check_supported(a)
    with pytest.raises(NotImplementedError, match="second"):
        dec(torch.zeros(1, 64, 3), torch.zeros(1, 64, 3), torch.zeros(1, 256), torch.zeros(1, 256),
            [None, torch.zeros(1, 256)], 'head')


def test_state_dict_manifest(states):
    lines = open(os.path.join(GOLDEN, "g9_manifest.txt")).read().strip().split("\n")
    mods = _modules(states)
    for tag, m in mods.items():
        want = [(ln.split(" ", 2)[1], eval(ln.split(" ", 2)[2])) for ln in lines if ln.startswith(tag + " ")]
        got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        assert got == want, tag
    assert mods["decoder"].hip_supported()
    assert not Decoder().hip_supported()


def test_decoder_twin_vs_golden_and_no_cpu_path(states, latents, golden):
    """tests/twins.py:decoder_forward_aten (the torch-op restatement the GPU tests cross-check the HIP training path
    with) against golden G3; and the product's Decoder.forward refuses CPU tensors in both modes (single backend)."""
    g = golden("g3_decoder")
    dec = _modules(states)["decoder"]
    zs, za = [t(v) for v in latents]
    p, r = t(g["p_64"]), t(g["r_64"])
    with torch.no_grad():
        fh, sh = twins.decoder_forward_aten(dec, p, r, zs[:, 0], za[:, 0], [t(g["sig_aud"]), None], 'head')
        ft, st = twins.decoder_forward_aten(dec, p, r, zs[:, 1], za[:, 1], t(g["sig_torso"]), 'torso')
        fl, sl = twins.decoder_forward_aten(dec, p, r, zs[:, 0], za[:, 0], [None, None], 'head')
    for got, ref in ((fh, "feat_head_64"), (sh, "sigma_head_64"), (ft, "feat_torso_64"), (st, "sigma_torso_64"),
                     (fl, "feat_listener_64"), (sl, "sigma_listener_64")):
        np.testing.assert_allclose(got.detach().numpy(), g[ref], rtol=1e-5, atol=2e-5)
    assert np.array_equal(dec.transform_points(p[:, :8]).numpy(), g["pe_p"])
    for ctx in (torch.no_grad(), torch.enable_grad()):
        with ctx, pytest.raises(RuntimeError, match="no CPU fallback"):
            dec(p, r, zs[:, 0], za[:, 0], [t(g["sig_aud"]), None], 'head')
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        run_nerf.composite_function(torch.zeros(1, 1, 2, 4), torch.zeros(1, 1, 2, 4, 3))
    with pytest.raises(RuntimeError, match="no CPU"):
        run_nerf.make_adam(torch.nn.Linear(3, 2).parameters(), 5e-4)


def _args():
    a = run_nerf.config_parser().parse_args(
        "--expname t --concate_bg --N_rand=256 --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
        "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000".split())
    return a


@pytest.mark.parametrize("step", [0, 300000, 400000])
def test_training_step_vs_golden(states, scene, latents, golden, step):
    """Host logic of the training step on CPU: the torch-op twin of the forward (tests/twins.py, MAIN:779-907) + the
    product's optimizer gating and LR schedule (run_nerf.optimizer_steps / update_lrate, MAIN:916-931, 1081-1094)
    against golden G8."""
    g = golden("g8_train_step")
    mods = _modules(states)
    args = _args()
    H, W = scene["H"], scene["W"]
    ds = [{"auds": t(scene["aud"]), "exp": t(scene["exp"]), "poses": t(scene["poses"]),
           "bc_img": t(scene["bg"]).float() / 255.0, "hwfcxy": [H, W, scene["focal"], scene["cx"], scene["cy"]],
           "near": 0.3, "far": 0.9}]
    sel = g["sel_yx"]
    tgt_h = t(synth.synth_tensor(0, "g8/th", (H, W, 3), 0.5)) + 0.5
    tgt_c = t(synth.synth_tensor(0, "g8/tc", (H, W, 3), 0.5)) + 0.5
    zs, za = [t(v) for v in latents]
    embed_fn, _ = nets.get_embedder(3, 0)
    opts = {k: torch.optim.Adam(m.parameters(), lr=5e-4, betas=(0.9, 0.999)) for k, m in mods.items()}
    loss, lh, lc, _, _ = twins.train_step_loss_aten(mods, ds, 0, 3, sel, tgt_h[sel[:, 0], sel[:, 1]],
                                                    tgt_c[sel[:, 0], sel[:, 1]], zs, za, step, args,
                                                    scene["aud"].shape[0], embed_fn, ds[0]["poses"][0, :3, :4])
    np.testing.assert_allclose([loss.item(), lh.item(), lc.item()], g[f"loss_{step}"], rtol=3e-6)
    for o in opts.values():
        o.zero_grad()
    loss.backward()
    for tag, m in mods.items():
        for k, p in m.named_parameters():
            ref = float(g[f"gnorm_{step}/{tag}/{k}"])
            got = 0.0 if p.grad is None else p.grad.double().norm().item()
            if ref < 0:
                assert got == 0
            else:
                assert abs(got - ref) <= 3e-4 * ref + 1e-9, (tag, k, got, ref)
    run_nerf.optimizer_steps(opts, step, args)
    for key in [k for k in g if k.startswith(f"after_{step}/")]:
        _, tag, name = key.split("/", 2)
        got = dict(mods[tag].named_parameters())[name].detach().reshape(-1)[: g[key].size].numpy()
        np.testing.assert_allclose(got, g[key].reshape(-1), rtol=0, atol=3e-6)
    lr = run_nerf.update_lrate(opts, step, args)
    assert opts["AudAttNet"].param_groups[0]["lr"] == 2 * lr and opts["ExpNet"].param_groups[0]["lr"] == 5e-4


def test_checkpoint_roundtrip(tmp_path, states, latents):
    mods = _modules(states)
    opts = {k: torch.optim.Adam(m.parameters(), lr=5e-4) for k, m in mods.items()}
    zs, za = [t(v) for v in latents]
    path = str(tmp_path / "000010.tar")
    run_nerf.save_checkpoint(path, 11, zs, za, mods, opts)
    ck = torch.load(path, weights_only=False)
    want = open(os.path.join(GOLDEN, "g9_manifest.txt")).read().strip().split("\n")
    keys = [ln for ln in want if ln.startswith("ckpt_keys ")][0].split(" ")[1:]
    assert sorted(ck.keys()) == sorted(keys)
    mods2 = _modules({k: {kk: np.zeros_like(v) for kk, v in st.items()} for k, st in states.items()})
    opts2 = {k: torch.optim.Adam(m.parameters(), lr=5e-4) for k, m in mods2.items()}
    step, zs2, za2 = run_nerf.load_checkpoint(path, mods2, opts2)
    assert step == 11 and torch.equal(zs2, zs) and torch.equal(za2, za)
    for k in mods:
        for a, b in zip(mods[k].state_dict().values(), mods2[k].state_dict().values()):
            assert torch.equal(a, b)


def test_select_coords_and_shards():
    rng = np.random.RandomState(0)
    sel = run_nerf.select_coords(450, 450, 2048, 0, None, rng)
    assert sel.shape == (2048, 2) and len({(a, b) for a, b in sel}) == 2048
    sel = run_nerf.select_coords(450, 450, 2048, 0.95, np.array([100, 120, 80, 90]), rng)
    inside = ((sel[:, 0] >= 100) & (sel[:, 0] <= 180) & (sel[:, 1] >= 120) & (sel[:, 1] <= 210)) | (sel[:, 0] >= 225)
    assert inside.sum() == int(2048 * 0.95)
    for R, P in ((202500, 8), (202500, 1), (7, 8), (64, 3)):
        cover = []
        for r in range(P):
            b, n, per = parallel.shard_range(R, P, r)
            assert per == -(-R // P) and 0 <= n <= per
            cover += list(range(b, b + n))
        assert cover == list(range(R))
    assert parallel.shard_range(202500, 8, 7) == (177191, 25309, 25313)


class _FakeFlat:
    """What training._FlatNet is to FlatGradBucket._adopt: one flat gradient buffer per module whose slices are the
    parameters' .grad (the HIP backward writes `_g_flat`; parameters a forward never used keep .grad = None)."""

    def __init__(self, module):
        self.params = list(module.parameters())
        self.flat = torch.zeros(sum(p.numel() for p in self.params))
        self._g_flat = None

    def backward(self, value, skip_last=True):
        g = self._g_flat
        if g is None or any(p.grad is not None for p in self.params):
            g = torch.zeros_like(self.flat)
            if not any(p.grad is not None for p in self.params):
                self._g_flat = g
        g.zero_()
        o = 0
        for i, p in enumerate(self.params):
            n = p.numel()
            if not (skip_last and i == len(self.params) - 1):
                g[o:o + n] = value
                p.grad = g[o:o + n].view_as(p)
            o += n


def _mixed_bucket_steps(rank, world):
    """ONE bucket over a network whose gradients live in its own flat buffer (adopted into the bucket after
    the first all_reduce_, one parameter always without a gradient - the decoder and its listener layers) and networks
    whose gradients are fresh torch-autograd tensors every step (conditioning networks without a SignalTrainer).  The
    adopted network's gradients must survive the copy path on every step; parameters without a gradient keep None."""
    torch.manual_seed(5)
    dec = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2, bias=True))
    cond = torch.nn.Linear(3, 2)
    idle = torch.nn.Linear(2, 2)                         # a network that runs no backward at all (PoseAttNet, smo_t == 0)
    fn = dec.__dict__["_dfn_flat"] = _FakeFlat(dec)
    bk = parallel.FlatGradBucket([dec, cond, idle])
    out = []
    for it in range(3):
        for m in (dec, cond, idle):
            for p in m.parameters():
                p.grad = None
        fn.backward(float(10 * it + rank + 1))
        cond(torch.full((2, 3), float(rank + 1 + it))).sum().backward()
        bk.all_reduce_()
        ps = list(dec.parameters())
        out.append((float(sum(p.grad.sum() for p in ps[:-1])), ps[-1].grad is None,
                    float(cond.weight.grad.sum()), all(p.grad is None for p in idle.parameters()),
                    ps[0].grad.data_ptr() == bk.views[0].data_ptr()))
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R = 1001
    begin, count, per = parallel.shard_range(R, world, rank)
    full = torch.arange(R * 3, dtype=torch.float32).reshape(R, 3)
    shard = torch.zeros(per, 3)
    shard[:count] = full[begin:begin + count]
    img = parallel.gather_rays(shard, R)
    ok_gather = torch.equal(img, full)
    torch.manual_seed(0)
    mods = [torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)]
    x = torch.full((4, 5), float(rank + 1))
    mods[1](mods[0](x)).sum().backward()
    mods[1].bias.grad = None                           # a missing grad counts as zero
    g_local = [None if p.grad is None else p.grad.clone() for m in mods for p in m.parameters()]
    bucket = parallel.FlatGradBucket(mods)
    bucket.all_reduce_()
    # replicas that start DIFFERENT (every rank initialises from its own RNG, as train() does without --resume) must be
    # identical after broadcast_replicas and stay identical through data-parallel steps with different data per rank
    torch.manual_seed(100 + rank)
    nets = {"a": torch.nn.Linear(6, 4), "b": torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))}
    opts = {k: torch.optim.Adam(m.parameters(), lr=1e-2) for k, m in nets.items()}
    if rank == 0:                       # rank 0 carries optimizer state (a resumed checkpoint), rank 1 none
        nets["b"](nets["a"](torch.ones(2, 6))).sum().backward()
        for o in opts.values():
            o.step(); o.zero_grad()
    lat = torch.randn(1, 2, 8)
    (lat,) = parallel.broadcast_replicas(nets, opts, [lat], src=0)
    bk = parallel.FlatGradBucket(list(nets.values()))
    for it in range(3):
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(1000 * rank + it))      # per-rank data
        for o in opts.values():
            o.zero_grad()
        (nets["b"](nets["a"](x)) ** 2).mean().backward()
        bk.all_reduce_()
        for o in opts.values():
            o.step()
    rep = np.concatenate([p.detach().numpy().reshape(-1) for k in sorted(nets) for p in nets[k].parameters()] +
                         [lat.numpy().reshape(-1)])
    mixed = _mixed_bucket_steps(rank, world)
    # strong-scaling split of the training step (bench.py c4s): the reference's 2048 rays split over the ranks
    b0, n0, per0 = parallel.shard_range(2048, world, rank)
    # plain numpy through the queue (tensors would travel as shared-memory file descriptors, which is fragile when
    # the producer exits early)
    q.put((rank, bool(ok_gather), int(bucket.numel), [None if g is None else g.numpy() for g in g_local],
           [None if p.grad is None else p.grad.detach().clone().numpy() for m in mods for p in m.parameters()], rep,
           (b0, n0, per0), mixed))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    except Exception:
        res = None
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    if res is None or any(p.exitcode != 0 for p in procs):
        return None
    return res


def test_gloo_world2_gather_and_grad_bucket():
    res = _run_world2() or _run_world2()       # one retry: the free port found above can be taken in between
    assert res is not None
    assert all(r[1] for r in res) and res[0][2] == 5 * 3 + 3 + 3 * 2 + 2
    for k in range(4):
        a, b = res[0][3][k], res[1][3][k]
        if a is None and b is None:                 # `.grad = None` stays None (single-rank semantics: Adam skips it)
            assert res[0][4][k] is None and res[1][4][k] is None
            continue
        z = np.zeros_like(res[0][4][k])
        want = ((a if a is not None else z) + (b if b is not None else z)) / 2
        assert np.allclose(res[0][4][k], want) and np.allclose(res[1][4][k], want)
    # replicas bit-identical after broadcast + 3 data-parallel Adam steps (they used to diverge from step 0)
    assert np.array_equal(res[0][5], res[1][5])
    assert res[0][6] == (0, 1024, 1024) and res[1][6] == (1024, 1024, 1024)
    # mixed bucket (adopted flat-buffer network + torch-autograd networks), three steps: the adopted network's averaged
    # gradient is (10 it + 1.5) per element on 3*4 + 4 + 4*2 = 24 elements every step (it was 0 from step 2), the parameter
    # without a gradient and the idle network keep None, the torch-autograd network averages too
    for r in (0, 1):
        for it, (dsum, last_none, csum, idle_none, in_bucket) in enumerate(res[r][7]):
            assert abs(dsum - 24 * (10 * it + 1.5)) < 1e-4 and last_none and idle_none and in_bucket
            assert abs(csum - 2 * 2 * 3 * (it + 1.5)) < 1e-4


def test_dropin_modules_importable():
    d = os.path.join(ROOT, "NeRFs", "DFANeRF")
    sys.path.insert(0, d)
    try:
        import run_nerf_com_trainExpLater as M
        import run_nerf_helpers as Hm
        import decoder as D
        import load_audface as L
        for name in ("render_rays", "composite_function", "calc_volume_weights", "encode_signal",
                     "encode_signal_torso", "config_parser", "train", "run_network", "create_nerf"):
            assert callable(getattr(M, name))
        for name in ("get_rays", "ndc_rays", "sample_pdf", "get_embedder", "AudioNet_W2L", "ExpressionEnc",
                     "AudioAttNet", "img2mse", "mse2psnr", "to8b"):
            assert hasattr(Hm, name)
        assert hasattr(D, "Decoder") and callable(L.load_audface_data_split)
    finally:
        sys.path.remove(d)
        for m in ("run_nerf_com_trainExpLater", "run_nerf_helpers", "decoder", "load_audface", "_bootstrap"):
            sys.modules.pop(m, None)


@pytest.mark.parametrize("step", [0, 300000])
def test_encode_signals_batch_matches_per_frame(states, scene, step):
    """Batched signal encoders == the per-frame encode_signal / encode_signal_torso, including
    the zero-padded windows at both ends of the sequence."""
    mods = _modules(states)
    n = scene["aud"].shape[0]
    ds = [{"auds": t(scene["aud"]), "exp": t(scene["exp"]), "poses": t(scene["poses"])}]
    embed_fn, _ = nets.get_embedder(3, 0)

    class A:
        nosmo_iters, smo_size, smo_torse_size = 300000, 4, 8
    ids = [0, 1, n // 2, n - 2, n - 1]
    with torch.no_grad():
        sb, stb = nets.encode_signals_batch(ds, 0, ids, mods["AudNet"], mods["ExpNet"], mods["AudAttNet"],
                                            mods["PoseAttNet"], step, A, n, embed_fn)
        for b, i in enumerate(ids):
            s1 = nets.encode_signal(ds, 0, i, 96, mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], step, A, n,
                                    embed_fn=embed_fn)[0]
            t1 = nets.encode_signal_torso(ds, 0, i, mods["PoseAttNet"], step, A, n, embed_fn=embed_fn)
            torch.testing.assert_close(sb[b], s1.reshape(-1), rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(stb[b], t1.reshape(-1), rtol=1e-4, atol=1e-5)


def test_select_coords_distinct_and_region_counts():
    """Pixel sampling of MAIN:786-820: distinct pixels, the requested share inside (face rect | lower half)."""
    rng = np.random.RandomState(7)
    H = W = 450
    for _ in range(3):
        sel = run_nerf.select_coords(H, W, 2048, 0, None, rng)
        assert sel.shape == (2048, 2) and sel.dtype == np.int64
        assert len({(int(y), int(x)) for y, x in sel}) == 2048
        assert sel.min() >= 0 and sel[:, 0].max() < H and sel[:, 1].max() < W
    rect = (100, 120, 150, 160)
    sel = run_nerf.select_coords(H, W, 2048, 0.95, rect, rng)
    y, x = sel[:, 0], sel[:, 1]
    inside = ((y >= rect[0]) & (y <= rect[0] + rect[2]) & (x >= rect[1]) & (x <= rect[1] + rect[3])) | (y >= H / 2)
    assert len({(int(a), int(b)) for a, b in sel}) == 2048 and int(inside.sum()) == int(2048 * 0.95)
    # dense request (more than a quarter of the population) still works
    s = run_nerf._choice_distinct(rng, 1000, 900)
    assert len(set(s.tolist())) == 900 and s.min() >= 0 and s.max() < 1000
    # roughly uniform: mean index of many draws is near the middle
    m = np.mean([run_nerf._choice_distinct(rng, 202500, 2048).mean() for _ in range(20)])
    assert abs(m - 101250) < 2500


def test_frame_writer_pipeline(tmp_path):
    """Output stage: images go through the pinned-buffer ring and the encoder thread in order, files are written."""
    from PIL import Image
    H, W = 24, 32
    w = run_nerf._FrameWriter(H, W, 2, depth=2)
    kept, imgs = [], []
    for i in range(5):
        a = torch.full((H, W, 3), 10 * i, dtype=torch.uint8)
        b = torch.full((H, W, 3), 10 * i + 1, dtype=torch.uint8)
        imgs.append(a.numpy().copy())
        w.submit([a, b], [str(tmp_path / f"com_{i}.jpg"), str(tmp_path / f"head_{i}.jpg") if i % 2 == 0 else None],
                 keep=kept)
    w.drain()
    assert len(kept) == 5 and all(np.array_equal(k, im) for k, im in zip(kept, imgs))
    for i in range(5):
        im = np.asarray(Image.open(tmp_path / f"com_{i}.jpg"))
        assert im.shape == (H, W, 3) and abs(int(im.mean()) - 10 * i) <= 2
        assert (tmp_path / f"head_{i}.jpg").exists() == (i % 2 == 0)


def test_frame_writer_mixed_keep_and_no_keep():
    """frames submitted with keep=None must neither append None entries to a keep list used by other submissions nor
    pile up markers when nothing is kept at all."""
    H, W = 8, 8
    img = lambda v: [torch.full((H, W, 3), v, dtype=torch.uint8)]
    w = run_nerf._FrameWriter(H, W, 1, depth=2)
    for i in range(6):                       # nothing kept at all: no bookkeeping grows
        w.submit(img(i), [None])
    w.drain()
    assert w._kept == {}
    kept = []
    for i in range(6):                       # then a loop that keeps every other frame
        w.submit(img(100 + i), [None], keep=kept if i % 2 == 0 else None)
    w.drain()
    assert [int(k[0, 0, 0]) for k in kept] == [100, 102, 104] and all(k is not None for k in kept)
    assert w._kept == {} and w.stats()["frames"] == 12 and w.stats()["encoder_busy_s"] >= 0.0


def test_pixel_sampler_matches_select_coords_semantics():
    """frames.PixelSampler (the device-side draw of MAIN:786-820; on CPU here) against run_nerf.select_coords (the host
    restatement of the same lines): N_rand distinct pixels; with sample_rate > 0 exactly int(N_rand * rate) of them
    inside (face rect | lower half) - the same counts select_coords produces; uniform coverage without a rate."""
    from dfanerf import frames
    H = W = 450
    dev = torch.device("cpu")
    s0 = frames.PixelSampler(H, W, 2048, 0, dev, seed=3)
    seen = []
    for _ in range(4):
        p = s0.draw().numpy()
        assert p.dtype == np.int32 and p.shape == (2048,) and len(set(p.tolist())) == 2048
        assert p.min() >= 0 and p.max() < H * W
        seen.append(p)
    assert not np.array_equal(seen[0], seen[1])                                  # the generator advances
    assert abs(np.mean([p.mean() for p in seen]) - (H * W - 1) / 2) < 4000       # uniform over the frame
    assert np.array_equal(frames.PixelSampler(H, W, 2048, 0, dev, seed=3).draw().numpy(), seen[0])     # seeded
    rect = np.array([100, 120, 150, 160])
    host = run_nerf.select_coords(H, W, 2048, 0.95, rect, np.random.RandomState(0))
    inside_host = ((host[:, 0] >= 100) & (host[:, 0] <= 250) & (host[:, 1] >= 120) & (host[:, 1] <= 280)) | (host[:, 0] >= H / 2)
    s1 = frames.PixelSampler(H, W, 2048, 0.95, dev, seed=4)
    for r in (rect, torch.as_tensor(rect)):
        p = s1.draw(r).numpy()
        y, x = p // W, p % W
        inside = ((y >= 100) & (y <= 250) & (x >= 120) & (x <= 280)) | (y >= H / 2)
        assert len(set(p.tolist())) == 2048 and int(inside.sum()) == int(inside_host.sum()) == int(2048 * 0.95)
        assert inside[:int(2048 * 0.95)].all() and not inside[int(2048 * 0.95):].any()
    with pytest.raises(ValueError):
        frames.PixelSampler(4, 4, 17, 0, dev)


def test_device_frame_cache_preload_lru_and_zero_reads_when_warm(tmp_path):
    """frames.DeviceFrameCache: every frame decoded once; a warm cache serves get() without touching a file; when the
    sequence does not fit the budget the least recently used frame is dropped and re-read on demand."""
    from PIL import Image
    from dfanerf import frames
    H, W, N = 12, 16, 6
    rs = np.random.RandomState(0)
    imgs = [rs.randint(0, 256, (2, H, W, 3)).astype(np.uint8) for _ in range(N)]
    ph, pc = [], []
    for i, (a, b) in enumerate(imgs):
        ph.append(str(tmp_path / f"h{i}.png")); pc.append(str(tmp_path / f"c{i}.png"))
        Image.fromarray(a).save(ph[-1]); Image.fromarray(b).save(pc[-1])
    c = frames.DeviceFrameCache(ph, pc, H, W, torch.device("cpu"))
    assert c.preload(range(N), workers=3) == N and c.host_reads == 2 * N
    for i in (3, 0, 5, 3):
        h, cm = c.get(i)
        assert h.dtype == torch.uint8 and tuple(h.shape) == (H * W, 3)
        assert np.array_equal(h.numpy().reshape(H, W, 3), imgs[i][0]) and np.array_equal(cm.numpy().reshape(H, W, 3), imgs[i][1])
    assert c.host_reads == 2 * N                                        # warm: no file was read
    small = frames.DeviceFrameCache(ph, pc, H, W, torch.device("cpu"), budget_bytes=3 * 2 * H * W * 3)
    assert small.capacity == 3
    for i in (0, 1, 2, 0, 3):                                           # 3 evicts 1 (0 was used more recently)
        small.get(i)
    assert small.host_reads == 8 and set(small.slots) == {2, 0, 3}
    small.get(1)
    assert small.host_reads == 10 and np.array_equal(small.get(1)[0].numpy().reshape(H, W, 3), imgs[1][0])
    with pytest.raises(ValueError):
        frames.DeviceFrameCache(ph, pc, H + 1, W, torch.device("cpu")).get(0)


def test_bench_union_of_launch_intervals():
    """bench.py's per-launch kernel time under the two-stream N > 1 schedule: co-running launches are counted for the time
    the kernel occupied the GPU (union of the intervals), not for the sum of their own durations."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    u = bench.union_length
    assert u([]) == 0.0
    assert u([(0.0, 1.0), (2.0, 3.5)]) == 2.5                               # one stream: the sum
    assert u([(2.0, 3.5), (0.0, 1.0)]) == 2.5                               # any order
    assert u([(0.0, 2.0), (1.0, 3.0), (2.5, 4.0)]) == 4.0                   # two streams, every launch shares half its time
    assert u([(0.0, 10.0), (1.0, 2.0), (3.0, 4.0)]) == 10.0                 # nested
    # two alternating streams, 8 launches of 66 ms that start every 33 ms: 33 ms per launch plus the last one's tail
    iv = [(33.0 * i, 33.0 * i + 66.0) for i in range(8)]
    assert abs(u(iv) / len(iv) - (33.0 * 7 + 66.0) / 8) < 1e-9
