"""End-to-end GPU tests of the drop-in driver: the reference's command lines (scripts/test_obama.sh and
scripts/train_obama.sh flag bundles) run against NeRFs/DFANeRF/run_nerf_com_trainExpLater.py of this repo on a
small synthetic dataset written in the reference's on-disk format."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import dfa_oracle as O
from conftest import ROOT
from dfanerf import synth

pytestmark = pytest.mark.gpu
H, W, F_TRAIN, F_VAL = 40, 56, 5, 3


def t(x):
    return torch.from_numpy(np.asarray(x))


@pytest.fixture(scope="module")
def dataset(tmp_path_factory, states, latents):
    from PIL import Image
    from dfanerf import nets, run_nerf
    from dfanerf.decoder import Decoder
    root = tmp_path_factory.mktemp("run")
    d = root / "dataset" / "obama"
    for sub in ("head_imgs", "com_imgs"):
        (d / sub).mkdir(parents=True)
    sc = synth.bench_scene(0, n_frames=F_TRAIN + F_VAL, H=H, W=W)
    rng = np.random.RandomState(0)
    Image.fromarray(sc["bg"]).save(d / "bc.jpg", quality=100, subsampling=0)
    for split, ids in (("train", range(F_TRAIN)), ("val", range(F_TRAIN, F_TRAIN + F_VAL))):
        frames = []
        for i in ids:
            frames.append({"img_id": i, "aud_id": i, "transform_matrix": sc["poses"][i].tolist(),
                           "face_rect": [5, 8, 20, 24]})
            for sub in ("head_imgs", "com_imgs"):
                Image.fromarray(rng.randint(0, 255, (H, W, 3), dtype=np.uint8)).save(d / sub / f"{i:06d}.jpg")
        json.dump({"focal_len": 150.0, "cx": W / 2.0, "cy": H / 2.0, "frames": frames},
                  open(d / f"transforms_{split}_ba.json", "w"))
    torch.save(t(sc["aud"]), d / "obama_aud.pt")
    torch.save({"exp_o": t(sc["exp"])}, d / "obama_64_32.pt")
    (d / "HeadNeRF_config_ba.txt").write_text("expname = obama_head\ndatadir = dataset/obama\n"
                                              "basedir = dataset/obama/logs\nnear = 0.3\nfar = 0.9\ntestskip = 1\n")
    # a checkpoint in the reference's .tar format
    mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
            "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in states[k].items()})
    opts = {k: torch.optim.Adam(m.parameters(), lr=5e-4) for k, m in mods.items()}
    ck = root / "dataset" / "train_together" / "obama_TrainExpLater_smoMix"
    ck.mkdir(parents=True)
    run_nerf.save_checkpoint(str(ck / "280000.tar"), 280000, t(latents[0]), t(latents[1]), mods, opts)
    return root, sc


COMMON = ("--config dataset/obama/HeadNeRF_config_ba.txt --last_dist=1e10 --datadir dataset/obama --concate_bg "
          "--sample_rate=0 --i_print=1 --i_test_person=10000 --chunk=2048 --win_size=16 --smo_size=4 "
          "--smo_torse_size 8 --train_together --all_speaker --sample_rate_mouth=0 --lrate_decay=500 --lrate=5e-4 "
          "--use_et_embed --nosmo_iters=300000 --dim_signal=96 --dim_aud=96 --n_object=1 "
          "--expname=obama_TrainExpLater_smoMix --aud_file=obama_aud.pt --use_deformation_field "
          "--exp_file=obama_64_32.pt --use_ba --noexp_iters 400000 "
          "--resume dataset/train_together/obama_TrainExpLater_smoMix/280000.tar")


def _run(root, extra, common=COMMON):
    cmd = [sys.executable, os.path.join(ROOT, "NeRFs", "DFANeRF", "run_nerf_com_trainExpLater.py")] + \
        (common + " " + extra).split()
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def _run2(root, extra, world=2):
    """the same CLI as `world` ranks (torch.distributed.run), all on GPU 0 over gloo (DFN_ONE_GPU)"""
    from conftest import free_port
    for attempt in range(2):          # one retry: a free port can be taken between the probe and the rendezvous
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()),
               os.path.join(ROOT, "NeRFs", "DFANeRF", "run_nerf_com_trainExpLater.py")] + (COMMON + " " + extra).split()
        r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=1500,
                           env=dict(os.environ, DFN_ONE_GPU="1", OMP_NUM_THREADS="4"))
        if os.environ.get("DFN_TEST_LOG_DIR"):          # (debugging aid: the ranks' complete output)
            with open(os.path.join(os.environ["DFN_TEST_LOG_DIR"], f"run2_world{world}_{attempt}_rc{r.returncode}.txt"), "a") as f:
                f.write(r.stdout + "\n==== stderr\n" + r.stderr)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def _oracle_frame_u8(root, sc, states, latents, k, n_coarse=64):
    """val frame k of the dataset through the oracle's frame loop -> uint8 (head, com) images"""
    from PIL import Image
    P = O.params_to_torch(states["decoder"])
    nets_o = {kk: O.params_to_torch(v) for kk, v in states.items() if kk != "decoder"}
    auds, exps, poses = [t(sc[kk])[F_TRAIN:] for kk in ("aud", "exp", "poses")]
    bg = t(np.asarray(Image.open(root / "dataset" / "obama" / "bc.jpg").convert("RGB"))).float() / 255.0
    pose_body = sc["poses"][0]                               # frame 0 of transforms_train_ba.json (MAIN:453-460)
    with torch.no_grad():
        sig = O.encode_signal(nets_o, auds, exps, k, 280000, 300000, 4, F_VAL)
        sigt = O.encode_signal_torso(nets_o, poses, k, 280000, 300000, 8, F_VAL)
        rh, rc = O.render_frame(P, H, W, 150.0, W / 2.0, H / 2.0, poses[k].numpy(), pose_body, bg, 0.3, 0.9,
                                t(latents[0]), t(latents[1]), sig, sigt, n_coarse, 0, 2)
    return O.to8b(rh.reshape(H, W, 3).numpy()), O.to8b(rc.reshape(H, W, 3).numpy())


def test_render_person_cli(dataset, states, latents):
    """scripts/test_obama.sh: --render_person over transforms_val_ba.json.  The frames are written as PNG (--image_ext:
    the kernel's uint8 output, losslessly) and compared with the oracle's frame loop + to8b: identical up to +-1 LSB on
    <= 0.2 % of the values - a wrong pose, signal window or background would move every pixel.  A second
    run writes the reference's .jpg files (same names, same count)."""
    from PIL import Image
    root, sc = dataset
    _run(root, "--render_person --test_file transforms_val_ba.json --N_rand=2048 --N_iters=600000 --image_ext png")
    out = root / "dataset" / "train_together" / "obama_TrainExpLater_smoMix" / "obama" / "person"
    files = sorted(os.listdir(out / "render_com"))
    assert files == [f"test_{i:06d}.png" for i in range(F_VAL)] and len(os.listdir(out / "render_head")) == F_VAL
    for k in (0, 1, F_VAL - 1):                              # first / last frame: zero-padded attention windows
        ref_h, ref_c = _oracle_frame_u8(root, sc, states, latents, k)
        for sub, ref in (("render_com", ref_c), ("render_head", ref_h)):
            img = np.asarray(Image.open(out / sub / f"test_{k:06d}.png").convert("RGB"))
            d = np.abs(img.astype(int) - ref.astype(int))
            assert d.max() <= 1 and (d > 0).mean() <= 2e-3, (k, sub, int(d.max()), float((d > 0).mean()))
    _run(root, "--render_person --test_file transforms_val_ba.json --N_rand=2048 --N_iters=600000 --render_video")
    assert sorted(f for f in os.listdir(out / "render_com") if f.endswith(".jpg")) == \
        [f"test_{i:06d}.jpg" for i in range(F_VAL)]
    img = np.asarray(Image.open(out / "render_com" / "test_000001.jpg")).astype(np.float32)
    ref = np.asarray(Image.open(out / "render_com" / "test_000001.png").convert("RGB")).astype(np.float32)
    assert np.abs(img - ref).mean() < 6.0                    # JPEG quality 95 of the same image (noisy background)
    # --N_samples is free upstream (MAIN:612-619); round 6: 32 and 128 render too - the CLI's frame against the oracle's loop at 32
    _run(root, "--render_person --test_file transforms_val_ba.json --N_rand=2048 --N_iters=600000 --image_ext png --N_samples 32")
    ref_h, ref_c = _oracle_frame_u8(root, sc, states, latents, 1, n_coarse=32)
    for sub, ref in (("render_com", ref_c), ("render_head", ref_h)):
        img = np.asarray(Image.open(out / sub / "test_000001.png").convert("RGB"))
        d = np.abs(img.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() <= 2e-3, ("N_samples 32", sub, int(d.max()), float((d > 0).mean()))


def test_render_person_cli_on_a_generate_test_jsons_file(dataset):
    """a test json in the format data_util/generate_test_jsons.py writes for driving the model with
    another audio track (frames copied from the source json with img_id / aud_id renumbered and the pose differences
    scaled by --param_scale, every other key kept) + its audio feature file: the loader and the renderer take it."""
    root, sc = dataset
    d = root / "dataset" / "obama"
    src = json.load(open(d / "transforms_val_ba.json"))
    n = len(src["frames"])
    arr = np.array([f["transform_matrix"] for f in src["frames"]], dtype=np.float32)
    diff = (arr[1:] - arr[:-1]) * 0.5                        # param_scale 0.5
    for i in range(n - 1):
        arr[i + 1] = arr[i] + diff[i]
    out = dict(src)
    out["frames"] = []
    for i in range(n):
        fr = dict(src["frames"][i])
        fr["transform_matrix"], fr["img_id"], fr["aud_id"] = arr[i].tolist(), i, i
        out["frames"].append(fr)
    json.dump(out, open(d / "transform_val_other.json", "w"))
    torch.save(t(sc["aud"])[:n].clone(), d / "other.pt")
    torch.save({"exp_o": t(sc["exp"])[:n].clone()}, d / "other_exp.pt")
    _run(root, "--render_person --test_file transform_val_other.json --aud_file other.pt --exp_file other_exp.pt "
               "--N_rand=2048 --N_iters=600000 --image_ext png --expname other_track "
               "--resume dataset/train_together/obama_TrainExpLater_smoMix/280000.tar")
    res = root / "dataset" / "train_together" / "other_track" / "obama" / "person" / "render_com"
    assert sorted(os.listdir(res)) == [f"test_{i:06d}.png" for i in range(n)]


def test_training_cli_writes_reference_checkpoint(dataset):
    """scripts/train_obama.sh: three optimisation steps from the checkpoint, then a new .tar with the reference's keys."""
    root, _ = dataset
    out = _run(root, "--N_rand=512 --N_iters=280004 --i_weights=2")
    base = root / "dataset" / "train_together" / "obama_TrainExpLater_smoMix"
    log = open(base / "loss.txt").read().strip().split("\n")
    assert len(log) == 4 and all(ln.startswith("[TRAIN] Iter: 28000") for ln in log), out[-500:]
    ck = torch.load(base / "280002.tar", weights_only=False)
    assert ck["global_step"] == 280002 and "network_PoseAttNet_state_dict" in ck and len(ck) == 13
    assert all(np.isfinite(float(ln.split("Com Loss: ")[1].split()[0])) for ln in log)
    assert "ground-truth frame pairs decoded to the device" in out            # the device-resident input stage ran
    # the face-rect / lower-half pixel split (MAIN:786-817) on the device sampler, and the periodic test of MAIN:943-1077:
    # render | ground truth side by side under the reference's file names, PSNR line in loss.txt
    from PIL import Image
    _run(root, "--N_rand=512 --N_iters=280002 --i_weights=100000 --sample_rate=0.9 --i_test_person=280002 --image_ext png")
    tdir = base / "obama" / "person" / "test_280002"
    assert sorted(os.listdir(tdir)) == ["test_000.png", "test_head_000.png"]
    assert np.asarray(Image.open(tdir / "test_000.png")).shape == (H, 2 * W, 3)
    log = open(base / "loss.txt").read().strip().split("\n")
    assert log[-1].startswith("[TEST] Iter: 280002 Object: 0_person PSNR: ")
    # the hierarchical training variant through the CLI (--hierarchical: the step differentiates row H, 64 + 128 samples, in
    # the 16-bit tier with its MX-fp8 recorder); three steps, finite losses, a checkpoint
    _run(root, "--N_rand=256 --N_iters=280005 --i_weights=280005 --i_print=1 --hierarchical --N_importance 128 --hip_tier bf16 "
               "--expname hier_train")
    hb = root / "dataset" / "train_together" / "hier_train"
    hl = [ln for ln in open(hb / "loss.txt").read().strip().split("\n") if ln.startswith("[TRAIN]")]
    assert len(hl) >= 3 and all(np.isfinite(float(ln.split("Com Loss: ")[1].split()[0])) for ln in hl), hl[-3:]
    assert (hb / "280005.tar").exists()
    # --use_expression (MAIN:412, decoder.py:219): the decoder then carries the expression layer the reference registers and, for
    # the one person it trains, never evaluates - a fresh model (no checkpoint has the layer), three steps, and the new .tar holds
    # it next to the layers that were trained
    _run(root, "--N_rand=256 --N_iters=3 --i_weights=2 --i_print=1 --use_expression --expname expr_train",
         common=COMMON[:COMMON.index("--resume")])
    eb = root / "dataset" / "train_together" / "expr_train"
    el = [ln for ln in open(eb / "loss.txt").read().strip().split("\n") if ln.startswith("[TRAIN]")]
    assert len(el) >= 2 and all(np.isfinite(float(ln.split("Com Loss: ")[1].split()[0])) for ln in el), el[-3:]
    ck = torch.load(eb / "000002.tar", weights_only=False)
    sd = ck["network_decoder_state_dict"]
    assert "expnet.weight" in sd and tuple(sd["expnet.weight"].shape) == (256, 256) and "fc_in.weight" in sd


@pytest.mark.parametrize("world", [2, 8])
def test_cli_with_several_ranks_on_one_gpu(dataset, world):
    """torchrun --nproc-per-node 2 / 8 of the drop-in CLI (all ranks on GPU 0 over gloo): --render_person writes byte-identical
    PNG frames to the single-process run (ray shards + one gather per frame), and a short data-parallel training run
    (replica broadcast, per-rank frames and pixels, gradient bucket, gated optimizers) writes its checkpoint."""
    from PIL import Image
    root, sc = dataset
    out = root / "dataset" / "train_together" / "obama_TrainExpLater_smoMix" / "obama" / "person"
    _run(root, "--render_person --test_file transforms_val_ba.json --N_rand=2048 --N_iters=600000 --image_ext png")
    one = {sub: [np.asarray(Image.open(out / sub / f"test_{k:06d}.png")).copy() for k in range(F_VAL)]
           for sub in ("render_com", "render_head")}
    for sub in one:
        for f in os.listdir(out / sub):
            os.remove(out / sub / f)
    _run2(root, "--render_person --test_file transforms_val_ba.json --N_rand=2048 --N_iters=600000 --image_ext png", world)
    for sub in one:
        assert sorted(os.listdir(out / sub)) == [f"test_{k:06d}.png" for k in range(F_VAL)]
        for k in range(F_VAL):
            assert np.array_equal(np.asarray(Image.open(out / sub / f"test_{k:06d}.png")), one[sub][k]), (sub, k)
    log = _run2(root, "--N_rand=256 --N_iters=280006 --i_weights=3 --hip_tier bf16", world)
    ck = root / "dataset" / "train_together" / "obama_TrainExpLater_smoMix"
    assert (ck / "280005.tar").exists(), log[-1500:]          # i_weights=3: saved at loop indices 280002, 280005
    lines = [ln for ln in open(ck / "loss.txt").read().split("\n") if ln.startswith("[TRAIN] Iter: 280006")]
    assert lines and np.isfinite(float(lines[-1].split("Com Loss: ")[1].split()[0]))


def test_cli_trains_and_renders_a_narrower_decoder_without_the_deformation_field(dataset):
    """--n_feat, --z_dim and --use_deformation_field are free upstream (MAIN:372-375, 411); round 5's CLI took one combination.
    From scratch: four training steps of a 128-wide decoder with 64-wide latent codes and no deformation field (exact tier), a
    checkpoint in the reference's format with the NARROW shapes, and --render_person from it in the f16 tier (both guards pass)."""
    root, _ = dataset
    common = COMMON.split(" --resume ")[0].replace("--use_deformation_field ", "").replace("--expname=obama_TrainExpLater_smoMix", "--expname=narrow")
    assert "--resume" not in common and "--use_deformation_field" not in common
    _run(root, "--N_rand=256 --N_iters=4 --i_weights=2 --n_feat 128 --z_dim 64 --hip_tier f32", common=common)
    base = root / "dataset" / "train_together" / "narrow"
    ck = torch.load(base / "000004.tar", map_location="cpu", weights_only=False)
    sd = ck["network_decoder_state_dict"]
    assert tuple(sd["blocks.3.weight"].shape) == (128, 128) and tuple(sd["fc_z.weight"].shape) == (128, 64) and \
        tuple(sd["fc_in.weight"].shape) == (128, 156) and not any(k.startswith("deform_net.") for k in sd)
    log = [ln for ln in open(base / "loss.txt").read().strip().split("\n") if ln.startswith("[TRAIN]")]
    assert len(log) == 4 and all(np.isfinite(float(ln.split("Com Loss: ")[1].split()[0])) for ln in log)
    out = _run(root, "--render_person --test_file transforms_val_ba.json --N_iters=600000 --image_ext png --n_feat 128 --z_dim 64 "
                     "--hip_tier f16 --hierarchical --N_importance 128 --resume dataset/train_together/narrow/000004.tar", common=common)
    assert "f16 tier: accuracy on" in out
    frames = sorted(os.listdir(base / "obama" / "person" / "render_com"))
    assert frames == [f"test_{k:06d}.png" for k in range(F_VAL)]


def test_n_object_default_stops_where_upstream_stops(dataset):
    """--n_object 2 is the flag's default (MAIN:378): upstream builds ONE dataset (`datadir = [args.datadir]`, MAIN:449) and stops
    in the per-person setup loop (`datadir[i]`, MAIN:499) with IndexError - before a frame is rendered.  Accepted at parse time
    (round 5 refused it) and reproduced: same exception, same place, nothing rendered."""
    root, _ = dataset
    cmd = [sys.executable, os.path.join(ROOT, "NeRFs", "DFANeRF", "run_nerf_com_trainExpLater.py")] + \
        (COMMON + " --render_person --test_file transforms_val_ba.json --N_iters=600000 --expname two_persons --n_object=2").split()
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "IndexError: list index out of range" in r.stderr and "--n_object 2" in r.stderr, r.stderr[-1500:]
    assert "unsupported configuration" not in r.stderr
    res = root / "dataset" / "train_together" / "two_persons" / "obama" / "person" / "render_com"
    assert not res.exists() or not os.listdir(res)


def test_cli_eight_ranks_with_eight_hardware_queues_each(dataset, monkeypatch):
    """the functional one-GPU modes cap the hardware queues per process (16 / world) because eight ranks x eight
    queues on ONE device died in start-up copies of c10d-gloo's device-tensor broadcast.  Round 6 stages the replicas through the host
    under gloo (parallel.broadcast_replicas): with EIGHT queues per process forced, the eight-rank render + training run must pass too
    (before: 3 of 4 boxes failed; after: 8 of 8 runs passed, profiles/r06g_world8_bcast.txt, r06p_world8_q8_soak.txt)."""
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    root, sc = dataset
    out = root / "dataset" / "train_together" / "obama_TrainExpLater_smoMix" / "obama" / "person"
    _run2(root, "--render_person --test_file transforms_val_ba.json --N_rand=2048 --N_iters=600000 --image_ext png", 8)
    for sub in ("render_com", "render_head"):
        assert sorted(f for f in os.listdir(out / sub) if f.endswith(".png")) == [f"test_{k:06d}.png" for k in range(F_VAL)]
    log = _run2(root, "--N_rand=256 --N_iters=280004 --i_weights=3 --hip_tier bf16", 8)
    lines = [ln for ln in open(root / "dataset" / "train_together" / "obama_TrainExpLater_smoMix" / "loss.txt").read().split("\n")
             if ln.startswith("[TRAIN] Iter: 280004")]
    assert lines and np.isfinite(float(lines[-1].split("Com Loss: ")[1].split()[0])), log[-1500:]


def test_f16_tier_refuses_an_out_of_range_checkpoint_loudly(dataset, states, latents):
    """--hip_tier f16 on a checkpoint whose activations exceed half precision: the CLI stops with the guard's message before it
    writes a frame (it used to render NaN frames silently); the same command on the in-range checkpoint reports its calibration."""
    from dfanerf import nets, run_nerf
    from dfanerf.decoder import Decoder
    root, _ = dataset
    base = "--render_person --test_file transforms_val_ba.json --N_rand=2048 --N_iters=600000 --image_ext png --hip_tier f16"
    out = _run(root, base)
    assert "f16 tier: calibrated on" in out and "max |activation|" in out
    st = dict(states)
    st["decoder"] = synth.scale_head_activations(states["decoder"], 1.0e4)
    mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
            "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in st[k].items()})
    opts = {k: torch.optim.Adam(m.parameters(), lr=5e-4) for k, m in mods.items()}
    ck = root / "dataset" / "train_together" / "big_acts"
    ck.mkdir(parents=True, exist_ok=True)
    run_nerf.save_checkpoint(str(ck / "280000.tar"), 280000, t(latents[0]), t(latents[1]), mods, opts)
    cmd = [sys.executable, os.path.join(ROOT, "NeRFs", "DFANeRF", "run_nerf_com_trainExpLater.py")] + \
        (COMMON + " " + base + " --expname big_acts --resume dataset/train_together/big_acts/280000.tar").split()
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "F16RangeError" in r.stderr and "--hip_tier bf16" in r.stderr, r.stderr[-1500:]
    res = root / "dataset" / "train_together" / "big_acts" / "obama" / "person" / "render_com"
    assert not res.exists() or not os.listdir(res)
    _run(root, base.replace("--hip_tier f16", "--hip_tier bf16") + " --expname big_acts --resume dataset/train_together/big_acts/280000.tar")
    assert len(os.listdir(res)) == F_VAL
    # --hip_tier auto (round 6): the same checkpoint is rendered - in the exact tier, and the run says why; the in-range one stays in f16
    for f in os.listdir(res):
        os.remove(res / f)
    out = _run(root, base.replace("--hip_tier f16", "--hip_tier auto") + " --expname big_acts --resume dataset/train_together/big_acts/280000.tar")
    assert "--hip_tier auto: rendering in the exact tier" in out and len(os.listdir(res)) == F_VAL
    out = _run(root, base.replace("--hip_tier f16", "--hip_tier auto"))
    assert "rendering in the exact tier" not in out and "f16 tier: accuracy on" in out
