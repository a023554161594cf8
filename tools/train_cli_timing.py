#!/usr/bin/env python3
"""Developer tool (GPU box): step time of the REAL training CLI (NeRFs/DFANeRF/run_nerf_com_trainExpLater.py with the
flag bundle of scripts/train_obama.sh) on a synthetic 450x450 dataset written in the reference's on-disk format, next to
`bench.py --workload c4` - the input stage (frames.py) must keep the loop GPU-bound: no image decode, no host-to-device
copy per step once the frames are resident.   python tools/train_cli_timing.py [steps] [bf16|f32]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import numpy as np
import torch
from PIL import Image
from dfanerf import nets, run_nerf, synth
from dfanerf.decoder import Decoder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tier = sys.argv[2] if len(sys.argv) > 2 else "bf16"
H = W = 450
F_TRAIN, F_VAL = 12, 2
root = tempfile.mkdtemp(prefix="dfn_cli_")
d = os.path.join(root, "dataset", "obama")
for sub in ("head_imgs", "com_imgs"):
    os.makedirs(os.path.join(d, sub))
sc = synth.bench_scene(0, n_frames=F_TRAIN + F_VAL)
st = synth.synth_all_states(0)
zs, za = synth.synth_latents(0)
rng = np.random.RandomState(0)
Image.fromarray(sc["bg"]).save(os.path.join(d, "bc.jpg"), quality=95)
t = lambda x: torch.from_numpy(np.asarray(x))
for split, ids in (("train", range(F_TRAIN)), ("val", range(F_TRAIN, F_TRAIN + F_VAL))):
    frames = []
    for i in ids:
        frames.append({"img_id": i, "aud_id": i, "transform_matrix": sc["poses"][i].tolist(), "face_rect": [100, 120, 150, 160]})
        for sub in ("head_imgs", "com_imgs"):
            Image.fromarray(rng.randint(0, 255, (H, W, 3), dtype=np.uint8)).save(os.path.join(d, sub, f"{i:06d}.jpg"))
    json.dump({"focal_len": sc["focal"], "cx": sc["cx"], "cy": sc["cy"], "frames": frames},
              open(os.path.join(d, f"transforms_{split}_ba.json"), "w"))
torch.save(t(sc["aud"]), os.path.join(d, "obama_aud.pt"))
torch.save({"exp_o": t(sc["exp"])}, os.path.join(d, "obama_64_32.pt"))
open(os.path.join(d, "HeadNeRF_config_ba.txt"), "w").write(
    "expname = obama_head\ndatadir = dataset/obama\nbasedir = dataset/obama/logs\nnear = 0.3\nfar = 0.9\ntestskip = 1\n")
mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
        "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
        "PoseAttNet": nets.AudioAttNet(42, 8)}
for k, m in mods.items():
    m.load_state_dict({kk: t(v) for kk, v in st[k].items()})
opts = {k: torch.optim.Adam(m.parameters(), lr=5e-4) for k, m in mods.items()}
ck = os.path.join(root, "dataset", "train_together", "obama_TrainExpLater_smoMix")
os.makedirs(ck)
run_nerf.save_checkpoint(os.path.join(ck, "300000.tar"), 300000, t(zs), t(za), mods, opts)
flags = ("--config dataset/obama/HeadNeRF_config_ba.txt --last_dist=1e10 --datadir dataset/obama --concate_bg --N_rand=2048 "
         "--sample_rate=0 --i_print=1000000 --i_test_person=100000000 --chunk=2048 --win_size=16 --smo_size=4 "
         "--smo_torse_size 8 --train_together --i_weights=100000000 --all_speaker --sample_rate_mouth=0 --lrate_decay=500 "
         "--lrate=5e-4 --use_et_embed --nosmo_iters=300000 --dim_signal=96 --dim_aud=96 --n_object=1 "
         "--expname=obama_TrainExpLater_smoMix --aud_file=obama_aud.pt --use_deformation_field --exp_file=obama_64_32.pt "
         f"--use_ba --noexp_iters 400000 --hip_tier {tier} "
         "--resume dataset/train_together/obama_TrainExpLater_smoMix/300000.tar").split()
script = os.path.join(ROOT, "NeRFs", "DFANeRF", "run_nerf_com_trainExpLater.py")
res = {}
for n in (100, 100, 100 + steps):                  # a discarded warm-up run (file cache, code objects), then two run lengths:
    t0 = time.perf_counter()                       # their difference is `steps` steady-state steps
    r = subprocess.run([sys.executable, script] + flags + [f"--N_iters={300000 + n}"], cwd=root, capture_output=True, text=True)
    res[n] = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
ms = (res[100 + steps] - res[100]) / steps * 1e3
print(json.dumps({"train_cli_ms_per_step": ms, "steps": steps, "N_rand": 2048, "tier": tier,
                  "wall_s": {str(k): round(v, 2) for k, v in res.items()},
                  "note": "scripts/train_obama.sh flag bundle, 450x450 synthetic dataset, steady-state steps = the "
                          "difference of two run lengths (start-up, frame decode and the checkpoint load cancel)"}))
