import os, sys
R = "/root/repo"
for d in ("tests", "dfa-nerf_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
from dfanerf import synth, nets, run_nerf, training
import test_gpu_train as T
t = T.t
scene = synth.bench_scene(0, n_frames=8); states = synth.synth_all_states(0); latents = synth.synth_latents(0)
dev = torch.device("cuda")
step, n = 300000, 2048
H, W = scene["H"], scene["W"]
flat_px = np.random.RandomState(11).permutation(H * W)[:n]
sel = np.stack([flat_px // W, flat_px % W], axis=1).astype(np.int64)
tgt_h = t(synth.synth_tensor(0, "g8/th", (H, W, 3), 0.5)) + 0.5
tgt_c = t(synth.synth_tensor(0, "g8/tc", (H, W, 3), 0.5)) + 0.5
mods = T._modules(states, dev)
args = run_nerf.config_parser().parse_args(
    "--expname t --concate_bg --N_rand=2048 --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
    "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000".split())
ds = [{"auds": t(scene["aud"]).to(dev), "exp": t(scene["exp"]).to(dev), "poses": t(scene["poses"]).to(dev),
       "bc_img": (t(scene["bg"]).float() / 255.0).to(dev), "hwfcxy": [H, W, scene["focal"], scene["cx"], scene["cy"]],
       "near": 0.3, "far": 0.9}]
zs, za = [t(v).to(dev) for v in latents]
embed_fn, _ = nets.get_embedder(3, 0)
buf = training.TrainBuffers("bf16", n, dev)
ys, xs = t(sel[:, 0]).to(dev), t(sel[:, 1]).to(dev)
sig = nets.encode_signal(ds, 0, 3, 96, mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], step, args, 8, embed_fn=embed_fn)
loss, lh, lc, _, _ = run_nerf.train_step_loss_hip(mods, ds, 0, 3, sel, tgt_h.to(dev)[ys, xs], tgt_c.to(dev)[ys, xs], zs, za,
                                                  step, args, 8, embed_fn, ds[0]["poses"][0], buf)
loss.backward(); torch.cuda.synchronize()
for f in (0, 1):
    for name, arr in (("dy", buf.dy[f]), ("act", buf.act[f])):
        rows = (arr.shape[1] - 128) // 32
        nb = rows // 32
        data = arr[:, :rows * 32].reshape(-1, nb, 1024)
        nz = ((data & 0x7f) != 0)
        blk_nz = nz.any(-1)                 # [tiles, nb]
        tile_nz = blk_nz.any(-1)
        print(f"field {f} {name}: elements nonzero {nz.float().mean():.3f}; 32x32 blocks with any nonzero {blk_nz.float().mean():.3f}; "
              f"whole tiles with any nonzero {tile_nz.float().mean():.3f}; tile pairs {tile_nz.reshape(-1, 2).any(-1).float().mean():.3f}")
        if name == "dy":
            # blocks whose values are all below 2^-6 of the tile-pair... relative magnitude: block amax (dequantised) vs global max
            sc = torch.exp2(arr[:, rows * 32:rows * 32 + nb].float() - 127.0)
            v = data.view(torch.float8_e4m3fn).float().abs().amax(-1) * sc          # [tiles, nb]
            g = v.max()
            for thr in (1e-3, 1e-4, 1e-6):
                print(f"      blocks with amax >= {thr:g} x global max: {(v >= thr * g).float().mean():.3f}; tiles: {(v.amax(-1) >= thr * g).float().mean():.3f}")
