#!/usr/bin/env python3
"""Developer tool (GPU box): the training forward (dfn_train_fwd: the fused renderer with its recorder on, 2048 rays, both
fields) alone, in a loop.   python tools/time_fwd.py [bf16|f32]     (DFN_LIB selects a variant library)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import torch
from dfanerf import training, synth, engine
from dfanerf.decoder import Decoder
from dfanerf._lib import lib, check

tier = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
n = 2048
buf = training.TrainBuffers(tier, n, dev)
dec = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_all_states(0)["decoder"].items()})
dec.to(dev)
flat = buf.bind(dec)
sc = synth.bench_scene(0, n_frames=2)
zs, za = [torch.from_numpy(v).to(dev)[0, :2].contiguous() for v in synth.synth_latents(0)]
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sh, stt = torch.randn(96, device=dev) * 0.1, torch.randn(42, device=dev) * 0.1
bias_t = C.c_void_p(buf.bias.data_ptr() + 4 * buf.nb[0])
check(lib.dfn_train_prepare(buf.tier, p(flat), p(sh), p(stt), p(zs), p(za), p(buf.packed[0]), p(buf.packed[1]),
                            p(buf.packed_T[0]), p(buf.packed_T[1]), p(buf.bias), bias_t, st), "prepare")
H, W = sc["H"], sc["W"]
bg = (torch.from_numpy(sc["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
pix = (torch.arange(n, dtype=torch.int32, device=dev) * 97) % (H * W)
fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], sc["near"], sc["far"],
                       1e10, 0, n, 64, 0, 2, True)
rh, rc = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
call = lambda: check(lib.dfn_train_fwd(buf.tier, C.byref(fr), p(buf.packed[0]), p(buf.packed[1]), p(buf.bias), bias_t, p(bg),
                                       None, p(pix), p(rh), p(rc), p(buf.samples), p(buf.act[0]), p(buf.masks[0]),
                                       p(buf.act[1]), p(buf.masks[1]), st), "dfn_train_fwd")
for _ in range(5): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): call()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
gb = sum(buf.act[f].numel() * buf.act[f].element_size() for f in (0, 1)) / 1e9
print(f"{tier}: {us:.1f} us per launch, act_T {gb:.2f} GB -> {gb / us * 1e3:.2f} TB/s written; rgb checksum {float(rh.sum() + rc.sum()):.4f}")
