#!/usr/bin/env python3
"""Developer experiment (CPU, no GPU needed): rendered-RGB accuracy of MIXED operand tiers of the decoder, emulated in torch.

VERDICT r02 item 3 asks for one decisive experiment on the headline kernel: bf16 MFMA operands for the seven trunk layers
(the bf16 instruction stream clocks ~5 % higher than the f16 one under the power limit) and f16 only "where the error
enters" (PE input layer + skip, sigma / rgb heads) - to be kept only if the full-frame PSNR against the exact tier stays
>= 49.4 dB (the north star's 0.05 dB clause, LABNOTES.md 3).  Before building that kernel this script measures what it could
deliver: the row-H pipeline of the oracle (coarse pass -> sample_pdf -> merged pass -> compositing; C2 geometry, head field,
64 + 128) with the decoder's GEMM operands (activations AND weights) rounded per layer to bf16 / f16 exactly as the kernel's
tiers do (f32 accumulation, f32 folded biases), on a strided subset of the 450 x 450 frame; PSNR against the same pipeline
with f32 operands.  The pure tiers calibrate the emulation against the GPU measurements (f16 56.5 dB, bf16 47.0 dB)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("dfa-nerf_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
import torch.nn.functional as F
import dfa_oracle as O
from dfanerf import synth

DT = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}


def q(x, t):
    return x if DT[t] is None else x.to(DT[t]).float()


def make_decoder(types):
    """types: operand type of the GEMMs 'in', 'blk0'..'blk6', 'skip', 'view' (feat_view + sigma_out + fc_view), 'out'"""
    def lin(P, name, x, t, cols=None):
        W = P[name + ".weight"] if cols is None else P[name + ".weight"][:, cols[0]:cols[1]]
        return q(x, t) @ q(W, t).T

    def fwd(P, p_in, ray_d, z_shape, z_app, signal, head_or_torso, return_intermediate=False):
        assert head_or_torso == "head"
        sig = signal[0]
        pe = O.posenc(p_in, O.N_FREQ_P)
        # per-frame constants are folded into f32 bias vectors by the fold kernel
        b_in = F.linear(sig, P["fc_in.weight"][:, 60:], P["fc_in.bias"]) + O._lin(P, "fc_z", z_shape)
        b_sk = F.linear(sig, P["fc_p_skips.0.weight"][:, 60:], P["fc_p_skips.0.bias"]) + O._lin(P, "fc_z_skips.0", z_shape)
        net = F.relu(lin(P, "fc_in", pe, types["in"], (0, 60)) + b_in)
        for i in range(7):
            net = F.relu(lin(P, f"blocks.{i}", net, types[f"blk{i}"]) + P[f"blocks.{i}.bias"])
            if i == 3:
                net = net + b_sk + lin(P, "fc_p_skips.0", pe, types["skip"], (0, 60))
        sigma = (lin(P, "sigma_out", net, types["view"]) + P["sigma_out.bias"]).squeeze(-1)
        d = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
        h = lin(P, "feat_view", net, types["view"]) + P["feat_view.bias"] + O._lin(P, "fc_z_view", z_app).unsqueeze(1) + \
            lin(P, "fc_view", O.posenc(d, O.N_FREQ_V), types["view"]) + P["fc_view.bias"]
        feat = torch.sigmoid(lin(P, "feat_out", F.relu(h), types["out"]) + P["feat_out.bias"])
        return feat, sigma
    return fwd


def tiers(trunk, rest, trunk_layers=range(7)):
    t = {k: rest for k in ("in", "skip", "view", "out")}
    t.update({f"blk{i}": (trunk if i in trunk_layers else rest) for i in range(7)})
    return t


def psnr(a, b):
    return float(-10 * torch.log10(((a.double() - b.double()) ** 2).mean()))


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    sc = synth.bench_scene(0, n_frames=8)
    st = synth.synth_all_states(0)
    zs, za = [torch.from_numpy(v) for v in synth.synth_latents(0)]
    P = O.params_to_torch(st["decoder"])
    nets = {k: O.params_to_torch(v) for k, v in st.items() if k != "decoder"}
    t = lambda x: torch.from_numpy(np.asarray(x))
    H, W = sc["H"], sc["W"]
    idx = torch.arange(0, H * W, (H * W) // n_rays)[:n_rays]
    with torch.no_grad():
        sig = O.encode_signal(nets, t(sc["aud"]), t(sc["exp"]), 0, 300000, 300000, 4, 8)
        o_h, d_h = O.get_rays(H, W, sc["focal"], sc["poses"][0][:3, :4], sc["cx"], sc["cy"])
        rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h)]
        bg = (t(sc["bg"]).float() / 255.0).reshape(-1, 3)[idx]
        variants = [("f32 operands (reference)", tiers("f32", "f32")),
                    ("f16 everywhere (the shipping throughput tier)", tiers("f16", "f16")),
                    ("bf16 everywhere (the training tier)", tiers("bf16", "bf16")),
                    ("MIXED: bf16 trunk (blocks 0-6), f16 input / skip / view / out", tiers("bf16", "f16")),
                    ("MIXED: bf16 blocks 1-5 only", tiers("bf16", "f16", range(1, 6))),
                    ("MIXED: bf16 blocks 0-3 only", tiers("bf16", "f16", range(0, 4))),
                    ("bf16 everywhere except f16 input + skip", {**tiers("bf16", "bf16"), "in": "f16", "skip": "f16"})]
        out = {}
        keep = O.decoder_forward
        try:
            for name, ty in variants:
                O.decoder_forward = make_decoder(ty)
                t0 = time.time()
                chunks = []
                for b in range(0, n_rays, 2048):
                    rh, _ = O.render_rays_chunk(P, rays[0][b:b + 2048], rays[1][b:b + 2048], rays[0][b:b + 2048],
                                                rays[1][b:b + 2048], bg[b:b + 2048], sc["near"], sc["far"], zs, za, sig, None,
                                                64, 128, 1)
                    chunks.append(rh)
                out[name] = torch.cat(chunks)
                ref = out[variants[0][0]]
                n_bf16 = sum(v == "bf16" for k, v in ty.items() if k.startswith("blk")) * 128 + \
                    (32 if ty["in"] == "bf16" else 0) + (32 if ty["skip"] == "bf16" else 0)      # of 1138 MFMAs (head)
                print(f"{name:66s}: PSNR vs f32 {psnr(out[name], ref):6.2f} dB   max |d| {float((out[name] - ref).abs().max()):.2e}   "
                      f"bf16 MFMAs {n_bf16}/1138   ({time.time() - t0:.0f} s)", flush=True)
        finally:
            O.decoder_forward = keep


if __name__ == "__main__":
    main()
