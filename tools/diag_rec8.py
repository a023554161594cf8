#!/usr/bin/env python3
"""Developer check (GPU box): the MX-fp8 arrays the 16-bit tier records (dfn_mlp.h "MX-fp8 recording") decoded in torch,
against the f32 tier's recorded arrays of the same forward / backward: act_T (decoder forward on points) and dy_T (dX chain)."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "dfa-nerf_amd", "oracle"):
    sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
from dfanerf import synth, training, engine
from dfanerf._lib import lib, check
from dfanerf.decoder import Decoder
dev = torch.device("cuda")
st = synth.synth_all_states(0)
zs, za = synth.synth_latents(0)
t = lambda x: torch.from_numpy(np.asarray(x))
dec = Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True)
dec.load_state_dict({k: t(v) for k, v in st["decoder"].items()})
dec.to(dev)
n = 256
g = torch.Generator().manual_seed(1)
pts = (torch.rand(n, 3, generator=g) - 0.5).to(dev)
dirs = torch.randn(n, 3, generator=g).to(dev)


def decode(arr, rows):
    nt = arr.shape[0]
    # per 32-row block: [point n][half h][register r] bytes, feature = (r & 3) + 8 (r >> 2) + 4 h
    raw = arr[:, :rows * 32].view(torch.float8_e4m3fn).float().view(nt, rows // 32, 32, 2, 16)
    r = torch.arange(16)
    feat = torch.stack([(r & 3) + 8 * (r >> 2) + 4 * h for h in (0, 1)])            # [2, 16]
    data = torch.empty(nt, rows // 32, 32, 32, device=arr.device)                  # [tile, block, feature, point]
    data[:, :, feat.reshape(-1).to(arr.device), :] = raw.permute(0, 1, 3, 4, 2).reshape(nt, rows // 32, 32, 32)
    data = data.reshape(nt, rows, 32)
    sc = torch.exp2(arr[:, rows * 32:rows * 32 + (rows + 31) // 32].float() - 127.0)           # [nt, rows/32]
    sc = sc.repeat_interleave(32, dim=1)[:, :rows]
    return (data * sc[..., None]).permute(1, 0, 2).reshape(rows, nt * 32)                      # [rows, NP]


for field, sig_n in ((0, 96), (1, 42)):
    sig = t(synth.synth_tensor(0, "g3/sig" if field == 0 else "g3/sigt", (sig_n,), 0.3)).to(dev)
    out = {}
    for tier in ("f32", "bf16"):
        tt = training.TIERS[tier]
        net = training._FlatNet.of(dec)
        net.refresh()
        pb = training._PointBuffers(tt, field, n, dev)
        zsd, zad = t(zs[0][field]).to(dev), t(za[0][field]).to(dev)
        stt = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda x: C.c_void_p(x.data_ptr())
        check(lib.dfn_fold_bias(tt, field, p(net.flat), p(sig), p(zsd), p(zad), p(pb.bias), stt), "fold")
        check(lib.dfn_pack_weights(tt, field, p(net.flat), p(pb.packed), stt), "pack")
        check(lib.dfn_pack_weights_bwd(tt, field, p(net.flat), p(pb.packed_T), stt), "packT")
        feat = torch.empty(n, 3, device=dev); sigma = torch.empty(n, device=dev)
        check(lib.dfn_decoder_train_fwd(tt, field, p(pb.packed), p(pb.bias), p(pts), p(dirs), n, p(feat), p(sigma), p(pb.samples),
                                        p(pb.act), p(pb.masks), stt), "fwd")
        ds = torch.zeros(pb.NP, 8, device=dev)
        gg = torch.Generator(device=dev).manual_seed(3)
        ds[:n, 4 * field:4 * field + 4] = torch.randn(n, 4, device=dev, generator=gg) * 1e-3
        check(lib.dfn_mlp_bwd(tt, field, p(pb.packed_T), p(pb.samples), p(ds), p(pb.masks), pb.NP, p(pb.dy), stt), "bwd")
        torch.cuda.synchronize()
        rows_a, rows_g = lib.dfn_train_rows(field, 0), lib.dfn_train_rows(field, 1)
        if tier == "f32":
            # f32 arrays are tile-major too: [tile][rows][32]
            a = pb.act.view(-1)[:pb.NP * rows_a].view(pb.NP // 32, rows_a, 32).permute(1, 0, 2).reshape(rows_a, pb.NP)
            d = pb.dy.view(-1)[:pb.NP * rows_g].view(pb.NP // 32, rows_g, 32).permute(1, 0, 2).reshape(rows_g, pb.NP)
        else:
            a, d = decode(pb.act, rows_a), decode(pb.dy, rows_g)
        out[tier] = (a[:, :n].cpu(), d[:, :n].cpu())
    for name, k in (("act_T", 0), ("dy_T", 1)):
        ref, got = out["f32"][k], out["bf16"][k]
        rows = ref.shape[0]
        print(f"field {field} {name}: rows {rows}")
        bad = []
        for r0 in range(0, rows, 32):
            a, b = ref[r0:r0 + 32], got[r0:r0 + 32]
            e = float((a - b).norm() / (a.norm() + 1e-30))
            # is it a row permutation inside the block?
            if e > 0.15:
                cc = (b @ a.T)
                match = cc.argmax(1).tolist()
                bad.append((r0, round(e, 2), match[:8]))
        print("   blocks with error > 15 %:", len(bad), "of", (rows + 31) // 32, bad[:6])
