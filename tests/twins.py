"""ATen twins of the product's HIP paths - TEST INFRASTRUCTURE ONLY.

The product (dfa-nerf_amd/dfanerf) is single-backend: every number on the render / training path comes out of the HIP
kernels.  What lives here are the same computations written with torch ops, so that tests can (a) check the host logic
(optimizer gating, LR schedule, checkpoint layout) on CPU against golden G8 and (b) cross-check the HIP kernels against
an independent autograd implementation.  Nothing under dfa-nerf_amd/ imports this module.

Reference semantics (paths under /root/reference/NeRFs/DFANeRF/): decoder.py:277-349 (Decoder.forward),
run_nerf_com_trainExpLater.py:146-179 (composite_function, calc_volume_weights), :779-907 (the training forward)."""
import torch
import torch.nn.functional as F

from dfanerf.helpers import img2mse
from dfanerf.nets import encode_signal, encode_signal_torso


def decoder_forward_aten(dec, p_in, ray_d, z_shape, z_app, signal, head_or_torso):
    """Decoder.forward (decoder.py:277-349) in ATen ops on `dec`'s parameters (autograd-capable, any device)."""
    if head_or_torso == 'head':
        signal = signal[0]
    p = dec.transform_points(p_in)
    if signal is not None:
        p = torch.cat((p, signal.expand(p.shape[1], -1).unsqueeze(0)), -1)
    if head_or_torso == 'torso':
        if dec.use_deformation_field:
            p = deform_forward_aten(dec.deform_net, p) + p
        net, p_skip = dec.fc_in_torso(p), dec.fc_p_skips_torso
    elif signal is not None:
        net, p_skip = dec.fc_in(p), dec.fc_p_skips
    else:
        net, p_skip = dec.fc_in_listener(p), dec.fc_p_skips_listener
    net = F.relu(net + dec.fc_z(z_shape).unsqueeze(1))
    k = 0
    for idx, layer in enumerate(dec.blocks):
        net = F.relu(layer(net))
        if (idx + 1) in dec.skips and idx < len(dec.blocks) - 1:
            net = net + dec.fc_z_skips[k](z_shape).unsqueeze(1) + p_skip[k](p)
            k += 1
    sigma_out = dec.sigma_out(net).squeeze(-1)
    net = dec.feat_view(net) + dec.fc_z_view(z_app).unsqueeze(1)
    d = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
    net = F.relu(net + dec.fc_view(dec.transform_points(d, views=True)))
    feat_out = torch.sigmoid(dec.feat_out(net))
    return feat_out, sigma_out


def deform_forward_aten(d, x):
    """DeformationField_ori.forward (decoder.py:109-134)."""
    def branch(blocks, skips, skip_in, out):
        net, k = x, 0
        for idx, layer in enumerate(blocks):
            net = F.relu(layer(net))
            if (idx + 1) in d.skips and idx < len(blocks) - 1:
                net = net + skips[k](skip_in)
                k += 1
        return out(net)
    embed, signal = x[..., :d.dim_embed], x[..., -d.dim_signal:]
    return torch.cat((branch(d.blocks_embed, d.fc_embed_skips, embed, d.out_embed),
                      branch(d.blocks_signal, d.fc_signal_skips, signal, d.out_signal)), -1)


def composite_function_aten(sigma, feat):
    if sigma.shape[0] > 1:
        denom = torch.sum(sigma, dim=0, keepdim=True)
        denom = torch.where(denom == 0, torch.full_like(denom, 1e-4), denom)
        return torch.sum(sigma, dim=0), (feat * (sigma / denom).unsqueeze(-1)).sum(0)
    return sigma.squeeze(0), feat.squeeze(0)


def calc_volume_weights_aten(z_vals, ray_vector, sigma, last_dist=1e10):
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], last_dist)], dim=-1)
    dists = dists * torch.norm(ray_vector, dim=-1, keepdim=True)
    alpha = 1. - torch.exp(-(F.relu(sigma) + 1e-6) * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), (1. - alpha + 1e-10)], dim=-1), dim=-1)
    return alpha * trans[..., :-1]


def bump_last(sigma, on):
    """relu(sigma) with +1e-6 on the last sample of the last stacked field (MAIN:692-694), out of place."""
    sigma = F.relu(sigma)
    if not on:
        return sigma
    bump = torch.zeros(sigma.shape[-1], device=sigma.device, dtype=sigma.dtype)
    bump[-1] = 1e-6
    return torch.cat([sigma[:-1], sigma[-1:] + bump], 0)


def get_rays_aten(H, W, focal, c2w, cx, cy):
    xs = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    ys = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    dirs = torch.stack([(xs - cx) / focal, -(ys - cy) / focal, -torch.ones_like(xs)], -1)
    c2w = c2w.float().cpu()
    rd = torch.stack([(dirs[..., 0] * c2w[k, 0] + dirs[..., 1] * c2w[k, 1]) + dirs[..., 2] * c2w[k, 2]
                      for k in range(3)], -1)
    return c2w[:3, 3].expand(rd.shape), rd


def train_step_loss_aten(nets, dataset, itr_obj, img_i, sel_yx, target_head, target_com, z_shape, z_app, global_step,
                         args, len_train, embed_fn, pose_torso):
    """Forward of one training step (MAIN:779-907) on the selected pixels, in ATen ops (autograd-capable, CPU or GPU).
    Returns loss, loss_head, loss_com, rgb_head, rgb_com."""
    dec = nets["decoder"]
    dev = next(dec.parameters()).device
    poses, bc_img = dataset[itr_obj]['poses'], dataset[itr_obj]['bc_img']
    H, W, focal, cx, cy = dataset[itr_obj]['hwfcxy']
    H, W = int(H), int(W)
    N = sel_yx.shape[0]
    signal = encode_signal(dataset, itr_obj, img_i, args.dim_aud, nets["AudNet"], nets["ExpNet"], nets["AudAttNet"],
                           global_step, args, len_train, embed_fn=embed_fn)
    signal_torso = encode_signal_torso(dataset, itr_obj, img_i, nets.get("PoseAttNet"), global_step, args, len_train,
                                       embed_fn=embed_fn)
    ys, xs = torch.as_tensor(sel_yx[:, 0], device=dev), torch.as_tensor(sel_yx[:, 1], device=dev)
    t_vals = torch.linspace(0., 1., steps=args.N_samples, device=dev)
    z_vals = (dataset[itr_obj]['near'] * (1. - t_vals) + dataset[itr_obj]['far'] * t_vals).expand(N, args.N_samples)

    def rays(pose):
        ro, rd = [x.to(dev) for x in get_rays_aten(H, W, focal, pose, cx, cy)]
        ro, rd = ro[ys, xs], rd[ys, xs]
        p = (ro[..., None, :] + rd[..., None, :] * z_vals[..., :, None]).reshape(1, -1, 3)
        r = rd.unsqueeze(1).expand(N, args.N_samples, 3).reshape(1, -1, 3)
        return rd, p, r
    rd_h, p_h, r_h = rays(poses[img_i, :3, :4])
    rd_t, p_t, r_t = rays(pose_torso)
    bc_rgb = bc_img[ys, xs].reshape(1, N, 1, 3)
    feat_h, sig_h = decoder_forward_aten(dec, p_h, r_h, z_shape[:, itr_obj * 2], z_app[:, itr_obj * 2], signal, 'head')
    feat_t, sig_t = decoder_forward_aten(dec, p_t, r_t, z_shape[:, itr_obj * 2 + 1], z_app[:, itr_obj * 2 + 1],
                                         signal_torso, 'torso')
    sig_h, feat_h = sig_h.reshape(1, N, -1), feat_h.reshape(1, N, args.N_samples, -1)
    sig_t, feat_t = sig_t.reshape(1, N, -1), feat_t.reshape(1, N, args.N_samples, -1)
    if args.concate_bg:
        feat_h = torch.cat((feat_h[..., :-1, :], bc_rgb), dim=-2)
        sig_t = torch.cat((sig_t[..., :-1], torch.zeros_like(sig_t[..., -1:])), -1)
    sigma = bump_last(torch.stack([sig_h], 0), args.concate_bg)
    sigma_to = bump_last(torch.stack([sig_h, sig_t], 0), args.concate_bg)
    ssum, fw = composite_function_aten(sigma, torch.stack([feat_h], 0))
    ssum_t, fw_t = composite_function_aten(sigma_to, torch.stack([feat_h, feat_t], 0))
    w_h = calc_volume_weights_aten(z_vals.unsqueeze(0), rd_h.unsqueeze(0), ssum, last_dist=args.last_dist)
    w_c = calc_volume_weights_aten(z_vals.unsqueeze(0), rd_t.unsqueeze(0), ssum_t, last_dist=args.last_dist)
    rgb_head = torch.sum(w_h.unsqueeze(-1) * fw, dim=-2).squeeze(0)
    rgb_com = torch.sum(w_c.unsqueeze(-1) * fw_t, dim=-2).squeeze(0)
    l_head = img2mse(rgb_head, target_head)
    l_com = img2mse(rgb_com, target_com)
    return l_com + l_head, l_head, l_com, rgb_head, rgb_com


# ---- the bias fold as differentiable torch ops (twin of dfn_fold_bias / dfn_fold_bias_bwd) ----------------------------
def _perm(n, device):
    """blob order [tile][half][16] -> feature index."""
    e = torch.arange(n, device=device)
    return 32 * (e >> 5) + (e & 3) + 8 * ((e & 15) >> 2) + 4 * ((e >> 4) & 1)


def _pad(v, n):
    return torch.cat([v, v.new_zeros(n - v.shape[0])]) if v.shape[0] < n else v


def fold_bias_torch(dec, sig_head, sig_torso, z_shape, z_app):
    """Differentiable twin of dfn_fold_bias (dfn_misc.hip: fold_kernel): [head blob | torso blob].
    z_shape, z_app: [2,256] rows (head, torso)."""
    dev = z_shape.device
    p256, p288, p64, p32 = _perm(256, dev), _perm(288, dev), _perm(64, dev), _perm(32, dev)
    sh, st = sig_head.reshape(-1), sig_torso.reshape(-1)

    def trunk(zs, za, b_in, b_skip):
        fczv = dec.fc_z_view(za)
        view = torch.cat([dec.feat_view.bias + fczv + dec.fc_view.bias, _pad(dec.sigma_out.bias, 32)])
        parts = [b_in[p256]] + [dec.blocks[l].bias[p256] for l in range(4)] + [b_skip[p256]] + \
                [dec.blocks[l].bias[p256] for l in range(4, 7)] + [view[p288], _pad(dec.feat_out.bias, 32)[p32]]
        return torch.cat(parts)
    zs0, zs1, za0, za1 = z_shape[0], z_shape[1], z_app[0], z_app[1]
    head = trunk(zs0, za0,
                 dec.fc_in.bias + dec.fc_in.weight[:, 60:] @ sh + dec.fc_z(zs0),
                 dec.fc_z_skips[0](zs0) + dec.fc_p_skips[0].bias + dec.fc_p_skips[0].weight[:, 60:] @ sh)
    d = dec.deform_net
    dv = [d.blocks_embed[0].bias + d.blocks_embed[0].weight[:, 60:] @ st,
          d.blocks_signal[0].bias + d.blocks_signal[0].weight[:, 60:] @ st,
          d.blocks_embed[1].bias, d.blocks_signal[1].bias, d.blocks_embed[2].bias, d.blocks_signal[2].bias,
          d.blocks_embed[3].bias, d.fc_embed_skips[0].bias, d.blocks_signal[3].bias, d.fc_signal_skips[0](st),
          d.blocks_embed[4].bias, d.blocks_signal[4].bias, _pad(d.out_embed.bias, 64), _pad(d.out_signal.bias + st, 64)]
    torso = torch.cat([v[p64] for v in dv] +
                      [trunk(zs1, za1, dec.fc_in_torso.bias + dec.fc_z(zs1),
                             dec.fc_z_skips[0](zs1) + dec.fc_p_skips_torso[0].bias)])
    return torch.cat([head, torso])


class RenderTrainFn(torch.autograd.Function):
    """(flat params [955242], bias blob [head|torso]) -> rgb_head [n,3], rgb_com [n,3] for the selected pixels: the HIP
    forward / backward of the renderer WITHOUT the fused fold (its inputs are plain autograd tensors)."""

    @staticmethod
    def forward(ctx, flat, bias, buf, frame, bg, pix_index):
        import ctypes as C
        from dfanerf._lib import check, lib
        from dfanerf.engine import _ptr, _stream
        flat_c = flat.detach().contiguous()
        bias_c = bias.detach().contiguous()
        t, st = buf.tier, _stream()
        for f in (0, 1):
            check(lib.dfn_pack_weights(t, f, _ptr(flat_c), _ptr(buf.packed[f]), st), "dfn_pack_weights")
            check(lib.dfn_pack_weights_bwd(t, f, _ptr(flat_c), _ptr(buf.packed_T[f]), st), "dfn_pack_weights_bwd")
        n = frame.ray_count
        rgb_h = torch.empty(n, 3, dtype=torch.float32, device=flat.device)
        rgb_c = torch.empty(n, 3, dtype=torch.float32, device=flat.device)
        bg_f32 = bg if bg.dtype == torch.float32 else None
        bg_u8 = bg if bg.dtype == torch.uint8 else None
        check(lib.dfn_train_fwd(t, C.byref(frame), _ptr(buf.packed[0]), _ptr(buf.packed[1]), _ptr(bias_c),
                                C.c_void_p(bias_c.data_ptr() + 4 * buf.nb[0]), _ptr(bg_f32), _ptr(bg_u8),
                                _ptr(pix_index), _ptr(rgb_h), _ptr(rgb_c), _ptr(buf.samples), _ptr(buf.act[0]),
                                _ptr(buf.masks[0]), _ptr(buf.act[1]), _ptr(buf.masks[1]), st), "dfn_train_fwd")
        ctx.buf, ctx.frame, ctx.bg, ctx.pix = buf, frame, bg, pix_index
        ctx.n_flat, ctx.dev = flat.numel(), flat.device
        return rgb_h, rgb_c

    @staticmethod
    def backward(ctx, d_h, d_c):
        import ctypes as C
        from dfanerf._lib import check, lib
        from dfanerf.engine import _ptr, _stream
        buf, frame, bg, st = ctx.buf, ctx.frame, ctx.bg, _stream()
        d_h = d_h.contiguous().float()
        d_c = d_c.contiguous().float()
        bg_f32 = bg if bg.dtype == torch.float32 else None
        bg_u8 = bg if bg.dtype == torch.uint8 else None
        check(lib.dfn_composite_bwd(C.byref(frame), _ptr(ctx.pix), _ptr(bg_f32), _ptr(bg_u8), _ptr(buf.samples),
                                    _ptr(d_h), _ptr(d_c), _ptr(buf.dsamples), st), "dfn_composite_bwd")
        g_flat = torch.zeros(ctx.n_flat, dtype=torch.float32, device=ctx.dev)
        g_bias = torch.empty(buf.nb[0] + buf.nb[1], dtype=torch.float32, device=ctx.dev)
        for f in (0, 1):
            check(lib.dfn_mlp_bwd(buf.tier, f, _ptr(buf.packed_T[f]), _ptr(buf.samples), _ptr(buf.dsamples),
                                  _ptr(buf.masks[f]), buf.NP, _ptr(buf.dy[f]), st), "dfn_mlp_bwd")
            check(lib.dfn_weight_grad(buf.tier, f, _ptr(buf.dy[f]), _ptr(buf.act[f]), buf.NP, _ptr(buf.ws[f]),
                                      _ptr(g_flat), st), "dfn_weight_grad")
            ws_b = torch.empty(check(lib.dfn_train_rows(f, 4), "rows"), dtype=torch.float32, device=ctx.dev)
            check(lib.dfn_bias_grad(buf.tier, f, _ptr(buf.dy[f]), buf.NP, _ptr(ws_b),
                                    C.c_void_p(g_bias.data_ptr() + (4 * buf.nb[0] if f else 0)), st), "dfn_bias_grad")
        return g_flat, g_bias, None, None, None, None


def render_train_unfused(dec, buf, frame, bg, pix_index, sig_head, sig_torso, z_shape, z_app):
    """The training render with the fold as differentiable torch ops around RenderTrainFn (HIP forward / backward of the
    renderer): the twin test_fused_fold_backward_matches_torch_fold compares FusedTrainFn with."""
    flat = torch.cat([p.reshape(-1) for p in dec.state_dict(keep_vars=True).values()])
    bias = fold_bias_torch(dec, sig_head, sig_torso, z_shape, z_app)
    return RenderTrainFn.apply(flat, bias, buf, frame, bg, pix_index)
