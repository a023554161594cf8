"""CPU oracle for the DFA-NeRF volumetric-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path; it is
imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline
leg and by nothing else.  The product (dfa-nerf_amd/) never imports it and has
no CPU fallback.

It is a restatement, in this repo's own functional style (plain tensors and
dicts of parameters, no nn.Module copies), of the algorithm in the reference's
Python files.  Every function cites the reference file:line it follows, with
paths relative to /root/reference/NeRFs/DFANeRF/:
    MAIN = run_nerf_com_trainExpLater.py, HELP = run_nerf_helpers.py,
    DEC  = decoder.py.

Parity pinning: the reference ships no tests and no golden vectors (SURVEY.md
section 4).  This oracle is pinned against vectors produced by importing the
reference's own modules on CPU in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; checked by
tests/test_oracle_golden.py).  Row H (the 64+128 hierarchical mode) has no
caller in the reference: its composition is defined here and in DESIGN.md and
is pinned only through its components (sample_pdf, decoder, compositing).

Arithmetic is PyTorch CPU fp32, the same ATen kernels the reference calls, so
the comparison with the reference is bitwise for the elementwise parts and
within GEMM-reassociation noise for the linear layers.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

N_FREQ_P = 10      # DEC:167 n_freq_posenc
N_FREQ_V = 4       # DEC:168 n_freq_posenc_views


def T(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


def params_to_torch(state):
    return {k: T(v).float() for k, v in state.items()}


# --------------------------------------------------------------------------
# A1 / A2: rays
# --------------------------------------------------------------------------
def linspace0(end, n):
    """torch.linspace(0, end, n) restated like linspace01 below (ATen linspace_kernel): f32 step end/(n-1), lower half step*i, upper
    half end - step*(n-1-i) as one fused multiply-add.  n = end + 1 gives the integers exactly."""
    if n <= 1:
        return torch.zeros(max(n, 0), dtype=torch.float32)
    step = np.float64(np.float32(end) / np.float32(n - 1))
    i = np.arange(n)
    lo = (step * i).astype(np.float32)
    hi = (np.float64(np.float32(end)) - step * (n - 1 - i)).astype(np.float32)
    return torch.from_numpy(np.where(i < n // 2, lo, hi).astype(np.float32))


def get_rays(H, W, focal, c2w, cx=None, cy=None, stride=1):
    """HELP:449-465.  x = column, y = row; ray index y*W+x after reshape(-1,3).
    rays_d[k] = (dx*R[k,0] + dy*R[k,1]) + dz*R[k,2], separate f32 roundings.  stride (dead upstream): W//stride x H//stride rays
    through linspace(0, W-1, W//stride) x linspace(0, H-1, H//stride)."""
    c2w = T(c2w).float()
    cx = W * .5 if cx is None else cx
    cy = H * .5 if cy is None else cy
    Hn, Wn = H // stride, W // stride
    xs = linspace0(W - 1, Wn)[None, :].expand(Hn, Wn)
    ys = linspace0(H - 1, Hn)[:, None].expand(Hn, Wn)
    dx = (xs - cx) / focal
    dy = -(ys - cy) / focal
    dz = -torch.ones_like(dx)
    R = c2w[:3, :3]
    comps = [(dx * R[k, 0] + dy * R[k, 1]) + dz * R[k, 2] for k in range(3)]
    rays_d = torch.stack(comps, -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """HELP:484-503."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    kx = -1. / (W / (2. * focal))
    ky = -1. / (H / (2. * focal))
    o0 = kx * o[..., 0] / o[..., 2]
    o1 = ky * o[..., 1] / o[..., 2]
    o2 = 1. + 2. * near / o[..., 2]
    d0 = kx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = ky * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2. * near / o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------
# A3: coarse samples
# --------------------------------------------------------------------------
def linspace01(n):
    """torch.linspace(0,1,n) restated (ATen RangeFactoriesKernel.cpp linspace_kernel): the f32
    step 1/(n-1); lower half step*i; upper half 1 - step*(n-1-i) evaluated as ONE fused
    multiply-add (single rounding).  Pinned bit-for-bit by golden G2; torch.linspace itself is
    not called so that the oracle does not depend on the host CPU's vector ISA."""
    step = np.float64(np.float32(1.0) / np.float32(n - 1))
    i = np.arange(n)
    lo = (step * i).astype(np.float32)
    hi = (1.0 - step * (n - 1 - i)).astype(np.float32)
    return torch.from_numpy(np.where(i < n // 2, lo, hi).astype(np.float32))


def coarse_z(near, far, n_samples):
    """MAIN:617-618.  z = near*(1-t) + far*t; identical for every ray."""
    t = linspace01(n_samples)
    near_t = near * torch.ones(1)
    far_t = far * torch.ones(1)
    return near_t * (1. - t) + far_t * t


def ray_points(rays_o, rays_d, z):
    """MAIN:638-641.  p = o + d*z (mul then add), dirs broadcast per sample.
    rays_o/d [R,3], z [R,S] or [S] -> p [R,S,3]."""
    if z.dim() == 1:
        z = z[None, :].expand(rays_d.shape[0], -1)
    return rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]


# --------------------------------------------------------------------------
# A4: positional encoding
# --------------------------------------------------------------------------
def posenc(p, n_freq, downscale=2.0):
    """DEC:257-275 ('normal' mode).  p/2, then per octave [sin(c_i p)(3), cos(c_i p)(3)]
    where c_i = float32(2**i * pi) is rounded BEFORE the multiply."""
    p = p / downscale
    out = []
    for i in range(n_freq):
        c = (2 ** i) * math.pi          # python double; torch casts it to f32 for the op
        out.append(torch.sin(c * p))
        out.append(torch.cos(c * p))
    return torch.cat(out, -1)


# --------------------------------------------------------------------------
# A5 / A6: decoder
# --------------------------------------------------------------------------
def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def deformation_field(P, x, dim_embed=60, dim_signal=42):
    """DEC:109-134 (DeformationField_ori; n_blocks=7 -> 5 hidden layers, skip after idx 3)."""
    embed = x[..., :dim_embed]
    signal = x[..., -dim_signal:]
    outs = []
    for kind, skip_in in (("embed", embed), ("signal", signal)):
        net = x
        for idx in range(5):
            net = F.relu(_lin(P, f"deform_net.blocks_{kind}.{idx}", net))
            if idx == 3:
                net = net + _lin(P, f"deform_net.fc_{kind}_skips.0", skip_in)
        outs.append(_lin(P, f"deform_net.out_{kind}", net))
    return torch.cat(outs, -1)


def decoder_forward(P, p_in, ray_d, z_shape, z_app, signal, head_or_torso,
                    return_intermediate=False):
    """DEC:277-349 with use_expression=False, use_viewdirs=True, final sigmoid; use_deformation_field = whether `P` holds the
    deform_net tensors (DEC:297: `if self.use_deformation_field and head_or_torso == 'torso'`).
    p_in, ray_d [1,N,3]; z_* [1,z_dim]; signal: [aud[1,96], None] for 'head',
    [1,42] or [42] for 'torso'.  Returns feat [1,N,3], sigma [1,N] (raw)."""
    if head_or_torso == 'head':
        signal = signal[0]
    pe = posenc(p_in, N_FREQ_P)
    if signal is not None:
        sig = signal.expand(pe.shape[1], -1).unsqueeze(0)
        p = torch.cat((pe, sig), -1)
    else:
        p = pe
    if head_or_torso == 'torso':
        if any(k.startswith("deform_net.") for k in P):
            p = deformation_field(P, p) + p
        in_name, skip_name = "fc_in_torso", "fc_p_skips_torso.0"
    elif head_or_torso == 'head':
        if signal is not None:
            in_name, skip_name = "fc_in", "fc_p_skips.0"
        else:
            in_name, skip_name = "fc_in_listener", "fc_p_skips_listener.0"
    else:
        raise Exception('Do not give head or torso!!')
    net = _lin(P, in_name, p) + _lin(P, "fc_z", z_shape).unsqueeze(1)
    net = F.relu(net)
    for idx in range(7):
        net = F.relu(_lin(P, f"blocks.{idx}", net))
        if idx == 3:
            net = net + _lin(P, "fc_z_skips.0", z_shape).unsqueeze(1)
            net = net + _lin(P, skip_name, p)
    sigma = _lin(P, "sigma_out", net).squeeze(-1)
    net = _lin(P, "feat_view", net) + _lin(P, "fc_z_view", z_app).unsqueeze(1)
    d = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
    net = F.relu(net + _lin(P, "fc_view", posenc(d, N_FREQ_V)))
    feat = torch.sigmoid(_lin(P, "feat_out", net))
    if return_intermediate:
        return feat, sigma, pe
    return feat, sigma


# --------------------------------------------------------------------------
# A7 / A8: per-frame conditioning signals
# --------------------------------------------------------------------------
def _leaky(x):
    return F.leaky_relu(x, 0.02)


def audnet_w2l(P, x):
    """HELP:165-178: 512 -> 256 -> 128 -> 64."""
    x = _leaky(_lin(P, "encoder.0", x))
    x = _leaky(_lin(P, "encoder.2", x))
    return _lin(P, "encoder.4", x)


def expression_enc(P, x):
    """HELP:182-193: 64 -> 32 -> 32."""
    return _lin(P, "encoder.2", _leaky(_lin(P, "encoder.0", x)))


def audio_att_net(P, x, dim_aud, seq_len):
    """HELP:210-240.  x [seq_len, dim] -> attention-weighted sum over the window [dim]."""
    y = x[..., :dim_aud].permute(1, 0).unsqueeze(0)
    for k in range(5):
        y = _leaky(F.conv1d(y, P[f"attentionConvNet.{2 * k}.weight"],
                            P[f"attentionConvNet.{2 * k}.bias"], padding=1))
    a = F.softmax(_lin(P, "attentionNet.0", y.view(1, seq_len)), dim=1).view(seq_len, 1)
    return torch.sum(a * x, dim=0)


def _window(x, img_i, half, length):
    """MAIN:36-57 / 86-102: rows [img_i-half, img_i+half) clipped to [0,length), zero padded."""
    left, right = img_i - half, img_i + half
    pad_l = max(0, -left)
    pad_r = max(0, right - length)
    win = x[max(left, 0):min(right, length)]
    if pad_l:
        win = torch.cat((torch.zeros_like(win)[:pad_l], win), 0)
    if pad_r:
        win = torch.cat((win, torch.zeros_like(win)[:pad_r]), 0)
    return win


def encode_signal(nets, auds, exps, img_i, global_step, nosmo_iters, smo_size, len_auds):
    """MAIN:28-75, object 0.  Returns [aud[1,96], None]."""
    if global_step >= nosmo_iters:
        half = int(smo_size / 2)
        a = audnet_w2l(nets["AudNet"], _window(auds, img_i, half, len_auds))
        e = expression_enc(nets["ExpNet"], _window(exps, img_i, half, len_auds))
        aud = audio_att_net(nets["AudAttNet"], torch.cat([a, e], 1), 96, smo_size).unsqueeze(0)
    else:
        a = audnet_w2l(nets["AudNet"], auds[img_i:img_i + 1])
        e = expression_enc(nets["ExpNet"], exps[img_i:img_i + 1])
        aud = torch.cat([a, e], 1)
    return [aud, None]


def pose_to_euler_trans(poses):
    """MAIN:182-204.  e = [atan2(R22,R12), asin(-R02), atan2(R00,-R01)], t = pose[:3,3]."""
    R = poses
    e0 = torch.atan2(R[:, 2, 2], R[:, 1, 2])
    e1 = torch.asin(-R[:, 0, 2])
    e2 = torch.atan2(R[:, 0, 0], -R[:, 0, 1])
    return torch.cat((torch.stack([e0, e1, e2], 1), poses[:, :3, 3]), 1)


def embed3(x):
    """HELP:21-70 get_embedder(3,0): [x, sin(x), cos(x), sin(2x), cos(2x), sin(4x), cos(4x)] -> 21."""
    out = [x]
    for f in (1.0, 2.0, 4.0):
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, -1)


def encode_signal_torso(nets, poses, img_i, global_step, nosmo_iters, smo_torse_size, len_poses):
    """MAIN:78-111.  [1,42] before nosmo_iters, [42] after (shape quirk kept)."""
    if global_step >= nosmo_iters:
        half = int(smo_torse_size / 2)
        left, right = max(img_i - half, 0), min(img_i + half, len_poses)
        et = pose_to_euler_trans(poses[left:right])
        pad_l = max(0, half - img_i)
        pad_r = max(0, img_i + half - len_poses)
        if pad_l:
            et = torch.cat((torch.zeros_like(et)[:pad_l], et), 0)
        if pad_r:
            et = torch.cat((et, torch.zeros_like(et)[:pad_r]), 0)
        emb = torch.cat((embed3(et[:, :3]), embed3(et[:, 3:])), 1)
        return audio_att_net(nets["PoseAttNet"], emb, 42, smo_torse_size)
    et = pose_to_euler_trans(poses[img_i].unsqueeze(0))
    return torch.cat((embed3(et[:, :3]), embed3(et[:, 3:])), 1)


# --------------------------------------------------------------------------
# A9-A12: compositing
# --------------------------------------------------------------------------
def composite_function(sigma, feat):
    """MAIN:146-166.  sigma [K,1,C,S], feat [K,1,C,S,3]."""
    if sigma.shape[0] > 1:
        denom = torch.sum(sigma, dim=0, keepdim=True)
        denom = torch.where(denom == 0, torch.full_like(denom, 1e-4), denom)
        w = sigma / denom
        return torch.sum(sigma, dim=0), (feat * w.unsqueeze(-1)).sum(0)
    return sigma.squeeze(0), feat.squeeze(0)


def calc_volume_weights(z_vals, ray_vector, sigma, last_dist=1e10):
    """MAIN:169-179.  z [1,C,S], ray_vector [1,C,3], sigma [1,C,S] -> weights [1,C,S]."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], last_dist)], -1)
    dists = dists * torch.norm(ray_vector, dim=-1, keepdim=True)
    alpha = 1. - torch.exp(-(F.relu(sigma) + 1e-6) * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1. - alpha + 1e-10], -1), -1)
    return alpha * trans[..., :-1]


def to8b(x):
    """HELP:17: truncating conversion."""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def integrate_fields(z, d_head, d_torso, sig_h, feat_h, sig_t, feat_t, bg, last_dist=1e10,
                     concate_bg=True):
    """MAIN:667-709 for one chunk.  z [C,S]; d_* [C,3]; sig_* [C,S] raw; feat_* [C,S,3];
    bg [C,3] or None.  sig_t/feat_t may be None (single field).
    Returns rgb_head [C,3], w_head [C,S], rgb_com, w_com (None if single field)."""
    sig_h = sig_h[None]
    feat_h = feat_h[None]
    if concate_bg and bg is not None:
        feat_h = torch.cat((feat_h[..., :-1, :], bg[None, :, None, :]), dim=-2)
    # MAIN:693 writes `sigma[-1,:,:,-1] += 1e-6` in place on the relu output; current autograd
    # rejects that in backward (relu saves its result), so the same values are formed out of place.
    bump = torch.zeros(sig_h.shape[-1])
    if concate_bg:
        bump[-1] = 1e-6
    s1 = F.relu(torch.stack([sig_h], 0))
    s1 = torch.cat([s1[:-1], s1[-1:] + bump], 0)
    f1 = torch.stack([feat_h], 0)
    ss, ff = composite_function(s1, f1)
    w_head = calc_volume_weights(z[None], d_head[None], ss, last_dist)
    rgb_head = torch.sum(w_head.unsqueeze(-1) * ff, dim=-2).squeeze(0)
    if sig_t is None:
        return rgb_head, w_head.squeeze(0), None, None
    sig_t = sig_t[None].clone()
    feat_t = feat_t[None]
    if concate_bg:
        sig_t[:, :, -1] = 0
    s2 = F.relu(torch.stack([sig_h, sig_t], 0))
    s2 = torch.cat([s2[:-1], s2[-1:] + bump], 0)
    f2 = torch.stack([feat_h, feat_t], 0)
    ss2, ff2 = composite_function(s2, f2)
    w_com = calc_volume_weights(z[None], d_torso[None], ss2, last_dist)
    rgb_com = torch.sum(w_com.unsqueeze(-1) * ff2, dim=-2).squeeze(0)
    return rgb_head, w_head.squeeze(0), rgb_com, w_com.squeeze(0)


# --------------------------------------------------------------------------
# A13: fine sampler
# --------------------------------------------------------------------------
def wave_sum64(x):
    """Sum over the last axis in the FIXED order the HIP kernels document (dfn_misc.hip:sample_pdf_kernel,
    dfn_render_kernels.h): element k is accumulated by lane k % 64 (k, k + 64, ... in sequence), then the 64 lanes are
    combined by a butterfly: partners at XOR distance 32, 16, 8, 4, 2, 1 (every lane ends with the same value: f32
    addition is commutative).  torch.sum's own order is not even stable across CPU vector widths, so a kernel cannot be
    bitwise against it; it can against this."""
    n = x.shape[-1]
    xp = F.pad(x, (0, (-n) % 64)).reshape(*x.shape[:-1], -1, 64)
    part = xp[..., 0, :].clone()
    for j in range(1, xp.shape[-2]):
        part = part + xp[..., j, :]
    lanes = torch.arange(64)
    for d in (32, 16, 8, 4, 2, 1):
        part = part + part[..., lanes ^ d]
    return part[..., :1]


def sample_pdf(bins, weights, n_samples, det=False, u=None, fixed_order=False):
    """HELP:537-581.  bins [R,nb], weights [R,nb-1] -> [R,n_samples].
    `u` overrides the uniform draws (the reference's pytest mode feeds
    np.random.seed(0) numbers here).  fixed_order: the normaliser sum(weights + 1e-5) in the kernels' documented
    order (wave_sum64) instead of torch.sum's - everything else is elementwise or sequential and identical: with it the
    HIP kernel matches this function BIT FOR BIT; without it this function matches the reference bit for bit on the
    machine that made golden G5."""
    weights = weights + 1e-5
    pdf = weights / (wave_sum64(weights) if fixed_order else torch.sum(weights, -1, keepdim=True))
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = linspace01(n_samples).expand(list(cdf.shape[:-1]) + [n_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


# --------------------------------------------------------------------------
# A15 / row H: whole-frame renderers
# --------------------------------------------------------------------------
def _eval_fields(P, o_h, d_h, o_t, d_t, z, z_shape, z_app, signal, signal_torso, fields):
    """Evaluate the head (and torso) field at samples z [C,S] of C rays."""
    C, S = z.shape
    p = ray_points(o_h, d_h, z).reshape(1, -1, 3)
    r = d_h[:, None, :].expand(C, S, 3).reshape(1, -1, 3)
    f_h, s_h = decoder_forward(P, p, r, z_shape[:, 0], z_app[:, 0], signal, 'head')
    f_h, s_h = f_h.reshape(C, S, 3), s_h.reshape(C, S)
    if fields == 1:
        return s_h, f_h, None, None
    p = ray_points(o_t, d_t, z).reshape(1, -1, 3)
    r = d_t[:, None, :].expand(C, S, 3).reshape(1, -1, 3)
    f_t, s_t = decoder_forward(P, p, r, z_shape[:, 1], z_app[:, 1], signal_torso, 'torso')
    return s_h, f_h, s_t.reshape(C, S), f_t.reshape(C, S, 3)


def render_rays_chunk(P, o_h, d_h, o_t, d_t, bg, near, far, z_shape, z_app, signal, signal_torso,
                      n_coarse=64, n_fine=0, fields=2, last_dist=1e10, return_aux=False):
    """One chunk of rays through the live renderer (MAIN:653-709) and, when
    n_fine > 0, through row H of SURVEY.md 8(a):
      coarse pass -> weights (head-only weights if fields==1, composited weights
      if fields==2) -> z_mid -> sample_pdf(z_mid, w[1:-1], n_fine, det=True) ->
      z_all = sort(cat(z, z_fine)) -> same decoder on all n_coarse+n_fine samples ->
      same compositing with S = n_coarse+n_fine."""
    C = d_h.shape[0]
    z = coarse_z(near, far, n_coarse)[None, :].expand(C, n_coarse)
    s_h, f_h, s_t, f_t = _eval_fields(P, o_h, d_h, o_t, d_t, z, z_shape, z_app, signal,
                                      signal_torso, fields)
    rgb_h, w_h, rgb_c, w_c = integrate_fields(z, d_h, d_t, s_h, f_h, s_t, f_t, bg, last_dist)
    aux = {"z_coarse": z, "w_head_coarse": w_h, "w_com_coarse": w_c,
           "rgb_head_coarse": rgb_h, "rgb_com_coarse": rgb_c}
    if n_fine > 0:
        w = w_h if fields == 1 else w_c
        z_mid = .5 * (z[..., 1:] + z[..., :-1])
        z_f = sample_pdf(z_mid, w[..., 1:-1], n_fine, det=True).detach()
        z_all, _ = torch.sort(torch.cat([z, z_f], -1), -1)
        s_h, f_h, s_t, f_t = _eval_fields(P, o_h, d_h, o_t, d_t, z_all, z_shape, z_app, signal,
                                          signal_torso, fields)
        rgb_h, w_h, rgb_c, w_c = integrate_fields(z_all, d_h, d_t, s_h, f_h, s_t, f_t, bg, last_dist)
        aux.update({"z_fine": z_f, "z_all": z_all})
    aux.update({"w_head": w_h, "w_com": w_c})
    if return_aux:
        return rgb_h, rgb_c, aux
    return rgb_h, rgb_c


def render_fixed_samples(P, o_h, d_h, o_t, d_t, bg, z, z_shape, z_app, signal, signal_torso, fields=2,
                         last_dist=1e10):
    """Decoder + compositing (MAIN:660-709) at given per-ray sample depths z [C,S]."""
    s_h, f_h, s_t, f_t = _eval_fields(P, o_h, d_h, o_t, d_t, z, z_shape, z_app, signal, signal_torso, fields)
    rgb_h, _, rgb_c, _ = integrate_fields(z, d_h, d_t, s_h, f_h, s_t, f_t, bg, last_dist)
    return rgb_h, rgb_c


def render_frame(P, H, W, focal, cx, cy, pose, pose_body, bg_img, near, far, z_shape, z_app,
                 signal, signal_torso, n_coarse=64, n_fine=0, fields=2, chunk=2048,
                 last_dist=1e10, ray_begin=0, ray_count=None):
    """The frame loop MAIN:633-715 (chunked, ragged last chunk).  bg_img [H,W,3] f32 in [0,1].
    Returns rgb_head [n,3], rgb_com [n,3] (None when fields==1) for rays
    [ray_begin, ray_begin+ray_count) in row-major y*W+x order."""
    o_h, d_h = get_rays(H, W, focal, T(pose)[:3, :4], cx, cy)
    o_t, d_t = get_rays(H, W, focal, T(pose_body)[:3, :4], cx, cy)
    o_h, d_h, o_t, d_t = [t.reshape(-1, 3) for t in (o_h, d_h, o_t, d_t)]
    bg = T(bg_img).float().reshape(-1, 3)
    n = H * W if ray_count is None else ray_count
    out_h, out_c = [], []
    for b in range(ray_begin, ray_begin + n, chunk):
        e = min(b + chunk, ray_begin + n)
        rh, rc = render_rays_chunk(P, o_h[b:e], d_h[b:e], o_t[b:e], d_t[b:e], bg[b:e], near, far,
                                   z_shape, z_app, signal, signal_torso, n_coarse, n_fine,
                                   fields, last_dist)
        out_h.append(rh)
        if rc is not None:
            out_c.append(rc)
    return torch.cat(out_h, 0), (torch.cat(out_c, 0) if out_c else None)


# --------------------------------------------------------------------------
# A16: one training step (loss + gradients); optimizer handling is in the tests
# --------------------------------------------------------------------------
def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10. * torch.log(x) / math.log(10.)


def train_loss(P, nets, sel_yx, H, W, focal, cx, cy, pose, pose_torso, bg_img, target_head,
               target_com, near, far, z_shape, z_app, auds, exps, poses, img_i, global_step,
               nosmo_iters, smo_size, smo_torse_size, len_train, n_samples=64, last_dist=1e10):
    """MAIN:779-907: signals -> rays at the selected pixels -> both fields, coarse only ->
    loss = mse(rgb_com_torso, target_com) + mse(rgb_head, target_head)."""
    signal = encode_signal(nets, auds, exps, img_i, global_step, nosmo_iters, smo_size, len_train)
    signal_torso = encode_signal_torso(nets, poses, img_i, global_step, nosmo_iters,
                                       smo_torse_size, len_train)
    o_h, d_h = get_rays(H, W, focal, T(pose)[:3, :4], cx, cy)
    o_t, d_t = get_rays(H, W, focal, T(pose_torso)[:3, :4], cx, cy)
    ys, xs = sel_yx[:, 0], sel_yx[:, 1]
    bg = T(bg_img)[ys, xs]
    rgb_h, rgb_c = render_rays_chunk(P, o_h[ys, xs], d_h[ys, xs], o_t[ys, xs], d_t[ys, xs], bg,
                                     near, far, z_shape, z_app, signal, signal_torso,
                                     n_coarse=n_samples, n_fine=0, fields=2, last_dist=last_dist)
    loss_head = img2mse(rgb_h, T(target_head)[ys, xs])
    loss_com = img2mse(rgb_c, T(target_com)[ys, xs])
    return loss_com + loss_head, loss_head, loss_com
