"""Host-side mirror of the reference's decoder.py: same constructor, same state_dict keys and registration
order (so the reference's .tar checkpoints load unchanged), same forward() signature.

Reference: /root/reference/NeRFs/DFANeRF/decoder.py:77-134 (DeformationField_ori), :137-349 (Decoder).

Single backend: forward() always runs the fused HIP decoder on the module's device -
  * no-grad (inference): dfn_decoder_fwd;
  * grad mode (training on explicit points, as the reference's loop calls it, MAIN:855-866): the same kernel with its
    recorder on and the HIP backward chain behind a torch.autograd.Function (training.DecoderTrainFn).
There is no ATen / CPU path in this module: CPU tensors raise.  (The torch-op restatement the tests cross-check against
lives in tests/twins.py.)
"""
import math

import torch
import torch.nn as nn


class DeformationField_ori(nn.Module):
    """Two independent 5x64 ReLU MLPs producing residuals for the PE (dim_embed) and the pose signal
    (dim_signal); skip connection after hidden layer index 3."""

    def __init__(self, dim_embed, dim_signal, hidden_size=64, n_blocks=7, skips=[4]):
        super().__init__()
        self.dim_embed, self.dim_signal, self.skips = dim_embed, dim_signal, skips
        n_hidden = n_blocks - 3
        self.blocks_embed = nn.ModuleList([nn.Linear(dim_embed + dim_signal, hidden_size)] +
                                          [nn.Linear(hidden_size, hidden_size) for _ in range(n_hidden)])
        self.out_embed = nn.Linear(hidden_size, dim_embed)
        self.blocks_signal = nn.ModuleList([nn.Linear(dim_embed + dim_signal, hidden_size)] +
                                           [nn.Linear(hidden_size, hidden_size) for _ in range(n_hidden)])
        self.out_signal = nn.Linear(hidden_size, dim_signal)
        n_skips = sum(i in skips for i in range(n_blocks - 1))
        if n_skips > 0:
            self.fc_embed_skips = nn.ModuleList([nn.Linear(dim_embed, hidden_size) for _ in range(n_skips)])
            self.fc_signal_skips = nn.ModuleList([nn.Linear(dim_signal, hidden_size) for _ in range(n_skips)])

    def forward(self, x):
        raise RuntimeError("DeformationField_ori is evaluated inside the fused HIP decoder (Decoder.forward, "
                           "head_or_torso='torso'); it has no stand-alone path")


class Decoder(nn.Module):
    """GIRAFFE-style conditional NeRF decoder with head / listener / torso input layers."""

    def __init__(self, hidden_size=128, n_blocks=8, n_blocks_view=1, dim_signal=64, skips=[4], use_viewdirs=True,
                 n_freq_posenc=10, dim_exp=256, dim_et_embed=42, n_freq_posenc_views=4, use_aud_net=False,
                 dim_aud=64, z_dim=64, rgb_out_dim=3, final_sigmoid_activation=True, downscale_p_by=2.,
                 positional_encoding="normal", use_wav2lip=False, dim_w2lfeature=512, gauss_dim_pos=10,
                 gauss_dim_view=4, gauss_std=4., use_deformation_field=False, use_expression=False, **kwargs):
        super().__init__()
        assert positional_encoding in ('normal', 'gauss')
        if positional_encoding == 'gauss':
            raise NotImplementedError("gauss positional encoding is never selected by the reference driver")
        self.use_viewdirs, self.skips = use_viewdirs, skips
        self.n_freq_posenc, self.n_freq_posenc_views = n_freq_posenc, n_freq_posenc_views
        self.downscale_p_by, self.z_dim = downscale_p_by, z_dim
        self.final_sigmoid_activation = final_sigmoid_activation
        self.n_blocks, self.n_blocks_view, self.dim_signal = n_blocks, n_blocks_view, dim_signal
        self.use_deformation_field, self.use_expression, self.use_wav2lip = use_deformation_field, use_expression, use_wav2lip
        self.positional_encoding = positional_encoding
        self.hidden_size, self.dim_et_embed = hidden_size, dim_et_embed
        dim_embed, dim_embed_view = 3 * n_freq_posenc * 2, 3 * n_freq_posenc_views * 2
        # registration order below == the reference's (decoder.py:207-255): it fixes the state_dict order
        if use_deformation_field:
            self.deform_net = DeformationField_ori(dim_embed, dim_et_embed)
        if use_expression:
            self.expnet = nn.Linear(dim_exp, hidden_size)
        if use_wav2lip:
            self.w2lnet = nn.Linear(dim_w2lfeature, hidden_size)
        self.fc_in = nn.Linear(dim_embed + dim_signal, hidden_size)
        self.fc_in_listener = nn.Linear(dim_embed, hidden_size)
        self.fc_in_torso = nn.Linear(dim_embed + dim_et_embed, hidden_size)
        if z_dim > 0:
            self.fc_z = nn.Linear(z_dim, hidden_size)
        self.blocks = nn.ModuleList([nn.Linear(hidden_size, hidden_size) for _ in range(n_blocks - 1)])
        n_skips = sum(i in skips for i in range(n_blocks - 1))
        if n_skips > 0:
            self.fc_z_skips = nn.ModuleList([nn.Linear(z_dim, hidden_size) for _ in range(n_skips)])
            self.fc_p_skips = nn.ModuleList([nn.Linear(dim_embed + dim_signal, hidden_size) for _ in range(n_skips)])
            self.fc_p_skips_listener = nn.ModuleList([nn.Linear(dim_embed, hidden_size) for _ in range(n_skips)])
            self.fc_p_skips_torso = nn.ModuleList([nn.Linear(dim_embed + dim_et_embed, hidden_size)
                                                   for _ in range(n_skips)])
        self.sigma_out = nn.Linear(hidden_size, 1)
        self.fc_z_view = nn.Linear(z_dim, hidden_size)
        self.feat_view = nn.Linear(hidden_size, hidden_size)
        self.fc_view = nn.Linear(dim_embed_view, hidden_size)
        self.feat_out = nn.Linear(hidden_size, rgb_out_dim)
        if use_viewdirs and n_blocks_view > 1:
            self.blocks_view = nn.ModuleList([nn.Linear(dim_embed_view + hidden_size, hidden_size)
                                              for _ in range(n_blocks_view - 1)])
        self._hip = {}            # tier -> (PackedDecoder, param version stamp)
        from ._lib import DECODER_UNUSED_PREFIXES
        self._dfn_flat_skip = DECODER_UNUSED_PREFIXES        # (training._FlatNet: not part of the flat parameter vector)

    # ---- positional encoding (decoder.py:257-275) ----------------------------------------------------------
    def transform_points(self, p, views=False):
        p = p / self.downscale_p_by
        L = self.n_freq_posenc_views if views else self.n_freq_posenc
        return torch.cat([torch.cat([torch.sin((2 ** i) * math.pi * p), torch.cos((2 ** i) * math.pi * p)], dim=-1)
                          for i in range(L)], dim=-1)

    # ---- HIP plumbing ------------------------------------------------------------------------------------------
    def hip_supported(self):
        return (0 < self.hidden_size <= 256 and 0 < self.z_dim <= 256 and self.n_blocks == 8 and list(self.skips) == [4] and
                self.dim_signal == 96 and self.dim_et_embed == 42 and self.n_freq_posenc == 10 and
                self.n_freq_posenc_views == 4 and self.use_viewdirs and self.n_blocks_view == 1 and
                self.final_sigmoid_activation and self.downscale_p_by == 2.)

    def _dfn_flat_layout(self):
        """training._FlatNet hook: None for the scripts' decoder (its flat parameter vector IS the library's); otherwise (first
        offset, padded shape per tensor of the flat vector): a narrower decoder, or one without the deformation field, trains inside
        the library's 256-wide layout (engine.padded_shape) - padded entries start at zero, get zero gradients and stay zero."""
        if self.hidden_size == 256 and self.z_dim == 256 and self.use_deformation_field:
            return None
        from . import engine
        from ._lib import DECODER_UNUSED_PREFIXES
        names = [k for k in self.state_dict().keys() if not k.startswith(DECODER_UNUSED_PREFIXES)]
        sd = self.state_dict(keep_vars=True)
        return (0 if self.use_deformation_field else engine.N_DEFORM_PARAMS), [engine.padded_shape(k, sd[k].shape) for k in names]

    def packed(self, tier="bf16"):
        """Kernel-ready weights for `tier`, repacked whenever a parameter has changed in place."""
        from . import engine
        if not self.hip_supported():
            raise NotImplementedError("the HIP path supports the scripts/test_obama.sh decoder configuration only")
        from ._lib import DECODER_UNUSED_PREFIXES
        # (use_expression / use_wav2lip: expnet / w2lnet are registered - checkpoint parity - and evaluated by nothing on this path)
        params = [v for k, v in self.state_dict().items() if not k.startswith(DECODER_UNUSED_PREFIXES)]
        stamp = (tuple(p._version for p in params), params[0].device, params[0].data_ptr())
        hit = self._hip.get(tier)
        if hit is None or hit[1][1:] != stamp[1:]:
            flat = engine.flatten_state(self.state_dict(), params[0].device)
            hit = (engine.PackedDecoder(flat, tier, fields=(0, 1, 2), z_dim=self.z_dim), stamp)
        elif hit[1][0] != stamp[0]:
            hit[0].flat.copy_(engine.flatten_state(self.state_dict(), params[0].device))
            hit[0].repack()
            hit = (hit[0], stamp)
        self._hip[tier] = hit
        return hit[0]

    # ---- forward (decoder.py:277-349) ---------------------------------------------------------------------------
    def forward(self, p_in, ray_d, z_shape=None, z_app=None, signal=None, head_or_torso=None, tier="f32"):
        if head_or_torso not in ('head', 'torso'):
            raise Exception('Do not give head or torso!!')
        if head_or_torso == 'head':
            if self.use_expression and signal[1] is not None:
                raise NotImplementedError("use_expression with an expression signal (signal[1]) is the reference's branch for a second "
                                          "person (itr_obj > 0, MAIN:72-75): the scripts train one (--n_object 1), whose "
                                          "signal[1] is None; expnet is a registered, unused layer on this path")
            signal = signal[0]
        if ray_d is None or z_shape is None or z_app is None:
            raise ValueError("Decoder.forward needs ray_d, z_shape and z_app (the reference's random-latent default, "
                             "decoder.py:282-287, is never used by its driver)")
        if not p_in.is_cuda:
            raise RuntimeError("Decoder.forward runs the fused HIP kernel and needs device tensors "
                               "(there is no CPU fallback)")
        from . import engine
        field = 1 if head_or_torso == 'torso' else (0 if signal is not None else 2)
        needs_grad = torch.is_grad_enabled() and (
            any(p.requires_grad for p in self.parameters()) or (signal is not None and signal.requires_grad))
        if needs_grad:
            from . import training
            return training.decoder_train(self, field, p_in, ray_d, z_shape, z_app, signal, tier)
        pk = self.packed(tier)
        bias = pk.fold_single(field, signal, z_shape.reshape(-1)[:self.z_dim], z_app.reshape(-1)[:self.z_dim])
        feat, sigma = engine.decoder_forward(pk, field, bias, p_in.reshape(-1, 3), ray_d.reshape(-1, 3))
        return feat.reshape(p_in.shape[0], -1, 3), sigma.reshape(p_in.shape[0], -1)
