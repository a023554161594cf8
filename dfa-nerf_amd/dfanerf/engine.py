"""Host-side driver of libdfanerf.so: device buffers, streams and argument marshalling.

PyTorch is plumbing here (device memory, the current HIP stream, torch.distributed); every number on the
render path comes out of the HIP kernels behind the C ABI (include/dfanerf.h).  Nothing in this module
computes on the CPU and nothing falls back to ATen."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (DECODER_UNUSED_PREFIXES, DfnFrame, FIELD_HEAD, FIELD_LISTENER, FIELD_TORSO, N_DECODER_PARAMS, TIER_BF16, TIER_F16, TIER_F32,
                   check, lib)

# "f16": v_mfma_f32_32x32x16_f16, the throughput tier (inference only); "bf16": also the 16-bit training tier
TIERS = {"f32": TIER_F32, "bf16": TIER_BF16, "f16": TIER_F16, TIER_F32: TIER_F32, TIER_BF16: TIER_BF16,
         TIER_F16: TIER_F16}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t, device):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device=device, dtype=torch.float32).contiguous()


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("dfanerf: no HIP device visible; the render path has no CPU fallback")


Z_WEIGHTS = ("fc_z.weight", "fc_z_skips.0.weight", "fc_z_view.weight")     # [256, z_dim]: the layers fed by the latent codes
# --n_feat < 256 (rendering only): the layers whose OUTPUT / INPUT is the hidden vector, padded with zero rows / columns to 256
_HID_ROWS = ("fc_in.", "fc_in_listener.", "fc_in_torso.", "fc_z.", "blocks.", "fc_z_skips.", "fc_p_skips.", "fc_p_skips_listener.",
             "fc_p_skips_torso.", "fc_z_view.", "feat_view.", "fc_view.")
_HID_COLS = ("blocks.", "sigma_out.", "feat_view.", "feat_out.")
N_DEFORM_PARAMS = 59750         # deform_net.* (decoder.py:84-105): the first 28 tensors of the flat vector (dfn_layout.h: P_DE0_W .. P_DSSK_B)


def padded_shape(name, shape):
    """shape of decoder tensor `name` in the library's 256-wide layout: hidden rows / columns and latent-code columns padded to 256"""
    shape = list(shape)
    if name in Z_WEIGHTS:
        shape[1] = 256
    if name.startswith(_HID_COLS) and name.endswith(".weight"):
        shape[1] = 256
    if name.startswith(_HID_ROWS):
        shape[0] = 256
    return tuple(shape)


def flatten_state(state, device):
    """decoder.state_dict() -> flat f32 device vector in registration order (dfn_layout.h:ParamId).
    --z_dim < 256 (round 6): the three layers the latent codes feed are [256, z_dim] and act on per-frame constants
    only (the fold, dfn_fold_bias) - they enter the library's [256, 256] slots padded with zero columns, and the codes are padded with
    zeros to match (pad_z): W_pad . z_pad = W . z exactly (zero products add nothing in any summation order).
    --n_feat < 256: a hidden unit with zero weights and zero bias outputs relu(0) = 0 and feeds zero columns - the
    network written out 256 wide is the same function, exactly (at the 256-wide network's cost).
    (Training such decoders: training._FlatNet keeps the same padded layout; the padded entries get zero gradients - relu'(0) = 0 -
    and stay zero.)  A decoder WITHOUT --use_deformation_field: the torso evaluates `deform(p) + p` (decoder.py:297-299) with an all-zero
    deformation network - every layer of it returns exactly 0 (relu(0) = 0, bias 0), so `p` passes unchanged, bit for bit."""
    parts = []
    if not any(k.startswith("deform_net.") for k in state):
        parts.append(torch.zeros(N_DEFORM_PARAMS, dtype=torch.float32, device=device))
    for k, v in state.items():
        if k.startswith(DECODER_UNUSED_PREFIXES):
            continue
        v = _f32c(v, device)
        want = padded_shape(k, v.shape)
        if want != tuple(v.shape):
            full = torch.zeros(want, dtype=torch.float32, device=device)
            full[tuple(slice(0, n) for n in v.shape)] = v
            v = full
        parts.append(v.reshape(-1))
    flat = torch.cat(parts)
    if flat.numel() != N_DECODER_PARAMS:
        raise ValueError(f"decoder has {flat.numel()} parameters; the HIP path supports the "
                         f"scripts/test_obama.sh architecture ({N_DECODER_PARAMS})")
    return flat


class PackedDecoder:
    """Kernel-ready weight streams of one decoder, per (tier, field).  Call repack() after an optimizer step."""

    def __init__(self, flat_params, tier="bf16", fields=(FIELD_HEAD, FIELD_TORSO), z_dim=256):
        require_gpu()
        if not 0 < int(z_dim) <= 256:
            raise ValueError(f"PackedDecoder: z_dim {z_dim} (1 ... 256)")
        self.z_dim = int(z_dim)         # width of the latent codes fold() is handed (padded with zeros to the library's 256)
        self.tier = TIERS[tier]
        self.flat = flat_params
        self.device = flat_params.device
        self.packed = {}
        self.f16_bounds = None          # f16 tier: calibrated max |activation| per layer (f16guard.activation_bounds), once known
        self.f16_weight_max = None
        for f in fields:
            nbytes = check(lib.dfn_packed_bytes(self.tier, f), "dfn_packed_bytes")
            self.packed[f] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.repack()

    def repack(self):
        if self.tier == TIERS["f16"]:
            # half precision's range (f16guard.py): a weight beyond it would be packed as inf.  One device reduction and one
            # host read per (re)pack - the f16 tier is inference only: it packs once per checkpoint, not once per step
            from . import f16guard
            self.f16_weight_max = f16guard.weight_bound(self.flat)
            f16guard.check(None, self.f16_weight_max, what="a decoder parameter")
        for f, buf in self.packed.items():
            check(lib.dfn_pack_weights(self.tier, f, _ptr(self.flat), _ptr(buf), _stream()), "dfn_pack_weights")

    def bias_floats(self, field):
        return check(lib.dfn_bias_floats(self.tier, field), "dfn_bias_floats")

    def fold(self, sig_head, sig_torso, z_shape, z_app, head_field=FIELD_HEAD, out=None):
        """Per-frame bias blob [head | torso].  z_shape / z_app: [2,256] rows (head, torso) or [256] for
        a single field.  sig_head [96] (None for the listener), sig_torso [42] (None = head only)."""
        dev = self.device
        nh = self.bias_floats(head_field)
        nt = self.bias_floats(FIELD_TORSO) if sig_torso is not None else 0
        if out is None:
            out = torch.empty(nh + nt, dtype=torch.float32, device=dev)
        zs, za = self.pad_z(z_shape), self.pad_z(z_app)
        sh = None if sig_head is None else _f32c(sig_head, dev).reshape(-1)
        check(lib.dfn_fold_bias(self.tier, head_field, _ptr(self.flat), _ptr(sh), _ptr(zs[0]), _ptr(za[0]),
                                _ptr(out), _stream()), "dfn_fold_bias(head)")
        if nt:
            st = _f32c(sig_torso, dev).reshape(-1)
            row = 1 if zs.shape[0] > 1 else 0
            check(lib.dfn_fold_bias(self.tier, FIELD_TORSO, _ptr(self.flat), _ptr(st), _ptr(zs[row]), _ptr(za[row]),
                                    C.c_void_p(out.data_ptr() + 4 * nh), _stream()), "dfn_fold_bias(torso)")
        return out

    def pad_z(self, z):
        """latent code(s) [..., z_dim] -> [rows, 256] f32 on the device, zero-padded (flatten_state pads the weights to match)"""
        z = _f32c(z, self.device).reshape(-1, self.z_dim)
        return z if self.z_dim == 256 else torch.nn.functional.pad(z, (0, 256 - self.z_dim)).contiguous()

    def fold_single(self, field, signal, z_shape, z_app):
        dev = self.device
        out = torch.empty(self.bias_floats(field), dtype=torch.float32, device=dev)
        # keep every temporary alive until the launch is enqueued (the caching allocator would otherwise
        # hand the same block to the next temporary)
        sg = None if signal is None else _f32c(signal, dev).reshape(-1)
        zs = self.pad_z(z_shape).reshape(-1)
        za = self.pad_z(z_app).reshape(-1)
        check(lib.dfn_fold_bias(self.tier, field, _ptr(self.flat), _ptr(sg), _ptr(zs), _ptr(za), _ptr(out),
                                _stream()), "dfn_fold_bias")
        return out


class SignalEncoder:
    """dfn_encode_signal / dfn_encode_signal_torso: the conditioning networks' forward in HIP (inference).  Holds flat
    f32 copies of the four networks' parameters (state_dict order); call refresh() after they change."""

    def __init__(self, aud_net, exp_net, att_net, pose_att_net, auds, exps, poses):
        require_gpu()
        self.nets = (aud_net, exp_net, att_net, pose_att_net)
        dev = auds.device
        self.auds, self.exps = _f32c(auds, dev), _f32c(exps, dev)
        self.poses = _f32c(poses, dev)
        self.n = int(self.auds.shape[0])
        self.pose_stride = int(self.poses[0].numel())
        self.device = dev
        self.refresh()

    def refresh(self):
        flat = lambda m: None if m is None else torch.cat(
            [v.detach().reshape(-1).to(self.device, torch.float32) for v in m.state_dict().values()]).contiguous()
        self.p_aud, self.p_exp, self.p_att, self.p_patt = [flat(m) for m in self.nets]

    def encode(self, frame_ids, smo_size, smo_torso_size, length=None):
        """-> sig [B,96], sig_torso [B,42] for the frames `frame_ids`; smo_* = 0 selects the unsmoothed branch;
        `length` = number of leading frames that form the sequence (zero padding beyond it), default all."""
        if isinstance(frame_ids, torch.Tensor):
            ids = frame_ids.to(device=self.device, dtype=torch.int32).reshape(-1)
        else:
            frame_ids = list(frame_ids)
            if len(frame_ids) == 1 and 0 <= int(frame_ids[0]) < (1 << 20):
                # one frame (the render loop): a slice of a device arange instead of a pageable host-to-device copy,
                # which would block the host until the previous frame's kernels have finished
                ar = getattr(self, "_arange", None)
                if ar is None or int(frame_ids[0]) >= ar.numel():
                    ar = self._arange = torch.arange(max(int(frame_ids[0]) + 1, 8192), dtype=torch.int32, device=self.device)
                ids = ar[int(frame_ids[0]):int(frame_ids[0]) + 1]
            else:
                ids = torch.as_tensor(frame_ids, dtype=torch.int32, device=self.device)
        n_total = self.n if length is None else int(length)
        B = ids.numel()
        sig = torch.empty(B, 96, dtype=torch.float32, device=self.device)
        sigt = torch.empty(B, 42, dtype=torch.float32, device=self.device)
        check(lib.dfn_encode_signal(_ptr(self.p_aud), _ptr(self.p_exp), _ptr(self.p_att), _ptr(self.auds),
                                    _ptr(self.exps), n_total, _ptr(ids), B, int(smo_size), _ptr(sig), _stream()),
              "dfn_encode_signal")
        check(lib.dfn_encode_signal_torso(_ptr(self.p_patt), _ptr(self.poses), self.pose_stride, n_total, _ptr(ids), B,
                                          int(smo_torso_size), _ptr(sigt), _stream()), "dfn_encode_signal_torso")
        return sig, sigt


_STREAMS = {}


def side_stream(device, high=False, role=None):
    """A side stream by ROLE ("wgrad", "sig_a", "sig_p"), one per device and role for the whole process: the device runs four
    hardware queues, and every further stream shares one with another stream and serialises with it (LABNOTES.md 7) - a second
    TrainBuffers / SignalTrainer / FramePrefetcher in the same process (bench.py's other workloads, the test renders of a
    training run, a second model) must reuse the first one's streams instead of creating more."""
    if role is None:
        return torch.cuda.Stream(device=device, priority=-1 if high else 0)
    d = torch.device(device)
    key = (d.index if d.index is not None else torch.cuda.current_device(), role, bool(high))
    s = _STREAMS.get(key)
    if s is None:
        s = _STREAMS[key] = torch.cuda.Stream(device=device, priority=-1 if high else 0)
    return s


class FramePrefetcher:
    """The per-frame front end (conditioning signals -> folded bias blob: two single-workgroup encoder launches and the fold,
    ~0.1 ms of latency chains that keep one compute unit busy) ONE FRAME AHEAD on a side stream: while frame k renders, frame
    k + 1's blob is produced underneath it; the render stream only waits for an event.  At 8 GPUs a rank's share of a frame
    is 4.4 ms, and the un-pipelined front end was 2.3 % of it (the part of a frame that does not shrink with the ray shard).
    Two blobs alternate; a blob is rewritten only after the render that read it (an event of the render stream)."""

    def __init__(self, encoder, packed, z_shape, z_app, smo_size, smo_torso_size, fields=2, length=None):
        self.enc, self.pk, self.zs, self.za = encoder, packed, z_shape, z_app
        self.smo, self.smo_t, self.fields, self.length = int(smo_size), int(smo_torso_size), int(fields), length
        self.side = side_stream(packed.device, role="wgrad")      # the process-wide general side stream
        # The blobs come from the CURRENT (main) stream's allocator pool, here, not from the side stream's inside _produce:
        # they are read on the render streams, and a block of the side stream's pool would go back to that pool when the
        # prefetcher is dropped - to be reused by the training step's weight-gradient stream while a render still reads it
        # (ADVICE r3).  Freed into the main pool the usual rule holds: the next user is ordered behind the main stream.
        nb = packed.bias_floats(FIELD_HEAD) + (packed.bias_floats(FIELD_TORSO) if self.fields == 2 else 0)
        self.slots = [{"bias": torch.empty(nb, dtype=torch.float32, device=packed.device), "ready": None, "free": None,
                       "frame": None} for _ in range(2)]
        # ... and they are WRITTEN on the side stream: tell the allocator, so that a blob freed while a queued _produce has
        # not run yet is not handed out again before the side stream got there
        for sl in self.slots:
            sl["bias"].record_stream(self.side)
        self.k = 0
        self._home = torch.cuda.current_stream(packed.device)
        self._stale = True                                 # the side stream has not seen the parameters yet

    def parameters_changed(self):
        """Call after the networks' parameters (or the packed weights / latent codes) were rewritten on the main stream: the
        next blob is produced behind the main stream's work instead of underneath it.  A prefetcher is otherwise meant to live
        for ONE render loop over fixed parameters (run_nerf builds one per loop): waiting for the main stream at every frame
        would serialise the front end with the render it is there to hide under."""
        self._stale = True

    def _produce(self, slot, frame):
        main = torch.cuda.current_stream(self.pk.device)
        if self._stale:
            self.side.wait_stream(main)                    # the parameters were written on the main stream
            self._stale = False
        if slot["free"] is not None:
            self.side.wait_event(slot["free"])             # the render that read this blob last
        with torch.cuda.stream(self.side):
            s2, t2 = self.enc.encode([int(frame)], self.smo, self.smo_t, length=self.length)
            slot["bias"] = self.pk.fold(s2[0], t2[0] if self.fields == 2 else None, self.zs, self.za, out=slot["bias"])
            slot["ready"] = torch.cuda.Event()
            slot["ready"].record(self.side)
        slot["frame"] = int(frame)

    def get(self, frame, next_frame=None):
        """-> the bias blob of `frame`, valid on the current stream (produced now if it was not prefetched).  `next_frame`:
        the frame to start underneath this one's render.  Call done() once the render that reads the blob is enqueued."""
        slot = self.slots[self.k]
        if slot["frame"] != int(frame):
            self._produce(slot, frame)
        cur = torch.cuda.current_stream(self.pk.device)
        cur.wait_event(slot["ready"])
        if cur != self._home:
            slot["bias"].record_stream(cur)                # read on a render stream other than the one it was allocated on
        self._cur, self._next = slot, next_frame
        return slot["bias"]

    def done(self):
        slot = self._cur
        slot["free"] = torch.cuda.Event()
        slot["free"].record(torch.cuda.current_stream(self.pk.device))
        slot["frame"] = None                               # consumed
        self.k ^= 1
        if self._next is not None:
            self._produce(self.slots[self.k], self._next)


def make_frame(H, W, focal, cx, cy, pose, pose_body, near, far, last_dist=1e10, ray_begin=0, ray_count=None,
               n_coarse=64, n_fine=0, fields=2, concate_bg=True):
    fr = DfnFrame()
    p = np.asarray(pose, np.float32)[:3, :4].reshape(-1)
    pb = np.asarray(pose_body if pose_body is not None else pose, np.float32)[:3, :4].reshape(-1)
    for i in range(12):
        fr.pose[i] = float(p[i])
        fr.pose_body[i] = float(pb[i])
    fr.H, fr.W = int(H), int(W)
    fr.focal, fr.cx, fr.cy = float(focal), float(cx), float(cy)
    fr.z_near, fr.z_far, fr.last_dist = float(near), float(far), float(last_dist)
    fr.ray_begin = int(ray_begin)
    fr.ray_count = int(H * W - ray_begin if ray_count is None else ray_count)
    fr.n_coarse, fr.n_fine, fr.fields, fr.concate_bg = int(n_coarse), int(n_fine), int(fields), int(bool(concate_bg))
    return fr


def render(packed, bias, frame, bg, pix_index=None, want_weights=False, out_head=None, out_com=None, want_z=False):
    """dfn_render_fwd.  bg: f32 [H*W,3] in [0,1] or uint8 [H*W,3] device tensor.
    Returns (rgb_head [n,3], rgb_com [n,3] or None[, w_head, w_com])."""
    dev = packed.device
    n = frame.ray_count
    two = frame.fields == 2
    rgb_h = out_head if out_head is not None else torch.empty(n, 3, dtype=torch.float32, device=dev)
    rgb_c = (out_com if out_com is not None else torch.empty(n, 3, dtype=torch.float32, device=dev)) if two else None
    for o in (rgb_h, rgb_c):
        if o is not None and (o.dtype != torch.float32 or not o.is_contiguous() or o.numel() != n * 3):
            raise ValueError("render: output buffers must be contiguous float32 [ray_count, 3]")
    S = frame.n_coarse + frame.n_fine
    w_h = torch.empty(n, S, dtype=torch.float32, device=dev) if want_weights else None
    w_c = torch.empty(n, S, dtype=torch.float32, device=dev) if (want_weights and two) else None
    z_v = torch.empty(n, S, dtype=torch.float32, device=dev) if want_z else None
    bg_f32 = bg if bg.dtype == torch.float32 else None
    bg_u8 = bg if bg.dtype == torch.uint8 else None
    if bg_f32 is None and bg_u8 is None:
        raise TypeError("bg must be float32 or uint8")
    nh = packed.bias_floats(FIELD_HEAD)
    bias_t = C.c_void_p(bias.data_ptr() + 4 * nh) if two else None
    if pix_index is not None:
        pix_index = pix_index.to(device=dev, dtype=torch.int32).contiguous()
    check(lib.dfn_render_fwd(packed.tier, C.byref(frame), _ptr(packed.packed[FIELD_HEAD]),
                             _ptr(packed.packed.get(FIELD_TORSO)) if two else None, _ptr(bias), bias_t,
                             _ptr(bg_f32), _ptr(bg_u8), _ptr(pix_index), _ptr(rgb_h), _ptr(rgb_c), _ptr(w_h),
                             _ptr(w_c), _ptr(z_v), _stream()), "dfn_render_fwd")
    out = (rgb_h, rgb_c)
    if want_weights:
        out += (w_h, w_c)
    if want_z:
        out += (z_v,)
    return out


def render_u8(packed, bias, frame, bg, pix_index=None, out_head=None, out_com=None):
    """dfn_render_fwd_u8: the same launch with to8b fused into the epilogue -> uint8 [n,3] images (head, composite)."""
    dev = packed.device
    n = frame.ray_count
    two = frame.fields == 2
    out_h = out_head if out_head is not None else torch.empty(n, 3, dtype=torch.uint8, device=dev)
    out_c = (out_com if out_com is not None else torch.empty(n, 3, dtype=torch.uint8, device=dev)) if two else None
    for o in (out_h, out_c):
        if o is not None and (o.dtype != torch.uint8 or not o.is_contiguous() or o.numel() != n * 3):
            raise ValueError("render_u8: output buffers must be contiguous uint8 [ray_count, 3]")
    bg_f32 = bg if bg.dtype == torch.float32 else None
    bg_u8 = bg if bg.dtype == torch.uint8 else None
    if bg_f32 is None and bg_u8 is None:
        raise TypeError("bg must be float32 or uint8")
    nh = packed.bias_floats(FIELD_HEAD)
    bias_t = C.c_void_p(bias.data_ptr() + 4 * nh) if two else None
    if pix_index is not None:
        pix_index = pix_index.to(device=dev, dtype=torch.int32).contiguous()
    check(lib.dfn_render_fwd_u8(packed.tier, C.byref(frame), _ptr(packed.packed[FIELD_HEAD]),
                                _ptr(packed.packed.get(FIELD_TORSO)) if two else None, _ptr(bias), bias_t,
                                _ptr(bg_f32), _ptr(bg_u8), _ptr(pix_index), _ptr(out_h), _ptr(out_c), _stream()),
          "dfn_render_fwd_u8")
    return out_h, out_c


def decoder_forward(packed, field, bias, points, dirs):
    """dfn_decoder_fwd: points/dirs [N,3] -> feat [N,3], sigma [N]."""
    dev = packed.device
    pts = _f32c(points, dev).reshape(-1, 3)
    dr = _f32c(dirs, dev).reshape(-1, 3)
    n = pts.shape[0]
    feat = torch.empty(n, 3, dtype=torch.float32, device=dev)
    sigma = torch.empty(n, dtype=torch.float32, device=dev)
    if n == 0:
        return feat, sigma
    check(lib.dfn_decoder_fwd(packed.tier, field, _ptr(packed.packed[field]), _ptr(bias), _ptr(pts), _ptr(dr), n,
                              _ptr(feat), _ptr(sigma), _stream()), "dfn_decoder_fwd")
    return feat, sigma


# ---- building blocks ------------------------------------------------------------------------------------------
def get_rays(H, W, focal, c2w, cx=None, cy=None, device="cuda", stride=1):
    require_gpu()
    cx = W * .5 if cx is None else cx
    cy = H * .5 if cy is None else cy
    m = (c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else np.asarray(c2w)).astype(np.float32)
    m = np.ascontiguousarray(m[:3, :4]).reshape(-1)
    arr = (C.c_float * 12)(*[float(v) for v in m])
    stride = int(stride)
    ro = torch.empty(H // stride, W // stride, 3, dtype=torch.float32, device=device)
    rd = torch.empty(H // stride, W // stride, 3, dtype=torch.float32, device=device)
    check(lib.dfn_get_rays_strided(int(H), int(W), stride, float(focal), float(cx), float(cy), arr, _ptr(ro), _ptr(rd), _stream()),
          "dfn_get_rays")
    return ro, rd


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    dev = rays_o.device
    ro, rd = _f32c(rays_o, dev), _f32c(rays_d, dev)
    oo, od = torch.empty_like(ro), torch.empty_like(rd)
    check(lib.dfn_ndc_rays(int(H), int(W), float(focal), float(near), _ptr(ro), _ptr(rd), ro.numel() // 3, _ptr(oo),
                           _ptr(od), _stream()), "dfn_ndc_rays")
    return oo, od


def sample_pdf(bins, weights, n_samples, det=False, u=None):
    dev = bins.device
    b, w = _f32c(bins, dev), _f32c(weights, dev)
    R, nb = b.shape
    if u is None and not det:
        u = torch.rand(R, n_samples, device=dev)
    uu = None if u is None else _f32c(u, dev)
    out = torch.empty(R, n_samples, dtype=torch.float32, device=dev)
    if R == 0:
        return out
    check(lib.dfn_sample_pdf(_ptr(b), _ptr(w), R, nb, int(n_samples), _ptr(uu), _ptr(out), _stream()),
          "dfn_sample_pdf")
    return out


def composite(sigma, feat):
    """sigma [K,...], feat [K,...,3] -> sigma_sum [...], feat_w [...,3]."""
    dev = sigma.device
    K = sigma.shape[0]
    s, f = _f32c(sigma, dev), _f32c(feat, dev)
    N = s.numel() // K
    ss = torch.empty(s.shape[1:], dtype=torch.float32, device=dev)
    fw = torch.empty(f.shape[1:], dtype=torch.float32, device=dev)
    check(lib.dfn_composite(_ptr(s), _ptr(f), K, N, _ptr(ss), _ptr(fw), _stream()), "dfn_composite")
    return ss, fw


def volume_weights(z_vals, ray_vector, sigma, last_dist=1e10):
    dev = sigma.device
    S = z_vals.shape[-1]
    z = _f32c(z_vals, dev).reshape(-1, S)
    r = _f32c(ray_vector, dev).reshape(-1, 3)
    sg = _f32c(sigma, dev).reshape(-1, S)
    w = torch.empty_like(sg)
    check(lib.dfn_volume_weights(_ptr(z), _ptr(r), _ptr(sg), z.shape[0], S, float(last_dist), _ptr(w), _stream()),
          "dfn_volume_weights")
    return w.reshape(sigma.shape)


def to8b(x):
    dev = x.device
    xx = _f32c(x, dev)
    out = torch.empty(xx.shape, dtype=torch.uint8, device=dev)
    check(lib.dfn_to8b(_ptr(xx), xx.numel(), _ptr(out), _stream()), "dfn_to8b")
    return out


def mfma_layout_probe(device="cuda"):
    out = torch.zeros(3, 32, 32, dtype=torch.float32, device=device)
    check(lib.dfn_debug_mfma_layout(_ptr(out), _stream()), "dfn_debug_mfma_layout")
    return out
