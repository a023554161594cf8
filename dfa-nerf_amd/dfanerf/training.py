"""Training step on the HIP path: torch autograd drives, the HIP kernels compute.

  conditioning signals [96] + [42] --> FusedTrainFn (dfn_train_prepare, dfn_train_fwd | dfn_composite_bwd, dfn_mlp_bwd,
      (SignalTrainer: HIP too)          dfn_signal_grad, dfn_weight_bias_grad, dfn_fold_bias_bwd) --> rgb_head, rgb_com
                                        --> MSE losses (MseLossFn; run_nerf_com_trainExpLater.py:902-907)

The decoder's forward AND backward run in the fused HIP kernels, including the bias fold and its backward; the
decoder's gradients are deposited into the parameters' .grad as slices of one flat buffer, and autograd only carries
d(signal) on into the conditioning networks (whose forward / backward are HIP as well: SignalTrainer).
DecoderTrainFn is the same chain for Decoder.forward on explicit points under autograd (the way the reference's own
training loop calls the decoder, MAIN:855-866).  The torch-op twins the tests compare against live in tests/twins.py."""
import ctypes as C

import numpy as np
import torch

from . import engine
from ._lib import ACT_E2M1, ACT_E4M3, FIELD_HEAD, FIELD_LISTENER, FIELD_TORSO, TRAIN_ACT_E4M3, DfnTrainLoss, check, lib
from .engine import TIERS, _ptr, _stream


import os
# the head field's weight-gradient GEMMs (HBM reads) run on a second stream underneath the torso field's dX chain (MFMAs +
# HBM writes): 2.19 -> 2.12 ms per step (interleaved A/B on one box); DFN_TRAIN_OVERLAP=0 turns it off
_OVERLAP = os.environ.get("DFN_TRAIN_OVERLAP", "1") == "1"
# developer switches inside the overlapped schedule (A/B timing): the head field's weight gradients on their own stream;
# the conditioning networks' streams at high priority (their single-workgroup kernels then get the first compute unit a big
# kernel's workgroup leaves instead of queueing behind its remaining workgroups)
_WGRAD_SIDE = os.environ.get("DFN_TRAIN_WGRAD_SIDE", "1") == "1"
_SIG_PRIO = os.environ.get("DFN_TRAIN_SIG_PRIO", "1") == "1"
# the main stream joins the conditioning networks' streams at the event behind dfn_signal_grad (1) or at their end (0: A/B)
_EVENT_JOIN = os.environ.get("DFN_TRAIN_EVENT_JOIN", "1") == "1"
# the torso field's dfn_signal_grad in front of its weight-gradient GEMMs (1) or next to them (0: A/B)
_SIG_FIRST = os.environ.get("DFN_TRAIN_SIG_FIRST", "1") == "1"
# the conditioning networks' backward kernels add their gradients into zero-filled buffers (0) or write them (1: no fill
# launches in front of them).  Measured (round 4, interleaved, four rounds): writing is SLOWER - c4 0.994 -> 1.014 ms: the
# single-workgroup kernel's 690 KB of gradient stores go faster into lines the fills just brought into L2 - so 0 stays
_SIG_SET = os.environ.get("DFN_TRAIN_SIG_SET", "0") == "1"
# the audio encoder's forward keeps its activations for its backward (1) or the backward recomputes them (0: A/B)
_SIG_KEEP = os.environ.get("DFN_TRAIN_SIG_KEEP", "1") == "1"
# the main stream waits for the head's weight-gradient chain in front of the torso's REDUCTION (1) or in front of its GEMMs (0: A/B)
_SPLIT_JOIN = os.environ.get("DFN_TRAIN_SPLIT_JOIN", "1") == "1"
# the torso field's dX chain BEFORE the head field's (1; A/B): the fields' roles in the overlapped schedule swapped.  Measured (interleaved,
# two boxes, profiles/r05p_ab_schedule.txt): the hierarchical step -1 % on one box and +-0 on the other, the 64-sample step 7 % SLOWER (the
# audio encoder's backward chain then starts behind the second dX chain and the next forward waits for it): off
# The f32 tier (round 6, the kernels at the end of the round: LABNOTES 10.8, three interleaved runs):
# 7.707 -> 7.659 ms with the torso first - there the head's shorter dX chain runs beside the torso's GEMMs and the conditioning chains have a
# millisecond of GEMMs to hide under either way - so the default depends on the tier (unset: f32 on, 16-bit off).
_TORSO_FIRST = os.environ.get("DFN_TRAIN_TORSO_FIRST")
# f32 tier: the last field's narrow weight-gradient GEMMs on the side stream, beside its 256 x 256 launch (1) or behind it on the main stream (0: A/B)
_NARROW_SIDE = os.environ.get("DFN_TRAIN_NARROW_SIDE", "1") == "1"
# the step's loss from the training forward's epilogue (1: dfn_train_fwd*_loss) or from its own launch (0: dfn_mse_loss_u8; A/B)
_LOSS_IN_FWD = os.environ.get("DFN_TRAIN_LOSS_IN_FWD", "1") == "1"


def _side_stream(device, high=False, role=None):
    """engine.side_stream with the signal streams' priority switch (DFN_TRAIN_SIG_PRIO)."""
    return engine.side_stream(device, high and _SIG_PRIO, role)


def _sync_flat(params, views):
    """Make `views` (slices of one flat f32 buffer, state_dict order) hold the values of `params`.  The first time the
    tensors are f32 and contiguous the flat buffer BECOMES their storage (p.data = view): optimizers, state_dict and
    load_state_dict then work on the flat buffer directly and no per-step copy is needed; if somebody re-points a
    parameter later, the next call notices and adopts it again.  Anything else: one multi-tensor copy per call."""
    if all(p.data_ptr() == v.data_ptr() for p, v in zip(params, views)):
        return
    with torch.no_grad():
        torch._foreach_copy_(views, [p.detach() for p in params])
        if all(p.dtype == torch.float32 and p.is_contiguous() and v.is_contiguous() and p.device == v.device for p, v in zip(params, views)):
            for p, v in zip(params, views):
                p.data = v


def _pad_z(z, rows):
    """latent codes [rows, z_dim] -> [rows, 256] f32, zero-padded (a narrower code meets zero-padded weight columns: engine.flatten_state)"""
    z = z.detach().reshape(rows, -1).float()
    return (z if z.shape[1] == 256 else torch.nn.functional.pad(z, (0, 256 - z.shape[1]))).contiguous()


def _grad_buffer(owner, name, like, params):
    """Zeroed flat gradient buffer for one backward.  The SAME buffer every step while the parameters' .grad are None
    when the backward starts (optimizer.zero_grad() with set_to_none, the default): the .grad views then keep their
    addresses from step to step and optim.HipAdam reuses its device-side tensor table.  If gradients are being
    accumulated (.grad still set) the buffer of the previous step is alive inside them: a fresh one is used."""
    if any(p.grad is not None for p in params):
        return torch.zeros_like(like)
    g = getattr(owner, name, None)
    if g is None or g.shape != like.shape or g.device != like.device:
        g = torch.zeros_like(like)
        setattr(owner, name, g)
    else:
        check(lib.dfn_zero_async(_ptr(g), g.numel() * 4, _stream()), "dfn_zero_async")      # (hipMemsetAsync: no ATen launch)
    return g


class PinnedUpload:
    """numpy -> device through a small ring of pinned staging buffers, asynchronously.  torch.as_tensor(ndarray,
    device=...) copies from pageable memory, which blocks the host until the stream has drained - once per training
    step that serialises the host with the GPU."""

    def __init__(self, slots=8):
        self.bufs, self.events, self.i = [None] * slots, [None] * slots, 0

    def __call__(self, arr, dtype, device):
        arr = np.ascontiguousarray(arr)
        k, self.i = self.i, (self.i + 1) % len(self.bufs)
        b = self.bufs[k]
        if b is None or b.numel() < arr.size or b.dtype != dtype:
            b = self.bufs[k] = torch.empty(max(arr.size, 1), dtype=dtype, pin_memory=True)
            self.events[k] = None
        if self.events[k] is not None:
            self.events[k].synchronize()          # the copy that last used this slot (8 uploads ago) is long done
        b.numpy()[:arr.size] = arr.reshape(-1)
        out = b[:arr.size].to(device, non_blocking=True).view(arr.shape)
        self.events[k] = torch.cuda.Event()
        self.events[k].record()
        return out


class TrainBuffers:
    """Device buffers of one training step, sized for `n_rays` rays of 64 + n_fine samples (reused across steps).
    n_fine = 0: the reference's coarse step (MAIN:855-899); 64 / 128: the hierarchical variant (dfn_train_fwd_hier)."""

    def __init__(self, tier, n_rays, device, n_fine=0, act_format=None, n_coarse=64):
        """act_format (16-bit tier): 'fp4' = the fused step records its GEMM inputs as MX-fp4 (e2m1, the default), 'e4m3' = as
        MX-fp8 (twice the bytes, 3 mantissa bits instead of 1): the run-time opt-out, one switch for an A/B of the two on real
        data (--hip_train_act, or DFN_TRAIN_ACT in the environment when the argument is None)."""
        self.flat = None            # [955242] f32 copy of the decoder parameters (state_dict order), see bind()
        self.flat_views = None
        self.tier = TIERS[tier]
        if self.tier not in (0, 1):
            raise ValueError("the training step runs in the f32 or the bf16 tier (f16 is the inference tier: gradients "
                             "underflow its exponent range)")
        if n_fine not in (0, 64, 128):
            raise ValueError("TrainBuffers: n_fine must be 0, 64 or 128")
        if n_coarse not in (32, 64, 128) or (n_fine and n_coarse != 64):
            raise ValueError("TrainBuffers: n_coarse (--N_samples) must be 32, 64 or 128, and 64 in the hierarchical step")
        self.n_coarse = int(n_coarse)
        self.n_fine, self.S = int(n_fine), int(n_coarse) + int(n_fine)
        self.n_rays, self.NP = n_rays, n_rays * self.S
        assert self.NP % 512 == 0, "N_rand must be a multiple of 8"
        rows = lambda f, w: check(lib.dfn_train_rows(f, w), "dfn_train_rows")
        # format of the fused step's recorded activations (include/dfanerf.h: DFN_ACT_E2M1 / DFN_ACT_E4M3; DFN_TRAIN_ACT_E4M3 in the
        # tier argument of the forward selects the e4m3 recorder)
        fmt = act_format if act_format is not None else os.environ.get("DFN_TRAIN_ACT", "fp4")
        if fmt not in ("fp4", "e4m3"):
            raise ValueError(f"TrainBuffers: act_format must be 'fp4' or 'e4m3', got {fmt!r}")
        self.act_format = ACT_E2M1 if fmt == "fp4" else ACT_E4M3
        self.fwd_tier = self.tier | (TRAIN_ACT_E4M3 if (self.tier == 1 and self.act_format == ACT_E4M3) else 0)
        act_sel = 6 if self.act_format == ACT_E2M1 else 8
        if self.tier == 1:
            # 16-bit tier: the recorded arrays are MX-fp8 (e4m3 + one E8M0 scale per 32-row block and 32-point tile, the
            # operand format of the block-scaled MFMA the weight-gradient GEMMs run on): [tile][dfn_train_rows(f, 6 / 7)] bytes
            # (act_T: MX-fp4 since round 4, 16 bytes per row and tile; + one tile: the weight-gradient GEMMs' 1-KiB DMA pieces
            # may take up to 512 bytes behind an odd last 512-byte block along)
            self.act = [torch.zeros(self.NP // 32 + 1, rows(f, act_sel), dtype=torch.uint8, device=device) for f in (0, 1)]
            self.dy = [torch.empty(self.NP // 32, rows(f, 7), dtype=torch.uint8, device=device) for f in (0, 1)]
        else:
            self.act = [torch.empty(rows(f, 0), self.NP, dtype=torch.float32, device=device) for f in (0, 1)]
            self.dy = [torch.empty(rows(f, 1), self.NP, dtype=torch.float32, device=device) for f in (0, 1)]
        self.masks = [torch.empty(self.NP // 32, rows(f, 2), 64, dtype=torch.int32, device=device) for f in (0, 1)]
        self.ws = [torch.empty(rows(f, 3), dtype=torch.float32, device=device) for f in (0, 1)]
        self.ws_sig = [torch.empty(rows(f, 5), dtype=torch.float32, device=device) for f in (0, 1)]
        self.samples = torch.empty(self.NP, 8, dtype=torch.float32, device=device)
        self.dsamples = torch.empty(self.NP, 8, dtype=torch.float32, device=device)
        # hierarchical step: what the forward tells the compositing backward about the merge (depths, ranks)
        self.z_all = torch.empty(n_rays, self.S, dtype=torch.float32, device=device) if n_fine else None
        self.ranks = torch.empty(n_rays, self.S, dtype=torch.uint8, device=device) if n_fine else None
        self.packed = [torch.empty(check(lib.dfn_packed_bytes(self.tier, f), "packed"), dtype=torch.uint8, device=device)
                       for f in (0, 1)]
        self.packed_T = [torch.empty(check(lib.dfn_packed_bwd_bytes(self.tier, f), "packed_T"), dtype=torch.uint8,
                                     device=device) for f in (0, 1)]
        self.nb = [check(lib.dfn_bias_floats(self.tier, f), "bias") for f in (0, 1)]
        self.bias = torch.empty(self.nb[0] + self.nb[1], dtype=torch.float32, device=device)

    def bind(self, dec):
        """Parameter list of `dec` in state_dict order and the matching views of one flat buffer (_FlatNet.of: one per
        module, shared by every consumer): the kernels read the flat buffer, which becomes the parameters' own storage
        (_sync_flat), and write gradients into a flat buffer whose slices become the parameters' .grad."""
        fn = _FlatNet.of(dec)
        fn.refresh()
        self.net, self.flat, self.params, self.offsets = fn, fn.flat, fn.params, fn.offsets
        return self.flat


class FusedTrainFn(torch.autograd.Function):
    """(sig_head [96], sig_torso [42]) -> rgb_head, rgb_com [n,3]: dfn_train_prepare (folds + weight streams) and the fused
    forward in HIP; the backward runs dfn_composite_bwd, per field dfn_mlp_bwd, dfn_signal_grad, dfn_weight_bias_grad and
    dfn_fold_bias_bwd (spread over up to four streams, see backward()), returns the gradients of the two signals to autograd
    (-> conditioning networks) and DEPOSITS the decoder gradients straight into the parameters' .grad as slices of one
    flat buffer (side effect of backward(), like a DDP hook: ~12 launches instead of the ~600 of the torch fold +
    cat/split autograd)."""

    @staticmethod
    def forward(ctx, sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, defer=None):
        return _fused_forward(ctx, sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, defer)

    @staticmethod
    def backward(ctx, d_h, d_c):
        return _fused_backward(ctx, d_h, d_c)


def _fused_forward(ctx, sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, defer, loss=None):
    # loss: a _lib.DfnTrainLoss -> the forward's epilogue forms the step's loss and d loss / d rgb (dfn_train_fwd*_loss)
    # defer: the SignalTrainer whose _SignalFn produced both signals (their backward then picks d(signal) up on its own
    # streams and the main stream never waits for it), or None
    ctx.defer = defer if _OVERLAP else None
    t, st = buf.tier, _stream()
    flat = buf.flat
    dev = flat.device
    sh = sig_head.detach().reshape(-1).float().contiguous()
    stt = sig_torso.detach().reshape(-1).float().contiguous()
    zs, za = _pad_z(z_shape, 2), _pad_z(z_app, 2)
    bias = buf.bias
    bias_t = C.c_void_p(bias.data_ptr() + 4 * buf.nb[0])
    # both folds and the four packed weight streams in one launch (six launches back to back cost 37 us of the step)
    check(lib.dfn_train_prepare(t, _ptr(flat), _ptr(sh), _ptr(stt), _ptr(zs), _ptr(za), _ptr(buf.packed[0]),
                                _ptr(buf.packed[1]), _ptr(buf.packed_T[0]), _ptr(buf.packed_T[1]), _ptr(bias), bias_t,
                                st), "dfn_train_prepare")
    n = frame.ray_count
    rgb_h = torch.empty(n, 3, dtype=torch.float32, device=dev)
    rgb_c = torch.empty(n, 3, dtype=torch.float32, device=dev)
    bg_f32 = bg if bg.dtype == torch.float32 else None
    bg_u8 = bg if bg.dtype == torch.uint8 else None
    if loss is not None and buf.n_fine:
        check(lib.dfn_train_fwd_hier_loss(buf.fwd_tier, C.byref(frame), _ptr(buf.packed[0]), _ptr(buf.packed[1]), _ptr(bias), bias_t,
                                          _ptr(bg_f32), _ptr(bg_u8), _ptr(pix_index), _ptr(rgb_h), _ptr(rgb_c),
                                          _ptr(buf.samples), _ptr(buf.act[0]), _ptr(buf.masks[0]), _ptr(buf.act[1]),
                                          _ptr(buf.masks[1]), _ptr(buf.z_all), _ptr(buf.ranks), C.byref(loss), st),
              "dfn_train_fwd_hier_loss")
    elif loss is not None:
        check(lib.dfn_train_fwd_loss(buf.fwd_tier, C.byref(frame), _ptr(buf.packed[0]), _ptr(buf.packed[1]), _ptr(bias), bias_t,
                                     _ptr(bg_f32), _ptr(bg_u8), _ptr(pix_index), _ptr(rgb_h), _ptr(rgb_c),
                                     _ptr(buf.samples), _ptr(buf.act[0]), _ptr(buf.masks[0]), _ptr(buf.act[1]),
                                     _ptr(buf.masks[1]), C.byref(loss), st), "dfn_train_fwd_loss")
    elif buf.n_fine:
        check(lib.dfn_train_fwd_hier(buf.fwd_tier, C.byref(frame), _ptr(buf.packed[0]), _ptr(buf.packed[1]), _ptr(bias), bias_t,
                                     _ptr(bg_f32), _ptr(bg_u8), _ptr(pix_index), _ptr(rgb_h), _ptr(rgb_c),
                                     _ptr(buf.samples), _ptr(buf.act[0]), _ptr(buf.masks[0]), _ptr(buf.act[1]),
                                     _ptr(buf.masks[1]), _ptr(buf.z_all), _ptr(buf.ranks), st), "dfn_train_fwd_hier")
    else:
        check(lib.dfn_train_fwd(buf.fwd_tier, C.byref(frame), _ptr(buf.packed[0]), _ptr(buf.packed[1]), _ptr(bias), bias_t,
                                _ptr(bg_f32), _ptr(bg_u8), _ptr(pix_index), _ptr(rgb_h), _ptr(rgb_c),
                                _ptr(buf.samples), _ptr(buf.act[0]), _ptr(buf.masks[0]), _ptr(buf.act[1]),
                                _ptr(buf.masks[1]), st), "dfn_train_fwd")
    ctx.buf, ctx.frame, ctx.bg, ctx.pix = buf, frame, bg, pix_index
    # the recorded arrays (and FusedTrainLossFn's d loss / d rgb) live in `buf` and are overwritten by the next forward
    # through it: a backward is only valid for the LAST forward (no gradient accumulation over micro-batches through one
    # TrainBuffers, no retain_graph across forwards) - stamped here, checked in _fused_backward
    buf._fwd_seq = ctx.seq = getattr(buf, "_fwd_seq", 0) + 1
    ctx.keep = (sh, stt, zs, za)
    ctx.sig_shapes = (sig_head.shape, sig_torso.shape)
    return rgb_h, rgb_c


def _fused_backward(ctx, d_h, d_c):
    buf, frame, bg, st = ctx.buf, ctx.frame, ctx.bg, _stream()
    if buf._fwd_seq != ctx.seq:
        raise RuntimeError("fused training step: backward of forward #%d, but forward #%d has since overwritten the "
                           "recorded activations of this TrainBuffers (one forward per backward; use a second TrainBuffers "
                           "for a second graph)" % (ctx.seq, buf._fwd_seq))
    sh, stt, zs, za = ctx.keep
    flat, dev = buf.flat, buf.flat.device
    d_h = d_h.contiguous().float()
    d_c = d_c.contiguous().float()
    bg_f32 = bg if bg.dtype == torch.float32 else None
    bg_u8 = bg if bg.dtype == torch.uint8 else None
    # the step's flat gradient buffer: the SAME one every step while the parameters' .grad are None when the backward starts
    # (_grad_buffer) - then the compositing backward's launch zeroes it on the way (dfn_composite_bwd*_z: its own fill was a
    # 9-us launch + a gap in front of the dX chain); a fresh buffer (gradient accumulation) comes zeroed from the allocator
    g_flat = getattr(buf.net, "_g_flat", None)
    # (dfn_composite_bwd*_z wants a 16-byte aligned fill: a gradient buffer adopted by parallel.FlatGradBucket is a slice of the
    # bucket at an arbitrary element offset - when the decoder does not come first it is only 4-byte aligned: plain fill then)
    reuse = g_flat is not None and g_flat.shape == flat.shape and g_flat.device == dev and g_flat.data_ptr() % 16 == 0 and \
        not any(p.grad is not None for p in buf.params)
    if not reuse:
        g_flat = _grad_buffer(buf.net, "_g_flat", flat, buf.params)
    zb, zn = (_ptr(g_flat), g_flat.numel()) if reuse else (None, 0)
    if buf.n_fine:
        check(lib.dfn_composite_bwd_hier_z(C.byref(frame), _ptr(ctx.pix), _ptr(bg_f32), _ptr(bg_u8), _ptr(buf.samples),
                                           _ptr(buf.z_all), _ptr(buf.ranks), _ptr(d_h), _ptr(d_c), _ptr(buf.dsamples), zb, zn,
                                           st), "dfn_composite_bwd_hier_z")
    else:
        check(lib.dfn_composite_bwd_z(C.byref(frame), _ptr(ctx.pix), _ptr(bg_f32), _ptr(bg_u8), _ptr(buf.samples),
                                      _ptr(d_h), _ptr(d_c), _ptr(buf.dsamples), zb, zn, st), "dfn_composite_bwd_z")
    g_bias = torch.empty(buf.nb[0] + buf.nb[1], dtype=torch.float32, device=dev)
    main = torch.cuda.current_stream(dev)
    # (with more than one rank RCCL brings a fifth stream; the package asks the runtime for eight hardware queues then
    # (dfanerf/__init__.py) - on the default four the weight-gradient stream shares the main stream's queue and this overlap
    # is lost: 1.32 -> 1.36 ms per step through RCCL on one GPU, LABNOTES.md 6)
    over = _OVERLAP and _WGRAD_SIDE
    if over and getattr(buf, "_side", None) is None:
        buf._side = _side_stream(dev, role="wgrad")
    side = buf._side if over else None
    # (g_flat: zeroed above, on the main stream: a many-workgroup fill on a side stream starves behind the dX chain's
    # workgroups - measured: 340 us for this 4-MB fill, and the head field's weight gradients queue behind it)
    if _OVERLAP:
        if getattr(buf, "_sig_streams", None) is None:
            buf._sig_streams = (_side_stream(dev, True, "sig_a"), _side_stream(dev, True, "sig_p"))
        tr = ctx.defer
        s_a, s_p = (tr.audio_stream(), tr.pose_stream()) if tr is not None else buf._sig_streams
        # dfn_signal_grad overwrites its half.  Deferred to a SignalTrainer the buffer is the trainer's own: its readers
        # run on the trainer's streams and - pipelined - nothing orders the main stream behind them before the next
        # encode(), so a buffer from the caching allocator would be handed out again (to the next step's pixel upload,
        # say) while the encoder backward still reads it.  Otherwise a fresh one (autograd may keep it as a .grad); the
        # waits below order the main stream behind its writers.
        if tr is not None:
            if getattr(tr, "_d_sig", None) is None:
                tr._d_sig = torch.empty(96 + 42, dtype=torch.float32, device=dev)
            d_sig = tr._d_sig
        else:
            d_sig = torch.empty(96 + 42, dtype=torch.float32, device=dev)
    else:
        d_sig = torch.empty(96 + 42, dtype=torch.float32, device=dev)
    def dx(f, stream):
        check(lib.dfn_mlp_bwd(buf.tier, f, _ptr(buf.packed_T[f]), _ptr(buf.samples), _ptr(buf.dsamples),
                              _ptr(buf.masks[f]), buf.NP, _ptr(buf.dy[f]), stream), "dfn_mlp_bwd")

    def dw(f, stream, g, with_sig, before_reduce=None, narrow_on=None):
        gb = C.c_void_p(g_bias.data_ptr() + (4 * buf.nb[0] if f else 0))
        if before_reduce is not None and narrow_on is not None:
            # f32 tier, the LAST field: its narrow GEMMs (HBM-bound, 72 KiB of LDS) on the other field's stream, beside its own 256 x 256
            # launch (matrix-pipe-bound, 64 KiB) instead of behind it - a workgroup of each fits one compute unit, and no fifth queue
            # (LABNOTES 10.8).  `narrow_on` = (stream, event): the event orders the side stream behind this field's dX chain.
            n_stream, n_ev = narrow_on
            n_ev.record(main)
            n_stream.wait_event(n_ev)
            for which, st_ in ((2, C.c_void_p(n_stream.cuda_stream)), (1, stream)):
                check(lib.dfn_weight_bias_grad_partials_part(buf.tier, f, buf.act_format, _ptr(buf.dy[f]), _ptr(buf.act[f]), buf.NP,
                                                             _ptr(buf.ws[f]), gb, which, st_), "dfn_weight_bias_grad_partials_part")
            before_reduce()
            check(lib.dfn_weight_bias_grad_reduce(buf.tier, f, buf.NP, _ptr(buf.ws[f]), _ptr(g), gb, stream),
                  "dfn_weight_bias_grad_reduce")
        elif before_reduce is None:
            check(lib.dfn_weight_bias_grad_fmt(buf.tier, f, buf.act_format, _ptr(buf.dy[f]), _ptr(buf.act[f]), buf.NP,
                                               _ptr(buf.ws[f]), _ptr(g), gb, stream), "dfn_weight_bias_grad_fmt")
        else:
            # the two stages apart: the GEMMs only write their own workspace; what must not overtake the other field's
            # reduction (the fields share most parameters: one += after the other, in a fixed order) is the REDUCTION
            check(lib.dfn_weight_bias_grad_partials(buf.tier, f, buf.act_format, _ptr(buf.dy[f]), _ptr(buf.act[f]), buf.NP,
                                                    _ptr(buf.ws[f]), gb, stream), "dfn_weight_bias_grad_partials")
            before_reduce()
            check(lib.dfn_weight_bias_grad_reduce(buf.tier, f, buf.NP, _ptr(buf.ws[f]), _ptr(g), gb, stream),
                  "dfn_weight_bias_grad_reduce")
        ds = C.c_void_p(d_sig.data_ptr() + (4 * 96 if f else 0)) if with_sig else None
        check(lib.dfn_fold_bias_bwd(buf.tier, FIELD_TORSO if f else FIELD_HEAD, _ptr(flat), _ptr(stt if f else sh),
                                    _ptr(zs[f]), _ptr(za[f]), gb, _ptr(g), ds, stream), "dfn_fold_bias_bwd")

    def dsig(f, stream):
        # d(signal) from the row sums of the dy_T rows behind it, without waiting for the weight gradients
        check(lib.dfn_signal_grad(buf.tier, f, _ptr(flat), _ptr(buf.dy[f]), buf.NP, _ptr(buf.ws_sig[f]),
                                  C.c_void_p(d_sig.data_ptr() + (4 * 96 if f else 0)), stream), "dfn_signal_grad")
    if _OVERLAP:
        # Three chains next to the main one.  (1) the head field's weight gradients (HBM reads) on a side stream
        # underneath the torso field's dX chain (MFMAs + HBM writes); the main stream waits for that chain before the
        # torso field's weight gradients, so the two fields accumulate into the one gradient buffer in a fixed order
        # (bit-reproducible; the side chain ends well before the dX chain it runs under).  (2, 3) d(signal) of each
        # field as soon as its dX chain is done (dfn_signal_grad) on the conditioning networks' streams: their backward
        # (single-workgroup latency chains, 0.26 ms) then runs underneath the weight-gradient GEMMs instead of behind
        # them.  Who consumes d_sig decides whether the main stream has to wait: _SignalFn.backward launches on those
        # same streams (ctx.defer).
        # (events behind each dfn_signal_grad: its fold backward is the only reader of the DECODER's parameters on the side
        # streams, so that is all the decoder's Adam has to stay behind - not the conditioning networks' whole backward chains)
        ev = None
        if tr is not None:
            ev = getattr(tr, "_dsig_ev", None)
            if ev is None:
                ev = tr._dsig_ev = (torch.cuda.Event(), torch.cuda.Event())
        torso_first = (buf.tier == 0) if _TORSO_FIRST is None else _TORSO_FIRST == "1"
        if tr is not None:
            tr._torso_first = torso_first
        fa, fb = (1, 0) if torso_first else (0, 1)           # the field whose dX chain runs first / second
        sig_st = {0: s_a, 1: s_p}
        dx(fa, st)
        # ONE event behind the first dX chain for both side chains (every record is a packet in the main queue in front of the
        # second dX chain: two cost 14 us between the two dX kernels)
        e_dx = getattr(buf, "_ev_dx", None)
        if e_dx is None:
            e_dx = buf._ev_dx = torch.cuda.Event()
        # (the first field's d(signal) stays on its encoder's stream: on the main stream between the two dX chains - what pays
        # for the second's, below - the step was 40 us LONGER, 1.02 -> 1.06 ms: the main chain is the critical one)
        e_dx.record(main)
        sig_st[fa].wait_event(e_dx)
        dsig(fa, C.c_void_p(sig_st[fa].cuda_stream))
        if ev is not None:
            ev[fa].record(sig_st[fa])
        if over:
            side.wait_event(e_dx)
            dw(fa, C.c_void_p(side.cuda_stream), g_flat, False)
        else:
            dw(fa, st, g_flat, False)
        dx(fb, st)
        if ev is not None and _SIG_FIRST:
            # The second field's d(signal) (row sums, their reduction, the fold backward: three small kernels, ~30 us) ON the main
            # stream, in front of its weight-gradient GEMMs, not next to them on its encoder's stream: the GEMMs'
            # workgroups own the compute units (144 KiB of LDS each) until they finish, so a kernel launched next to them
            # starts when they end - the pose network's whole backward chain (row sums -> fold backward -> encoder backward ->
            # Adam -> the next step's encoder forward, ~130 us) then ran BEHIND the GEMMs and the next step's forward waited
            # for it.  On the main stream it also costs no cross-queue hand-over (a wait on another queue's event is 15-20 us:
            # dX -> pose stream -> main was 59 us between the dX chain and the GEMMs).  The encoder's stream picks d(signal) up
            # behind ev[fb] (_SignalFn.backward).
            dsig(fb, st)
            ev[fb].record(main)
        else:
            sig_st[fb].wait_stream(main)
            dsig(fb, C.c_void_p(sig_st[fb].cuda_stream))
            if ev is not None:
                ev[fb].record(sig_st[fb])
        if over and _SPLIT_JOIN:
            # the main stream joins the first field's chain (GEMMs, reduction, fold backward on the side stream) in front of the
            # second's REDUCTION, not in front of its GEMMs: the cross-queue wait sat between d(signal) and a 130-us kernel that
            # does not depend on it - 17 us of the critical path (profiles/r04g_c4_timeline.txt)
            narrow_on = None
            if buf.tier == 0 and _NARROW_SIDE:
                e_n = getattr(buf, "_ev_narrow", None)
                if e_n is None:
                    e_n = buf._ev_narrow = torch.cuda.Event()
                narrow_on = (side, e_n)
            dw(fb, st, g_flat, False, before_reduce=lambda: main.wait_stream(side), narrow_on=narrow_on)
        else:
            if over:
                main.wait_stream(side)
            dw(fb, st, g_flat, False)
        if tr is None:          # torch autograd consumes d_sig on the main stream
            main.wait_stream(s_a)
            main.wait_stream(s_p)
        else:
            tr._deferred = True
    else:
        # one stream, the SAME kernels in the same order per buffer (d(signal) through dfn_signal_grad here too): bit for
        # bit what the overlapped schedule computes - the reference the stream-schedule test holds it against
        for f in (0, 1):
            dx(f, st)
            dsig(f, st)
            dw(f, st, g_flat, False)
    buf.net.deposit(g_flat, touched=_decoder_touched(buf.net, (0, 1)))
    return (d_sig[:96].reshape(ctx.sig_shapes[0]), d_sig[96:].reshape(ctx.sig_shapes[1]), None, None, None, None,
            None, None, None)


class _FlatNet:
    """Flat f32 copy of one network's parameters (state_dict order) + the matching gradient buffer.  One per module
    (_FlatNet.of): the flat buffer becomes the parameters' storage, so two binders of one module would fight over it."""

    def __init__(self, module):
        self.module = module
        sd = module.state_dict(keep_vars=True)
        skip = tuple(getattr(module, "_dfn_flat_skip", ()))      # (decoder: layers the kernels never evaluate, _lib.DECODER_UNUSED_PREFIXES)
        if skip:
            sd = {k: v for k, v in sd.items() if not k.startswith(skip)}
        self.names, self.params = list(sd.keys()), list(sd.values())
        # (owning sub-module, attribute name) of every entry, for the cheap identity check of of()
        self.slots = []
        for name in self.names:
            mod, _, key = name.rpartition(".")
            self.slots.append((module.get_submodule(mod) if mod else module, key))
        self._dep = None
        dev = self.params[0].device
        # A module may ask for a PADDED layout (Decoder._dfn_flat_layout: a narrower decoder, or one without the deformation field,
        # inside the library's 256-wide flat vector): the flat buffer is then a zero-filled vector of the library's size, every
        # parameter a strided corner of its padded slot - copied in before a step (refresh), its gradient copied out after it (deposit);
        # the padded entries stay zero (relu'(0) = 0: their gradients are zero, and nothing ever steps them).
        lay = getattr(module, "_dfn_flat_layout", None)
        lay = lay() if callable(lay) else None
        self.padded = lay is not None
        if self.padded:
            from ._lib import N_DECODER_PARAMS
            o, self.shapes = lay
            assert len(self.shapes) == len(self.params)
            self.flat = torch.zeros(N_DECODER_PARAMS, dtype=torch.float32, device=dev)
            self.offsets, self.views = [], []
            for p, shp in zip(self.params, self.shapes):
                self.offsets.append(o)
                self.views.append(self._corner(self.flat, o, shp, p))
                o += int(np.prod(shp))
            assert o == N_DECODER_PARAMS, (o, N_DECODER_PARAMS)
            return
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o)
            o += p.numel()
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]

    @staticmethod
    def _corner(flat, o, shp, p):
        """parameter p's values inside its padded slot [o, o + prod(shp)) of `flat`: the leading corner of the slot viewed as `shp`"""
        return flat[o:o + int(np.prod(shp))].view(shp)[tuple(slice(0, n) for n in p.shape)]

    @staticmethod
    def of(module):
        fn = module.__dict__.get("_dfn_flat")
        if fn is not None:
            # per-step check: the same Parameter objects, read straight from the modules that own them (no state_dict
            # walk: that costs 0.15 ms per training step over the five networks)
            for (mod, key), p in zip(fn.slots, fn.params):
                if (mod._parameters.get(key) if key in mod._parameters else mod._buffers.get(key)) is not p:
                    fn = None
                    break
            if fn is not None and fn.params[0].device != fn.flat.device:
                fn = None
        if fn is None:
            fn = module.__dict__["_dfn_flat"] = _FlatNet(module)
        return fn

    def refresh(self):
        _sync_flat(self.params, self.views)

    def deposit(self, grad_flat, touched=None):
        """.grad of every parameter (of those in `touched`, a list of bools, if given) = its slice of grad_flat (added
        to an existing .grad).  Parameters outside `touched` keep .grad = None, as torch autograd leaves parameters a
        forward never used (the listener layers; the other field's layers when one field is evaluated)."""
        # the same view OBJECTS while the buffer and the selection stay (optim.HipAdam recognises an unchanged step by them)
        key = (grad_flat.data_ptr(), id(touched))
        if self.padded:
            # gradients of the padded layout: one multi-tensor copy of the parameters' corners into contiguous tensors of their own
            # (the same ones every step while the buffer and the selection stay: optim.HipAdam keeps its table)
            if self._dep is None or self._dep[0] != key:
                sel = [(p, self._corner(grad_flat, o, shp, p)) for i, (p, o, shp) in enumerate(zip(self.params, self.offsets, self.shapes))
                       if p.requires_grad and (touched is None or touched[i])]
                self._dep = (key, [p for p, _ in sel], [c for _, c in sel], [torch.empty_like(p) for p, _ in sel], grad_flat)
            _, ps, corners, own, _ = self._dep
            fresh = [i for i, p in enumerate(ps) if p.grad is None]
            if fresh:
                torch._foreach_copy_([own[i] for i in fresh], [corners[i] for i in fresh])
            for i, p in enumerate(ps):
                if p.grad is None:
                    p.grad = own[i]
                else:
                    p.grad.add_(corners[i])
            return
        if self._dep is None or self._dep[0] != key:
            views = [(p, grad_flat[o:o + p.numel()].view_as(p)) for i, (p, o) in enumerate(zip(self.params, self.offsets))
                     if p.requires_grad and (touched is None or touched[i])]
            self._dep = (key, views, grad_flat)
        for p, g in self._dep[1]:
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)


_HEAD_ONLY = ("fc_in.", "fc_p_skips.")
_TORSO_ONLY = ("deform_net.", "fc_in_torso.", "fc_p_skips_torso.")
_LISTENER = ("fc_in_listener.", "fc_p_skips_listener.")


def _decoder_touched(net, fields):
    """Per parameter of the decoder: does a forward through `fields` (0 head, 1 torso, 2 listener) use it?  (decoder.py:297-325)"""
    key = ("touched", tuple(fields))
    hit = net.__dict__.get(key)
    if hit is None:
        def used(name):
            if name.startswith(_LISTENER):
                return 2 in fields
            if name.startswith(_HEAD_ONLY):
                return 0 in fields
            if name.startswith(_TORSO_ONLY):
                return 1 in fields
            return True
        hit = net.__dict__[key] = [used(n) for n in net.names]
    return hit


class SignalTrainer:
    """Rows A7 / A8 of the training step in HIP: dfn_encode_signal* forward and dfn_encode_signal*_bwd backward for the
    frame being trained (4 launches instead of the ~200 of the torch modules + autograd).  The returned signals carry
    an autograd node whose backward deposits the four networks' gradients into their parameters' .grad."""

    def __init__(self, aud_net, exp_net, att_net, pose_att_net, auds, exps, poses):
        engine.require_gpu()
        self.nets = [_FlatNet.of(m) for m in (aud_net, exp_net, att_net, pose_att_net)]
        dev = auds.device
        self.auds, self.exps = auds.detach().float().contiguous(), exps.detach().float().contiguous()
        self.poses = poses.detach().float().contiguous()
        self.pose_stride = int(self.poses[0].numel())
        self.device = dev
        self._pipelined, self._fresh = False, True

    def audio_stream(self):
        """stream of the audio / expression encoder's BACKWARD (None: DFN_TRAIN_OVERLAP=0); its forward runs on the caller's
        stream (the decoder's forward waits for it anyway)"""
        return self._streams()[0]

    def pose_stream(self):
        """second stream for the pose encoder, forward and backward (None: DFN_TRAIN_OVERLAP=0)"""
        return self._streams()[1]

    def _streams(self):
        if not _OVERLAP:
            return None, None
        if getattr(self, "_side", None) is None:
            self._side = (_side_stream(self.device, True, "sig_a"), _side_stream(self.device, True, "sig_p"))
        return self._side

    def adopt_optimizers(self, opts):
        """Pipeline the conditioning networks across steps: their optimizers (dict name -> optim.HipAdam, the keys of
        run_nerf.create_nerf) step on the streams their gradients are produced on, and encode() then runs the next step's
        forward there too - all of it underneath the decoder's weight-gradient GEMMs of the step before; the main stream
        only waits for the two signals.  Everything that writes these networks' parameters must go through those optimizers
        (or call resync() afterwards)."""
        if not _OVERLAP:
            return
        s_a, s_p = self._streams()
        for k, s in (("AudNet", s_a), ("ExpNet", s_a), ("AudAttNet", s_a), ("PoseAttNet", s_p)):
            if k in opts and hasattr(opts[k], "_step"):
                opts[k].dfn_stream = s
                opts[k].dfn_join_later = True       # encode() / join() order the main stream behind the update
        self._pipelined = True
        self._fresh = True

    def join(self):
        """Order the current stream behind everything queued on the conditioning networks' streams (their backward, Adam,
        the next forward).  encode() does it for the training step; call it before anything ELSE reads these networks'
        parameters or gradients on the current stream (a periodic test render, a checkpoint, gradient clipping)."""
        cur = torch.cuda.current_stream(self.device)
        for s in self._streams():
            if s is not None:
                cur.wait_stream(s)

    def resync(self):
        """the parameters were written on the current stream (load_state_dict, ...): the next encode() waits for it"""
        self._fresh = True

    def frame_id(self, frame):
        """[1] int32 device tensor holding `frame` without a host-to-device copy (a pageable copy blocks the host until
        the stream has drained: it serialises every step with the previous one)."""
        ar = getattr(self, "_frame_ids", None)
        if ar is None or frame >= ar.numel():
            ar = self._frame_ids = torch.arange(max(int(frame) + 1, 4096), dtype=torch.int32, device=self.device)
            self._fresh = True          # made on the current stream: the encoder streams wait for it once
        return ar[frame:frame + 1]

    def encode(self, frame, smo_size, smo_torso_size, length):
        for n in self.nets:
            n.refresh()
        if getattr(self, "_anchor", None) is None:
            self._anchor = torch.zeros(1, device=self.device, requires_grad=True)       # keeps the node in the graph
            # two sets of outputs, used alternately: with adopt_optimizers() the next step's forward runs while the
            # previous step's fold backward may still have to read its signals
            self._outs = [(torch.empty(1, 96, dtype=torch.float32, device=self.device),
                           torch.empty(1, 42, dtype=torch.float32, device=self.device)) for _ in range(2)]
            self._flip = 0
        return _SignalFn.apply(self._anchor, self, int(frame), int(smo_size), int(smo_torso_size), int(length))


class _SignalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, tr, frame, smo, smo_t, length):
        dev, st = tr.device, _stream()
        ids = tr.frame_id(frame)
        tr._flip ^= 1
        sig, sigt = tr._outs[tr._flip]
        a, e, t, p = [n.flat for n in tr.nets]
        # the two encoders are independent single-workgroup latency chains: the pose one runs on a second stream
        # underneath the audio / expression one.  Pipelined (adopt_optimizers): both run on the streams their parameters
        # were just updated on, behind the previous step's backward + Adam there and ahead of whatever the main stream is
        # still doing for the previous step; it waits for them only on the first call (parameters loaded on the main stream)
        s_a, s_p = tr._streams()
        main = torch.cuda.current_stream(dev)
        piped = s_a is not None and getattr(tr, "_pipelined", False)
        if s_p is not None and (not piped or tr._fresh):
            s_p.wait_stream(main)
        if piped and tr._fresh:
            s_a.wait_stream(main)
        tr._fresh = False
        st_t = st if s_p is None else C.c_void_p(s_p.cuda_stream)
        st_a = C.c_void_p(s_a.cuda_stream) if piped else st
        check(lib.dfn_encode_signal_torso(_ptr(p), _ptr(tr.poses), tr.pose_stride, length, _ptr(ids), 1, smo_t,
                                          _ptr(sigt), st_t), "dfn_encode_signal_torso")
        # the audio encoder also leaves the activations its backward needs (round 4: that backward is the step's one exposed
        # side chain, and 40 % of it re-ran this forward): valid for the LAST encode only (ctx.keep_seq)
        if _SIG_KEEP:
            if getattr(tr, "_keep", None) is None:
                tr._keep = torch.empty(check(lib.dfn_encode_signal_keep_floats(), "keep"), dtype=torch.float32, device=dev)
                tr._keep_seq = 0
            tr._keep_seq += 1
            ctx.keep_seq = tr._keep_seq
            check(lib.dfn_encode_signal_keep(_ptr(a), _ptr(e), _ptr(t), _ptr(tr.auds), _ptr(tr.exps), length, _ptr(ids), smo,
                                             _ptr(sig), _ptr(tr._keep), st_a), "dfn_encode_signal_keep")
        else:
            ctx.keep_seq = -1
            check(lib.dfn_encode_signal(_ptr(a), _ptr(e), _ptr(t), _ptr(tr.auds), _ptr(tr.exps), length, _ptr(ids), 1, smo,
                                        _ptr(sig), st_a), "dfn_encode_signal")
        if s_p is not None:
            main.wait_stream(s_p)
        if piped:
            main.wait_stream(s_a)
        ctx.tr, ctx.args = tr, (frame, smo, smo_t, length)
        return sig, sigt

    @staticmethod
    def backward(ctx, d_sig, d_sigt):
        tr, (frame, smo, smo_t, length) = ctx.tr, ctx.args
        st = _stream()
        a, e, t, p = [n.flat for n in tr.nets]
        # d_sig / d_sigt were produced on the audio / pose streams by FusedTrainFn.backward (dfn_signal_grad, right after each
        # field's dX chain) when it deferred to this trainer (tr._deferred): everything below, the zeroing of the gradient
        # buffers included, is then launched on those streams and never waits for the main one, which meanwhile runs the
        # weight-gradient GEMMs; the main stream joins them at the end.  A d_sig from anywhere else lives on the main
        # stream: the side streams wait for it first.
        s_a, s_p = tr.audio_stream(), tr.pose_stream()
        main = torch.cuda.current_stream(tr.device)
        deferred, tr._deferred = getattr(tr, "_deferred", False), False
        if s_a is not None and not deferred:
            s_a.wait_stream(main)
            s_p.wait_stream(main)
        elif s_p is not None and getattr(tr, "_dsig_ev", None) is not None:
            s_p.wait_event(tr._dsig_ev[1])          # (the torso's d(signal) may have been produced on the main stream: _SIG_FIRST)
            if getattr(tr, "_torso_first", False):
                s_a.wait_event(tr._dsig_ev[0])      # (the head's then: the torso's dX chain ran first, _TORSO_FIRST)
        st_a = st if s_a is None else C.c_void_p(s_a.cuda_stream)
        st_t = st if s_p is None else C.c_void_p(s_p.cuda_stream)
        def buffers(side, stream, nets):
            # _grad_buffer on an explicit stream (dfn_zero_async: no torch stream context per fill)
            out = []
            for n in nets:
                g = getattr(n, "_g_flat", None)
                if g is None or g.shape != n.flat.shape or g.device != n.flat.device or \
                        any(q.grad is not None for q in n.params):
                    g = torch.empty_like(n.flat)
                    if side is not None:
                        side.wait_stream(main)      # fresh memory of the main stream's pool: its last user ran there
                    if not any(q.grad is not None for q in n.params):
                        n._g_flat = g
                if not _SIG_SET:
                    check(lib.dfn_zero_async(_ptr(g), g.numel() * 4, stream), "dfn_zero_async")
                out.append(g)
            return out
        g = buffers(s_a, st_a, tr.nets[:3]) + buffers(s_p, st_t, tr.nets[3:])
        d_sig = d_sig.contiguous().float()
        d_sigt = d_sigt.contiguous().float()
        # (*_set: the kernels WRITE the gradients - one writer per element and call -: no fill launches in front of them; a
        # network that takes no part (attention before --nosmo_iters) is not written and not deposited below)
        bwd_t = lib.dfn_encode_signal_torso_bwd_set if _SIG_SET else lib.dfn_encode_signal_torso_bwd
        bwd_a = lib.dfn_encode_signal_bwd_set if _SIG_SET else lib.dfn_encode_signal_bwd
        check(bwd_t(_ptr(p), _ptr(tr.poses), tr.pose_stride, length, frame, smo_t, _ptr(d_sigt), _ptr(g[3]), st_t),
              "dfn_encode_signal_torso_bwd")
        if not _SIG_SET and ctx.keep_seq >= 0 and ctx.keep_seq == getattr(tr, "_keep_seq", -2):
            check(lib.dfn_encode_signal_bwd_kept(_ptr(a), _ptr(e), _ptr(t), _ptr(tr.auds), _ptr(tr.exps), length, frame, smo,
                                                 _ptr(d_sig), _ptr(tr._keep), _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), st_a),
                  "dfn_encode_signal_bwd_kept")
        else:
            check(bwd_a(_ptr(a), _ptr(e), _ptr(t), _ptr(tr.auds), _ptr(tr.exps), length, frame, smo, _ptr(d_sig), _ptr(g[0]),
                        _ptr(g[1]), _ptr(g[2]), st_a), "dfn_encode_signal_bwd")
        if s_a is not None:
            # Order the main stream behind these chains - ALWAYS: dfn_signal_grad's fold backward reads the DECODER's
            # parameters on these streams, and the decoder's Adam (main stream) must not overtake it.  (Leaving the join to
            # the next encode() saved two queue barriers and was a race: with 256 rays the main stream reaches Adam before
            # the starved side kernels have run; the 3000-step bitwise soak of tests/test_gpu_train.py caught it.)
            # Round 4: when d(signal) was deferred to this trainer, the readers of the decoder's parameters are the two
            # dfn_signal_grad calls FusedTrainFn.backward issued, and it recorded an event behind each: the main stream waits for
            # those instead of for the whole chains (the torso side's row sums run next to the torso's weight-gradient GEMMs and
            # its encoder backward ended 45 us after them: that tail was on the step's critical path for no reason).
            ev = getattr(tr, "_dsig_ev", None)
            # (only with adopted optimizers - tr._pipelined -: the conditioning networks' Adam then steps on these streams;
            # an optimizer on the main stream needs the whole chain behind it, and so does anything that reads their .grad)
            if deferred and ev is not None and _EVENT_JOIN and getattr(tr, "_pipelined", False):
                main.wait_event(ev[0])
                main.wait_event(ev[1])
            else:
                main.wait_stream(s_a)
                main.wait_stream(s_p)
        tr.nets[0].deposit(g[0])
        tr.nets[1].deposit(g[1])
        if smo > 0:
            tr.nets[2].deposit(g[2])
        if smo_t > 0:
            tr.nets[3].deposit(g[3])
        return None, None, None, None, None, None


def render_train(dec, buf, frame, bg, pix_index, sig_head, sig_torso, z_shape, z_app, signal_trainer=None):
    """Differentiable (w.r.t. the decoder parameters and the two signals) two-field render of the
    pixels `pix_index` [n] (int32, y*W+x): coarse (frame.n_fine == 0, the reference's step) or hierarchical (64 / 128
    fine samples at detached depths; buf = TrainBuffers(..., n_fine=...)).  Returns rgb_head, rgb_com [n,3].  Fold and its backward run in HIP, the
    decoder gradients are deposited into .grad by the backward (FusedTrainFn).  One forward per backward: the recorded
    activations live in `buf` and the next forward overwrites them.  signal_trainer: the SignalTrainer whose encode()
    produced BOTH signals for this call (None otherwise): d(signal) then stays on its streams (FusedTrainFn.backward)."""
    if frame.ray_count != buf.n_rays or (pix_index is not None and pix_index.numel() != buf.n_rays):
        raise ValueError(f"render_train: the buffers were sized for {buf.n_rays} rays, the frame has {frame.ray_count}"
                         f" (pix_index {None if pix_index is None else pix_index.numel()})")
    if frame.n_fine != buf.n_fine or frame.n_coarse != buf.n_coarse or frame.fields != 2:
        raise ValueError(f"render_train: the buffers were sized for {buf.n_coarse} + {buf.n_fine} samples, the frame asks for "
                         f"{frame.n_coarse} + {frame.n_fine} (fields {frame.fields}: the training step renders both fields)")
    buf.bind(dec)
    if not sig_head.requires_grad:      # keep the Function in the graph even when no conditioning net trains
        sig_head = sig_head.detach().requires_grad_(True)
    if signal_trainer is not None:
        # d(signal) may stay on the trainer's streams only if NOTHING sits between its autograd node and this one: any
        # torch op in between (an index, a reshape) would run its backward on the main stream, ahead of the side streams
        node = sig_head.grad_fn
        if node is None or node is not sig_torso.grad_fn or type(node).__name__ != "_SignalFnBackward":
            signal_trainer = None
    return FusedTrainFn.apply(sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, signal_trainer)


def render_train_loss(dec, buf, frame, bg, pix_index, sig_head, sig_torso, z_shape, z_app, img_head, img_com,
                      signal_trainer=None):
    """render_train + the step's loss against uint8 ground-truth frames [H*W,3] resident on the device, as one autograd node
    (FusedTrainLossFn): -> loss (= loss_com + loss_head), loss_head, loss_com, rgb_head, rgb_com."""
    if frame.ray_count != buf.n_rays or pix_index is None or pix_index.numel() != buf.n_rays:
        raise ValueError(f"render_train_loss: the buffers were sized for {buf.n_rays} rays, the frame has {frame.ray_count}")
    if frame.n_fine != buf.n_fine or frame.n_coarse != buf.n_coarse or frame.fields != 2:
        raise ValueError(f"render_train_loss: the buffers were sized for {buf.n_coarse} + {buf.n_fine} samples, the frame asks for "
                         f"{frame.n_coarse} + {frame.n_fine}")
    buf.bind(dec)
    if not sig_head.requires_grad:
        sig_head = sig_head.detach().requires_grad_(True)
    if signal_trainer is not None:
        node = sig_head.grad_fn
        if node is None or node is not sig_torso.grad_fn or type(node).__name__ != "_SignalFnBackward":
            signal_trainer = None
    return FusedTrainLossFn.apply(sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, signal_trainer, img_head, img_com)


# ---- the loss ------------------------------------------------------------------------------------------------------------
class MseLossFn(torch.autograd.Function):
    """(rgb_head, rgb_com [n,3]) -> losses [2] = (img2mse(rgb_head, target_head), img2mse(rgb_com, target_com)) with the
    targets gathered from uint8 frames resident on the device (frames.DeviceFrameCache) INSIDE the kernel:
    dfn_mse_loss_u8 = MAIN:791-800 + :902-907 + their autograd in one launch (upstream: two index_selects, two uint8 ->
    float conversions, two divisions, two MSE forwards and their backward kernels)."""

    @staticmethod
    def forward(ctx, rgb_head, rgb_com, img_head, img_com, pix):
        n = rgb_head.shape[0]
        if img_head.dtype != torch.uint8 or img_com.dtype != torch.uint8 or pix.dtype != torch.int32:
            raise TypeError("MseLossFn: uint8 frames and int32 pixel ids expected")
        rh, rc = rgb_head.detach().contiguous(), rgb_com.detach().contiguous()
        losses = torch.empty(3, dtype=torch.float32, device=rh.device)      # (head, com, their sum)
        d_h, d_c = torch.empty_like(rh), torch.empty_like(rc)
        check(lib.dfn_mse_loss_u8(_ptr(rh), _ptr(rc), _ptr(img_head), _ptr(img_com), _ptr(pix), n, _ptr(losses), _ptr(d_h),
                                  _ptr(d_c), _stream()), "dfn_mse_loss_u8")
        ctx.save_for_backward(d_h, d_c)
        # two outputs, not one [2] tensor to be indexed by the caller: every index op would add a SelectBackward (a zeros
        # fill + a copy each, then an add of the two) to the backward - nine 5-us launches between the forward and the dX
        # chain of a 2-ms step
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_h, g_c):
        d_h, d_c = ctx.saved_tensors
        if g_h is not None and g_c is not None and g_h.data_ptr() == g_c.data_ptr():
            # loss = l_head + l_com hands both the same gradient tensor: one multi-tensor launch instead of two
            out = torch._foreach_mul((d_h, d_c), g_h)
            return out[0], out[1], None, None, None
        return (None if g_h is None else d_h * g_h), (None if g_c is None else d_c * g_c), None, None, None


class FusedTrainLossFn(torch.autograd.Function):
    """FusedTrainFn and MseLossFn as ONE autograd node: (sig_head, sig_torso) -> (loss, loss_head, loss_com, rgb_head, rgb_com)
    with loss = loss_com + loss_head formed inside dfn_mse_loss_u8 (MAIN:902-907).  The loss kernel already leaves
    d loss / d rgb; when the step's backward is started with the buffers' own unit gradient (training.backward(loss, buf)) the
    backward goes straight into the compositing backward: no ATen launch between the forward and the dX chain (as two nodes
    autograd ran an add, a ones fill and a multi-tensor multiply there).  rgb_* are returned for inspection only (not
    differentiable through this node: a loss on them - --use_L1 - takes the two-node route)."""

    @staticmethod
    def forward(ctx, sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, defer, img_head, img_com):
        if img_head.dtype != torch.uint8 or img_com.dtype != torch.uint8 or pix_index.dtype != torch.int32:
            raise TypeError("FusedTrainLossFn: uint8 frames and int32 pixel ids expected")
        n, dev = frame.ray_count, buf.flat.device
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        if getattr(buf, "_d_rgb", None) is None or buf._d_rgb[0].shape[0] != n:
            buf._d_rgb = (torch.empty(n, 3, dtype=torch.float32, device=dev), torch.empty(n, 3, dtype=torch.float32, device=dev))
            # the epilogue's partial sums + ticket: zero once, every launch leaves the ticket zero again
            buf._loss_ws = torch.zeros(check(lib.dfn_train_loss_floats(n), "dfn_train_loss_floats"), dtype=torch.float32,
                                       device=dev)
        d_h, d_c = buf._d_rgb
        if _LOSS_IN_FWD:
            # the loss and d loss / d rgb come out of the forward's epilogue: no launch between the forward and the backward
            arg = DfnTrainLoss(img_head.data_ptr(), img_com.data_ptr(), d_h.data_ptr(), d_c.data_ptr(), losses.data_ptr(),
                               buf._loss_ws.data_ptr())
            rgb_h, rgb_c = _fused_forward(ctx, sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, defer, arg)
        else:
            rgb_h, rgb_c = _fused_forward(ctx, sig_head, sig_torso, buf, frame, bg, pix_index, z_shape, z_app, defer)
            check(lib.dfn_mse_loss_u8(_ptr(rgb_h), _ptr(rgb_c), _ptr(img_head), _ptr(img_com), _ptr(pix_index), n,
                                      _ptr(losses), _ptr(d_h), _ptr(d_c), _stream()), "dfn_mse_loss_u8")
        ctx.d = (d_h, d_c)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(rgb_h, rgb_c)
        return losses[2], losses[0], losses[1], rgb_h, rgb_c

    @staticmethod
    def backward(ctx, g_sum, g_h, g_c, _gh, _gc):
        d_h, d_c = ctx.d
        unit = unit_gradient(ctx.buf)
        if not (g_sum is not None and g_h is None and g_c is None and g_sum.data_ptr() == unit.data_ptr()):
            # any other upstream gradient: scale explicitly (torch ops)
            zero = torch.zeros((), dtype=torch.float32, device=d_h.device)
            gs = zero if g_sum is None else g_sum
            d_h = d_h * (gs + (zero if g_h is None else g_h))
            d_c = d_c * (gs + (zero if g_c is None else g_c))
        return _fused_backward(ctx, d_h, d_c) + (None, None)


def unit_gradient(buf):
    """the 0-dim ones tensor of `buf`: hand it to backward() as the loss's gradient (training.backward) and autograd neither
    fills a fresh one nor does FusedTrainLossFn have to multiply by it"""
    u = getattr(buf, "_unit", None)
    if u is None:
        u = buf._unit = torch.ones((), dtype=torch.float32, device=buf.samples.device)
    return u


def backward(loss, buf):
    """loss.backward() with the buffers' unit gradient (see FusedTrainLossFn)."""
    torch.autograd.backward(loss, grad_tensors=unit_gradient(buf))


def mse_losses(rgb_head, rgb_com, img_head, img_com, pix):
    """-> (loss_head, loss_com) 0-dim tensors; img_*: uint8 [H*W,3] device frames, pix: int32 [n]."""
    return MseLossFn.apply(rgb_head, rgb_com, img_head, img_com, pix)


# ---- composite_function / calc_volume_weights under autograd ---------------------------------------------------------
class CompositeFn(torch.autograd.Function):
    """composite_function (MAIN:146-166) as an autograd node: dfn_composite forward, dfn_composite_grad backward - what a
    reference-shaped training loop (MAIN:888-889) differentiates through.  sigma [K,...], feat [K,...,3]."""

    @staticmethod
    def forward(ctx, sigma, feat):
        s, f = sigma.detach().float().contiguous(), feat.detach().float().contiguous()
        ctx.save_for_backward(s, f)
        ctx.set_materialize_grads(False)
        return engine.composite(s, f)

    @staticmethod
    def backward(ctx, d_ss, d_fw):
        s, f = ctx.saved_tensors
        K = s.shape[0]
        d_s, d_f = torch.empty_like(s), torch.empty_like(f)
        d_ss = None if d_ss is None else d_ss.float().contiguous()
        d_fw = None if d_fw is None else d_fw.float().contiguous()
        check(lib.dfn_composite_grad(_ptr(s), _ptr(f), K, s.numel() // K, _ptr(d_ss), _ptr(d_fw), _ptr(d_s), _ptr(d_f),
                                     _stream()), "dfn_composite_grad")
        return d_s, d_f


class VolumeWeightsFn(torch.autograd.Function):
    """calc_volume_weights (MAIN:169-179) as an autograd node: dfn_volume_weights forward, dfn_volume_weights_grad backward
    (gradient of sigma; the depths and ray vectors are constants of the training loop, MAIN:838-841)."""

    @staticmethod
    def forward(ctx, z_vals, ray_vector, sigma, last_dist):
        S = z_vals.shape[-1]
        z = z_vals.detach().float().contiguous().reshape(-1, S)
        r = ray_vector.detach().float().contiguous().reshape(-1, 3)
        sg = sigma.detach().float().contiguous().reshape(-1, S)
        ctx.save_for_backward(z, r, sg)
        ctx.last_dist, ctx.shape = float(last_dist), sigma.shape
        return engine.volume_weights(z, r, sg, last_dist).reshape(sigma.shape)

    @staticmethod
    def backward(ctx, d_w):
        z, r, sg = ctx.saved_tensors
        d_w = d_w.float().contiguous()
        d_s = torch.empty_like(sg)
        check(lib.dfn_volume_weights_grad(_ptr(z), _ptr(r), _ptr(sg), sg.shape[0], sg.shape[1], ctx.last_dist, _ptr(d_w),
                                          _ptr(d_s), _stream()), "dfn_volume_weights_grad")
        return None, None, d_s.reshape(ctx.shape), None


# ---- Decoder.forward on explicit points under autograd -------------------------------------------------------------
class _PointBuffers:
    """Device buffers of one decoder-on-points training call (one field), sized for NP = ceil32(n) points."""

    def __init__(self, tier, field, n, device):
        self.tier, self.field, self.n = tier, field, n
        # whole tiles; the 16-bit tier's weight-gradient GEMMs contract PAIRS of tiles (block-scaled MFMA, K = 64 points)
        self.NP = NP = (n + 63) // 64 * 64 if tier == 1 else (n + 31) // 32 * 32
        rows = lambda w: check(lib.dfn_train_rows(field, w), "dfn_train_rows")
        if tier == 1:       # MX-fp8 (TrainBuffers); zeros: a padding tile the forward never writes must read as 0.0 x 2^-127
            # (selector 8: the decoder-on-points recorder keeps e4m3 activations - dfn_weight_bias_grad_fmt(.., DFN_ACT_E4M3, ..))
            self.act = torch.zeros(NP // 32, rows(8), dtype=torch.uint8, device=device)
            self.dy = torch.empty(NP // 32, rows(7), dtype=torch.uint8, device=device)
        else:
            self.act = torch.empty(rows(0), NP, dtype=torch.float32, device=device)
            self.dy = torch.empty(rows(1), NP, dtype=torch.float32, device=device)
        self.masks = torch.zeros(NP // 32, rows(2), 64, dtype=torch.int32, device=device)
        self.ws = torch.empty(rows(3), dtype=torch.float32, device=device)
        self.samples = torch.zeros(NP, 8, dtype=torch.float32, device=device)
        self.packed = torch.empty(check(lib.dfn_packed_bytes(tier, field), "packed"), dtype=torch.uint8, device=device)
        self.packed_T = torch.empty(check(lib.dfn_packed_bwd_bytes(tier, field), "packed_T"), dtype=torch.uint8,
                                    device=device)
        self.nb = check(lib.dfn_bias_floats(tier, field), "bias")
        self.bias = torch.empty(self.nb, dtype=torch.float32, device=device)


class DecoderTrainFn(torch.autograd.Function):
    """signal [96] / [42] -> feat [n,3], sigma [n] of ONE field at explicit points: dfn_fold_bias + dfn_decoder_train_fwd
    (the fused decoder with its recorder on); backward = dfn_mlp_bwd, dfn_weight_bias_grad, dfn_fold_bias_bwd: returns
    d(signal) to autograd and deposits the decoder gradients into the parameters' .grad (like FusedTrainFn).
    The points and directions get no gradient (upstream they come from get_rays / linspace: constants)."""

    @staticmethod
    def forward(ctx, signal, net, pb, pts, dirs, zs, za):
        t, f, st = pb.tier, pb.field, _stream()
        flat, dev = net.flat, net.flat.device
        sg = signal.detach().reshape(-1).float().contiguous() if signal.numel() else None      # (listener: no signal)
        check(lib.dfn_fold_bias(t, f, _ptr(flat), _ptr(sg), _ptr(zs), _ptr(za), _ptr(pb.bias), st), "dfn_fold_bias")
        check(lib.dfn_pack_weights(t, f, _ptr(flat), _ptr(pb.packed), st), "dfn_pack_weights")
        check(lib.dfn_pack_weights_bwd(t, f, _ptr(flat), _ptr(pb.packed_T), st), "dfn_pack_weights_bwd")
        feat = torch.empty(pb.n, 3, dtype=torch.float32, device=dev)
        sigma = torch.empty(pb.n, dtype=torch.float32, device=dev)
        check(lib.dfn_decoder_train_fwd(t, f, _ptr(pb.packed), _ptr(pb.bias), _ptr(pts), _ptr(dirs), pb.n, _ptr(feat),
                                        _ptr(sigma), _ptr(pb.samples), _ptr(pb.act), _ptr(pb.masks), st),
              "dfn_decoder_train_fwd")
        ctx.net, ctx.pb, ctx.keep, ctx.sig_shape = net, pb, (sg, zs, za, pts, dirs), signal.shape
        return feat, sigma

    @staticmethod
    def backward(ctx, d_feat, d_sigma):
        net, pb, (sg, zs, za, _, _), st = ctx.net, ctx.pb, ctx.keep, _stream()
        t, f, dev = pb.tier, pb.field, net.flat.device
        o = 4 if f == FIELD_TORSO else 0          # (the listener is the head's program: the head's output slots)
        ds = torch.zeros(pb.NP, 8, dtype=torch.float32, device=dev)
        ds[:pb.n, o] = d_sigma.reshape(-1)
        ds[:pb.n, o + 1:o + 4] = d_feat.reshape(-1, 3)
        check(lib.dfn_mlp_bwd(t, f, _ptr(pb.packed_T), _ptr(pb.samples), _ptr(ds), _ptr(pb.masks), pb.NP, _ptr(pb.dy),
                              st), "dfn_mlp_bwd")
        g_flat = _grad_buffer(net, "_g_flat", net.flat, net.params)
        g_bias = torch.empty(pb.nb, dtype=torch.float32, device=dev)
        d_sig = torch.zeros(0 if sg is None else sg.numel(), dtype=torch.float32, device=dev)
        check(lib.dfn_weight_bias_grad_fmt(t, f, 0, _ptr(pb.dy), _ptr(pb.act), pb.NP, _ptr(pb.ws), _ptr(g_flat), _ptr(g_bias),
                                           st), "dfn_weight_bias_grad_fmt")       # (0 = DFN_ACT_E4M3)
        check(lib.dfn_fold_bias_bwd(t, f, _ptr(net.flat), _ptr(sg), _ptr(zs), _ptr(za), _ptr(g_bias), _ptr(g_flat),
                                    _ptr(d_sig) if sg is not None else None, st), "dfn_fold_bias_bwd")
        net.deposit(g_flat, touched=_decoder_touched(net, (f,)))
        return d_sig.reshape(ctx.sig_shape), None, None, None, None, None, None


def decoder_train(dec, field, p_in, ray_d, z_shape, z_app, signal, tier="f32"):
    """Decoder.forward(p_in [B,N,3], ray_d [B,N,3], ...) under autograd, in HIP: -> feat [B,N,3], sigma [B,N].
    field: 0 head, 1 torso, 2 listener (`signal` None: the head's program on fc_in_listener / fc_p_skips_listener,
    decoder.py:306-307, 322-323).  Gradients flow to the decoder's parameters (deposited into .grad) and to `signal`."""
    if field not in (FIELD_HEAD, FIELD_TORSO, FIELD_LISTENER):
        raise ValueError(f"decoder_train: field {field}")
    if not dec.hip_supported():
        raise NotImplementedError("the HIP path supports the scripts/test_obama.sh decoder configuration only")
    t = TIERS["bf16" if tier in ("f16", 2) else tier]
    net = _FlatNet.of(dec)
    net.refresh()
    dev = net.flat.device
    pts = p_in.detach().reshape(-1, 3).to(dev, torch.float32).contiguous()
    dirs = ray_d.detach().reshape(-1, 3).to(dev, torch.float32).contiguous()
    n = pts.shape[0]
    # a fresh set of buffers per call: several forwards may be alive before their backwards run (head, then torso)
    pb = _PointBuffers(t, field, n, dev)
    zs = _pad_z(z_shape.detach().reshape(-1)[:dec.z_dim].to(dev), 1).reshape(-1)
    za = _pad_z(z_app.detach().reshape(-1)[:dec.z_dim].to(dev), 1).reshape(-1)
    if field == FIELD_LISTENER:
        # no conditioning signal: an anchor keeps the node in the graph (its "gradient" is empty)
        signal = torch.zeros(0, dtype=torch.float32, device=dev, requires_grad=True)
    elif not signal.requires_grad:
        signal = signal.detach().requires_grad_(True)
    feat, sigma = DecoderTrainFn.apply(signal, net, pb, pts, dirs, zs, za)
    return feat.reshape(p_in.shape[0], -1, 3), sigma.reshape(p_in.shape[0], -1)
