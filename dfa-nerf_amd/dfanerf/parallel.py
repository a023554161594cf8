"""Multi-GPU partitioning of the render path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Rays are independent given (pose, per-frame signals, weights), so there is no collective inside the
renderer:
  * inference: rank r renders rays [r*ceil(R/P), min(R, (r+1)*ceil(R/P))) of every frame; ONE all_gather of the
    padded RGB shards per frame (<= 304 KB per rank for a 450x450 frame);
  * training: each rank draws its own frame and rays; ONE all_reduce(sum) of a single flat fp32 gradient
    bucket holding all five networks (1,138,656 floats = 4.55 MB), scaled by 1/P, before the gated optimizer
    steps.  Adam states are replicated; z_shape / z_app are constants."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment; no-op for a single process."""
    world, rank, local = env_world()
    # DFN_ONE_GPU=1 (developer / test switch): every rank on GPU 0 over gloo - a functional run of the multi-rank paths on a
    # one-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("DFN_ONE_GPU") and torch.cuda.is_available():
        local, backend = 0, backend or "gloo"
        torch.cuda.set_device(0)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return world, rank, local


def shard_range(n_rays, world, rank):
    """(begin, count, per) for rank: contiguous blocks of per = ceil(n_rays / world) rays."""
    per = (n_rays + world - 1) // world
    begin = min(rank * per, n_rays)
    return begin, max(0, min(n_rays, begin + per) - begin), per


def gather_rays(shard, n_rays, group=None):
    """shard [per, C] (rows beyond this rank's count are padding) -> [n_rays, C] on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return shard[:n_rays]
    out = torch.empty(world * shard.shape[0], *shard.shape[1:], dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(out, shard.contiguous(), group=group)
    return out[:n_rays]


def broadcast_replicas(nets, opts, extra=(), src=0, group=None):
    """Make every rank's replica identical to rank `src`'s: all parameters and buffers of `nets` (dict of modules), the
    per-parameter Adam state of `opts` (dict of optimizers; state entries that exist on `src` are created elsewhere) and
    the tensors in `extra` (returned, broadcast).  Called once after create_nerf / load_checkpoint: FlatGradBucket only
    averages gradients, so replicas that start different stay different."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(extra)
    # gloo (the functional modes: every rank on one GPU over gloo, the CPU tests) is handed HOST tensors only.  Given device
    # tensors, c10d's gloo backend stages them through pinned buffers with copies on streams of its own - and with eight ranks on
    # ONE device and eight hardware queues per process that start-up phase died in 2 of 3 runs with a GPU memory fault inside one
    # of those copies (an ATen elementwise copy of a 256 x 256 weight), caching allocator on or off; staged through the host by
    # us it passed 3 of 3 on the same box (round 6, profiles/r06g_world8_bcast.txt; plain PyTorch processes with the same
    # process / queue counts and no gloo never fault: tools/queue_oversub_repro.py).  Production ranks use RCCL, which takes
    # device tensors.  DFN_BCAST_HOST=0: the old route (developer switch).
    via_host = os.environ.get("DFN_BCAST_HOST", "1") == "1" and dist.get_backend(group) == "gloo"

    def bcast(t):
        if via_host and t.is_cuda:
            h = t.detach().cpu()
            dist.broadcast(h, src=src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=group)
    with torch.no_grad():
        for k in sorted(nets):
            for t in list(nets[k].parameters()) + list(nets[k].buffers()):
                bcast(t.data)
        for k in sorted(opts):
            # through the HOST: a pickled device tensor unpickles onto the SOURCE rank's device on every receiver (each rank
            # would open a context and allocate on GPU `src`); load_state_dict casts to the parameters' device
            sd = [_to_device(opts[k].state_dict(), "cpu")] if dist.get_rank(group) == src else [None]
            dist.broadcast_object_list(sd, src=src, group=group)
            if dist.get_rank(group) != src:
                dev = next(iter(nets[k].parameters())).device if k in nets else None
                opts[k].load_state_dict(_to_device(sd[0], dev))
        out = []
        for t in extra:
            t = t.contiguous()
            bcast(t)
            out.append(t)
    return out


def _to_device(obj, dev):
    if isinstance(obj, torch.Tensor):
        return obj.to(dev) if dev is not None else obj
    if isinstance(obj, dict):
        return {k: _to_device(v, dev) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(v, dev) for v in obj)
    return obj


def multi_rank_schedule():
    """True when the training step runs next to a collective: more than one rank - or DFN_FORCE_MULTIRANK=1 with an
    initialised process group of ONE rank (developer switch: the multi-rank stream schedule and the bucket all_reduce over
    the real RCCL backend on a box with one GPU; RCCL refuses two ranks on one device)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("DFN_FORCE_MULTIRANK") == "1"


class FlatGradBucket:
    """One flat fp32 buffer for the gradients of several modules -> a single all_reduce per step.

    The HIP training path writes ONE flat gradient buffer per network (training._FlatNet, `.grad` = views of it).  After the
    first step those buffers ARE slices of the bucket (_adopt): from then on a step's all_reduce runs in place on what the
    backward kernels wrote - no gather copy, no scaling pass (ReduceOp.AVG on RCCL), no re-pointing of sixty `.grad`s;
    measured on one MI355X through RCCL (one rank, bench.py DFN_BENCH_RCCL_WORLD1): 95 us of a 1.4-ms step.  Gradients that
    live anywhere else (torch autograd's own tensors, accumulated gradients) take the general path: two multi-tensor copies
    around the collective."""

    def __init__(self, modules):
        self.modules = list(modules)
        self.params = [p for m in self.modules for p in m.parameters()]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._ptrs = [v.data_ptr() for v in self.views]
        self._spans, off = [], 0                     # (first element, one past the last, parameters) per module
        for m in self.modules:
            ps = list(m.parameters())
            n = sum(p.numel() for p in ps)
            self._spans.append((off, off + n, ps))
            off += n

    def _adopt(self):
        """The networks' flat gradient buffers become slices of the bucket (modules bound by training._FlatNet whose flat
        layout is exactly their parameters, in order: all five networks of the training step)."""
        for m, (a, b, ps) in zip(self.modules, self._spans):
            fn = m.__dict__.get("_dfn_flat")
            if fn is not None and fn.flat.numel() == b - a and len(fn.params) == len(ps) and \
                    all(x is y for x, y in zip(fn.params, ps)):
                fn._g_flat = self.flat[a:b]

    def _reduce(self, group):
        world = dist.get_world_size(group)
        if world > 1 and dist.get_backend(group) == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if world > 1:
                self.flat.mul_(1.0 / world)

    def all_reduce_(self, group=None):
        """Average the gradients over the ranks in place (missing grads count as zero)."""
        if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not multi_rank_schedule()):
            return
        grads = [p.grad for p in self.params]
        if any(g is not None for g in grads) and \
                all(g is None or g.data_ptr() == ptr for g, ptr in zip(grads, self._ptrs)):
            # in place.  A module NONE of whose parameters has a gradient ran no backward this step: its slice still holds
            # an older step's values (parameters without a gradient inside a module that ran are zero: the backward zeroed
            # the module's whole buffer); `.grad = None` stays None.
            for a, b, ps in self._spans:
                if all(p.grad is None for p in ps):
                    self.flat[a:b].zero_()
            self._reduce(group)
            return
        # general path.  Some gradients may ALREADY live in the bucket (an adopted network next to torch-autograd ones):
        # those are left where they are - zeroing the whole bucket here wiped them (the decoder stopped
        # training from step 2 whenever the conditioning networks' gradients came from torch autograd).  Only the views of
        # parameters WITHOUT a gradient are zeroed (they count as zero in the average), only foreign gradients are copied.
        zero = [v for v, g in zip(self.views, grads) if g is None]
        copy = [(v, g) for v, g, ptr in zip(self.views, grads, self._ptrs) if g is not None and g.data_ptr() != ptr]
        if zero:
            torch._foreach_zero_(zero)
        if copy:
            torch._foreach_copy_([v for v, _ in copy], [g for _, g in copy])
        self._reduce(group)
        # `.grad = None` stays None, as on the in-place path and in single-rank training (Adam skips the parameter; which
        # parameters carry a gradient is decided by the step gating, identical on every rank)
        for p, v, g in zip(self.params, self.views, grads):
            if g is not None:
                p.grad = v
        self._adopt()


class StepReducer:
    """The training step's gradient exchange in THREE collectives, each on the stream its gradients are produced on
    (training.SignalTrainer): the audio-side networks' 180,785 gradients as soon as their backward is done (under the torso
    field's dX chain), PoseAttNet's 2,629 after the torso's, the decoder's 955,242 behind its weight-gradient GEMMs on the
    current stream.  With ONE bucket the conditioning networks' Adam waited for the decoder's last GEMM, and the next step's
    audio encoder (a 77-us single-workgroup chain the decoder forward needs) was exposed at the start of every step instead
    of running underneath the previous step's GEMMs.  Every rank issues the collectives in the same order (audio, pose,
    decoder) - also the order they become ready in -; the backend runs them on its one stream in that order."""

    AUDIO = ("AudNet", "ExpNet", "AudAttNet")
    POSE = ("PoseAttNet",)

    def __init__(self, nets, opts=None, signal_trainer=None):
        """nets: dict name -> module as run_nerf.create_nerf returns it ("decoder" + the conditioning networks)."""
        self.trainer = signal_trainer
        self.opts = opts or {}
        early = signal_trainer is not None and getattr(signal_trainer, "_pipelined", False) and \
            signal_trainer.audio_stream() is not None and all(k in nets for k in self.AUDIO + self.POSE) and \
            set(nets) == set(self.AUDIO + self.POSE + ("decoder",))
        if early:
            self.side = [(self.AUDIO, FlatGradBucket([nets[k] for k in self.AUDIO]), signal_trainer.audio_stream),
                         (self.POSE, FlatGradBucket([nets[k] for k in self.POSE]), signal_trainer.pose_stream)]
            self.dec = FlatGradBucket([nets["decoder"]])
        else:                     # no side streams to run the early collectives on: one bucket, one collective
            self.side = []
            self.dec = FlatGradBucket(list(nets.values()))

    def all_reduce_(self, group=None):
        for names, bucket, stream in self.side:
            with torch.cuda.stream(stream()):
                bucket.all_reduce_(group)
            for k in names:                            # their step() need not wait for the current stream's collective
                o = self.opts.get(k)
                if o is not None:
                    o.dfn_reduced_on_stream = True
        self.dec.all_reduce_(group)
