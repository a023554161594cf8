"""HipAdam: torch.optim.Adam (run_nerf_com_trainExpLater.py:522-547, betas (0.9, 0.999)) whose step() is ONE
dfn_adam_multi launch per parameter group.

torch's fused multi-tensor kernel gives a block a 64 K-element chunk: the decoder's 68 tensors become ~80 blocks and
its step takes 104 us on an MI355X, the five optimizers of a training step 250 us; dfn_adam_multi cuts tensors into
2048-element chunks and is bound by the bytes it moves.  Same update rule and the same state / state_dict layout as
torch.optim.Adam(fused=True) (state[p] = {"step": 0-dim f32 device tensor, "exp_avg", "exp_avg_sq"}), so checkpoints
move freely between the two; parameters of one group at different step counts (a resumed reference checkpoint has no
state for the parameters its autograd never reached) are stepped in one launch per distinct count; anything this class
does not cover (weight decay, amsgrad, maximize, non-f32 or non-contiguous tensors) goes through torch's own step()."""
import ctypes as C
import math

import numpy as np
import torch

from ._lib import lib, check

ADAM_CHUNK = 2048          # DFN_ADAM_CHUNK in include/dfanerf.h


def _bump_versions(tensors):
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is None:                                  # older torch: an in-place no-op per tensor list
        torch._foreach_add_(list(tensors), 0.0)
        return
    try:
        inc(tensors)                                 # iterable accepted by recent versions
    except TypeError:
        for t in tensors:
            inc(t)


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, fused=True)
        # group index -> {"pkey", "gkey", "buckets": [{"ps", "t", "items", "chunks", "n_chunks", "host"}]}: one bucket
        # (= one launch) per distinct step count of the group's parameters
        self._cache = {}

    # ---- state bookkeeping: torch keeps a step counter per parameter, a launch needs one per bucket ---------------
    def _sync_steps(self):
        for c in self._cache.values():
            for b in c["buckets"]:
                steps = [self.state[p]["step"] for p in b["ps"]]
                torch._foreach_zero_(steps)
                torch._foreach_add_(steps, float(b["t"]))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache.clear()

    def _torch_step(self, closure):
        self._sync_steps()
        self._cache.clear()
        return super().step(closure)

    def _build(self, gi, ps, pkey, gkey):
        old = self._cache.get(gi)
        if old is not None and old["pkey"] == pkey:
            # same parameters, new gradient buffers: the counts carry on
            counts = {id(p): b["t"] for b in old["buckets"] for p in b["ps"]}
        else:
            if old is not None:
                self._sync_steps()
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # one device-to-host copy when the table is (re)built - e.g. after load_state_dict: a resumed reference
            # checkpoint holds no state for parameters whose gradients were None upstream, so counts can differ
            steps = torch.stack([self.state[p]["step"].reshape(()).to(ps[0].device) for p in ps]).cpu()
            counts = {id(p): int(s) for p, s in zip(ps, steps.tolist())}
        dev = ps[0].device
        buckets = []
        for t in sorted(set(counts[id(p)] for p in ps)):
            bps = [p for p in ps if counts[id(p)] == t]
            items = np.zeros((len(bps), 5), dtype=np.int64)
            chunks = []
            for i, p in enumerate(bps):
                st = self.state[p]
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and v.dtype == torch.float32):
                    return None
                items[i] = (p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                chunks.extend((i, k) for k in range((p.numel() + ADAM_CHUNK - 1) // ADAM_CHUNK))
            # pinned + asynchronous: a pageable upload would block the host until the stream has drained
            h_items = torch.from_numpy(items).pin_memory()
            h_chunks = torch.from_numpy(np.asarray(chunks, dtype=np.int32).reshape(-1, 2)).pin_memory()
            buckets.append({"ps": bps, "t": t, "n_chunks": len(chunks), "host": (h_items, h_chunks),
                            "items": h_items.to(dev, non_blocking=True), "chunks": h_chunks.to(dev, non_blocking=True)})
        c = {"pkey": pkey, "gkey": gkey, "buckets": buckets, "ps": [], "gs": []}
        self._cache[gi] = c
        # built once: the table uploads above ran on the current stream, the launches may go to self.dfn_stream
        torch.cuda.current_stream(dev).synchronize()
        return c

    def step(self, closure=None):
        """One launch per group and distinct step count.  `self.dfn_stream` (training.SignalTrainer.adopt_optimizers): run
        the step on that stream instead of the current one - the conditioning networks' gradients are produced there early in
        the backward, so their update (and the next step's encoder forward behind it) need not queue behind the decoder's
        weight-gradient GEMMs.  The current stream is ordered behind the update (unless `dfn_join_later`: the stream's owner
        does that where the parameters are next read); with more than one rank the side stream first waits for the current
        one (the gradient all-reduce runs there) unless parallel.StepReducer exchanged these gradients on the side stream."""
        s = getattr(self, "dfn_stream", None)
        if s is None:
            return self._step(closure, None)
        cur = torch.cuda.current_stream(s.device)
        from .parallel import multi_rank_schedule
        # (dfn_reduced_on_stream, set per step by parallel.StepReducer: these gradients were exchanged on `s` itself)
        if multi_rank_schedule() and not self.__dict__.pop("dfn_reduced_on_stream", False):
            s.wait_stream(cur)
        out = self._step(closure, s)
        if not getattr(self, "dfn_join_later", False):     # the owner of the stream orders its readers itself
            cur.wait_stream(s)                             # (SignalTrainer: encode() and join())
        return out

    # torch wraps Optimizer.step of every subclass with its profiler / hook machinery (~40 us per call, four optimizers per
    # training step of 1.8 ms): training loops that register no optimizer hooks call this name (run_nerf.optimizer_steps)
    step_unhooked = step

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def _fast_plan(self, gi, group):
        """The cached launch table of group gi if nothing it was built from has moved: same parameters with gradients, the
        same gradient TENSORS (training._FlatNet.deposit hands out the same view objects while the flat gradient buffer
        stays) and the same parameter storage.  None: take the checking path."""
        c = self._cache.get(gi)
        if c is None:
            return None
        i, ps, gs, pk = 0, c["ps"], c["gs"], c["pkey"]
        n = len(ps)
        for p in group["params"]:
            g = p.grad
            if g is None:
                continue
            if i >= n or p is not ps[i] or g is not gs[i] or p.data_ptr() != pk[i]:
                return None
            i += 1
        return c if i == n else None

    @torch.no_grad()
    def _step(self, closure=None, stream=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plans = []
        for gi, group in enumerate(self.param_groups):
            if group.get("weight_decay", 0) or group.get("amsgrad") or group.get("maximize") or \
                    not isinstance(group["lr"], (int, float)):
                return self._fallback(stream) or loss
            c = self._fast_plan(gi, group)
            if c is None:
                ps = [p for p in group["params"] if p.grad is not None]
                if not ps:
                    continue
                for p in ps:
                    g = p.grad
                    if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous()
                            and g.is_contiguous() and g.device == p.device and not g.is_sparse):
                        return self._fallback(stream) or loss
                pkey = tuple(p.data_ptr() for p in ps)
                gkey = tuple(p.grad.data_ptr() for p in ps)
                c = self._cache.get(gi)
                if c is None or c["pkey"] != pkey or c["gkey"] != gkey:
                    c = self._build(gi, ps, pkey, gkey)
                    if c is None:
                        return self._fallback(stream) or loss
                c["ps"], c["gs"] = ps, [p.grad for p in ps]
            elif not c["ps"]:
                continue
            plans.append((group, c))
        for group, c in plans:
            b1, b2 = group["betas"]
            for b in c["buckets"]:
                b["t"] += 1
                t = b["t"]
                cs = stream if stream is not None else torch.cuda.current_stream(b["ps"][0].device)
                check(lib.dfn_adam_multi(C.c_void_p(b["items"].data_ptr()), C.c_void_p(b["chunks"].data_ptr()),
                                         b["n_chunks"], float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                         float(1.0 - b1 ** t), float(math.sqrt(1.0 - b2 ** t)), C.c_void_p(cs.cuda_stream)),
                      "dfn_adam_multi")
                # the kernel wrote the parameters behind torch's back: bump their version counters like an in-place op
                # would, or everything keyed on them (Decoder.packed()'s repack-on-change, autograd's saved-tensor
                # checks) goes stale
                _bump_versions(b["ps"])
        return loss

    def _fallback(self, stream):
        if stream is None:
            return self._torch_step(None)
        with torch.cuda.stream(stream):
            return self._torch_step(None)
