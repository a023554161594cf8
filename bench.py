#!/usr/bin/env python3
"""bench.py - throughput of the fused DFA-NeRF renderer on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one 450x450 audio-driven frame through the hot path: per-frame conditioning signals (HIP encoders)
-> bias fold -> fused render (rays -> 64 coarse -> sample_pdf -> 64+128 merged samples -> decoder MLP ->
compositing) -> (N > 1) RCCL all-gather of the RGB shards.  Weights, background and the audio/expression/
pose features are resident in HBM before the timed region.  For N > 1 the rays of every frame are sharded
across the ranks (strong scaling: total work per step is fixed), the partition SURVEY.md 8(e) names.

The headline tier is f16 (v_mfma_f32_32x32x16_f16): the 16-bit tier whose rendered RGB holds the north star's accuracy
clause on the full frame (tests/test_gpu_parity.py::test_full_frame_psnr_16bit_tiers_vs_f32_tier, >= 49.4 dB against
the exact f32 tier).  --tier bf16 is the 16-bit training tier (faster by ~5 %, 47 dB), --tier f32 the exact tier.

Prints ONE JSON line on rank 0 (fields: see the driver contract), including
  roofline     - MFMA roofline of the dominant kernel (render_kernel), from algorithmic FLOPs and HIP-event
                 timings of the launches inside the timed region; `clock_ghz` = the shader clock under load read inside
                 the kernel (s_memtime / s_memrealtime): the launch is power-bound, the 2.5 PF peak assumes 2.4 GHz;
  sustained    - the same step repeated for >= --sustain-seconds (default 10 s) after the K timed steps: the K steps
                 the driver asks for last well under a second, the chip's DVFS settles later;
  cpu_baseline - the oracle (oracle/dfa_oracle.py, a port of the reference's CPU path) timed on a bounded
                 sample of the same workload on this host's cores, at the best of a sweep over thread counts
                 (rank 0, N == 1 only);
  other_workloads - (N == 1) short runs of the other BASELINE configs in the same process: c3 (two-field render), c1
                 (coarse), c4 (training step), so that they are driver-timed numbers too.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))

import numpy as np
import torch
import torch.distributed as dist

# algorithmic FLOPs per sample point: 2 * MACs of the layers the reference evaluates
FLOP_PT_HEAD, FLOP_PT_TORSO = 1222656.0, 1285120.0
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}       # dense MFMA peaks, MI355X_MICROARCH.md

WORKLOADS = {
    # name: (n_fine, fields, description)
    "c1": (0, 1, "Obama head NeRF, 450x450, coarse-only 64 samples/ray (configs[0] geometry, on GPU)"),
    "c2": (128, 1, "Obama head NeRF inference, 450x450, 64+128 hierarchical (configs[1])"),
    "c3": (128, 2, "Obama head+torso two-NeRF composite render, 450x450, 64+128 (configs[2])"),
    # training step (configs[3]): reference semantics = coarse 64 samples, both fields, fwd+bwd, 5 gated Adams;
    # data parallel: every rank draws its own frame + 2048 rays (weak scaling), one flat-bucket all_reduce
    "c4": (0, 2, "Obama training step (fwd+bwd, Adam), N_rand=2048 per GPU, data-parallel RCCL grad all-reduce (configs[3])"),
    # the strong-scaling variant SURVEY.md 8(e) asks to report as well: the reference's 2048 rays split over the ranks
    "c4s": (0, 2, "Obama training step (fwd+bwd, Adam), N_rand=2048 GLOBAL (2048/N per GPU), RCCL grad all-reduce"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)        # 300 frames = ~10 s: a sustained number by default
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--tier", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (after the sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=10.0,
                    help="length of the sustained run after the K timed steps (0 = off; skipped when the K steps "
                         "themselves already lasted that long)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other workloads (N == 1)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(args, sc, st, zs, za, n_fine, fields):
    """Oracle (port of the reference CPU path) on chunks of frame 0.  Thread sweep first (256-ray pieces, ~1-2 s each):
    torch's intra-op parallelism on 2048-ray chunks does not scale to every core of a 128-core host, and an
    oversubscribed baseline would flatter the GPU.  Then whole 2048-ray chunks at the best count until ~cpu-seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dfa_oracle as O
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    P = O.params_to_torch(st["decoder"])
    nets = {k: O.params_to_torch(v) for k, v in st.items() if k != "decoder"}
    auds, exps, poses = [torch.from_numpy(sc[k]) for k in ("aud", "exp", "poses")]
    H, W = sc["H"], sc["W"]
    bg = torch.from_numpy(sc["bg"]).float() / 255.0

    def run(begin, n, chunk):
        t0 = time.perf_counter()
        O.render_frame(P, H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], bg, sc["near"],
                       sc["far"], torch.from_numpy(zs), torch.from_numpy(za), sig, sigt, 64, n_fine, fields, chunk,
                       ray_begin=begin, ray_count=n)
        return time.perf_counter() - t0
    with torch.no_grad():
        sig = O.encode_signal(nets, auds, exps, 0, 300000, 300000, 4, auds.shape[0])
        sigt = O.encode_signal_torso(nets, poses, 0, 300000, 300000, 8, poses.shape[0])
        run(0, 256, 2048)                                        # warm-up (not timed)
        sweep = {}
        for nt in sorted({n for n in (8, 16, 32, 64, 128, cores) if n <= cores}):
            torch.set_num_threads(nt)
            run(0, 128, 2048)
            sweep[nt] = 1024 / run(101250, 1024, 2048)           # rays/s on a 1024-ray piece from the middle of the frame
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        chunk, done, t_used = 2048, 0, 0.0
        while t_used < args.cpu_seconds and done < H * W:
            n = min(chunk, H * W - done)
            t_used += run(done, n, chunk)
            done += n
    return {"value": done / t_used, "unit": "rays/s", "cores": int(best), "kind": "port",
            "host_physical_cores": int(cores),
            "thread_sweep_rays_per_s": {str(k): round(v, 1) for k, v in sweep.items()},
            "sample": f"{done} rays ({done // chunk} chunks of 2048) of frame 0, same workload, fp32, "
                      f"torch {torch.__version__} CPU, {best} threads (best of the sweep), {t_used:.1f} s"}


# ---------------------------------------------------------------------------------------------------------------
def bench_training(args, workload, steps, warmup, world, rank, dev, sustain_s=0.0):
    """configs[3]: one optimisation step per `step`: signals -> fold -> fused HIP forward (recorder on) -> MSE
    losses -> HIP backward (compositing, dX chain, weight-gradient GEMMs) -> flat-bucket all_reduce -> gated Adams.
    Ground-truth frames and background are resident uint8 device tensors; pixels are drawn and targets gathered on
    the device (frames.PixelSampler, dfn_mse_loss_u8): no host image read, no host-to-device copy per step."""
    from dfanerf import nets, parallel, run_nerf, synth, training
    from dfanerf.decoder import Decoder
    tier = "bf16" if args.tier == "f16" else args.tier           # the 16-bit training tier (f16 is inference only)
    desc = WORKLOADS[workload][2]
    strong = workload == "c4s"
    N_RAND = 2048 // world if strong else 2048
    assert N_RAND % 8 == 0
    sc = synth.bench_scene(0, n_frames=8)
    st = synth.synth_all_states(0)
    H, W = sc["H"], sc["W"]
    t = lambda x: torch.from_numpy(np.asarray(x))
    mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
            "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in st[k].items()})
        m.to(dev)
    a = run_nerf.config_parser().parse_args(
        f"--expname b --concate_bg --N_rand={N_RAND} --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
        "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000".split())
    ds = [{"auds": t(sc["aud"]).to(dev), "exp": t(sc["exp"]).to(dev), "poses": t(sc["poses"]).to(dev),
           "bc_img": (t(sc["bg"]).float() / 255.0).to(dev), "hwfcxy": [H, W, sc["focal"], sc["cx"], sc["cy"]],
           "near": sc["near"], "far": sc["far"]}]
    zs, za = [t(v).to(dev) for v in synth.synth_latents(0)]
    embed_fn, _ = nets.get_embedder(3, 0)
    opts = {k: run_nerf.make_adam(m.parameters(), 5e-4) for k, m in mods.items()}
    buf = training.TrainBuffers(tier, N_RAND, dev)
    buf.signal_trainer = training.SignalTrainer(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                                ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    buf.signal_trainer.adopt_optimizers(opts)
    bucket = parallel.FlatGradBucket(list(mods.values())) if world > 1 else None
    rng = np.random.RandomState(100 + rank)
    rng_frame = np.random.RandomState(100) if strong else rng      # strong: ONE frame per step on all ranks (MAIN:779)
    # the training input stage as train() runs it (dfanerf/frames.py): uint8 ground-truth frames resident on the device,
    # pixels drawn on the device, targets gathered inside the loss kernel
    from dfanerf import frames
    g8 = torch.Generator(device=dev).manual_seed(7)
    gt = [(torch.randint(0, 256, (H * W, 3), dtype=torch.uint8, device=dev, generator=g8),
           torch.randint(0, 256, (H * W, 3), dtype=torch.uint8, device=dev, generator=g8)) for _ in range(8)]
    # (DFN_BENCH_FIFTH_STREAM: developer switch - the pixel draw on a stream of its own, i.e. five streams on four hardware queues)
    sampler = frames.PixelSampler(H, W, N_RAND, 0, dev, seed=100 + rank, pipeline=True,
                                  stream=None if os.environ.get("DFN_BENCH_FIFTH_STREAM") else buf.signal_trainer.pose_stream())
    gstep = 300000                                   # all five optimizers' gates exercised except ExpNet

    host_t = [0.0] * 5 if os.environ.get("DFN_BENCH_HOST_TIMING") else None      # developer switch: host time by section

    def step():
        c = time.perf_counter
        t0 = c()
        img_i = int(rng_frame.randint(0, 8))
        pix = sampler.draw()
        loss, *_ = run_nerf.train_step_loss_hip(mods, ds, 0, img_i, pix, gt[img_i][0], gt[img_i][1], zs, za, gstep, a,
                                                8, embed_fn, ds[0]["poses"][0], buf)
        t1 = c()
        for o in opts.values():
            o.zero_grad()
        t2 = c()
        loss.backward()
        t3 = c()
        if bucket is not None:
            bucket.all_reduce_()
        run_nerf.optimizer_steps(opts, gstep, a)
        t4 = c()
        run_nerf.update_lrate(opts, gstep, a)
        if host_t is not None:
            for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, c() - t4)):
                host_t[k] += d
        return loss

    def host_report(n):
        if host_t is not None and rank == 0:
            names = ("draw + forward", "zero_grad", "backward", "optimizers", "lr")
            print("host ms/step: " + ", ".join(f"{nm} {1e3 * v / n:.3f}" for nm, v in zip(names, host_t)) +
                  f"  (sum {1e3 * sum(host_t) / n:.3f})", file=sys.stderr)
            host_t[:] = [0.0] * 5

    def timed(n):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        assert torch.isfinite(loss)
        return dt
    for _ in range(warmup):
        step()
    host_report(max(warmup, 1))
    dt = timed(steps)
    host_report(steps)
    out = None
    sus = None
    if sustain_s > 0 and dt < sustain_s:
        n_sus = max(steps, int(sustain_s / (dt / steps)))
        dts = timed(n_sus)
        sus = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3, "value": N_RAND * world * n_sus / dts}
    if rank == 0:
        flop_ray = 3 * 64 * (FLOP_PT_HEAD + FLOP_PT_TORSO)          # fwd + 2x bwd, SURVEY.md 8(d)
        ach = flop_ray * N_RAND * world * steps / dt / 1e12
        # The step is bound by the traffic of what the forward records for the backward, not by the MFMAs.  Algorithmic
        # bytes per step and GPU (every array touched once per use, no re-reads): the forward writes the GEMM inputs
        # act_T, the dX chain writes the pre-activation gradients dy_T, the weight-gradient GEMMs read both; rows from
        # dfn_train_rows, NP = 64 * N_rand points, element = the tier's type.
        from dfanerf._lib import lib as _l
        esz = 2 if tier == "bf16" else 4
        NP = 64 * N_RAND
        act_b = sum(_l.dfn_train_rows(f, 0) for f in (0, 1)) * NP * esz
        dy_b = sum(_l.dfn_train_rows(f, 1) for f in (0, 1)) * NP * esz
        step_bytes = 2 * (act_b + dy_b)          # recorded activations and pre-activation gradients: written once, read once (wgrad)
        gbs = step_bytes * world * steps / dt / 1e9
        out = {
            "metric": f"training rays/sec (whole node), N_rand={N_RAND} per GPU, 64 coarse samples, 2 fields, fwd+bwd+Adam",
            "value": N_RAND * world * steps / dt, "unit": "rays/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": tier, "data": "synthetic",
            "config": {"workload": desc, "H": H, "W": W, "N_rand_per_gpu": N_RAND, "n_coarse": 64, "fields": 2,
                       "parallelism": f"dp{world}, one flat-bucket all_reduce (1,138,656 floats)"},
            "roofline": {"bound": "hbm", "kernel": "whole step (render_kernel<train>, mlp_bwd, wgrad_lds)",
                         "achieved": gbs, "peak": 8000.0 * world, "unit": "GB/s", "frac": gbs / (8000.0 * world),
                         "traffic": None, "bytes_per_step_per_gpu": step_bytes,
                         "mfma": {"achieved_tflops": ach, "peak_tflops": PEAK_TFLOPS[tier] * world,
                                  "frac": ach / (PEAK_TFLOPS[tier] * world), "flop_per_ray": flop_ray}}}
        if sus:
            out["sustained"] = sus
    return out


# ---------------------------------------------------------------------------------------------------------------
def bench_render(args, workload, tier, steps, warmup, world, rank, dev, sustain_s=0.0):
    from dfanerf import engine, nets, synth
    from dfanerf._lib import check, lib
    n_fine, fields, desc = WORKLOADS[workload]
    F = 8                                              # frames of the audio-driven sequence (configs[4] batch)
    sc = synth.bench_scene(0, n_frames=F)
    st = synth.synth_all_states(0)
    zs, za = synth.synth_latents(0)
    H, W = sc["H"], sc["W"]
    R = H * W

    # ---- everything resident on the device before timing --------------------------------------------------
    flat = engine.flatten_state(st["decoder"], dev)
    pk = engine.PackedDecoder(flat, tier, fields=(0, 1) if fields == 2 else (0,))
    bg = torch.from_numpy(sc["bg"]).reshape(-1, 3).to(dev)                       # uint8, as the loader has it
    aud_net, exp_net = nets.AudioNet_W2L().to(dev), nets.ExpressionEnc().to(dev)
    att, patt = nets.AudioAttNet(96, 4).to(dev), nets.AudioAttNet(42, 8).to(dev)
    for m, k in ((aud_net, "AudNet"), (exp_net, "ExpNet"), (att, "AudAttNet"), (patt, "PoseAttNet")):
        m.load_state_dict({kk: torch.from_numpy(v) for kk, v in st[k].items()})
    ds = [{"auds": torch.from_numpy(sc["aud"]).to(dev), "exp": torch.from_numpy(sc["exp"]).to(dev),
           "poses": torch.from_numpy(sc["poses"]).to(dev)}]

    class A:
        nosmo_iters, smo_size, smo_torse_size = 300000, 4, 8
    zs_d, za_d = torch.from_numpy(zs[0]).to(dev), torch.from_numpy(za[0]).to(dev)

    per = (R + world - 1) // world                       # rank r renders [r*per, min(R,(r+1)*per))
    begin = rank * per
    count = max(0, min(R, begin + per) - begin)
    # one [per, 3] shard per image the workload produces (head; head + two-field composite for c3), gathered together
    n_img = 2 if fields == 2 else 1
    shard = torch.zeros(n_img, per, 3, dtype=torch.float32, device=dev)
    gathered = torch.empty(world, n_img, per, 3, dtype=torch.float32, device=dev) if world > 1 else None
    state = {"bias": None}
    ev = []

    # conditioning networks in HIP (dfn_encode_signal / dfn_encode_signal_torso)
    enc = engine.SignalEncoder(aud_net, exp_net, att, patt, ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    fid = [torch.tensor([f], dtype=torch.int32, device=dev) for f in range(F)]
    probe = torch.zeros(2, dtype=torch.int64, device=dev)

    def step(i, timed):
        f = i % F
        s2, t2 = enc.encode(fid[f], A.smo_size, A.smo_torse_size)           # 2 launches: [1,96], [1,42]
        sig, sigt = s2[0], (t2[0] if fields == 2 else None)
        state["bias"] = pk.fold(sig, sigt, zs_d, za_d, out=state["bias"])
        fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][f], sc["pose_body"], sc["near"],
                               sc["far"], ray_begin=begin, ray_count=count, n_fine=n_fine, fields=fields)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        engine.render(pk, state["bias"], fr, bg, out_head=shard[0, :count], out_com=shard[1, :count] if fields == 2 else None)
        if timed:
            e1.record()
            ev.append((e0, e1))
        if world > 1:
            dist.all_gather_into_tensor(gathered.view(world * n_img, per, 3), shard)     # concatenation form: every backend
            return gathered
        return shard

    def timed(n, i0):
        ev.clear()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            img = step(i0 + i, True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        assert torch.isfinite(img).all()
        return dt, float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")

    def clock():
        """effective shader clock of the last launch (GHz), read inside the kernel"""
        c = probe.cpu().numpy()
        return float(c[0]) / float(c[1]) * 0.1 if c[1] > 0 else None

    for i in range(warmup):
        step(i, False)
    check(lib.dfn_debug_clock_probe(probe.data_ptr()), "dfn_debug_clock_probe")
    try:
        dt, kern_ms = timed(steps, warmup)
        ghz = clock()
        sus = None
        if sustain_s > 0 and dt < sustain_s:
            n_sus = max(steps, int(sustain_s / (dt / steps)))
            dts, kms = timed(n_sus, warmup + steps)
            sus = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3, "value": R * n_sus / dts,
                   "kernel_ms": kms, "clock_ghz": clock()}
    finally:
        lib.dfn_debug_clock_probe(None)
    if rank != 0:
        return None
    ms_step = dt / steps * 1e3
    # Algorithmic FLOPs of a ray = (decoder evaluations the workload NEEDS) x (2 MACs of the layers the reference
    # evaluates per point).  Row H needs 64 + n_fine evaluations per field: there is one network,
    # so the 64 coarse outputs are reused in the merged pass (identical results).  SURVEY.md 8(d) priced the
    # composition as the NeRF lineage codes it (coarse points evaluated twice: 64 + 64 + n_fine); that figure is
    # reported beside it as `reference_composition_*` - it would put the exact-f32 tier above 100 % of peak, so
    # it is not what `achieved` / `frac` use.
    per_pt = FLOP_PT_HEAD + (FLOP_PT_TORSO if fields == 2 else 0.0)
    evals = 64 + n_fine
    evals_ref = 64 + ((64 + n_fine) if n_fine > 0 else 0)
    flop_ray = evals * per_pt
    achieved = flop_ray * count / (kern_ms * 1e-3) / 1e12
    ref_comp = evals_ref * per_pt * count / (kern_ms * 1e-3) / 1e12
    # HBM bytes per launch cannot be counted from inside this process: they come from the committed rocprofv3
    # PMC passes of this same command (profiles/traffic.json, see profiles/README.md); null if none was taken
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tr = json.load(f).get(f"{workload}_{tier}")
        if tr:
            traffic, traffic_src = tr["hbm_bytes_per_launch"], "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
    except OSError:
        pass
    peak = PEAK_TFLOPS[tier]
    out = {
        "metric": "rays/sec (whole node) at 450x450, 64c+128f samples" if n_fine else
                  "rays/sec (whole node) at 450x450, 64 coarse samples",
        "value": R * steps / dt, "unit": "rays/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms_step, "ms_per_frame": ms_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": tier, "data": "synthetic",
        "config": {"workload": desc, "H": H, "W": W, "n_coarse": 64, "n_fine": n_fine, "fields": fields,
                   "frames": F, "rays_per_step": R,
                   "parallelism": f"rays sharded over {world} GPU(s), all_gather of RGB" if world > 1
                   else "single GPU"},
        "roofline": {"bound": "mfma", "kernel": "render_kernel", "achieved": achieved,
                     "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kern_ms, "clock_ghz": ghz,
                     "flop_per_ray": flop_ray, "decoder_evals_per_ray_per_field": evals,
                     "reference_composition": {"decoder_evals_per_ray_per_field": evals_ref,
                                               "flop_per_ray": evals_ref * per_pt, "tflops": ref_comp,
                                               "frac": ref_comp / peak},
                     "rays_per_launch": count},
    }
    if sus:
        sus["roofline_frac"] = flop_ray * count / (sus["kernel_ms"] * 1e-3) / 1e12 / peak
        out["sustained"] = sus
    out["_scene"] = (sc, st, zs, za, n_fine, fields)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rccl_ranks = 1
    # DFN_BENCH_ONE_GPU=1 (developer switch): every rank on GPU 0 over the gloo backend - a FUNCTIONAL run of the multi-rank
    # code (ray shards, the per-frame gather, the gradient bucket, the optimizers' stream rules) on a box with one GPU;
    # RCCL refuses two ranks on one device.  Its timings mean nothing.
    one_gpu = bool(os.environ.get("DFN_BENCH_ONE_GPU"))
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                        # the ranks RCCL actually connected
        rccl_ranks = int(ones.item())
    if world != args.gpus and rank == 0:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; using {world}", file=sys.stderr)

    train_wl = args.workload in ("c4", "c4s")
    if train_wl:
        out = bench_training(args, args.workload, args.steps, args.warmup, world, rank, dev, args.sustain_seconds)
    else:
        out = bench_render(args, args.workload, args.tier, args.steps, args.warmup, world, rank, dev,
                           args.sustain_seconds)
    extra = {}
    if world == 1 and not args.no_extra:
        # the other BASELINE configs, short runs in the same process (driver-timed, not builder-only numbers)
        for wl, k, w in (("c3", 40, 5), ("c1", 60, 5), ("c4", 150, 20)):
            if wl == args.workload:
                continue
            try:
                r = bench_training(args, wl, k, w, 1, 0, dev) if wl == "c4" else \
                    bench_render(args, wl, args.tier, k, w, 1, 0, dev)
                r.pop("_scene", None)
                extra[wl] = {kk: r[kk] for kk in ("metric", "value", "unit", "steps", "ms_per_step", "dtype", "roofline")}
                extra[wl]["workload"] = r["config"]["workload"]
            except Exception as e:                      # never lose the headline line to a side measurement
                extra[wl] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        scene = out.pop("_scene", None)
        out["rccl_ranks"] = rccl_ranks
        if world == 1 and not args.no_cpu_baseline and scene is not None:
            out["cpu_baseline"] = cpu_baseline(args, *scene)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if extra:
            out["other_workloads"] = extra
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
