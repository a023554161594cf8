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
                 timings of the launches inside the timed region (`kernel_ms`; under the N > 1 schedule, where two launches
                 co-run on the two render streams, the union of the launch intervals per launch, with the plain mean of
                 the event pairs beside it as `kernel_ms_event_pairs`); `clock_ghz` = the shader clock under load read inside
                 the kernel (s_memtime / s_memrealtime): the launch is power-bound, the 2.5 PF peak assumes 2.4 GHz;
  sustained    - the same step repeated for >= --sustain-seconds (default 10 s) after the K timed steps: the K steps
                 the driver asks for last well under a second, the chip's DVFS settles later;
  cpu_baseline - the oracle (oracle/dfa_oracle.py, a port of the reference's CPU path) timed on a bounded
                 sample of the same workload on this host's cores (rank 0, N == 1 only): `value` = the best of
                 {one process at the best thread count of a sweep, P processes x T threads filling every physical
                 core for T in 16, 32}; `cores` = the cores that configuration used;
  other_workloads - (N == 1) short runs of the other BASELINE configs in the same process: c3 (two-field render), c1
                 (coarse), c2_f32 (the exact tier: the one that meets "within 1e-4 PSNR"), c4 / c4h (training step, coarse and
                 hierarchical, 16-bit training tier - `dtype` spells out its operand formats) and c4_f32 (the training step
                 in the exact tier), so that they are driver-timed numbers too;
  f16_range    - (N == 1, f16 tier) the range guard's calibration (dfanerf/f16guard.py) on the bench scene, after the timed loops: max
                 |activation| over 256 rays of each frame in the exact tier and max |parameter| against half precision's 65504; and
                 the accuracy guard's numbers on the same sample: psnr_vs_f32_db (f16 images against the exact tier's, worse of
                 head / composite), the worst frame, the gate "within 0.05 dB" implies for a 30-dB model;
  parity_check - (N == 1) after the timed loops, 64 rays of the LAST timed frame rendered again in the timed configuration
                 and compared with the CPU oracle (outside the timed region): binds the timed launch to the parity suite.

`python bench.py --gpus N` (N > 1) without a torchrun environment launches its own N ranks (torch.distributed.run on
127.0.0.1, a probed free port) - the driver's plain `python bench.py --gpus 8` and the torchrun form give the same run.
For N > 1 the per-frame gather is issued async_op=True on a double-buffered shard: frame k's gather runs on RCCL's
stream underneath frame k + 1's render; `--workload c5` renders an 8-frame batch per step with ONE
gather for the batch (configs[4]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))

import dfanerf  # noqa: F401,E402  (before the first GPU call: multi-rank processes ask the runtime for eight hardware queues)
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
    # the hierarchical variant SURVEY.md 8(d) asks to report beside c4: 64 coarse + 128 fine samples (fine depths detached)
    "c4h": (128, 2, "Obama training step, hierarchical 64+128 samples (fine depths detached), N_rand=2048 per GPU, fwd+bwd+Adam"),
    # configs[4]: a step = a batch of 8 audio-driven frames (each the C2 frame), rays of every frame sharded over the ranks,
    # ONE gather for the whole batch
    "c5": (128, 1, "8-frame audio-driven batch inference, 450x450, 64+128, rays sharded across the GPUs, one RCCL gather per batch (configs[4])"),
}
TRAIN_WORKLOADS = ("c4", "c4s", "c4h")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)        # 300 frames = ~10 s: a sustained number by default
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["all"],
                    help="all = the headline (c2) plus short runs of the other configs in `other_workloads` at ANY N: "
                         "c3, c5, c4, c4s under the process group (N > 1), per-workload error capture")
    ap.add_argument("--tier", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample (after the sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=10.0,
                    help="length of the sustained run after the K timed steps (0 = off; skipped when the K steps "
                         "themselves already lasted that long)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other workloads (N == 1)")
    ap.add_argument("--size", type=int, default=450, choices=[450, 512],
                    help="frame size of the render workloads (512 = what scripts/process_data.sh emits; the counter passes of "
                         "`c2_512` in profiles/traffic.json use it)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the 64-ray oracle check after the timed loops (N == 1)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)      # internal: one process of cpu_baseline.multi_process
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
            sweep[nt] = 2048 / run(101250, 2048, 2048)           # rays/s on ONE WHOLE 2048-ray chunk from the middle of the frame
        best = max(sweep, key=sweep.get)                         # (round 4 swept 1024-ray pieces: its optimum moved between 8 and 32)
        torch.set_num_threads(best)
        chunk, done, t_used = 2048, 0, 0.0
        while t_used < args.cpu_seconds and done < H * W:
            n = min(chunk, H * W - done)
            t_used += run(done, n, chunk)
            done += n
    out = {"value": done / t_used, "unit": "rays/s", "cores": int(best), "kind": "port",
           "host_physical_cores": int(cores), "single_process_value": done / t_used, "single_process_threads": int(best),
           "thread_sweep_rays_per_s": {str(k): round(v, 1) for k, v in sweep.items()},
           "sample": f"{done} rays ({done // chunk} chunks of 2048) of frame 0, same workload, fp32, "
                     f"torch {torch.__version__} CPU, {best} threads (best of the sweep), {t_used:.1f} s.  The sweep's optimum is "
                     "the limit of torch's INTRA-OP parallelism on 2048-ray chunks (more threads are slower), not the host's "
                     "capacity: `multi_process` runs that many-thread process several times side by side"}
    # The host's capacity - the north star's "reference CPU render timed on the host cores of the same box (core count stated)":
    # P processes of T threads each with P * T = every physical core, side by side on disjoint chunks of the frame, for T in
    # {16, 32} (one process cannot use a 128-core host: the sweep's optimum is 8-32 threads).  The BEST of these (and of the
    # single process) is `value`, with cores = P * T: the steadier and the fairer figure (round 4's line reported the single
    # process at the sweep's noisy optimum: 582-712 rays/s by run).
    mp_runs = []
    if not os.environ.get("DFN_BENCH_NO_CPU_MP"):
        import subprocess
        # (round 6: THREE repetitions of TWO whole 2048-ray chunks per process - every chunk is the complete path of its rays,
        # coarse pass, sample_pdf, fine pass, compositing - and the MEDIAN repetition is the figure; round 5 timed one chunk per
        # process once and moved by +-5 % from run to run.  One thread count per run: 16 where the host has 32+ cores.)
        REPS, CHUNKS = 3, 2
        for T in ((16,) if int(cores) // 16 >= 2 else (32, 8, 4)):
            n_proc = int(cores) // T
            if n_proc < 2:
                continue
            try:
                span = H * W - CHUNKS * REPS * chunk
                procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                                           f"{(i * CHUNKS * REPS * chunk) % span},{CHUNKS},{T},{n_fine},{fields},{REPS}"],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                          env=dict(os.environ, OMP_NUM_THREADS=str(T), MKL_NUM_THREADS=str(T)))
                         for i in range(n_proc)]
                res = []
                for pr in procs:
                    o, _ = pr.communicate(timeout=900)
                    res.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
                # the processes run the same amount of work side by side: a repetition's rate = the sum of the processes' rates
                rates = sorted(sum(r["rays"] / r["seconds"][k] for r in res) for k in range(REPS))
                secs = max(sum(r["seconds"]) for r in res)
                mp_runs.append({"value": rates[len(rates) // 2], "unit": "rays/s", "processes": n_proc, "threads_per_process": T,
                                "cores": n_proc * T, "seconds": secs, "chunks_per_process": CHUNKS * REPS, "repetitions": REPS,
                                "repetition_rates": [round(v, 1) for v in rates],
                                "spread": round((rates[-1] - rates[0]) / rates[len(rates) // 2], 4)})
                break
            except Exception as e:                      # a reported extra: never lose the line to it
                mp_runs.append({"error": f"{type(e).__name__}: {e}", "threads_per_process": T})
    good = [m for m in mp_runs if "value" in m]
    if good:
        top = max(good, key=lambda m: m["value"])
        out["multi_process"] = top
        out["multi_process_runs"] = mp_runs
        if top["value"] > out["value"]:
            out["value"], out["cores"] = top["value"], int(top["cores"])
            out["sample"] = (f"{top['processes']} processes x {top['threads_per_process']} threads = {top['cores']} of the host's "
                             f"{int(cores)} physical cores, each process {top['repetitions']} repetitions of {top['chunks_per_process'] // top['repetitions']} "
                             f"whole 2048-ray chunks of frame 0 (coarse pass, sample_pdf, fine pass, compositing) concurrently "
                             f"({top['seconds']:.1f} s), median repetition (spread {top['spread'] * 100:.1f} %), same workload, fp32, "
                             f"torch {torch.__version__} CPU: the oracle on "
                             f"every host core.  One process alone: {out['single_process_value']:.0f} rays/s at {best} threads, the "
                             "best of the thread sweep (torch's intra-op parallelism on 2048-ray chunks stops scaling at 8-32 threads)")
    elif mp_runs:
        out["multi_process"] = mp_runs[-1]
    return out


def cpu_worker(spec):
    """one process of cpu_baseline.multi_process: `chunks` 2048-ray chunks of frame 0 from ray `begin` on `threads` threads"""
    begin, chunks, threads, n_fine, fields, reps = ([int(v) for v in spec.split(",")] + [1])[:6]
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dfa_oracle as O
    from dfanerf import synth
    torch.set_num_threads(threads)
    sc = synth.bench_scene(0, n_frames=8)
    st = synth.synth_all_states(0)
    zs, za = synth.synth_latents(0)
    P = O.params_to_torch(st["decoder"])
    nets = {k: O.params_to_torch(v) for k, v in st.items() if k != "decoder"}
    auds, exps, poses = [torch.from_numpy(sc[k]) for k in ("aud", "exp", "poses")]
    H, W = sc["H"], sc["W"]
    bg = torch.from_numpy(sc["bg"]).float() / 255.0
    with torch.no_grad():
        sig = O.encode_signal(nets, auds, exps, 0, 300000, 300000, 4, auds.shape[0])
        sigt = O.encode_signal_torso(nets, poses, 0, 300000, 300000, 8, poses.shape[0])

        def run(b, n):
            t0 = time.perf_counter()
            O.render_frame(P, H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][0], sc["pose_body"], bg, sc["near"], sc["far"],
                           torch.from_numpy(zs), torch.from_numpy(za), sig, sigt, 64, n_fine, fields, 2048, ray_begin=b, ray_count=n)
            return time.perf_counter() - t0
        run(0, 256)
        if reps > 1:
            run(begin, 2048)          # one whole untimed chunk: the repetitions of a run rose by 4-5 % each from a cold start (round 6)
        secs = [sum(run(begin + 2048 * (chunks * k + c), 2048) for c in range(chunks)) for k in range(reps)]
    print(json.dumps({"rays": 2048 * chunks, "seconds": secs}))


# ---------------------------------------------------------------------------------------------------------------
def bench_training(args, workload, steps, warmup, world, rank, dev, sustain_s=0.0, tier=None):
    """configs[3]: one optimisation step per `step`: signals -> fold -> fused HIP forward (recorder on) -> MSE
    losses -> HIP backward (compositing, dX chain, weight-gradient GEMMs) -> flat-bucket all_reduce -> gated Adams.
    Ground-truth frames and background are resident uint8 device tensors; pixels are drawn and targets gathered on
    the device (frames.PixelSampler, dfn_mse_loss_u8): no host image read, no host-to-device copy per step."""
    from dfanerf import nets, parallel, run_nerf, synth, training
    from dfanerf.decoder import Decoder
    tier = tier or ("bf16" if args.tier == "f16" else args.tier)           # the 16-bit training tier (f16 is inference only)
    n_fine, _, desc = WORKLOADS[workload]
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
        (f"--expname b --concate_bg --N_rand={N_RAND} --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
         "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000" +
         (f" --hierarchical --N_importance {n_fine}" if n_fine else "")).split())
    ds = [{"auds": t(sc["aud"]).to(dev), "exp": t(sc["exp"]).to(dev), "poses": t(sc["poses"]).to(dev),
           "bc_img": (t(sc["bg"]).float() / 255.0).to(dev), "hwfcxy": [H, W, sc["focal"], sc["cx"], sc["cy"]],
           "near": sc["near"], "far": sc["far"]}]
    zs, za = [t(v).to(dev) for v in synth.synth_latents(0)]
    embed_fn, _ = nets.get_embedder(3, 0)
    opts = {k: run_nerf.make_adam(m.parameters(), 5e-4) for k, m in mods.items()}
    buf = training.TrainBuffers(tier, N_RAND, dev, n_fine=n_fine)
    buf.signal_trainer = training.SignalTrainer(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                                ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    buf.signal_trainer.adopt_optimizers(opts)
    bucket = parallel.StepReducer(mods, opts, buf.signal_trainer) if world > 1 or parallel.multi_rank_schedule() else None
    if os.environ.get("DFN_BENCH_ONE_BUCKET"):       # developer switch: the single-collective form
        bucket = parallel.FlatGradBucket(list(mods.values())) if bucket is not None else None
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
                                  stream=None if os.environ.get("DFN_BENCH_FIFTH_STREAM") else run_nerf.draw_stream(buf))
    gstep = 300000                                   # all five optimizers' gates exercised except ExpNet

    host_t = [0.0] * 5 if os.environ.get("DFN_BENCH_HOST_TIMING") else None      # developer switch: host time by section

    # DFN_BENCH_GPU_PHASES=1 (developer switch): events on the main stream at the phase boundaries of every step - untraced GPU
    # time of forward (incl. waiting for the conditioning signals) / backward / optimizer.  NOT free: four timing events per
    # step took the step from 1.00 to 1.37 ms (an event record in a queue is a barrier packet with a system-scope release:
    # ~13 us between two kernels, tools/event_cost.py) - read the split, not the sum
    phases = [] if os.environ.get("DFN_BENCH_GPU_PHASES") else None

    def mark(tag):
        if phases is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            phases.append((tag, e))

    def step():
        c = time.perf_counter
        t0 = c()
        mark("start")
        img_i = int(rng_frame.randint(0, 8))
        pix = sampler.draw()
        loss, *_ = run_nerf.train_step_loss_hip(mods, ds, 0, img_i, pix, gt[img_i][0], gt[img_i][1], zs, za, gstep, a,
                                                8, embed_fn, ds[0]["poses"][0], buf)
        mark("fwd")
        t1 = c()
        for o in opts.values():
            o.zero_grad()
        t2 = c()
        training.backward(loss, buf)             # loss.backward() started with the buffers' unit gradient, as run_nerf.train does
        mark("bwd")
        t3 = c()
        if bucket is not None:
            bucket.all_reduce_()
        run_nerf.optimizer_steps(opts, gstep, a)
        mark("opt")
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
        dt_rank = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        assert torch.isfinite(loss)
        state["dt_rank"] = dt_rank
        state["loss"] = float(loss)
        return dt
    state = {}
    for _ in range(warmup):
        step()
    host_report(max(warmup, 1))
    if phases is not None:
        phases.clear()
    dt = timed(steps)
    if phases is not None and rank == 0:
        acc, n4 = {}, len(phases) // 4
        for k in range(1, n4):                       # (skip the first step)
            (_, e0), (_, e1), (_, e2), (_, e3) = phases[4 * k:4 * k + 4]
            prev = phases[4 * k - 1][1]
            for tag, a_, b_ in (("idle before the step's first event", prev, e0), ("draw + signals wait + prepare + forward + loss", e0, e1),
                                ("backward", e1, e2), ("reduce + Adam", e2, e3)):
                acc[tag] = acc.get(tag, 0.0) + a_.elapsed_time(b_)
        print("GPU ms/step on the main stream: " + ", ".join(f"{k_} {v / (n4 - 1):.4f}" for k_, v in acc.items()) +
              f"  (sum {sum(acc.values()) / (n4 - 1):.4f})", file=sys.stderr)
    rank_ms = all_ranks(state["dt_rank"] / steps * 1e3, world, dev)
    host_report(steps)
    out = None
    sus = None
    if sustain_s > 0 and dt < sustain_s:
        n_sus = max(steps, int(sustain_s / (dt / steps)))
        dts = timed(n_sus)
        sus = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3, "value": N_RAND * world * n_sus / dts}
    if rank == 0:
        S = 64 + n_fine
        flop_ray = 3 * S * (FLOP_PT_HEAD + FLOP_PT_TORSO)          # fwd + 2x bwd, SURVEY.md 8(d) (C4: 481.5 MFLOP/ray)
        ach = flop_ray * N_RAND * world * steps / dt / 1e12
        # SURVEY.md 8(d) prices the training step as MFMA-bound: `frac` is the MFMA fraction.  What actually bounds the step
        # today is the traffic of what the forward records for the backward (LABNOTES.md 7): `traffic` = bytes the DESIGN moves
        # per step and GPU - the forward writes the GEMM inputs act_T, the dX chain the pre-activation gradients dy_T, the
        # weight-gradient GEMMs read both (rows from dfn_train_rows, NP = S * N_rand points, element = the tier's type) -
        # next to the ALGORITHMIC bytes of the step (pixel ids, targets, background, the four weight streams, the parameter
        # read of the pack, the gradient write): the ratio is the traffic the recorded design adds.
        from dfanerf._lib import lib as _l
        NP = S * N_RAND
        if tier == "bf16":       # MX-fp8 recording: [tile][rows x 32 e4m3 bytes + scale block]
            act_b = sum(_l.dfn_train_rows(f, 6) for f in (0, 1)) * (NP // 32)
            dy_b = sum(_l.dfn_train_rows(f, 7) for f in (0, 1)) * (NP // 32)
        else:
            act_b = sum(_l.dfn_train_rows(f, 0) for f in (0, 1)) * NP * 4
            dy_b = sum(_l.dfn_train_rows(f, 1) for f in (0, 1)) * NP * 4
        step_bytes = 2 * (act_b + dy_b)          # recorded activations and pre-activation gradients: written once, read once (wgrad)
        t_id = 1 if tier == "bf16" else 0
        streams = sum(_l.dfn_packed_bytes(t_id, f) + _l.dfn_packed_bwd_bytes(t_id, f) for f in (0, 1))
        alg_bytes = N_RAND * (4 + 3 + 3 + 3 + 24) + 2 * streams + 2 * 4 * 1138656
        # measured HBM bytes per step (rocprofv3 PMC passes of this command, profiles/traffic.json), when a pass was committed
        traffic, traffic_src = step_bytes, ("design bytes per step and GPU: recorded activations + pre-activation gradients, "
                                            "written once and read once")
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tr = json.load(f).get(f"{workload}_{tier}")
            if tr and world == 1:
                traffic = tr["hbm_bytes_per_launch"]
                traffic_src = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel, summed over the step's launches)"
        except OSError:
            pass
        gbs = traffic * world * steps / dt / 1e9
        peak = PEAK_TFLOPS[tier] * world
        out = {
            "metric": f"training rays/sec (whole node), N_rand={N_RAND} per GPU, {S} samples, 2 fields, fwd+bwd+Adam",
            "value": N_RAND * world * steps / dt, "unit": "rays/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            # what the step computes in: the 16-bit training tier is bf16 MFMAs in the forward and the dX chain, and the weight
            # gradients on block-scaled MX operands (dY e4m3 x recorded activations e2m1 or e4m3) - the label says so
            "vs_baseline": None, "dtype": (tier if tier != "bf16" else
                                           "bf16 fwd+dX, mx-fp8(e4m3) x mx-" + ("fp4(e2m1)" if buf.act_format == 1 else "fp8(e4m3)") + " wgrad"),
            "data": "synthetic",
            "config": {"workload": desc, "H": H, "W": W, "N_rand_per_gpu": N_RAND, "n_coarse": 64, "n_fine": n_fine,
                       "recorded_activation_format": (("e2m1 (MX-fp4)" if buf.act_format == 1 else "e4m3 (MX-fp8)") if tier == "bf16" else "f32"),
                       "fields": 2, "parallelism": f"dp{world}, three in-place all_reduces per step, each on the stream its gradients are produced on: audio-side networks (180,785 floats), PoseAttNet (2,629), decoder (955,242)"},
            "roofline": {"bound": "mfma", "kernel": "whole step (render_kernel<train>, mlp_bwd, " + ("wgrad_mx)" if tier == "bf16" else "wgrad_full + wgrad_narrow)"),
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "flop_per_ray": flop_ray,
                         "traffic": traffic, "traffic_source": traffic_src, "recorded_bytes_per_step": step_bytes,
                         "algorithmic_bytes": alg_bytes, "traffic_over_algorithmic": traffic / alg_bytes,
                         "hbm": {"achieved_gbs": gbs, "peak_gbs": 8000.0 * world, "frac": gbs / (8000.0 * world)}},
            "per_rank": {"ms_per_step": rank_ms}}
        if sus:
            out["sustained"] = sus
        if os.environ.get("DFN_BENCH_PRINT_LOSS"):      # (tests: the loss of the last timed step, rank 0)
            out["last_loss"] = state["loss"]
    return out


# ---------------------------------------------------------------------------------------------------------------
def all_ranks(x, world, dev):
    """[value of every rank] (float64), gathered through an all_reduce of a one-hot vector (every backend has that)"""
    if world == 1:
        return [float(x)]
    v = torch.zeros(world, dtype=torch.float64, device=dev)
    v[dist.get_rank()] = float(x)
    dist.all_reduce(v)
    return [float(t) for t in v.cpu()]


def parity_check(sc, st, zs, za, pk, n_fine, fields, frame, dev, tier):
    """64 rays of frame `frame` (the last timed one) through the HIP path exactly as the timed loop ran it - HIP signal
    encoders, fold, fused render in the timed tier - against the CPU oracle (oracle/dfa_oracle.py: its own signal encoders,
    decoder, compositing), OUTSIDE the timed region.  Two distances: against the oracle evaluated at the depths the kernel
    sampled (decoder + compositing; sample_pdf's `denom < 1e-5` switch makes depths in empty space rounding-sensitive) and
    against the oracle's whole pipeline with its own fine sampler."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dfa_oracle as O
    from dfanerf import engine, nets
    H, W = sc["H"], sc["W"]
    idx = np.arange(0, H * W, (H * W) // 64)[:64].astype(np.int32)
    t = lambda x: torch.from_numpy(np.asarray(x))
    mods = {"AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
            "PoseAttNet": nets.AudioAttNet(42, 8)}
    for k, m in mods.items():
        m.load_state_dict({kk: t(v) for kk, v in st[k].items()})
        m.to(dev)
    auds, exps, poses = [t(sc[k]).to(dev) for k in ("aud", "exp", "poses")]
    enc = engine.SignalEncoder(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"], auds, exps, poses)
    s2, t2 = enc.encode([frame], 4, 8)
    bias = pk.fold(s2[0], t2[0] if fields == 2 else None, t(zs[0]).to(dev), t(za[0]).to(dev))
    bg8 = t(sc["bg"]).reshape(-1, 3).to(dev)
    fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][frame], sc["pose_body"], sc["near"], sc["far"],
                           ray_count=len(idx), n_fine=n_fine, fields=fields)
    out = engine.render(pk, bias, fr, bg8, pix_index=t(idx).to(dev), want_z=True)
    torch.cuda.synchronize()
    rh, rc, z = out[0].cpu(), (out[1].cpu() if fields == 2 else None), out[-1].cpu()
    P = O.params_to_torch(st["decoder"])
    onets = {k: O.params_to_torch(v) for k, v in st.items() if k != "decoder"}
    bg = (t(sc["bg"]).float() / 255.0).reshape(-1, 3)[idx]
    with torch.no_grad():
        sig = O.encode_signal(onets, t(sc["aud"]), t(sc["exp"]), frame, 300000, 300000, 4, sc["aud"].shape[0])
        sigt = O.encode_signal_torso(onets, t(sc["poses"]), frame, 300000, 300000, 8, sc["poses"].shape[0])
        o_h, d_h = O.get_rays(H, W, sc["focal"], sc["poses"][frame][:3, :4], sc["cx"], sc["cy"])
        o_t, d_t = O.get_rays(H, W, sc["focal"], sc["pose_body"][:3, :4], sc["cx"], sc["cy"])
        rays = [x.reshape(-1, 3)[idx] for x in (o_h, d_h, o_t, d_t)]
        fh, fc = O.render_fixed_samples(P, *rays, bg, z, t(zs), t(za), sig, sigt, fields)
        ph, pc = O.render_rays_chunk(P, *rays, bg, sc["near"], sc["far"], t(zs), t(za), sig, sigt, 64, n_fine, fields)
    def dist_(a, b):
        d = (a - b).double()
        mse = float((d ** 2).mean())
        return float(d.abs().max()), (10.0 * np.log10(1.0 / mse) if mse > 0 else float("inf"))
    got, ref_f, ref_p = (rc, fc, pc) if fields == 2 else (rh, fh, ph)       # the final image of the workload
    m_f, p_f = dist_(got, ref_f)
    m_p, p_p = dist_(got, ref_p)
    return {"rays": int(len(idx)), "frame": int(frame), "tier": tier, "image": "composite" if fields == 2 else "head",
            "max_abs_rgb": m_f, "psnr_db": p_f, "reference": "oracle decoder + compositing at the depths the kernel sampled",
            "vs_oracle_pipeline": {"max_abs_rgb": m_p, "psnr_db": p_p,
                                   "reference": "oracle end to end (own sample_pdf; a fine depth that flips sample_pdf's "
                                                "`denom < 1e-5` switch moves within one coarse bin)"}}


def power_ceiling(pk, tier, dev, kernel_tflops):
    """What the matrix pipe sustains under the chip's power limit on THIS box, measured in-process right after the timed loops
    (outside the timed region, ~1 s): dfn_debug_mfma_chain = render_kernel's inner loop reduced to its cost drivers, on the
    renderer's own operand statistics - A fragments = the first 32 KiB of the packed weight stream the timed launches read,
    B operands = post-ReLU-like activations (half zeros, |N(0,1)| otherwise) in the tier's type.  `bare_chain` = MFMAs only
    (operands in registers), `renderer_mix` = + one 1-KiB LDS fragment read per MFMA and two convert / max instructions per
    MFMA (what the decoder's epilogue cannot avoid).  The 2.5 PFLOP/s peak needs 2.4 GHz, which the part only holds when the
    multipliers do not toggle (LABNOTES.md 4.6; tools/vendor_gemm_ceiling.py calibrates the same ceiling with hipBLASLt)."""
    import ctypes as C
    from dfanerf._lib import check as chk, lib
    from dfanerf.engine import TIERS
    t = TIERS[tier]
    frags = pk.packed[0][:32768].contiguous()
    g = torch.Generator(device=dev).manual_seed(5)
    act = torch.randn(32768, device=dev, generator=g).abs() * (torch.rand(32768, device=dev, generator=g) < 0.5)
    b = act.to(torch.float16 if tier == "f16" else torch.bfloat16).contiguous()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks, iters = 4 * cus, 12000
    out = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    res = {}
    for name, (l, v) in (("bare_chain", (0, 0)), ("renderer_mix", (2, 4))):
        best = None
        for rep in range(3):
            n = 500 if rep == 0 else iters                       # (warm-up first)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            chk(lib.dfn_debug_mfma_chain(t, l, v, C.c_void_p(frags.data_ptr()), C.c_void_p(b.data_ptr()), n, blocks,
                                         C.c_void_p(out.data_ptr()), C.c_void_p(clk.data_ptr()), st), "dfn_debug_mfma_chain")
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ms = e0.elapsed_time(e1)
                c = clk.cpu().numpy()
                tf = blocks * 8 * n * 32 * 32768.0 / (ms * 1e-3) / 1e12
                if best is None or tf > best[0]:
                    best = (tf, float(c[0]) / float(c[1]) * 0.1 if c[1] > 0 else None)
        res[name] = {"tflops": best[0], "frac_of_peak": best[0] / PEAK_TFLOPS[tier], "clock_ghz": best[1]}
    res["frac_of_ceiling"] = kernel_tflops / res["bare_chain"]["tflops"]          # the timed kernel against the bare chain
    res["frac_of_renderer_mix"] = kernel_tflops / res["renderer_mix"]["tflops"]
    res["operands"] = "A: the packed weight stream's first 32 fragments; B: post-ReLU-like (half zeros, |N(0,1)|), " + tier
    return res


def bench_render(args, workload, tier, steps, warmup, world, rank, dev, sustain_s=0.0, check=False, size=450):
    from dfanerf import engine, nets, synth
    from dfanerf._lib import check as chk, lib
    n_fine, fields, desc = WORKLOADS[workload]
    F = 8                                              # frames of the audio-driven sequence (configs[4] batch)
    B = 8 if workload == "c5" else 1                   # frames per step = frames per gather
    sc = synth.bench_scene(0, n_frames=F, H=size, W=size)
    if size != 450:                                    # same field of view as the 450 x 450 scene
        sc["focal"] = 1200.0 * size / 450
        desc = desc.replace("450x450", f"{size}x{size}")
    st = synth.synth_all_states(0)
    if os.environ.get("DFN_BENCH_ACT_SCALE"):
        # developer switch (the DFN_EXP_CLAMPCVT build): the HEAD field's hidden activations scaled by s - a ReLU
        # network is positively homogeneous, so scaling what enters each layer besides the previous activations (input layer,
        # latent / skip / view projections: weights and biases; hidden layers: biases) scales every activation by s and
        # nothing else.  With s = 1/16 the synthetic network's activations (max 13) stay below 1.
        st["decoder"] = synth.scale_head_activations(st["decoder"], float(os.environ["DFN_BENCH_ACT_SCALE"]))
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
    # one [per, 3] shard per image the workload produces (head; head + two-field composite for c3) and frame of the batch,
    # gathered together.  TWO sets of buffers: the gather of step k is issued async_op=True (RCCL's own stream) and only
    # waited for when its buffers come round again at step k + 2, so it runs underneath the render of step k + 1.
    n_img = 2 if fields == 2 else 1
    shards = [torch.zeros(B, n_img, per, 3, dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty(world, B, n_img, per, 3, dtype=torch.float32, device=dev) if (world > 1 or dist.is_initialized())
                else None for _ in range(2)]
    works = [None, None]
    ev = []

    # conditioning networks in HIP (dfn_encode_signal / dfn_encode_signal_torso)
    enc = engine.SignalEncoder(aud_net, exp_net, att, patt, ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    prefetch = engine.FramePrefetcher(enc, pk, zs_d, za_d, A.smo_size, A.smo_torse_size, fields=fields)
    probe = torch.zeros(2, dtype=torch.int64, device=dev)

    # N > 1: consecutive FRAMES alternate between two render streams.  A rank's shard is 12.4 rounds of workgroups at 8 ranks:
    # on one stream the 13th, partial round leaves 64 % of the compute units idle (5 % of the frame); the next frame's
    # workgroups, launched on the other stream, start on them (tools/shard_scaling.py --two-streams, one GPU: a 1/8 shard
    # 4.09 -> 3.90 ms per frame, 95.6 % -> 100 % of linear; 1/4: 99.2 -> 100 %).  N == 1: one stream, as profiled.
    main_stream = torch.cuda.current_stream(dev)
    # (DFN_BENCH_RCCL_WORLD1: a process group of ONE rank on the real RCCL backend runs the whole N > 1 schedule - two render
    # streams, the prefetcher, the asynchronous all_gather_into_tensor on RCCL's own stream, eight hardware queues - on a box
    # with one GPU: everything a rank of an 8-GPU run does per frame except the wire)
    from dfanerf import parallel
    multi = world > 1 or parallel.multi_rank_schedule()
    if multi:
        rstreams = [engine.side_stream(dev, role="render_a"), engine.side_stream(dev, role="render_b")]
        for rs in rstreams:
            rs.wait_stream(main_stream)                  # the set-up above ran on the main stream
    else:
        rstreams = [main_stream, main_stream]

    def step(i, timed):
        k = i & 1
        used = sorted({(i * B + b) & 1 for b in range(B)})
        if works[k] is not None:
            for u in used:                               # the gather that last used these buffers (two steps ago)
                with torch.cuda.stream(rstreams[u]):
                    works[k].wait()
            works[k] = None
        shard = shards[k]
        for b in range(B):
            g = i * B + b
            f = g % F
            rs = rstreams[g & 1]
            with torch.cuda.stream(rs):
                # conditioning signals (2 encoder launches) + bias fold of THIS frame were started underneath the previous
                # frame's render (engine.FramePrefetcher); the next frame's start underneath this one's
                bias = prefetch.get(f, next_frame=(g + 1) % F)
                fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][f], sc["pose_body"], sc["near"],
                                       sc["far"], ray_begin=begin, ray_count=count, n_fine=n_fine, fields=fields)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(rs)
                engine.render(pk, bias, fr, bg, out_head=shard[b, 0, :count],
                              out_com=shard[b, 1, :count] if fields == 2 else None)
                if timed:
                    e1.record(rs)
                    ev.append((e0, e1))
                prefetch.done()
        if multi:
            last = rstreams[(i * B + B - 1) & 1]
            for u in used:                               # (a batch: frames of the step ran on both streams)
                if rstreams[u] is not last:
                    last.wait_stream(rstreams[u])
            with torch.cuda.stream(last):
                # concatenation form (every backend takes it); async: the collective waits for this stream's work so far and
                # the NEXT step's render does not wait for the collective
                works[k] = dist.all_gather_into_tensor(gathered[k].view(world * B * n_img, per, 3),
                                                       shard.view(B * n_img, per, 3), async_op=True)
            return gathered[k]
        return shard

    def drain():
        for k in (0, 1):
            if works[k] is not None:
                works[k].wait()
                works[k] = None

    def timed(n, i0):
        ev.clear()
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            img = step(i0 + i, True)
        drain()                                          # every gather of the timed steps has completed
        torch.cuda.synchronize()
        dt_rank = time.perf_counter() - t0               # this rank's own time, before it waits for the others
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        assert torch.isfinite(img).all()
        state["last"] = (i0 + n - 1, img)
        return dt, launch_ms(), dt_rank

    def launch_ms():
        """Average duration of a render_kernel launch from the HIP events around every launch of the timed region.  With ONE
        render stream that is the mean of the event pairs.  Under the N > 1 schedule consecutive frames run on two alternating
        streams and their launches CO-RUN (a launch's last, partial round of workgroups shares the chip with the next
        launch): a pair's own elapsed time then counts the time it shared, so the figure is the length of the UNION of the
        launches' [start, end] intervals (all events placed on the first event's clock) over the number of launches - the
        time the kernel occupied the GPU per launch.  Both are in the line (`kernel_ms`, `kernel_ms_event_pairs`)."""
        if not ev:
            state["pair_ms"] = float("nan")
            return float("nan")
        state["pair_ms"] = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        if not multi:
            return state["pair_ms"]
        z = ev[0][0]
        return union_length([(z.elapsed_time(a), z.elapsed_time(b)) for a, b in ev]) / len(ev)

    state = {}

    def gather_check():
        """N > 1, rank 0, outside the timed region: the frame the LAST timed step gathered (its last frame, if a batch) against
        the same frame rendered whole by this rank alone - every rank's shard, as it arrived, bit for bit."""
        i_last, img = state["last"]
        f = (i_last * B + B - 1) % F
        got = img.view(world, B, n_img, per, 3)[:, B - 1].permute(1, 0, 2, 3).reshape(n_img, world * per, 3)[:, :R].clone()
        s2, t2 = enc.encode([f], A.smo_size, A.smo_torse_size)
        bias = pk.fold(s2[0], t2[0] if fields == 2 else None, zs_d, za_d)
        fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][f], sc["pose_body"], sc["near"], sc["far"],
                               ray_begin=0, ray_count=R, n_fine=n_fine, fields=fields)
        whole = torch.empty(n_img, R, 3, dtype=torch.float32, device=dev)
        engine.render(pk, bias, fr, bg, out_head=whole[0], out_com=whole[1] if fields == 2 else None)
        torch.cuda.synchronize()
        return {"frame": f, "identical": bool(torch.equal(got, whole)), "max_abs_diff": float((got - whole).abs().max())}

    def clock():
        """effective shader clock of the last launch (GHz), read inside the kernel"""
        c = probe.cpu().numpy()
        return float(c[0]) / float(c[1]) * 0.1 if c[1] > 0 else None

    for i in range(warmup):
        step(i, False)
    chk(lib.dfn_debug_clock_probe(probe.data_ptr()), "dfn_debug_clock_probe")
    try:
        dt, kern_ms, dt_rank = timed(steps, warmup)
        pair_ms = state["pair_ms"]
        ghz = clock()
        sus = None
        if sustain_s > 0 and dt < sustain_s:
            n_sus = max(steps, int(sustain_s / (dt / steps)))
            dts, kms, _ = timed(n_sus, warmup + steps)
            sus = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3, "value": R * B * n_sus / dts,
                   "kernel_ms": kms, "clock_ghz": clock()}
    finally:
        lib.dfn_debug_clock_probe(None)
    # per-rank figures (every rank contributes): its own wall time per step and its render_kernel time per launch
    rank_ms = all_ranks(dt_rank / steps * 1e3, world, dev)
    rank_kern = all_ranks(kern_ms, world, dev)
    gcheck = None
    if multi and rank == 0:
        try:
            gcheck = gather_check()
        except Exception as e:                          # never lose the line to the check; the failure is in the line
            gcheck = {"error": f"{type(e).__name__}: {e}"}
    # the collective alone (nothing to overlap with): 20 gathers of the step's shard, synchronised
    gather_ms = None
    if multi:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_gather_into_tensor(gathered[0].view(world * B * n_img, per, 3), shards[0].view(B * n_img, per, 3))
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) / 20 * 1e3
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
            tr = json.load(f).get(f"{'c2' if workload == 'c5' else workload}{'_512' if size == 512 else ''}_{tier}")
        if tr and world == 1:
            traffic, traffic_src = tr["hbm_bytes_per_launch"], "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
    except OSError:
        pass
    peak = PEAK_TFLOPS[tier]
    out = {
        "metric": f"rays/sec (whole node) at {H}x{W}, 64c+128f samples" if n_fine else
                  f"rays/sec (whole node) at {H}x{W}, 64 coarse samples",
        "value": R * B * steps / dt, "unit": "rays/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms_step, "ms_per_frame": ms_step / B, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": tier, "data": "synthetic",
        "config": {"workload": desc, "H": H, "W": W, "n_coarse": 64, "n_fine": n_fine, "fields": fields,
                   "frames": F, "frames_per_step": B, "rays_per_step": R * B,
                   "parallelism": (f"rays of every frame sharded over {world} GPUs ({per} rays per rank), one async "
                                   f"all_gather_into_tensor of the RGB shards per {'batch of 8 frames' if B > 1 else 'frame'}, "
                                   "double-buffered (overlaps the next step's render)") if world > 1 else "single GPU"},
        "roofline": {"bound": "mfma", "kernel": "render_kernel", "achieved": achieved,
                     "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kern_ms,
                     "kernel_ms_event_pairs": pair_ms, "render_streams": 2 if multi else 1, "clock_ghz": ghz,
                     "flop_per_ray": flop_ray, "decoder_evals_per_ray_per_field": evals,
                     "reference_composition": {"decoder_evals_per_ray_per_field": evals_ref,
                                               "flop_per_ray": evals_ref * per_pt, "tflops": ref_comp,
                                               "frac": ref_comp / peak},
                     "rays_per_launch": count,
                     # the whole job against the whole node's peak, from the wall clock (N > 1: consecutive frames overlap
                     # on two streams, so a launch's event time includes its neighbours' head and tail)
                     "whole_job": {"tflops": flop_ray * R * B * steps / dt / 1e12,
                                   "frac": flop_ray * R * B * steps / dt / 1e12 / (peak * world)}},
        "per_rank": {"ms_per_step": rank_ms, "render_kernel_ms": rank_kern},
    }
    if multi:
        out["gather"] = {"ms_alone": gather_ms, "bytes_per_rank": int(B * n_img * per * 12), "async": True,
                         "collective": "all_gather_into_tensor", "render_streams": 2}
        out["gather_check"] = gcheck
    if sus:
        sus["roofline_frac"] = flop_ray * count / (sus["kernel_ms"] * 1e-3) / 1e12 / peak
        out["sustained"] = sus
    if check and world == 1 and tier in ("f16", "bf16") and not os.environ.get("DFN_BENCH_NO_CEILING"):
        try:                                        # after the timed loops; never lose the headline line to it
            pc = out["roofline"]["power_ceiling"] = power_ceiling(pk, tier, dev, out["sustained"]["roofline_frac"] * peak
                                                                  if sus else achieved)
            # the same as SCALAR keys of `roofline` (a line parser that keeps only scalars keeps these): what the matrix pipe
            # sustains under the power limit on THIS box as a fraction of the 2.5 PF peak - the bare MFMA chain and the
            # renderer's instruction mix - and the timed kernel against the latter
            out["roofline"]["ceiling_bare_frac"] = pc["bare_chain"]["frac_of_peak"]
            out["roofline"]["ceiling_mix_frac"] = pc["renderer_mix"]["frac_of_peak"]
            out["roofline"]["frac_of_mix"] = pc["frac_of_renderer_mix"]
            out["roofline"]["ceiling_mix_clock_ghz"] = pc["renderer_mix"]["clock_ghz"]
        except Exception as e:
            out["roofline"]["power_ceiling"] = {"error": f"{type(e).__name__}: {e}"}
    if check and world == 1 and tier == "f16":
        # the f16 tier's range guard (dfanerf/f16guard.py), as the render CLI runs it before a sequence: max |activation| per
        # layer over 256 rays of each of the F frames in the exact tier, max |parameter|; outside the timed region
        try:
            from dfanerf import f16guard
            frs, sh, stt = [], [], []
            for f in range(F):
                s2, t2 = enc.encode([f], A.smo_size, A.smo_torse_size)
                sh.append(s2[0])
                stt.append(t2[0])
                frs.append(engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][f], sc["pose_body"], sc["near"], sc["far"]))
            b = f16guard.activation_bounds(flat, frs, sh, stt, zs_d, za_d, bg, n_fine=n_fine)
            top = f16guard.check(b, pk.f16_weight_max)
            out["f16_range"] = {"max_activation": top, "max_parameter": pk.f16_weight_max, "f16_max": f16guard.F16_MAX,
                                "margin": f16guard.MARGIN, "frames": F, "rays_per_frame": 256,
                                "worst_layer": max(((v, f"{fld}: {ly}") for fld, d in b.items() for ly, v in d.items()))[1]}
            # ... and its accuracy guard (round 6): the same 256 rays x F frames in the f16 tier AND the exact tier, both fields,
            # the workload's n_fine - PSNR of the f16 images against the exact tier's, whole sample and worst frame, next to the
            # gate the north star's "within 0.05 dB" needs for a 30-dB model (f16guard.psnr_gate)
            pk16, pk32 = engine.PackedDecoder(flat, "f16"), engine.PackedDecoder(flat, "f32")     # (both fields, whatever the workload packs)
            gen, blocks = torch.Generator(device="cpu").manual_seed(0), []
            with torch.no_grad():
                for f in range(F):
                    pix = torch.randperm(H * W, generator=gen)[:256].to(torch.int32).to(dev)
                    img = {}
                    for name, p_ in (("f16", pk16), ("f32", pk32)):
                        fr = engine.make_frame(H, W, sc["focal"], sc["cx"], sc["cy"], sc["poses"][f], sc["pose_body"], sc["near"],
                                               sc["far"], ray_count=256, n_fine=n_fine, fields=2)
                        img[name] = engine.render(p_, p_.fold(sh[f], stt[f], zs_d, za_d), fr, bg, pix_index=pix)
                    blocks.append({"head": (img["f16"][0], img["f32"][0], None), "com": (img["f16"][1], img["f32"][1], None)})
            acc = f16guard.accuracy_stats(blocks)
            out["f16_range"].update({"psnr_vs_f32_db": round(min(a["psnr_db"] for a in acc.values()), 2),
                                     "psnr_vs_f32_worst_frame_db": round(min(a["worst_block_db"] for a in acc.values()), 2),
                                     "psnr_gate_db": round(f16guard.psnr_gate(f16guard.DEFAULT_MODEL_PSNR), 2),
                                     "psnr_by_image": {k: round(a["psnr_db"], 2) for k, a in acc.items()},
                                     "n_fine": int(n_fine)})
        except Exception as e:
            out["f16_range"] = {"error": f"{type(e).__name__}: {e}"}
    if check and world == 1:
        try:
            out["parity_check"] = parity_check(sc, st, zs, za, pk, n_fine, fields, ((warmup + steps) * B - 1) % F, dev, tier)
        except Exception as e:                      # never lose the headline line to the check; the failure is in the line
            out["parity_check"] = {"error": f"{type(e).__name__}: {e}"}
    out["_scene"] = (sc, st, zs, za, n_fine, fields)
    return out


def union_length(intervals):
    """Total length covered by a list of (start, end) intervals (any order, overlaps counted once)."""
    iv = sorted(intervals)
    if not iv:
        return 0.0
    busy, (lo, hi) = 0.0, iv[0]
    for a, b in iv[1:]:
        if a > hi:
            busy, lo, hi = busy + (hi - lo), a, b
        else:
            hi = max(hi, b)
    return busy + (hi - lo)


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment: run the N ranks ourselves - the command the driver
    documents, on 127.0.0.1 and a probed free port (one retry: a free port can be taken between the probe and the
    rendezvous).  The ranks' stdout / stderr pass through; rank 0 prints the JSON line."""
    import subprocess
    one_gpu = bool(os.environ.get("DFN_BENCH_ONE_GPU"))
    if not one_gpu and torch.cuda.device_count() < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) visible "
              "(DFN_BENCH_ONE_GPU=1 runs every rank on GPU 0 over gloo: functional, not a measurement)", file=sys.stderr)
        return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = 1
    for _ in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.call(cmd, env=env)
        if rc == 0:
            break
    return rc


def main():
    args = parse()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
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
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if one_gpu else "nccl"
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                        # the ranks RCCL actually connected
        rccl_ranks = int(ones.item())
    # DFN_BENCH_RCCL_WORLD1=1 (developer switch, one GPU): a process group of ONE rank on the real RCCL backend and the
    # multi-rank training schedule (DFN_FORCE_MULTIRANK: bucket all_reduce through RCCL's own stream, head weight gradients on
    # the main stream, optimizer streams ordered behind the collective) - what N > 1 runs per rank, minus the exchange.
    if world == 1 and os.environ.get("DFN_BENCH_RCCL_WORLD1"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ["DFN_FORCE_MULTIRANK"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        backend = "nccl (one rank, DFN_BENCH_RCCL_WORLD1)"
    if world != args.gpus and rank == 0:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; using {world}", file=sys.stderr)

    run_all = args.workload == "all"
    headline = "c2" if run_all else args.workload
    train_wl = headline in TRAIN_WORKLOADS
    if train_wl:
        out = bench_training(args, headline, args.steps, args.warmup, world, rank, dev, args.sustain_seconds)
    else:
        out = bench_render(args, headline, args.tier, args.steps, args.warmup, world, rank, dev,
                           args.sustain_seconds, check=not args.no_parity_check, size=args.size)
    extra = {}
    multi = world > 1 or dist.is_initialized()
    if (world == 1 and not args.no_extra) or run_all:
        # the other BASELINE configs, short runs in the same process (driver-timed, not builder-only numbers); c2_f32 = the
        # exact tier, the one that meets the north star's "within 1e-4 PSNR" clause.  Under a process group (--workload all
        # at N > 1, or DFN_BENCH_RCCL_WORLD1) every rank runs the same list in the same order; a workload that raises on a
        # rank is recorded in the line (`error`) and, after the ranks have agreed on it, the list goes on.
        todo = [("c3", "c3", args.tier, 40, 5), ("c1", "c1", args.tier, 60, 5), ("c2_f32", "c2", "f32", 5, 1),
                ("c5", "c5", args.tier, 4, 1), ("c4", "c4", args.tier, 150, 20), ("c4h", "c4h", args.tier, 60, 10),
                # the training step in the EXACT tier (f32 MFMAs, f32 recording): the reference's own arithmetic, driver-timed
                ("c4_f32", "c4", "f32", 20, 3)]
        if world == 1 and not multi:
            # 512 x 512: the frame size the reference's own preprocessing emits (scripts/process_data.sh:4)
            todo.insert(3, ("c2_512", "c2", args.tier, 10, 2))
        if world > 1:
            todo = [("c3", "c3", args.tier, 40, 5), ("c5", "c5", args.tier, 4, 1), ("c4", "c4", args.tier, 150, 20),
                    ("c4s", "c4s", args.tier, 150, 20)]
        for name, wl, tier, k, w in todo:
            if wl == headline and tier == args.tier and name != "c2_512":
                continue
            r, err = None, None
            try:
                if wl in TRAIN_WORKLOADS:
                    r = bench_training(args, wl, k, w, world, rank, dev, tier="f32" if name == "c4_f32" else None)
                else:
                    r = bench_render(args, wl, tier, k, w, world, rank, dev, size=512 if name == "c2_512" else 450,
                                     check=(name == "c2_f32" and not args.no_parity_check))
            except Exception as e:                      # never lose the headline line to a side measurement
                err = f"{type(e).__name__}: {e}"
            if world > 1:
                try:                                     # agree: a rank that failed marks the workload failed on rank 0's line
                    bad = torch.tensor([1.0 if err else 0.0], device=dev)
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                    if bad.item() > 0 and err is None:
                        err = "failed on another rank"
                except Exception as e:
                    err = err or f"{type(e).__name__}: {e}"
            if rank != 0:
                continue
            if err or r is None:
                extra[name] = {"error": err or "no result"}
                print(f"bench.py: workload {name} failed: {err}", file=sys.stderr)
                continue
            r.pop("_scene", None)
            extra[name] = {kk: r[kk] for kk in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "ms_per_frame", "dtype",
                                                "scaling", "roofline", "parity_check", "per_rank", "gather", "gather_check")
                           if kk in r}
            extra[name]["workload"] = r["config"]["workload"]
            print(f"bench.py: workload {name}: {r['ms_per_step']:.3f} ms/step", file=sys.stderr)
    if rank == 0:
        scene = out.pop("_scene", None)
        out["rccl_ranks"] = rccl_ranks
        out["backend"] = backend
        if world == 1 and not args.no_cpu_baseline and scene is not None:
            out["cpu_baseline"] = cpu_baseline(args, *scene)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if extra:
            out["other_workloads"] = extra
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
