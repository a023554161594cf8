"""Worker of tests/test_gpu_multirank.py: launched 2 or 8 times by torch.distributed.run, ALL ranks on GPU 0 over the gloo backend
(RCCL refuses two ranks on one device), so that the product's multi-rank code paths run on hardware on a one-GPU box:
FrameRenderer.render_image (ray shards + ONE gather per frame) and the data-parallel training step (FlatGradBucket all-reduce,
replica broadcast, the optimizers' stream rules with more than one rank).  Prints one line 'MULTIRANK_OK ...' from rank 0."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
import numpy as np
import torch
import torch.distributed as dist

from dfanerf import frames, nets, parallel, run_nerf, synth, training
from dfanerf.decoder import Decoder


def t(x):
    return torch.from_numpy(np.asarray(x))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    st, sc = synth.synth_all_states(0), synth.bench_scene(0, n_frames=4)
    zs, za = [t(v).to(dev) for v in synth.synth_latents(0)]
    H, W = sc["H"], sc["W"]
    args = run_nerf.config_parser().parse_args(
        "--expname t --concate_bg --N_rand=512 --sample_rate=0 --smo_size=4 --smo_torse_size 8 --use_et_embed "
        "--dim_signal=96 --dim_aud=96 --n_object=1 --use_deformation_field --noexp_iters 400000 --hip_tier f16 "
        "--hierarchical --N_importance 128".split())

    def modules(seed_shift):
        mods = {"decoder": Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
                "AudNet": nets.AudioNet_W2L(), "ExpNet": nets.ExpressionEnc(), "AudAttNet": nets.AudioAttNet(96, 4),
                "PoseAttNet": nets.AudioAttNet(42, 8)}
        for k, m in mods.items():
            m.load_state_dict({kk: t(v) + (1e-3 * seed_shift if v.dtype.kind == "f" else 0) for kk, v in st[k].items()})
            m.to(dev)
        return mods

    # ---- inference: every rank renders its ray shard, one gather per frame; against the whole frame rendered alone --------
    mods = modules(0)
    bg = (t(sc["bg"]).float() / 255.0).to(dev)
    R = run_nerf.FrameRenderer(mods["decoder"], zs, za, bg, [H, W, sc["focal"], sc["cx"], sc["cy"]], sc["near"], sc["far"], args)
    sig = [torch.randn(1, 96, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.1, None]
    sigt = torch.randn(42, device=dev, generator=torch.Generator(device=dev).manual_seed(2)) * 0.1
    pose, pose_body = sc["poses"][1], sc["pose_body"]
    ok_img = True
    for out_u8 in (False, True):
        rh, rc = R.render_image(pose, pose_body, sig, sigt, fields=2, out_u8=out_u8)          # sharded + gathered
        fh, fc = R.render(pose, pose_body, sig, sigt, fields=2, out_u8=out_u8)                # this rank alone, all rays
        ok_img &= bool(torch.equal(rh.reshape(-1, 3), fh.reshape(-1, 3)) and torch.equal(rc.reshape(-1, 3), fc.reshape(-1, 3)))

    # the two halves as the frame loops use them: frame k + 1 is begun (rendered, its gather issued) BEFORE frame k is
    # finished - two gathers in flight on the two buffer sets - and a third frame reuses the first set
    hs = [R.render_image_begin(sc["poses"][f], pose_body, sig, sigt, fields=2, out_u8=True) for f in (0, 2)]
    got = [tuple(x.clone() for x in R.render_image_end(hs[0]))]
    hs.append(R.render_image_begin(sc["poses"][3], pose_body, sig, sigt, fields=2, out_u8=True))
    got += [tuple(x.clone() for x in R.render_image_end(h)) for h in hs[1:]]
    for f, (gh, gc) in zip((0, 2, 3), got):
        fh, fc = R.render(sc["poses"][f], pose_body, sig, sigt, fields=2, out_u8=True)
        ok_img &= bool(torch.equal(gh.reshape(-1, 3), fh) and torch.equal(gc.reshape(-1, 3), fc))

    # the render-person loop of run_nerf.train(): the front end one frame ahead (engine.FramePrefetcher), consecutive frames
    # on the two render streams, the gather of frame k underneath the render of frame k + 1 - 24 frames in a row, every
    # gathered frame against the frame one rank renders whole from the same blob
    from dfanerf import engine
    enc = engine.SignalEncoder(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"], t(sc["aud"]).to(dev),
                               t(sc["exp"]).to(dev), t(sc["poses"]).to(dev))
    pk = mods["decoder"].packed(R.tier)
    refs = []
    for f in range(4):
        s2, t2 = enc.encode([f], 4, 8)
        refs.append(tuple(x.clone() for x in R.render(sc["poses"][f], pose_body, None, None, fields=2, out_u8=True,
                                                      bias=pk.fold(s2[0], t2[0], R.zs, R.za))))
    pf = engine.FramePrefetcher(enc, pk, R.zs, R.za, 4, 8, fields=2)
    order = [(7 * k + (k >> 2)) % 4 for k in range(24)]
    pending, ok_loop = None, True

    def finish(h, f):
        gh, gc = R.render_image_end(h)
        return bool(torch.equal(gh.reshape(-1, 3), refs[f][0]) and torch.equal(gc.reshape(-1, 3), refs[f][1]))
    for k, f in enumerate(order):
        bias = pf.get(f, order[k + 1] if k + 1 < len(order) else None)
        h = R.render_image_begin(sc["poses"][f], pose_body, None, None, out_u8=True, bias=bias)
        with torch.cuda.stream(h.get("stream") or torch.cuda.current_stream(dev)):
            pf.done()
        if pending is not None:
            ok_loop &= finish(*pending)
        pending = (h, f)
    ok_loop &= finish(*pending)
    ok_img &= ok_loop

    # ---- training: replicas that start DIFFERENT, broadcast, three data-parallel steps on different frames ---------------
    mods = modules(rank)                                   # rank 1 starts from other values
    opts = {k: run_nerf.make_adam(m.parameters(), 5e-4) for k, m in mods.items()}
    parallel.broadcast_replicas(mods, opts)
    ds = [{"auds": t(sc["aud"]).to(dev), "exp": t(sc["exp"]).to(dev), "poses": t(sc["poses"]).to(dev), "bc_img": bg,
           "hwfcxy": [H, W, sc["focal"], sc["cx"], sc["cy"]], "near": sc["near"], "far": sc["far"]}]
    buf = training.TrainBuffers("bf16", 512, dev)
    buf.signal_trainer = training.SignalTrainer(mods["AudNet"], mods["ExpNet"], mods["AudAttNet"], mods["PoseAttNet"],
                                                ds[0]["auds"], ds[0]["exp"], ds[0]["poses"])
    buf.signal_trainer.adopt_optimizers(opts)
    bucket = parallel.StepReducer(mods, opts, buf.signal_trainer)          # as run_nerf.train() builds it
    assert len(bucket.side) == 2
    sampler = frames.PixelSampler(H, W, 512, 0, dev, seed=50 + rank, pipeline=True, stream=buf.signal_trainer.pose_stream())
    gen = torch.Generator(device=dev).manual_seed(9 + rank)
    gt = (torch.randint(0, 256, (H * W, 3), dtype=torch.uint8, device=dev, generator=gen),
          torch.randint(0, 256, (H * W, 3), dtype=torch.uint8, device=dev, generator=gen))
    embed_fn, _ = nets.get_embedder(3, 0)
    assert parallel.multi_rank_schedule()
    losses = []
    for k in range(3):
        loss, *_ = run_nerf.train_step_loss_hip(mods, ds, 0, (k + rank) % 4, sampler.draw(), gt[0], gt[1], zs, za, 300000,
                                                args, 4, embed_fn, ds[0]["poses"][0], buf)
        for o in opts.values():
            o.zero_grad()
        loss.backward()
        bucket.all_reduce_()
        run_nerf.optimizer_steps(opts, 300000, args)
        losses.append(float(loss))
    buf.signal_trainer.join()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for m in mods.values() for p in m.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    same = all(bool(torch.equal(both[0], b)) for b in both[1:])          # every replica against rank 0's (world = 2 or 8)
    if not same and rank == 0:                                            # which network, which rank, by how much
        off = 0
        for name, m in mods.items():
            n = sum(p.numel() for p in m.parameters())
            d = [float((both[0][off:off + n] - b[off:off + n]).abs().max()) for b in both[1:]]
            print(f"MULTIRANK_DIFF {name}: max |rank 0 - rank r| for r = 1.. : {d}", flush=True)
            off += n
    moved = bool((flat != torch.cat([t(v).reshape(-1).to(dev) for k in mods for v in
                                     [st[k][n] for n, _ in mods[k].named_parameters()]])).any())
    # (finite: the parameters too - torch.equal is False on NaNs, so a non-finite gradient anywhere would read as "replicas differ")
    finite = bool(all(np.isfinite(losses)) and torch.isfinite(flat).all())
    if rank == 0:
        print(f"MULTIRANK_OK image={ok_img} replicas_identical={same} trained={moved} finite={finite}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
