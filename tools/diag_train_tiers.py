import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "dfa-nerf_amd", "oracle"): sys.path.insert(0, os.path.join(R, d))
import numpy as np, torch
import dfa_oracle as O
from dfanerf import synth, engine, training
import test_gpu_train_hier as T
scene = synth.bench_scene(0, n_frames=8); states = synth.synth_all_states(0); latents = synth.synth_latents(0)
t = T.t
onets = {k: O.params_to_torch(v) for k, v in states.items() if k != "decoder"}
with torch.no_grad():
    SIG = O.encode_signal(onets, t(scene["aud"]), t(scene["exp"]), 3, 0, 300000, 4, 8)[0]
    SIGT = O.encode_signal_torso(onets, t(scene["poses"]), 3, 0, 300000, 8, 8).reshape(-1)
print("signal rms", float(SIG.pow(2).mean().sqrt()), float(SIGT.pow(2).mean().sqrt()))
_orig = synth.synth_tensor
def _patched(seed, name, shape, scale):
    if name == "g3/sig": return SIG.reshape(shape).numpy().copy()
    if name == "g3/sigt": return SIGT.reshape(shape).numpy().copy()
    return _orig(seed, name, shape, scale)
if os.environ.get("DIAG_ENCODED"): synth.synth_tensor = _patched
def run(tier, n_fine, n):
    if n_fine:
        r = T._hier_step(states, scene, latents, tier, n_fine, n)
    else:
        # coarse variant through the same helper: patch
        dev = torch.device("cuda")
        H, W = scene["H"], scene["W"]
        zs, za = [t(v).to(dev) for v in latents]
        bg = (t(scene["bg"]).float() / 255.0).reshape(-1, 3).to(dev)
        pix = (torch.arange(n, dtype=torch.int64) * 3163) % (H * W)
        frame = engine.make_frame(H, W, scene["focal"], scene["cx"], scene["cy"], scene["poses"][1], scene["pose_body"], 0.3, 0.9, 1e10, 0, n, 64, 0, 2, True)
        tgt = torch.rand(n, 3, generator=torch.Generator().manual_seed(5))
        dec = T._decoder(states, dev)
        sh = t(synth.synth_tensor(0, "g3/sig", (1, 96), 0.8)).to(dev).requires_grad_(True)
        st = t(synth.synth_tensor(0, "g3/sigt", (42,), 0.8)).to(dev).requires_grad_(True)
        buf = training.TrainBuffers(tier, n, dev)
        rh, rc = training.render_train(dec, buf, frame, bg, pix.to(dev, torch.int32), sh, st, zs[0, :2], za[0, :2])
        loss = ((rh - tgt.to(dev)) ** 2).mean() + ((rc - tgt.to(dev)) ** 2).mean()
        loss.backward(); torch.cuda.synchronize()
        r = dict(z=O.coarse_z(0.3,0.9,64)[None].expand(n,64).contiguous(), d_sh=sh.grad.cpu(), d_st=st.grad.cpu(), pix=pix, tgt=tgt, bg=bg.cpu(), sh=sh.detach().cpu(), st=st.detach().cpu(),
                 grads={k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in dec.named_parameters()})
    H, W = scene["H"], scene["W"]
    P = O.params_to_torch(states["decoder"])
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    sh_o = r["sh"].clone().requires_grad_(True); st_o = r["st"].clone()[None].requires_grad_(True)
    o_h, d_h = O.get_rays(H, W, scene["focal"], scene["poses"][1][:3, :4], scene["cx"], scene["cy"])
    o_t, d_t = O.get_rays(H, W, scene["focal"], scene["pose_body"][:3, :4], scene["cx"], scene["cy"])
    rays = [x.reshape(-1, 3)[r["pix"]] for x in (o_h, d_h, o_t, d_t)]
    zs, za = [t(v) for v in latents]
    oh, oc = O.render_fixed_samples(Pg, *rays, r["bg"][r["pix"]], r["z"], zs, za, [sh_o, None], st_o, 2)
    lo = ((oh - r["tgt"]) ** 2).mean() + ((oc - r["tgt"]) ** 2).mean(); lo.backward()
    rel = lambda a, b: float((a.reshape(-1) - b.reshape(-1)).norm() / (b.norm() + 1e-30))
    worst_n, worst_d, wk = 0, 0, None
    for k, g in r["grads"].items():
        if g is None: continue
        ref = Pg[k].grad; rn = float(ref.norm())
        if rn == 0: continue
        en = abs(float(g.norm()) - rn) / rn; ed = rel(g, ref)
        if ed > worst_d: worst_d, wk = ed, k
        worst_n = max(worst_n, en)
    print(f"{tier} n_fine={n_fine} n={n}: d_sh {rel(r['d_sh'], sh_o.grad):.3f} d_st {rel(r['d_st'], st_o.grad):.3f} worst norm err {worst_n:.3f} worst dir err {worst_d:.3f} ({wk})", flush=True)
for tier, nf, n in (("bf16", 0, 64), ("bf16", 128, 64), ("bf16", 0, 256), ("bf16", 128, 256), ("f32", 128, 64)):
    run(tier, nf, n)
