"""CPU-only checks of the host side of the C ABI: the library loads, exports every symbol of
include/dfanerf.h, and the pack plan (dfn_plan.cpp) reproduces the reference decoder when the kernel's
dataflow is emulated from it in numpy (no compute call into the library, no GPU)."""
import os
import re

import numpy as np
import pytest
import torch

import dfa_oracle as O
from conftest import ROOT
from dfanerf import _lib, synth

E = {0: 4, 1: 8}
UPT = {0: 4, 1: 2}


def tile_feat(h, r):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def kslot_to_slot(tier, u, h, e):
    t, r = u // UPT[tier], (u % UPT[tier]) * E[tier] + e
    return 32 * t + tile_feat(h, r)


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "dfanerf.h")).read()
    decl = set(re.findall(r"\b(dfn_[a-z0-9_]+)\s*\(", hdr))
    assert decl, "no declarations found"
    import ctypes
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in dfanerf.h but not exported"
    assert set(_lib.EXPORTS) == decl
    assert b"gfx950" in _lib.lib.dfn_version()


def test_developer_switches_cannot_reach_the_product_library(tmp_path):
    """the kernels' timing / ablation switches render wrong results at full speed.  (1) the in-tree library
    is not a developer build; (2) defining any of them without DFN_DEV_BUILD is a compile error (csrc/dfn_devguard.h);
    (3) build.sh refuses DFN_EXTRA_FLAGS without DFN_DEV_BUILD=1 before it compiles anything."""
    import subprocess
    assert b"DEV" not in _lib.lib.dfn_version()
    src = tmp_path / "probe.cpp"
    src.write_text('#include "dfn_layout.h"\nint main() { return dfn::SLAB_FRAGS > 0 ? 0 : 1; }\n')
    inc = ["-I", os.path.join(ROOT, "dfa-nerf_amd", "csrc"), "-I", os.path.join(ROOT, "include")]
    base = ["g++", "-std=c++17", "-fsyntax-only", str(src)] + inc
    assert subprocess.run(base, capture_output=True).returncode == 0
    for sw in ("DFN_EXP_NOEPI", "DFN_REC8_NOAMAX", "DFN_NOMASK", "DFN_TIMING", "DFN_WL_NOMFMA"):
        r = subprocess.run(base + ["-D" + sw], capture_output=True, text=True)
        assert r.returncode != 0 and "DFN_DEV_BUILD" in r.stderr, (sw, r.stderr[-300:])
        assert subprocess.run(base + ["-D" + sw, "-DDFN_DEV_BUILD=1"], capture_output=True).returncode == 0, sw
    # every switch the sources mark "wrong results" is one the guard knows
    import glob
    import re
    guard = open(os.path.join(ROOT, "dfa-nerf_amd", "csrc", "dfn_devguard.h")).read()
    marked = set()
    for f in glob.glob(os.path.join(ROOT, "dfa-nerf_amd", "csrc", "*.h*")):
        for ln in open(f):
            m = re.match(r"\s*#\s*(?:ifdef|elif defined\(|if defined\()\s*(DFN_\w+).*wrong results", ln)
            if m:
                marked.add(m.group(1))
    assert len(marked) >= 10 and all(f"defined({sw})" in guard for sw in marked), sorted(sw for sw in marked if f"defined({sw})" not in guard)
    r = subprocess.run(["bash", os.path.join(ROOT, "dfa-nerf_amd", "build.sh")], capture_output=True, text=True,
                       env=dict(os.environ, DFN_EXTRA_FLAGS="-DDFN_EXP_NOEPI"), timeout=60)
    assert r.returncode == 2 and "DFN_DEV_BUILD" in r.stderr, r.stderr[-300:]
    assert b"DEV" not in _lib.lib.dfn_version() and os.path.exists(_lib.LIB_PATH)        # ... and it touched nothing


def test_error_paths_without_gpu():
    assert _lib.lib.dfn_packed_bytes(7, 0) < 0 and b"bad tier" in _lib.lib.dfn_last_error()
    assert _lib.lib.dfn_bias_floats(0, 9) < 0
    assert _lib.lib.dfn_pack_plan(1, 0, None, 0) == 1152 * 512
    small = np.zeros(4, np.int32)
    assert _lib.lib.dfn_pack_plan(1, 0, small.ctypes.data, 4) == -3      # DFN_E_SIZE
    with pytest.raises(_lib.DfnError):
        _lib.check(_lib.lib.dfn_pack_weights(1, 0, None, None, None), "pack")
    # argument validation of every launch entry point happens before any HIP call: checkable without a GPU
    L, N = _lib.lib, None
    fr = _lib.DfnFrame()
    fr.n_coarse, fr.n_fine, fr.fields, fr.ray_count, fr.H, fr.W = 64, 128, 1, 8, 4, 4
    one = 4096      # any non-null address: the calls below must fail before they would use it
    assert L.dfn_render_fwd(1, None, one, N, one, N, one, N, N, one, N, N, N, N, N) == -1
    assert L.dfn_render_fwd(1, _lib.C.byref(fr), N, N, one, N, one, N, N, one, N, N, N, N, N) == -1      # no weights
    fr.n_fine = 96
    assert L.dfn_render_fwd(1, _lib.C.byref(fr), one, N, one, N, one, N, N, one, N, N, N, N, N) == -1
    assert b"n_fine" in L.dfn_last_error()
    fr.n_fine, fr.fields = 128, 2
    assert L.dfn_render_fwd_u8(1, _lib.C.byref(fr), one, N, one, N, one, N, N, one, N, N) == -1          # torso inputs missing
    assert L.dfn_fold_bias(1, 0, one, N, one, one, one, N) == -1 and b"signal" in L.dfn_last_error()
    assert L.dfn_fold_bias_bwd(1, 0, one, one, one, one, N, one, N, N) == -1
    assert L.dfn_encode_signal(one, one, one, one, one, 8, one, 1, 3, one, N) == -1 and b"smo_size" in L.dfn_last_error()
    assert L.dfn_encode_signal(one, one, N, one, one, 8, one, 1, 4, one, N) == -1                         # attention missing
    assert L.dfn_encode_signal_torso(one, one, 10, 8, one, 1, 8, one, N) == -1                            # pose stride
    assert L.dfn_encode_signal_bwd(one, one, one, one, one, 8, 0, 4, N, one, one, one, N) == -1
    assert L.dfn_encode_signal_torso_bwd(one, one, 16, 8, 0, 0, one, N, N) == 0                           # smo 0: nothing to do
    assert L.dfn_bias_grad(1, 0, N, 64, one, one, N) == -1 and L.dfn_weight_grad(1, 5, one, one, 64, one, one, N) == -1
    # the f16 tier is inference only: pack / fold / render accept it, the training entry points refuse it
    assert L.dfn_packed_bytes(2, 0) == L.dfn_packed_bytes(1, 0) > 0 and L.dfn_bias_floats(2, 1) == L.dfn_bias_floats(1, 1)
    assert L.dfn_packed_bwd_bytes(2, 0) == -1 and L.dfn_weight_grad(2, 0, one, one, 64, one, one, N) == -1
    assert L.dfn_decoder_train_fwd(2, 0, one, one, one, one, 32, one, one, one, one, one, N) == -1
    assert L.dfn_decoder_train_fwd(1, 3, one, one, one, one, 32, one, one, one, one, one, N) == -1        # no such field (2 = listener: trains since round 6)
    assert L.dfn_decoder_fwd(1, 0, one, one, N, one, 4, one, one, N) == -1
    assert L.dfn_adam_multi(N, N, 4, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.03, N) == -1                          # no tables
    assert L.dfn_adam_multi(one, one, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.03, N) == -1                      # t = 0: bias_c1 = 0
    assert L.dfn_adam_multi(N, N, 0, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.03, N) == 0                           # nothing to do
    assert L.dfn_sample_pdf(one, one, 4, 300, 8, N, one, N) == -1                                         # nb <= 256
    # the entry points of the training step's schedule (round 2)
    assert L.dfn_signal_grad(1, 0, one, one, 48, one, one, N) == -1 and b"multiple of 32" in L.dfn_last_error()
    assert L.dfn_signal_grad(1, 2, one, one, 64, one, one, N) == -1 and L.dfn_signal_grad(2, 0, one, one, 64, one, one, N) == -1
    assert L.dfn_signal_grad(1, 0, one, one, 64, N, one, N) == -1                                         # no workspace
    assert L.dfn_train_prepare(1, one, one, N, one, one, one, one, one, one, one, one, N) == -1            # torso signal missing
    assert L.dfn_train_prepare(2, one, one, one, one, one, one, one, one, one, one, one, N) == -1          # f16: inference only
    assert L.dfn_zero_async(N, 16, N) == -1 and L.dfn_zero_async(one, -1, N) == -1 and L.dfn_zero_async(one, 0, N) == 0
    assert L.dfn_train_rows(0, 5) == 128 * 512 + L.dfn_bias_floats(1, 0) and L.dfn_train_rows(1, 5) > 0
    # round 4: the training forward with the loss in its epilogue - a loss block with any null member is refused
    fr.n_fine, fr.fields, fr.ray_count = 0, 2, 8
    args16 = (1, _lib.C.byref(fr), one, one, one, one, one, N, one, one, one, one, one, one, one, one)
    assert L.dfn_train_fwd_loss(*args16, None, N) == -1 and b"loss" in L.dfn_last_error()
    bad = _lib.DfnTrainLoss(one, one, one, one, one, None)                                                # no workspace
    assert L.dfn_train_fwd_loss(*args16, _lib.C.byref(bad), N) == -1
    assert L.dfn_train_fwd_hier_loss(*args16, one, one, None, N) == -1
    assert L.dfn_train_loss_floats(2048) >= 2 * 256 + 1 and L.dfn_train_loss_floats(0) > 0
    assert L.dfn_weight_bias_grad_fmt(1, 0, 7, one, one, 64, one, one, one, N) == -1 and b"act_format" in L.dfn_last_error()
    # DFN_TRAIN_ACT_E4M3 rides in the tier argument of the training forwards only: 16-bit tier only, rejected everywhere else
    assert L.dfn_packed_bytes(1 | _lib.TRAIN_ACT_E4M3, 0) < 0
    assert L.dfn_train_fwd(0 | _lib.TRAIN_ACT_E4M3, *args16[1:], N) == -1 and b"DFN_TRAIN_ACT_E4M3" in L.dfn_last_error()   # f32 tier
    assert L.dfn_train_rows(0, 8) == 2 * (L.dfn_train_rows(0, 6) - 128) + 128           # e4m3 rows are twice the e2m1 rows
    # the weight gradients in two stages (GEMMs into the workspace | the reduction of the slices)
    assert L.dfn_weight_bias_grad_partials(1, 0, 1, N, one, 64, one, one, N) == -1                        # no dy_T
    assert L.dfn_weight_bias_grad_partials(1, 0, 1, one, one, 64, one, N, N) == -1 and b"dbias" in L.dfn_last_error()
    assert L.dfn_weight_bias_grad_reduce(1, 0, 64, one, N, one, N) == -1                                  # no grad_flat
    assert L.dfn_weight_bias_grad_reduce(2, 0, 64, one, one, one, N) == -1                                # f16: inference only


class Reader:
    """Walks the packed stream in consumption order and rebuilds dense weights of each tile group."""

    def __init__(self, tier, field, flat):
        self.tier, self.plan, self.flat, self.pos = tier, _lib.pack_plan(tier, field), flat, 0

    def group(self, G, KU, nslots):
        W = np.zeros((32 * G, nslots))
        e_n = E[self.tier]
        for ku in range(KU):
            for g in range(G):
                frag = self.plan[self.pos:self.pos + 64 * e_n].reshape(64, e_n)
                self.pos += 64 * e_n
                for lane in range(64):
                    i, h = lane & 31, lane >> 5
                    for e in range(e_n):
                        idx = frag[lane, e]
                        if idx >= 0:
                            W[32 * g + i, kslot_to_slot(self.tier, ku, h, e)] = self.flat[idx]
        return W

    def layer(self, OT, KU, nslots):
        return np.concatenate([self.group(2, KU, nslots) for _ in range(OT // 2)], 0)

    def layer_skip(self, OT, KU, nslots, KU2, nslots2):
        a, b = [], []
        for _ in range(OT // 2):
            a.append(self.group(2, KU, nslots))
            b.append(self.group(2, KU2, nslots2))
        return np.concatenate(a, 0), np.concatenate(b, 0)


def flat_params(state):
    return np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in state.values()])


def emulate(tier, field, st, pe, pev, sig, zs, za):
    """numpy restatement of mlp_head / mlp_torso (dfn_mlp.h) driven by the packed stream.
    pe [N,60], pev [N,24] -> feat [N,3], sigma [N].  Biases folded like dfn_misc.hip:fold_kernel."""
    P = {k: np.asarray(v, np.float64) for k, v in st.items()}
    rd = Reader(tier, field, flat_params(st).astype(np.float64))
    u = UPT[tier]
    relu = lambda x: np.maximum(x, 0)
    N = pe.shape[0]
    pe64 = np.zeros((N, 64)); pe64[:, :60] = pe
    v32 = np.zeros((N, 32)); v32[:, :24] = pev
    fcz = P["fc_z.weight"] @ zs + P["fc_z.bias"]
    fczs = P["fc_z_skips.0.weight"] @ zs + P["fc_z_skips.0.bias"]
    fczv = P["fc_z_view.weight"] @ za + P["fc_z_view.bias"]
    if field in (0, 2):
        nm = ("fc_in", "fc_p_skips.0") if field == 0 else ("fc_in_listener", "fc_p_skips_listener.0")
        b_in = P[nm[0] + ".bias"] + fcz
        b_sk = P[nm[1] + ".bias"] + fczs
        if field == 0:
            b_in = b_in + P[nm[0] + ".weight"][:, 60:] @ sig
            b_sk = b_sk + P[nm[1] + ".weight"][:, 60:] @ sig
        act = relu(pe64 @ rd.layer(8, 2 * u, 64).T + b_in)
        pvec, kup, nps = pe64, 2 * u, 64
    else:
        w = lambda n: P[f"deform_net.{n}.weight"]
        b = lambda n: P[f"deform_net.{n}.bias"]
        ve = relu(pe64 @ rd.layer(2, 2 * u, 64).T + b("blocks_embed.0") + w("blocks_embed.0")[:, 60:] @ sig)
        vs = relu(pe64 @ rd.layer(2, 2 * u, 64).T + b("blocks_signal.0") + w("blocks_signal.0")[:, 60:] @ sig)
        ve = relu(ve @ rd.layer(2, 2 * u, 64).T + b("blocks_embed.1"))
        vs = relu(vs @ rd.layer(2, 2 * u, 64).T + b("blocks_signal.1"))
        ve = relu(ve @ rd.layer(2, 2 * u, 64).T + b("blocks_embed.2"))
        vs = relu(vs @ rd.layer(2, 2 * u, 64).T + b("blocks_signal.2"))
        w3, wsk = rd.layer_skip(2, 2 * u, 64, 2 * u, 64)
        ve = relu(ve @ w3.T + b("blocks_embed.3")) + b("fc_embed_skips.0") + pe64 @ wsk.T
        vs = relu(vs @ rd.layer(2, 2 * u, 64).T + b("blocks_signal.3")) + b("fc_signal_skips.0") + \
            w("fc_signal_skips.0") @ sig
        ve = relu(ve @ rd.layer(2, 2 * u, 64).T + b("blocks_embed.4"))
        vs = relu(vs @ rd.layer(2, 2 * u, 64).T + b("blocks_signal.4"))
        eo = ve @ rd.layer(2, 2 * u, 64).T + np.pad(b("out_embed"), (0, 4))
        so = vs @ rd.layer(2, 2 * u, 64).T + np.pad(b("out_signal") + sig, (0, 22))
        pd = np.concatenate([eo + pe64, so], 1)
        act = relu(pd @ rd.layer(8, 4 * u, 128).T + P["fc_in_torso.bias"] + fcz)
        b_sk = P["fc_p_skips_torso.0.bias"] + fczs
        pvec, kup, nps = pd, 4 * u, 128
    for l in range(3):
        act = relu(act @ rd.layer(8, 8 * u, 256).T + P[f"blocks.{l}.bias"])
    w4, wsk = rd.layer_skip(8, 8 * u, 256, kup, nps)
    act = relu(act @ w4.T + P["blocks.3.bias"]) + b_sk + pvec @ wsk.T
    for l in range(4, 7):
        act = relu(act @ rd.layer(8, 8 * u, 256).T + P[f"blocks.{l}.bias"])
    rows = []
    for tg in range(4):
        rows.append(act @ rd.group(2, 8 * u, 256).T + v32 @ rd.group(2, u, 32).T)
    hid = relu(np.concatenate(rows, 1) + P["feat_view.bias"] + fczv + P["fc_view.bias"])
    sg = act @ rd.group(1, 8 * u, 256).T + v32 @ rd.group(1, u, 32).T
    sigma = sg[:, 0] + P["sigma_out.bias"][0]
    assert np.abs(sg[:, 1:]).max() == 0          # rows 1..31 of the sigma tile are structural zeros
    out = hid @ rd.group(1, 8 * u, 256).T
    assert np.abs(out[:, 3:]).max() == 0
    feat = 1 / (1 + np.exp(-(out[:, :3] + P["feat_out.bias"])))
    frag_elems = 64 * E[tier]
    assert rd.pos % frag_elems == 0 and (rd.plan[rd.pos:] == -1).all()     # only slab padding remains
    return feat, sigma


@pytest.mark.parametrize("tier", [0, 1])
@pytest.mark.parametrize("field", [0, 1, 2])
def test_plan_reproduces_reference_decoder(tier, field, golden, states, latents):
    g = golden("g3_decoder")
    st = states["decoder"]
    zs, za = latents
    p, r = torch.from_numpy(g["p_64"][:, :48]), torch.from_numpy(g["r_64"][:, :48])
    pe = O.posenc(p, 10)[0].double().numpy()
    pev = O.posenc(r / torch.norm(r, dim=-1, keepdim=True), 4)[0].double().numpy()
    fi = 1 if field == 1 else 0
    sig = {0: g["sig_aud"][0], 1: g["sig_torso"][0], 2: None}[field]
    feat, sigma = emulate(tier, field, st, pe, pev, None if sig is None else sig.astype(np.float64),
                          zs[0, fi].astype(np.float64), za[0, fi].astype(np.float64))
    name = {0: "head", 1: "torso", 2: "listener"}[field]
    np.testing.assert_allclose(feat, g[f"feat_{name}_64"][0, :48], atol=2e-5, rtol=0)
    np.testing.assert_allclose(sigma, g[f"sigma_{name}_64"][0, :48], atol=2e-4, rtol=1e-5)


def test_f16_tier_shares_the_bf16_plan():
    """The two 16-bit tiers differ in the operand type only: same fragment order, same structural zeros."""
    for field in (0, 1, 2):
        assert np.array_equal(_lib.pack_plan(2, field), _lib.pack_plan(1, field))


def test_flat_param_order_matches_state_dict(states):
    """dfn_layout.h:ParamId order == decoder.state_dict() order: spot-check offsets through the plan."""
    st = states["decoder"]
    off, offs = 0, {}
    for k, v in st.items():
        offs[k] = off
        off += v.size
    assert off == _lib.N_DECODER_PARAMS
    plan = _lib.pack_plan(1, 0)
    used = plan[plan >= 0]
    lo, hi = offs["fc_in.weight"], offs["fc_in.weight"] + st["fc_in.weight"].size
    first = plan[:64 * 8 * 8]                 # first tile pair of fc_in
    assert ((first < 0) | ((first >= lo) & (first < hi))).all()
    for unused in ("fc_in_listener.weight", "fc_in_torso.weight", "deform_net.out_embed.weight"):
        lo, hi = offs[unused], offs[unused] + st[unused].size
        assert not ((used >= lo) & (used < hi)).any()
