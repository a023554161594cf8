"""Deterministic closed-form synthetic weights and inputs.

There is no trained checkpoint and no dataset in the reference tree
(/root/reference/.MISSING_LARGE_BLOBS), so goldens, parity tests and the bench
all run on weights produced by an integer hash of (seed, tensor name, flat
index).  Nothing here depends on a random-number library version: the same
bytes come out in the build container (where the reference is imported to
produce tests/golden/) and on the GPU box (where the reference does not exist).

Shapes follow the reference modules constructed the way scripts/test_obama.sh
constructs them (NeRFs/DFANeRF/run_nerf_com_trainExpLater.py:518-547):
Decoder(z_dim=256, hidden_size=256, dim_signal=96, use_deformation_field=True),
AudioNet_W2L, ExpressionEnc, AudioAttNet(96, 4), AudioAttNet(42, 8).
"""
import zlib

import numpy as np

HIDDEN = 256
Z_DIM = 256
DIM_PE = 60          # 3 * 10 * 2   (decoder.py:203)
DIM_PE_VIEW = 24     # 3 * 4 * 2    (decoder.py:204)
DIM_SIGNAL = 96      # --dim_signal=96 in scripts/*.sh
DIM_ET = 42          # 2 * (3 + 3*2*3), get_embedder(3, 0) on euler and trans
DEFORM_HIDDEN = 64
# synthetic density / colour calibration (see synth_decoder_state)
SIGMA_W, SIGMA_B, FEAT_W = 1.7, -27.0, 0.05


def decoder_shapes(hidden=HIDDEN, z_dim=Z_DIM, dim_signal=DIM_SIGNAL, dim_et=DIM_ET):
    """state_dict key -> shape, in the registration order of decoder.py:207-251."""
    s = {}

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    h = DEFORM_HIDDEN
    # deform_net (DeformationField_ori, decoder.py:84-105)
    lin("deform_net.blocks_embed.0", h, DIM_PE + dim_et)
    for k in range(1, 5):
        lin(f"deform_net.blocks_embed.{k}", h, h)
    lin("deform_net.out_embed", DIM_PE, h)
    lin("deform_net.blocks_signal.0", h, DIM_PE + dim_et)
    for k in range(1, 5):
        lin(f"deform_net.blocks_signal.{k}", h, h)
    lin("deform_net.out_signal", dim_et, h)
    lin("deform_net.fc_embed_skips.0", h, DIM_PE)
    lin("deform_net.fc_signal_skips.0", h, dim_et)
    lin("fc_in", hidden, DIM_PE + dim_signal)
    lin("fc_in_listener", hidden, DIM_PE)
    lin("fc_in_torso", hidden, DIM_PE + dim_et)
    lin("fc_z", hidden, z_dim)
    for k in range(7):
        lin(f"blocks.{k}", hidden, hidden)
    lin("fc_z_skips.0", hidden, z_dim)
    lin("fc_p_skips.0", hidden, DIM_PE + dim_signal)
    lin("fc_p_skips_listener.0", hidden, DIM_PE)
    lin("fc_p_skips_torso.0", hidden, DIM_PE + dim_et)
    lin("sigma_out", 1, hidden)
    lin("fc_z_view", hidden, z_dim)
    lin("feat_view", hidden, hidden)
    lin("fc_view", hidden, DIM_PE_VIEW)
    lin("feat_out", 3, hidden)
    return s


def audnet_shapes():
    return {"encoder.0.weight": (256, 512), "encoder.0.bias": (256,),
            "encoder.2.weight": (128, 256), "encoder.2.bias": (128,),
            "encoder.4.weight": (64, 128), "encoder.4.bias": (64,)}


def expnet_shapes():
    return {"encoder.0.weight": (32, 64), "encoder.0.bias": (32,),
            "encoder.2.weight": (32, 32), "encoder.2.bias": (32,)}


def attnet_shapes(dim_aud, seq_len):
    s = {}
    chans = [dim_aud, 16, 8, 4, 2, 1]
    for k in range(5):
        s[f"attentionConvNet.{2 * k}.weight"] = (chans[k + 1], chans[k], 3)
        s[f"attentionConvNet.{2 * k}.bias"] = (chans[k + 1],)
    s["attentionNet.0.weight"] = (seq_len, seq_len)
    s["attentionNet.0.bias"] = (seq_len,)
    return s


def _hash_uniform(seed, name, n):
    """n values in [-1, 1) from a 64-bit integer mix of (seed, crc32(name), index)."""
    idx = np.arange(n, dtype=np.uint64)
    salt = (0x9E3779B97F4A7C15 * (zlib.crc32(name.encode()) + 1)) & 0xFFFFFFFFFFFFFFFF
    salt ^= (int(seed) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        x = idx + np.uint64(salt)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    # top 24 bits -> exactly representable f32 in [0,1)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (2.0 * u - 1.0).astype(np.float32)


def synth_tensor(seed, name, shape, scale, offset=0.0):
    n = int(np.prod(shape))
    return (_hash_uniform(seed, name, n) * np.float32(scale) + np.float32(offset)).reshape(shape)


def synth_state(shapes, seed, prefix, gain=1.0, overrides=None):
    """Uniform(-a, a) with a = gain*sqrt(6/fan_in) for weights (keeps ReLU
    activations O(1) through depth), a = 0.1 for biases."""
    overrides = overrides or {}
    out = {}
    for k, shp in shapes.items():
        if k.endswith(".weight"):
            fan_in = int(np.prod(shp[1:]))
            a = gain * np.sqrt(6.0 / fan_in)
        else:
            a = 0.1
        a = overrides.get(k, a)
        off = 0.0
        if isinstance(a, tuple):
            a, off = a
        out[k] = synth_tensor(seed, prefix + "/" + k, shp, a, off)
    return out


def synth_decoder_state(seed=0, z_dim=Z_DIM, hidden=HIDDEN):
    """sigma_out is scaled up so that relu(sigma) spans roughly [0, 30] on the
    bench frustum: otherwise every ray is pure background."""
    shapes = decoder_shapes(hidden=hidden, z_dim=z_dim)
    ov = {"sigma_out.weight": SIGMA_W, "sigma_out.bias": (0.0, SIGMA_B),
          "feat_out.weight": FEAT_W}
    return synth_state(shapes, seed, "decoder", overrides=ov)


def synth_all_states(seed=0):
    return {
        "decoder": synth_decoder_state(seed),
        "AudNet": synth_state(audnet_shapes(), seed, "AudNet"),
        "ExpNet": synth_state(expnet_shapes(), seed, "ExpNet"),
        "AudAttNet": synth_state(attnet_shapes(96, 4), seed, "AudAttNet"),
        "PoseAttNet": synth_state(attnet_shapes(42, 8), seed, "PoseAttNet"),
    }


def synth_latents(seed=0, n_object=1, z_dim=Z_DIM):
    """z_shape, z_app [1, 2*n_object, z_dim] ~ roughly N(0,1): sum of 4 uniforms, rescaled."""
    def g(name):
        n = 2 * n_object * z_dim
        acc = np.zeros(n, np.float32)
        for k in range(4):
            acc += _hash_uniform(seed, f"{name}/{k}", n)
        return (acc * np.float32(np.sqrt(3.0 / 4.0))).reshape(1, 2 * n_object, z_dim)
    return g("z_shape"), g("z_app")


def synth_features(seed=0, n_frames=8):
    aud = synth_tensor(seed, "aud", (n_frames, 512), 1.7)        # ~unit variance
    exp = synth_tensor(seed, "exp", (n_frames, 64), 0.5)         # ~0.3 std
    return aud, exp


def euler_pose(euler, trans):
    """4x4 camera-to-world from XYZ euler angles (rad) and translation."""
    a, b, c = [float(v) for v in euler]
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = rx @ ry @ rz
    m[:3, 3] = trans
    return m.astype(np.float32)


def bench_scene(seed=0, n_frames=8, H=450, W=450):
    """The synthetic workload of SURVEY.md 8(d): poses, intrinsics, bg, features."""
    poses = []
    for f in range(n_frames):
        poses.append(euler_pose((0.05 + 0.01 * f, -0.1 + 0.02 * f, 0.02), (0.01, -0.02, 0.6)))
    poses = np.stack(poses)
    pose_body = euler_pose((0.0, 0.0, 0.0), (0.0, 0.0, 0.6))
    bg = ((_hash_uniform(seed, "bg", H * W * 3) * 0.5 + 0.5) * 255.0).astype(np.uint8).reshape(H, W, 3)
    aud, exp = synth_features(seed, n_frames)
    return dict(H=H, W=W, focal=1200.0, cx=W / 2.0, cy=H / 2.0, near=0.3, far=0.9,
                poses=poses, pose_body=pose_body, bg=bg, aud=aud, exp=exp)


def scale_head_activations(decoder_state, s):
    """The same decoder with the HEAD field's hidden activations scaled by s: a ReLU network is positively homogeneous, so
    scaling what enters each layer besides the previous activations (input layer, latent / skip / view projections: weights
    and biases; hidden layers: biases) scales every activation by s, and the two output layers undo it (weights / s): the
    head image is unchanged in exact arithmetic.  (The torso shares the trunk: its image changes.)  Used by the f16 range
    guard's tests (s = 1e4: activations beyond half precision) and bench.py's DFN_BENCH_ACT_SCALE."""
    dec = dict(decoder_state)
    both = ("fc_in", "fc_z", "fc_z_skips.0", "fc_p_skips.0", "fc_z_view", "fc_view")
    for k in list(dec):
        base, _, leaf = k.rpartition(".")
        if base in both or (leaf == "bias" and (base.startswith("blocks.") or base == "feat_view")):
            dec[k] = (dec[k] * np.float32(s)).astype(np.float32)
        elif k in ("sigma_out.weight", "feat_out.weight"):
            dec[k] = (dec[k] / np.float32(s)).astype(np.float32)
    return dec
