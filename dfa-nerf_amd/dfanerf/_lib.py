"""ctypes binding of libdfanerf.so (the C ABI declared in include/dfanerf.h).

The library is the product: there is no Python/CPU fallback.  If the shared object is missing the import
fails loudly and tells the caller how to build it."""
import ctypes as C
import os

import torch  # noqa: F401  (first: the library must bind to the HIP runtime torch has loaded, not a second copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFN_LIB: developer override (A/B builds of the kernels); the shipped path is the in-tree library
LIB_PATH = os.environ.get("DFN_LIB") or os.path.join(_HERE, "libdfanerf.so")

TIER_F32, TIER_BF16, TIER_F16 = 0, 1, 2
TRAIN_ACT_E4M3 = 0x100      # or'ed into the tier of dfn_train_fwd*: e4m3 instead of e2m1 activations (include/dfanerf.h)
ACT_E4M3, ACT_E2M1 = 0, 1
FIELD_HEAD, FIELD_TORSO, FIELD_LISTENER = 0, 1, 2
N_DECODER_PARAMS = 955242
# Linear layers the reference's Decoder registers with use_expression / use_wav2lip (decoder.py:219-228) and never evaluates for the
# one person its scripts train (itr_obj 0: signal = [aud, None], MAIN:70; w2lnet is used nowhere): they are parameters of the
# module (state_dict, checkpoints, optimizer) and no part of the kernels' flat parameter vector
DECODER_UNUSED_PREFIXES = ("expnet.", "w2lnet.")


class DfnFrame(C.Structure):
    _fields_ = [("pose", C.c_float * 12), ("pose_body", C.c_float * 12), ("H", C.c_int), ("W", C.c_int),
                ("focal", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("z_near", C.c_float),
                ("z_far", C.c_float), ("last_dist", C.c_float), ("ray_begin", C.c_int), ("ray_count", C.c_int),
                ("n_coarse", C.c_int), ("n_fine", C.c_int), ("fields", C.c_int), ("concate_bg", C.c_int)]


class DfnTrainLoss(C.Structure):
    _fields_ = [("img_head", C.c_void_p), ("img_com", C.c_void_p), ("d_rgb_head", C.c_void_p), ("d_rgb_com", C.c_void_p),
                ("losses", C.c_void_p), ("workspace", C.c_void_p)]


class DfnError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first "
            f"(python -c 'import __graft_entry__ as g; g.build()' or dfa-nerf_amd/build.sh). "
            f"There is no CPU fallback for the render path.")
    lib = C.CDLL(LIB_PATH)
    vp, fp, ip, lg, i32 = C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int
    sig = {
        "dfn_last_error": (C.c_char_p, []),
        "dfn_version": (C.c_char_p, []),
        "dfn_packed_bytes": (lg, [i32, i32]),
        "dfn_pack_weights": (i32, [i32, i32, fp, vp, vp]),
        "dfn_pack_plan": (lg, [i32, i32, ip, lg]),
        "dfn_bias_floats": (lg, [i32, i32]),
        "dfn_encode_signal": (i32, [fp, fp, fp, fp, fp, i32, ip, i32, i32, fp, vp]),
        "dfn_encode_signal_torso": (i32, [fp, fp, i32, i32, ip, i32, i32, fp, vp]),
        "dfn_encode_signal_bwd": (i32, [fp, fp, fp, fp, fp, i32, i32, i32, fp, fp, fp, fp, vp]),
        "dfn_encode_signal_torso_bwd": (i32, [fp, fp, i32, i32, i32, i32, fp, fp, vp]),
        "dfn_encode_signal_bwd_set": (i32, [fp, fp, fp, fp, fp, i32, i32, i32, fp, fp, fp, fp, vp]),
        "dfn_encode_signal_keep": (i32, [fp, fp, fp, fp, fp, i32, ip, i32, fp, fp, vp]),
        "dfn_encode_signal_keep_floats": (lg, []),
        "dfn_encode_signal_bwd_kept": (i32, [fp, fp, fp, fp, fp, i32, i32, i32, fp, fp, fp, fp, fp, vp]),
        "dfn_encode_signal_torso_bwd_set": (i32, [fp, fp, i32, i32, i32, i32, fp, fp, vp]),
        "dfn_fold_bias": (i32, [i32, i32, fp, fp, fp, fp, fp, vp]),
        "dfn_fold_bias_bwd": (i32, [i32, i32, fp, fp, fp, fp, fp, fp, fp, vp]),
        "dfn_adam_multi": (i32, [vp, vp, i32, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, C.c_float, vp]),
        "dfn_render_fwd": (i32, [i32, C.POINTER(DfnFrame), vp, vp, fp, fp, fp, vp, ip, fp, fp, fp, fp, fp, vp]),
        "dfn_render_fwd_u8": (i32, [i32, C.POINTER(DfnFrame), vp, vp, fp, fp, fp, vp, ip, vp, vp, vp]),
        "dfn_decoder_fwd": (i32, [i32, i32, vp, fp, fp, fp, lg, fp, fp, vp]),
        "dfn_decoder_train_fwd": (i32, [i32, i32, vp, fp, fp, fp, lg, fp, fp, fp, vp, vp, vp]),
        "dfn_get_rays": (i32, [i32, i32, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), fp, fp, vp]),
        "dfn_get_rays_strided": (i32, [i32, i32, i32, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), fp, fp, vp]),
        "dfn_ndc_rays": (i32, [i32, i32, C.c_float, C.c_float, fp, fp, lg, fp, fp, vp]),
        "dfn_sample_pdf": (i32, [fp, fp, lg, i32, i32, fp, fp, vp]),
        "dfn_composite": (i32, [fp, fp, i32, lg, fp, fp, vp]),
        "dfn_volume_weights": (i32, [fp, fp, fp, lg, i32, C.c_float, fp, vp]),
        "dfn_composite_grad": (i32, [fp, fp, i32, lg, fp, fp, fp, fp, vp]),
        "dfn_volume_weights_grad": (i32, [fp, fp, fp, lg, i32, C.c_float, fp, fp, vp]),
        "dfn_to8b": (i32, [fp, lg, vp, vp]),
        "dfn_debug_mfma_layout": (i32, [fp, vp]),
        "dfn_debug_clock_probe": (i32, [vp]),
        "dfn_debug_mfma_chain": (i32, [i32, i32, i32, vp, vp, i32, i32, fp, vp, vp]),
        "dfn_train_rows": (lg, [i32, i32]),
        "dfn_packed_bwd_bytes": (lg, [i32, i32]),
        "dfn_pack_weights_bwd": (i32, [i32, i32, fp, vp, vp]),
        "dfn_train_prepare": (i32, [i32, fp, fp, fp, fp, fp, vp, vp, vp, vp, fp, fp, vp]),
        "dfn_train_fwd": (i32, [i32, C.POINTER(DfnFrame), vp, vp, fp, fp, fp, vp, ip, fp, fp, fp, vp, vp, vp, vp, vp]),
        "dfn_train_fwd_hier": (i32, [i32, C.POINTER(DfnFrame), vp, vp, fp, fp, fp, vp, ip, fp, fp, fp, vp, vp, vp, vp, fp, vp, vp]),
        "dfn_train_loss_floats": (lg, [i32]),
        "dfn_train_fwd_loss": (i32, [i32, C.POINTER(DfnFrame), vp, vp, fp, fp, fp, vp, ip, fp, fp, fp, vp, vp, vp, vp,
                                     C.POINTER(DfnTrainLoss), vp]),
        "dfn_train_fwd_hier_loss": (i32, [i32, C.POINTER(DfnFrame), vp, vp, fp, fp, fp, vp, ip, fp, fp, fp, vp, vp, vp, vp, fp,
                                          vp, C.POINTER(DfnTrainLoss), vp]),
        "dfn_composite_bwd_hier": (i32, [C.POINTER(DfnFrame), ip, fp, vp, fp, fp, vp, fp, fp, fp, vp]),
        "dfn_composite_bwd_hier_z": (i32, [C.POINTER(DfnFrame), ip, fp, vp, fp, fp, vp, fp, fp, fp, fp, lg, vp]),
        "dfn_sample_pixels": (i32, [i32, i32, i32, i32, ip, C.c_uint64, C.c_uint64, ip, ip, vp]),
        "dfn_mse_loss_u8": (i32, [fp, fp, vp, vp, ip, i32, fp, fp, fp, vp]),
        "dfn_composite_bwd": (i32, [C.POINTER(DfnFrame), ip, fp, vp, fp, fp, fp, fp, vp]),
        "dfn_composite_bwd_z": (i32, [C.POINTER(DfnFrame), ip, fp, vp, fp, fp, fp, fp, fp, lg, vp]),
        "dfn_mlp_bwd": (i32, [i32, i32, vp, fp, fp, vp, lg, vp, vp]),
        "dfn_wgrad_plan": (lg, [i32, i32, ip, lg]),
        "dfn_weight_grad": (i32, [i32, i32, vp, vp, lg, fp, fp, vp]),
        "dfn_weight_bias_grad": (i32, [i32, i32, vp, vp, lg, fp, fp, fp, vp]),
        "dfn_weight_bias_grad_fmt": (i32, [i32, i32, i32, vp, vp, lg, fp, fp, fp, vp]),
        "dfn_weight_bias_grad_partials": (i32, [i32, i32, i32, vp, vp, lg, fp, fp, vp]),
        "dfn_weight_bias_grad_reduce": (i32, [i32, i32, lg, fp, fp, fp, vp]),
        "dfn_weight_bias_grad_partials_part": (i32, [i32, i32, i32, vp, vp, lg, fp, fp, i32, vp]),
        "dfn_bias_grad": (i32, [i32, i32, vp, lg, fp, fp, vp]),
        "dfn_zero_async": (i32, [vp, lg, vp]),
        "dfn_signal_grad": (i32, [i32, i32, fp, vp, lg, fp, fp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    # A library built with developer / timing switches (csrc/dfn_devguard.h: ablations that compute WRONG results at full
    # speed, instrumented kernels) says so in its version string.  It is only ever loaded on purpose: through DFN_LIB (the
    # variant libraries of tools/build_variant.sh) - never as the in-tree product library.
    if b"DEV" in lib.dfn_version() and not os.environ.get("DFN_LIB"):
        raise ImportError(f"{LIB_PATH} was built with developer switches ({lib.dfn_version().decode()}): rebuild it with "
                          "dfa-nerf_amd/build.sh --clean (no DFN_EXTRA_FLAGS), or select a variant library explicitly with DFN_LIB")
    return lib, sorted(sig)


lib, EXPORTS = _load()


def check(rc, what=""):
    if rc < 0:
        raise DfnError(f"{what}: {lib.dfn_last_error().decode()} (code {rc})")
    return rc


def pack_plan(tier, field):
    """Host-side pack plan as a numpy int32 array (no GPU needed)."""
    import numpy as np
    n = check(lib.dfn_pack_plan(tier, field, None, 0), "dfn_pack_plan")
    out = np.empty(n, np.int32)
    check(lib.dfn_pack_plan(tier, field, out.ctypes.data, n), "dfn_pack_plan")
    return out
