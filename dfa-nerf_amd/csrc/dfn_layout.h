// dfn_layout.h - register/LDS/HBM layout shared by the host-side pack planner and the gfx950 kernels.
//
// Everything in the fused decoder is organised around the C/D fragment map of the CDNA4 32x32 MFMA
// (v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32 share it):
//     lane l, accumulator register r  ->  row (r&3) + 8*(r>>2) + 4*(l>>5),  column l&31.
// We compute OUT^T[feature][point] = W[feature][k] * ACT^T[k][point], so rows are output features and
// columns are sample points: lane (n = l&31, h = l>>5) ends a layer holding, for ITS point n, the 16
// features {(r&3)+8*(r>>2)+4h} of every 32-feature tile.  The contraction index of an MFMA is free to
// permute as long as A and B agree, so the NEXT layer consumes those registers directly as its B
// operand (k-slot (h, e) of k-unit u  <->  accumulator register r = (u % UPT)*E + e) and the weights
// (A operand) are pre-permuted to match when they are packed.  No LDS transpose, no cross-lane moves:
// a wave carries its 32 points through the whole MLP in registers.
#pragma once
#include <stdint.h>
#include "dfn_devguard.h"

#if defined(__HIPCC__)
#define DFN_HD __host__ __device__ inline
#else
#define DFN_HD inline
#endif

namespace dfn {

// ---- precision tiers ---------------------------------------------------------------------------
enum Tier : int { TIER_F32 = 0, TIER_BF16 = 1, TIER_F16 = 2 };
// k-slots per half-wave per k-unit: one k-unit = what one 16-byte A-fragment read feeds.
//   bf16: 8 bf16 per lane  = one v_mfma_f32_32x32x16_bf16  (K=16)
//   f16 : 8 f16 per lane   = one v_mfma_f32_32x32x16_f16   (K=16; same rate and fragment map as bf16, 10 mantissa
//                            bits instead of 7: the throughput tier whose rendered RGB stays within the accuracy clause;
//                            inference only - gradients would underflow f16's 5-bit exponent)
//   f32 : 4 f32 per lane   = four v_mfma_f32_32x32x2_f32   (K=2 each)
DFN_HD constexpr bool tier_is16(int tier) { return tier != TIER_F32; }
DFN_HD int tier_E(int tier) { return tier_is16(tier) ? 8 : 4; }
DFN_HD int tier_UPT(int tier) { return tier_is16(tier) ? 2 : 4; }        // k-units per 32-feature tile
DFN_HD int tier_elem_bytes(int tier) { return tier_is16(tier) ? 2 : 4; }

constexpr int FRAG_BYTES = 1024;            // one A fragment: 64 lanes x 16 B, lane-linear
constexpr int SLAB_FRAGS = 32;              // ring slot = 32 fragments = 32 KiB
constexpr int SLAB_BYTES = SLAB_FRAGS * FRAG_BYTES;

// feature (within a 32-tile) held by half h, accumulator register r
DFN_HD int tile_feat(int h, int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// inverse: feature f in [0,32) -> (h, r)
DFN_HD void tile_feat_inv(int f, int* h, int* r) {
    *h = (f >> 2) & 1;
    *r = (f & 3) + 4 * (f >> 3);
}
// slot index of a vector stored tile-wise: slot = 32*t + tile_feat(h, r)
// k-slot (unit u, half h, element e) of a vector consumed as B operand:
DFN_HD int kslot_to_slot(int tier, int u, int h, int e) {
    const int E = tier_E(tier), UPT = tier_UPT(tier);
    const int t = u / UPT, r = (u % UPT) * E + e;
    return 32 * t + tile_feat(h, r);
}

// ---- architecture (scripts/test_obama.sh: hidden 256, z 256, dim_signal 96, deform on) -----------
constexpr int HID = 256, ZDIM = 256, NPE = 60, NPEV = 24, NSIG = 96, NET = 42, DH = 64;

// Flat parameter buffer = decoder.state_dict() values concatenated in registration order
// (/root/reference/NeRFs/DFANeRF/decoder.py:207-251; tests/golden/g9_manifest.txt).
enum ParamId : int {
    P_DE0_W, P_DE0_B, P_DE1_W, P_DE1_B, P_DE2_W, P_DE2_B, P_DE3_W, P_DE3_B, P_DE4_W, P_DE4_B,
    P_DEO_W, P_DEO_B,
    P_DS0_W, P_DS0_B, P_DS1_W, P_DS1_B, P_DS2_W, P_DS2_B, P_DS3_W, P_DS3_B, P_DS4_W, P_DS4_B,
    P_DSO_W, P_DSO_B,
    P_DESK_W, P_DESK_B, P_DSSK_W, P_DSSK_B,
    P_FCIN_W, P_FCIN_B, P_FCINL_W, P_FCINL_B, P_FCINT_W, P_FCINT_B,
    P_FCZ_W, P_FCZ_B,
    P_BLK0_W, P_BLK0_B, P_BLK1_W, P_BLK1_B, P_BLK2_W, P_BLK2_B, P_BLK3_W, P_BLK3_B,
    P_BLK4_W, P_BLK4_B, P_BLK5_W, P_BLK5_B, P_BLK6_W, P_BLK6_B,
    P_FCZSK_W, P_FCZSK_B, P_FCPSK_W, P_FCPSK_B, P_FCPSKL_W, P_FCPSKL_B, P_FCPSKT_W, P_FCPSKT_B,
    P_SIGMA_W, P_SIGMA_B, P_FCZV_W, P_FCZV_B, P_FEATV_W, P_FEATV_B, P_FCV_W, P_FCV_B,
    P_FEATO_W, P_FEATO_B,
    P_COUNT
};

struct ParamShape { int rows, cols; };   // bias: rows = n, cols = 1

DFN_HD constexpr ParamShape param_shape(int id) {
    switch (id) {
    case P_DE0_W: case P_DS0_W: return {DH, NPE + NET};
    case P_DE1_W: case P_DE2_W: case P_DE3_W: case P_DE4_W:
    case P_DS1_W: case P_DS2_W: case P_DS3_W: case P_DS4_W: return {DH, DH};
    case P_DE0_B: case P_DE1_B: case P_DE2_B: case P_DE3_B: case P_DE4_B:
    case P_DS0_B: case P_DS1_B: case P_DS2_B: case P_DS3_B: case P_DS4_B:
    case P_DESK_B: case P_DSSK_B: return {DH, 1};
    case P_DEO_W: return {NPE, DH};
    case P_DEO_B: return {NPE, 1};
    case P_DSO_W: return {NET, DH};
    case P_DSO_B: return {NET, 1};
    case P_DESK_W: return {DH, NPE};
    case P_DSSK_W: return {DH, NET};
    case P_FCIN_W: case P_FCPSK_W: return {HID, NPE + NSIG};
    case P_FCINL_W: case P_FCPSKL_W: return {HID, NPE};
    case P_FCINT_W: case P_FCPSKT_W: return {HID, NPE + NET};
    case P_FCZ_W: case P_FCZSK_W: case P_FCZV_W: return {HID, ZDIM};
    case P_BLK0_W: case P_BLK1_W: case P_BLK2_W: case P_BLK3_W: case P_BLK4_W: case P_BLK5_W:
    case P_BLK6_W: case P_FEATV_W: return {HID, HID};
    case P_SIGMA_W: return {1, HID};
    case P_SIGMA_B: return {1, 1};
    case P_FCV_W: return {HID, NPEV};
    case P_FEATO_W: return {3, HID};
    case P_FEATO_B: return {3, 1};
    default: return {HID, 1};   // every remaining id is a 256-wide bias
    }
}

DFN_HD constexpr int param_numel(int id) { return param_shape(id).rows * param_shape(id).cols; }
// offsets of the tensors in the flat parameter vector: a compile-time table (a runtime id costs one load, not a
// 60-iteration loop over the shapes - the fold kernels index it with computed ids)
struct ParamOffsets { int v[P_COUNT + 1]; };
DFN_HD constexpr ParamOffsets make_param_offsets() {
    ParamOffsets t = {};
    int off = 0;
    for (int i = 0; i < P_COUNT; ++i) {
        t.v[i] = off;
        off += param_numel(i);
    }
    t.v[P_COUNT] = off;
    return t;
}
DFN_HD constexpr int param_offset(int id) {
    constexpr ParamOffsets T = make_param_offsets();
    return T.v[id];
}
static_assert(make_param_offsets().v[P_COUNT] == 955242, "flat decoder parameter count");
constexpr int N_DECODER_PARAMS = 955242;     // checked against param_offset(P_COUNT) at load time

// ---- input-vector slot maps -------------------------------------------------------------------------
// PE64:   slot s <-> reference PE column s (decoder.py:271-274: 6*octave + 3*is_cos + axis), s < 60;
//         slots 60..63 are zero padding.  The identity map puts PE column f in exactly the register
//         where a 64-wide GEMM leaves output feature f, so the torso's residual `deform(p) + p`
//         (decoder.py:299) is a lane-local register add.
// VIEW32: slot s <-> view-direction PE column s, s < 24; slots 24..31 zero.
// DPE64 / DSIG64 (torso, after the deformation field): slot s <-> column s (s < 60 resp. s < 42).
DFN_HD int pe_slot_to_ref(int slot) { return slot < NPE ? slot : -1; }
DFN_HD int view_slot_to_ref(int slot) { return slot < NPEV ? slot : -1; }

// ---- programs ---------------------------------------------------------------------------------------
enum Field : int { FIELD_HEAD = 0, FIELD_TORSO = 1 };

// fragment / bias-blob sizes of each field's program, per tier (filled by the planner, checked by the
// kernels with static constants)
struct ProgramInfo {
    int n_frags;        // A fragments in the packed weight stream of one MLP pass
    int n_slabs;        // ring slots consumed by one pass
    int n_bias;         // floats in the per-frame bias blob
};

}  // namespace dfn
