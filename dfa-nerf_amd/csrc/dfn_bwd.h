// dfn_bwd.h - backward of the fused decoder MLP for one wavefront = 32 sample points (training).
//
// dL/d(input) of every layer is the SAME machinery as the forward (dfn_mlp.h) run on transposed weight
// streams: G_in^T[i][n] = sum_o W[o][i] * DY^T[o][n], so A = W^T fragments (packed by dfn_plan.cpp:
// build_bwd_plan in the op order of this file), B = the pre-activation gradient the previous backward GEMM
// left in this lane's registers, masked with the forward's ReLU bits.  Every pre-activation gradient is also
// written feature-major per 32-point tile ([tile][rows][32]) for the weight-gradient GEMMs (dfn_train.hip: wgrad_kernel).
//
// Reference semantics being differentiated: decoder.py:277-349 and :109-134 (torch autograd of those ops;
// pinned by golden G8).
#pragma once
#include "dfn_mlp.h"

namespace dfn {

// rows of the feature-major gradient array dy_T
struct GradMap {
    // trunk
    static constexpr int T_DY0 = 0, T_DY4 = 4 * 256, T_G4 = 5 * 256, T_DY5 = 6 * 256, T_DYV = 9 * 256,
                         T_DSIG = 10 * 256, T_DYO = 10 * 256 + 32, T_ROWS = 10 * 256 + 64;
    static constexpr int H_TRUNK = 0, H_ROWS = T_ROWS;
    // torso deformation nets: DE0 DS0 DE1 DS1 DE2 DS2 DE3 GE3 DS3 DE4 DS4 DEO DSO GS3 (14 x 64), then the trunk
    static constexpr int S_DE0 = 0, S_DS0 = 64, S_DE1 = 128, S_DS1 = 192, S_DE2 = 256, S_DS2 = 320, S_DE3 = 384,
                         S_GE3 = 448, S_DS3 = 512, S_DE4 = 576, S_DS4 = 640, S_DEO = 704, S_DSO = 768, S_GS3 = 832,
                         S_TRUNK = 896, S_ROWS = 896 + T_ROWS;
};

struct BwdIO {
    void* dy_T;                 // tile-major [pass][rows][32]
    const unsigned* masks;      // forward ReLU bits [pass][mask_dwords][64]
    int rows;
    long pass;
    int mask_dwords;
};

// g (accumulators of a tile pair) *= ReLU mask bits (word = pair index)
DFN_DEV void apply_mask(f32x16 (&acc)[2], unsigned bits) {
#ifndef DFN_NOMASK
#pragma unroll
    for (int b = 0; b < 32; ++b) {      // bit b sign-extended to a word (v_bfe_i32), ANDed onto the value: 2 ops per value
        const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)bits, mask_pos(b), 1);
        const float x = acc[b >> 4][b & 15];          // a scalar copy: __builtin_bit_cast of a vector ELEMENT miscompiles
        acc[b >> 4][b & 15] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & keep);
    }
#endif
}
// 16-bit tiers: the same on the PACKED operand words of the pair (tiles t0, t0 + 1 of v).  Word j = 8 g + 4 h + e of the pair
// holds values 2 j (low half) and 2 j + 1 (high half), whose bits sit at j and 16 + j of the mask dword (mask_pos): shifting
// both halves of the dword left by 15 - j puts each value's bit on its half's sign, an arithmetic shift right by 15 spreads it
// over the half - v_pk_lshlrev_b16, v_pk_ashrrev_i16, v_and_b32: 3 VALU per word = 1.5 per value.  (The per-value form above
// compiles to v_and + v_cmp_ne + v_cndmask + a second v_and per VALUE: 40 % of the dX kernels' vector instructions.)  asm: so
// that it stays that way; a zeroed bf16 half = the zeroed f32 value converted.
template <int TIER, int NT>
DFN_DEV void apply_mask_packed(Vec<TIER, NT>& v, int t0, unsigned bits) {
    static_assert(tier_is16(TIER), "packed operand words");
#ifndef DFN_NOMASK
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int gh = 0; gh < 4; ++gh) {             // register 2 (t0 + g) + h of v, g = gh >> 1, h = gh & 1: words 4 gh .. 4 gh + 3
        u32x4_ q = __builtin_bit_cast(u32x4_, v.u[2 * t0 + gh]);
        unsigned w[4] = {q[0], q[1], q[2], q[3]};          // scalar copies (__builtin_bit_cast of a vector ELEMENT miscompiles)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * gh + e;
            // ONE asm per word, tied to the word: as separate statements the scheduler hoisted the sixteen masks of a pair
            // above its MFMAs (they only depend on the mask dword) and spilled 100 registers
            unsigned m;
            if (j < 15)
                asm("v_pk_lshlrev_b16 %1, %3, %2 op_sel_hi:[0,1]\n\tv_pk_ashrrev_i16 %1, 15, %1 op_sel_hi:[0,1]\n\tv_and_b32 %0, %0, %1"
                    : "+v"(w[e]), "=&v"(m) : "v"(bits), "n"(15 - j));
            else
                asm("v_pk_ashrrev_i16 %1, 15, %2 op_sel_hi:[0,1]\n\tv_and_b32 %0, %0, %1" : "+v"(w[e]), "=&v"(m) : "v"(bits));
        }
        const u32x4_ r = {w[0], w[1], w[2], w[3]};
        v.u[2 * t0 + gh] = __builtin_bit_cast(typename std::remove_reference<decltype(v.u[0])>::type, r);
    }
#endif
}
#ifndef DFN_PACKED_MASK
#define DFN_PACKED_MASK 1
#endif
DFN_DEV unsigned mask_word(const BwdIO& io, int dword, int lane) {
#ifdef DFN_EXP_CONSTMASK      // timing experiment (wrong results): no mask LOADS (a value the compiler cannot fold)
    return (unsigned)io.mask_dwords * 0x9E3779B1u + (unsigned)(dword * 64 + lane) * 0x85EBCA6Bu;
#endif
    const gchar* mb = uniform_ptr(io.masks + ((long)io.pass * io.mask_dwords + dword) * 64);
    return *(const __attribute__((address_space(1))) unsigned*)(mb + (unsigned)lane * 4u);
}

template <int TIER, int NT, class CT>
DFN_DEV void put(const BwdIO& io, int row0, const Vec<TIER, NT>& v, const CT& c) {
#ifndef DFN_NOPUT
#ifdef DFN_PUT_EIGHTH     // timing experiment (wrong results): one tile in eight is stored - the chain stays alive, 7/8 of the stores go
    store_tiles_T<TIER, NT>(io.dy_T, io.rows, io.pass, row0, v, 0, 1, c);
#elif defined(DFN_PUT_SMALL)      // timing experiment (wrong results): every workgroup writes the same 8 tiles - the stores are issued, HBM is not
    store_vec_T<TIER, NT>(io.dy_T, io.rows, io.pass & 7, row0, v, c);
#else
    store_vec_T<TIER, NT>(io.dy_T, io.rows, io.pass, row0, v, c);
#endif
#endif
}

// Pin a finished vector where the source computes it.  The torso kernel's skip-path product (fc_p_skips_torso^T x g4) is
// only consumed ~500 MFMAs later; left free, the scheduler sinks its 64 MFMAs down to that use and keeps their INPUTS alive
// instead - 64 weight fragments copied from the LDS ring to scratch memory (256 spilled VGPRs).
// f32 tier, torso kernel: the skip path's product (64 registers per lane) waits in LDS while the lower half of the trunk runs.
// The f32 dX kernels run one wave per SIMD with all 512 registers; the head kernel needs 402, the torso kernel with this vector
// alive across the four-layer loop needed 438 MORE than it has (1756 bytes of scratch per lane: 3.48 ms where the head kernel
// takes 1.18 - round 5, profiles/r05l_c4_f32_kernel_stats.csv).  The ring leaves 64 KiB of the CU's 160: 4 waves x 64 x 64 floats.
constexpr int PARK_BYTES_PER_WAVE = 64 * 64 * 4;
template <class CT> DFN_DEV void park_store(const CT& c, const Vec<TIER_F32, 4>& v) {
    lds_f32* p = (lds_f32*)(c.ring + RING_BYTES + c.wave * PARK_BYTES_PER_WAVE) + c.lane;
#pragma unroll
    for (int k = 0; k < 64; ++k) p[64 * k] = v.v[k];
}
template <class CT> DFN_DEV void park_load(const CT& c, Vec<TIER_F32, 4>& v) {
    const lds_f32* p = (const lds_f32*)(c.ring + RING_BYTES + c.wave * PARK_BYTES_PER_WAVE) + c.lane;
#pragma unroll
    for (int k = 0; k < 64; ++k) v.v[k] = p[64 * k];
}
template <int TIER, int NT> DFN_DEV void pin_vec(Vec<TIER, NT>& v) {
    if constexpr (tier_is16(TIER)) {
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) asm volatile("" : "+v"(v.u[i]));
    }
}

// DFN_PUT_SPREAD: how the dy_T stores of a layer's INPUT vector (the previous layer's finished output, alive as this
// layer's B operand anyway) are issued.  0: one burst in front of the layer, all eight waves of the workgroup at once.
// 1: spread between the MFMAs of the whole layer (PutSide: one 16-byte MX-fp8 tile store every T / NTB k-steps).
// History (bf16 recording, round 2): spreading 64 word stores per vector bought 5 % (head 225 -> 213 us); with one tile in
// eight stored the head kernel took 139 us - the other 75-85 us were the 0.69 GB of stores.  Round 3 (MX-fp8, point-major
// blocks: 8 store instructions and half the bytes per vector): head 182 us, of which the store instructions 24, the block
// amax 10, the ReLU mask application 8; without any recording 133 us (ablation builds, tools/time_dx.py).  NOT the cause
// (each ablated, +-3 %): the vmcnt wait of the slab hand-over (vmcnt(63)), the ReLU mask loads, the depth of the fragment
// prefetch (compiler-sunk ds_reads pinned by sched_barrier), the instruction cache (0.3 % misses).
#ifndef DFN_PUT_SPREAD
#define DFN_PUT_SPREAD 1
#endif
#ifndef DFN_TORSO_G4_SPREAD
#define DFN_TORSO_G4_SPREAD 1
#endif
#ifndef DFN_TORSO_DY0_SPREAD
#define DFN_TORSO_DY0_SPREAD 1
#endif
// (MX-fp8, dfn_mlp.h: the vector's NTB 16-byte tile stores are spread evenly over the consuming layer's T = OT / 2 x KU
// k-steps; the scales of its tile pairs are found in front of the layer)
template <int TIER, int NTB, int KU, int T, class CT> struct PutSide {
    const BwdIO& io;
    const Vec<TIER, NTB>& v;
    int row0, tg;
    const CT& c;
    const Q8* qs;                   // scale of tile pair p (NTB == 1: of the one tile)
    static constexpr int STRIDE = T >= NTB ? T / NTB : 1, PER = T >= NTB ? 1 : (NTB + T - 1) / T;      // k-steps per tile / tiles per k-step
    DFN_DEV void tile(int t) const {
#ifndef DFN_NOPUT
        if constexpr (TIER == TIER_BF16) {
            if (t < NTB) store_tile8<NTB>(io.dy_T, io.rows, io.pass, row0, v, t, 0, qs[t >> 1], c);
        }
#endif
    }
    DFN_DEV void operator()(int ku) const {
        if (row0 < 0) return;
        const int s = tg * KU + ku;
        if constexpr (TIER == TIER_F32) {       // one value per store instruction: the vector's 16 NTB values over the layer's T k-steps
#ifndef DFN_NOPUT
            constexpr int N = 16 * NTB, PER32 = (N + T - 1) / T;
#pragma unroll
            for (int w = 0; w < PER32; ++w)
                if (s * PER32 + w < N) store_val_T32(io.dy_T, io.rows, io.pass, row0, s * PER32 + w, v.v[s * PER32 + w], c);
#endif
        } else if constexpr (T >= NTB) {
            if (s % STRIDE == 0) tile(s / STRIDE);
        } else {
#pragma unroll
            for (int w = 0; w < PER; ++w) tile(s * PER + w);
        }
    }
};
// scales of the tile pairs of `v` + their bytes in dy_T (in front of the layer that streams the vector out)
template <int TIER, int NTB, class CT>
DFN_DEV void put_scales(const BwdIO& io, int row0, const Vec<TIER, NTB>& v, Q8 (&qs)[(NTB + 1) / 2], const CT& c) {
    if constexpr (TIER == TIER_BF16) {
        if (row0 < 0) return;
#pragma unroll
        for (int p = 0; p < (NTB + 1) / 2; ++p) {
            const int np = NTB >= 2 ? 2 : 1;
            qs[p] = q8_of_tiles<NTB>(v, 2 * p, np);
            store_scale8(io.dy_T, io.rows, io.pass, row0, 2 * p, 0, np, qs[p], c);
        }
    }
}
// DFN_PUT32_SPREAD: the same for the f32 tier (one wave per SIMD: a burst of 128 store instructions in front of a layer stops
// the wave, and with it the matrix pipe, until the memory queue has drained)
#ifndef DFN_PUT32_SPREAD
#define DFN_PUT32_SPREAD 1
#endif
template <int TIER> constexpr bool put_spread() {
    return (TIER == TIER_BF16 && DFN_PUT_SPREAD != 0) || (TIER == TIER_F32 && DFN_PUT32_SPREAD != 0);
}

// out[OT tiles] = (W^T x in) [* mask]; mask_dword0 < 0: no mask.  put_row >= 0: `in` is written to rows put_row.. of dy_T
// on the way (PutSide)
template <int TIER, int OT, int KU, int NTB, class CT>
DFN_DEV void bwd_layer(Vec<TIER, OT>& out, const Vec<TIER, NTB>& in, int mask_dword0, const BwdIO& io, int& f,
                       Fetch<TIER>& fe, Stream& s, const CT& c, int put_row = -1) {
    if constexpr (!put_spread<TIER>()) {
        if (put_row >= 0) put<TIER, NTB>(io, put_row, in, c);
        put_row = -1;
    }
    Q8 qs[(NTB + 1) / 2];
    put_scales<TIER, NTB>(io, put_row, in, qs, c);
#pragma unroll
    for (int tg = 0; tg < OT / 2; ++tg) {
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
#ifdef DFN_TIMING
        const unsigned long long q0 = __builtin_readcyclecounter();      // (s_memtime: scalar memory, costs an lgkmcnt(0) each)
#endif
        gemm_group<TIER, 2, KU, NTB>(acc, in, f, fe, s, c, NoHook{},
                                     PutSide<TIER, NTB, KU, (OT / 2) * KU, CT>{io, in, put_row, tg, c, qs});
#ifdef DFN_TIMING
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
        const unsigned long long q1 = __builtin_readcyclecounter();
#endif
        if constexpr (tier_is16(TIER) && DFN_PACKED_MASK) {
            acc_to_vec<TIER, 2, OT, false>(acc, out, 2 * tg);
            if (mask_dword0 >= 0) apply_mask_packed<TIER, OT>(out, 2 * tg, mask_word(io, mask_dword0 + tg, c.lane));
        } else {
            if (mask_dword0 >= 0) apply_mask(acc, mask_word(io, mask_dword0 + tg, c.lane));
            acc_to_vec<TIER, 2, OT, false>(acc, out, 2 * tg);
        }
#ifdef DFN_TIMING
        asm volatile("" : "+v"(out.u[4 * tg]), "+v"(out.u[4 * tg + 3]));
        const unsigned long long q2 = __builtin_readcyclecounter();
        s.t_issue += q1 - q0;        // MFMA phase of the pair (fragment waits included)
        s.t_epi += q2 - q1;          // mask + convert epilogue
#endif
    }
}
// out = (W1^T x in1 + W2^T x in2) [* mask]
template <int TIER, int OT, int KU1, int NTB1, int KU2, int NTB2, class CT>
DFN_DEV void bwd_layer2(Vec<TIER, OT>& out, const Vec<TIER, NTB1>& in1, const Vec<TIER, NTB2>& in2,
                        int mask_dword0, const BwdIO& io, int& f, Fetch<TIER>& fe, Stream& s, const CT& c,
                        int put_row1 = -1) {
    if constexpr (!put_spread<TIER>()) {
        if (put_row1 >= 0) put<TIER, NTB1>(io, put_row1, in1, c);
        put_row1 = -1;
    }
    Q8 qs[(NTB1 + 1) / 2];
    put_scales<TIER, NTB1>(io, put_row1, in1, qs, c);
#pragma unroll
    for (int tg = 0; tg < OT / 2; ++tg) {
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
        gemm_group<TIER, 2, KU1, NTB1>(acc, in1, f, fe, s, c, NoHook{},
                                       PutSide<TIER, NTB1, KU1, (OT / 2) * KU1, CT>{io, in1, put_row1, tg, c, qs});
        gemm_group<TIER, 2, KU2, NTB2>(acc, in2, f, fe, s, c);
        if constexpr (tier_is16(TIER) && DFN_PACKED_MASK) {
            acc_to_vec<TIER, 2, OT, false>(acc, out, 2 * tg);
            if (mask_dword0 >= 0) apply_mask_packed<TIER, OT>(out, 2 * tg, mask_word(io, mask_dword0 + tg, c.lane));
        } else {
            if (mask_dword0 >= 0) apply_mask(acc, mask_word(io, mask_dword0 + tg, c.lane));
            acc_to_vec<TIER, 2, OT, false>(acc, out, 2 * tg);
        }
    }
}


// ---- backward fragment counts (transposed streams) -----------------------------------------------------------
template <int TIER> struct BProg {
    using P = Prog<TIER>;
    static constexpr int UPT = P::UPT, KU_ACT = P::KU_ACT, KU_D = P::KU_D, KU_T = UPT;      // KU_T: one 32-row tile
    // trunk: BO (8 tiles x 1-tile K), BV (8 x (256 + 32)), B7 B6 B5 (8 x 256), B4..B1 (8 x 256)
    static constexpr int T_FRAGS = 8 * KU_T + 8 * (KU_ACT + KU_T) + 7 * 8 * KU_ACT;
    static constexpr int H_FRAGS = T_FRAGS;
    static constexpr int H_SLABS = (H_FRAGS + SLAB_FRAGS - 1) / SLAB_FRAGS;
    // torso tail: skip path 4 tiles x 256 (right after B5), fc_in_torso^T 4 x 256, then the deformation nets:
    // EO^T SO^T (2 x 64 each), E4 S4, E3 S3, E2 S2, E1 S1 (2 tiles x 64 each)
    static constexpr int S_FRAGS = T_FRAGS + 2 * 4 * KU_ACT + 10 * 2 * KU_D;
    static constexpr int S_SLABS = (S_FRAGS + SLAB_FRAGS - 1) / SLAB_FRAGS;
};

struct BwdIn {
    float dsigma, dpre[3];      // dL/dsigma_raw and dL/d(pre-sigmoid rgb) of this lane's point (lanes 0..31)
};

// Backward of the trunk.  On return `dy0` holds dL/d(pre-activation of the first layer) (masked), and, when
// TORSO, `gpd_skip` holds fc_p_skips_torso^T x g4 (the skip path's contribution to dL/d pd).
template <int TIER, bool TORSO, class CT>
DFN_DEV void bwd_trunk(const BwdIn& in, Vec<TIER, 8>& dy0, Vec<TIER, 4>& gpd_skip, const BwdIO& io, int g_trunk,
                       int m_trunk, int& f, Fetch<TIER>& fe, Stream& s, const CT& c) {
    using B = BProg<TIER>;
    Vec<TIER, 8> cur, nxt;
    // Every 8-tile gradient vector is written to dy_T by the layer that CONSUMES it (put_row of bwd_layer: its stores go out
    // between that layer's MFMAs, PutSide), not in a burst behind the layer that produced it.
    // feat_out^T: d(pre-rgb) [3 of a 32-row tile] -> g_h, masked with h > 0
    {
        Vec<TIER, 1> dout;
#pragma unroll
        for (int L = 0; L < 16; ++L) dout.set(L, (c.half == 0 && L < 3) ? in.dpre[L < 3 ? L : 0] : 0.f);
        put<TIER, 1>(io, g_trunk + GradMap::T_DYO, dout, c);
        bwd_layer<TIER, 8, B::KU_T, 1>(cur, dout, m_trunk + RecMap::TM_H, io, f, fe, s, c);
    }
    // [feat_view ; sigma_out]^T -> g_a7, masked with a7 > 0
    {
        Vec<TIER, 1> dsig;
#pragma unroll
        for (int L = 0; L < 16; ++L) dsig.set(L, (c.half == 0 && L == 0) ? in.dsigma : 0.f);
        put<TIER, 1>(io, g_trunk + GradMap::T_DSIG, dsig, c);
        bwd_layer2<TIER, 8, B::KU_ACT, 8, B::KU_T, 1>(nxt, cur, dsig, m_trunk + RecMap::TM_A5 + 8, io, f, fe, s, c,
                                                      g_trunk + GradMap::T_DYV);
        cur = nxt;                                                                  // dy7
    }
    // blocks[6]^T, blocks[5]^T -> dy6, dy5
    bwd_layer<TIER, 8, B::KU_ACT, 8>(nxt, cur, m_trunk + RecMap::TM_A5 + 4, io, f, fe, s, c, g_trunk + GradMap::T_DY5 + 512);
    cur = nxt;
    bwd_layer<TIER, 8, B::KU_ACT, 8>(nxt, cur, m_trunk + RecMap::TM_A5, io, f, fe, s, c, g_trunk + GradMap::T_DY5 + 256);
    cur = nxt;
    // blocks[4]^T -> g4 = dL/d a4 (post-skip, no activation)
    bwd_layer<TIER, 8, B::KU_ACT, 8>(nxt, cur, -1, io, f, fe, s, c, g_trunk + GradMap::T_DY5);
    cur = nxt;
#if DFN_TORSO_G4_SPREAD
    if constexpr (TORSO) bwd_layer<TIER, 4, B::KU_ACT, 8>(gpd_skip, cur, -1, io, f, fe, s, c, g_trunk + GradMap::T_G4);   // fc_p_skips_torso^T
    else put<TIER, 8>(io, g_trunk + GradMap::T_G4, cur, c);           // (masked in place next: no consumer layer to ride on)
#else
    put<TIER, 8>(io, g_trunk + GradMap::T_G4, cur, c);                // (masked in place below)
    if constexpr (TORSO) bwd_layer<TIER, 4, B::KU_ACT, 8>(gpd_skip, cur, -1, io, f, fe, s, c);   // fc_p_skips_torso^T
#endif
    if constexpr (TORSO) pin_vec(gpd_skip);
    if constexpr (TORSO && TIER == TIER_F32) park_store(c, gpd_skip);      // (read back by bwd_torso)
    // dy4 = g4 * [y4 > 0]
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned bits = mask_word(io, m_trunk + RecMap::TM_A4R + w, c.lane);
        if constexpr (tier_is16(TIER)) {
            apply_mask_packed<TIER, 8>(cur, 2 * w, bits);
        } else {
#pragma unroll
            for (int b = 0; b < 32; ++b)
                if (!((bits >> mask_pos(b)) & 1u)) cur.set(32 * w + b, 0.f);
        }
    }
    // blocks[3..0]^T -> dy3 .. dy0; layer l writes its input: dy4 (l = 3), then dy3 .. dy1
    if constexpr (TIER == TIER_F32) {
        // f32 tier: a runtime loop whose iterations all restart the fragment index at the same compile-time value - a
        // 256 x 256 layer is a whole number of slabs and of fetch-ring turns, so the slab phase and the ring slot repeat
        // and every fragment position of the body stays static (training step 14.3 -> 11.8 ms).  The bf16 kernels
        // are faster fully unrolled (torso 299 against 416 us: the loop-carried vectors make the allocator spill).
        constexpr int LAYER_FRAGS = 8 * B::KU_ACT;
        static_assert(LAYER_FRAGS % SLAB_FRAGS == 0 && LAYER_FRAGS % PF_DEPTH == 0, "layer = whole slabs and ring turns");
        const int f0 = f;
#pragma nounroll
        for (int l = 3; l >= 0; --l) {
            int fl = f0;
            bwd_layer<TIER, 8, B::KU_ACT, 8>(nxt, cur, m_trunk + RecMap::TM_A0 + 4 * l, io, fl, fe, s, c,
                                             g_trunk + (l == 3 ? GradMap::T_DY4 : GradMap::T_DY0 + 256 * (l + 1)));
            cur = nxt;
        }
        f = f0 + 4 * LAYER_FRAGS;
    } else {
#pragma unroll
        for (int l = 3; l >= 0; --l) {
            bwd_layer<TIER, 8, B::KU_ACT, 8>(nxt, cur, m_trunk + RecMap::TM_A0 + 4 * l, io, f, fe, s, c,
                                             g_trunk + (l == 3 ? GradMap::T_DY4 : GradMap::T_DY0 + 256 * (l + 1)));
            cur = nxt;
        }
    }
    if constexpr (!TORSO || !DFN_TORSO_DY0_SPREAD) put<TIER, 8>(io, g_trunk + GradMap::T_DY0, cur, c);      // torso: the caller's next layer writes dy0
    dy0 = cur;
}

template <int TIER, class CT>
DFN_DEV void bwd_head(const BwdIn& in, const BwdIO& io, Stream& s, const CT& c) {
    int f = 0;
    Fetch<TIER> fe;
    fe.prime(s, c);
    Vec<TIER, 8> dy0;
    Vec<TIER, 4> unused;
    bwd_trunk<TIER, false>(in, dy0, unused, io, GradMap::H_TRUNK, RecMap::H_MTRUNK, f, fe, s, c);
}

template <int TIER, class CT>
DFN_DEV void bwd_torso(const BwdIn& in, const BwdIO& io, Stream& s, const CT& c) {
    using B = BProg<TIER>;
    int f = 0;
    Fetch<TIER> fe;
    fe.prime(s, c);
    Vec<TIER, 8> dy0;
    Vec<TIER, 4> gsk;
    bwd_trunk<TIER, true>(in, dy0, gsk, io, GradMap::S_TRUNK, RecMap::S_MTRUNK, f, fe, s, c);
    // dL/d pd = fc_in_torso^T x dy0 + (skip path)
    Vec<TIER, 4> gpd;
    bwd_layer<TIER, 4, B::KU_ACT, 8>(gpd, dy0, -1, io, f, fe, s, c, DFN_TORSO_DY0_SPREAD ? GradMap::S_TRUNK + GradMap::T_DY0 : -1);
    if constexpr (TIER == TIER_F32) park_load(c, gsk);
#pragma unroll
    for (int L = 0; L < 64; ++L) gpd.set(L, gpd.get(L) + gsk.get(L));
    // pd = [out_embed(.) + pe ; out_signal(.) + signal]: the GEMM outputs get g_pd unchanged
    Vec<TIER, 2> ge, gs, gn;
#pragma unroll
    for (int L = 0; L < 32; ++L) {
        ge.set(L, gpd.get(L));
        gs.set(L, gpd.get(32 + L));
    }
    put<TIER, 2>(io, GradMap::S_DEO, ge, c);
    put<TIER, 2>(io, GradMap::S_DSO, gs, c);
    // out_embed^T / out_signal^T -> dE4, dS4 (masked with ve4 / vs4 > 0)
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, ge, RecMap::S_MD0 + 8, io, f, fe, s, c);  ge = gn;
    put<TIER, 2>(io, GradMap::S_DE4, ge, c);
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, gs, RecMap::S_MD0 + 9, io, f, fe, s, c);  gs = gn;
    put<TIER, 2>(io, GradMap::S_DS4, gs, c);
    // blocks_*[4]^T -> gradient of the post-skip vectors ve3 / vs3; then the pre-skip ReLU masks
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, ge, -1, io, f, fe, s, c);  ge = gn;
    put<TIER, 2>(io, GradMap::S_GE3, ge, c);
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, gs, -1, io, f, fe, s, c);  gs = gn;
    put<TIER, 2>(io, GradMap::S_GS3, gs, c);
    {
        const unsigned be = mask_word(io, RecMap::S_MD0 + 6, c.lane), bs = mask_word(io, RecMap::S_MD0 + 7, c.lane);
        if constexpr (tier_is16(TIER)) {
            apply_mask_packed<TIER, 2>(ge, 0, be);
            apply_mask_packed<TIER, 2>(gs, 0, bs);
        } else {
#pragma unroll
            for (int b = 0; b < 32; ++b) {
                if (!((be >> mask_pos(b)) & 1u)) ge.set(b, 0.f);
                if (!((bs >> mask_pos(b)) & 1u)) gs.set(b, 0.f);
            }
        }
    }
    put<TIER, 2>(io, GradMap::S_DE3, ge, c);
    put<TIER, 2>(io, GradMap::S_DS3, gs, c);
    // blocks_*[3]^T, [2]^T, [1]^T
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, ge, RecMap::S_MD0 + 4, io, f, fe, s, c);  ge = gn;
    put<TIER, 2>(io, GradMap::S_DE2, ge, c);
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, gs, RecMap::S_MD0 + 5, io, f, fe, s, c);  gs = gn;
    put<TIER, 2>(io, GradMap::S_DS2, gs, c);
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, ge, RecMap::S_MD0 + 2, io, f, fe, s, c);  ge = gn;
    put<TIER, 2>(io, GradMap::S_DE1, ge, c);
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, gs, RecMap::S_MD0 + 3, io, f, fe, s, c);  gs = gn;
    put<TIER, 2>(io, GradMap::S_DS1, gs, c);
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, ge, RecMap::S_MD0 + 0, io, f, fe, s, c);  ge = gn;
    put<TIER, 2>(io, GradMap::S_DE0, ge, c);
    bwd_layer<TIER, 2, B::KU_D, 2>(gn, gs, RecMap::S_MD0 + 1, io, f, fe, s, c);  gs = gn;
    put<TIER, 2>(io, GradMap::S_DS0, gs, c);
}

}  // namespace dfn
