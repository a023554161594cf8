// dfn_mlp.h - the fused decoder MLP for one wavefront = 32 sample points, gfx950 only.
//
// Reference semantics: /root/reference/NeRFs/DFANeRF/decoder.py:277-349 (Decoder.forward, head and
// torso branches) and :109-134 (DeformationField_ori.forward).
//
// Data flow (see dfn_layout.h for the fragment maps):
//   * weights arrive as a stream of 1 KiB A-fragments, packed on the host side of the C-ABI in exactly
//     the order this file consumes them (dfn_plan.cpp mirrors the op order below), moved L2 -> LDS by
//     LDS-DMA (global_load_lds_dwordx4) into a 3-slot ring of 32-fragment slabs shared by all waves of
//     the workgroup;
//   * activations never leave registers: the accumulator registers of layer l ARE the B operand of
//     layer l+1 (after bias/ReLU and, in the bf16 tier, a pairwise f32->bf16 pack);
//   * GEMMs run tile-major: two 32-feature output tiles (2 x 16 accumulator registers) are finished over
//     the whole K before the next pair starts, so only 32 accumulator registers are live and the
//     bias/ReLU/pack epilogue of one pair overlaps the MFMAs of the next;
//   * per-frame constant inputs (audio/expression signal, pose signal, z_shape, z_app) are folded into
//     bias vectors by dfn_fold_kernel and enter as accumulator initial values.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "dfn_layout.h"

namespace dfn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DFN_DEV __device__ __forceinline__
// 1: in the bf16 inference render kernels the fragment reads and the LDS-DMA are inline asm with counted lgkmcnt waits
// (-2.2 % on C2; LABNOTES.md 4.5).  The build checks the generated ISA with tools/check_inflight.py: no instruction may
// touch an asm read's destination before the wait that retires it.  0 = everything through the compiler.
#ifndef DFN_ASM_FETCH
#define DFN_ASM_FETCH 1
#endif
// explicit LDS address space: every LDS access must be a ds_* instruction (a flat access would wait on
// vmcnt and drain the weight prefetch)
#define DFN_LDS __attribute__((address_space(3)))
typedef DFN_LDS char lds_char;
typedef const __attribute__((address_space(1))) char gchar_c;
typedef DFN_LDS float lds_f32;
typedef DFN_LDS f32x4 lds_f32x4;
typedef DFN_LDS u32x4 lds_u32x4;

template <int TIER> struct TierCfg;
template <> struct TierCfg<TIER_BF16> {
    static constexpr int E = 8, UPT = 2, WAVES = 8, THREADS = 512, LOADS_PER_SLAB = 4;
};
template <> struct TierCfg<TIER_F16> : TierCfg<TIER_BF16> {};
template <> struct TierCfg<TIER_F32> {
    static constexpr int E = 4, UPT = 4, WAVES = 4, THREADS = 256, LOADS_PER_SLAB = 8;
};

constexpr int RING_SLOTS = 3;
constexpr int RING_BYTES = RING_SLOTS * SLAB_BYTES;

// ---- a vector of NT 32-feature tiles held as MFMA B operand ------------------------------------------
// local slot L (0..16*NT-1) of this lane = tile L>>4, accumulator register L&15.
template <int TIER, int NT> struct Vec;
template <int NT> struct Vec<TIER_BF16, NT> {
    bf16x8 u[2 * NT];
    DFN_DEV void set(int L, float x) { u[L >> 3][L & 7] = (__bf16)x; }
    DFN_DEV float get(int L) const { return (float)u[L >> 3][L & 7]; }
};
template <int NT> struct Vec<TIER_F16, NT> {
    f16x8 u[2 * NT];
    DFN_DEV void set(int L, float x) { u[L >> 3][L & 7] = (_Float16)x; }
    DFN_DEV float get(int L) const { return (float)u[L >> 3][L & 7]; }
};
template <int NT> struct Vec<TIER_F32, NT> {
    float v[16 * NT];
    DFN_DEV void set(int L, float x) { v[L] = x; }
    DFN_DEV float get(int L) const { return v[L]; }
};

// ---- weight stream -------------------------------------------------------------------------------------
// All state is wave-uniform.  The stream of one MLP pass is nslab[field] slabs; pass p of a workgroup runs
// field (sched >> p) & 1 (bit mask set by the kernel: 0 = always field 0).
struct Stream {
    // packed blobs (global) and slabs per pass of field 0 / field 1.  Scalars on purpose: as two-element arrays they
    // were indexed by the runtime field bit, which kept the WHOLE struct in scratch memory (88 bytes per lane, every
    // cursor update a scratch store + load) in the two-field kernels
    const char* base0;
    const char* base1;
    int nslab0, nslab1;
    unsigned sched;          // field of pass p = bit p (passes beyond bit 31: field 0)
    // prefetch cursor
    const char* pf_ptr;
    int pf_left;             // slabs left in the pass the cursor is in
    int pass;                // index of the pass the consumer starts next
    const char* next_ptr;    // stream of the pass after the one being consumed
    int next_left;
    unsigned pf_slot;        // ring slot the next prefetch lands in
    unsigned rd_off;         // LDS byte offset (within the ring) of the slab being consumed
    unsigned rd_slot;
    unsigned rd_vaddr;       // per lane: LDS address of this lane's 16 bytes of fragment 0 of the slab being consumed
    int pf_owed;             // pieces of the slab two ahead still to be issued
#ifdef DFN_TIMING
    unsigned long long t_wait, t_bar, t_issue, t_epi;
#endif
};

// One LDS-DMA piece: this wave's k-th 1 KiB fragment of the slab the prefetch cursor points at.
template <int TIER, bool ASM = false>
DFN_DEV void stream_issue_piece(const Stream& s, lds_char* ring, int wave, int lane, int k) {
    using C = TierCfg<TIER>;
    const int f = k * C::WAVES + wave;
    if constexpr (ASM) {
        // asm on purpose: with the builtin ("LDS DMA" to the compiler) in the function, hipcc treats lgkmcnt as out of
        // order and puts s_waitcnt lgkmcnt(0) in front of every MFMA whose operands came from LDS
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)ring + s.pf_slot * SLAB_BYTES +
                                                            (unsigned)f * FRAG_BYTES);
        // wave-uniform base in an SGPR pair + 32-bit lane offset (the "saddr" form): no 64-bit VALU address arithmetic per
        // piece (interleaved A/B on one box, f16 C2: 35.10 -> 34.42 ms)
        const gchar_c* sb = (const gchar_c*)(s.pf_ptr + f * FRAG_BYTES);
        const unsigned voff = (unsigned)lane * 16u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(dst) : "memory", "m0");
    } else {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(s.pf_ptr + (size_t)lane * 16 + f * FRAG_BYTES),
            (DFN_LDS void*)(ring + s.pf_slot * SLAB_BYTES + f * FRAG_BYTES), 16, 0, 0);
    }
}
// advance the prefetch cursor to the next slab; at the end of a pass it jumps to the stream of the next pass
// (next_ptr / next_left, set when the CONSUMER starts a pass: the cursor is only two slabs ahead of it).
// Two cheap selects on purpose: anything heavier is turned into a branch per slab, which cuts the MLP into
// 32-MFMA basic blocks that the register allocator spills around (the two-field kernel wrote 21 GB of scratch
// per frame that way).
DFN_DEV void stream_cursor_next(Stream& s) {
    s.pf_slot = (s.pf_slot + 1 == RING_SLOTS) ? 0u : s.pf_slot + 1;
    const int left = s.pf_left - 1;
    const bool wrap = left == 0;
    s.pf_ptr = wrap ? s.next_ptr : s.pf_ptr + SLAB_BYTES;
    s.pf_left = wrap ? s.next_left : left;
}
// the consumer starts pass s.pass: publish where the cursor goes after the end of this pass
DFN_DEV void stream_pass_begin(Stream& s) {
    const int np = s.pass + 1;
    const int f = (np < 32) ? ((s.sched >> np) & 1u) : 0;
    s.next_ptr = f ? s.base1 : s.base0;
    s.next_left = f ? s.nslab1 : s.nslab0;
    s.pass = np;
}
template <int TIER, bool ASM = false>
DFN_DEV void stream_issue(Stream& s, lds_char* ring, int wave, int lane) {
#pragma unroll
    for (int k = 0; k < TierCfg<TIER>::LOADS_PER_SLAB; ++k) stream_issue_piece<TIER, ASM>(s, ring, wave, lane, k);
    stream_cursor_next(s);
}
// A pass whose last slab is partial stops reading fragments before every piece of the slab two ahead went out:
// the next pass issues the rest before its first fragment (pass boundaries only).
template <int TIER, bool ASM = false>
DFN_DEV void stream_flush(Stream& s, lds_char* ring, int wave, int lane) {
    constexpr int L = TierCfg<TIER>::LOADS_PER_SLAB;
    if (s.pf_owed > 0) {
        for (int k = L - s.pf_owed; k < L; ++k) stream_issue_piece<TIER, ASM>(s, ring, wave, lane, k);
        stream_cursor_next(s);
        s.pf_owed = 0;
    }
}

template <int TIER, bool ASM = false>
DFN_DEV void stream_begin(Stream& s, lds_char* ring, int wave, int lane) {
    s.pf_ptr = s.base0;
    s.pf_left = s.nslab0;
    s.pass = 0;
    s.next_ptr = s.base0;
    s.next_left = s.nslab0;
    s.pf_slot = 0;
    s.rd_slot = RING_SLOTS - 1;     // the first slab_advance moves it to slot 0
    s.rd_off = 0;
    s.rd_vaddr = 0;
    s.pf_owed = 0;
    stream_issue<TIER, ASM>(s, ring, wave, lane);     // slab 0
    stream_issue<TIER, ASM>(s, ring, wave, lane);     // slab 1
}

// Called by every wave right before it reads the first fragment of the next slab (slab g).  The DMA of slab g+2
// (into the slot slab g-1 releases here) is NOT issued here: a burst of 32 loads per CU right behind the barrier
// keeps the vector-memory issue path busy for ~500 cycles during which no wave can issue an MFMA (in-order
// issue).  Its pieces are spread over the fragment reads of slab g instead (Fetch::load).
template <int TIER>
DFN_DEV void slab_advance(Stream& s, lds_char* ring, int wave, int lane) {
    using C = TierCfg<TIER>;
    // my share of slab g has landed (only slab g+1's pieces are younger), and my fragment reads of slab g-1 have
    // returned (its slot is refilled during slab g; the compiler knows nothing about that hazard)
#ifdef DFN_TIMING
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
#ifdef DFN_EXP_VMCNT      // timing experiment (WRONG results: the weight slab may not have landed): what do the stores in the vmcnt queue cost?
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DFN_EXP_VMCNT) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::LOADS_PER_SLAB) : "memory");
#endif
#ifdef DFN_TIMING
    const unsigned long long t1 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();   // everyone's share landed; everyone is done with slab g-1
    asm volatile("" ::: "memory");
#ifdef DFN_TIMING
    const unsigned long long t2 = __builtin_readcyclecounter();
    s.t_wait += t1 - t0; s.t_bar += t2 - t1;
#endif
    s.rd_slot = (s.rd_slot + 1 == RING_SLOTS) ? 0u : s.rd_slot + 1;
    s.rd_off = s.rd_slot * SLAB_BYTES;
    s.rd_vaddr = (unsigned)(unsigned long)ring + (unsigned)lane * 16u + s.rd_off;
    s.pf_owed += C::LOADS_PER_SLAB;
}

// ---- GEMM pieces ------------------------------------------------------------------------------------------
// Training-mode recorder: the forward pass leaves what the backward kernels need (all null otherwise).
//   act_T : feature-major activations per 32-point tile, [tile][rows][32] (bf16 / f32 by tier): the inputs of
//           every GEMM, read by the weight-gradient GEMMs (contraction over the sample points);
//   masks : ReLU masks as bits, [pass][dword][64 lanes], read by the dX chain.
struct Rec {
    void* act_T;    // tile-major: [pass][rows][32 points]; every store of a pass is base + compile-time offset
    unsigned* masks;
    int rows;       // rows per tile of act_T for this field
    long pass;      // tile (pass) index of this field = point / 32
    int mask_dwords;
};

// per-wave constants threaded through the ops; REC = training recorder on; ASMF = fragment reads / LDS-DMA as inline asm
// with counted waits (only kernels whose registers do not spill: an in-flight asm destination must never be copied)
// ASMD = only the LDS-DMA as inline asm (fragment reads stay with the compiler): for the kernels that also STORE a lot
// (recorder, dX chain).  With the builtin in the function hipcc drains the whole vector-memory queue - every store
// still in flight - at each use of an ordinary load (a ReLU mask word, a spill reload); without it the waits are counted.
// PIPE = software-pipelined layers (layer_pipe below): the convert / ReLU epilogue of a tile pair and the bias reads of the
// next one are issued between the MFMAs of the pair in between, on a second set of accumulators (+32 registers).
template <bool REC, bool ASMF = false, bool ASMD = false, bool PIPE = false, bool ACT4 = false> struct CtxT {
    static constexpr bool rec_on = REC;
    static constexpr bool act_fp4 = ACT4;            // the recorder writes act_T as MX-fp4 (dfn_mlp.h "Round 4")
#ifndef DFN_TRAIN_ASMF      // 1: the asm fragment fetch in the recording (training forward) kernels too: 413 -> 402 us; tools/check_inflight.py
                            // (run by build.sh on their ISA as well) proves no spill or copy touches an in-flight destination
#define DFN_TRAIN_ASMF 1
#endif
    static constexpr bool asm_fetch = (ASMF || (REC && DFN_TRAIN_ASMF != 0)) && (!REC || DFN_TRAIN_ASMF != 0) && (DFN_ASM_FETCH != 0);
    static constexpr bool asm_dma = asm_fetch || ASMD;
    static constexpr bool pipe = PIPE && !REC;
    lds_char* ring;
    int wave, lane, half;
    Rec rec;
};
typedef CtxT<false> Ctx;

template <int TIER> struct ActT;
template <> struct ActT<TIER_BF16> { typedef __bf16 type; };
template <> struct ActT<TIER_F16> { typedef _Float16 type; };
template <> struct ActT<TIER_F32> { typedef float type; };

// store a B-operand vector feature-major into a tile-major array [tile][rows][32]: element (row0 + feature, n).
// All offsets from the tile base are compile-time constants.
// a pointer the caller knows to be wave-uniform, pinned to an SGPR pair
// (returned as a GLOBAL address-space pointer: through an integer the compiler would fall back to flat_store)
typedef __attribute__((address_space(1))) char gchar;
DFN_DEV gchar* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gchar*)(((unsigned long)hi << 32) | lo);
}
// A global store in its "saddr" form, written by hand: wave-uniform base in an SGPR pair + 32-bit lane offset + immediate.  From
// `base + lane_offset + constant` hipcc builds a 64-bit vector address (v_lshl_add_u64) and emits the `off` form: two address
// registers per lane through the memory pipeline per store instead of one.  The recorders' stores are what the f32 training kernels
// lose their matrix pipe to (one wave per SIMD): forward 2625 -> 2584 us, dX (torso) 1284 -> 1237 us, the step 7.97 -> 7.83 ms
// (profiles/r06x_*).  `off` = a compile-time byte offset >= 0 (what exceeds the 12-bit immediate goes to the base: scalar ALU).
// No "memory" clobber: nothing in these kernels reads the recorded arrays back.  hipcc's hazard recogniser does not look into
// inline asm: the 16-byte form carries its own two wait states (gfx940 family: a store of more than 8 bytes followed by a write of
// its data registers; with ONE the MX-fp8 tiles came out wrong - 13 of 14 bf16 training tests failed, tools/r06y_run.sh).
#define DFN_GSTORE(INSN, TAIL, ubase, voff, off, x)                                                                            \
    do {                                                                                                                       \
        gchar* ub_ = (ubase) + ((off) & ~4095);                                                                                \
        asm volatile(INSN " %0, %1, %2 offset:%3" TAIL ::"v"(voff), "v"(x), "s"(ub_), "n"((off) & 4095));                      \
    } while (0)
#define DFN_GSTORE_B32_NT(ubase, voff, off, x) DFN_GSTORE("global_store_dword", " nt", ubase, voff, off, x)
#define DFN_GSTORE_B32(ubase, voff, off, x) DFN_GSTORE("global_store_dword", "", ubase, voff, off, x)
#define DFN_GSTORE_B64_NT(ubase, voff, off, x) DFN_GSTORE("global_store_dwordx2", " nt", ubase, voff, off, x)
#define DFN_GSTORE_B128_NT(ubase, voff, off, x) DFN_GSTORE("global_store_dwordx4", " nt\n\ts_nop 1", ubase, voff, off, x)

// ---- MX-fp8 recording (the 16-bit TRAINING tier) -----------------------------------------------------------------------
// What the forward records for the backward (act_T) and what the dX chain leaves for the weight-gradient GEMMs (dy_T) is
// written ONCE and read ONCE, 3 GB each way per 2048-ray step in bf16: the training step is bound by that traffic, not by
// its MFMAs.  Both arrays are therefore stored as OCP fp8 (e4m3) with one power-of-two scale (E8M0) per block of
// (tile pair = 64 features) x (32 points) - the MX block format gfx950's v_mfma_scale_f32_32x32x64_f8f6f4 consumes
// directly (dfn_wgrad_bf16.hip: contraction over the points, the K block of 32 IS one 32-point tile): half the bytes, and
// the rounding (2^-4 relative, independent per element) averages out over the 131,072 points a weight gradient sums:
// measured 0.1 % (median) / 1.6 % (worst tensor) of the bf16 step's own gradient, invisible next to the 1.7 % the bf16
// operands cost against the f32 oracle (tools/diag_fp8_record.py).  Power-of-two scales make the result independent of the
// block size as long as nothing under- or overflows.
// Layout per 32-point tile: one 1-KiB block per 32-row (feature) block, then REC8_SCALE_BYTES of scales, one per block.
// Inside a block the bytes are POINT-major, [point n][half h][accumulator register r] = feature tile_feat(h, r) of point n:
// exactly the 16 bytes lane (n, h) holds of the block after the conversion, so a block is written by ONE 16-byte store
// per lane (1 KiB contiguous per instruction) with no cross-lane traffic.  The weight-gradient GEMMs need the transpose
// (lane = feature row, registers = points): they get it from the LDS transpose read ds_read_b64_tr_b8 (dfn_wgrad_bf16.hip;
// tools/tr8_probe.hip pins its semantics).  Round 3's first version transposed 4 x 4 bytes across lane quads in the
// producers (two DPP moves + two v_perm_b32 per dword, four dword stores per block): the transposes cost 13 us and the
// store instructions most of 56 us of the 430-us training forward.
#ifndef DFN_DPP_ASM
#define DFN_DPP_ASM 1
#endif
constexpr int REC8_SCALE_BYTES = 128;        // >= rows / 32 of every array (torso dy_T: 110)
DFN_HD constexpr long rec8_tile_bytes(int rows) { return (long)rows * 32 + REC8_SCALE_BYTES; }
// Round 4: the recorded ACTIVATIONS (act_T: the GEMM inputs, half of the recorded bytes) are MX-fp4 - e2m1 values (0 .5 1 1.5 2 3 4 6
// and their negatives) under the same E8M0 scale per (tile pair = 64 features) x (32 points), the narrowest operand format of
// v_mfma_scale_f32_32x32x64_f8f6f4 (B operand: blgp 4; the pre-activation gradients dy_T stay e4m3: their blocks span too
// many binades - tools/diag_mx_narrow.py, profiles/r04_mx_narrow_emulation.txt: dy in e2m3 fails the gradient gates, act in
// e2m1 holds them: worst tensor 7.5 % of 15 %, gradient norms unchanged, the 200-step loss curve 0.05 % off the f32 tier's).
// A weight gradient sums 131,072 points: the rounding of a 1-mantissa-bit value (up to 25 %, independent per element)
// averages out like the e4m3 one did.  Layout per 32-point tile: one 512-BYTE block per 32-row block - [point n][half h][8
// bytes = 16 nibbles, nibble r = accumulator register r = feature tile_feat(h, r)], a lane's two dwords - then the scale
// bytes.  The consumer transposes with ds_read_b64_tr_b4 (tools/fp4_probe.hip pins the conversion, the operand's K order
// and the transpose on the hardware).  Run-time opt-out (round 5): DFN_TRAIN_ACT_E4M3 in the tier argument of dfn_train_fwd* selects
// the kernels whose recorder writes e4m3 activations (render_kernel<.., ACT4 = false>, dfn_render_bf16e.hip): A/B runs of the two
// formats on real data in one process (dfanerf.training.TrainBuffers(act_format=...), --hip_train_act, DFN_TRAIN_ACT).
// WHICH recorder writes MX-fp4: the fused training step's (render_kernel<.., TRAIN>: 131,072+ points per weight gradient).  The
// decoder-on-points recorder (decoder_kernel<.., REC>: Decoder.forward under autograd, any number of points - a reference-shaped loop,
// tests with a few thousand points) keeps e4m3: with 4,096 points the e2m1 rounding no longer averages out (whole-tensor
// error 9.5 % where the gate is 8 %).  The format is a property of the recording context (CtxT<..>::act_fp4) and an argument of the
// weight-gradient entry points (dfn_weight_bias_grad_fmt).
DFN_HD constexpr int act_row_bytes(bool fp4) { return fp4 ? 16 : 32; }          // bytes of one feature row of a 32-point tile
DFN_HD constexpr long act_tile_bytes(int rows, bool fp4) { return (long)rows * act_row_bytes(fp4) + REC8_SCALE_BYTES; }
struct Q8 {
    float scale;        // power of two: stored value = x / scale
    unsigned e8;        // its biased exponent (E8M0)
};
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_q __attribute__((ext_vector_type(2)));
// scale of tiles [t0, t0 + n) of v: amax over the wave (all 32 points), wave-uniform
// NONNEG: the values are ReLU outputs (no sign bits to clear)
// FP4: the scale of an e2m1 block: amax / scale in (3, 6] (6 = the format's largest value: nothing saturates)
template <int NT, bool NONNEG = false, bool FP4 = false> DFN_DEV Q8 q8_of_tiles(const Vec<TIER_BF16, NT>& v, int t0, int n) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    // a TREE of packed maxima (four independent chains, then their maximum): as one chain the 15 dependent v_pk_max_u16 of a
    // tile pair sit in front of everything the wave issues next (in-order issue), MFMAs included
    u16x2 m4[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t >= t0 && t < t0 + n)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4_ q = __builtin_bit_cast(u32x4_, v.u[2 * t + h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {    // |x| of a bf16 orders like its bit pattern: packed unsigned 16-bit max
                    const unsigned a = NONNEG ? q[e] : q[e] & 0x7fff7fffu;      // (a scalar: __builtin_bit_cast of a vector ELEMENT miscompiles)
                    m4[e] = __builtin_elementwise_max(m4[e], __builtin_bit_cast(u16x2, a));
                }
            }
    const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(m4[0], m4[1]), __builtin_elementwise_max(m4[2], m4[3]));
    unsigned x = max((unsigned)m[0], (unsigned)m[1]);
    // row maxima by DPP (quad swaps, half-row mirror, row mirror), then the four rows by readlane: an SGPR.  As asm: the
    // update_dpp builtin compiles to v_mov + v_mov_dpp + v_max per step (12 VALU), v_max_u32_dpp with the DPP on its own first
    // operand is one; dst = src0 = src1, so a lane whose source lane is off keeps its value.  s_nop 1: the two wait states a
    // DPP read needs behind the VALU write of its operand (the compiler does not see into the asm).
#if !DFN_DPP_ASM          // (the dX kernels: the asm form costs them 18 more spilled registers and is not faster there)
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xf, 0xf, false));
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xf, 0xf, false));
#else
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0"
        : "+v"(x));
#endif
    const unsigned a = max(max((unsigned)__builtin_amdgcn_readlane((int)x, 0), (unsigned)__builtin_amdgcn_readlane((int)x, 16)),
                           max((unsigned)__builtin_amdgcn_readlane((int)x, 32), (unsigned)__builtin_amdgcn_readlane((int)x, 48)));
    // amax in [2^(E-127), 2^(E-126)); scale = 2^(E-127-7): |x| / scale < 256 (e4m3 holds 448; no saturation mode needed)
    const unsigned E = a >> 7;                                   // bf16: sign(1) exponent(8) mantissa(7)
    Q8 q;
#ifdef DFN_REC8_NOAMAX        // timing experiment (wrong results): a fixed scale, no amax
    q.e8 = 120u;
    q.scale = __builtin_bit_cast(float, q.e8 << 23);
    return q;
#endif
    if constexpr (FP4) {
        // amax = 2^(E-127) (1 + m / 128): m <= 64 -> scale 2^(E-129), amax / scale = 4 (1 + m / 128) in [4, 6]; else 2^(E-128): (3, 4)
        const unsigned e = E + ((a & 0x7fu) > 64u ? 1u : 0u);
        q.e8 = e > 3u ? e - 2u : 1u;
    } else {
        q.e8 = E > 8u ? E - 7u : 1u;
    }
    q.scale = __builtin_bit_cast(float, q.e8 << 23);
    return q;
}
// tile t of v -> 16 e2m1 nibbles = ONE 8-byte store at [point][half] of the 512-byte block (MX-fp4 activations, above)
template <int NT, class CT>
DFN_DEV void store_tile4(void* arr, int rows, long tile, int row0, const Vec<TIER_BF16, NT>& v, int t, int t_first, const Q8& q,
                         const CT& c) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ out;
#pragma unroll
    for (int k = 0; k < 2; ++k) {            // registers 8 k .. 8 k + 7 -> dword k, register r in nibble r & 7
        const u32x4_ w = __builtin_bit_cast(u32x4_, v.u[2 * t + k]);
        const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];      // scalar copies (bit_cast of a vector ELEMENT miscompiles)
        unsigned o = 0;
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, __builtin_bit_cast(bf16x2_q, w0), q.scale, 0);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, __builtin_bit_cast(bf16x2_q, w1), q.scale, 1);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, __builtin_bit_cast(bf16x2_q, w2), q.scale, 2);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, __builtin_bit_cast(bf16x2_q, w3), q.scale, 3);
        out[k] = o;
    }
#ifdef DFN_REC8_NOSTORE       // timing experiment (wrong results): everything but the store instruction
    asm volatile("" ::"v"(out));
    return;
#endif
    gchar* ubase = uniform_ptr((char*)arr + tile * act_tile_bytes(rows, true) + (long)row0 * 16);
    const unsigned voff = (unsigned)((c.lane & 31) * 16 + c.half * 8);
    DFN_GSTORE_B64_NT(ubase, voff, (t - t_first) * 512, out);
}
// act_T tile store / scale store in the build's activation format
template <int NT, class CT>
DFN_DEV void store_tile_act(void* arr, int rows, long tile, int row0, const Vec<TIER_BF16, NT>& v, int t, int t_first, const Q8& q,
                            const CT& c);
template <class CT>
DFN_DEV void store_scale_act(void* arr, int rows, long tile, int row0, int t0, int t_first, int n, const Q8& q, const CT& c) {
    gchar* sb = uniform_ptr((char*)arr + tile * act_tile_bytes(rows, CT::act_fp4) + (long)rows * act_row_bytes(CT::act_fp4) + (row0 >> 5) + (t0 - t_first));
    if (c.lane < n) *(__attribute__((address_space(1))) unsigned char*)(sb + c.lane) = (unsigned char)q.e8;
}
// ONE store instruction: tile t of v (this lane's 16 features of the 32-row block, one point) -> 16 fp8 bytes at
// [point][half] of the block; 8 conversions, no cross-lane traffic.
template <int NT, class CT>
DFN_DEV void store_tile8(void* arr, int rows, long tile, int row0, const Vec<TIER_BF16, NT>& v, int t, int t_first, const Q8& q,
                         const CT& c) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    typedef short s16x2_ __attribute__((ext_vector_type(2)));
    u32x4_ out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {            // registers 4 k .. 4 k + 3 -> dword k
        const u32x4_ w = __builtin_bit_cast(u32x4_, v.u[2 * t + (k >> 1)]);
        const unsigned w_lo = w[2 * (k & 1)], w_hi = w[2 * (k & 1) + 1];      // scalar copies: __builtin_bit_cast of a vector ELEMENT miscompiles
        s16x2_ o;                                // (both halves are written: no v_mov 0 per dword)
        o = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o, __builtin_bit_cast(bf16x2_q, w_lo), q.scale, false);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(o, __builtin_bit_cast(bf16x2_q, w_hi), q.scale, true);
        out[k] = __builtin_bit_cast(unsigned, o);
    }
#ifdef DFN_REC8_NOSTORE       // timing experiment (wrong results): everything but the store instruction
    asm volatile("" ::"v"(out));
    return;
#endif
    gchar* ubase = uniform_ptr((char*)arr + tile * rec8_tile_bytes(rows) + (long)row0 * 32);
    const unsigned voff = (unsigned)((c.lane & 31) * 32 + c.half * 16);
    DFN_GSTORE_B128_NT(ubase, voff, (t - t_first) * 1024, out);
}
// the scale bytes of tiles [t0, t0 + n) of a vector whose tile t_first sits at row row0
template <class CT>
DFN_DEV void store_scale8(void* arr, int rows, long tile, int row0, int t0, int t_first, int n, const Q8& q, const CT& c) {
    gchar* sb = uniform_ptr((char*)arr + tile * rec8_tile_bytes(rows) + (long)rows * 32 + (row0 >> 5) + (t0 - t_first));
    if (c.lane < n) *(__attribute__((address_space(1))) unsigned char*)(sb + c.lane) = (unsigned char)q.e8;
}
template <int NT, class CT>
DFN_DEV void store_tile_act(void* arr, int rows, long tile, int row0, const Vec<TIER_BF16, NT>& v, int t, int t_first, const Q8& q,
                            const CT& c) {
    if constexpr (CT::act_fp4) store_tile4<NT>(arr, rows, tile, row0, v, t, t_first, q, c);
    else store_tile8<NT>(arr, rows, tile, row0, v, t, t_first, q, c);
}
// tiles [t0, t0 + n) of v -> rows row0 + 32 (t - t0) ...   (ACT: the array is act_T, in the build's activation format)
template <int TIER, int NT, class CT, bool ACT = false>
DFN_DEV void store_tiles_T(void* arr, int rows, long tile, int row0, const Vec<TIER, NT>& v, int t0, int n, const CT& c) {
    typedef typename ActT<TIER>::type T;
    // address = wave-uniform base (SGPR pair) + 32-bit per-lane offset + immediate: the "saddr" form of global_store.
    // A per-lane 64-bit pointer per row block cost a v_lshl_add_u64 per store (the row offsets exceed the 4 KiB
    // immediate) and the register allocator spilled those pointers around the MFMA loops.
    gchar* ubase = uniform_ptr((T*)arr + (tile * rows + row0) * 32);
    if constexpr (TIER == TIER_BF16) {
        // MX-fp8 (above): tile pairs share a scale; one 16-byte store per tile
#pragma unroll
        for (int t = 0; t < NT; t += 2)
            if (t >= t0 && t < t0 + n) {
                const int np = (t + 1 < t0 + n && t + 1 < NT) ? 2 : 1;
                const Q8 q = q8_of_tiles<NT, false, ACT && CT::act_fp4>(v, t, np);
                if constexpr (ACT) store_scale_act(arr, rows, tile, row0, t, t0, np, q, c);
                else store_scale8(arr, rows, tile, row0, t, t0, np, q, c);
#pragma unroll
                for (int k = 0; k < 2; ++k)             // (constant trip count: np may be a run-time value)
                    if (k < np && t + k < NT) {
                        if constexpr (ACT) store_tile_act<NT>(arr, rows, tile, row0, v, t + k, t0, q, c);
                        else store_tile8<NT>(arr, rows, tile, row0, v, t + k, t0, q, c);
                    }
            }
    } else {
#ifdef DFN_REC32_NOSTORE      // timing experiment (wrong results): the f32 recorder without its store instructions
        return;
#endif
#pragma unroll
        for (int L = 0; L < 16 * NT; ++L)
            if ((L >> 4) >= t0 && (L >> 4) < t0 + n) {
                const int f = 32 * ((L >> 4) - t0) + tile_feat(0, L & 15);
                const unsigned voff = (unsigned)(4 * c.half * 32 + (c.lane & 31)) * (unsigned)sizeof(T);
                if constexpr (sizeof(T) == 4) {
                    const float x = v.get(L);
                    DFN_GSTORE_B32_NT(ubase, voff, f * 128, x);
                } else {
                    __builtin_nontemporal_store((T)v.get(L), (__attribute__((address_space(1))) T*)(ubase + f * 32 * (int)sizeof(T) + voff));
                }
            }
    }
}
template <int TIER, int NT, class CT, bool ACT = false>
DFN_DEV void store_vec_T(void* arr, int rows, long tile, int row0, const Vec<TIER, NT>& v, const CT& c) {
    store_tiles_T<TIER, NT, CT, ACT>(arr, rows, tile, row0, v, 0, NT, c);
}
template <int TIER, int NT, class CT>
DFN_DEV void rec_vec(const CT& c, int row0, const Vec<TIER, NT>& v) {
    if constexpr (!CT::rec_on) return;
    else store_vec_T<TIER, NT, CT, true>(c.rec.act_T, c.rec.rows, c.rec.pass, row0, v, c);
}

// one value of an f32 vector (local slot L; tile 0 of the vector at row row0) -> its place in a tile-major array: what
// store_tiles_T does for whole tiles, one store instruction at a time (the recorder and the dX chain spread them between MFMAs)
template <class CT>
DFN_DEV void store_val_T32(void* arr, int rows, long tile, int row0, int L, float x, const CT& c) {
#ifdef DFN_REC32_NOSTORE      // timing experiment (wrong results)
    return;
#endif
    gchar* ubase = uniform_ptr((float*)arr + (tile * rows + row0) * 32);
    const int f = 32 * (L >> 4) + tile_feat(0, L & 15);
    const unsigned voff = (unsigned)(4 * c.half * 32 + (c.lane & 31)) * 4u;
    DFN_GSTORE_B32_NT(ubase, voff, f * 128, x);
}

// The A-fragment stream of a pass is strictly sequential (fragment f lives in slab f/32 at position f%32),
// so fragments are prefetched PF_DEPTH ahead into a small register ring that is carried across ops and
// layers: LDS latency hides behind the MFMAs of earlier fragments.  `fp` = index of the next fragment to
// PREFETCH; both indices are compile-time after inlining/unrolling (in the runtime layer loops only their
// slab phase matters, and one 256x256 layer is a whole number of slabs and of ring turns).
// 1: the seven trunk layers as straight-line code (no `act = nxt` copies: 384 v_mov per pass, one in eight of the kernel's
// non-MFMA vector instructions - the SIMD's vector issue port is what the MFMAs compete for - at +21 KB of code per MLP);
// 0: two runtime loops of one layer body each.  (A ping-pong loop of two bodies makes hipcc spill ~60 VGPRs.)
#ifndef DFN_TRUNK_UNROLL
#define DFN_TRUNK_UNROLL 0
#endif
#ifndef DFN_PF_DEPTH
#define DFN_PF_DEPTH 4
#endif
constexpr int PF_DEPTH = DFN_PF_DEPTH;
template <int TIER, class CT> constexpr bool use_asm_fetch() { return tier_is16(TIER) && CT::asm_fetch; }
// (the asm fragment READS are for the 16-bit tiers only.)  The asm DMA serves every kernel of the f32 tier (round 6; before,
// this function answered tier_is16 whenever asm_fetch was on, and the f32 kernels got the builtin: a 64-bit vector address per
// piece and, with "LDS DMA" in the function, s_waitcnt lgkmcnt(0) in front of the MFMAs that read LDS - f32 training forward
// 2562 -> 2496 us, c2_f32 328.2 -> 318.7 ms = 0.921 -> 0.948 of the f32 MFMA peak, the same 138.47 dB; profiles/r06z_*).
// DFN_F32_ASM_DMA = 0: the builtin again.
#ifndef DFN_F32_PIN_FRAGS
#define DFN_F32_PIN_FRAGS 1
#endif
#ifndef DFN_F32_ASM_DMA
#define DFN_F32_ASM_DMA 1
#endif
template <int TIER, class CT> constexpr bool use_asm_dma() { return tier_is16(TIER) ? CT::asm_dma : (DFN_F32_ASM_DMA != 0 || CT::asm_dma); }

#define DFN_FRAG_CASE(K)                                                                                      \
    case K:                                                                                                   \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"((K) * FRAG_BYTES) : "memory"); \
        break;
DFN_DEV void frag_read(u32x4& dst, unsigned vaddr, int pos) {      // pos = fragment index within the slab (constant)
    switch (pos) {
        DFN_FRAG_CASE(0) DFN_FRAG_CASE(1) DFN_FRAG_CASE(2) DFN_FRAG_CASE(3) DFN_FRAG_CASE(4) DFN_FRAG_CASE(5)
        DFN_FRAG_CASE(6) DFN_FRAG_CASE(7) DFN_FRAG_CASE(8) DFN_FRAG_CASE(9) DFN_FRAG_CASE(10) DFN_FRAG_CASE(11)
        DFN_FRAG_CASE(12) DFN_FRAG_CASE(13) DFN_FRAG_CASE(14) DFN_FRAG_CASE(15) DFN_FRAG_CASE(16) DFN_FRAG_CASE(17)
        DFN_FRAG_CASE(18) DFN_FRAG_CASE(19) DFN_FRAG_CASE(20) DFN_FRAG_CASE(21) DFN_FRAG_CASE(22) DFN_FRAG_CASE(23)
        DFN_FRAG_CASE(24) DFN_FRAG_CASE(25) DFN_FRAG_CASE(26) DFN_FRAG_CASE(27) DFN_FRAG_CASE(28) DFN_FRAG_CASE(29)
        DFN_FRAG_CASE(30) DFN_FRAG_CASE(31)
        default: __builtin_unreachable();
    }
}
#undef DFN_FRAG_CASE
// wait until at most `younger` LDS operations issued after the read of `r` are outstanding (they return in order); the
// "+v" ties the wait to the register so that the MFMA consuming it cannot be scheduled above it
DFN_DEV void frag_wait(u32x4& r, int younger) {
    switch (younger) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r)); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r)); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(r)); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(r)); break;
        default: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(r)); break;
    }
}
static_assert(PF_DEPTH >= 1 && PF_DEPTH <= 8, "frag_wait covers up to 7 younger reads");
// the runtime layer loops restart the fragment index at the same value every iteration: the register ring index
// f % PF_DEPTH only stays consistent if a layer (128 fragments) is a whole number of ring turns (depth 6 renders garbage)
static_assert((PF_DEPTH & (PF_DEPTH - 1)) == 0, "PF_DEPTH must be a power of two");

// Hook of the slab hand-over.  The kernels that store a lot (recorder, dX chain) issue the stores of a finished tile
// pair HERE, right behind the wait of slab_advance, not where the pair finishes: vmcnt counts loads and stores in one
// in-order queue, so the next hand-over (which must see this wave's LDS-DMA pieces landed) also waits for every store
// issued before those pieces - behind the hand-over a store burst has a whole slab period to drain while the MFMAs
// run; at the end of a pair it had none (measured: stores cost their full HBM time ON TOP of the compute).
struct NoHook { DFN_DEV void operator()() const {} };
struct NoSide { DFN_DEV void operator()(int) const {} };      // work issued between the MFMAs of a tile group: side(k-step)

template <int TIER> struct Fetch {
    u32x4 buf[PF_DEPTH];
    template <class CT> DFN_DEV void load(int slot, int fp, Stream& s, const CT& c) { load(slot, fp, s, c, NoHook{}); }
    template <class CT, class H> DFN_DEV void load(int slot, int fp, Stream& s, const CT& c, H&& hook) {
        using C = TierCfg<TIER>;
        constexpr int GAP = SLAB_FRAGS / C::LOADS_PER_SLAB;       // fragment reads between two DMA pieces
        if (fp % SLAB_FRAGS == 0) {
            slab_advance<TIER>(s, c.ring, c.wave, c.lane);
            hook();
        }
        if constexpr (use_asm_fetch<TIER, CT>()) frag_read(buf[slot], s.rd_vaddr, fp % SLAB_FRAGS);
        else buf[slot] = *(const lds_u32x4*)(c.ring + c.lane * 16 + s.rd_off + (fp % SLAB_FRAGS) * FRAG_BYTES);
#ifdef DFN_EXP_DBLLDS       // experiment: what does the LDS fragment traffic cost?  read every fragment a second time
        {                   // (same results, +100 % fragment reads; the neighbouring fragment so that the data differ)
            const u32x4 dup = *(const volatile lds_u32x4*)(c.ring + c.lane * 16 + s.rd_off + ((fp + 1) % SLAB_FRAGS) * FRAG_BYTES);
            asm volatile("" ::"v"(dup));
        }
#endif
        if (fp % GAP == GAP / 2) {                                 // one piece of the slab two ahead
            const int k = (fp % SLAB_FRAGS) / GAP;                 // compile-time
            stream_issue_piece<TIER, use_asm_dma<TIER, CT>()>(s, c.ring, c.wave, c.lane, k);
            --s.pf_owed;
            if (k == C::LOADS_PER_SLAB - 1) stream_cursor_next(s);
        }
    }
    // start of a pass: fragments 0..PF_DEPTH-1
    template <class CT> DFN_DEV void prime(Stream& s, const CT& c) {
        stream_flush<TIER, use_asm_dma<TIER, CT>()>(s, c.ring, c.wave, c.lane);
        stream_pass_begin(s);
#pragma unroll
        for (int i = 0; i < PF_DEPTH; ++i) load(i, i, s, c);
    }
};

// One tile-group: acc[g] (g < G output tiles) += W x b over k-units [0, KU) of b.
// Fragments are consumed in stream order [ku][g]; `f` is the running fragment index of the pass.
// TAIL: number of fragments that follow this group in the pass (-1 = plenty): no prefetch past the end.
template <int TIER, int G, int KU, int NTB, int TAIL = -1, class CT, class H = NoHook, class SD = NoSide>
DFN_DEV void gemm_group(f32x16 (&acc)[G], const Vec<TIER, NTB>& b, int& f, Fetch<TIER>& fe, Stream& s,
                        const CT& c, H&& hook = H{}, SD&& side = SD{}) {
#pragma unroll
    for (int ku = 0; ku < KU; ++ku) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int left = (KU - ku) * G - g - 1;           // fragments after this one in the group
            constexpr bool ASM = use_asm_fetch<TIER, CT>();
            if constexpr (ASM) {
                const int after = (TAIL < 0) ? PF_DEPTH : left + TAIL;             // ... in the pass
                frag_wait(fe.buf[f % PF_DEPTH], after < PF_DEPTH - 1 ? after : PF_DEPTH - 1);     // fragment f has landed
            }
            const u32x4 a = fe.buf[f % PF_DEPTH];
            if constexpr (!ASM) {
                if (TAIL < 0 || left + TAIL >= PF_DEPTH) fe.load(f % PF_DEPTH, f + PF_DEPTH, s, c, hook);
            }
            if constexpr (TIER == TIER_BF16) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b.u[ku],
                                                                 acc[g], 0, 0, 0);
            } else if constexpr (TIER == TIER_F16) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), b.u[ku],
                                                                acc[g], 0, 0, 0);
            } else {
                const f32x4 af = __builtin_bit_cast(f32x4, a);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], b.v[4 * ku + e], acc[g], 0, 0, 0);
            }
            if constexpr (ASM) {      // refill the slot just consumed (the MFMA has read its operands when it issued)
                if (TAIL < 0 || left + TAIL >= PF_DEPTH) fe.load(f % PF_DEPTH, f + PF_DEPTH, s, c, hook);
            }
#ifdef DFN_GEMM_SCHEDBAR
            // compiler-scheduled fragment reads: pin every (read of fragment f + PF_DEPTH, MFMA of fragment f) pair where the
            // source puts it.  Left alone, the machine scheduler sinks each ds_read_b128 next to the MFMA that consumes it
            // (fewer live registers) and the LDS latency of every fragment is exposed: s_waitcnt lgkmcnt(0/1) in front of
            // every MFMA, matrix pipe 35 % busy in the dX kernels
            if constexpr (!ASM) __builtin_amdgcn_sched_barrier(DFN_GEMM_SCHEDBAR);
#else
            // f32 tier, kernels that do not record (inference, dX chain): the same pin (round 6, -DDFN_GEMM_SCHEDBAR=0 builds:
            // c2_f32 317.8 -> 314.7 ms, dX 1230 -> 1216 us per field; the recording forward LOSES 20 us with it and stays free;
            // profiles/r06za_*)
            if constexpr (!ASM && TIER == TIER_F32 && !CT::rec_on && (DFN_F32_PIN_FRAGS != 0)) __builtin_amdgcn_sched_barrier(0);
#endif
            ++f;
        }
        side(ku);
#ifdef DFN_PIPE_SCHEDBAR
        // pin the side work of a k-step between its MFMAs and the next step's: without it the scheduler pulls the
        // convert / ReLU units together right behind the last MFMA of their accumulators (where they wait for it)
        if constexpr (!std::is_same<typename std::decay<SD>::type, NoSide>::value) __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

// ReLU as a signed-integer max: negative floats (sign bit set) are negative ints; one v_max_i32, no
// canonicalisation of the MFMA result
DFN_DEV float relu_(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }

// bias vectors live in LDS as [tile][half][16]
template <int G>
DFN_DEV void acc_init(f32x16 (&acc)[G], const lds_f32* bias_lds, int half) {
#ifdef DFN_EXP_NOINIT     // timing experiment (wrong results): accumulators start at zero, no bias reads from LDS
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    return;
#endif
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const lds_f32x4* p = (const lds_f32x4*)(bias_lds + g * 32 + half * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[g][4 * q + e] = v[e];
        }
    }
}
// acc = relu(acc) + bias   (where a constant joins after the ReLU: decoder.py:316-321, 118-119)
template <int G>
DFN_DEV void acc_relu_add(f32x16 (&acc)[G], const lds_f32* bias_lds, int half) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const lds_f32x4* p = (const lds_f32x4*)(bias_lds + g * 32 + half * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[g][4 * q + e] = relu_(acc[g][4 * q + e]) + v[e];
        }
    }
}
// accumulators of G tiles -> tiles [t0, t0+G) of the next layer's B operand.
// bf16 tier with ReLU: pack first (v_cvt_pk_bf16_f32), then ReLU on the packed pairs as a signed 16-bit max
// (v_pk_max_i16: a negative bf16 is a negative int16; relu(round(x)) == round(relu(x))): 1 VALU per value pair
// less than max-then-pack.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int TIER, int G, int NT, bool RELU>
DFN_DEV void acc_to_vec(const f32x16 (&acc)[G], Vec<TIER, NT>& v, int t0) {
#ifdef DFN_EXP_NOEPI      // timing experiment (wrong results): what does the convert / ReLU epilogue cost?  One value per
    {                     // tile keeps the data dependency on the accumulators alive
#pragma unroll
        for (int g = 0; g < G; ++g) v.set(16 * (t0 + g), acc[g][0]);
        return;
    }
#endif
    if constexpr (tier_is16(TIER)) {
        typedef typename std::conditional<TIER == TIER_F16, f16x2, bf16x2>::type pk2;     // v_cvt_pk_{f16,bf16}_f32 (RNE)
        typedef typename std::conditional<TIER == TIER_F16, f16x8, bf16x8>::type pk8;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 x = {acc[g][8 * h + 2 * e], acc[g][8 * h + 2 * e + 1]};
                    const pk2 pk = __builtin_convertvector(x, pk2);
                    if constexpr (RELU) {
                        const s16x2 m = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), (s16x2)(0));   // v_pk_max_i16
                        w[e] = __builtin_bit_cast(unsigned, m);
                    } else {
                        w[e] = __builtin_bit_cast(unsigned, pk);
                    }
                }
                const u32x4 q = {w[0], w[1], w[2], w[3]};
                v.u[2 * (t0 + g) + h] = __builtin_bit_cast(pk8, q);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = RELU ? relu_(acc[g][r]) : acc[g][r];
                v.set(16 * (t0 + g) + r, x);
            }
    }
}

// training recorder, per tile pair.  Values (post-activation, as the next layer sees them): rows row0 + {0..63} of
// act_T - bf16 tier: the packed operand words acc_to_vec made (tiles t0, t0 + 1 of `out`); issued at the next slab
// hand-over (NoHook above).  ReLU bits: one dword per lane from the accumulators, stored at once (rec_mask_pair).
template <int TIER, int NT, class CT>
DFN_DEV void rec_vals(const CT& c, int row0, const Vec<TIER, NT>& out, int t0) {
    if constexpr (CT::rec_on)
        if (row0 >= 0) store_tiles_T<TIER, NT, CT, true>(c.rec.act_T, c.rec.rows, c.rec.pass, row0, out, t0, 2, c);
}
// ReLU bit of value b (= 16 g + r: tile g of the pair, accumulator register r) of a tile pair inside its mask dword: value b
// lives in the (b & 1) half of packed operand word b >> 1, and the 16-bit tiers build the dword FROM those words (below)
DFN_HD constexpr int mask_pos(int b) { return ((b & 1) << 4) | (b >> 1); }
template <class CT>
DFN_DEV void rec_mask_pair(const CT& c, int mask_dword, const f32x16 (&acc)[2]) {
#ifdef DFN_REC_NOMASK         // timing experiment (wrong results): no ReLU bits at all (f32 route)
    return;
#endif
    if constexpr (CT::rec_on) {
        if (mask_dword >= 0) {
            unsigned bits = 0;
#pragma unroll
            for (int b = 0; b < 32; ++b) bits |= (acc[b >> 4][b & 15] > 0.f) ? (1u << mask_pos(b)) : 0u;
            gchar* mb = uniform_ptr(c.rec.masks + ((long)c.rec.pass * c.rec.mask_dwords + mask_dword) * 64);
            DFN_GSTORE_B32(mb, (unsigned)c.lane * 4u, 0, bits);      // (ordinary: the dX kernels read these next)
        }
    }
}

// The same dword from the pair's packed, ReLU'd operand words (16-bit tiers): a positive bf16 / f16 is a non-zero 16-bit
// pattern >= 1, so v_pk_min_u16(word, 0x00010001) is (bit of the low half) | (bit of the high half) << 16 and one
// v_lshl_or_b32 drops both at mask_pos: 1 VALU per value instead of the 3 (compare, select, or) of the f32 route - a
// quarter of the training forward's vector instructions were mask bits.
template <int TIER, int NT, class CT>
DFN_DEV void rec_mask_pair_packed(const CT& c, int mask_dword, const Vec<TIER, NT>& out, int t0) {
#ifdef DFN_REC_NOMASK         // timing experiment (wrong results): no ReLU bits at all
    return;
#endif
    if constexpr (CT::rec_on && tier_is16(TIER)) {
        if (mask_dword >= 0) {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            unsigned b2[2] = {0, 0};          // two v_lshl_or_b32 chains (the compiler's shift + v_or3 tree: 1.7 VALU per word)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4_ q = __builtin_bit_cast(u32x4_, out.u[2 * t0 + k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned w = q[e];          // (a scalar: __builtin_bit_cast of a vector ELEMENT miscompiles)
                    // asm: LLVM rewrites min(u16x2, {1, 1}) into two 16-bit compares, two selects and a v_perm (5 VALU per
                    // word; the training forward carried 4,000 of them)
                    unsigned t;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(w), "s"(0x00010001u));
                    asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(b2[e & 1]) : "v"(t), "n"(4 * k + e));
                }
            }
            const unsigned bits = b2[0] | b2[1];
#ifdef DFN_REC_NOMASKSTORE    // timing experiment (wrong results): the bits, not their store
            asm volatile("" ::"v"(bits));
            return;
#endif
            gchar* mb = uniform_ptr(c.rec.masks + ((long)c.rec.pass * c.rec.mask_dwords + mask_dword) * 64);
            DFN_GSTORE_B32(mb, (unsigned)c.lane * 4u, 0, bits);
        }
    }
}

// ---- software-pipelined tile pairs (16-bit inference kernels) ----------------------------------------------------------
// Word w (0..15) of a tile pair's epilogue: accumulator registers (2j, 2j+1) of tile w >> 3 -> one packed 32-bit word of the
// next layer's operand (v_cvt_pk_{f16,bf16}_f32 + v_pk_max_i16), exactly what acc_to_vec does for the whole pair.
template <int TIER, int NT, bool RELU>
DFN_DEV void epi_word(const f32x16 (&acc)[2], Vec<TIER, NT>& v, int t0, int w) {
    typedef typename std::conditional<TIER == TIER_F16, f16x2, bf16x2>::type pk2;
    typedef typename std::conditional<TIER == TIER_F16, f16x8, bf16x8>::type pk8;
    const int g = w >> 3, h = (w >> 2) & 1, e = w & 3;
    const f32x2 x = {acc[g][8 * h + 2 * e], acc[g][8 * h + 2 * e + 1]};
    const pk2 pk = __builtin_convertvector(x, pk2);
    unsigned word;
    if constexpr (RELU) word = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), (s16x2)(0)));
    else word = __builtin_bit_cast(unsigned, pk);
    u32x4 q = __builtin_bit_cast(u32x4, v.u[2 * (t0 + g) + h]);
    q[e] = word;
    v.u[2 * (t0 + g) + h] = __builtin_bit_cast(pk8, q);
}
// bias read q (0..7) of a tile pair's accumulator initialisation (one ds_read_b128 = 4 registers): acc_init, piecewise
DFN_DEV void init_quad(f32x16 (&acc)[2], const lds_f32* bias_lds, int half, int q) {
    const int g = q >> 2, qq = q & 3;
    const f32x4 v = ((const lds_f32x4*)(bias_lds + g * 32 + half * 16))[qq];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[g][4 * qq + e] = v[e];
}
// Side work of tile pair tg while its MFMAs issue: k-steps [0, KU/2) convert the PREVIOUS pair (other accumulator set),
// k-steps [KU/2, KU) read the biases of the NEXT pair into that set.  All indices are compile-time after unrolling.
template <int TIER, int OT, int KU, bool RELU> struct PipeSide {
    static constexpr int ES = KU / 2 > 0 ? KU / 2 : 1, WPS = (16 + ES - 1) / ES;          // epilogue steps, words per step
    static constexpr int BS = KU - ES > 0 ? KU - ES : 1, QPS = (8 + BS - 1) / BS;          // bias steps, reads per step
    f32x16 (&other)[2];
    Vec<TIER, OT>& out;
    const lds_f32* bias_next;       // bias of pair tg + 1 (null: none)
    int t_prev;                     // first tile of the previous pair (-1: none)
    int half;
    DFN_DEV void operator()(int ku) const {
        if (t_prev >= 0 && ku < ES) {
#pragma unroll
            for (int w = 0; w < WPS; ++w)
                if (ku * WPS + w < 16) epi_word<TIER, OT, RELU>(other, out, t_prev, ku * WPS + w);
        }
        if (bias_next && ku >= ES) {
#pragma unroll
            for (int q = 0; q < QPS; ++q)
                if ((ku - ES) * QPS + q < 8) init_quad(other, bias_next, half, (ku - ES) * QPS + q);
        }
    }
};
template <int TIER, int OT, int KU, int NTB, bool RELU, class CT>
DFN_DEV void layer_pipe(Vec<TIER, OT>& out, const Vec<TIER, NTB>& in, const lds_f32* bias, int& f, Fetch<TIER>& fe,
                        Stream& s, const CT& c) {
    static_assert(OT % 2 == 0 && KU >= 2, "tile pairs");
    f32x16 acc[2][2];
    acc_init<2>(acc[0], bias, c.half);
#pragma unroll
    for (int tg = 0; tg < OT / 2; ++tg) {
        const int cur = tg & 1;
        const PipeSide<TIER, OT, KU, RELU> side = {acc[cur ^ 1], out, tg + 1 < OT / 2 ? bias + (tg + 1) * 64 : nullptr,
                                                   tg > 0 ? 2 * (tg - 1) : -1, c.half};
        gemm_group<TIER, 2, KU, NTB>(acc[cur], in, f, fe, s, c, NoHook{}, side);
        if (KU - PipeSide<TIER, OT, KU, RELU>::ES < 1 && tg + 1 < OT / 2) acc_init<2>(acc[cur ^ 1], bias + (tg + 1) * 64, c.half);
    }
    acc_to_vec<TIER, 2, OT, RELU>(acc[(OT / 2 - 1) & 1], out, OT - 2);        // the last pair: not overlapped
}

// DFN_REC_SPREAD (training recorder, 16-bit tier): 1 = the scale search and the two 16-byte tile stores of tile pair tg - 1
// go out between the MFMAs of pair tg (RecSide), like the dX chain's PutSide (dfn_bwd.h); 0 = as one burst at the next slab
// hand-over.
#ifndef DFN_REC_SPREAD
#define DFN_REC_SPREAD 1
#endif
template <int TIER, int OT, int KU, class CT, bool NONNEG> struct RecSide {
    const CT& c;
    const Vec<TIER, OT>& out;
    int rec_row, prev;              // prev: the pair whose values are stored (-1: none)
    mutable Q8 q;                   // the pair's scale, found at k-step 0
    // k-step 0 finds the pair's scale; its two 16-byte tile stores (MX-fp8, above) go out at k-steps S0 and S1
    static constexpr int S0 = KU > 1 ? 1 : 0, S1 = KU > 2 ? 1 + (KU - 1) / 2 : S0;
    DFN_DEV void operator()(int ku) const {
        if constexpr (CT::rec_on && TIER == TIER_BF16) {
            if (prev < 0 || rec_row < 0) return;
            if (ku == 0) {
                q = q8_of_tiles<OT, NONNEG, CT::act_fp4>(out, 2 * prev, 2);
                store_scale_act(c.rec.act_T, c.rec.rows, c.rec.pass, rec_row, 2 * prev, 0, 2, q, c);
            }
            if (ku == S0) store_tile_act<OT>(c.rec.act_T, c.rec.rows, c.rec.pass, rec_row, out, 2 * prev, 0, q, c);
            if (ku == S1) store_tile_act<OT>(c.rec.act_T, c.rec.rows, c.rec.pass, rec_row, out, 2 * prev + 1, 0, q, c);
        }
    }
};

// DFN_REC32_SPREAD (training recorder, f32 tier): 1 = the 32 value stores and the ReLU bits of tile pair tg - 1 go out one by one
// between the MFMAs of pair tg (the kernel runs ONE wave per SIMD: a burst of 32 store instructions fills the memory queue and
// the wave - and with it the matrix pipe - waits until it drains); 0 = as one burst at the next slab hand-over.
#ifndef DFN_REC32_SPREAD
#define DFN_REC32_SPREAD 1
#endif
template <int OT, int KU, class CT> struct RecSide32 {
    const CT& c;
    const Vec<TIER_F32, OT>& out;
    int rec_row, mask_dword, prev;  // mask_dword: of pair `prev` (-1: none); prev: the pair whose values go out (-1: none)
    unsigned& bits;
    static constexpr int VPS = (32 + KU - 1) / KU;          // values per k-step
    DFN_DEV void operator()(int ku) const {
        if constexpr (CT::rec_on) {
            if (prev < 0) return;
#pragma unroll
            for (int w = 0; w < VPS; ++w) {
                const int b = ku * VPS + w;
                if (b < 32) {
                    const float x = out.v[32 * prev + b];
                    if (rec_row >= 0) store_val_T32(c.rec.act_T, c.rec.rows, c.rec.pass, rec_row + 64 * prev, b, x, c);
#ifndef DFN_REC_NOMASK
                    if (mask_dword >= 0) bits |= (x > 0.f) ? (1u << mask_pos(b)) : 0u;     // (relu(x) > 0 <=> x > 0: the accumulator's bit)
#endif
                }
            }
#ifndef DFN_REC_NOMASK
            if (ku == KU - 1 && mask_dword >= 0) {
                gchar* mb = uniform_ptr(c.rec.masks + ((long)c.rec.pass * c.rec.mask_dwords + mask_dword) * 64);
                DFN_GSTORE_B32(mb, (unsigned)c.lane * 4u, 0, bits);
            }
#endif
        }
    }
};

// out[OT tiles] = act( bias + W x in ), tile pairs; KU = k-units of `in` used
template <int TIER, int OT, int KU, int NTB, bool RELU, class CT>
DFN_DEV void layer(Vec<TIER, OT>& out, const Vec<TIER, NTB>& in, const lds_f32* bias, int& f, Fetch<TIER>& fe,
                   Stream& s, const CT& c, int rec_row = -1, int rec_mask = -1) {
    static_assert(OT % 2 == 0, "tile pairs");
    if constexpr (CT::pipe && tier_is16(TIER) && OT >= 4 && KU >= 2) {
        layer_pipe<TIER, OT, KU, NTB, RELU>(out, in, bias, f, fe, s, c);
        return;
    }
    if constexpr (CT::rec_on && TIER == TIER_F32 && (DFN_REC32_SPREAD != 0) && KU >= 8) {
        unsigned bits = 0;
#pragma unroll
        for (int tg = 0; tg < OT / 2; ++tg) {
            f32x16 acc[2];
            acc_init<2>(acc, bias + tg * 64, c.half);
            bits = 0;
            gemm_group<TIER, 2, KU, NTB>(acc, in, f, fe, s, c, NoHook{},
                                         RecSide32<OT, KU, CT>{c, out, rec_row, (rec_mask < 0 || tg == 0) ? -1 : rec_mask + tg - 1, tg - 1, bits});
            acc_to_vec<TIER, 2, OT, RELU>(acc, out, 2 * tg);
            if (tg == OT / 2 - 1) rec_mask_pair(c, rec_mask < 0 ? -1 : rec_mask + tg, acc);       // the last pair: a burst
        }
        if (rec_row >= 0) rec_vals<TIER>(c, rec_row + 64 * (OT / 2 - 1), out, OT - 2);
        return;
    }
    constexpr bool SPREAD = CT::rec_on && TIER == TIER_BF16 && (DFN_REC_SPREAD != 0);
    bool pend = false;          // the values of pair tg - 1 wait for the next slab hand-over
#pragma unroll
    for (int tg = 0; tg < OT / 2; ++tg) {
        f32x16 acc[2];
        acc_init<2>(acc, bias + tg * 64, c.half);
        if constexpr (SPREAD) {
            gemm_group<TIER, 2, KU, NTB>(acc, in, f, fe, s, c, NoHook{}, RecSide<TIER, OT, KU, CT, RELU>{c, out, rec_row, tg - 1, {}});
        } else {
            auto flush = [&] {
                if (pend) rec_vals<TIER>(c, rec_row + 64 * (tg - 1), out, 2 * (tg - 1));
                pend = false;
            };
            gemm_group<TIER, 2, KU, NTB>(acc, in, f, fe, s, c, flush);
            flush();                // no hand-over inside this group
        }
        acc_to_vec<TIER, 2, OT, RELU>(acc, out, 2 * tg);
        if constexpr (RELU && tier_is16(TIER)) rec_mask_pair_packed<TIER, OT>(c, rec_mask < 0 ? -1 : rec_mask + tg, out, 2 * tg);
        else rec_mask_pair(c, rec_mask < 0 ? -1 : rec_mask + tg, acc);
        pend = CT::rec_on && rec_row >= 0;
    }
    if (pend) rec_vals<TIER>(c, rec_row + 64 * (OT / 2 - 1), out, OT - 2);
}
// out = relu(bias + W x in) + bias2 + W2 x in2      (no activation after the skip)
template <int TIER, int OT, int KU, int NTB, int KU2, int NTB2, class CT>
DFN_DEV void layer_skip(Vec<TIER, OT>& out, const Vec<TIER, NTB>& in, const lds_f32* bias,
                        const Vec<TIER, NTB2>& in2, const lds_f32* bias2, int& f, Fetch<TIER>& fe,
                        Stream& s, const CT& c, int rec_row = -1, int mask_dword0 = -1) {
    bool pend = false;
#pragma unroll
    for (int tg = 0; tg < OT / 2; ++tg) {
        f32x16 acc[2];
        acc_init<2>(acc, bias + tg * 64, c.half);
        auto flush = [&] {
            if (pend) rec_vals<TIER>(c, rec_row + 64 * (tg - 1), out, 2 * (tg - 1));
            pend = false;
        };
        gemm_group<TIER, 2, KU, NTB>(acc, in, f, fe, s, c, flush);
        rec_mask_pair(c, mask_dword0 < 0 ? -1 : mask_dword0 + tg, acc);                 // ReLU bits of the pre-skip value
        acc_relu_add<2>(acc, bias2 + tg * 64, c.half);
        gemm_group<TIER, 2, KU2, NTB2>(acc, in2, f, fe, s, c, flush);
        flush();
        acc_to_vec<TIER, 2, OT, false>(acc, out, 2 * tg);                               // post-skip value (no ReLU)
        pend = CT::rec_on && rec_row >= 0;
    }
    if (pend) rec_vals<TIER>(c, rec_row + 64 * (OT / 2 - 1), out, OT - 2);
}

// ---- positional encodings ----------------------------------------------------------------------------------
// decoder.py:257-275: p/2, then per octave [sin(c_i p)(3), cos(c_i p)(3)], c_i = float32(2^i * pi).
// Slot s of the PE vector is reference column s = 6*octave + 3*is_cos + axis (identity map, dfn_layout.h);
// this lane (half h) holds, in local slot L, column 32*(L>>4) + tile_feat(h, L&15).
__device__ __constant__ float PE_FREQ[10] = {
    (float)(1.0 * 3.14159265358979323846),   (float)(2.0 * 3.14159265358979323846),
    (float)(4.0 * 3.14159265358979323846),   (float)(8.0 * 3.14159265358979323846),
    (float)(16.0 * 3.14159265358979323846),  (float)(32.0 * 3.14159265358979323846),
    (float)(64.0 * 3.14159265358979323846),  (float)(128.0 * 3.14159265358979323846),
    (float)(256.0 * 3.14159265358979323846), (float)(512.0 * 3.14159265358979323846)};

template <int TIER, int NT, int NCOL>
DFN_DEV void posenc(Vec<TIER, NT>& v, const float (&p)[3], int half) {
    const float ph[3] = {p[0] * 0.5f, p[1] * 0.5f, p[2] * 0.5f};    // p / downscale_p_by (exact)
#pragma unroll
    for (int L = 0; L < 16 * NT; ++L) {
        const int c0 = 32 * (L >> 4) + tile_feat(0, L & 15), c1 = c0 + 4;      // compile-time
        const bool ok0 = c0 < NCOL, ok1 = c1 < NCOL;
        if (!ok0 && !ok1) {
            v.set(L, 0.f);
            continue;
        }
        const int o0 = ok0 ? c0 / 6 : 0, o1 = ok1 ? c1 / 6 : 0;
        const int a0 = (c0 % 6) % 3, a1 = (c1 % 6) % 3;
        const bool cos0 = (c0 % 6) >= 3, cos1 = (c1 % 6) >= 3;
        float val;
        if constexpr (TIER == TIER_F32) {
            // argument rounded like the reference: fl32(fl32(2^i*pi) * fl32(p/2)); accurate sin/cos
            const float x = half ? __fmul_rn(PE_FREQ[o1], ph[a1]) : __fmul_rn(PE_FREQ[o0], ph[a0]);
            if (cos0 == cos1) {
                val = cos0 ? cosf(x) : sinf(x);
            } else {
                float sv, cv;
                sincosf(x, &sv, &cv);
                val = (half ? cos1 : cos0) ? cv : sv;
            }
        } else {
            // hardware sin takes revolutions: 2^i*pi*(p/2) = 2*pi * (2^(i-1) * p/2) (exact power-of-two
            // scaling, no rounded pi); cos(x) = sin(x + 1/4 rev)
            const float r0 = ph[a0] * ((float)(1 << o0) * 0.5f), r1 = ph[a1] * ((float)(1 << o1) * 0.5f);
            const float q0 = cos0 ? 0.25f : 0.f, q1 = cos1 ? 0.25f : 0.f;
            const float rev = __builtin_amdgcn_fractf(half ? r1 : r0) + (half ? q1 : q0);
            val = __builtin_amdgcn_sinf(rev);
        }
        const bool ok = half ? ok1 : ok0;
        v.set(L, ok ? val : 0.f);
    }
}

// ---- per-field programs: fragment and bias-blob offsets ------------------------------------------------------
// Op order of one pass (dfn_plan.cpp mirrors it):
//   head : IN(PE) | L1 L2 L3 | L4+SKIP(PE) | L5 L6 L7 | VIEW(act+view, 9 tiles) | OUT(1 tile)
//   torso: E0 S0 (PE) | E1 S1 E2 S2 | E3+ESKIP(PE) S3 | E4 S4 | EO SO | IN(pd) | L1..L3 | L4+SKIP(pd) |
//          L5..L7 | VIEW | OUT
template <int TIER> struct Prog {
    static constexpr int UPT = TierCfg<TIER>::UPT;
    static constexpr int KU_PE = 2 * UPT, KU_VIEW = UPT, KU_ACT = 8 * UPT, KU_D = 2 * UPT, KU_PD = 4 * UPT;
    static constexpr int F_LAYER = 8 * KU_ACT;          // fragments of one 256x256 layer (multiple of 32)
    static constexpr int F_TAIL = 9 * (KU_ACT + KU_VIEW) + KU_ACT;
    // head
    static constexpr int H_FRAGS = 8 * KU_PE + 7 * F_LAYER + 8 * KU_PE + F_TAIL;
    static constexpr int H_SLABS = (H_FRAGS + SLAB_FRAGS - 1) / SLAB_FRAGS;
    // head bias blob (floats): in(256) L1..L4(4x256) skip(256) L5..L7(3x256) view(288) out(32)
    static constexpr int H_B_IN = 0, H_B_L1 = 256, H_B_SKIP = 5 * 256, H_B_L5 = 6 * 256, H_B_VIEW = 9 * 256,
                         H_B_OUT = 9 * 256 + 288, H_NBIAS = 9 * 256 + 288 + 32;
    // torso
    static constexpr int F_D = 2 * KU_D;                // fragments of one 64x64 layer
    static constexpr int T_F_DEFORM = 4 * KU_PE + 10 * F_D + 2 * KU_PE;   // E0 S0 | 10 64x64 layers | ESKIP
    static constexpr int T_FRAGS = T_F_DEFORM + 8 * KU_PD + 7 * F_LAYER + 8 * KU_PD + F_TAIL;
    static constexpr int T_SLABS = (T_FRAGS + SLAB_FRAGS - 1) / SLAB_FRAGS;
    // torso bias blob: E0 S0 E1 S1 E2 S2 E3 ESKIP S3 SSKIP E4 S4 EO SO (14 x 64), then the trunk like head
    static constexpr int T_B_E0 = 0, T_B_S0 = 64, T_B_E1 = 128, T_B_S1 = 192, T_B_E2 = 256, T_B_S2 = 320,
                         T_B_E3 = 384, T_B_ESKIP = 448, T_B_S3 = 512, T_B_SSKIP = 576, T_B_E4 = 640,
                         T_B_S4 = 704, T_B_EO = 768, T_B_SO = 832, T_B_IN = 896;
    static constexpr int T_B_L1 = T_B_IN + 256, T_B_SKIP = T_B_IN + 5 * 256, T_B_L5 = T_B_IN + 6 * 256,
                         T_B_VIEW = T_B_IN + 9 * 256, T_B_OUT = T_B_VIEW + 288, T_NBIAS = T_B_OUT + 32;
};

// d/|d| of the point's ray, read from LDS only when the view layer needs it (keeps it out of registers)
struct DhatRef {
    const volatile lds_f32* p;
    int stride;
    DFN_DEV void load(float (&d)[3]) const {
        d[0] = p[0];
        d[1] = p[stride];
        d[2] = p[2 * stride];
    }
};

// rows of act_T and dwords of the mask array (training recorder); trunk offsets are relative to R_TRUNK
struct RecMap {
    // trunk: a0..a7 (8 x 256), h (256), view PE (32)
    static constexpr int T_A0 = 0, T_H = 8 * 256, T_VIEW = 9 * 256, T_ROWS = 9 * 256 + 32;
    static constexpr int TM_A0 = 0, TM_A4R = 4 * 4, TM_A5 = 5 * 4, TM_H = 8 * 4, TM_DWORDS = 9 * 4;   // a0..a3,a4r,a5..a7,h
    // head: pe(64) then the trunk
    static constexpr int H_PE = 0, H_TRUNK = 64, H_ROWS = 64 + T_ROWS, H_MTRUNK = 0, H_MDWORDS = TM_DWORDS;
    // torso: pe(64), ve0 vs0 ve1 vs1 ve2 vs2 ve3 vs3 ve4 vs4 (10 x 64), pd(128), trunk
    static constexpr int S_PE = 0, S_D0 = 64, S_PD = 64 + 640, S_TRUNK = S_PD + 128, S_ROWS = S_TRUNK + T_ROWS;
    static constexpr int S_MD0 = 0, S_MTRUNK = 10, S_MDWORDS = 10 + TM_DWORDS;     // one dword per 64-wide vector
};

struct MlpOut {
    float sigma, r, g, b;      // valid in lanes 0..31 (half 0): raw sigma, sigmoid rgb of point lane&31
};

DFN_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// shared trunk: L1..L3, L4+skip, L5..L7, view layer, rgb head.  `act` holds relu(first layer) on entry;
// f_l1 is the slab phase (fragment index mod 32) at L1.
template <int TIER, int NTP, int KUP, class CT>
DFN_DEV MlpOut mlp_trunk(Vec<TIER, 8>& act, const Vec<TIER, NTP>& pvec, const DhatRef& dref,
                         const lds_f32* bias, int b_l1, int b_skip, int b_l5, int b_view, int b_out,
                         int f_l1, Fetch<TIER>& fe, Stream& s, const CT& c, int r_trunk, int m_trunk) {
    using P = Prog<TIER>;
    Vec<TIER, 8> nxt;
    // blocks[0..6] with the skip after blocks[3] (decoder.py:313-325), ping-ponging between the two operand vectors:
    // the trunk's output a7 ends up in `nxt`.  Every layer starts at the same slab phase f_l1 (a layer is 128 fragments,
    // the skip adds a multiple of 32).  DFN_TRUNK_UNROLL: see below.
    const auto blk_bias = [&](int k) { return bias + (k < 4 ? b_l1 + 256 * k : b_l5 + 256 * (k - 4)); };
    const auto blk_row = [&](int k) { return r_trunk + RecMap::T_A0 + 256 * (k + 1); };          // output of blocks[k]
    const auto blk_mask = [&](int k) { return m_trunk + (k < 4 ? RecMap::TM_A0 + 4 * (k + 1) : RecMap::TM_A5 + 4 * (k - 4)); };
    int f = f_l1;
#if DFN_TRUNK_UNROLL
    // straight-line: seven layer bodies, the operand vectors alternate by name (no copies, no loop-carried vectors)
#define DFN_PLAIN(OUT, IN, K) f = f_l1; layer<TIER, 8, P::KU_ACT, 8, true>(OUT, IN, blk_bias(K), f, fe, s, c, blk_row(K), blk_mask(K))
    DFN_PLAIN(nxt, act, 0);
    DFN_PLAIN(act, nxt, 1);
    DFN_PLAIN(nxt, act, 2);
    f = f_l1;       // blocks[3], then the skip: relu(.) + fc_z_skips(z) + fc_p_skips(p)   (decoder.py:316-325)
    layer_skip<TIER, 8, P::KU_ACT, 8, KUP, NTP>(act, nxt, bias + b_l1 + 256 * 3, pvec, bias + b_skip, f, fe, s, c,
                                                r_trunk + RecMap::T_A0 + 256 * 4, m_trunk + RecMap::TM_A4R);
    DFN_PLAIN(nxt, act, 4);
    DFN_PLAIN(act, nxt, 5);
    DFN_PLAIN(nxt, act, 6);
#undef DFN_PLAIN
#else
    // runtime loops (one layer body each): the 64-register copy after every layer is the price of the small code
    for (int l = 0; l < 3; ++l) {
        f = f_l1;
        layer<TIER, 8, P::KU_ACT, 8, true>(nxt, act, bias + b_l1 + 256 * l, f, fe, s, c,
                                           r_trunk + RecMap::T_A0 + 256 * (l + 1), m_trunk + RecMap::TM_A0 + 4 * (l + 1));
        act = nxt;
    }
    f = f_l1;       // blocks[3], then the skip: relu(.) + fc_z_skips(z) + fc_p_skips(p)   (decoder.py:316-325)
    layer_skip<TIER, 8, P::KU_ACT, 8, KUP, NTP>(nxt, act, bias + b_l1 + 256 * 3, pvec, bias + b_skip, f, fe, s, c,
                                                r_trunk + RecMap::T_A0 + 256 * 4, m_trunk + RecMap::TM_A4R);
    act = nxt;
    for (int l = 0; l < 3; ++l) {
        f = f_l1;
        layer<TIER, 8, P::KU_ACT, 8, true>(nxt, act, bias + b_l5 + 256 * l, f, fe, s, c,
                                           r_trunk + RecMap::T_A0 + 256 * (5 + l), m_trunk + RecMap::TM_A5 + 4 * l);
        act = nxt;
    }
#endif
    // the trunk's output a7: in `nxt` after the straight-line form, in `act` after the loops; the other vector takes the
    // view layer's output
    Vec<TIER, 8>& a7 = DFN_TRUNK_UNROLL ? nxt : act;
    Vec<TIER, 8>& hid = DFN_TRUNK_UNROLL ? act : nxt;
    // feat_view (+ sigma_out as row 0 of a 9th tile) on [act ; view PE]   (decoder.py:329-340)
    MlpOut o;
    {
        Vec<TIER, 1> vview;
        float dhat[3];
        dref.load(dhat);
        posenc<TIER, 1, NPEV>(vview, dhat, c.half);
        rec_vec<TIER, 1>(c, r_trunk + RecMap::T_VIEW, vview);
        bool pend = false;      // recorder values of pair tg - 1, issued at the next slab hand-over (NoHook)
        int ptg = 0;
        auto flush = [&] {
            if (pend) rec_vals<TIER>(c, r_trunk + RecMap::T_H + 64 * ptg, hid, 2 * ptg);
            pend = false;
        };
#pragma unroll
        for (int tg = 0; tg < 4; ++tg) {
            f32x16 acc[2];
            acc_init<2>(acc, bias + b_view + tg * 64, c.half);
            gemm_group<TIER, 2, P::KU_ACT, 8>(acc, a7, f, fe, s, c, flush);
            gemm_group<TIER, 2, P::KU_VIEW, 1>(acc, vview, f, fe, s, c, flush);
            flush();
            acc_to_vec<TIER, 2, 8, true>(acc, hid, 2 * tg);
            if constexpr (tier_is16(TIER)) rec_mask_pair_packed<TIER, 8>(c, m_trunk + RecMap::TM_H + tg, hid, 2 * tg);
            else rec_mask_pair(c, m_trunk + RecMap::TM_H + tg, acc);
            pend = CT::rec_on;
            ptg = tg;
        }
        f32x16 acc1[1];
        acc_init<1>(acc1, bias + b_view + 256, c.half);
        gemm_group<TIER, 1, P::KU_ACT, 8>(acc1, a7, f, fe, s, c, flush);
        gemm_group<TIER, 1, P::KU_VIEW, 1>(acc1, vview, f, fe, s, c, flush);
        flush();
        o.sigma = acc1[0][0];
    }
    // feat_out + sigmoid   (decoder.py:344-347)
    {
        f32x16 acc1[1];
        acc_init<1>(acc1, bias + b_out, c.half);
        gemm_group<TIER, 1, P::KU_ACT, 8, 0>(acc1, hid, f, fe, s, c);     // last op of the pass
        o.r = sigmoidf_(acc1[0][0]);
        o.g = sigmoidf_(acc1[0][1]);
        o.b = sigmoidf_(acc1[0][2]);
    }
    return o;
}

// ---- head pass: decoder.py:291-349 with head_or_torso == 'head' ----------------------------------------------
template <int TIER, class CT>
DFN_DEV MlpOut mlp_head(const float (&p)[3], const DhatRef& dhat, const lds_f32* bias, Stream& s,
                        const CT& c) {
    using P = Prog<TIER>;
    Vec<TIER, 2> pe;
    posenc<TIER, 2, NPE>(pe, p, c.half);
    rec_vec<TIER, 2>(c, RecMap::H_PE, pe);
    Vec<TIER, 8> act;
    int f = 0;
    Fetch<TIER> fe;
    fe.prime(s, c);
    layer<TIER, 8, P::KU_PE, 2, true>(act, pe, bias + P::H_B_IN, f, fe, s, c, RecMap::H_TRUNK + RecMap::T_A0,
                                      RecMap::H_MTRUNK + RecMap::TM_A0);
    return mlp_trunk<TIER, 2, P::KU_PE>(act, pe, dhat, bias, P::H_B_L1, P::H_B_SKIP, P::H_B_L5, P::H_B_VIEW,
                                        P::H_B_OUT, f % SLAB_FRAGS, fe, s, c, RecMap::H_TRUNK, RecMap::H_MTRUNK);
}

// ---- torso pass: deformation field (decoder.py:109-134, 297-299) then the trunk ---------------------------
template <int TIER, class CT>
DFN_DEV MlpOut mlp_torso(const float (&p)[3], const DhatRef& dhat, const lds_f32* bias, Stream& s,
                         const CT& c) {
    using P = Prog<TIER>;
    Vec<TIER, 2> pe;
    posenc<TIER, 2, NPE>(pe, p, c.half);
    rec_vec<TIER, 2>(c, RecMap::S_PE, pe);
    Vec<TIER, 2> ve, vs, vn;
    int f = 0;
    Fetch<TIER> fe;
    fe.prime(s, c);
    // deformation vector k (0: ve0, 1: vs0, 2: ve1, ...) is recorded as the next GEMM's input, with its ReLU bits
#define DFN_RD(k) RecMap::S_D0 + 64 * (k), RecMap::S_MD0 + (k)
    layer<TIER, 2, P::KU_PE, 2, true>(ve, pe, bias + P::T_B_E0, f, fe, s, c, DFN_RD(0));
    layer<TIER, 2, P::KU_PE, 2, true>(vs, pe, bias + P::T_B_S0, f, fe, s, c, DFN_RD(1));
    layer<TIER, 2, P::KU_D, 2, true>(vn, ve, bias + P::T_B_E1, f, fe, s, c, DFN_RD(2));  ve = vn;
    layer<TIER, 2, P::KU_D, 2, true>(vn, vs, bias + P::T_B_S1, f, fe, s, c, DFN_RD(3));  vs = vn;
    layer<TIER, 2, P::KU_D, 2, true>(vn, ve, bias + P::T_B_E2, f, fe, s, c, DFN_RD(4));  ve = vn;
    layer<TIER, 2, P::KU_D, 2, true>(vn, vs, bias + P::T_B_S2, f, fe, s, c, DFN_RD(5));  vs = vn;
    // skips join after the ReLU of layer idx 3 (decoder.py:118-119, 128-129)
    layer_skip<TIER, 2, P::KU_D, 2, P::KU_PE, 2>(vn, ve, bias + P::T_B_E3, pe, bias + P::T_B_ESKIP, f, fe, s, c,
                                                 DFN_RD(6));
    ve = vn;
    {   // signal net: its skip input is the per-frame pose signal -> a constant added after the ReLU
        f32x16 acc[2];
        acc_init<2>(acc, bias + P::T_B_S3, c.half);
        gemm_group<TIER, 2, P::KU_D, 2>(acc, vs, f, fe, s, c);
        rec_mask_pair(c, RecMap::S_MD0 + 7, acc);
        acc_relu_add<2>(acc, bias + P::T_B_SSKIP, c.half);
        acc_to_vec<TIER, 2, 2, false>(acc, vn, 0);
        rec_vals<TIER>(c, RecMap::S_D0 + 64 * 7, vn, 0);
        vs = vn;
    }
    layer<TIER, 2, P::KU_D, 2, true>(vn, ve, bias + P::T_B_E4, f, fe, s, c, DFN_RD(8));  ve = vn;
    layer<TIER, 2, P::KU_D, 2, true>(vn, vs, bias + P::T_B_S4, f, fe, s, c, DFN_RD(9));  vs = vn;
#undef DFN_RD
    // p = deform(p) + p  (decoder.py:299): PE column j sits in the register where the GEMM leaves output
    // j (identity slot map); the signal half of the residual is folded into the SO bias by the fold kernel.
    Vec<TIER, 4> pd;      // tiles 0,1: deformed PE (60 valid); tiles 2,3: deformed pose signal (42 valid)
    {
        f32x16 acc[2];
        acc_init<2>(acc, bias + P::T_B_EO, c.half);
        gemm_group<TIER, 2, P::KU_D, 2>(acc, ve, f, fe, s, c);
#pragma unroll
        for (int L = 0; L < 32; ++L) pd.set(L, acc[L >> 4][L & 15] + pe.get(L));
        acc_init<2>(acc, bias + P::T_B_SO, c.half);
        gemm_group<TIER, 2, P::KU_D, 2>(acc, vs, f, fe, s, c);
#pragma unroll
        for (int L = 0; L < 32; ++L) pd.set(32 + L, acc[L >> 4][L & 15]);
    }
    rec_vec<TIER, 4>(c, RecMap::S_PD, pd);
    Vec<TIER, 8> act;
    layer<TIER, 8, P::KU_PD, 4, true>(act, pd, bias + P::T_B_IN, f, fe, s, c, RecMap::S_TRUNK + RecMap::T_A0,
                                      RecMap::S_MTRUNK + RecMap::TM_A0);
    return mlp_trunk<TIER, 4, P::KU_PD>(act, pd, dhat, bias, P::T_B_L1, P::T_B_SKIP, P::T_B_L5, P::T_B_VIEW,
                                        P::T_B_OUT, f % SLAB_FRAGS, fe, s, c, RecMap::S_TRUNK, RecMap::S_MTRUNK);
}

}  // namespace dfn
