// dfn_wgrad_bf16.hip - weight gradients of the 16-bit training tier: dW = dY . X^T contracted over the sample points,
// one workgroup per (GEMM, slice of the points), operands shared through LDS, MX-fp8 operands on the block-scaled MFMA.
//
// Every GEMM of the two fields is at most 256 x 256 (dfn_plan.cpp: build_wgrad_plan), so one workgroup of 8 waves
// owns the WHOLE output of a GEMM for its slice of the points: each byte of dY / X is fetched from HBM once per
// GEMM that reads it.  The operands are what the forward's recorder and the dX chain left (dfn_mlp.h: "MX-fp8
// recording"): per 32-point tile [rows][32 points] e4m3 bytes + one E8M0 scale per 32-row block, i.e. exactly the
// operand format of v_mfma_scale_f32_32x32x64_f8f6f4 with the contraction index K = points: a K block of 32 (the unit a
// scale applies to) IS one tile, an instruction contracts two tiles, and the dequantisation is fused into the MFMA.
// Lane (m = l & 31, kh = l >> 5) of an A (B) fragment holds row m's points 16 kh .. + 15 of the first tile (bytes 0-15)
// and of the second tile (bytes 16-31); the scale of the first tile is taken from the lanes with kh = 0, of the second from
// kh = 1 (tools/mx_probe.hip pins this on the hardware).
// A 32-row block of one tile is ONE contiguous KiB, POINT-major ([point n][half h][register r], what the producers' lanes
// hold), which LDS-DMA (global_load_lds_dwordx4) copies linearly into a ring of steps without touching registers.  The
// transpose to the MFMA's operand layout is the LDS read: ds_read_b64_tr_b8 hands lane i of a 16-lane group byte i & 7 of
// the eight 8-byte rows named by the group's lanes of parity i >> 3 (tools/tr8_probe.hip).  Lane i names (point 16 kh + 8 q
// + (i >> 1), byte block i & 1 of half h = group & 1): lanes 0-7 of the group then receive registers 0-7, lanes 8-15
// registers 8-15 of eight points - MFMA row m = 16 h + r is feature tile_feat(h, r), a relabelling of the rows inside a
// 32-row block that the epilogue undoes.  Four reads (q = 0, 1 of both tiles) make a fragment.
// Differentiates decoder.py:277-349 (the Linear layers of both fields) like wgrad_kernel (dfn_train.hip), which
// stays for the f32 tier.
#include <hip/hip_runtime.h>
#include "dfn_bwd.h"
#include "dfn_layout.h"
#include "dfn_mlp.h"
#include "dfn_train.h"

namespace dfn {

constexpr int WL_WAVES = 8, WL_THREADS = 64 * WL_WAVES;
#ifndef DFN_WL_DEPTH
#define DFN_WL_DEPTH 4
#endif
// Cache policy of the operand stream: every byte is read once by one workgroup -> non-temporal (measured, both fields, alone:
// DMA stream only 339 -> 319 us, whole kernel 448 -> 438 us; sc1 / sc0 sc1: no change).  Also measured and NOT adopted: the
// arrays laid out [row block][tile] (sequential 1-KiB streams instead of 8-KiB runs at a 77-KB stride): the same 4.4-4.5 TB/s;
// every other piece through registers instead of LDS-DMA: the same - the stream is bound on the memory side, not by the
// LDS-DMA path of a compute unit.  Where the kernel's time goes (both fields, alone; timing builds DFN_WL_NOMFMA / DFN_WL_NOLDS):
// DMA stream + barriers 311 us, + the MFMAs on constant operands 386 us, + the transpose reads = the kernel, 439 us.  A
// register-double-buffered variant (reads of pair q + 1 under the MFMAs of pair q; needs 4 waves x 128 x 128 outputs to fit
// the registers) can at best reach the 386-us row: not built.  Built and measured out: the two waves of a SIMD out of phase
// (waves 4-7 multiply the fragments they read for the PREVIOUS pair, then read this one's, so that one wave's transpose reads run
// under its partner's MFMAs; no second fragment set) - 472 us instead of 444, the step 1.5 % slower.
#ifndef DFN_WL_POL
#define DFN_WL_POL 1
#endif
#if DFN_WL_POL == 1
#define DFN_WL_POLICY " nt"
#else
#define DFN_WL_POLICY ""
#endif
constexpr int WL_DEPTH = DFN_WL_DEPTH;          // ring depth in steps
constexpr int WL_STEP_BYTES = 32 * 1024;        // 32 operand tiles (1 KiB each) per step at most
constexpr int WL_PIECES = 4;                    // 1 KiB DMA pieces per wave and step at most (32 per step)
// The E8M0 scales of a chunk of the slice (16 bytes per 32-point tile: up to 8 dY row blocks + 8 X row blocks of the GEMM)
// are staged into LDS with ordinary loads BEFORE the chunk's DMA pipeline starts: an ordinary vector load inside the loop
// would make every wait on it drain the LDS-DMA pieces in flight (one in-order vmcnt queue).
constexpr int WL_CHUNK_PAIRS = 512;             // tile pairs per chunk: 1024 tiles x 16 B = 16 KiB of scales
constexpr int WL_SCALE_BYTES = 2 * WL_CHUNK_PAIRS * 16;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef int i32x2 __attribute__((ext_vector_type(2)));
// one MFMA fragment (8 dwords) of a 32-row block from its point-major LDS images of two tiles: p0 / p1 = this lane's
// transpose-read address in the first / second tile for q = 0 (see the header)
DFN_DEV i32x8 frag_tr8(const lds_char* p0, const lds_char* p1) {
    const i32x2 a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((DFN_LDS i32x2*)p0), b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((DFN_LDS i32x2*)(p0 + 256));
    const i32x2 c = __builtin_amdgcn_ds_read_tr8_b64_v2i32((DFN_LDS i32x2*)p1), d = __builtin_amdgcn_ds_read_tr8_b64_v2i32((DFN_LDS i32x2*)(p1 + 256));
    const i32x8 f = {a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
    return f;
}
// B fragment of a 32-row block of MX-fp4 activations (dfn_mlp.h, round 4): the block's LDS image is 512 bytes, [point][half][8
// bytes = 16 nibbles r = 0..15]; ds_read_b64_tr_b4 hands lane i of a 16-lane group nibble i of the sixteen 8-byte rows named by
// the group's lanes (tools/fp4_probe.hip): lane j names (point p0 + j, half h = group & 1), so lane i receives feature
// tile_feat(h, i) of 16 consecutive points - MFMA row m = 16 h + i, the same relabelling as the fp8 A operand's.  An fp4 operand
// lane (m, kh) holds ALL 32 points of tile kh (nibbles 0..31 = K 0..31 of K block kh): two reads of ONE tile, not one of each.
// p = this lane's row address for points 0..15 of its tile's block.
DFN_DEV i32x8 frag_tr4(const lds_char* p) {
    const i32x2 a = __builtin_amdgcn_ds_read_tr4_b64_v2i32((DFN_LDS i32x2*)p), b = __builtin_amdgcn_ds_read_tr4_b64_v2i32((DFN_LDS i32x2*)(p + 256));
    const i32x8 f = {a[0], a[1], b[0], b[1], 0, 0, 0, 0};
    return f;
}
template <int N> DFN_DEV void wl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// ---- the 256 x 256 GEMMs (8 of the ~14-29 GEMMs of a field, 73 % of its operand bytes) -----------------------------------
// Round 3's loop served every shape with run-time tile counts: per MFMA it executed 19 scalar and 11.6 other vector instructions
// (every fragment read and every MFMA sat in its own basic block behind a branch; 33 SGPRs spilled to VGPR lanes) and the matrix
// pipe was 19 % busy (profiles/r03e_c4_bf16_pmc.txt).  For M = N = 256 everything is a constant of the wave: a step is ONE tile
// pair = 2 x 16 operand blocks = 32 DMA pieces, four per wave - tile u = k >> 1, operand A / B = k & 1, block = the wave's index
// -; a wave owns 4 x 2 output tiles (row group rg = wave & 1, column group CG = wave >> 1, a template parameter) and the bias
// row sums of block 4 rg + CG; the E8M0 scales are selected by the MFMA's op_sel (byte i of the dword the lane read) instead
// of shifted; the DMA pieces are addressed as wave-uniform SGPR base + constant lane offset.
DFN_DEV void wl_dma(unsigned voff, const char* sbase, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" DFN_WL_POLICY ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
}
// stage the E8M0 scales of tile pairs [c0, c1) into LDS: [tile of the chunk][16] = A row blocks 0..7 | B row blocks 0..7
DFN_DEV void wl_stage_scales(lds_char* sc_lds, const unsigned char* scA, const unsigned char* scB, long strideA, long strideB,
                             long c0, long c1, int mts, int nts) {
    wl_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    for (int e = threadIdx.x; e < (int)(2 * (c1 - c0)) * 16; e += WL_THREADS) {
        const long t = 2 * c0 + (e >> 4);
        const int k = e & 15;
        unsigned char v = 127;
        if (k < 8) { if (k < mts) v = scA[t * strideA + k]; }
        else if (k - 8 < nts) v = scB[t * strideB + (k - 8)];
        *(DFN_LDS unsigned char*)(sc_lds + e) = v;
    }
    wl_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int CG, bool B4>
DFN_DEV void wl_full(lds_char* lds, const WOp& o, const void* dy_T, const void* act_T, long p0, long p1, int g_rows, int a_rows,
                     int ks, float* C, long c_stride, const int* e_of, float* dbias, int n_bias, int wave, int lane) {
    lds_char* sc_lds = lds + WL_DEPTH * WL_STEP_BYTES;
    const long strideA = rec8_tile_bytes(g_rows), strideB = act_tile_bytes(a_rows, B4);
    const int rg = wave & 1, kh = lane >> 5, hh = (lane >> 4) & 1, li = lane & 15;
    const bool do_bias = dbias && o.bias_owner;
    const unsigned char* scA = (const unsigned char*)dy_T + (long)g_rows * 32 + (o.a_row >> 5);
    const unsigned char* scB = (const unsigned char*)act_T + (long)a_rows * act_row_bytes(B4) + (o.b_row >> 5);
    const unsigned lds_base = (unsigned)(unsigned long)lds;
    // DMA pieces (1 KiB each).  dY: block `wave` of both tiles.  Activations, e4m3: block `wave` of both tiles (4 pieces per wave and
    // step); MX-fp4: a block is 512 bytes, a piece = a PAIR of blocks, 4 per tile - waves 0-3 fetch pair `wave` of tile 0,
    // waves 4-7 pair `wave - 4` of tile 1 (3 pieces per wave and step).  LDS image of tile u: dY blocks at u * 16 KiB + 1 KiB x
    // i, activation blocks at u * 16 KiB + 8 KiB + (1 KiB | 512 B) x j.
    constexpr int NPW = B4 ? 3 : 4;                 // pieces per wave and step
    const unsigned voffA = (unsigned)((o.a_row + 32 * wave) * 32 + 16 * lane);
    const unsigned voffB = B4 ? (unsigned)((o.b_row + 64 * (wave & 3)) * 16 + 16 * lane) : (unsigned)((o.b_row + 32 * wave) * 32 + 16 * lane);
    const unsigned dA = (unsigned)wave * 1024u;
    const unsigned dB = B4 ? (unsigned)(wave >> 2) * 16384u + 8192u + (unsigned)(wave & 3) * 1024u : (unsigned)(8 + wave) * 1024u;
    // this lane's transpose-read address (q = 0) in the wave's first A / B block of tile 0 of slot 0
    const lds_char* rdA = lds + (16 * kh + (li >> 1)) * 32 + hh * 16 + (li & 1) * 8 + 4 * rg * 1024;
    const lds_char* rdB = B4 ? lds + kh * 16384 + 8192 + 2 * CG * 512 + li * 16 + hh * 8        // (tile kh: see frag_tr4)
                                  : rdA + (8 + 2 * CG - 4 * rg) * 1024;
    f32x16 acc[4][2], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accb[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0][r] = acc[i][1][r] = 0.f;
    }
    i32x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 0x38383838;           // e4m3 1.0
    for (long c0 = p0, c1; c0 < p1; c0 = c1) {
        c1 = (c0 + WL_CHUNK_PAIRS < p1) ? c0 + WL_CHUNK_PAIRS : p1;
        const long n_steps = c1 - c0;                            // one tile pair per step
        // scales of the chunk: one (tile, operand) = 8 bytes per thread, its eight byte loads in flight together (the general
        // path's loop, a byte per thread and trip, is eight dependent HBM round trips for a 128-pair slice: ~15 us of ~150)
        wl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        for (int e = threadIdx.x; e < (int)(4 * (c1 - c0)); e += WL_THREADS) {
            const long t = 2 * c0 + (e >> 1);
            const unsigned char* src = (e & 1) ? scB + t * strideB : scA + t * strideA;
            unsigned lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo |= (unsigned)src[k] << (8 * k);
                hi |= (unsigned)src[4 + k] << (8 * k);
            }
            i32x2 v = {(int)lo, (int)hi};
            *(DFN_LDS i32x2*)(sc_lds + (e >> 1) * 16 + (e & 1) * 8) = v;
        }
        wl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* gA = (const char*)dy_T + 2 * c0 * strideA;
        const char* gB = (const char*)act_T + (2 * c0 + (B4 ? (wave >> 2) : 0)) * strideB;
        unsigned issue_off = 0;                                  // byte offset of the slot the next issue fills
        auto issue = [&]() {
            const unsigned m = lds_base + issue_off;
            wl_dma(voffA, gA, m + dA);
            wl_dma(voffA, gA + strideA, m + 16384u + dA);
            if constexpr (B4) {
                wl_dma(voffB, gB, m + dB);               // (gB already points at this wave's tile of the pair)
            } else {
                wl_dma(voffB, gB, m + dB);
                wl_dma(voffB, gB + strideB, m + 16384u + dB);
            }
            gA += 2 * strideA;
            gB += 2 * strideB;
            issue_off = issue_off + WL_STEP_BYTES == WL_DEPTH * WL_STEP_BYTES ? 0u : issue_off + WL_STEP_BYTES;
        };
#pragma unroll
        for (int s = 0; s < WL_DEPTH - 1; ++s)
            if (s < n_steps) issue();
        const lds_char* scp = sc_lds + kh * 16;
        unsigned rd_off = 0;
        for (long s = 0; s < n_steps; ++s) {
            if (s + WL_DEPTH - 2 < n_steps) wl_wait_vm<NPW * (WL_DEPTH - 2)>();
            else wl_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + WL_DEPTH - 1 < n_steps) issue();
            const lds_char* pa = rdA + rd_off;
            const lds_char* pb = rdB + rd_off;
            rd_off = rd_off + WL_STEP_BYTES == WL_DEPTH * WL_STEP_BYTES ? 0u : rd_off + WL_STEP_BYTES;
            const int sa4 = *(const DFN_LDS int*)(scp + 4 * rg);
            const int sb2 = *(const DFN_LDS unsigned short*)(scp + 8 + 2 * CG);
            scp += 32;
            i32x8 a[4], b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (B4) b[j] = frag_tr4(pb + j * 512);
                else b[j] = frag_tr8(pb + j * 1024, pb + j * 1024 + 16384);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag_tr8(pa + i * 1024, pa + i * 1024 + 16384);
#define WL_MM(I)                                                                                                         \
    acc[I][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[I], b[0], acc[I][0], 0, (B4 ? 4 : 0), I, sa4, 0, sb2);       \
    acc[I][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[I], b[1], acc[I][1], 0, (B4 ? 4 : 0), I, sa4, 1, sb2);
            WL_MM(0) WL_MM(1) WL_MM(2) WL_MM(3)
#undef WL_MM
            if (do_bias) accb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[CG], ones, accb, 0, 0, CG, sa4, 0, 127);
        }
    }
    // epilogue: as the general path (one writer per element and slice; MFMA row m of a block = feature tile_feat(m >> 4, m & 15))
    auto feat_of = [](int m) { return tile_feat(m >> 4, m & 15); };
    if (do_bias && (lane & 31) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = e_of[o.a_row + 32 * (4 * rg + CG) + feat_of(tile_feat(lane >> 5, r))];
            if (e >= 0) dbias[(long)ks * n_bias + e] = accb[r];
        }
    }
    float* c = C + (long)ks * c_stride + o.c_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (4 * rg + i) + feat_of(tile_feat(lane >> 5, r)), col = 32 * (2 * CG + j) + feat_of(lane & 31);
                c[(long)row * 256 + col] = acc[i][j][r];
            }
        }
}

// ---- the other fixed shapes: 256 x {128, 64, 32}, 32 x 256, 64 x 64 -------------------------------------------------------------
// tools/wl_trace.py (round 4, per-workgroup clocks): with the 256 x 256 GEMMs on wl_full (0.91 us per 32-KiB step, the head field at
// 5.3 TB/s) the launch waited for the shapes the general loop still served - 256 x 64 / 128: 1.06-1.1 us per step; the torso's
// fifteen 64 x 64 GEMMs of the deformation field: 2.5 us per step, because with 2 x 2 output tiles ONE wave did all the fragment
// reads and MFMAs of a step's four tile pairs while seven waited at the barrier - 240 of the torso's 432 workgroups, the whole second
// round of the launch.  Here the eight waves form an RGN x CGN x KWN grid: RGN x CGN over the output tiles (TM x TN per wave) and
// KWN over the tile PAIRS of a step (64 x 64: 2 x 1 x 4 - every wave one pair and one row tile per step); the KWN partial sums of
// an output tile are added through LDS at the end, in wave order (fixed: bit-reproducible like everything else here).
template <int MTS, int NTS, int RGN, int CGN, int KWN, bool B4>
DFN_DEV void wl_static(lds_char* lds, const WOp& o, const void* dy_T, const void* act_T, long p0, long p1, int g_rows, int a_rows,
                       int ks, float* C, long c_stride, const int* e_of, float* dbias, int n_bias, int wave, int lane) {
    static_assert(RGN * CGN * KWN == WL_WAVES && MTS % RGN == 0 && NTS % CGN == 0, "wave grid");
    constexpr int TM = MTS / RGN, TN = NTS / CGN;
    // 1-KiB pieces per tile: one per dY block; the activation blocks one each (e4m3) or in pairs of 512-byte blocks (MX-fp4; an
    // odd last block takes the following 512 bytes along - inside the array: TrainBuffers pads act_T by a tile)
    constexpr int NTB = B4 ? (NTS + 1) / 2 : NTS, NTL = MTS + NTB;
    constexpr int BBLK = B4 ? 512 : 1024;                          // LDS bytes of an activation block
    constexpr int PPS = NTL <= 4 ? 4 : (NTL <= 8 ? 2 : 1);              // tile pairs per step (<= 32 KiB)
    constexpr int NP = PPS * 2 * NTL, NWM = (NP + WL_WAVES - 1) / WL_WAVES;      // DMA pieces per step; per wave at most
    static_assert(PPS % KWN == 0 && NP <= 32 && TM <= 4 && TN <= 4, "shape");
    lds_char* sc_lds = lds + WL_DEPTH * WL_STEP_BYTES;
    const long strideA = rec8_tile_bytes(g_rows), strideB = act_tile_bytes(a_rows, B4);
    const int rg = wave % RGN, cg = (wave / RGN) % CGN, kw = wave / (RGN * CGN);
    const int kh = lane >> 5, hh = (lane >> 4) & 1, li = lane & 15;
    const bool do_bias = dbias && o.bias_owner && cg == 0;              // the row sums of a dY block: by the waves of column group 0
    const unsigned char* scA = (const unsigned char*)dy_T + (long)g_rows * 32 + (o.a_row >> 5);
    const unsigned char* scB = (const unsigned char*)act_T + (long)a_rows * act_row_bytes(B4) + (o.b_row >> 5);
    const unsigned lds_base = (unsigned)(unsigned long)lds;
    // this wave's DMA pieces: piece p = wave + 8 k of the step = (tile u of the step, operand block tl)
    unsigned voff[NWM], dst[NWM];
    int sub[NWM];
    bool isb[NWM];
    const int n_w = (NP - wave + WL_WAVES - 1) / WL_WAVES;
#pragma unroll
    for (int k = 0; k < NWM; ++k) {
        const int p = wave + WL_WAVES * k;
        const int u = p / NTL, tl = p - u * NTL;
        isb[k] = tl >= MTS;
        voff[k] = isb[k] ? (unsigned)((o.b_row + (B4 ? 64 : 32) * (tl - MTS)) * act_row_bytes(B4) + 16 * lane)
                         : (unsigned)((o.a_row + 32 * tl) * 32 + 16 * lane);
        sub[k] = u;
        dst[k] = (unsigned)((u * NTL + tl) * 1024);
    }
    const lds_char* rd0 = lds + (16 * kh + (li >> 1)) * 32 + hh * 16 + (li & 1) * 8;      // transpose-read address, q = 0, block 0
    const lds_char* rd4 = lds + li * 16 + hh * 8;                                          // ... of an MX-fp4 block (frag_tr4)
    f32x16 acc[TM][TN], accb[TM];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            accb[i][r] = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j][r] = 0.f;
        }
    i32x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 0x38383838;           // e4m3 1.0
    for (long c0 = p0, c1; c0 < p1; c0 = c1) {
        c1 = (c0 + WL_CHUNK_PAIRS < p1) ? c0 + WL_CHUNK_PAIRS : p1;
        const long n_steps = (c1 - c0 + PPS - 1) / PPS;
        // scales of the chunk: one (tile, operand) per thread, its byte loads in flight together
        wl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        for (int e = threadIdx.x; e < (int)(4 * (c1 - c0)); e += WL_THREADS) {
            const long t = 2 * c0 + (e >> 1);
            const bool b = e & 1;
            const unsigned char* src = b ? scB + t * strideB : scA + t * strideA;
            unsigned lo = 0x7f7f7f7fu, hi = 0x7f7f7f7fu;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < (b ? NTS : MTS)) {
                    unsigned& w = k < 4 ? lo : hi;
                    w = (w & ~(0xffu << (8 * (k & 3)))) | ((unsigned)src[k] << (8 * (k & 3)));
                }
            i32x2 v = {(int)lo, (int)hi};
            *(DFN_LDS i32x2*)(sc_lds + (e >> 1) * 16 + (e & 1) * 8) = v;
        }
        wl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        unsigned issue_off = 0;
        long tt = 2 * c0;                                        // first tile of the next step to issue
        const long t_last = 2 * c1 - 1;
        auto issue = [&]() {
            const unsigned m = lds_base + issue_off;
#pragma unroll
            for (int k = 0; k < NWM; ++k)
                if (k < n_w) {
                    long t = tt + sub[k];
                    t = t < t_last ? t : t_last;             // ragged last step: refetch the last tile (never multiplied)
                    const char* sb = isb[k] ? (const char*)act_T + t * strideB : (const char*)dy_T + t * strideA;
                    wl_dma(voff[k], sb, m + dst[k]);
                }
            tt += 2 * PPS;
            issue_off = issue_off + WL_STEP_BYTES == WL_DEPTH * WL_STEP_BYTES ? 0u : issue_off + WL_STEP_BYTES;
        };
#pragma unroll
        for (int s = 0; s < WL_DEPTH - 1; ++s)
            if (s < n_steps) issue();
        unsigned rd_off = 0;
        for (long s = 0; s < n_steps; ++s) {
            if (s + WL_DEPTH - 2 < n_steps) {
                switch (n_w) {
                    case 1: wl_wait_vm<1 * (WL_DEPTH - 2)>(); break;
                    case 2: wl_wait_vm<2 * (WL_DEPTH - 2)>(); break;
                    case 3: wl_wait_vm<3 * (WL_DEPTH - 2)>(); break;
                    default: wl_wait_vm<4 * (WL_DEPTH - 2)>(); break;
                }
            } else {
                wl_wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + WL_DEPTH - 1 < n_steps) issue();
            const lds_char* slot = rd0 + rd_off;
            const lds_char* slot4 = rd4 + rd_off;
            rd_off = rd_off + WL_STEP_BYTES == WL_DEPTH * WL_STEP_BYTES ? 0u : rd_off + WL_STEP_BYTES;
            const long pp = c0 + s * PPS;
#pragma unroll
            for (int uu = 0; uu < PPS / KWN; ++uu) {
                const int u = kw + KWN * uu;                     // this wave's tile pair of the step
                if (pp + u >= c1) break;
                const lds_char* sc = sc_lds + (2 * (pp + u - c0) + kh) * 16;
                int sa, sb;
                if constexpr (TM == 1) sa = *(const DFN_LDS unsigned char*)(sc + rg);
                else if constexpr (TM == 2) sa = *(const DFN_LDS unsigned short*)(sc + 2 * rg);
                else sa = *(const DFN_LDS int*)(sc + 4 * rg);
                if constexpr (TN == 1) sb = *(const DFN_LDS unsigned char*)(sc + 8 + cg);
                else if constexpr (TN == 2) sb = *(const DFN_LDS unsigned short*)(sc + 8 + 2 * cg);
                else sb = *(const DFN_LDS int*)(sc + 8 + 4 * cg);
                const lds_char* b0 = slot + (2 * u) * NTL * 1024;       // first tile of the pair
                const lds_char* b1 = b0 + NTL * 1024;                   // second tile
                i32x8 a[TM], b[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (B4)          // tile kh of the pair, its activation block TN cg + j
                        b[j] = frag_tr4(slot4 + (2 * u + kh) * NTL * 1024 + MTS * 1024 + (TN * cg + j) * BBLK);
                    else
                        b[j] = frag_tr8(b0 + (MTS + TN * cg + j) * 1024, b1 + (MTS + TN * cg + j) * 1024);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = frag_tr8(b0 + (TM * rg + i) * 1024, b1 + (TM * rg + i) * 1024);
#define WL_M1(I, J) if constexpr (I < TM && J < TN) acc[I][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[I], b[J], acc[I][J], 0, (B4 ? 4 : 0), I, sa, J, sb);
#define WL_MR(I) WL_M1(I, 0) WL_M1(I, 1) WL_M1(I, 2) WL_M1(I, 3)
                WL_MR(0) WL_MR(1) WL_MR(2) WL_MR(3)
#undef WL_MR
#undef WL_M1
                if (do_bias) {
#define WL_B1(I) if constexpr (I < TM) accb[I] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[I], ones, accb[I], 0, 0, I, sa, 0, 127);
                    WL_B1(0) WL_B1(1) WL_B1(2) WL_B1(3)
#undef WL_B1
                }
            }
        }
    }
    auto feat_of = [](int m) { return tile_feat(m >> 4, m & 15); };
    if constexpr (KWN > 1) {
        // add the KWN partial sums of every output tile through LDS (the ring is idle: every piece has landed and been read)
        wl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        float DFN_LDS* red = (float DFN_LDS*)lds;
        constexpr int PER_WAVE = (TM * TN + TM) * 16 * 64;               // floats: [tile or bias tile][register][lane]
        static_assert(PER_WAVE * 4 * WL_WAVES <= WL_DEPTH * WL_STEP_BYTES, "the partial sums fit the (idle) ring");
        if (kw > 0) {
            float DFN_LDS* mine = red + (long)wave * PER_WAVE;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[((TM * TN + i) * 16 + r) * 64 + lane] = accb[i][r];
            }
        }
        __builtin_amdgcn_s_barrier();
        if (kw > 0) return;
#pragma unroll
        for (int q = 1; q < KWN; ++q) {
            const float DFN_LDS* other = red + (long)(wave + q * RGN * CGN) * PER_WAVE;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += other[((i * TN + j) * 16 + r) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 16; ++r) accb[i][r] += other[((TM * TN + i) * 16 + r) * 64 + lane];
            }
        }
    }
    if (do_bias && (lane & 31) == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = e_of[o.a_row + 32 * (TM * rg + i) + feat_of(tile_feat(lane >> 5, r))];
                if (e >= 0) dbias[(long)ks * n_bias + e] = accb[i][r];
            }
    }
    float* c = C + (long)ks * c_stride + o.c_off;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (TM * rg + i) + feat_of(tile_feat(lane >> 5, r)), col = 32 * (TN * cg + j) + feat_of(lane & 31);
                c[(long)row * (32 * NTS) + col] = acc[i][j][r];
            }
}

#ifdef DFN_WL_TRACE          // developer build (tools/build_variant.sh ... -DDFN_WL_TRACE): per-workgroup start / end on the 100-MHz clock
__device__ unsigned long long* g_wl_trace = nullptr;
struct WlTraceScope {
    unsigned long long t0;
    int op, ks;
    __device__ WlTraceScope(int op_, int ks_) : t0(__builtin_amdgcn_s_memrealtime()), op(op_), ks(ks_) {}
    __device__ ~WlTraceScope() {
        if (g_wl_trace && threadIdx.x == 0) {
            unsigned long long* r = g_wl_trace + 4 * blockIdx.x;
            r[0] = t0; r[1] = __builtin_amdgcn_s_memrealtime(); r[2] = (unsigned long long)op; r[3] = (unsigned long long)ks;
        }
    }
};
#endif
template <bool B4>
__global__ __launch_bounds__(WL_THREADS) void wgrad_mx_kernel(const WOp* ops, const WItem* items, const void* __restrict__ dy_T,
                                                               const void* __restrict__ act_T, long n_tiles, int g_rows, int a_rows,
                                                               float* C, long c_stride, const int* e_of,
                                                               float* dbias, int n_bias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* lds = (lds_char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one workgroup = slice ks of ksplit of one GEMM; the split is PER GEMM, proportional to its operand rows (launch_wgrad_bf16)
    const WItem it = items[blockIdx.x];
    const WOp o = ops[it.op];
    const int ks = it.ks, ksplit = it.n;
#ifdef DFN_WL_TRACE
    WlTraceScope trace_scope(it.op, ks);
#endif
    const long pairs = n_tiles / 2;                                     // (NP is a multiple of 512: n_tiles is even)
    const long per = (pairs + ksplit - 1) / ksplit;
    const long p0 = ks * per, p1 = (p0 + per < pairs) ? p0 + per : pairs;      // this slice's tile PAIRS
    if (p0 >= p1) return;                                               // whole workgroup
#ifndef DFN_WL_NOFULL
    if (o.M == 256 && o.N == 256) {                                     // (whole workgroup) the specialised path
        switch (wave >> 1) {
            case 0: wl_full<0, B4>(lds, o, dy_T, act_T, p0, p1, g_rows, a_rows, ks, C, c_stride, e_of, dbias, n_bias, wave, lane); break;
            case 1: wl_full<1, B4>(lds, o, dy_T, act_T, p0, p1, g_rows, a_rows, ks, C, c_stride, e_of, dbias, n_bias, wave, lane); break;
            case 2: wl_full<2, B4>(lds, o, dy_T, act_T, p0, p1, g_rows, a_rows, ks, C, c_stride, e_of, dbias, n_bias, wave, lane); break;
            default: wl_full<3, B4>(lds, o, dy_T, act_T, p0, p1, g_rows, a_rows, ks, C, c_stride, e_of, dbias, n_bias, wave, lane); break;
        }
        return;
    }
#define WL_ARGS lds, o, dy_T, act_T, p0, p1, g_rows, a_rows, ks, C, c_stride, e_of, dbias, n_bias, wave, lane
    if (o.M == 256 && o.N == 128) { wl_static<8, 4, 4, 2, 1, B4>(WL_ARGS); return; }
    if (o.M == 256 && o.N == 64) { wl_static<8, 2, 8, 1, 1, B4>(WL_ARGS); return; }
    if (o.M == 256 && o.N == 32) { wl_static<8, 1, 8, 1, 1, B4>(WL_ARGS); return; }
    if (o.M == 32 && o.N == 256) { wl_static<1, 8, 1, 8, 1, B4>(WL_ARGS); return; }
    if (o.M == 64 && o.N == 64) { wl_static<2, 2, 2, 1, 4, B4>(WL_ARGS); return; }
#undef WL_ARGS
#endif
    const int mts = o.M / 32, nts = o.N / 32, ntl = mts + nts;          // operand tiles per 32 points
    const int pps = ntl <= 4 ? 4 : (ntl <= 8 ? 2 : 1);                  // tile pairs per step (<= 32 KiB)
    const int np = pps * 2 * ntl;                                       // DMA pieces per step, <= 32
    const int n_w = (np - wave + WL_WAVES - 1) / WL_WAVES;              // ... of which this wave issues n_w (1..4)
    const long strideA = rec8_tile_bytes(g_rows), strideB = act_tile_bytes(a_rows, B4);
    lds_char* sc_lds = lds + WL_DEPTH * WL_STEP_BYTES;                  // [tile of the chunk][16]: A row blocks 0..7 | B row blocks 0..7

    // ---- this wave's DMA pieces: the same (tile of the step, operand tile) every step ------------------------
    const char* src[WL_PIECES];       // per lane: address of its 16 bytes in tile 0
    long stride[WL_PIECES];           // bytes per tile of that operand array
    int sub[WL_PIECES];               // tile of the step (0 .. 2 pps - 1)
    unsigned dst[WL_PIECES];          // LDS offset inside the step's slot
    {
#pragma unroll
        for (int k = 0; k < WL_PIECES; ++k) {
            const int p = wave + WL_WAVES * k;
            const int u = p / ntl, tl = p - u * ntl;            // tile of the step, operand tile
            const bool isb = tl >= mts;
            const int r = isb ? o.b_row + 32 * (tl - mts) : o.a_row + 32 * tl;       // first row of the block: 1 KiB, copied linearly
            // (an MX-fp4 activation block is 512 bytes: this general loop - shapes without a static path - still moves a KiB
            // per block and takes the following 512 bytes along, inside the array: TrainBuffers pads act_T by a tile)
            src[k] = (const char*)(isb ? act_T : dy_T) + (long)r * (isb ? act_row_bytes(B4) : 32) + 16 * lane;
            stride[k] = isb ? strideB : strideA;
            sub[k] = u;
            dst[k] = (unsigned)((u * ntl + tl) * 1024);
        }
    }
    const unsigned lds_base = (unsigned)(unsigned long)lds;
    unsigned issue_slot = 0;                         // steps are issued in order: slot of the next issue = step % WL_DEPTH
    long c0 = p0, c1 = p0;                           // the chunk of tile pairs being processed
    auto issue = [&](long s) {                       // DMA of step s of the chunk into slot s % WL_DEPTH
        const unsigned slot = lds_base + issue_slot * WL_STEP_BYTES;
        issue_slot = issue_slot + 1 == WL_DEPTH ? 0u : issue_slot + 1;
        const long tt = 2 * (c0 + s * pps);
#pragma unroll
        for (int k = 0; k < WL_PIECES; ++k)
            if (k < n_w) {
                long t = tt + sub[k];
                t = t < 2 * c1 ? t : 2 * c1 - 1;     // ragged last step: refetch the last tile (never multiplied)
                const char* a = src[k] + t * stride[k];
                const unsigned d = __builtin_amdgcn_readfirstlane(slot + dst[k]);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" DFN_WL_POLICY ::"v"(a), "s"(d) : "memory", "m0");
            }
    };

    // ---- this wave's part of the output: rows 128 rg .. + 127, columns 64 cg .. + 63 -------------------------
    const int rg = wave & 1, cg = wave >> 1;
    const int mt_n = max(0, min(4, mts - 4 * rg)), nt_n = max(0, min(2, nts - 2 * cg));
    // bias gradients = row sums of dY: operand tile 4 rg + cg times a tile of ones, one more MFMA per tile pair and wave
    const bool do_bias = dbias && o.bias_owner && cg < mt_n;
    // operand reads (transpose reads, header): this lane names point 16 kh + 8 q + (i >> 1), half h, byte block i & 1
    const int kh = lane >> 5, hh = (lane >> 4) & 1, li = lane & 15;
    const int rd = (16 * kh + (li >> 1)) * 32 + hh * 16 + (li & 1) * 8;         // q = 0; q = 1: + 8 points = + 256 bytes
    // scale bytes in memory: [tile][rows x 32 | row block]; the MFMA takes the first tile's from the lanes kh = 0, the
    // second's from kh = 1
    const unsigned char* scA = (const unsigned char*)dy_T + (long)g_rows * 32 + (o.a_row >> 5);
    const unsigned char* scB = (const unsigned char*)act_T + (long)a_rows * act_row_bytes(B4) + (o.b_row >> 5);
    f32x16 acc[4][2], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accb[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0][r] = acc[i][1][r] = 0.f;
    }
    i32x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = 0x38383838;           // e4m3 1.0

    for (c0 = p0; c0 < p1; c0 = c1) {
        c1 = (c0 + WL_CHUNK_PAIRS < p1) ? c0 + WL_CHUNK_PAIRS : p1;
        const long n_steps = (c1 - c0 + pps - 1) / pps;
        // ---- stage this chunk's scales (nothing of the previous chunk is in flight or being read any more)
        wl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        for (int e = threadIdx.x; e < (int)(2 * (c1 - c0)) * 16; e += WL_THREADS) {
            const long t = 2 * c0 + (e >> 4);
            const int k = e & 15;
            unsigned char v = 127;
            if (k < 8) { if (k < mts) v = scA[t * strideA + k]; }
            else if (k - 8 < nts) v = scB[t * strideB + (k - 8)];
            *(DFN_LDS unsigned char*)(sc_lds + e) = v;
        }
        wl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_slot = 0;
#pragma unroll
        for (int s = 0; s < WL_DEPTH - 1; ++s)
            if (s < n_steps) issue(s);
        unsigned rd_slot = 0;
        for (long s = 0; s < n_steps; ++s) {
            // my pieces of step s have landed (only those of the next WL_DEPTH - 2 steps are younger) and my reads of
            // step s - 1 have returned; after the barrier that holds for every wave, and slot (s - 1) % WL_DEPTH is refilled
            if (s + WL_DEPTH - 2 < n_steps) {
                switch (n_w) {
                    case 1: wl_wait_vm<1 * (WL_DEPTH - 2)>(); break;
                    case 2: wl_wait_vm<2 * (WL_DEPTH - 2)>(); break;
                    case 3: wl_wait_vm<3 * (WL_DEPTH - 2)>(); break;
                    default: wl_wait_vm<4 * (WL_DEPTH - 2)>(); break;
                }
            } else {
                wl_wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + WL_DEPTH - 1 < n_steps) issue(s + WL_DEPTH - 1);
            const lds_char* slot = lds + rd_slot * WL_STEP_BYTES;
            rd_slot = rd_slot + 1 == WL_DEPTH ? 0u : rd_slot + 1;
            const long pp = c0 + s * pps;
#ifdef DFN_WL_NOMFMA          // timing experiment (wrong results): the DMA stream and the barriers alone
            if (pp >= 0) continue;
#endif
            for (int u = 0; u < pps; ++u) {
                if (pp + u >= c1) break;
                // the scales this lane supplies: of the pair's first tile (kh = 0) or second tile (kh = 1)
                const lds_char* sc = sc_lds + (2 * (pp + u - c0) + kh) * 16;
                const unsigned sa4 = *(const DFN_LDS unsigned*)(sc + 4 * rg);
                const unsigned sb2 = *(const DFN_LDS unsigned short*)(sc + 8 + 2 * cg);
                const lds_char* b0 = slot + (2 * u) * ntl * 1024 + rd;          // first tile of the pair
                const lds_char* b1 = b0 + ntl * 1024;                           // second tile
                i32x8 a[4], b[2];
#ifdef DFN_WL_NOLDS          // timing experiment (wrong results): the MFMAs on constant operands, no fragment reads
                for (int i = 0; i < 4; ++i) a[i] = ones;
                for (int j = 0; j < 2; ++j) b[j] = ones;
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
#else
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < mt_n && (nt_n > 0 || (do_bias && i == cg))) a[i] = frag_tr8(b0 + (4 * rg + i) * 1024, b1 + (4 * rg + i) * 1024);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (j < nt_n) {
                        if constexpr (B4) b[j] = frag_tr4((kh ? b1 : b0) - rd + (mts + 2 * cg + j) * 1024 + (lane & 15) * 16 + ((lane >> 4) & 1) * 8);
                        else b[j] = frag_tr8(b0 + (mts + 2 * cg + j) * 1024, b1 + (mts + 2 * cg + j) * 1024);
                    }
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (i < mt_n && j < nt_n)
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 0, (B4 ? 4 : 0), 0, (int)(sa4 >> (8 * i)), 0,
                                                                                        (int)(sb2 >> (8 * j)));
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i == cg) accb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], ones, accb, 0, 0, 0, (int)(sa4 >> (8 * i)), 0, 127);
                }
            }
        }
    }

    // Split-K WITHOUT atomics: this workgroup owns slice `ks` of the partial arrays (C: [ksplit][c_stride], dbias:
    // [ksplit][n_bias]); every element of a slice has exactly one writer, and reduce_scatter_kernel / reduce_bias_kernel
    // add the slices in index order - the gradients are bit-reproducible run to run (float atomicAdd was not).
    // MFMA row / column m of a 32-row block = 16 h + r  <->  feature tile_feat(h, r) (the transpose reads' relabelling)
    auto feat_of = [](int m) { return tile_feat(m >> 4, m & 15); };
    if (do_bias && (lane & 31) == 0) {          // every column of accb holds the row sums: take column 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = e_of[o.a_row + 32 * (4 * rg + cg) + feat_of(tile_feat(lane >> 5, r))];
            if (e >= 0) dbias[(long)ks * n_bias + e] = accb[r];
        }
    }
    float* c = C + (long)ks * c_stride + o.c_off;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (i < mt_n && j < nt_n) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * (4 * rg + i) + feat_of(tile_feat(lane >> 5, r)), col = 32 * (2 * cg + j) + feat_of(lane & 31);
                    c[(long)row * o.N + col] = acc[i][j][r];
                }
            }
}

hipError_t launch_wgrad_bf16(int field, bool act_fp4, const WOp* ops_dev, const WItem* items_dev, int n_items, const void* dy_T,
                             const void* act_T, long NP, float* C, long c_stride, const int* e_of, float* dbias, int n_bias,
                             hipStream_t st) {
    constexpr int lds = WL_DEPTH * WL_STEP_BYTES + WL_SCALE_BYTES;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_mx_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wgrad_mx_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    if ((NP / 32) & 1) return hipErrorInvalidValue;          // tile pairs (the MFMA contracts two 32-point tiles)
    const bool torso = field == FIELD_TORSO;
    const int g_rows = torso ? GradMap::S_ROWS : GradMap::H_ROWS, a_rows = torso ? RecMap::S_ROWS : RecMap::H_ROWS;
    if (act_fp4)
        hipLaunchKernelGGL(wgrad_mx_kernel<true>, dim3(n_items), dim3(WL_THREADS), lds, st, ops_dev, items_dev, dy_T, act_T, NP / 32,
                           g_rows, a_rows, C, c_stride, e_of, dbias, n_bias);
    else
        hipLaunchKernelGGL(wgrad_mx_kernel<false>, dim3(n_items), dim3(WL_THREADS), lds, st, ops_dev, items_dev, dy_T, act_T, NP / 32,
                           g_rows, a_rows, C, c_stride, e_of, dbias, n_bias);
    return hipGetLastError();
}

}  // namespace dfn
#ifdef DFN_WL_TRACE
extern "C" int dfn_debug_wl_trace(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfn::g_wl_trace), &p, sizeof(p));
}
#endif
