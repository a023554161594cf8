// dfn_wgrad_bf16.hip - weight gradients of the bf16 tier: dW = dY . X^T contracted over the sample points,
// one workgroup per (GEMM, slice of the points), operands shared through LDS.
//
// Every GEMM of the two fields is at most 256 x 256 (dfn_plan.cpp: build_wgrad_plan), so one workgroup of 8 waves
// owns the WHOLE output of a GEMM for its slice of the points: each byte of dY / X is fetched from HBM once per
// GEMM that reads it (wgrad_kernel<bf16> fetched the dY rows once per 128 output columns, per wave, into registers).
// The operands are tile-major ([tile of 32 points][rows][32], dfn_mlp.h: Rec): the rows a GEMM needs of one tile
// are ONE contiguous run, which LDS-DMA (global_load_lds_dwordx4) copies into a ring of 4 x 32 KiB steps without
// touching registers.  The DMA writes LDS linearly (lane i -> 16 bytes at 16 i) but every lane names its own source
// address: a piece is 16 rows x 64 bytes = 1 KiB of contiguous memory, four consecutive lanes fetch the four
// 16-byte chunks of one row (one 64-byte segment per quad: full address-coalescing rate; a first version with one
// row per lane ran at a quarter of it), in the order chunk = slot ^ ((row >> 2) & 3).  That XOR swizzle makes the
// MFMA operand reads (lane = row, all lanes the same chunk, i.e. a 64-byte stride) free of bank conflicts: the 16
// lanes of every ds_read_b128 lane group land on 16 different 16-byte slots of the 256-byte bank row.
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
constexpr int WL_DEPTH = DFN_WL_DEPTH;          // ring depth in steps (5 x 32 KiB = the compute unit's whole LDS)
constexpr int WL_STEP_BYTES = 32 * 1024;        // 16 operand tiles (512 rows x 32 points) per step at most
constexpr int WL_PIECES = 4;                    // 1 KiB DMA pieces per wave and step at most (32 per step)

template <int N> DFN_DEV void wl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(WL_THREADS) void wgrad_lds_kernel(const WOp* ops, const int* order, const void* dy_T,
                                                                const void* act_T, long n_tiles, int g_rows, int a_rows,
                                                                int ksplit, float* C, long c_stride, const int* e_of,
                                                                float* dbias, int n_bias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* lds = (lds_char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WOp o = ops[order[blockIdx.x / ksplit]];
    const int ks = blockIdx.x % ksplit;
    const long per = (n_tiles + ksplit - 1) / ksplit;
    const long t0 = ks * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    if (t0 >= t1) return;                                               // whole workgroup
    const int mts = o.M / 32, nts = o.N / 32, ntl = mts + nts;          // operand tiles per 32 points
    const int tps = ntl <= 4 ? 4 : (ntl <= 8 ? 2 : 1);                  // 32-point tiles per step (<= 32 KiB)
    const int np = tps * ntl * 2;                                       // DMA pieces per step, <= 32
    const int n_w = (np - wave + WL_WAVES - 1) / WL_WAVES;              // ... of which this wave issues n_w (1..4)
    const long n_steps = (t1 - t0 + tps - 1) / tps;

    // ---- this wave's DMA pieces: the same (sub-tile, operand tile, k-half) every step -----------------------
    const char* src[WL_PIECES];       // per lane: address of its 16 bytes in tile 0
    long stride[WL_PIECES];           // bytes per tile of that operand array
    int sub[WL_PIECES];               // sub-tile of the step
    unsigned dst[WL_PIECES];          // LDS offset inside the step's slot
#pragma unroll
    for (int k = 0; k < WL_PIECES; ++k) {
        const int p = wave + WL_WAVES * k;
        const int u = p / (ntl * 2), rem = p - u * (ntl * 2);
        const int tl = rem >> 1, hf = rem & 1;                  // operand tile, upper / lower 16 rows
        const bool isb = tl >= mts;
        const int r = 16 * hf + (lane >> 2);                    // row of the tile
        const int row = (isb ? o.b_row + 32 * (tl - mts) : o.a_row + 32 * tl) + r;
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        src[k] = (const char*)(isb ? act_T : dy_T) + ((long)row * 32 + 8 * chunk) * 2;
        stride[k] = (long)(isb ? a_rows : g_rows) * 64;
        sub[k] = u;
        dst[k] = (unsigned)(((u * ntl + tl) * 2 + hf) * 1024);
    }
    const unsigned lds_base = (unsigned)(unsigned long)lds;
    unsigned issue_slot = 0;                         // steps are issued in order: slot of the next issue = step % WL_DEPTH
    auto issue = [&](long s) {                       // DMA of step s into slot s % WL_DEPTH
        const unsigned slot = lds_base + issue_slot * WL_STEP_BYTES;
        issue_slot = issue_slot + 1 == WL_DEPTH ? 0u : issue_slot + 1;
        const long tt = t0 + s * tps;
#pragma unroll
        for (int k = 0; k < WL_PIECES; ++k)
            if (k < n_w) {
                long t = tt + sub[k];
                t = t < t1 ? t : t1 - 1;             // ragged last step: refetch the last tile (never multiplied)
                const char* a = src[k] + t * stride[k];
                const unsigned d = __builtin_amdgcn_readfirstlane(slot + dst[k]);
#ifdef DFN_WL_NT
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(a), "s"(d) : "memory", "m0");
#else
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(d) : "memory", "m0");
#endif
            }
    };

    // ---- this wave's part of the output: rows 128 rg .. + 127, columns 64 cg .. + 63 -------------------------
    const int rg = wave & 1, cg = wave >> 1;
    const int mt_n = max(0, min(4, mts - 4 * rg)), nt_n = max(0, min(2, nts - 2 * cg));
    // bias gradients = row sums of dY: operand tile 4 rg + cg times a tile of ones, two MFMAs per 32 points and wave
    const bool do_bias = dbias && o.bias_owner && cg < mt_n;
    // operand reads: MFMA lane l holds row l & 31, points 8 (l >> 5) + 16 kh .. + 7 = chunk (l >> 5) + 2 kh of the row
    const int rd_sw = (lane >> 5) ^ ((lane >> 2) & 3);
    const int rd0 = (lane & 31) * 64 + rd_sw * 16, rd1 = (lane & 31) * 64 + (rd_sw ^ 2) * 16;
    f32x16 acc[4][2], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accb[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0][r] = acc[i][1][r] = 0.f;
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

#pragma unroll
    for (int s = 0; s < WL_DEPTH - 1; ++s)
        if (s < n_steps) issue(s);
    unsigned rd_slot = 0;
    for (long s = 0; s < n_steps; ++s) {
        // my pieces of step s have landed (only those of the next WL_DEPTH - 2 steps are younger) and my reads of
        // step s - 1 have returned; after the barrier that holds for every wave, and slot (s - 1) % 4 is refilled
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
        const long tt = t0 + s * tps;
#ifdef DFN_WL_NOMFMA          // timing experiment (wrong results): the DMA stream and the barriers alone
        if (tt >= 0) continue;
#endif
        for (int u = 0; u < tps; ++u) {
            if (tt + u >= t1) break;
            const lds_char* base = slot + u * ntl * 2048;
            bf16x8 a[4][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < mt_n && (nt_n > 0 || (do_bias && i == cg))) {
                    a[i][0] = *(const DFN_LDS bf16x8*)(base + (4 * rg + i) * 2048 + rd0);
                    a[i][1] = *(const DFN_LDS bf16x8*)(base + (4 * rg + i) * 2048 + rd1);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (j < nt_n) {
                    b[j][0] = *(const DFN_LDS bf16x8*)(base + (mts + 2 * cg + j) * 2048 + rd0);
                    b[j][1] = *(const DFN_LDS bf16x8*)(base + (mts + 2 * cg + j) * 2048 + rd1);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (i < mt_n && j < nt_n) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    }
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i == cg) {
                        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], ones, accb, 0, 0, 0);
                        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], ones, accb, 0, 0, 0);
                    }
            }
        }
    }

    // Split-K WITHOUT atomics: this workgroup owns slice `ks` of the partial arrays (C: [ksplit][c_stride], dbias:
    // [ksplit][n_bias]); every element of a slice has exactly one writer, and reduce_scatter_kernel / reduce_bias_kernel
    // add the slices in index order - the gradients are bit-reproducible run to run (float atomicAdd was not).
    if (do_bias && (lane & 31) == 0) {          // every column of accb holds the row sums: take column 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = e_of[o.a_row + 32 * (4 * rg + cg) + tile_feat(lane >> 5, r)];
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
                    const int row = 32 * (4 * rg + i) + tile_feat(lane >> 5, r), col = 32 * (2 * cg + j) + (lane & 31);
                    c[(long)row * o.N + col] = acc[i][j][r];
                }
            }
}

hipError_t launch_wgrad_bf16(int field, const WOp* ops_dev, const int* order_dev, int n_ops, const void* dy_T,
                             const void* act_T, long NP, int ksplit, float* C, long c_stride, const int* e_of,
                             float* dbias, int n_bias, hipStream_t st) {
    constexpr int lds = WL_DEPTH * WL_STEP_BYTES;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    const bool torso = field == FIELD_TORSO;
    const int g_rows = torso ? GradMap::S_ROWS : GradMap::H_ROWS, a_rows = torso ? RecMap::S_ROWS : RecMap::H_ROWS;
    hipLaunchKernelGGL(wgrad_lds_kernel, dim3(n_ops * ksplit), dim3(WL_THREADS), lds, st, ops_dev, order_dev, dy_T, act_T,
                       NP / 32, g_rows, a_rows, ksplit, C, c_stride, e_of, dbias, n_bias);
    return hipGetLastError();
}

}  // namespace dfn
