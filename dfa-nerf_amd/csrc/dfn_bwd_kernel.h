// dfn_bwd_kernel.h - the MLP backward (dX chain) kernel and its launcher, as templates over the tier.
// Instantiated in two translation units (compile time): dfn_bwd_bf16.hip and dfn_train.hip (f32 tier).
#pragma once
#include <hip/hip_runtime.h>
#include "dfn_bwd.h"
#include "dfn_layout.h"
#include "dfn_mlp.h"
#include "dfn_train.h"

namespace dfn {

#ifdef DFN_TIMING     // developer build: per wave {total cycles, cycles in the hand-over waitcnt, in the barrier, 100 MHz ticks}
__device__ unsigned long long g_bwd_timing[6 * 8192];
#endif

// ================================================================================================
// MLP backward: one wave = one 32-point tile, one pass of the transposed weight stream
// ================================================================================================
template <int TIER, bool TORSO>
__global__ __launch_bounds__(TierCfg<TIER>::THREADS, TierCfg<TIER>::THREADS / 256) void mlp_bwd_kernel(
    const MlpBwdArgs A) {
    using C = TierCfg<TIER>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    lds_char* lds = (lds_char*)smem;
#ifndef DFN_BWD_ASMF
#define DFN_BWD_ASMF 0
#endif
    // LDS-DMA as asm: the ReLU mask loads must not drain the dY stores.  DFN_BWD_ASMF: the fragment reads as asm with counted
    // waits too (16-bit tiers), like the inference kernels
    typedef CtxT<false, (DFN_BWD_ASMF != 0), true> CtxB;
    const CtxB ctx = {lds, wave, lane, lane >> 5, {}};
    Stream s;
    s.base0 = s.base1 = A.wblob_T;
    s.nslab0 = s.nslab1 = A.nslab;
    s.sched = 0;
#ifdef DFN_TIMING
    s.t_wait = s.t_bar = s.t_issue = s.t_epi = 0;
    const unsigned long long tt0 = __builtin_readcyclecounter(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    stream_begin<TIER, use_asm_dma<TIER, CtxB>()>(s, lds, wave, lane);
    const long n_tiles = A.NP / 32;
    BwdIO io;
    io.dy_T = A.dy_T;
    io.masks = A.masks;
    io.rows = TORSO ? GradMap::S_ROWS : GradMap::H_ROWS;
    io.mask_dwords = TORSO ? RecMap::S_MDWORDS : RecMap::H_MDWORDS;
    __syncthreads();
    // One pass per workgroup by default (gridDim = tiles / 8).  DFN_BWD_PERSIST: the launcher sizes the grid to one workgroup
    // per compute unit and a workgroup walks tiles blockIdx, + gridDim, ... with the weight stream running on across its
    // passes - no launch / ring-fill / drain bubble between the two rounds a compute unit runs; measured neutral.
    for (long t8 = blockIdx.x; t8 * C::WAVES < n_tiles; t8 += gridDim.x) {
        const long tile_raw = t8 * C::WAVES + wave;
        const long tile = tile_raw < n_tiles ? tile_raw : n_tiles - 1;     // idle waves redo the last tile (same values)
        const long p = tile * 32 + (lane & 31);
        BwdIn in;
        {
            const int o = TORSO ? 4 : 0;
            const float* ds = A.dsamples + p * 8 + o;
            const float* sm = A.samples + p * 8 + o;
            in.dsigma = ds[0];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float y = sm[1 + k];
                in.dpre[k] = ds[1 + k] * (y * (1.0f - y));                 // sigmoid'
            }
        }
        io.pass = tile;
        if constexpr (TORSO) bwd_torso<TIER>(in, io, s, ctx);
        else bwd_head<TIER>(in, io, s, ctx);
    }
#ifdef DFN_TIMING
    if (lane == 0) {
        const long w = ((long)blockIdx.x * C::WAVES + wave) & 8191;
        g_bwd_timing[6 * w + 0] = __builtin_readcyclecounter() - tt0;
        g_bwd_timing[6 * w + 1] = s.t_wait;
        g_bwd_timing[6 * w + 2] = s.t_bar;
        g_bwd_timing[6 * w + 3] = __builtin_amdgcn_s_memrealtime() - tr0;
        g_bwd_timing[6 * w + 4] = s.t_issue;
        g_bwd_timing[6 * w + 5] = s.t_epi;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int TIER, bool TORSO> inline hipError_t launch_mlp_bwd_t(const MlpBwdArgs& A, hipStream_t st) {
    using C = TierCfg<TIER>;
    // (f32 torso kernel: + the parking area of the skip path's product, dfn_bwd.h: park_store)
    const int lds = RING_BYTES + ((TIER == TIER_F32 && TORSO) ? C::WAVES * PARK_BYTES_PER_WAVE : 0);
    static_assert(RING_BYTES + 4 * PARK_BYTES_PER_WAVE <= 160 * 1024, "LDS budget of the f32 torso dX kernel");
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_bwd_kernel<TIER, TORSO>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    const long n_tiles = A.NP / 32;
    int blocks = (int)((n_tiles + C::WAVES - 1) / C::WAVES);
#ifndef DFN_BWD_PERSIST
#define DFN_BWD_PERSIST 0      // measured: 213 -> 212 us (head), 237 -> 238 us (torso): the kernel is power-bound (LABNOTES.md 7)
#endif
    if (DFN_BWD_PERSIST) {
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
            if (cus <= 0) cus = 256;
        }
        // one workgroup per compute unit (the ring fills its LDS), each with the same number of passes
        const int rounds = (blocks + cus - 1) / cus;
        blocks = (blocks + rounds - 1) / rounds;
    }
    hipLaunchKernelGGL((mlp_bwd_kernel<TIER, TORSO>), dim3(blocks), dim3(C::THREADS), lds, st, A);
    return hipGetLastError();
}
}  // namespace dfn
