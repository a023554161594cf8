// dfn_bwd_kernel.h - the MLP backward (dX chain) kernel and its launcher, as templates over the tier.
// Instantiated in two translation units (compile time): dfn_bwd_bf16.hip and dfn_train.hip (f32 tier).
#pragma once
#include <hip/hip_runtime.h>
#include "dfn_bwd.h"
#include "dfn_layout.h"
#include "dfn_mlp.h"
#include "dfn_train.h"

namespace dfn {

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
    stream_begin<TIER, use_asm_dma<TIER, CtxB>()>(s, lds, wave, lane);
    const long n_tiles = A.NP / 32;
    const long tile_raw = (long)blockIdx.x * C::WAVES + wave;
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
    BwdIO io;
    io.dy_T = A.dy_T;
    io.masks = A.masks;
    io.rows = TORSO ? GradMap::S_ROWS : GradMap::H_ROWS;
    io.pass = tile;
    io.mask_dwords = TORSO ? RecMap::S_MDWORDS : RecMap::H_MDWORDS;
    __syncthreads();
    if constexpr (TORSO) bwd_torso<TIER>(in, io, s, ctx);
    else bwd_head<TIER>(in, io, s, ctx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int TIER, bool TORSO> inline hipError_t launch_mlp_bwd_t(const MlpBwdArgs& A, hipStream_t st) {
    using C = TierCfg<TIER>;
    const int lds = RING_BYTES;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_bwd_kernel<TIER, TORSO>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    const long n_tiles = A.NP / 32;
    const int blocks = (int)((n_tiles + C::WAVES - 1) / C::WAVES);
    hipLaunchKernelGGL((mlp_bwd_kernel<TIER, TORSO>), dim3(blocks), dim3(C::THREADS), lds, st, A);
    return hipGetLastError();
}
}  // namespace dfn
