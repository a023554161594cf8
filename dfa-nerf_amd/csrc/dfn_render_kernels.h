// dfn_render_kernels.h - fused frame renderer and fused decoder for gfx950 (MI355X): the kernel templates.
// Instantiated once per precision tier in dfn_render_{f32,bf16,f16}.hip (separate translation units: they build
// in parallel); dfn_render.hip dispatches on the tier.
//
// One wavefront = one ray.  A workgroup of W waves (8 in the bf16 tier, 4 in the f32 tier) walks W rays
// through: ray generation -> 64 coarse samples -> [head MLP (+ torso MLP)] on 32-sample tiles ->
// online alpha compositing -> (optional) inverse-CDF fine sampling, rank merge -> the same MLPs on the
// merged 64+n_fine samples -> compositing -> 12 (or 24) bytes of RGB per ray.  Nothing but the packed
// weight stream, the per-frame bias blob and one background pixel per ray is read from memory.
//
// Reference semantics (paths under /root/reference/NeRFs/DFANeRF/):
//   rays            run_nerf_helpers.py:449-465 (get_rays)
//   coarse z        run_nerf_com_trainExpLater.py:612-619, 638-641
//   decoder         decoder.py:277-349, 109-134       (dfn_mlp.h)
//   bg / sigma fix  run_nerf_com_trainExpLater.py:669-671, 678-679, 688-694
//   composite       run_nerf_com_trainExpLater.py:146-166 (composite_function)
//   weights         run_nerf_com_trainExpLater.py:169-179 (calc_volume_weights), 706-709
//   fine sampling   run_nerf_helpers.py:537-581 (sample_pdf); composition = SURVEY.md 8(a) row H
#pragma once
#include <hip/hip_runtime.h>
#include "dfn_layout.h"
#include "dfn_mlp.h"
#include "dfn_params.h"

namespace dfn {

// ---- exact-rounding helpers (no fma contraction: the reference is separate f32 mul/add) -------------
DFN_DEV float mul_(float a, float b) { return __fmul_rn(a, b); }
DFN_DEV float add_(float a, float b) { return __fadd_rn(a, b); }
DFN_DEV float sub_(float a, float b) { return __fsub_rn(a, b); }
DFN_DEV float div_(float a, float b) { return __fdiv_rn(a, b); }

// torch.linspace(0,1,n)[i] (see oracle/dfa_oracle.py:linspace01)
DFN_DEV float linspace01(int i, int n) {
    const float step = div_(1.0f, (float)(n - 1));
    return (i < n / 2) ? mul_(step, (float)i) : fmaf(-step, (float)(n - 1 - i), 1.0f);
}

DFN_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// pinhole ray through pixel `pix` (row-major y*W+x): get_rays, run_nerf_helpers.py:449-465
DFN_DEV void make_ray(const float* pose, int pix, int W, float focal, float cx, float cy, float (&o)[3],
                      float (&d)[3]) {
    const int y = pix / W, x = pix - y * W;
    const float dx = div_(sub_((float)x, cx), focal);
    const float dy = div_(-sub_((float)y, cy), focal);
    const float dz = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = add_(add_(mul_(dx, pose[4 * k + 0]), mul_(dy, pose[4 * k + 1])), mul_(dz, pose[4 * k + 2]));
        o[k] = pose[4 * k + 3];
    }
}

DFN_DEV float norm3(const float (&d)[3]) {
    return sqrtf(add_(add_(mul_(d[0], d[0]), mul_(d[1], d[1])), mul_(d[2], d[2])));
}

// ---- per-wave ray state, kept in LDS so that nothing but the sample point is live across an MLP pass ----
enum RayState : int {
    RS_OH = 0, RS_DH = 3, RS_OT = 6, RS_DT = 9, RS_NH = 12, RS_NT = 13, RS_DHAT_H = 14, RS_DHAT_T = 17,
    RS_BG = 20, RS_TH = 23, RS_TC = 24, RS_RGB_H = 25, RS_RGB_C = 28, RS_COUNT = 32
};

// One 32-sample tile of calc_volume_weights + the weighted colour sum
// (run_nerf_com_trainExpLater.py:169-179, 706-709).  sigma >= 0 already composited; dist already multiplied
// by the ray norm.  T (running transmittance) and rgb (running sums) live in LDS; returns the weight.
DFN_DEV float integrate_tile(volatile lds_f32* st, int rs_T, int rs_rgb, float sigma, float dist,
                             const float (&col)[3], int n, int lane) {
    const float alpha = sub_(1.0f, expf(-mul_(add_(fmaxf(sigma, 0.f), 1e-6f), dist)));
    const float v = add_(sub_(1.0f, alpha), 1e-10f);
    float inc = v;      // inclusive prefix product over the 32 samples of the tile
#pragma unroll
    for (int dlt = 1; dlt < 32; dlt <<= 1) {
        const float up = __shfl_up(inc, dlt, 32);
        if (n >= dlt) inc *= up;
    }
    float exc = __shfl_up(inc, 1, 32);
    if (n == 0) exc = 1.0f;
    const float T = st[rs_T];
    const float w = alpha * (T * exc);
    float part[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float x = w * col[k];
#pragma unroll
        for (int dlt = 16; dlt >= 1; dlt >>= 1) x += __shfl_xor(x, dlt, 32);
        part[k] = x;
    }
    const float Tn = T * __shfl(inc, 31, 32);
    const float r0 = st[rs_rgb + 0] + part[0], r1 = st[rs_rgb + 1] + part[1], r2 = st[rs_rgb + 2] + part[2];
    wave_lds_fence();
    if (lane == 0) {
        st[rs_T] = Tn;
        st[rs_rgb + 0] = r0;
        st[rs_rgb + 1] = r1;
        st[rs_rgb + 2] = r2;
    }
    wave_lds_fence();
    return w;
}

template <int TIER, bool TWO = false> struct KernelLds {
    using P = Prog<TIER>;
    static constexpr int BIAS_H = RING_BYTES;
    static constexpr int BIAS_T = BIAS_H + P::H_NBIAS * 4;
    static constexpr int SCRATCH = BIAS_T + P::T_NBIAS * 4;
    // per wave (floats): zall[192] | M[4][192] | rank8[128 bytes] | state[32]
    //   zall   sample depths: the 64 coarse z, then the merged, sorted 64 + n_fine
    //   M      per merged sample (sigma, r, g, b) of the field set being composited; while the coarse pass and
    //          sample_pdf run its first 256 floats hold tmp[64] cdf[64] zf[128]
    //   rank8  merged rank of fine sample j (u8)
    // (decoder_kernel keeps its per-lane d/|d| [3][64] at float 544 of the same area)
    //   keepc  (two-field kernel only) the coarse samples' two-field mix (ssum, fm) [4][64]
    static constexpr int Z_ALL = 0, M_OFF = 192, M_STRIDE = 192, RANK8 = 960, STATE = 992, KEEPC = 1024;
    static constexpr int PARK_H = 256;       // inside M: the coarse samples' head outputs [4][64] until the merge
    static constexpr int SCRATCH_FLOATS = TWO ? 1280 : 1024;
    static constexpr int SCRATCH_PER_WAVE = SCRATCH_FLOATS * 4;
    static constexpr int TOTAL = SCRATCH + TierCfg<TIER>::WAVES * SCRATCH_PER_WAVE;
    static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

// ================================================================================================
// Frame renderer
// ================================================================================================
// Pass order of one ray (= one wave): coarse tiles 0,1 (head, then torso when two fields are rendered), the
// fine sampler, then the decoder on the n_fine NEW points only - there is one network (SURVEY.md 8(a) row H,
// step 4), so its outputs at the 64 coarse points are kept from the coarse pass instead of being evaluated a
// second time: head on the fine tiles, compositing of the head image over the merged samples, torso on the fine
// tiles, compositing of the two-field image.  Every sample's (sigma, rgb) is bit-identical to what a pass over
// all 64 + n_fine merged points gives (an MFMA column depends on its own point only).
// TRAIN: 0 = inference; 1 = the training forward of the reference's step (MAIN:855-899: coarse samples only, recorder on);
// 2 = the hierarchical training forward (row H under autograd: the fine depths are constants - no gradient through
// sample_pdf, as in the NeRF lineage - and the loss sees the merged 64 + n_fine samples: every point is evaluated ONCE with
// the recorder on, the coarse points in the coarse pass, the fine points in the fine passes, recorded in evaluation
// order; the backward gets the merged depths and each point's merged rank to composite them in depth order)
// ACT4 (training forwards only): the recorder writes act_T as MX-fp4 (the default of the fused step) or as MX-fp8 e4m3 (the
// run-time opt-out: DFN_TRAIN_ACT_E4M3 or'ed into the tier of the dfn_train_fwd* calls; instantiated in dfn_render_bf16e.hip)
template <int TIER, bool TWO, int TRAIN, bool ACT4 = true>
__global__ __launch_bounds__(TierCfg<TIER>::THREADS, TierCfg<TIER>::THREADS / 256) void render_kernel(
    const RenderArgs A) {
    using C = TierCfg<TIER>;
    using L = KernelLds<TIER, TWO>;
    using P = Prog<TIER>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;
    const DfnFrame& F = A.frame;
    lds_char* lds = (lds_char*)smem;
    // inference: asm fragment fetch (DFN_ASM_FETCH); training: asm LDS-DMA only.  Pipelined layers in the head-only
    // 16-bit kernel (the two-field kernel has no 32 registers to spare)
#ifndef DFN_PIPE
#define DFN_PIPE 1
#endif
#ifndef DFN_PIPE_TWO         // 1: pipelined layers in the two-field kernel too.  Round 4, measured: 256 VGPRs + 15 spilled (60 B of scratch: the
                             // build's no-scratch check for the 16-bit inference kernels would refuse it), C3 72.39 -> 71.86 ms (-0.7 %,
                             // interleaved, three rounds): not adopted
#define DFN_PIPE_TWO 0
#endif
#ifndef DFN_PIPE_TWO_HEAD    // 1: in the two-field kernel the HEAD's passes run the pipelined layers, the torso's (whose deformation
                             // field and skip input hold the registers the second accumulator set needs) the plain ones: 256
                             // VGPRs, nothing spilled, the same bits; C3 70.62 -> 70.24-70.36 ms (-0.4 %, interleaved, three rounds -
                             // the kernel sits at its power ceiling, LABNOTES.md 4.6: a better schedule is paid back in clock)
#define DFN_PIPE_TWO_HEAD 1
#endif
    typedef CtxT<TRAIN != 0, TRAIN == 0, TRAIN != 0, (DFN_PIPE != 0) && TRAIN == 0 && (!TWO || DFN_PIPE_TWO != 0), TRAIN != 0 && ACT4> CtxK;      // (fused step: act_T in MX-fp4 unless ACT4 is off)
    CtxK ctx = {lds, wave, lane, lane >> 5, {}};
    typedef CtxT<TRAIN != 0, TRAIN == 0, TRAIN != 0, (DFN_PIPE != 0) && TRAIN == 0 && (!TWO || DFN_PIPE_TWO != 0 || DFN_PIPE_TWO_HEAD != 0), TRAIN != 0 && ACT4> CtxH;   // the head's passes
    constexpr bool two = TWO;
    const int NF = TRAIN == 1 ? 0 : F.n_fine;     // TRAIN == 1: the training forward is the reference's coarse renderer
    const bool hier = NF > 0;
    const int KF = NF / 32;                        // fine tiles
    // coarse samples per ray (--N_samples, MAIN:612-619): 64 in the scripts; 32 and 128 too when there is no fine pass (round 6:
    // the coarse loop walks NC / 32 tiles; the hierarchical sampler is a 64-lane wave program and keeps NC = 64, dfn_api.hip)
    const int NC = hier ? 64 : F.n_coarse;
    const int S = NC + NF;

    Stream s;
    s.base0 = A.wblob[0];
    s.base1 = A.wblob[1];
    s.nslab0 = A.nslab[0];
    s.nslab1 = A.nslab[1];
    // coarse: H T per coarse tile (NC / 32 of them: H T H T for the scripts' 64); fine: KF x H, then KF x T
    const int KC = (F.n_fine > 0 && TRAIN != 1) ? 2 : F.n_coarse / 32;
    s.sched = two ? ((0xAAu & ((1u << (2 * KC)) - 1u)) | (((1u << KF) - 1u) << (2 * KC + KF))) : 0u;
#ifdef DFN_TIMING
    s.t_wait = s.t_bar = s.t_issue = 0;
    unsigned long long T_mlp = 0, T_pdf = 0;
    const unsigned long long T_start = __builtin_readcyclecounter();
    const unsigned long long R_start = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef DFN_PRIO_YOUNG      // experiment (static priority for the second-dispatched half): no effect on C2, not enabled
    if (wave >= C::WAVES / 2) __builtin_amdgcn_s_setprio(1);
#endif
    const bool probe = A.clock_probe && blockIdx.x == gridDim.x / 2 && wave == 0;       // wave-uniform
    unsigned long long probe_c0 = 0, probe_r0 = 0;
    if (probe) {
        probe_c0 = __builtin_readcyclecounter();
        probe_r0 = __builtin_amdgcn_s_memrealtime();
    }
    stream_begin<TIER, use_asm_dma<TIER, CtxK>()>(s, lds, wave, lane);
    {
        lds_f32* bl = (lds_f32*)(lds + L::BIAS_H);
        const int nb = P::H_NBIAS + (TWO ? P::T_NBIAS : 0);
        for (int i = tid; i < nb; i += C::THREADS) bl[i] = A.bias[i];
    }
    const lds_f32* bias_h = (const lds_f32*)(lds + L::BIAS_H);
    const lds_f32* bias_t = (const lds_f32*)(lds + L::BIAS_T);

    lds_f32* scr = (lds_f32*)(lds + L::SCRATCH + wave * L::SCRATCH_PER_WAVE);
    lds_f32* zall = scr + L::Z_ALL;      // [192]
    lds_f32* zc = zall;                  // [64]  the coarse z (until the merge overwrites the area)
    lds_f32* M = scr + L::M_OFF;         // [4][192]
    lds_f32* tmp = M;                    // [64]  coarse weights, then pdf / bins
    lds_f32* cdf = M + 64;               // [64]
    lds_f32* zf = M + 128;               // [128]
    DFN_LDS unsigned char* rank8 = (DFN_LDS unsigned char*)(scr + L::RANK8);
    volatile lds_f32* st = scr + L::STATE;

    // ---- this wave's ray -----------------------------------------------------------------------
    const int r_raw = blockIdx.x * C::WAVES + wave;
    const bool valid = r_raw < F.ray_count;
    {
        const int r = valid ? r_raw : F.ray_count - 1;
        const int pix = A.pix_index ? A.pix_index[r] : F.ray_begin + r;
        float bg[3];
        if (A.bg_u8) {
#pragma unroll
            for (int k = 0; k < 3; ++k) bg[k] = div_((float)A.bg_u8[(size_t)pix * 3 + k], 255.0f);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) bg[k] = A.bg_f32[(size_t)pix * 3 + k];
        }
        float oh[3], dh[3], ot[3], dt[3];
        make_ray(F.pose, pix, F.W, F.focal, F.cx, F.cy, oh, dh);
        make_ray(F.pose_body, pix, F.W, F.focal, F.cx, F.cy, ot, dt);
        const float nh = norm3(dh), nt = norm3(dt);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                st[RS_OH + k] = oh[k];
                st[RS_DH + k] = dh[k];
                st[RS_OT + k] = ot[k];
                st[RS_DT + k] = dt[k];
                st[RS_DHAT_H + k] = div_(dh[k], nh);      // decoder.py:337
                st[RS_DHAT_T + k] = div_(dt[k], nt);
                st[RS_BG + k] = bg[k];
            }
            st[RS_NH] = nh;
            st[RS_NT] = nt;
            st[RS_TH] = 1.0f;
            st[RS_TC] = 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) st[RS_RGB_H + k] = st[RS_RGB_C + k] = 0.f;
        }
        // coarse z: near*(1-t) + far*t (run_nerf_com_trainExpLater.py:617-618)
        for (int i = lane; i < NC; i += 64) {
            const float t = linspace01(i, NC);
            zall[i] = add_(mul_(F.z_near, sub_(1.0f, t)), mul_(F.z_far, t));
        }
    }
    __syncthreads();     // bias blob + ray state visible (also drains the first two slab loads)

    const bool cbg = F.concate_bg != 0;
    const DhatRef dref_h = {st + RS_DHAT_H, 1}, dref_t = {st + RS_DHAT_T, 1};

    // head-image inputs of one sample: background colour / sigma bump on the last sample (:669-671, :693)
    auto head_inputs = [&](float sg_h, float (&fh)[3], bool last, float& s1) {
        if (cbg && last) { fh[0] = st[RS_BG]; fh[1] = st[RS_BG + 1]; fh[2] = st[RS_BG + 2]; }
        s1 = fmaxf(sg_h, 0.f);                     // head-only image: K = 1 (composite_function is a squeeze)
        if (cbg && last) s1 = add_(s1, 1e-6f);
    };
    // composite_function over {head, torso} for one sample (:158-162, :678-679, :694); fh after head_inputs
    auto combine = [&](float sg_h, const float (&fh)[3], float sg_t, const float (&ft)[3], bool last, float& ssum,
                       float (&fm)[3]) {
        if (cbg && last) sg_t = 0.f;
        const float sh = fmaxf(sg_h, 0.f);
        float stt = fmaxf(sg_t, 0.f);
        if (cbg && last) stt = add_(stt, 1e-6f);   // last stacked field
        ssum = add_(sh, stt);
        const float den = (ssum == 0.f) ? 1e-4f : ssum;
        const float wh = div_(sh, den), wt = div_(stt, den);
#pragma unroll
        for (int k = 0; k < 3; ++k) fm[k] = add_(mul_(fh[k], wh), mul_(ft[k], wt));
    };
    // calc_volume_weights + colour sum over the S merged samples held in M (4 rows: sigma, r, g, b)
    auto composite_merged = [&](bool head_image, float* w_out) {
        for (int t = 0; t < S / 32; ++t) {
            const int si = t * 32 + n;
            const float z = ((volatile lds_f32*)zall)[si];
            const bool last = (si == S - 1);
            const float znext = ((volatile lds_f32*)zall)[last ? si : si + 1];
            const float dz = last ? F.last_dist : sub_(znext, z);
            const volatile lds_f32* Mv = M;
            float sg = Mv[si], col[3] = {Mv[L::M_STRIDE + si], Mv[2 * L::M_STRIDE + si], Mv[3 * L::M_STRIDE + si]};
            float w;
            if (head_image) {
                float s1;
                head_inputs(sg, col, last, s1);
                w = integrate_tile(st, RS_TH, RS_RGB_H, s1, mul_(dz, st[RS_NH]), col, n, lane);
            } else {
                w = integrate_tile(st, RS_TC, RS_RGB_C, sg, mul_(dz, st[RS_NT]), col, n, lane);
            }
            if (valid && lane < 32) {
                if (w_out) w_out[(size_t)r_raw * S + si] = w;
                if (head_image && A.z_out) A.z_out[(size_t)r_raw * S + si] = z;
            }
        }
    };

    // what the coarse pass leaves for the merged compositing, parked in LDS so that nothing of it is live in
    // registers across the MLP passes: the coarse samples' head outputs (inside M until the merge scatters them to
    // their ranks) and, with two fields, their mix (ssum, fm)
    lds_f32* park_h = M + L::PARK_H;          // [4][64]
    lds_f32* keepc = scr + L::KEEPC;          // [4][64]  (allocated for the two-field kernel only)
    int rank_c = lane;

    enum { PH_COARSE = 0, PH_FINE_H = 1, PH_FINE_T = 2 };
    int phase = PH_COARSE, tile = 0;
    // one call site per MLP: the loop is a small state machine around them
    for (;;) {
        const int idx = tile * 32 + n;
        const int ri = (phase == PH_COARSE) ? idx : (int)rank8[idx];       // sample's index in zall
        const float z = ((volatile lds_f32*)zall)[ri];
        MlpOut a = {}, b = {};
        if (phase != PH_FINE_T) {
            float p[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = add_(st[RS_OH + k], mul_(st[RS_DH + k], z));
            if constexpr (TRAIN != 0) {      // idle waves (ray >= ray_count) record into the last ray's slots: same values
                // tiles of a ray in evaluation order: coarse 0, 1, then the fine tiles
                const long rr = valid ? r_raw : F.ray_count - 1;
                ctx.rec = {A.act_T[0], A.masks[0], RecMap::H_ROWS, rr * (S / 32) + (phase == PH_COARSE ? tile : 2 + tile),
                           RecMap::H_MDWORDS};
            }
#ifdef DFN_TIMING
            const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
            if constexpr (std::is_same<CtxH, CtxK>::value) {
                a = mlp_head<TIER>(p, dref_h, bias_h, s, ctx);
            } else {
                const CtxH ctx_h = {ctx.ring, ctx.wave, ctx.lane, ctx.half, ctx.rec};
                a = mlp_head<TIER>(p, dref_h, bias_h, s, ctx_h);
            }
#ifdef DFN_TIMING
            T_mlp += __builtin_readcyclecounter() - tm0;
#endif
        }
        if (two && phase != PH_FINE_H) {
            float p[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = add_(st[RS_OT + k], mul_(st[RS_DT + k], z));
            if constexpr (TRAIN != 0) {
                const long rr = valid ? r_raw : F.ray_count - 1;
                ctx.rec = {A.act_T[1], A.masks[1], RecMap::S_ROWS, rr * (S / 32) + (phase == PH_COARSE ? tile : 2 + tile),
                           RecMap::S_MDWORDS};
            }
            b = mlp_torso<TIER>(p, dref_t, bias_t, s, ctx);
        }

        if (phase == PH_COARSE) {
            const int si = idx;
            if (TRAIN != 0 && valid && lane < 32) {      // raw outputs, evaluation order: [ray][64 coarse | n_fine fine][8]
                float* so = A.samples_out + ((size_t)r_raw * S + si) * 8;
                so[0] = a.sigma; so[1] = a.r; so[2] = a.g; so[3] = a.b;
                so[4] = b.sigma; so[5] = b.r; so[6] = b.g; so[7] = b.b;
            }
            // results live in lanes 0..31; mirror them so that both halves run the same arithmetic
            const bool last = (si == NC - 1);
            const float znext = ((volatile lds_f32*)zall)[last ? si : si + 1];
            const float dz = last ? F.last_dist : sub_(znext, z);
            const float sg_h = __shfl(a.sigma, n);
            float fh[3] = {__shfl(a.r, n), __shfl(a.g, n), __shfl(a.b, n)};
            if (hier && lane < 32) {
                park_h[si] = sg_h; park_h[64 + si] = fh[0]; park_h[128 + si] = fh[1]; park_h[192 + si] = fh[2];
            }
            float s1;
            head_inputs(sg_h, fh, last, s1);
            const float w_h = integrate_tile(st, RS_TH, RS_RGB_H, s1, mul_(dz, st[RS_NH]), fh, n, lane);
            float w_c = 0.f;
            if (two) {
                const float sg_t = __shfl(b.sigma, n);
                const float ft[3] = {__shfl(b.r, n), __shfl(b.g, n), __shfl(b.b, n)};
                float ssum, fm[3];
                combine(sg_h, fh, sg_t, ft, last, ssum, fm);
                if (hier && lane < 32) {
                    keepc[si] = ssum; keepc[64 + si] = fm[0]; keepc[128 + si] = fm[1]; keepc[192 + si] = fm[2];
                }
                w_c = integrate_tile(st, RS_TC, RS_RGB_C, ssum, mul_(dz, st[RS_NT]), fm, n, lane);
            }
            if (hier) {
                if (lane < 32) tmp[si] = two ? w_c : w_h;
            } else if (valid && lane < 32) {
                if (A.w_head) A.w_head[(size_t)r_raw * NC + si] = w_h;
                if (A.w_com && two) A.w_com[(size_t)r_raw * NC + si] = w_c;
                if (A.z_out) A.z_out[(size_t)r_raw * NC + si] = z;
            }
            if (++tile < NC / 32) continue;
            if (!hier) break;
#ifdef DFN_TIMING
            const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
            // ---- sample_pdf(z_mid, weights[1:-1], n_fine, det=True), run_nerf_helpers.py:537-581 ----
            wave_lds_fence();
            const float wp = (lane >= 1 && lane <= 62) ? add_(tmp[lane], 1e-5f) : 0.f;
            // sum(weights + 1e-5) in the library's documented order (sample_pdf_kernel, oracle wave_sum64): element k
            // (= coarse weight k + 1) sits in lane k, lanes combine by a butterfly at XOR distances 32, 16, ..., 1
            float Ssum = __shfl(wp, (lane + 1) & 63);
#pragma unroll
            for (int dlt = 32; dlt >= 1; dlt >>= 1) Ssum += __shfl_xor(Ssum, dlt);
            const float pdf = div_(wp, Ssum);
            const float zmid = (lane < 63) ? mul_(0.5f, add_(zc[lane + 1], zc[lane])) : 0.f;
            // torch.cumsum as the reference's CPU path computes it: accumulator in DOUBLE (at::acc_type<float, false>),
            // every prefix rounded to f32 on output.  Here as a 6-step wave scan instead of the 62-step sequential loop,
            // bit-identical to it: every pdf entry is an f32 in [1e-5 / 1.0007, 1] (compositing weights are >= 0 and sum
            // to <= 1) and every prefix is < 2, i.e. a multiple of 2^-40 below 2 = 41 bits: double additions of them are
            // EXACT in any order.  (The stand-alone sample_pdf_kernel takes arbitrary weights and keeps the loop.)
            double c = (double)pdf;          // lanes 0 and 63 hold 0
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const double up = __shfl_up(c, dlt);
                if (lane >= dlt) c += up;
            }
            cdf[lane] = (lane <= 62) ? (float)c : 3.0e38f;
            wave_lds_fence();
            tmp[lane] = zmid;                // bins
            wave_lds_fence();
            for (int m = 0; m < NF / 64; ++m) {
                const int j = lane + 64 * m;
                const float u = linspace01(j, NF);
                int inds = 0;                // searchsorted(cdf[0..62], u, right=True)
#pragma unroll
                for (int stp = 32; stp >= 1; stp >>= 1) {
                    const int t = inds + stp;
                    if (t <= 63 && cdf[t - 1] <= u) inds = t;
                }
                const int below = max(inds - 1, 0), above = min(inds, 62);
                const float cb = cdf[below], ca = cdf[above];
                const float bb = tmp[below], ba = tmp[above];
                float den = sub_(ca, cb);
                if (den < 1e-5f) den = 1.0f;
                const float t = div_(sub_(u, cb), den);
                zf[j] = add_(bb, mul_(t, sub_(ba, bb)));
            }
            wave_lds_fence();
            // ---- z_all = sort(cat(z, z_fine)): ranks by counting, no sortedness assumption on z_fine ----
            const float myc = zc[lane];
            int rank_f[3] = {0, 0, 0};
            float myf[3] = {0, 0, 0};
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m < NF / 64) {
                    myf[m] = zf[lane + 64 * m];
                    int cnt = 0;             // coarse values <= mine (zc strictly increasing)
#pragma unroll
                    for (int stp = 64; stp >= 1; stp >>= 1) {
                        const int t = cnt + stp;
                        if (t <= 64 && zc[t - 1] <= myf[m]) cnt = t;
                    }
                    rank_f[m] = cnt;
                }
            }
            // The deterministic fine samples come out non-decreasing (inverse CDF of an increasing u) except for 1-ulp
            // effects of the lerp rounding at bin borders.  Sorted: a fine sample's rank among the fine samples is its
            // index, a coarse sample's is a binary search - instead of the NF-step counting loop (6 k of the 30 k cycles
            // this section used to take per ray).  Checked per ray; the counting loop stays as the general path.
            bool sorted_f = true;
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (m < NF / 64) {
                    const int j = lane + 64 * m;
                    if (j + 1 < NF && zf[j + 1] < myf[m]) sorted_f = false;
                }
            if (__builtin_amdgcn_ballot_w64(!sorted_f) == 0) {
#pragma unroll
                for (int m = 0; m < 3; ++m) rank_f[m] += lane + 64 * m;       // fine samples before mine: exactly j
                int cntf = 0;                                                   // fine values < my coarse value
#pragma unroll
                for (int stp = 128; stp >= 1; stp >>= 1) {
                    const int t = cntf + stp;
                    if (t <= NF && zf[t - 1] < myc) cntf = t;
                }
                rank_c += cntf;
            } else {
                for (int i = 0; i < NF; ++i) {
                    const float v = zf[i];
                    rank_c += (v < myc) ? 1 : 0;
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const int j = lane + 64 * m;
                        rank_f[m] += (v < myf[m] || (v == myf[m] && i < j)) ? 1 : 0;
                    }
                }
            }
            float keep_h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) keep_h[q] = park_h[q * 64 + lane];
            wave_lds_fence();                // every read of zc / zf / tmp / cdf / park_h is done: the areas are reused
            zall[rank_c] = myc;
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (m < NF / 64) {
                    zall[rank_f[m]] = myf[m];
                    rank8[lane + 64 * m] = (unsigned char)rank_f[m];
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) M[q * L::M_STRIDE + rank_c] = keep_h[q];
            if constexpr (TRAIN == 2) {      // merged rank of every evaluated point, for the compositing backward
                if (valid) {
                    unsigned char* ro = A.ranks_out + (size_t)r_raw * S;
                    ro[lane] = (unsigned char)rank_c;
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        if (m < NF / 64) ro[64 + lane + 64 * m] = (unsigned char)rank_f[m];
                }
            }
            if (lane == 0) {                 // the merged compositing starts from scratch
                st[RS_TH] = 1.0f;
                st[RS_TC] = 1.0f;
#pragma unroll
                for (int k = 0; k < 3; ++k) st[RS_RGB_H + k] = st[RS_RGB_C + k] = 0.f;
            }
            wave_lds_fence();
#ifdef DFN_TIMING
            T_pdf += __builtin_readcyclecounter() - tp0;
#endif
            phase = PH_FINE_H;
            tile = 0;
        } else if (phase == PH_FINE_H) {
            if (TRAIN == 2 && valid && lane < 32) {
                float* so = A.samples_out + ((size_t)r_raw * S + 64 + idx) * 8;
                so[0] = a.sigma; so[1] = a.r; so[2] = a.g; so[3] = a.b;
            }
            if (lane < 32) {
                M[ri] = a.sigma;
                M[L::M_STRIDE + ri] = a.r;
                M[2 * L::M_STRIDE + ri] = a.g;
                M[3 * L::M_STRIDE + ri] = a.b;
            }
            if (++tile < KF) continue;
            wave_lds_fence();
            composite_merged(true, A.w_head);
            if (!two) break;
            phase = PH_FINE_T;
            tile = 0;
        } else {
            if (TRAIN == 2 && valid && lane < 32) {
                float* so = A.samples_out + ((size_t)r_raw * S + 64 + idx) * 8;
                so[4] = b.sigma; so[5] = b.r; so[6] = b.g; so[7] = b.b;
            }
            if (lane < 32) {                 // the sample's head outputs are in M: replace them by the two-field mix
                const volatile lds_f32* Mv = M;
                const float sg_h = Mv[ri];
                float fh[3] = {Mv[L::M_STRIDE + ri], Mv[2 * L::M_STRIDE + ri], Mv[3 * L::M_STRIDE + ri]};
                const bool last = (ri == S - 1);
                float s1, ssum, fm[3];
                head_inputs(sg_h, fh, last, s1);
                const float ft[3] = {b.r, b.g, b.b};
                combine(sg_h, fh, b.sigma, ft, last, ssum, fm);
                M[ri] = ssum;
                M[L::M_STRIDE + ri] = fm[0];
                M[2 * L::M_STRIDE + ri] = fm[1];
                M[3 * L::M_STRIDE + ri] = fm[2];
            }
            if (++tile < KF) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) M[q * L::M_STRIDE + rank_c] = keepc[q * 64 + lane];
            wave_lds_fence();
            composite_merged(false, A.w_com);
            break;
        }
    }
#ifdef DFN_TIMING
    if (valid && lane == 0 && A.z_out) {
        const unsigned long long T_end = __builtin_readcyclecounter();
        const unsigned long long R_end = __builtin_amdgcn_s_memrealtime();
        float* o = A.z_out + (size_t)r_raw * S + 64;
        o[0] = (float)(T_end - T_start); o[1] = (float)(R_end - R_start); o[2] = (float)T_mlp; o[3] = (float)T_pdf;
        o[4] = (float)s.t_wait; o[5] = (float)s.t_bar; o[6] = (float)s.t_issue; o[7] = (float)wave;
    }
#endif
    if (valid && lane == 0) {
        if (A.out_u8) {      // to8b (run_nerf_helpers.py:17): (255 * clip(x, 0, 1)) truncated, straight from the epilogue
            unsigned char* o8h = (unsigned char*)A.rgb_head;
            unsigned char* o8c = (unsigned char*)A.rgb_com;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o8h[(size_t)r_raw * 3 + k] = (unsigned char)(int)mul_(255.0f, fminf(fmaxf(st[RS_RGB_H + k], 0.f), 1.f));
                if (two && o8c)
                    o8c[(size_t)r_raw * 3 + k] = (unsigned char)(int)mul_(255.0f, fminf(fmaxf(st[RS_RGB_C + k], 0.f), 1.f));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                A.rgb_head[(size_t)r_raw * 3 + k] = st[RS_RGB_H + k];
                if (two && A.rgb_com) A.rgb_com[(size_t)r_raw * 3 + k] = st[RS_RGB_C + k];
            }
        }
    }
    if constexpr (TRAIN != 0) {
        // dfn_train_fwd_loss: the step's loss and d loss / d rgb from the epilogue (include/dfanerf.h).  Ray: its target pixel,
        // its row of d_rgb, its squared error; workgroup: its rays' in ray order; the last workgroup to get here (ticket in
        // the caller's workspace): the workgroups' in workgroup order.
        if (A.loss.losses) {                                         // (uniform over the launch)
            const int total = 3 * F.ray_count;
            if (lane == 0) {
                float se_h = 0.f, se_c = 0.f;
                if (valid) {
                    const int pix = A.pix_index ? A.pix_index[r_raw] : F.ray_begin + r_raw;
                    const float scale = div_(2.0f, (float)total);
                    unsigned char th[3], tc[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        th[k] = A.loss.img_head[(size_t)pix * 3 + k];
                        tc[k] = A.loss.img_com[(size_t)pix * 3 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float dh = sub_(st[RS_RGB_H + k], div_((float)th[k], 255.0f));
                        const float dc = sub_(st[RS_RGB_C + k], div_((float)tc[k], 255.0f));
                        se_h = add_(se_h, mul_(dh, dh));
                        se_c = add_(se_c, mul_(dc, dc));
                        A.loss.d_rgb_head[(size_t)r_raw * 3 + k] = mul_(scale, dh);
                        A.loss.d_rgb_com[(size_t)r_raw * 3 + k] = mul_(scale, dc);
                    }
                }
                st[RS_TH] = se_h;
                st[RS_TC] = se_c;
            }
            __syncthreads();
            const int nwg = gridDim.x;
            float* part = A.loss.workspace;
            unsigned* ticket = (unsigned*)(A.loss.workspace + 2 * nwg);
            volatile lds_f32* st0 = (lds_f32*)(lds + L::SCRATCH) + L::STATE;      // wave 0's state block
            if (tid == 0) {
                float a = 0.f, b = 0.f;
                for (int w = 0; w < C::WAVES; ++w) {
                    volatile lds_f32* sw = (lds_f32*)(lds + L::SCRATCH + w * L::SCRATCH_PER_WAVE) + L::STATE;
                    a = add_(a, sw[RS_TH]);
                    b = add_(b, sw[RS_TC]);
                }
                part[blockIdx.x] = a;
                part[nwg + blockIdx.x] = b;
                __threadfence();
                st0[RS_NH] = (atomicAdd(ticket, 1u) == (unsigned)nwg - 1u) ? 1.0f : 0.0f;
            }
            __syncthreads();
            if (st0[RS_NH] != 0.0f && wave == 0) {
                __threadfence();
                float a = 0.f, b = 0.f;
                for (int k = lane; k < nwg; k += 64) {            // lane l: workgroups l, l + 64, ... in sequence
                    a = add_(a, ((volatile float*)part)[k]);
                    b = add_(b, ((volatile float*)part)[nwg + k]);
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {               // then the lanes by a butterfly (same value in every lane)
                    a = add_(a, __shfl_xor(a, m, 64));
                    b = add_(b, __shfl_xor(b, m, 64));
                }
                if (lane == 0) {
                    const float lh = div_(a, (float)total), lc = div_(b, (float)total);
                    A.loss.losses[0] = lh;                          // img2mse(rgb_head, target_head)
                    A.loss.losses[1] = lc;                          // img2mse(rgb_com, target_com)
                    A.loss.losses[2] = add_(lc, lh);                // the step's loss (MAIN:902-907: loss_com + loss_head)
                    *ticket = 0u;                                   // ready for the next launch on this workspace
                }
            }
        }
    }
    if (probe && lane == 0) {
        A.clock_probe[0] = __builtin_readcyclecounter() - probe_c0;
        A.clock_probe[1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
    }
    // drain the prefetched slabs before the LDS allocation is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ================================================================================================
// Decoder.forward on explicit points (decoder.py:277-349): 32 points per wave, one MLP pass
// ================================================================================================
// REC: the training recorder is on (dfn_decoder_train_fwd): every GEMM input, the ReLU bits and the raw outputs are
// stored for the backward kernels, exactly as render_kernel<.., TRAIN> does for the points it generates itself.
template <int TIER, bool TORSO, bool REC = false>
__global__ __launch_bounds__(TierCfg<TIER>::THREADS, TierCfg<TIER>::THREADS / 256) void decoder_kernel(
    const DecoderArgs A) {
    using C = TierCfg<TIER>;
    using L = KernelLds<TIER>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;
    lds_char* lds = (lds_char*)smem;
    typedef CtxT<REC, false, REC> CtxD;
    CtxD ctx = {lds, wave, lane, lane >> 5, {}};
    ctx.rec.act_T = nullptr;
    ctx.rec.masks = nullptr;

    Stream s;
    s.base0 = s.base1 = A.wblob;
    s.nslab0 = s.nslab1 = A.nslab;
    s.sched = 0;
    stream_begin<TIER, use_asm_dma<TIER, CtxD>()>(s, lds, wave, lane);
    constexpr bool torso = TORSO;
    lds_f32* bias_l = (lds_f32*)(lds + (torso ? L::BIAS_T : L::BIAS_H));
    for (int i = tid; i < A.n_bias; i += C::THREADS) bias_l[i] = A.bias[i];

    lds_f32* scr = (lds_f32*)(lds + L::SCRATCH + wave * L::SCRATCH_PER_WAVE);
    volatile lds_f32* dl = scr + 544;      // per-lane d/|d|: [3][64]
    // idle waves (tile >= n_tiles) redo the last tile and idle lanes (point >= n) the last point: same values, so the
    // recorder's stores of a redone tile are harmless
    const long n_tiles = (A.n_points + 31) / 32;
    const long tile_raw = (long)blockIdx.x * C::WAVES + wave;
    const long tile = tile_raw < n_tiles ? tile_raw : n_tiles - 1;
    const long pt_raw = tile * 32 + n;
    const long pt = pt_raw < A.n_points ? pt_raw : A.n_points - 1;
    float p[3];
    {
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = A.points[pt * 3 + k];
            d[k] = A.dirs[pt * 3 + k];
        }
        const float nd = norm3(d);
#pragma unroll
        for (int k = 0; k < 3; ++k) dl[64 * k + lane] = div_(d[k], nd);
    }
    __syncthreads();
    const DhatRef dref = {dl + lane, 64};
    if constexpr (REC)
        ctx.rec = {A.act_T, A.masks, torso ? RecMap::S_ROWS : RecMap::H_ROWS, tile, torso ? RecMap::S_MDWORDS : RecMap::H_MDWORDS};
    MlpOut o;
    if constexpr (torso) o = mlp_torso<TIER>(p, dref, bias_l, s, ctx);
    else o = mlp_head<TIER>(p, dref, bias_l, s, ctx);
    if (lane < 32) {
        if (tile_raw < n_tiles && pt_raw < A.n_points) {
            A.feat[pt_raw * 3 + 0] = o.r;
            A.feat[pt_raw * 3 + 1] = o.g;
            A.feat[pt_raw * 3 + 2] = o.b;
            A.sigma[pt_raw] = o.sigma;
        }
        if constexpr (REC) {       // padding points of the last tile included (finite values; their gradients are zero)
            float* so = A.samples + pt_raw * 8 + (torso ? 4 : 0);
            so[0] = o.sigma; so[1] = o.r; so[2] = o.g; so[3] = o.b;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- per-tier launchers (instantiated by dfn_render_<tier>.hip)
template <typename K> static hipError_t set_lds(K kernel, int lds) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}
template <int TIER, bool TWO, int TRAIN = 0, bool ACT4 = true> static hipError_t launch_render_t(const RenderArgs& A, hipStream_t st) {
    using C = TierCfg<TIER>;
    const int lds = KernelLds<TIER, TWO>::TOTAL;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = set_lds(render_kernel<TIER, TWO, TRAIN, ACT4>, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int blocks = (A.frame.ray_count + C::WAVES - 1) / C::WAVES;
    hipLaunchKernelGGL((render_kernel<TIER, TWO, TRAIN, ACT4>), dim3(blocks), dim3(C::THREADS), lds, st, A);
    return hipGetLastError();
}
template <int TIER, bool TORSO, bool REC = false> static hipError_t launch_decoder_t(const DecoderArgs& A, hipStream_t st) {
    using C = TierCfg<TIER>;
    const int lds = KernelLds<TIER>::TOTAL;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = set_lds(decoder_kernel<TIER, TORSO, REC>, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const long per_block = (long)C::WAVES * 32;
    const int blocks = (int)((A.n_points + per_block - 1) / per_block);
    hipLaunchKernelGGL((decoder_kernel<TIER, TORSO, REC>), dim3(blocks), dim3(C::THREADS), lds, st, A);
    return hipGetLastError();
}

hipError_t launch_train_bf16_e4m3(const RenderArgs& A, hipStream_t st);      // dfn_render_bf16e.hip
// all launches of one tier; TRAINABLE: the tier has a training forward (recorder on)
template <int TIER, bool TRAINABLE> static hipError_t launch_render_tier(const RenderArgs& A, hipStream_t st) {
    const bool two = A.frame.fields == 2;
    if (A.samples_out) {     // training step: two fields, recorder on; coarse only (MAIN:855-899) or hierarchical (row H)
        if constexpr (TRAINABLE) {
            // 16-bit tier with the e4m3 opt-out for act_T: those two kernels are a translation unit of their own
            if (TIER == TIER_BF16 && A.act_e4m3) return launch_train_bf16_e4m3(A, st);
            return A.frame.n_fine > 0 ? launch_render_t<TIER, true, 2>(A, st) : launch_render_t<TIER, true, 1>(A, st);
        } else return hipErrorInvalidValue;
    }
    return two ? launch_render_t<TIER, true>(A, st) : launch_render_t<TIER, false>(A, st);
}
template <int TIER, bool TRAINABLE> static hipError_t launch_decoder_tier(const DecoderArgs& A, hipStream_t st) {
    if (A.act_T) {           // training forward on explicit points: recorder on
        if constexpr (TRAINABLE)
            return A.field == FIELD_TORSO ? launch_decoder_t<TIER, true, true>(A, st) : launch_decoder_t<TIER, false, true>(A, st);
        else return hipErrorInvalidValue;
    }
    return A.field == FIELD_TORSO ? launch_decoder_t<TIER, true>(A, st) : launch_decoder_t<TIER, false>(A, st);
}

}  // namespace dfn
