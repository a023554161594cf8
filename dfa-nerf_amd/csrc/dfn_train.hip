// dfn_train.hip - training kernels: MLP backward (dX chain), compositing backward, weight-gradient GEMMs,
// gradient scatter and bias gradients.  The forward of a training step is render_kernel with the recorder
// switched on (dfn_render.hip / dfn_mlp.h: Rec).
//
// Differentiates run_nerf_com_trainExpLater.py:855-907 (two fields, coarse samples, composite, weights,
// weighted colour sums) and decoder.py:277-349 / 109-134.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "dfn_bwd_kernel.h"

namespace dfn {

hipError_t launch_mlp_bwd(int tier, int field, const MlpBwdArgs& A, hipStream_t st) {
    const bool torso = field == FIELD_TORSO;
    if (tier == TIER_BF16) return launch_mlp_bwd_bf16(torso, A, st);      // dfn_bwd_bf16.hip
    return torso ? launch_mlp_bwd_t<TIER_F32, true>(A, st) : launch_mlp_bwd_t<TIER_F32, false>(A, st);
}
void bwd_program_info(int tier, int field, ProgramInfo* out) {
    const bool torso = field == FIELD_TORSO;
    if (tier == TIER_BF16) {
        using B = BProg<TIER_BF16>;
        *out = torso ? ProgramInfo{B::S_FRAGS, B::S_SLABS, 0} : ProgramInfo{B::H_FRAGS, B::H_SLABS, 0};
    } else {
        using B = BProg<TIER_F32>;
        *out = torso ? ProgramInfo{B::S_FRAGS, B::S_SLABS, 0} : ProgramInfo{B::H_FRAGS, B::H_SLABS, 0};
    }
}

// ================================================================================================
// compositing backward: one wave per ray, 64 coarse samples = 64 lanes
// ================================================================================================
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float up = __shfl_up(inc, d);
        if (lane >= d) inc *= up;
    }
    float exc = __shfl_up(inc, 1);
    return lane == 0 ? 1.0f : exc;
}
__device__ __forceinline__ float wave_suffix_sum_excl(float v, int lane) {      // sum over lanes > lane
    float inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float dn = __shfl_down(inc, d);
        if (lane + d < 64) inc += dn;
    }
    float exc = __shfl_down(inc, 1);
    return lane == 63 ? 0.f : exc;
}
// weights w_i = a_i T_i of run_nerf_com_trainExpLater.py:169-179 and dL/d(sigma_i) given q_i = c_i . G
struct WB {
    float w, ds;
};
__device__ __forceinline__ void composite_zero_fill(const CompositeBwdArgs& A) {
    if (!A.zero_buf) return;
    const long n4 = A.zero_floats >> 2, stride = (long)gridDim.x * blockDim.x;
    float4* q = (float4*)A.zero_buf;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && threadIdx.x < (A.zero_floats & 3)) A.zero_buf[(n4 << 2) + threadIdx.x] = 0.f;
}
// KL consecutive samples per lane (KL = 1: --N_samples 32 / 64, lanes beyond the ray's samples idle; KL = 2: 128), round 6.
// Sample i = lane KL + k.  The scans run over the lanes' local products / sums and are carried across a lane's KL positions;
// with KL = 1 and 64 samples the arithmetic is the round 1-5 kernel's.
template <int KL>
__device__ __forceinline__ void weights_bwd_n(const float (&sigma)[KL], const float (&dist)[KL], const float (&q)[KL],
                                              const bool (&live)[KL], int lane, WB (&out)[KL]) {
    float e[KL], a[KL], v[KL], T[KL], wq[KL];
    float lp = 1.0f;
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        e[k] = expf(-((fmaxf(sigma[k], 0.f) + 1e-6f) * dist[k]));
        a[k] = live[k] ? 1.0f - e[k] : 0.f;
        v[k] = live[k] ? 1.0f - a[k] + 1e-10f : 1.0f;
        lp *= v[k];
    }
    float Tb = wave_excl_prod(KL == 1 ? v[0] : lp, lane);
    float ls = 0.f;
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        T[k] = Tb;
        Tb *= v[k];
        out[k].w = a[k] * T[k];
        wq[k] = out[k].w * q[k];
        ls += wq[k];
    }
    float suf = wave_suffix_sum_excl(KL == 1 ? wq[0] : ls, lane);      // over the lanes behind this one
#pragma unroll
    for (int k = KL - 1; k >= 0; --k) {
        const float da = T[k] * q[k] - suf / v[k];
        out[k].ds = da * dist[k] * e[k];             // d a / d sigma = dist * exp(-(sigma + 1e-6) dist)
        suf += wq[k];
    }
}
template <int KL>
__global__ void composite_bwd_kernel(const CompositeBwdArgs A) {
    composite_zero_fill(A);
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= A.frame.ray_count) return;
    const DfnFrame& F = A.frame;
    const int NC = F.n_coarse;
    const int pix = A.pix_index ? A.pix_index[ray] : F.ray_begin + (int)ray;
    // geometry (same arithmetic as the forward)
    const int y = pix / F.W, x = pix - y * F.W;
    const float dx = __fdiv_rn(__fsub_rn((float)x, F.cx), F.focal), dy = __fdiv_rn(-__fsub_rn((float)y, F.cy), F.focal);
    float nrm[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float* P = b ? F.pose_body : F.pose;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, P[4 * k]), __fmul_rn(dy, P[4 * k + 1])),
                                      __fmul_rn(-1.0f, P[4 * k + 2]));
            acc = __fadd_rn(acc, __fmul_rn(d, d));
        }
        nrm[b] = sqrtf(acc);
    }
    const float step = __fdiv_rn(1.0f, (float)(NC - 1));
    auto tval = [&](int i) { return (i < NC / 2) ? __fmul_rn(step, (float)i) : fmaf(-step, (float)(NC - 1 - i), 1.0f); };
    auto zval = [&](int i) {
        const float t = tval(i);
        return __fadd_rn(__fmul_rn(F.z_near, __fsub_rn(1.0f, t)), __fmul_rn(F.z_far, t));
    };
    const bool cbg = F.concate_bg != 0;
    bool live[KL], last[KL], h_is_bg[KL];
    float dz[KL], sg_h[KL], sg_t_raw[KL], ch[KL][3], ct[KL][3], sh[KL];
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        const int i = lane * KL + k;
        live[k] = i < NC;
        const int ii = live[k] ? i : NC - 1;
        last[k] = ii == NC - 1;
        dz[k] = last[k] ? F.last_dist : __fsub_rn(zval(ii + 1), zval(ii));
        const float* sm = A.samples + (ray * NC + ii) * 8;
        sg_h[k] = sm[0];
        sg_t_raw[k] = sm[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ch[k][c] = sm[1 + c];
            ct[k][c] = sm[5 + c];
        }
        h_is_bg[k] = cbg && last[k];
        if (h_is_bg[k]) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                ch[k][c] = A.bg_u8 ? __fdiv_rn((float)A.bg_u8[(size_t)pix * 3 + c], 255.0f) : A.bg_f32[(size_t)pix * 3 + c];
        }
        sh[k] = fmaxf(sg_h[k], 0.f);
    }
    float Gh[3], Gc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Gh[k] = A.d_rgb_head[ray * 3 + k];
        Gc[k] = A.d_rgb_com ? A.d_rgb_com[ray * 3 + k] : 0.f;
    }
    float d_sg_h[KL], d_ch[KL][3], d_sg_t[KL], d_ct[KL][3];
    // ---- head-only image: s = relu(sg_h) (+1e-6 at the last sample), colour ch
    {
        float s1[KL], dist[KL], q[KL];
        WB r[KL];
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            s1[k] = h_is_bg[k] ? sh[k] + 1e-6f : sh[k];
            dist[k] = dz[k] * nrm[0];
            q[k] = ch[k][0] * Gh[0] + ch[k][1] * Gh[1] + ch[k][2] * Gh[2];
        }
        weights_bwd_n<KL>(s1, dist, q, live, lane, r);
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            d_sg_h[k] = (sg_h[k] > 0.f) ? r[k].ds : 0.f;
            d_sg_t[k] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d_ch[k][c] = h_is_bg[k] ? 0.f : r[k].w * Gh[c];
                d_ct[k][c] = 0.f;
            }
        }
    }
    // ---- composite image (run_nerf_com_trainExpLater.py:146-166)
    if (A.frame.fields == 2) {
        float ssum[KL], dist[KL], q[KL], sg_t[KL], st[KL], den[KL], wh[KL], wt[KL];
        bool zero[KL];
        WB r[KL];
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            sg_t[k] = (cbg && last[k]) ? 0.f : sg_t_raw[k];
            st[k] = fmaxf(sg_t[k], 0.f);
            if (cbg && last[k]) st[k] += 1e-6f;
            ssum[k] = sh[k] + st[k];
            zero[k] = ssum[k] == 0.f;
            den[k] = zero[k] ? 1e-4f : ssum[k];
            wh[k] = sh[k] / den[k];
            wt[k] = st[k] / den[k];
            float cm[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) cm[c] = ch[k][c] * wh[k] + ct[k][c] * wt[k];
            q[k] = cm[0] * Gc[0] + cm[1] * Gc[1] + cm[2] * Gc[2];
            dist[k] = dz[k] * nrm[1];
        }
        weights_bwd_n<KL>(ssum, dist, q, live, lane, r);
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            const float gch = ch[k][0] * Gc[0] + ch[k][1] * Gc[1] + ch[k][2] * Gc[2];
            const float gct = ct[k][0] * Gc[0] + ct[k][1] * Gc[1] + ct[k][2] * Gc[2];
            const float dwh = r[k].w * gch, dwt = r[k].w * gct;
            // wh = sh/den, wt = st/den, den = sh + st (or the constant 1e-4)
            float dsh = r[k].ds + dwh / den[k], dst = r[k].ds + dwt / den[k];
            if (!zero[k]) {
                const float common = (dwh * sh[k] + dwt * st[k]) / (den[k] * den[k]);
                dsh -= common;
                dst -= common;
            }
            if (sg_h[k] > 0.f) d_sg_h[k] += dsh;
            if (sg_t[k] > 0.f && !(cbg && last[k])) d_sg_t[k] += dst;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (!h_is_bg[k]) d_ch[k][c] += r[k].w * Gc[c] * wh[k];
                d_ct[k][c] += r[k].w * Gc[c] * wt[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KL; ++k) {
        if (!live[k]) continue;
        float* out = A.dsamples + (ray * NC + lane * KL + k) * 8;
        out[0] = d_sg_h[k];
        out[4] = d_sg_t[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            out[1 + c] = d_ch[k][c];
            out[5 + c] = d_ct[k][c];
        }
    }
}
hipError_t launch_composite_bwd(const CompositeBwdArgs& A, hipStream_t st) {
    const int blocks = (A.frame.ray_count + 3) / 4;
    if (A.frame.n_coarse == 128) hipLaunchKernelGGL(composite_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, A);
    else if (A.frame.n_coarse == 64 || A.frame.n_coarse == 32) hipLaunchKernelGGL(composite_bwd_kernel<1>, dim3(blocks), dim3(256), 0, st, A);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ================================================================================================
// compositing backward over the MERGED samples of the hierarchical training step (SURVEY.md 8(a) row H under autograd)
// ================================================================================================
// One wave per ray, S = 64 K merged samples (K = 2: 64 + 64, K = 3: 64 + 128): lane l owns the K consecutive merged
// positions l K .. l K + K - 1.  The forward left every evaluated point's raw outputs in EVALUATION order (coarse points,
// then the fine ones: the order of the recorded activations, which is what the dX chain walks), its merged rank, and the
// merged depths; the fine depths are constants (no gradient through sample_pdf), so this is the same differentiation as
// composite_bwd_kernel with run-time depths and the scans carried across a lane's K positions.
template <int K>
__global__ __launch_bounds__(256) void composite_bwd_hier_kernel(const CompositeBwdArgs A) {
    constexpr int S = 64 * K;
    __shared__ unsigned char inv_s[4][S];
    composite_zero_fill(A);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * 4 + wv;
    if (ray >= A.frame.ray_count) return;
    const DfnFrame& F = A.frame;
    const int pix = A.pix_index ? A.pix_index[ray] : F.ray_begin + (int)ray;
    const int y = pix / F.W, x = pix - y * F.W;
    const float dx = __fdiv_rn(__fsub_rn((float)x, F.cx), F.focal), dy = __fdiv_rn(-__fsub_rn((float)y, F.cy), F.focal);
    float nrm[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float* P = b ? F.pose_body : F.pose;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, P[4 * k]), __fmul_rn(dy, P[4 * k + 1])),
                                      __fmul_rn(-1.0f, P[4 * k + 2]));
            acc = __fadd_rn(acc, __fmul_rn(d, d));
        }
        nrm[b] = sqrtf(acc);
    }
    // inverse of the rank map: merged position -> evaluation index
    unsigned char* inv = inv_s[wv];
#pragma unroll
    for (int m = 0; m < K; ++m) inv[A.ranks[ray * S + lane + 64 * m]] = (unsigned char)(lane + 64 * m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool cbg = F.concate_bg != 0, two = F.fields == 2;
    float Gh[3], Gc[3], bgc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Gh[k] = A.d_rgb_head[ray * 3 + k];
        Gc[k] = A.d_rgb_com ? A.d_rgb_com[ray * 3 + k] : 0.f;
        bgc[k] = A.bg_u8 ? __fdiv_rn((float)A.bg_u8[(size_t)pix * 3 + k], 255.0f) : A.bg_f32[(size_t)pix * 3 + k];
    }
    // per owned position: everything the two images' backward needs
    int ev[K];
    float dz[K], sg_h[K], sg_t[K], ch[K][3], ct[K][3];
    bool lastp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = lane * K + k;
        ev[k] = inv[i];
        lastp[k] = i == S - 1;
        const float z = A.z_all[ray * S + i];
        dz[k] = lastp[k] ? F.last_dist : __fsub_rn(A.z_all[ray * S + i + 1], z);
        const float4* sm = (const float4*)(A.samples + ((size_t)ray * S + ev[k]) * 8);
        const float4 q0 = sm[0], q1 = sm[1];
        sg_h[k] = q0.x; ch[k][0] = q0.y; ch[k][1] = q0.z; ch[k][2] = q0.w;
        sg_t[k] = q1.x; ct[k][0] = q1.y; ct[k][1] = q1.z; ct[k][2] = q1.w;
        if (cbg && lastp[k]) { ch[k][0] = bgc[0]; ch[k][1] = bgc[1]; ch[k][2] = bgc[2]; }
    }
    // weights + d/d(sigma) of one image over the merged samples: s (effective sigma), dist, q = colour . G per position
    auto image_bwd = [&](const float (&s)[K], const float (&dist)[K], const float (&q)[K], float (&w)[K], float (&ds)[K]) {
        float e[K], a[K], v[K], pre[K];
        float prod = 1.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            e[k] = expf(-((fmaxf(s[k], 0.f) + 1e-6f) * dist[k]));
            a[k] = 1.0f - e[k];
            v[k] = 1.0f - a[k] + 1e-10f;
            pre[k] = prod;                       // product of this lane's earlier positions
            prod *= v[k];
        }
        const float T0 = wave_excl_prod(prod, lane);          // transmittance in front of this lane's first position
        float wq_sum = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            w[k] = a[k] * (T0 * pre[k]);
            wq_sum += w[k] * q[k];
        }
        const float suf0 = wave_suffix_sum_excl(wq_sum, lane);     // sum of w q over the later lanes
        float later = 0.f;                                         // ... plus this lane's later positions
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            const float da = (T0 * pre[k]) * q[k] - (suf0 + later) / v[k];
            ds[k] = da * dist[k] * e[k];
            later += w[k] * q[k];
        }
    };
    float d_sg_h[K], d_sg_t[K], d_ch[K][3], d_ct[K][3];
    {   // head-only image
        float s1[K], dist[K], q[K], w[K], ds[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float sh = fmaxf(sg_h[k], 0.f);
            s1[k] = (cbg && lastp[k]) ? sh + 1e-6f : sh;
            dist[k] = dz[k] * nrm[0];
            q[k] = ch[k][0] * Gh[0] + ch[k][1] * Gh[1] + ch[k][2] * Gh[2];
        }
        image_bwd(s1, dist, q, w, ds);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool h_is_bg = cbg && lastp[k];
            d_sg_h[k] = sg_h[k] > 0.f ? ds[k] : 0.f;
            d_sg_t[k] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d_ch[k][c] = h_is_bg ? 0.f : w[k] * Gh[c];
                d_ct[k][c] = 0.f;
            }
        }
    }
    if (two) {   // composite image (run_nerf_com_trainExpLater.py:146-166)
        float ss[K], dist[K], q[K], w[K], ds[K], sh[K], stt[K], den[K], wh[K], wt[K];
        bool zero[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float sgt = (cbg && lastp[k]) ? 0.f : sg_t[k];
            sh[k] = fmaxf(sg_h[k], 0.f);
            stt[k] = fmaxf(sgt, 0.f);
            if (cbg && lastp[k]) stt[k] += 1e-6f;
            ss[k] = sh[k] + stt[k];
            zero[k] = ss[k] == 0.f;
            den[k] = zero[k] ? 1e-4f : ss[k];
            wh[k] = sh[k] / den[k];
            wt[k] = stt[k] / den[k];
            dist[k] = dz[k] * nrm[1];
            float cm[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) cm[c] = ch[k][c] * wh[k] + ct[k][c] * wt[k];
            q[k] = cm[0] * Gc[0] + cm[1] * Gc[1] + cm[2] * Gc[2];
        }
        image_bwd(ss, dist, q, w, ds);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool h_is_bg = cbg && lastp[k];
            const float gch = ch[k][0] * Gc[0] + ch[k][1] * Gc[1] + ch[k][2] * Gc[2];
            const float gct = ct[k][0] * Gc[0] + ct[k][1] * Gc[1] + ct[k][2] * Gc[2];
            const float dwh = w[k] * gch, dwt = w[k] * gct;
            float dsh = ds[k] + dwh / den[k], dst = ds[k] + dwt / den[k];
            if (!zero[k]) {
                const float common = (dwh * sh[k] + dwt * stt[k]) / (den[k] * den[k]);
                dsh -= common;
                dst -= common;
            }
            if (sg_h[k] > 0.f) d_sg_h[k] += dsh;
            if (sg_t[k] > 0.f && !(cbg && lastp[k])) d_sg_t[k] += dst;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (!h_is_bg) d_ch[k][c] += w[k] * Gc[c] * wh[k];
                d_ct[k][c] += w[k] * Gc[c] * wt[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float4* out = (float4*)(A.dsamples + ((size_t)ray * S + ev[k]) * 8);
        out[0] = make_float4(d_sg_h[k], d_ch[k][0], d_ch[k][1], d_ch[k][2]);
        out[1] = make_float4(d_sg_t[k], d_ct[k][0], d_ct[k][1], d_ct[k][2]);
    }
}
hipError_t launch_composite_bwd_hier(const CompositeBwdArgs& A, hipStream_t st) {
    const int blocks = (A.frame.ray_count + 3) / 4;
    const int K = (64 + A.frame.n_fine) / 64;
    if (K == 2) hipLaunchKernelGGL(composite_bwd_hier_kernel<2>, dim3(blocks), dim3(256), 0, st, A);
    else if (K == 3) hipLaunchKernelGGL(composite_bwd_hier_kernel<3>, dim3(blocks), dim3(256), 0, st, A);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ================================================================================================
// weight gradients (f32 tier): C[M x N] = A[M x NP] * B[N x NP]^T, contraction over the sample points, split over slices of the
// points (one partial array per slice, no atomics; launch_reduce_scatter adds the slices in index order).
// Two kernels, both with the operands through LDS: wgrad_full_kernel (the 256 x 256 GEMMs, 89 % of a field's FLOPs) and
// wgrad_narrow_kernel (everything else, all shapes in one launch).  Rounds 2-5 ran the narrow GEMMs in a kernel that read its
// operands straight from L2 / HBM, one wave per 2 x 4-tile macro-tile, one wave per SIMD (420 registers): latency-bound,
// 0.48 ms (head) / 0.64 ms (torso) for 11 % of the FLOPs (profiles/r05end_c4_f32_timeline.txt).
// ================================================================================================
// One GEMM (or a block of its row tiles) for one slice of the points, by one workgroup of four waves, the operands through LDS
// once per workgroup (round 5, for the 256 x 256 GEMMs; before that a general kernel read every operand tile from L2 / HBM once
// per wave macro-tile that needed it - 32-byte pieces of 128-byte lines, 46 % of the f32 MFMA peak with one wave per SIMD):
//   * a step = one 32-point tile: the block's dy_T rows and act_T rows of the tile are CONTIGUOUS in memory (tile-major arrays)
//     -> 1-KiB LDS-DMA pieces of 8 rows each, dealt round-robin to the four waves;
//   * the LDS image is swizzled at the SOURCE (the DMA writes lane i's 16 bytes at base + 16 i; each lane chooses what it
//     fetches): 16-byte chunk c of row r sits at slot 8 r + (c ^ ((r >> 1) & 7)) - the sixteen rows of a ds_read_b128 lane
//     group then hit sixteen different slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS);
//   * the four waves as an RG x (4 / RG) grid over the block's MT x NT output tiles;
//   * NS stages, ONE barrier per step: "tile t has landed" and "everybody is done with tile t - 1" are the same barrier, behind
//     it the stage of tile t - 1 is refilled with tile t + NS - 1 (round 5 had two barriers per step around two stages).
// wgrad_full_kernel: the 256 x 256 GEMMs (8 of a field's 13 / 26, 89 % of its FLOPs), <8, 8, 4, 2>: wave w owns output rows
// 64 w .. 64 w + 63 x all 256 columns (256 accumulator registers), one workgroup per compute unit; 8 GEMMs x 32 slices = 256
// workgroups = one round: 1.02-1.05 ms per field.  Its step is HALF a tile (16 points, two 32-KiB stages, 324 registers; the
// 32-point step with two 64-KiB stages: 356 registers and 0.04-0.08 ms more per training step, profiles/r06s_*, r06u_*).
//
// wgrad_narrow_kernel: every GEMM that is NOT 256 x 256 (11 % of a field's weight-gradient FLOPs, but a third of its operand
// bytes), all of them in ONE launch: one workgroup per WNItem = (GEMM, block of its row tiles, slice of the points).  (Round 5
// tried one launch per narrow shape: 32-416 workgroups of little work each, one launch after the other - slower than the
// general kernel.  Side by side in one launch they fill the chip.)  What the narrow shapes need on top:
//   * 2-4 stages, as many as fit 72 KiB: a stage of a 64 x 64 GEMM is 16 KiB and its step 16 MFMAs per wave - two stages
//     would leave the workgroup waiting for memory most of the time;
//   * 72 KiB of LDS and < 128 registers: two workgroups per compute unit (the second hides the first's barriers and waits);
//   * GEMMs with 256 dy_T rows are cut into two blocks of 128 rows (their act_T rows are read twice: 8-32 KiB per step): every
//     item then holds at most 4 output tiles per wave and the launch's longest workgroup is 128 steps x 64 MFMAs;
//   * a dy_T row block no GEMM reads (N = 0: its layer multiplies a per-frame constant) is summed straight from memory.
// Same products in the same order per output element as before (MFMA m of an 8-point group pairs point m with point m + 4, groups
// in point order), same row-sum order: the slices' partial sums are bit-identical to rounds 2-5.
// (Shapes: dfn_train.h wn_shape_of; anything else is refused when the plan is built, dfn_api.hip.)
template <int MT, int NT, int RG, int NS, int LDS_MAX = 72 * 1024, int PT = 32>
__device__ __forceinline__ void wgrad_lds_part(const WOp& o, int m_tile0, int ks, const float* dy_T, const float* act_T, long n_tiles,
                                               int g_rows, int a_rows, int ksplit, float* C, long c_stride, const int* e_of,
                                               float* dbias, int n_bias, lds_char* lds) {
    // PT = points per step: 32 (a whole tile: 128-byte rows, 8 rows per 1-KiB DMA piece) or 16 (half a tile: 64-byte half rows,
    // 16 rows per piece, chunk c of row r at slot 4 r + (c ^ ((r >> 2) & 3)) - the same sixteen-different-slots rule)
    constexpr int CG = 4 / RG, MW = MT / RG, NW = NT / CG;
    constexpr int HP = 32 / PT, CH = PT / 4, RP = 256 / PT, QN = PT / 8;
    constexpr int PIECES = (MT + NT) * 32 / RP, PA = MT * 32 / RP, PW = PIECES / 4;
    constexpr int STAGE = PIECES * 1024;
    static_assert((PT == 32 || PT == 16) && PIECES % 4 == 0, "step");
    static_assert(MT % RG == 0 && NT % CG == 0 && NS >= 2 && NS <= 4 && NS * STAGE <= LDS_MAX && (NS - 1) * PW <= 63, "shape");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave / CG, cg = wave % CG;
    const long per = (n_tiles + ksplit - 1) / ksplit;
    const long t0 = ks * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    if (t0 >= t1) return;                                                // a slice without points: the reduction skips it
    const long n_steps = (t1 - t0) * HP;
    const int a_row0 = o.a_row + 32 * m_tile0;
    auto swz = [](int r) { return PT == 32 ? (r >> 1) & 7 : (r >> 2) & 3; };
    auto issue = [&](int stage, long u) {
        const long t = t0 + u / HP;
        const int half = (int)(u % HP);
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int p = 4 * k + wave;                                  // wave-uniform
            const bool isb = p >= PA;
            const int r = RP * (isb ? p - PA : p) + lane / CH;
            const int c = (lane % CH) ^ swz(r);
            const float* base = isb ? act_T + (t * (long)a_rows + o.b_row) * 32 : dy_T + (t * (long)g_rows + a_row0) * 32;
            const gchar_c* sb = (const gchar_c*)uniform_ptr(base);
            const unsigned voff = (unsigned)(r * 128 + half * (PT * 4) + c * 16);
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)lds + (unsigned)(stage * STAGE + p * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(dst) : "memory", "m0");
        }
    };
    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int rl = lane & 31, h = lane >> 5;
    const bool do_bias = dbias && o.bias_owner && cg == 0;               // wave-uniform
    float rs[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) rs[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < n_steps) issue(s, s);
    int st = 0;                                                          // stage of step u
    for (long u = 0; u < n_steps; ++u) {
        // this wave's pieces of step u have landed once at most the pieces of the younger steps in flight are outstanding
        const long younger = (n_steps - 1 - u < NS - 2) ? n_steps - 1 - u : NS - 2;
        if (NS >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
        else if (NS >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                     // ... and everybody else's; everybody is also done reading the stage of step u - 1
        if (u + NS - 1 < n_steps) issue(st == 0 ? NS - 1 : st - 1, u + NS - 1);
        const lds_char* sa = lds + st * STAGE;
        const lds_char* sb_ = sa + PA * 1024;
        // the operands of 8-point group q + 1 are read while the MFMAs of group q issue (two register sets: left to itself the
        // compiler reads a group's operands into the registers the previous group's last MFMA has just released, and the first
        // MFMA of every group waits for LDS)
        f32x4 av[2][MW], bv[2][NW];
        auto fetch = [&](int q) {
            const int c = 2 * q + h;
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                const int r = 32 * (MW * rg + i) + rl;
                av[q & 1][i] = *(const lds_f32x4*)(sa + (r * CH + (c ^ swz(r))) * 16);
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int r = 32 * (NW * cg + j) + rl;
                bv[q & 1][j] = *(const lds_f32x4*)(sb_ + (r * CH + (c ^ swz(r))) * 16);
            }
        };
        fetch(0);
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            if (q < QN - 1) fetch(q + 1);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][i][m], bv[q & 1][j][m], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < MW; ++i) rs[i] += (av[q & 1][i][0] + av[q & 1][i][1]) + (av[q & 1][i][2] + av[q & 1][i][3]);
            }
        }
        st = (st + 1 == NS) ? 0 : st + 1;
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const float tot = rs[i] + __shfl_xor(rs[i], 32);
            if (h == 0) {
                const int e = e_of[a_row0 + 32 * (MW * rg + i) + rl];
                if (e >= 0) dbias[(long)ks * n_bias + e] = tot;                 // one writer per (slice, element)
            }
        }
    }
    float* c = C + (long)ks * c_stride + o.c_off;
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (m_tile0 + MW * rg + i) + tile_feat(h, r), col = 32 * (NW * cg + j) + rl;
                c[(long)row * o.N + col] = acc[i][j][r];
            }
}
// row sums only (N = 0): up to 64 dy_T rows from row tile m_tile0 on, summed over the slice's points straight from memory.
// Thread (row = tid & 63, quarter = tid >> 6) adds points 8 quarter .. 8 quarter + 7 of every tile in tile order, then the four
// quarters are added in index order: a fixed order.
__device__ __forceinline__ void wgrad_rows_part(const WOp& o, int m_tile0, int ks, const float* dy_T, long n_tiles, int g_rows, int ksplit,
                                                const int* e_of, float* dbias, int n_bias, lds_char* lds) {
    const long per = (n_tiles + ksplit - 1) / ksplit;
    const long t0 = ks * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    if (t0 >= t1 || !dbias || !o.bias_owner) return;
    const int row = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int r = 32 * m_tile0 + row;                                    // row inside the GEMM's block
    float s = 0.f;
    if (r < o.M) {
        const float* p = dy_T + ((long)t0 * g_rows + o.a_row + r) * 32 + 8 * quarter;
        for (long t = t0; t < t1; ++t, p += (long)g_rows * 32) {
            const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
            s += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
        }
    }
    lds_f32* red = (lds_f32*)lds;
    red[quarter * 64 + row] = s;
    __syncthreads();
    if (quarter == 0 && r < o.M) {
        const float tot = (red[row] + red[64 + row]) + (red[128 + row] + red[192 + row]);
        const int e = e_of[o.a_row + r];
        if (e >= 0) dbias[(long)ks * n_bias + e] = tot;
    }
}
constexpr int WN_LDS_BYTES = 72 * 1024;
__global__ __launch_bounds__(256, 2) void wgrad_narrow_kernel(const WOp* ops, const WNItem* items, const float* dy_T, const float* act_T,
                                                              long n_tiles, int g_rows, int a_rows, int ksplit, float* C,
                                                              long c_stride, const int* e_of, float* dbias, int n_bias) {
    extern __shared__ __attribute__((aligned(16))) char wn_smem[];
    lds_char* lds = (lds_char*)wn_smem;
    const WNItem it = items[blockIdx.x];
    const WOp o = ops[it.op];
#define DFN_WN(MT, NT, RG, NS) \
    wgrad_lds_part<MT, NT, RG, NS>(o, it.m_tile0, it.ks, dy_T, act_T, n_tiles, g_rows, a_rows, ksplit, C, c_stride, e_of, dbias, n_bias, lds)
    switch (it.shape) {                                                  // (uniform over the workgroup)
        case WN_4x4: DFN_WN(4, 4, 2, 2); break;                          // 128 x 128 of a 256 x 128 GEMM: wave = 2 x 2 tiles, 32-KiB stages
        case WN_4x2: DFN_WN(4, 2, 4, 3); break;                          // 128 x 64 of a 256 x 64 GEMM: wave = 1 x 2 tiles, 24-KiB stages
        case WN_1x8: DFN_WN(1, 8, 1, 2); break;                          // 32 x 256: wave = 1 x 2 tiles, 36-KiB stages
        case WN_4x1: DFN_WN(4, 1, 4, 3); break;                          // 128 x 32 of a 256 x 32 GEMM: wave = 1 tile, 20-KiB stages
        case WN_2x2: DFN_WN(2, 2, 2, 4); break;                          // 64 x 64: wave = 1 tile, 16-KiB stages
        default: wgrad_rows_part(o, it.m_tile0, it.ks, dy_T, n_tiles, g_rows, ksplit, e_of, dbias, n_bias, lds); break;
    }
#undef DFN_WN
}

// the 256 x 256 GEMMs (two 32-KiB stages of 16 points; `full_ops`: their indices in `ops`, dfn_api.hip)
__global__ __launch_bounds__(256) void wgrad_full_kernel(const WOp* ops, const int* full_ops, const float* dy_T, const float* act_T,
                                                          long n_tiles, int g_rows, int a_rows, int ksplit, float* C, long c_stride,
                                                          const int* e_of, float* dbias, int n_bias) {
    extern __shared__ __attribute__((aligned(16))) char wf1_smem[];
    const WOp o = ops[full_ops[blockIdx.x / ksplit]];
    wgrad_lds_part<8, 8, 4, 2, 64 * 1024, 16>(o, 0, blockIdx.x % ksplit, dy_T, act_T, n_tiles, g_rows, a_rows, ksplit, C, c_stride, e_of,
                                              dbias, n_bias, (lds_char*)wf1_smem);
}

// hipFuncSetAttribute once per (kernel, device): a second device of the process needs it too
template <typename K> static hipError_t lds_attr_once(K kernel, int bytes, bool (&done)[64]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !done[dev]) {
        e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
    return hipSuccess;
}

hipError_t launch_wgrad(int tier, int field, const WOp* ops_dev, const int* full_ops_dev, int n_full, const WNItem* nitems_dev,
                        int n_nitems, const void* dy_T, const void* act_T, long NP, int ksplit, float* C, long c_stride,
                        const int* e_of, float* dbias, int n_bias, hipStream_t st) {
    const bool torso = field == FIELD_TORSO;
    const int g_rows = torso ? GradMap::S_ROWS : GradMap::H_ROWS, a_rows = torso ? RecMap::S_ROWS : RecMap::H_ROWS;
    if (tier != TIER_F32) return hipErrorInvalidValue;          // bf16: launch_wgrad_bf16
    const float *dy = (const float*)dy_T, *ac = (const float*)act_T;
    hipError_t e;
    if (n_full > 0) {           // the 256 x 256 GEMMs (8 of a field's GEMMs, 89 % of its FLOPs): one workgroup per (GEMM, slice)
        constexpr int lds = 2 * (8 + 8) * 2 * 1024;             // two stages of 16 points
        static bool done[64] = {};
        if ((e = lds_attr_once(wgrad_full_kernel, lds, done)) != hipSuccess) return e;
        hipLaunchKernelGGL(wgrad_full_kernel, dim3(n_full * ksplit), dim3(256), lds, st, ops_dev, full_ops_dev, dy, ac, NP / 32,
                           g_rows, a_rows, ksplit, C, c_stride, e_of, dbias, n_bias);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (n_nitems > 0) {         // everything else, side by side in one launch
        static bool done[64] = {};
        if ((e = lds_attr_once(wgrad_narrow_kernel, WN_LDS_BYTES, done)) != hipSuccess) return e;
        hipLaunchKernelGGL(wgrad_narrow_kernel, dim3(n_nitems), dim3(256), WN_LDS_BYTES, st, ops_dev, nitems_dev, dy, ac, NP / 32, g_rows,
                           a_rows, ksplit, C, c_stride, e_of, dbias, n_bias);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// ================================================================================================
// pixel sampling: run_nerf_com_trainExpLater.py:786-820 in one launch, one workgroup
// ================================================================================================
// n DISTINCT pixels, uniform over the subsets, in random order (np.random.choice(replace=False) upstream); with
// rect_num > 0 the first rect_num of them from (face rect | lower half of the image), the other n - rect_num from the
// complement (MAIN:786-817).  Sequential rejection sampling, in parallel: M = 8192 candidates drawn with a counter-based
// generator (splitmix64 of (seed, counter, i)); candidate i is kept iff no earlier candidate has the same pixel (an LDS
// hash table holds the smallest index per pixel: atomicMin, so the result does not depend on thread timing); a block
// scan ranks the kept candidates of each class in draw order.  status[0] / status[1] = kept candidates per class (must
// be >= the request; the host wrapper sizes the request so that it always is).
constexpr int SP_M = 8192, SP_TABLE = 16384, SP_THREADS = 1024, SP_PER = SP_M / SP_THREADS;
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(SP_THREADS) void sample_pixels_kernel(int H, int W, int n, int rect_num, const int* rect,
                                                                    unsigned long long seed, unsigned long long counter,
                                                                    int* out, int* status) {
    extern __shared__ unsigned sp_lds[];
    // [SP_TABLE]  smallest candidate index of the pixel that claimed the slot, 0xffffffff = empty.  The pixel itself is
    // cand[entry]: a claimed slot only ever receives candidates of ITS pixel, so that stays valid whatever the timing, and
    // the frame size is not limited by the entry's bits (round 3 packed (pixel << 13 | index) into the word: H * W < 2^18,
    // one short of the 512 x 512 frames the reference's preprocessing emits, scripts/process_data.sh:4)
    unsigned* table = sp_lds;
    unsigned* cand = sp_lds + SP_TABLE;          // [SP_M]
    unsigned* part = cand + SP_M;                // [2][SP_THREADS] per-thread counts, then their exclusive prefix
    const int t = threadIdx.x;
    const unsigned HW = (unsigned)(H * W);
    for (int e = t; e < SP_TABLE; e += SP_THREADS) table[e] = 0xffffffffu;
    const unsigned long long base = splitmix64(seed ^ (counter * 0xD1B54A32D192ED03ull));
#pragma unroll
    for (int q = 0; q < SP_PER; ++q) {           // thread t owns the CONSECUTIVE candidates t * SP_PER + q (draw order)
        const int i = t * SP_PER + q;
        const unsigned r = (unsigned)(splitmix64(base + (unsigned long long)i) >> 32);
        cand[i] = (unsigned)(((unsigned long long)r * HW) >> 32);       // uniform in [0, HW) up to HW / 2^32
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SP_PER; ++q) {
        const int i = t * SP_PER + q;
        const unsigned p = cand[i];
        unsigned slot = (p * 2654435761u) >> 18;                         // 14 bits
        for (;;) {
            const unsigned cur = atomicCAS(&table[slot], 0xffffffffu, (unsigned)i);
            if (cur == 0xffffffffu) break;
            if (cand[cur] == p) { atomicMin(&table[slot], (unsigned)i); break; }
            slot = (slot + 1) & (SP_TABLE - 1);
        }
    }
    __syncthreads();
    int ry0 = 0, rx0 = 0, ry1 = -1, rx1 = -1;
    if (rect_num > 0) { ry0 = rect[0]; rx0 = rect[1]; ry1 = ry0 + rect[2]; rx1 = rx0 + rect[3]; }
    unsigned keep = 0, cls = 0;                  // bit q: candidate kept / candidate inside (face rect | lower half)
    int c_in = 0, c_out = 0;
#pragma unroll
    for (int q = 0; q < SP_PER; ++q) {
        const int i = t * SP_PER + q;
        const unsigned p = cand[i];
        unsigned slot = (p * 2654435761u) >> 18;
        while (cand[table[slot]] != p) slot = (slot + 1) & (SP_TABLE - 1);
        const bool first = table[slot] == (unsigned)i;
        const int y = (int)(p / (unsigned)W), x = (int)(p - (unsigned)y * (unsigned)W);
        const bool inside = rect_num > 0 && ((y >= ry0 && y <= ry1 && x >= rx0 && x <= rx1) || 2 * y >= H);
        if (first) {
            keep |= 1u << q;
            if (inside) { cls |= 1u << q; ++c_in; } else ++c_out;
        }
    }
    part[t] = (unsigned)c_in;
    part[SP_THREADS + t] = (unsigned)c_out;
    __syncthreads();
    if (t < 128) {                               // exclusive prefix over the 1024 per-thread counts, both classes: wave w
        const int which = t >> 6, lane = t & 63; // = class; lane owns 16 consecutive entries
        unsigned* a = part + which * SP_THREADS + lane * 16;
        unsigned loc[16], sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { loc[k] = sum; sum += a[k]; }
        unsigned inc = sum;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned up = __shfl_up(inc, d);
            if (lane >= d) inc += up;
        }
        const unsigned excl = inc - sum;
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = excl + loc[k];
        if (lane == 63 && status) status[which] = (int)inc;
    }
    __syncthreads();
    int r_in = (int)part[t], r_out = (int)part[SP_THREADS + t];
    const int n_out = n - rect_num;
#pragma unroll
    for (int q = 0; q < SP_PER; ++q) {
        if (!((keep >> q) & 1u)) continue;
        const unsigned p = cand[t * SP_PER + q];
        if ((cls >> q) & 1u) {
            if (r_in < rect_num) out[r_in] = (int)p;
            ++r_in;
        } else {
            if (r_out < n_out) out[rect_num + r_out] = (int)p;
            ++r_out;
        }
    }
}
hipError_t launch_sample_pixels(int H, int W, int n, int rect_num, const int* rect, unsigned long long seed,
                                unsigned long long counter, int* out, int* status, hipStream_t st) {
    constexpr int lds = (SP_TABLE + SP_M + 2 * SP_THREADS) * 4;
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)sample_pixels_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL(sample_pixels_kernel, dim3(1), dim3(SP_THREADS), lds, st, H, W, n, rect_num, rect, seed, counter, out,
                       status);
    return hipGetLastError();
}

// ================================================================================================
// loss: target gather + the two MSEs + their gradients in one launch
// ================================================================================================
// run_nerf_com_trainExpLater.py:791-800 (target[select_coords] of the head and the composite image), :902-907
// (img2mse(rgb_com_torso, target_com) + img2mse(rgb_head, target_head)) and what torch autograd makes of them
// (d loss / d rgb = 2 (rgb - target) / (3 n)).  Targets are uint8 images resident on the device (/ 255 like LOAD:58-60).
// One workgroup, fixed reduction order: thread t adds its elements t, t + 1024, ... in sequence, then an LDS tree.  Round 4:
// the gathers of eight elements are in flight together (the loop used to walk pix -> image byte -> next element as six
// DEPENDENT round trips per thread: 17 us between the forward and the dX chain); the additions keep their order, so the
// results are bit for bit what they were.  (A several-workgroup version with a last-block ticket was 2 us faster and needed a
// device-global counter: the library keeps no mutable global state, include/dfanerf.h.)
__global__ __launch_bounds__(1024) void mse_loss_kernel(const float* __restrict__ rgb_head, const float* __restrict__ rgb_com,
                                                        const unsigned char* __restrict__ img_head,
                                                        const unsigned char* __restrict__ img_com,
                                                        const int* __restrict__ pix, int n, float* losses, float* d_head,
                                                        float* d_com) {
    __shared__ float red[2][1024];
    const int t = threadIdx.x, total = 3 * n;
    const float scale = __fdiv_rn(2.0f, (float)total);
    float sh = 0.f, sc = 0.f;
    constexpr int B = 8;
    for (int e0 = t; e0 < total; e0 += B * 1024) {
        float rh[B], rc[B];
        unsigned char th[B], tc[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {                    // every load of the batch before any use
            const int e = e0 + k * 1024;
            if (e < total) {
                const int r = e / 3, c = e - 3 * r;
                const size_t src = (size_t)pix[r] * 3 + c;
                th[k] = img_head[src];
                tc[k] = img_com[src];
                rh[k] = rgb_head[e];
                rc[k] = rgb_com[e];
            }
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const int e = e0 + k * 1024;
            if (e < total) {
                const float dh = __fsub_rn(rh[k], __fdiv_rn((float)th[k], 255.0f));
                const float dc = __fsub_rn(rc[k], __fdiv_rn((float)tc[k], 255.0f));
                sh = __fadd_rn(sh, __fmul_rn(dh, dh));
                sc = __fadd_rn(sc, __fmul_rn(dc, dc));
                d_head[e] = __fmul_rn(scale, dh);
                d_com[e] = __fmul_rn(scale, dc);
            }
        }
    }
    red[0][t] = sh;
    red[1][t] = sc;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) {
        if (t < s) {
            red[0][t] = __fadd_rn(red[0][t], red[0][t + s]);
            red[1][t] = __fadd_rn(red[1][t], red[1][t + s]);
        }
        __syncthreads();
    }
    if (t == 0) {
        losses[0] = __fdiv_rn(red[0][0], (float)total);      // img2mse(rgb_head, target_head)
        losses[1] = __fdiv_rn(red[1][0], (float)total);      // img2mse(rgb_com, target_com)
        losses[2] = __fadd_rn(losses[1], losses[0]);          // the step's loss (MAIN:902-907: loss_com + loss_head)
    }
}
hipError_t launch_mse_loss(const float* rgb_head, const float* rgb_com, const unsigned char* img_head,
                           const unsigned char* img_com, const int* pix, int n, float* losses, float* d_head, float* d_com,
                           hipStream_t st) {
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(1024), 0, st, rgb_head, rgb_com, img_head, img_com, pix, n, losses,
                       d_head, d_com);
    return hipGetLastError();
}

// grad_flat[map[i]] += parts[0][i] + parts[1][i] + ... (fixed order): the second stage of the split-K reduction
__global__ void reduce_scatter_kernel(const int* map, const float* parts, long n, long stride, int slices, float* grad_flat) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int dst = map[i];
    if (dst < 0) return;
    float a = 0.f;
    for (int k = 0; k < slices; ++k) a += parts[(long)k * stride + i];
    grad_flat[dst] += a;          // every parameter appears at most once per field
}
hipError_t launch_reduce_scatter(const int* map, const float* parts, long n, long stride, int slices, float* grad_flat,
                                 hipStream_t st) {
    hipLaunchKernelGGL(reduce_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, map, parts, n, stride,
                       slices, grad_flat);
    return hipGetLastError();
}
// both second stages of the split-K reduction in one launch (bf16 tier: the row sums ride along in the GEMMs): blocks
// [0, bias_blocks) add the bias slices, the others the weight slices - two dependent 10-25 us launches less per field
__device__ __forceinline__ int filled_slices(int n, long units) {       // slices of an n-way split that hold points
    const long per = (units + n - 1) / n;
    return (int)((units + per - 1) / per);
}
__global__ void reduce_both_kernel(const int* map, const float* parts, long n, long stride, int slices, float* grad_flat,
                                   const int* rows, const float* bparts, int n_bias, float* dbias, int bias_blocks,
                                   const unsigned char* blk_n, const unsigned char* bias_n, long units) {
    if ((int)blockIdx.x < bias_blocks) {
        const int e = blockIdx.x * blockDim.x + threadIdx.x;
        if (e >= n_bias) return;
        float a = 0.f;
        if (rows[e] >= 0) {
            const int sl = bias_n ? filled_slices(bias_n[e], units) : slices;
            for (int k = 0; k < sl; ++k) a += bparts[(long)k * n_bias + e];
        }
        dbias[e] = a;
        return;
    }
    const long i = (long)(blockIdx.x - bias_blocks) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int dst = map[i];
    if (dst < 0) return;
    const int sl = blk_n ? filled_slices(blk_n[blockIdx.x - bias_blocks], units) : slices;     // (one GEMM per 256-element block)
    float a = 0.f;
    for (int k = 0; k < sl; ++k) a += parts[(long)k * stride + i];
    grad_flat[dst] += a;
}
hipError_t launch_reduce_both(const int* map, const float* parts, long n, long stride, int slices, float* grad_flat,
                              const int* rows, const float* bparts, int n_bias, float* dbias, const unsigned char* blk_n,
                              const unsigned char* bias_n, long units, hipStream_t st) {
    const int bias_blocks = (n_bias + 255) / 256;
    hipLaunchKernelGGL(reduce_both_kernel, dim3((unsigned)(bias_blocks + (n + 255) / 256)), dim3(256), 0, st, map, parts, n,
                       stride, slices, grad_flat, rows, bparts, n_bias, dbias, bias_blocks, blk_n, bias_n, units);
    return hipGetLastError();
}
__global__ void reduce_bias_kernel(const int* rows, const float* parts, int n_bias, int slices, float* dbias) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_bias) return;
    float a = 0.f;
    if (rows[e] >= 0)
        for (int k = 0; k < slices; ++k) a += parts[(long)k * n_bias + e];
    dbias[e] = a;
}
hipError_t launch_reduce_bias(const int* rows, const float* parts, int n_bias, int slices, float* dbias, hipStream_t st) {
    hipLaunchKernelGGL(reduce_bias_kernel, dim3((n_bias + 255) / 256), dim3(256), 0, st, rows, parts, n_bias, slices, dbias);
    return hipGetLastError();
}

// ---- the 16-bit tier's recorded arrays are MX-fp8 (dfn_mlp.h: "MX-fp8 recording"): per 32-point tile one KiB per 32-row
// block, point-major ([point][half h][register r] = feature tile_feat(h, r)), + one E8M0 scale per block.  Sum of row `row`
// over the 32 points of tile t, dequantised: the row's byte of every point's 32-byte record.
__device__ __forceinline__ float rec8_row_sum(const unsigned char* arr, long t, int rows, int row) {
    const unsigned char* base = arr + t * rec8_tile_bytes(rows);
    int h, r;
    tile_feat_inv(row & 31, &h, &r);
    const unsigned* p = (const unsigned*)(base + (long)(row >> 5) * 1024 + h * 16 + (r & ~3));
    const int sh = 8 * (r & 3);
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 32; ++n) s += __builtin_amdgcn_cvt_f32_fp8((int)(p[n * 8] >> sh), 0);
    const unsigned e8 = base[(long)rows * 32 + (row >> 5)];
    return s * __uint_as_float(e8 << 23);
}

// d(bias blob)[e] = sum over points of dy_T[.., row_of[e], ..]   (tile-major array [tile][rows][32]).
// Streaming row sums: thread = one row, block = 256 consecutive rows, blockIdx.y = a slice of the tiles; a wave
// reads 64 rows x 32 points = one contiguous 4 KiB (bf16) / 8 KiB (f32) run per tile.  The partial sum of slice y goes
// to parts[y][e_of[row]] (bias element of a row, -1 = none; one writer per slice and element), reduce_bias_kernel adds the
// slices in order.
template <typename T>
__global__ void bias_grad_kernel(const int* e_of, const T* dy_T, long n_tiles, int rows, float* parts, int n_bias) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int e = e_of[row];
    if (e < 0) return;
    const long per = (n_tiles + gridDim.y - 1) / gridDim.y;
    const long t0 = blockIdx.y * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    float acc = 0.f;
    for (long t = t0; t < t1; ++t) {
        if constexpr (sizeof(T) == 1) {          // 16-bit tier: MX-fp8
            acc += rec8_row_sum((const unsigned char*)dy_T, t, rows, row);
        } else {
            const uint4* p = (const uint4*)(dy_T + (t * rows + row) * 32);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint4 v = p[q];
                acc += (__uint_as_float(v.x) + __uint_as_float(v.y)) + (__uint_as_float(v.z) + __uint_as_float(v.w));
            }
        }
    }
    parts[(long)blockIdx.y * n_bias + e] = acc;
}
hipError_t launch_bias_grad(int tier, int field, const int* e_of, const int* bias_rows, int n_bias, const void* dy_T,
                            long NP, float* parts, float* dbias, hipStream_t st) {
    const int rows = field == FIELD_TORSO ? GradMap::S_ROWS : GradMap::H_ROWS;
    const dim3 grid((rows + 255) / 256, BIAS_GRAD_SLICES);
    if (tier == TIER_BF16)
        hipLaunchKernelGGL(bias_grad_kernel<unsigned char>, grid, dim3(256), 0, st, e_of, (const unsigned char*)dy_T, NP / 32,
                           rows, parts, n_bias);
    else
        hipLaunchKernelGGL(bias_grad_kernel<float>, grid, dim3(256), 0, st, e_of, (const float*)dy_T, NP / 32, rows, parts,
                           n_bias);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    return launch_reduce_bias(bias_rows, parts, n_bias, BIAS_GRAD_SLICES, dbias, st);
}


// ---- d(signal) ahead of the weight gradients ---------------------------------------------------------------------------
// The conditioning networks' backward (dfn_encode_signal*_bwd: single-workgroup latency chains, 0.2 ms) only needs
// d(signal), and d(signal) only needs the row sums of the few dy_T rows whose bias elements fold a signal term (head:
// fc_in / fc_p_skips = 512 rows, torso: four deformation vectors = 256 rows of ~2700).  Summing those rows right after the
// dX chain (8 % of dy_T, one extra streaming read) lets that whole chain run on a side stream underneath the weight-gradient
// GEMMs instead of behind them.  One wave = 64 rows = one contiguous 4 KiB (bf16) run per tile, blockIdx.y = slice of the
// tiles, parts[slice][i]; sig_reduce_kernel adds the slices in index order (bit-reproducible).
template <typename T>
__global__ __launch_bounds__(64) void sig_rows_kernel(const int* __restrict__ row_of, int n_sig, const T* __restrict__ dy_T,
                                                      long n_tiles, int rows, float* __restrict__ parts) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int row = row_of[i];
    const long per = (n_tiles + gridDim.y - 1) / gridDim.y;
    const long t0 = blockIdx.y * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    float acc = 0.f;
#pragma unroll 4
    for (long t = t0; t < t1; ++t) {
        if constexpr (sizeof(T) == 1) {          // 16-bit tier: MX-fp8
            acc += rec8_row_sum((const unsigned char*)dy_T, t, rows, row);
        } else {
            const uint4* p = (const uint4*)(dy_T + (t * rows + row) * 32);
            uint4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = p[q];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                acc += (__uint_as_float(v[q].x) + __uint_as_float(v[q].y)) + (__uint_as_float(v[q].z) + __uint_as_float(v[q].w));
        }
    }
    parts[(long)blockIdx.y * n_sig + i] = acc;
}
// MX-fp8 arrays: one wave = one 32-row block of dy_T over a slice of the tiles.  A block of a tile is ONE contiguous KiB,
// [point n][half h][register r]: lane L takes its 16 bytes (point L >> 1, half L & 1: sixteen features of one point) with
// one dwordx4 load - 1 KiB per load instruction instead of the 64 useful bytes a row-per-lane walk gets -, dequantises with
// v_cvt_pk_f32_fp8 and the tile's E8M0 scale (the wave's scale bytes of 64 tiles are fetched by one load, v_readlane per
// tile), and accumulates sixteen sums; the 32 points are added at the end (five xor steps), fixed order: bit-reproducible.
// row_of[32 b .. 32 b + 31] must be the rows of ONE aligned block (checked by the launcher's caller: signal rows are whole
// 64-row vectors); parts[slice][i] as before.
__global__ __launch_bounds__(64) void sig_rows8_kernel(const int* __restrict__ row_of, int n_sig, const unsigned char* __restrict__ dy_T,
                                                       long n_tiles, int rows, float* __restrict__ parts) {
    __shared__ float sums[32];
    const int lane = threadIdx.x;
    const int my_row = row_of[blockIdx.x * 32 + (lane & 31)];
    const int rb = __builtin_amdgcn_readfirstlane(my_row >> 5);
    const long per = (n_tiles + gridDim.y - 1) / gridDim.y;
    const long t0 = blockIdx.y * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    const long stride = rec8_tile_bytes(rows);
    const unsigned char* blk = dy_T + (long)rb * 1024 + 16 * lane;
    const unsigned char* scl = dy_T + (long)rows * 32 + rb;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (long c = t0; c < t1; c += 64) {
        const int n = (int)((t1 - c < 64) ? t1 - c : 64);
        const unsigned sc = lane < n ? scl[(c + lane) * stride] : 127u;
        for (int u0 = 0; u0 < n; u0 += 8) {              // eight tiles' loads in flight (constant trip counts inside)
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (u0 + j < n) v[j] = *(const uint4*)(blk + (c + u0 + j) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (u0 + j >= n) break;
                const float scale = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)sc, u0 + j) << 23);
                const unsigned w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[q], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[q], true);
                    acc[4 * q + 0] = fmaf(lo[0], scale, acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(lo[1], scale, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(hi[0], scale, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(hi[1], scale, acc[4 * q + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int m = 2; m < 64; m <<= 1) acc[k] += __shfl_xor(acc[k], m, 64);
    }
    if (lane < 2) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sums[lane * 16 + k] = acc[k];
    }
    __syncthreads();
    if (lane < 32) {
        int h, r;
        tile_feat_inv(my_row & 31, &h, &r);
        parts[(long)blockIdx.y * n_sig + blockIdx.x * 32 + lane] = sums[h * 16 + r];
    }
}
// one wave per element: the lanes fetch the slices in parallel, the adds stay in index order inside a lane and in a fixed tree
// across the lanes (the serial loop was 128 dependent HBM latencies: 62 us on the critical path of the step)
__global__ __launch_bounds__(256) void sig_reduce_kernel(const int* __restrict__ elem_of, int n_sig, const float* __restrict__ parts, int slices,
                                                         float* __restrict__ dbias) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_sig) return;
    float a = 0.f;
    for (int k = lane; k < slices; k += 64) a += parts[(long)k * n_sig + i];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) dbias[elem_of[i]] = a;
}
hipError_t launch_signal_rows(int tier, int field, const int* row_of, const int* elem_of, int n_sig, const void* dy_T, long NP,
                              float* parts, float* dbias, hipStream_t st) {
    const int rows = field == FIELD_TORSO ? GradMap::S_ROWS : GradMap::H_ROWS;
    const long n_tiles = NP / 32;
    int slices = (int)(n_tiles < SIG_ROW_SLICES ? n_tiles : SIG_ROW_SLICES);
    const dim3 grid(n_sig / 64, slices);
    if (tier == TIER_BF16) {
        // few signal rows (the torso's): more, shorter slices - the partial sums' area holds SIG_ROW_SLICES x 512 floats whatever
        // n_sig is, and with 128 slices the torso's 4 row blocks were 512 waves walking 32 tiles each, two per compute unit:
        // latency-, not byte-bound (17 us for 16 MB, on the step's main stream); down to eight tiles (one batch of loads) per wave
        const long cap = (long)SIG_ROW_SLICES * 512 / n_sig, want = n_tiles / 8;
        if (want > slices) slices = (int)(want < cap ? want : cap);
    }
    if (tier == TIER_BF16)
        hipLaunchKernelGGL(sig_rows8_kernel, dim3(n_sig / 32, slices), dim3(64), 0, st, row_of, n_sig, (const unsigned char*)dy_T,
                           n_tiles, rows, parts);
    else
        hipLaunchKernelGGL(sig_rows_kernel<float>, grid, dim3(64), 0, st, row_of, n_sig, (const float*)dy_T, n_tiles, rows,
                           parts);
    hipLaunchKernelGGL(sig_reduce_kernel, dim3((n_sig + 3) / 4), dim3(256), 0, st, elem_of, n_sig, parts, slices, dbias);
    return hipGetLastError();
}

}  // namespace dfn
