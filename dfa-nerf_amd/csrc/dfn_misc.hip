// dfn_misc.hip - the small kernels around the fused renderer: weight packing, per-frame bias folding and
// the stand-alone building blocks kept for API parity with the reference's Python functions.
//
// Reference (paths under /root/reference/NeRFs/DFANeRF/): get_rays run_nerf_helpers.py:449-465;
// ndc_rays :484-503; sample_pdf :537-581; composite_function run_nerf_com_trainExpLater.py:146-166;
// calc_volume_weights :169-179; to8b run_nerf_helpers.py:17.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <hip/hip_bf16.h>
#include "dfn_layout.h"
#include "dfn_mlp.h"
#include "dfn_misc.h"

namespace dfn {

// ---- pack: gather flat params through the plan ---------------------------------------------------------
__device__ __forceinline__ void pack_body(const int* __restrict__ plan, const float* __restrict__ params, void* out, long n,
                                          int tier, long i) {
    if (i >= n) return;
    const int src = plan[i];
    const float v = src >= 0 ? params[src] : 0.f;
    if (tier == TIER_BF16) ((__bf16*)out)[i] = (__bf16)v;       // round-to-nearest-even
    else if (tier == TIER_F16) ((_Float16*)out)[i] = (_Float16)v;
    else ((float*)out)[i] = v;
}
__global__ void pack_kernel(const int* __restrict__ plan, const float* __restrict__ params, void* out, long n,
                            int tier) {
    pack_body(plan, params, out, n, tier, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
hipError_t launch_pack(const int* plan, const float* params, void* out, long n, int tier, hipStream_t st) {
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, plan, params, out, n,
                       tier);
    return hipGetLastError();
}

// ---- fold: per-frame bias vectors ----------------------------------------------------------------------------
__device__ float pval(const float* P, int pid, int r, int c) {
    return P[param_offset(pid) + r * param_shape(pid).cols + c];
}
// dot(W[pid][row, c0:c0+n], v[0:n])
__device__ float rowdot(const float* P, int pid, int row, int c0, int n, const float* v) {
    const float* w = P + param_offset(pid) + row * param_shape(pid).cols + c0;
    float a = 0.f;
    for (int k = 0; k < n; ++k) a = fmaf(w[k], v[k], a);
    return a;
}

// One WAVE per bias element (i = wave index, the same for all 64 lanes).  Blob layout = Prog<TIER>::*_B_* offsets; within a
// vector, element t*32 + h*16 + r is feature 32*t + tile_feat(h, r).  The bias layout does not depend on the tier.
// An element is a few bias parameters plus up to two dot products of 96 ... 256 terms with a weight row: the lanes stride
// over the row (coalesced 256-byte reads) and the partial sums are added by a butterfly - a fixed order, so the blob is
// reproducible.  (One THREAD per element walked its row alone, 64 uncoalesced rows per load instruction: the fold took
// 10 + 7 us per frame and most of prepare_kernel's 28 us in front of every training forward.)
__device__ void fold_body(int field, const float* __restrict__ P, const float* __restrict__ sig,
                          const float* __restrict__ zs, const float* __restrict__ za, float* out, int n, int i) {
    using PG = Prog<TIER_BF16>;
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    // lane 0 carries the terms that are not sums over a row; every lane its share of the rows
    auto pval = [&](const float* P_, int pid, int r, int c) { return lane == 0 ? dfn::pval(P_, pid, r, c) : 0.f; };
    auto rowdot = [&](const float* P_, int pid, int row, int c0, int nn, const float* vv) {
        const float* w = P_ + param_offset(pid) + row * param_shape(pid).cols + c0;
        float a = 0.f;
        for (int k = lane; k < nn; k += 64) a = fmaf(w[k], vv[k], a);
        return a;
    };
    // locate the vector this element belongs to
    int base, f;
    auto feat_of = [](int e) { return 32 * (e >> 5) + tile_feat((e >> 4) & 1, e & 15); };
    float v = 0.f;
    if (field != FIELD_TORSO) {
        const bool lis = (field == 2);
        const int in_w = lis ? P_FCINL_W : P_FCIN_W, in_b = lis ? P_FCINL_B : P_FCIN_B;
        const int sk_w = lis ? P_FCPSKL_W : P_FCPSK_W, sk_b = lis ? P_FCPSKL_B : P_FCPSK_B;
        if (i < PG::H_B_L1) {                      // IN: fc_in.b + fc_in.W[:,60:]·sig + fc_z(z_shape)
            f = feat_of(i);
            v = pval(P, in_b, f, 0) + pval(P, P_FCZ_B, f, 0) + rowdot(P, P_FCZ_W, f, 0, ZDIM, zs);
            if (!lis) v += rowdot(P, in_w, f, NPE, NSIG, sig);
        } else if (i < PG::H_B_SKIP) {             // L1..L4: blocks[0..3].bias
            base = i - PG::H_B_L1;
            f = feat_of(base & 255);
            v = pval(P, P_BLK0_B + 2 * (base >> 8), f, 0);
        } else if (i < PG::H_B_L5) {               // SKIP: fc_z_skips(z) + fc_p_skips.b + W[:,60:]·sig
            f = feat_of(i - PG::H_B_SKIP);
            v = pval(P, P_FCZSK_B, f, 0) + rowdot(P, P_FCZSK_W, f, 0, ZDIM, zs) + pval(P, sk_b, f, 0);
            if (!lis) v += rowdot(P, sk_w, f, NPE, NSIG, sig);
        } else if (i < PG::H_B_VIEW) {             // L5..L7
            base = i - PG::H_B_L5;
            f = feat_of(base & 255);
            v = pval(P, P_BLK4_B + 2 * (base >> 8), f, 0);
        } else if (i < PG::H_B_OUT) {              // VIEW: feat_view.b + fc_z_view(z_app) + fc_view.b ; sigma row
            f = feat_of(i - PG::H_B_VIEW);
            if (f < 256)
                v = pval(P, P_FEATV_B, f, 0) + pval(P, P_FCZV_B, f, 0) + rowdot(P, P_FCZV_W, f, 0, ZDIM, za) +
                    pval(P, P_FCV_B, f, 0);
            else v = (f == 256) ? pval(P, P_SIGMA_B, 0, 0) : 0.f;
        } else {                                   // OUT: feat_out.b
            f = feat_of(i - PG::H_B_OUT);
            v = f < 3 ? pval(P, P_FEATO_B, f, 0) : 0.f;
        }
    } else {
        if (i < PG::T_B_IN) {                      // deformation nets: 14 vectors of 64
            const int vec = i >> 6;
            f = feat_of(i & 63);
            switch (vec) {
            case 0: v = pval(P, P_DE0_B, f, 0) + rowdot(P, P_DE0_W, f, NPE, NET, sig); break;
            case 1: v = pval(P, P_DS0_B, f, 0) + rowdot(P, P_DS0_W, f, NPE, NET, sig); break;
            case 2: v = pval(P, P_DE1_B, f, 0); break;
            case 3: v = pval(P, P_DS1_B, f, 0); break;
            case 4: v = pval(P, P_DE2_B, f, 0); break;
            case 5: v = pval(P, P_DS2_B, f, 0); break;
            case 6: v = pval(P, P_DE3_B, f, 0); break;
            case 7: v = pval(P, P_DESK_B, f, 0); break;                                        // ESKIP
            case 8: v = pval(P, P_DS3_B, f, 0); break;
            case 9: v = pval(P, P_DSSK_B, f, 0) + rowdot(P, P_DSSK_W, f, 0, NET, sig); break;  // SSKIP
            case 10: v = pval(P, P_DE4_B, f, 0); break;
            case 11: v = pval(P, P_DS4_B, f, 0); break;
            case 12: v = f < NPE ? pval(P, P_DEO_B, f, 0) : 0.f; break;                         // EO
            default: v = f < NET ? pval(P, P_DSO_B, f, 0) + (lane == 0 ? sig[f] : 0.f) : 0.f; break;   // SO + signal
            }
        } else if (i < PG::T_B_L1) {               // IN: fc_in_torso.b + fc_z(z_shape)
            f = feat_of(i - PG::T_B_IN);
            v = pval(P, P_FCINT_B, f, 0) + pval(P, P_FCZ_B, f, 0) + rowdot(P, P_FCZ_W, f, 0, ZDIM, zs);
        } else if (i < PG::T_B_SKIP) {
            base = i - PG::T_B_L1;
            f = feat_of(base & 255);
            v = pval(P, P_BLK0_B + 2 * (base >> 8), f, 0);
        } else if (i < PG::T_B_L5) {
            f = feat_of(i - PG::T_B_SKIP);
            v = pval(P, P_FCZSK_B, f, 0) + rowdot(P, P_FCZSK_W, f, 0, ZDIM, zs) + pval(P, P_FCPSKT_B, f, 0);
        } else if (i < PG::T_B_VIEW) {
            base = i - PG::T_B_L5;
            f = feat_of(base & 255);
            v = pval(P, P_BLK4_B + 2 * (base >> 8), f, 0);
        } else if (i < PG::T_B_OUT) {
            f = feat_of(i - PG::T_B_VIEW);
            if (f < 256)
                v = pval(P, P_FEATV_B, f, 0) + pval(P, P_FCZV_B, f, 0) + rowdot(P, P_FCZV_W, f, 0, ZDIM, za) +
                    pval(P, P_FCV_B, f, 0);
            else v = (f == 256) ? pval(P, P_SIGMA_B, 0, 0) : 0.f;
        } else {
            f = feat_of(i - PG::T_B_OUT);
            v = f < 3 ? pval(P, P_FEATO_B, f, 0) : 0.f;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) out[i] = v;
}
__global__ void fold_kernel(int field, const float* __restrict__ P, const float* __restrict__ sig,
                            const float* __restrict__ zs, const float* __restrict__ za, float* out, int n) {
    fold_body(field, P, sig, zs, za, out, n, (blockIdx.x * blockDim.x + threadIdx.x) >> 6);
}
// Everything a training step derives from the parameters before its forward, in ONE launch (dfn_train_prepare): the two
// fields' bias folds and their four packed weight streams (forward + transposed).  Six launches of 3-10 us each, back to
// back on one stream, cost 37 us of a 2-ms step; the jobs are independent, so one grid covers them (blocks of 256
// threads; a block's job = the range its index falls into).
__global__ __launch_bounds__(256) void prepare_kernel(PrepareJobs J) {
    int b = blockIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (b < J.pack_blocks[k]) {
            pack_body(J.plan[k], J.params, J.out[k], J.n[k], J.tier, (long)b * 256 + threadIdx.x);
            return;
        }
        b -= J.pack_blocks[k];
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        if (b < J.fold_blocks[f]) {
            fold_body(f, J.params, J.sig[f], J.zs[f], J.za[f], J.bias[f], J.nb[f], b * 4 + (threadIdx.x >> 6));
            return;
        }
        b -= J.fold_blocks[f];
    }
}
hipError_t launch_prepare(PrepareJobs J, hipStream_t st) {
    int blocks = 0;
    for (int k = 0; k < 4; ++k) blocks += (J.pack_blocks[k] = (int)((J.n[k] + 255) / 256));
    for (int f = 0; f < 2; ++f) blocks += (J.fold_blocks[f] = (J.nb[f] + 3) / 4);          // a wave per bias element
    hipLaunchKernelGGL(prepare_kernel, dim3(blocks), dim3(256), 0, st, J);
    return hipGetLastError();
}
hipError_t launch_fold(int field, const float* params, const float* sig, const float* zs, const float* za,
                       float* out, int n, hipStream_t st) {
    hipLaunchKernelGGL(fold_kernel, dim3((n + 3) / 4), dim3(256), 0, st, field, params, sig, zs, za, out, n);
    return hipGetLastError();
}

// ---- fold backward: gradient of the bias blob -> decoder parameters and the conditioning signal ------------------
// The fold is linear: bias[i] = sum of bias parameters + sum of rowdot(W[row, c0:c0+n], v).  One thread per bias
// element applies, for its g = dbias[i]: dP[bias term] += g, dW[row, c0+k] += g v[k], dv[k] += g W[row, c0+k].
// Inside one launch every parameter element is touched by exactly one thread (plain +=; launches on a stream
// serialise); dv is shared by all rows -> atomics.  z_shape / z_app are constants upstream (never passed to an
// optimizer, MAIN:522-547): no gradient is produced for them.
struct FoldGrad {      // one WAVE per bias element: lane 0 owns the bias terms, the lanes stride over a row's columns
    float* G; float g; int lane;
    __device__ void b(int pid, int f) const { if (lane == 0) G[param_offset(pid) + f] += g; }
    __device__ void w(int pid, int row, int c0, int n, const float* v) const {
        const long o = param_offset(pid) + (long)row * param_shape(pid).cols + c0;
        for (int k = lane; k < n; k += 64) G[o + k] += g * v[k];
    }
};
__global__ void fold_bwd_kernel(int field, const float* __restrict__ sig, const float* __restrict__ zs,
                                const float* __restrict__ za, const float* __restrict__ dbias, float* G, int n) {
    using PG = Prog<TIER_BF16>;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    auto feat_of = [](int e) { return 32 * (e >> 5) + tile_feat((e >> 4) & 1, e & 15); };
    const FoldGrad q = {G, dbias[i], (int)(threadIdx.x & 63)};
    int base, f;
    auto trunk_tail = [&](int i, int b_l1, int b_skip, int b_l5, int b_view, int b_out, int skb) {
        // shared by head / listener / torso: everything after IN except the signal columns of SKIP
        if (i < b_skip) {
            base = i - b_l1; f = feat_of(base & 255); q.b(P_BLK0_B + 2 * (base >> 8), f);
        } else if (i < b_l5) {
            f = feat_of(i - b_skip); q.b(P_FCZSK_B, f); q.w(P_FCZSK_W, f, 0, ZDIM, zs); q.b(skb, f);
        } else if (i < b_view) {
            base = i - b_l5; f = feat_of(base & 255); q.b(P_BLK4_B + 2 * (base >> 8), f);
        } else if (i < b_out) {
            f = feat_of(i - b_view);
            if (f < 256) { q.b(P_FEATV_B, f); q.b(P_FCZV_B, f); q.w(P_FCZV_W, f, 0, ZDIM, za); q.b(P_FCV_B, f); }
            else if (f == 256) q.b(P_SIGMA_B, 0);
        } else {
            f = feat_of(i - b_out); if (f < 3) q.b(P_FEATO_B, f);
        }
    };
    if (field != FIELD_TORSO) {
        const bool lis = (field == 2);
        const int in_w = lis ? P_FCINL_W : P_FCIN_W, in_b = lis ? P_FCINL_B : P_FCIN_B;
        const int sk_w = lis ? P_FCPSKL_W : P_FCPSK_W, sk_b = lis ? P_FCPSKL_B : P_FCPSK_B;
        if (i < PG::H_B_L1) {
            f = feat_of(i); q.b(in_b, f); q.b(P_FCZ_B, f); q.w(P_FCZ_W, f, 0, ZDIM, zs);
            if (!lis) q.w(in_w, f, NPE, NSIG, sig);
        } else {
            trunk_tail(i, PG::H_B_L1, PG::H_B_SKIP, PG::H_B_L5, PG::H_B_VIEW, PG::H_B_OUT, sk_b);
            if (!lis && i >= PG::H_B_SKIP && i < PG::H_B_L5) q.w(sk_w, feat_of(i - PG::H_B_SKIP), NPE, NSIG, sig);
        }
    } else {
        if (i < PG::T_B_IN) {
            const int vec = i >> 6;
            f = feat_of(i & 63);
            switch (vec) {
            case 0: q.b(P_DE0_B, f); q.w(P_DE0_W, f, NPE, NET, sig); break;
            case 1: q.b(P_DS0_B, f); q.w(P_DS0_W, f, NPE, NET, sig); break;
            case 2: q.b(P_DE1_B, f); break;
            case 3: q.b(P_DS1_B, f); break;
            case 4: q.b(P_DE2_B, f); break;
            case 5: q.b(P_DS2_B, f); break;
            case 6: q.b(P_DE3_B, f); break;
            case 7: q.b(P_DESK_B, f); break;
            case 8: q.b(P_DS3_B, f); break;
            case 9: q.b(P_DSSK_B, f); q.w(P_DSSK_W, f, 0, NET, sig); break;
            case 10: q.b(P_DE4_B, f); break;
            case 11: q.b(P_DS4_B, f); break;
            case 12: if (f < NPE) q.b(P_DEO_B, f); break;
            default: if (f < NET) q.b(P_DSO_B, f); break;
            }
        } else if (i < PG::T_B_L1) {
            f = feat_of(i - PG::T_B_IN); q.b(P_FCINT_B, f); q.b(P_FCZ_B, f); q.w(P_FCZ_W, f, 0, ZDIM, zs);
        } else {
            trunk_tail(i, PG::T_B_L1, PG::T_B_SKIP, PG::T_B_L5, PG::T_B_VIEW, PG::T_B_OUT, P_FCPSKT_B);
        }
    }
}
// d(signal)[k] = sum over the bias elements whose fold has a W[., c0 + k] * signal[k] term (+ the identity term of
// the torso's SO vector).  One wave per k, lanes stride over the elements: no atomics, one writer per output.
struct SigTerm { int base, len, pid, c0; };
__global__ void fold_bwd_sig_kernel(int field, const float* __restrict__ P, const float* __restrict__ dbias, float* dsig,
                                    bool overwrite) {
    using PG = Prog<TIER_BF16>;
    const int k = blockIdx.x, lane = threadIdx.x;
    auto feat_of = [](int e) { return 32 * (e >> 5) + tile_feat((e >> 4) & 1, e & 15); };
    const SigTerm head[2] = {{PG::H_B_IN, 256, P_FCIN_W, NPE}, {PG::H_B_SKIP, 256, P_FCPSK_W, NPE}};
    const SigTerm torso[3] = {{PG::T_B_E0, 64, P_DE0_W, NPE}, {PG::T_B_S0, 64, P_DS0_W, NPE}, {PG::T_B_SSKIP, 64, P_DSSK_W, 0}};
    const bool t = field == FIELD_TORSO;
    float acc = 0.f;
    for (int q = 0; q < (t ? 3 : 2); ++q) {
        const SigTerm m = t ? torso[q] : head[q];
        const long o = param_offset(m.pid) + m.c0 + k;
        const int cols = param_shape(m.pid).cols;
        for (int e = lane; e < m.len; e += 64) acc += dbias[m.base + e] * P[o + (long)feat_of(e) * cols];
    }
    if (t)      // SO: bias = out_signal.bias + signal  (identity)
        for (int e = lane; e < 64; e += 64) if (feat_of(e) == k) acc += dbias[PG::T_B_SO + e];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) dsig[k] = overwrite ? acc : dsig[k] + acc;
}
hipError_t launch_fold_bwd(int field, const float* params, const float* sig, const float* zs, const float* za,
                           const float* dbias, float* grad_flat, float* dsig, int n, hipStream_t st) {
    hipLaunchKernelGGL(fold_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, st, field, sig, zs, za, dbias, grad_flat, n);
    if (dsig && field != 2)
        hipLaunchKernelGGL(fold_bwd_sig_kernel, dim3(field == FIELD_TORSO ? NET : NSIG), dim3(64), 0, st, field, params,
                           dbias, dsig, false);
    return hipGetLastError();
}
// d(signal) alone (dfn_signal_grad): dbias needs to hold the elements sig_term_elements() lists, nothing else
hipError_t launch_fold_bwd_sig(int field, const float* params, const float* dbias, float* dsig, bool overwrite,
                               hipStream_t st) {
    hipLaunchKernelGGL(fold_bwd_sig_kernel, dim3(field == FIELD_TORSO ? NET : NSIG), dim3(64), 0, st, field, params, dbias,
                       dsig, overwrite);
    return hipGetLastError();
}
// the bias-blob elements fold_bwd_sig_kernel reads (the vectors whose fold has a signal term), ascending; 512 / 256
int sig_term_elements(int field, int* out) {
    using PG = Prog<TIER_BF16>;
    int n = 0;
    auto run = [&](int base, int len) { for (int e = 0; e < len; ++e) out[n++] = base + e; };
    if (field == FIELD_TORSO) { run(PG::T_B_E0, 64); run(PG::T_B_S0, 64); run(PG::T_B_SSKIP, 64); run(PG::T_B_SO, 64); }
    else { run(PG::H_B_IN, 256); run(PG::H_B_SKIP, 256); }
    return n;
}


// ---- get_rays / ndc_rays ----------------------------------------------------------------------------------------
struct Pose12 { float m[12]; };
// torch.linspace(0, end, n)[i] in f32 (ATen linspace_kernel; oracle/dfa_oracle.py: linspace): step = end / (n - 1); lower half step * i,
// upper half end - step * (n - 1 - i) as ONE fused multiply-add.  n = end + 1 (stride 1): the integers, exactly.
__device__ __forceinline__ float linspace0(int i, int n, float end) {
    if (n <= 1) return 0.f;
    const float step = __fdiv_rn(end, (float)(n - 1));
    return (i < n / 2) ? __fmul_rn(step, (float)i) : fmaf(-step, (float)(n - 1 - i), end);
}
// H, W: the image; Hn x Wn = (H / stride) x (W / stride) rays through linspace(0, W - 1, Wn) x linspace(0, H - 1, Hn) (HELP:451)
__global__ void get_rays_kernel(int H, int W, int Hn, int Wn, float focal, float cx, float cy, Pose12 c2w, float* ro, float* rd) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= Hn * Wn) return;
    const int yi = pix / Wn, xi = pix - yi * Wn;
    const float x = linspace0(xi, Wn, (float)(W - 1)), y = linspace0(yi, Hn, (float)(H - 1));
    const float dx = __fdiv_rn(__fsub_rn(x, cx), focal);
    const float dy = __fdiv_rn(-__fsub_rn(y, cy), focal);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, c2w.m[4 * k]), __fmul_rn(dy, c2w.m[4 * k + 1])),
                                  __fmul_rn(-1.0f, c2w.m[4 * k + 2]));
        rd[(size_t)pix * 3 + k] = d;
        ro[(size_t)pix * 3 + k] = c2w.m[4 * k + 3];
    }
}
hipError_t launch_get_rays(int H, int W, int stride, float focal, float cx, float cy, const float* c2w_host, float* ro,
                           float* rd, hipStream_t st) {
    Pose12 p;
    for (int i = 0; i < 12; ++i) p.m[i] = c2w_host[i];
    const int Hn = H / stride, Wn = W / stride;
    hipLaunchKernelGGL(get_rays_kernel, dim3((Hn * Wn + 255) / 256), dim3(256), 0, st, H, W, Hn, Wn, focal, cx, cy, p, ro, rd);
    return hipGetLastError();
}

__global__ void ndc_rays_kernel(float kx, float ky, float z_near, const float* ro, const float* rd, long n,
                                float* oo, float* od) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o0 = ro[i * 3], o1 = ro[i * 3 + 1], o2 = ro[i * 3 + 2];
    const float d0 = rd[i * 3], d1 = rd[i * 3 + 1], d2 = rd[i * 3 + 2];
    const float t = __fdiv_rn(-__fadd_rn(z_near, o2), d2);
    const float p0 = __fadd_rn(o0, __fmul_rn(t, d0)), p1 = __fadd_rn(o1, __fmul_rn(t, d1)),
                p2 = __fadd_rn(o2, __fmul_rn(t, d2));
    oo[i * 3 + 0] = __fdiv_rn(__fmul_rn(kx, p0), p2);
    oo[i * 3 + 1] = __fdiv_rn(__fmul_rn(ky, p1), p2);
    oo[i * 3 + 2] = __fadd_rn(1.0f, __fdiv_rn(__fmul_rn(2.0f, z_near), p2));
    od[i * 3 + 0] = __fmul_rn(kx, __fsub_rn(__fdiv_rn(d0, d2), __fdiv_rn(p0, p2)));
    od[i * 3 + 1] = __fmul_rn(ky, __fsub_rn(__fdiv_rn(d1, d2), __fdiv_rn(p1, p2)));
    od[i * 3 + 2] = __fdiv_rn(__fmul_rn(-2.0f, z_near), p2);
}
hipError_t launch_ndc_rays(int H, int W, float focal, float z_near, const float* ro, const float* rd, long n,
                           float* oo, float* od, hipStream_t st) {
    // -1./(W/(2.*focal)) evaluated in double like the Python scalar expression, then used as an f32 scalar
    const float kx = (float)(-1.0 / ((double)W / (2.0 * (double)focal)));
    const float ky = (float)(-1.0 / ((double)H / (2.0 * (double)focal)));
    hipLaunchKernelGGL(ndc_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kx, ky, z_near, ro, rd,
                       n, oo, od);
    return hipGetLastError();
}

// ---- sample_pdf: one wavefront per ray -----------------------------------------------------------------------------
__device__ float linspace01_(int i, int n) {
    const float step = __fdiv_rn(1.0f, (float)(n - 1));
    return (i < n / 2) ? __fmul_rn(step, (float)i) : fmaf(-step, (float)(n - 1 - i), 1.0f);
}
__global__ void sample_pdf_kernel(const float* bins, const float* weights, long R, int nb, int ns, const float* u_in,
                                  float* out) {
    __shared__ float cdf_s[4][256];
    __shared__ float bin_s[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * 4 + w;
    if (ray >= R) return;                       // whole wave exits together
    float* cdf = cdf_s[w];
    float* bn = bin_s[w];
    const int nw = nb - 1;
    // weights + 1e-5, sum
    float part = 0.f;
    for (int k = lane; k < nw; k += 64) part += __fadd_rn(weights[ray * nw + k], 1e-5f);
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    for (int k = lane; k < nw; k += 64) cdf[k + 1] = __fdiv_rn(__fadd_rn(weights[ray * nw + k], 1e-5f), part);
    for (int k = lane; k < nb; k += 64) bn[k] = bins[ray * nb + k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {      // torch.cumsum on the CPU (the reference's goldens): sequential, accumulator in DOUBLE
        double c = 0.0;   // (at::acc_type<float, false>), every prefix rounded to f32 on output
        cdf[0] = 0.f;
        for (int k = 1; k < nb; ++k) {
            c += (double)cdf[k];
            cdf[k] = (float)c;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = lane; j < ns; j += 64) {
        const float u = u_in ? u_in[ray * ns + j] : linspace01_(j, ns);
        int lo = 0, hi = nb;                    // first index with cdf > u  == searchsorted(right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, nb - 1);
        float den = __fsub_rn(cdf[above], cdf[below]);
        if (den < 1e-5f) den = 1.0f;
        const float t = __fdiv_rn(__fsub_rn(u, cdf[below]), den);
        out[ray * ns + j] = __fadd_rn(bn[below], __fmul_rn(t, __fsub_rn(bn[above], bn[below])));
    }
}
hipError_t launch_sample_pdf(const float* bins, const float* weights, long R, int nb, int ns, const float* u,
                             float* out, hipStream_t st) {
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, bins, weights, R, nb, ns, u,
                       out);
    return hipGetLastError();
}

// ---- composite_function ------------------------------------------------------------------------------------------------
__global__ void composite_kernel(const float* sigma, const float* feat, int K, long N, float* ssum, float* fw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (K == 1) {
        ssum[i] = sigma[i];
        for (int c = 0; c < 3; ++c) fw[i * 3 + c] = feat[i * 3 + c];
        return;
    }
    float tot = 0.f;
    for (int k = 0; k < K; ++k) tot = __fadd_rn(tot, sigma[k * N + i]);
    const float den = (tot == 0.f) ? 1e-4f : tot;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const float w = __fdiv_rn(sigma[k * N + i], den);
        for (int c = 0; c < 3; ++c) acc[c] = __fadd_rn(acc[c], __fmul_rn(feat[(k * N + i) * 3 + c], w));
    }
    ssum[i] = tot;
    for (int c = 0; c < 3; ++c) fw[i * 3 + c] = acc[c];
}
hipError_t launch_composite(const float* sigma, const float* feat, int K, long N, float* ssum, float* fw,
                            hipStream_t st) {
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, sigma, feat, K, N, ssum,
                       fw);
    return hipGetLastError();
}

// ---- calc_volume_weights: one wavefront per ray, sequential-in-lane cumprod blocks + wave scan -----------------------
__global__ void volume_weights_kernel(const float* z, const float* ray, const float* sigma, long R, int S,
                                      float last_dist, float* wout) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + w;
    if (r >= R) return;
    const float dx = ray[r * 3], dy = ray[r * 3 + 1], dzv = ray[r * 3 + 2];
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dzv, dzv)));
    const int per = (S + 63) / 64;             // consecutive samples per lane
    const int s0 = lane * per;
    float alpha[16], pre[16];
    float prod = 1.0f;
    for (int k = 0; k < per && k < 16; ++k) {
        const int s = s0 + k;
        float a = 0.f;
        if (s < S) {
            const float d = (s == S - 1) ? last_dist : __fsub_rn(z[r * S + s + 1], z[r * S + s]);
            const float dist = __fmul_rn(d, nrm);
            a = __fsub_rn(1.0f, expf(-__fmul_rn(__fadd_rn(fmaxf(sigma[r * S + s], 0.f), 1e-6f), dist)));
        }
        alpha[k] = a;
        pre[k] = prod;                         // product of this lane's earlier factors
        if (s < S) prod = prod * __fadd_rn(__fsub_rn(1.0f, a), 1e-10f);
    }
    float inc = prod;                          // inclusive scan of lane products
    for (int d = 1; d < 64; d <<= 1) {
        const float up = __shfl_up(inc, d);
        if (lane >= d) inc *= up;
    }
    float exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 1.0f;
    for (int k = 0; k < per && k < 16; ++k) {
        const int s = s0 + k;
        if (s < S) wout[r * S + s] = alpha[k] * (exc * pre[k]);
    }
}
hipError_t launch_volume_weights(const float* z, const float* ray, const float* sigma, long R, int S,
                                 float last_dist, float* w, hipStream_t st) {
    hipLaunchKernelGGL(volume_weights_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, z, ray, sigma, R, S,
                       last_dist, w);
    return hipGetLastError();
}

// ---- backward of composite_function / calc_volume_weights (the stand-alone building blocks under autograd) ----------------
// What torch autograd computes through run_nerf_com_trainExpLater.py:146-179 as a reference-shaped training loop calls them
// (MAIN:888-899).  composite_function: w_k = sigma_k / den with den = sum_k sigma_k, and den := 1e-4 written IN PLACE where the
// sum is 0 - an index_put_ of a constant, so no gradient reaches sigma through den at those samples.
__global__ void composite_grad_kernel(const float* sigma, const float* feat, int K, long N, const float* d_ssum,
                                      const float* d_fw, float* d_sigma, float* d_feat) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float gs = d_ssum ? d_ssum[i] : 0.f;
    float g[3] = {0.f, 0.f, 0.f};
    if (d_fw)
        for (int c = 0; c < 3; ++c) g[c] = d_fw[i * 3 + c];
    if (K == 1) {
        d_sigma[i] = gs;
        for (int c = 0; c < 3; ++c) d_feat[i * 3 + c] = g[c];
        return;
    }
    float tot = 0.f;
    for (int k = 0; k < K; ++k) tot = __fadd_rn(tot, sigma[k * N + i]);
    const bool replaced = tot == 0.f;
    const float den = replaced ? 1e-4f : tot, inv = 1.0f / den;
    float through_den = 0.f;                      // sum_k <g, feat_k> sigma_k / den^2
    for (int k = 0; k < K; ++k) {
        const float* f = feat + (k * N + i) * 3;
        through_den += (g[0] * f[0] + g[1] * f[1] + g[2] * f[2]) * sigma[k * N + i];
    }
    through_den = replaced ? 0.f : through_den * inv * inv;
    for (int k = 0; k < K; ++k) {
        const float* f = feat + (k * N + i) * 3;
        const float w = sigma[k * N + i] * inv;
        d_sigma[k * N + i] = gs + (g[0] * f[0] + g[1] * f[1] + g[2] * f[2]) * inv - through_den;
        for (int c = 0; c < 3; ++c) d_feat[(k * N + i) * 3 + c] = w * g[c];
    }
}
hipError_t launch_composite_grad(const float* sigma, const float* feat, int K, long N, const float* d_ssum, const float* d_fw,
                                 float* d_sigma, float* d_feat, hipStream_t st) {
    hipLaunchKernelGGL(composite_grad_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, sigma, feat, K, N, d_ssum,
                       d_fw, d_sigma, d_feat);
    return hipGetLastError();
}

// calc_volume_weights: w_s = alpha_s T_s, T_s = prod_{j<s} f_j, f_j = 1 - alpha_j + 1e-10.
//   d alpha_s = T_s (g_s - G_s),  G_s = sum_{j>s} g_j alpha_j prod_{s<m<j} f_m = b_{s+1} + f_{s+1} G_{s+1}  (b_j = g_j alpha_j)
// (products of the f BETWEEN s and j only: no division by an f that may be 1e-10).  One wavefront per ray like the forward:
// each lane owns `per` consecutive samples, the affine maps x -> b + f x are composed inside the lane and scanned across the
// wave from the far end.  d sigma_s = [sigma_s > 0] d alpha_s dist_s exp(-(relu(sigma_s) + 1e-6) dist_s).
__global__ void volume_weights_grad_kernel(const float* z, const float* ray, const float* sigma, long R, int S,
                                           float last_dist, const float* d_w, float* d_sigma) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + w;
    if (r >= R) return;
    const float dx = ray[r * 3], dy = ray[r * 3 + 1], dzv = ray[r * 3 + 2];
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dzv, dzv)));
    const int per = (S + 63) / 64;
    const int s0 = lane * per;
    float alpha[16], pre[16], dist[16], ex[16], gw[16];
    float prod = 1.0f;
    for (int k = 0; k < per && k < 16; ++k) {
        const int s = s0 + k;
        float a = 0.f, e = 1.f, dd = 0.f, g = 0.f;
        if (s < S) {
            const float d = (s == S - 1) ? last_dist : __fsub_rn(z[r * S + s + 1], z[r * S + s]);
            dd = __fmul_rn(d, nrm);
            e = expf(-__fmul_rn(__fadd_rn(fmaxf(sigma[r * S + s], 0.f), 1e-6f), dd));
            a = __fsub_rn(1.0f, e);
            g = d_w[r * S + s];
        }
        alpha[k] = a; dist[k] = dd; ex[k] = e; gw[k] = g;
        pre[k] = prod;
        if (s < S) prod = prod * __fadd_rn(__fsub_rn(1.0f, a), 1e-10f);
    }
    float inc = prod;                          // T: inclusive scan of the lane products, as in the forward
    for (int d = 1; d < 64; d <<= 1) {
        const float up = __shfl_up(inc, d);
        if (lane >= d) inc *= up;
    }
    float exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 1.0f;
    // the lane's composed map x -> B + F x over its samples, first sample outermost
    float F = 1.0f, B = 0.f;
    for (int k = per < 16 ? per - 1 : 15; k >= 0; --k) {
        const int s = s0 + k;
        if (s < S) {
            const float f = __fadd_rn(__fsub_rn(1.0f, alpha[k]), 1e-10f), b = gw[k] * alpha[k];
            B = b + f * B;                     // M_s o (F, B)
            F = f * F;
        }
    }
    float sF = F, sB = B;                      // suffix composition over lanes >= this one
    for (int d = 1; d < 64; d <<= 1) {
        const float oF = __shfl_down(sF, d), oB = __shfl_down(sB, d);
        if (lane + d < 64) { sB = sB + sF * oB; sF = sF * oF; }
    }
    float G = __shfl_down(sB, 1);              // G of this lane's LAST sample = (maps of all later lanes)(0)
    if (lane == 63) G = 0.f;
    for (int k = per < 16 ? per - 1 : 15; k >= 0; --k) {
        const int s = s0 + k;
        if (s < S) {
            const float T = exc * pre[k];
            const float d_alpha = T * (gw[k] - G);
            d_sigma[r * S + s] = sigma[r * S + s] > 0.f ? d_alpha * dist[k] * ex[k] : 0.f;
            G = gw[k] * alpha[k] + __fadd_rn(__fsub_rn(1.0f, alpha[k]), 1e-10f) * G;
        }
    }
}
hipError_t launch_volume_weights_grad(const float* z, const float* ray, const float* sigma, long R, int S, float last_dist,
                                      const float* d_w, float* d_sigma, hipStream_t st) {
    hipLaunchKernelGGL(volume_weights_grad_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, z, ray, sigma, R, S,
                       last_dist, d_w, d_sigma);
    return hipGetLastError();
}

// ---- zero fill (dfn_zero_async) -------------------------------------------------------------------------------------------
__global__ void zero_words_kernel(unsigned* p, long n) {
    const long n4 = n >> 2;
    uint4* q = (uint4*)p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) q[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0;
}
hipError_t launch_zero_words(unsigned* p, long n, hipStream_t st) {
    const long n4 = n >> 2;
    const int blocks = (int)std::min<long>(1024, std::max<long>(1, (n4 + 255) / 256));
    hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, st, p, n);
    return hipGetLastError();
}

// ---- power-ceiling probe (dfn_debug_mfma_chain) -----------------------------------------------------------------------------
// render_kernel's inner loop reduced to its cost drivers (tools/mfma_power_probe.hip is the stand-alone original, LABNOTES.md 4.6):
// 8 waves per workgroup, one workgroup per compute unit (150 KiB of LDS requested), every wave issuing the tier's 32x32x16 MFMA
// on two alternating accumulator sets with LDS2 1-KiB fragment reads and VALU2 convert / max instructions per TWO MFMAs, on the
// caller's operands.  (0, 0) = the bare chain: what the matrix pipe sustains on those operands under the chip's power limit;
// (2, 4) = the renderer's instruction mix.
typedef unsigned pc_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pc_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pc_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16, int LDS2, int VALU2>
__global__ __launch_bounds__(512) void mfma_chain_kernel(const pc_u32x4* __restrict__ frag_src, const pc_u32x4* __restrict__ b_src,
                                                         int iters, float* out, unsigned long long* clk) {
    extern __shared__ pc_u32x4 pc_slab[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 64; i += 512) pc_slab[i] = frag_src[i];
    __syncthreads();
    pc_u32x4 b[4];
    for (int k = 0; k < 4; ++k) b[k] = b_src[(threadIdx.x * 4 + k) & 4095];
    f32x16 acc0 = {}, acc1 = {};
    pc_u32x4 a0 = pc_slab[lane], a1 = pc_slab[64 + lane];
    unsigned e0 = b[0][0], e1 = b[1][1], e2 = b[2][2], e3 = b[3][3];
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {          // 32 MFMAs per iteration
            const int f = (2 * k) % 32;
            if (LDS2 >= 1) a0 = pc_slab[f * 64 + lane];
            if (LDS2 >= 2) a1 = pc_slab[(f + 1) * 64 + lane];
            if constexpr (F16) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pc_f16x8, a0), __builtin_bit_cast(pc_f16x8, b[k & 3]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pc_f16x8, a1), __builtin_bit_cast(pc_f16x8, b[(k + 1) & 3]), acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pc_bf16x8, a0), __builtin_bit_cast(pc_bf16x8, b[k & 3]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pc_bf16x8, a1), __builtin_bit_cast(pc_bf16x8, b[(k + 1) & 3]), acc1, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < VALU2; ++v) {
                unsigned& x = (v & 3) == 0 ? e0 : (v & 3) == 1 ? e1 : (v & 3) == 2 ? e2 : e3;
                if (v & 1) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(e0 ^ e2));
                else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(__builtin_bit_cast(float, x)), "v"(__builtin_bit_cast(float, x ^ 0x3f800000u)));
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum + (float)(e0 ^ e1 ^ e2 ^ e3);
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
hipError_t launch_mfma_chain(bool f16, int lds2, int valu2, const void* frags, const void* b, int iters, int blocks, float* out,
                             unsigned long long* clk, hipStream_t st) {
    constexpr int lds = 150 * 1024;
#define PC_GO(F, L, V)                                                                                                        \
    {                                                                                                                         \
        hipError_t e = hipFuncSetAttribute((const void*)mfma_chain_kernel<F, L, V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e != hipSuccess) return e;                                                                                        \
        hipLaunchKernelGGL((mfma_chain_kernel<F, L, V>), dim3(blocks), dim3(512), lds, st, (const pc_u32x4*)frags,              \
                           (const pc_u32x4*)b, iters, out, clk);                                                              \
        return hipGetLastError();                                                                                             \
    }
    if (lds2 == 0 && valu2 == 0) { if (f16) PC_GO(true, 0, 0) else PC_GO(false, 0, 0) }
    if (lds2 == 2 && valu2 == 4) { if (f16) PC_GO(true, 2, 4) else PC_GO(false, 2, 4) }
    // what a different register blocking would change (tools/power_mix_sweep.py): half the fragment reads (one per two MFMAs:
    // two point tiles per weight fragment), no fragment reads, no epilogue instructions
    if (lds2 == 1 && valu2 == 4) { if (f16) PC_GO(true, 1, 4) else PC_GO(false, 1, 4) }
    if (lds2 == 0 && valu2 == 4) { if (f16) PC_GO(true, 0, 4) else PC_GO(false, 0, 4) }
    if (lds2 == 2 && valu2 == 0) { if (f16) PC_GO(true, 2, 0) else PC_GO(false, 2, 0) }
    if (lds2 == 1 && valu2 == 2) { if (f16) PC_GO(true, 1, 2) else PC_GO(false, 1, 2) }
    if (lds2 == 2 && valu2 == 2) { if (f16) PC_GO(true, 2, 2) else PC_GO(false, 2, 2) }
#undef PC_GO
    return hipErrorInvalidValue;
}

// ---- to8b ---------------------------------------------------------------------------------------------------------------
__global__ void to8b_kernel(const float* x, long n, unsigned char* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = fminf(fmaxf(x[i], 0.f), 1.f);
    out[i] = (unsigned char)(int)__fmul_rn(255.0f, v);      // truncation, like numpy astype(uint8)
}
hipError_t launch_to8b(const float* x, long n, unsigned char* out, hipStream_t st) {
    hipLaunchKernelGGL(to8b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, out);
    return hipGetLastError();
}

// ---- MFMA fragment-map probe ------------------------------------------------------------------------------------------------
// A[i][k] = 1 + i + 64*kslot ... chosen so that D[i][j] identifies both the row/col map and the (half, slot)
// pairing: A(i, half h, slot e) = (i+1) * (1 if (h,e)==(H0,E0) else 0), B(j, h, e) likewise with (j+1)*...
// -> D[i][j] = (i+1)*(j+1) iff the same (half, slot) of A and B are paired; written out through the map
// the kernels assume (row = tile_feat(h, r), col = lane&31).
__global__ void mfma_probe_kernel(float* out) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    // bf16: slot (h=1, e=5)
    {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (__bf16)((h == 1 && e == 5) ? (float)(i + 1) : 0.f);
            b[e] = (__bf16)((h == 1 && e == 5) ? (float)(2 * i + 1) : 0.f);
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out[(tile_feat(h, r)) * 32 + i] = c[r];
    }
    // f32: k-slot h == 1
    {
        const float a = (h == 1) ? (float)(i + 1) : 0.f, b = (h == 1) ? (float)(2 * i + 1) : 0.f;
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out[1024 + (tile_feat(h, r)) * 32 + i] = c[r];
    }
    // f16: slot (h=1, e=5), the bf16 probe with f16 operands
    {
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (_Float16)((h == 1 && e == 5) ? (float)(i + 1) : 0.f);
            b[e] = (_Float16)((h == 1 && e == 5) ? (float)(2 * i + 1) : 0.f);
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out[2048 + (tile_feat(h, r)) * 32 + i] = c[r];
    }
}
hipError_t launch_mfma_probe(float* out, hipStream_t st) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, st, out);
    return hipGetLastError();
}


// ---- Adam over a list of tensors (run_nerf_com_trainExpLater.py:522-547: torch.optim.Adam, betas (0.9, 0.999)) -------
// torch's own multi-tensor kernel hands one block a 64 K-element chunk: the decoder's 68 tensors (955 k parameters)
// make ~80 blocks on 256 CUs and take 104 us per step; with 2048-element chunks the same update is ~600 blocks and
// bound by its 27 MB of traffic.  One block = one chunk of one tensor (chunk table built by the host once).
__global__ __launch_bounds__(256) void adam_multi_kernel(const DfnAdamItem* items, const int2* chunks, float lr, float beta2,
                                                          float om_beta1, float om_beta2, float eps, float bias_c1,
                                                          float bias_c2_sqrt) {
    const int2 ch = chunks[blockIdx.x];
    const DfnAdamItem it = items[ch.x];
    const long i0 = (long)ch.y * DFN_ADAM_CHUNK;
    const long i1 = (i0 + DFN_ADAM_CHUNK < it.n) ? i0 + DFN_ADAM_CHUNK : it.n;
    const float step_size = lr / bias_c1;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float g = it.grad[i];
        float m = it.exp_avg[i], v = it.exp_avg_sq[i];
        m = m + (g - m) * om_beta1;                             // lerp, as torch; 1 - beta rounded from double (host)
        v = beta2 * v + om_beta2 * g * g;
        const float denom = sqrtf(v) / bias_c2_sqrt + eps;
        it.param[i] -= step_size * m / denom;
        it.exp_avg[i] = m;
        it.exp_avg_sq[i] = v;
    }
}
hipError_t launch_adam_multi(const DfnAdamItem* items, const void* chunks, int n_chunks, float lr, double beta1, double beta2,
                             float eps, float bias_c1, float bias_c2_sqrt, hipStream_t st) {
    if (n_chunks <= 0) return hipSuccess;
    // 1 - beta in double first: 1.0f - 0.999f is off by 1.3e-5 relative, which would scale exp_avg_sq
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, st, items, (const int2*)chunks, lr, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), eps, bias_c1, bias_c2_sqrt);
    return hipGetLastError();
}

}  // namespace dfn
