// dfn_signal.hip - the per-frame conditioning signals as two fused kernels, forward only
// (inference; training differentiates the torch twins in dfanerf/nets.py).
//
// Reference (paths under /root/reference/NeRFs/DFANeRF/):
//   encode_signal        run_nerf_com_trainExpLater.py:28-75   (window + zero padding :36-57)
//   encode_signal_torso  run_nerf_com_trainExpLater.py:78-111
//   rot_to_euler / pose_to_euler_trans  :182-204
//   AudioNet_W2L  run_nerf_helpers.py:165-178   ExpressionEnc :182-193   AudioAttNet :210-240
//   Embedder / get_embedder(3, 0)  run_nerf_helpers.py:21-70
//
// One workgroup (4 waves) = one frame.  Linear layers: a wave owns an output feature, its lanes stride over the
// inputs (coalesced weight reads), the window's rows share every weight load.  Everything else is a few hundred
// FLOPs on LDS.  Parameters arrive flattened in state_dict order per network.
#include <hip/hip_runtime.h>
#include "dfn_signal.h"

namespace dfn {

constexpr int SIG_MAX_WIN = 8;       // largest attention window (smo_size / smo_torse_size)
constexpr int SIG_KEEP_ACT = 8 * (256 + 128 + 32 + 96);      // floats of kept activations; behind them: d h1 [8][256] | d h2 [8][128]
constexpr int SIG_THREADS = 1024;     // 16 waves: the kernels are latency-bound chains of small layers

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.02f * x; }

// y[t][o] = act(b[o] + sum_k W[o][k] x[t][k]),  x: LDS [S][K], y: LDS [S][M].  A wave owns output rows; all of a row's
// weights are requested before the first FMA (K/64 <= 8 independent loads in flight: the layer is latency-bound).
template <int K>
__device__ void linear_rows(const float* __restrict__ W, const float* __restrict__ b, int M, const float* x, float* y,
                            int S, bool act) {
    constexpr int KU = (K + 63) / 64;
    // R output rows per iteration: their R * KU weight loads are all in flight together (the layer is a chain of
    // first-touch HBM / L2 latencies on ONE compute unit; with one row at a time AudioNet's 512 -> 256 layer alone took
    // 16 round trips per wave)
    // (measured: R = 4 / 8 made the kernels SLOWER, 77 -> 100 us forward, 181 -> 216 us backward: fewer waves have work -
    // 256 rows / 16 waves / 4 - and the first-touch latency is paid per wave anyway; R = 1 stays)
    constexpr int R = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int o0 = wave * R; o0 < M; o0 += nw * R) {
        float wv[R][KU], bo[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int o = o0 + r;
            const float* w = W + (long)o * K;
#pragma unroll
            for (int u = 0; u < KU; ++u) wv[r][u] = (o < M && lane + 64 * u < K) ? w[lane + 64 * u] : 0.f;
            bo[r] = o < M ? b[o] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int o = o0 + r;
            if (o >= M) break;
#pragma unroll
            for (int t = 0; t < SIG_MAX_WIN; ++t) {
                if (t < S) {
                    float a = 0.f;
#pragma unroll
                    for (int u = 0; u < KU; ++u)
                        if (lane + 64 * u < K) a = fmaf(wv[r][u], x[t * K + lane + 64 * u], a);
                    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
                    if (lane == 0) {
                        a += bo[r];
                        y[t * M + o] = act ? leaky(a) : a;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// number of parameters of AudioAttNet(D, S): convs D->16->8->4->2->1 (k3) + Linear(S,S)
__device__ __host__ constexpr int att_param_count(int D, int S) {
    return 16 * D * 3 + 16 + 8 * 16 * 3 + 8 + 4 * 8 * 3 + 4 + 2 * 4 * 3 + 2 + 1 * 2 * 3 + 1 + S * S + S;
}
constexpr int ATT_PARAMS_MAX = att_param_count(96, SIG_MAX_WIN);
// the attention net's few thousand parameters are read many times by few threads: stage them in LDS
__device__ const float* stage_att(const float* __restrict__ P, float* dst, int D, int S) {
    const int n = att_param_count(D, S);
    for (int e = threadIdx.x; e < n; e += blockDim.x) dst[e] = P[e];
    __syncthreads();
    return dst;
}

// AudioAttNet on the window feat [S][D] (LDS): conv stack D->16->8->4->2->1 (k3, p1, LeakyReLU .02 after every conv),
// Linear(S,S), softmax, weighted sum of the rows -> out[D].  P = flattened state_dict of the net.
__device__ void attention(const float* __restrict__ P, int D, int S, const float* feat, float* buf0, float* buf1,
                          float* out) {
    const int chans[6] = {D, 16, 8, 4, 2, 1};
    const float* cur = feat;        // [S][Cin] (row = time step)
    int cin_stride = D;
    float* nxt = buf0;
    long off = 0;
    for (int l = 0; l < 5; ++l) {
        const int ci = chans[l], co = chans[l + 1];
        const float* W = P + off;                 // [co][ci][3]
        const float* B = W + (long)co * ci * 3;
        for (int e = threadIdx.x; e < co * S; e += blockDim.x) {
            const int o = e / S, t = e - o * S;
            float a = B[o];
            for (int c = 0; c < ci; ++c) {
                const float* w = W + ((long)o * ci + c) * 3;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int tt = t + j - 1;
                    if (tt >= 0 && tt < S) a = fmaf(w[j], cur[tt * cin_stride + c], a);
                }
            }
            nxt[t * co + o] = leaky(a);
        }
        __syncthreads();
        off += (long)co * ci * 3 + co;
        cur = nxt;
        cin_stride = co;
        nxt = (nxt == buf0) ? buf1 : buf0;
    }
    // cur: [S][1] conv output; attentionNet.0 = Linear(S, S) then softmax over the window
    const float* LW = P + off;
    const float* LB = LW + S * S;
    __shared__ float att[SIG_MAX_WIN];
    if (threadIdx.x == 0) {
        float z[SIG_MAX_WIN], m = -3.0e38f;
        for (int i = 0; i < S; ++i) {
            float a = LB[i];
            for (int j = 0; j < S; ++j) a = fmaf(LW[i * S + j], cur[j], a);
            z[i] = a;
            m = fmaxf(m, a);
        }
        float sum = 0.f;
        for (int i = 0; i < S; ++i) { z[i] = expf(z[i] - m); sum += z[i]; }
        for (int i = 0; i < S; ++i) att[i] = z[i] / sum;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f;
        for (int t = 0; t < S; ++t) a = fmaf(att[t], feat[t * D + d], a);
        out[d] = a;
    }
}

// ---- A7 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SIG_THREADS) void encode_signal_kernel(const float* __restrict__ PA, const float* __restrict__ PE,
                                                                    const float* __restrict__ PT, const float* __restrict__ auds,
                                                                    const float* __restrict__ exps, int N,
                                                                    const int* __restrict__ frame_ids, int smo, float* out,
                                                                    float* keep) {
    extern __shared__ float lds[];
    const int S = smo > 0 ? smo : 1, half = smo / 2;
    float* xa = lds;                       // [S][512]
    float* h1 = xa + SIG_MAX_WIN * 512;    // [S][256]
    float* h2 = h1 + SIG_MAX_WIN * 256;    // [S][128]
    float* xe = h2 + SIG_MAX_WIN * 128;    // [S][64]
    float* e1 = xe + SIG_MAX_WIN * 64;     // [S][32]
    float* ft = e1 + SIG_MAX_WIN * 32;     // [S][96] = cat(AudNet, ExpNet)
    float* b0 = ft + SIG_MAX_WIN * 96;     // [S][16]
    float* b1 = b0 + SIG_MAX_WIN * 16;     // [S][16]
    float* a64 = b1 + SIG_MAX_WIN * 16;    // [S][64]
    float* e32 = a64 + SIG_MAX_WIN * 64;   // [S][32]
    float* ps = e32 + SIG_MAX_WIN * 32;    // attention parameters
    const int f = frame_ids[blockIdx.x];
    for (int e = threadIdx.x; e < S * 512; e += blockDim.x) {
        const int t = e >> 9, k = e & 511, src = smo > 0 ? f - half + t : f;
        xa[t * 512 + k] = (smo == 0 || (src >= 0 && src < N)) ? auds[(long)src * 512 + k] : 0.f;      // zero rows outside, MAIN:36-57
    }
    for (int e = threadIdx.x; e < S * 64; e += blockDim.x) {
        const int t = e >> 6, k = e & 63, src = smo > 0 ? f - half + t : f;
        xe[t * 64 + k] = (smo == 0 || (src >= 0 && src < N)) ? exps[(long)src * 64 + k] : 0.f;
    }
    __syncthreads();
    // AudioNet_W2L: 512 -> 256 -> 128 -> 64
    // (round 4, measured and not kept: this first layer - three quarters of the encoder's weights - as a 16-workgroup launch in
    // front of the chain: next to the weight-gradient GEMMs its workgroups wait for compute-unit slots, 51 + 54 us instead of 77)
    linear_rows<512>(PA, PA + 131072, 256, xa, h1, S, true);
    linear_rows<256>(PA + 131328, PA + 131328 + 32768, 128, h1, h2, S, true);
    linear_rows<128>(PA + 164224, PA + 164224 + 8192, 64, h2, a64, S, false);
    // ExpressionEnc: 64 -> 32 -> 32
    linear_rows<64>(PE, PE + 2048, 32, xe, e1, S, true);
    linear_rows<32>(PE + 2080, PE + 2080 + 1024, 32, e1, e32, S, false);
    for (int e = threadIdx.x; e < S * 96; e += blockDim.x) {
        const int t = e / 96, d = e - t * 96;
        ft[t * 96 + d] = d < 64 ? a64[t * 64 + d] : e32[t * 32 + d - 64];
    }
    __syncthreads();
    // training (one frame): the activations the backward needs, so that it does not run the three AudioNet layers - 690 KB of
    // weights through one compute unit - a second time (SIG_KEEP_FLOATS: [h1 S x 256 | h2 S x 128 | e1 S x 32 | ft S x 96])
    if (keep && blockIdx.x == 0) {
        for (int e = threadIdx.x; e < S * 256; e += blockDim.x) keep[e] = h1[e];
        for (int e = threadIdx.x; e < S * 128; e += blockDim.x) keep[SIG_MAX_WIN * 256 + e] = h2[e];
        for (int e = threadIdx.x; e < S * 32; e += blockDim.x) keep[SIG_MAX_WIN * 384 + e] = e1[e];
        for (int e = threadIdx.x; e < S * 96; e += blockDim.x) keep[SIG_MAX_WIN * 416 + e] = ft[e];
    }
    float* o = out + (long)blockIdx.x * 96;
    if (smo > 0) attention(stage_att(PT, ps, 96, S), 96, S, ft, b0, b1, o);
    else
        for (int d = threadIdx.x; d < 96; d += blockDim.x) o[d] = ft[d];
}

// ---- A8 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SIG_THREADS) void encode_signal_torso_kernel(const float* __restrict__ PT,
                                                                          const float* __restrict__ poses, int pose_stride,
                                                                          int N, const int* __restrict__ frame_ids, int smo,
                                                                          float* out) {
    __shared__ float emb[SIG_MAX_WIN * 42], b0[SIG_MAX_WIN * 16], b1[SIG_MAX_WIN * 16], ps[att_param_count(42, SIG_MAX_WIN)];
    const int S = smo > 0 ? smo : 1, half = smo / 2;
    const int f = frame_ids[blockIdx.x];
    for (int e = threadIdx.x; e < S * 6; e += blockDim.x) {
        const int t = e / 6, c = e - t * 6, src = smo > 0 ? f - half + t : f;
        float v = 0.f;                                   // zero rows of (euler, trans) outside the sequence, MAIN:96-103
        if (smo == 0 || (src >= 0 && src < N)) {
            const float* R = poses + (long)src * pose_stride;        // rows of the pose matrix, 4 floats each
            if (c == 0) v = atan2f(R[2 * 4 + 2], R[1 * 4 + 2]);
            else if (c == 1) v = asinf(-R[0 * 4 + 2]);
            else if (c == 2) v = atan2f(R[0 * 4 + 0], -R[0 * 4 + 1]);
            else v = R[(c - 3) * 4 + 3];
        }
        // get_embedder(3, 0): [x, sin x, cos x, sin 2x, cos 2x, sin 4x, cos 4x] per 3-vector; two of them -> 42
        const int g = c / 3, a = c - 3 * g;
        float* o = emb + t * 42 + g * 21;
        o[a] = v;
        o[3 + a] = sinf(v);          o[6 + a] = cosf(v);
        o[9 + a] = sinf(v * 2.0f);   o[12 + a] = cosf(v * 2.0f);
        o[15 + a] = sinf(v * 4.0f);  o[18 + a] = cosf(v * 4.0f);
    }
    __syncthreads();
    float* o = out + (long)blockIdx.x * 42;
    if (smo > 0) attention(stage_att(PT, ps, 42, S), 42, S, emb, b0, b1, o);
    else
        for (int d = threadIdx.x; d < 42; d += blockDim.x) o[d] = emb[d];
}

hipError_t launch_encode_signal(const float* aud_params, const float* exp_params, const float* att_params,
                                const float* auds, const float* exps, int N, const int* frame_ids, int n_frames, int smo,
                                float* out, float* keep, hipStream_t st) {
    const size_t lds = sizeof(float) * (SIG_MAX_WIN * (512 + 256 + 128 + 64 + 32 + 96 + 16 + 16 + 64 + 32) + ATT_PARAMS_MAX);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)encode_signal_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL(encode_signal_kernel, dim3(n_frames), dim3(SIG_THREADS), lds, st, aud_params, exp_params,
                       att_params, auds, exps, N, frame_ids, smo, out, keep);
    return hipGetLastError();
}
hipError_t launch_encode_signal_torso(const float* att_params, const float* poses, int pose_stride, int N,
                                      const int* frame_ids, int n_frames, int smo, float* out, hipStream_t st) {
    hipLaunchKernelGGL(encode_signal_torso_kernel, dim3(n_frames), dim3(SIG_THREADS), 0, st, att_params, poses,
                       pose_stride, N, frame_ids, smo, out);
    return hipGetLastError();
}

}  // namespace dfn

// ==========================================================================================================================
// Backward (training, one frame per step as upstream: MAIN:737-941).  Each kernel recomputes its forward into LDS, then
// back-propagates d(out) into the parameter gradients (+= into flat buffers laid out like the parameters).  The inputs
// (audio / expression features, poses) are data: no gradient is produced for them.
// ==========================================================================================================================
namespace dfn {

__device__ __forceinline__ float dleaky(float post) { return post > 0.f ? 1.0f : 0.02f; }   // slope > 0: sign(pre) = sign(post)

// gradients of y = act(b + W x) over the S rows:  GW[o][k] += sum_t dy[t][o] x[t][k];  Gb[o] += sum_t dy[t][o]
// (dy already multiplied by act'), and optionally dx[t][k] = sum_o W[o][k] dy[t][o].
// SET: the gradients are WRITTEN, not added (every element has exactly one writer per call): the caller's buffers need no
// zero fill (round 4: three 6-us fills sat in front of the audio chain's 190-us backward, which a training step can end on).
template <int K, bool SET = false, bool DW = true>
__device__ void linear_rows_bwd(const float* __restrict__ W, float* GW, float* Gb, int M, const float* x, const float* dy,
                                float* dx, int S) {
    constexpr int KU = (K + 63) / 64;
    constexpr int R = 1;                       // rows per iteration (more was slower, see linear_rows)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int o0 = wave * R; DW && o0 < M; o0 += nw * R) {      // (DW = false: the weight gradients come from signal_dw12_kernel)
        float old[R][KU];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float* g = GW + (long)(o0 + r) * K;
#pragma unroll
            for (int u = 0; u < KU; ++u) old[r][u] = (!SET && o0 + r < M && lane + 64 * u < K) ? g[lane + 64 * u] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int o = o0 + r;
            if (o >= M) break;
            float* g = GW + (long)o * K;
            float sb = 0.f;
#pragma unroll
            for (int t = 0; t < SIG_MAX_WIN; ++t) {
                if (t < S) {
                    const float d = dy[t * M + o];
                    sb += d;
#pragma unroll
                    for (int u = 0; u < KU; ++u)
                        if (lane + 64 * u < K) old[r][u] = fmaf(d, x[t * K + lane + 64 * u], old[r][u]);
                }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
                if (lane + 64 * u < K) g[lane + 64 * u] = old[r][u];
            if (lane == 0) Gb[o] = SET ? sb : Gb[o] + sb;
        }
    }
    if (dx) {
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float a[SIG_MAX_WIN];
#pragma unroll
            for (int t = 0; t < SIG_MAX_WIN; ++t) a[t] = 0.f;
            for (int o0 = 0; o0 < M; o0 += 8) {
                float w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = W[(long)(o0 + q) * K + k];           // M is a multiple of 8
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int t = 0; t < SIG_MAX_WIN; ++t)
                        if (t < S) a[t] = fmaf(w[q], dy[t * M + o0 + q], a[t]);
            }
#pragma unroll
            for (int t = 0; t < SIG_MAX_WIN; ++t)
                if (t < S) dx[t * K + k] = a[t];
        }
    }
    __syncthreads();
}

// Forward of AudioAttNet keeping every activation: acts[l] = output of conv l (post-LeakyReLU), [S][chans[l+1]];
// returns att[S] in LDS.  Then the backward: d_out[D] -> G (+=, layout of P) and d_feat[S][D] (+= the direct path).
struct AttWork {
    float* act[5];      // LDS: [S][16], [S][8], [S][4], [S][2], [S][1]
    float* dact[2];     // LDS ping-pong for the gradients, [S][D] each (largest layer input)
    float* att;         // [S]
    float* dz;          // [S]
};
__device__ void attention_fwd_keep(const float* __restrict__ P, int D, int S, const float* feat, const AttWork& w) {
    const int chans[6] = {D, 16, 8, 4, 2, 1};
    const float* cur = feat;
    int cs = D;
    long off = 0;
    for (int l = 0; l < 5; ++l) {
        const int ci = chans[l], co = chans[l + 1];
        const float* W = P + off;
        const float* B = W + (long)co * ci * 3;
        for (int e = threadIdx.x; e < co * S; e += blockDim.x) {
            const int o = e / S, t = e - o * S;
            float a = B[o];
            for (int c = 0; c < ci; ++c) {
                const float* wv = W + ((long)o * ci + c) * 3;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int tt = t + j - 1;
                    if (tt >= 0 && tt < S) a = fmaf(wv[j], cur[tt * cs + c], a);
                }
            }
            w.act[l][t * co + o] = leaky(a);
        }
        __syncthreads();
        off += (long)co * ci * 3 + co;
        cur = w.act[l];
        cs = co;
    }
    const float* LW = P + off;
    const float* LB = LW + S * S;
    if (threadIdx.x == 0) {
        float z[SIG_MAX_WIN], m = -3.0e38f;
        for (int i = 0; i < S; ++i) {
            float a = LB[i];
            for (int j = 0; j < S; ++j) a = fmaf(LW[i * S + j], cur[j], a);
            z[i] = a;
            m = fmaxf(m, a);
        }
        float sum = 0.f;
        for (int i = 0; i < S; ++i) { z[i] = expf(z[i] - m); sum += z[i]; }
        for (int i = 0; i < S; ++i) w.att[i] = z[i] / sum;
    }
    __syncthreads();
}
template <bool SET = false>
__device__ void attention_bwd(const float* __restrict__ P, float* G, int D, int S, const float* feat, const float* d_out,
                              const AttWork& w, float* d_feat /* [S][D], overwritten */) {
    const int chans[6] = {D, 16, 8, 4, 2, 1};
    long offs[6];
    offs[0] = 0;
    for (int l = 0; l < 5; ++l) offs[l + 1] = offs[l] + (long)chans[l + 1] * chans[l] * 3 + chans[l + 1];
    // out[d] = sum_t att[t] feat[t][d]
    if (threadIdx.x < S) {
        float a = 0.f;
        for (int d = 0; d < D; ++d) a = fmaf(d_out[d], feat[threadIdx.x * D + d], a);
        w.dz[threadIdx.x] = a;                       // d att[t] for now
    }
    for (int e = threadIdx.x; e < S * D; e += blockDim.x) d_feat[e] = w.att[e / D] * d_out[e % D];
    __syncthreads();
    // softmax, then Linear(S,S): z = LW c5 + LB
    const float* LW = P + offs[5];
    float* GLW = G + offs[5];
    float* GLB = GLW + S * S;
    const float* c5 = w.act[4];                      // [S][1]
    float* dcur = w.dact[0];                         // gradient w.r.t. the current layer's OUTPUT (post-activation)
    if (threadIdx.x == 0) {
        float dot = 0.f, dz[SIG_MAX_WIN];
        for (int j = 0; j < S; ++j) dot = fmaf(w.att[j], w.dz[j], dot);
        for (int i = 0; i < S; ++i) dz[i] = w.att[i] * (w.dz[i] - dot);
        for (int i = 0; i < S; ++i) {
            GLB[i] = SET ? dz[i] : GLB[i] + dz[i];
            for (int j = 0; j < S; ++j) GLW[i * S + j] = SET ? dz[i] * c5[j] : GLW[i * S + j] + dz[i] * c5[j];
        }
        for (int j = 0; j < S; ++j) {
            float a = 0.f;
            for (int i = 0; i < S; ++i) a = fmaf(LW[i * S + j], dz[i], a);
            dcur[j] = a;                             // d c5[j]  ([S][1])
        }
    }
    __syncthreads();
    // conv stack backwards
    for (int l = 4; l >= 0; --l) {
        const int ci = chans[l], co = chans[l + 1];
        const float* W = P + offs[l];
        float* GW = G + offs[l];
        float* GB = GW + (long)co * ci * 3;
        const float* x = l == 0 ? feat : w.act[l - 1];
        const int xs = l == 0 ? D : ci;
        const float* y = w.act[l];
        // d pre = d y * leaky'(y)   (in place)
        for (int e = threadIdx.x; e < S * co; e += blockDim.x) dcur[e] *= dleaky(y[e]);
        __syncthreads();
        for (int e = threadIdx.x; e < co * ci * 3; e += blockDim.x) {
            const int o = e / (ci * 3), r = e - o * ci * 3, c = r / 3, j = r - 3 * c;
            float a = 0.f;
            for (int t = 0; t < S; ++t) {
                const int tt = t + j - 1;
                if (tt >= 0 && tt < S) a = fmaf(dcur[t * co + o], x[tt * xs + c], a);
            }
            GW[e] = SET ? a : GW[e] + a;
        }
        for (int o = threadIdx.x; o < co; o += blockDim.x) {
            float a = 0.f;
            for (int t = 0; t < S; ++t) a += dcur[t * co + o];
            GB[o] = SET ? a : GB[o] + a;
        }
        float* dnext = (l == 0) ? nullptr : ((dcur == w.dact[0]) ? w.dact[1] : w.dact[0]);
        for (int e = threadIdx.x; e < S * ci; e += blockDim.x) {
            const int tt = e / ci, c = e - tt * ci;
            float a = 0.f;
            for (int o = 0; o < co; ++o) {
                const float* wv = W + ((long)o * ci + c) * 3;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = tt - j + 1;
                    if (t >= 0 && t < S) a = fmaf(wv[j], dcur[t * co + o], a);
                }
            }
            if (l == 0) d_feat[tt * D + c] += a;     // the conv stack reads feat[..., :D]
            else dnext[e] = a;
        }
        __syncthreads();
        if (l > 0) dcur = dnext;
    }
}

template <bool SET, bool KEPT>
__global__ __launch_bounds__(SIG_THREADS) void encode_signal_bwd_kernel(
    const float* __restrict__ PA, const float* __restrict__ PE, const float* __restrict__ PT, const float* __restrict__ auds,
    const float* __restrict__ exps, int N, int f, int smo, const float* __restrict__ d_out, float* GA, float* GE, float* GT,
    const float* __restrict__ kept) {
    extern __shared__ float lds[];
    const int S = smo > 0 ? smo : 1, half = smo / 2;
    float* xa = lds;                       // [S][512]
    float* h1 = xa + SIG_MAX_WIN * 512;    // [S][256]
    float* h2 = h1 + SIG_MAX_WIN * 256;    // [S][128]
    float* xe = h2 + SIG_MAX_WIN * 128;    // [S][64]
    float* e1 = xe + SIG_MAX_WIN * 64;     // [S][32]
    float* ft = e1 + SIG_MAX_WIN * 32;     // [S][96]
    float* a64 = ft + SIG_MAX_WIN * 96;    // [S][64]
    float* e32 = a64 + SIG_MAX_WIN * 64;   // [S][32]
    float* dft = e32 + SIG_MAX_WIN * 32;   // [S][96]
    float* g0 = dft + SIG_MAX_WIN * 96;    // [S][256] gradient scratch
    float* g1 = g0 + SIG_MAX_WIN * 256;    // [S][256]
    float* aw = g1 + SIG_MAX_WIN * 256;    // attention work: 5 activations + att + dz
    float* ps = aw + SIG_MAX_WIN * 33;     // attention parameters
    AttWork w;
    w.act[0] = aw; w.act[1] = w.act[0] + SIG_MAX_WIN * 16; w.act[2] = w.act[1] + SIG_MAX_WIN * 8;
    w.act[3] = w.act[2] + SIG_MAX_WIN * 4; w.act[4] = w.act[3] + SIG_MAX_WIN * 2;
    w.att = w.act[4] + SIG_MAX_WIN; w.dz = w.att + SIG_MAX_WIN;
    w.dact[0] = g0; w.dact[1] = g1;
    __shared__ float dout_s[96];
    for (int e = threadIdx.x; e < S * 512; e += blockDim.x) {
        const int t = e >> 9, k = e & 511, src = smo > 0 ? f - half + t : f;
        xa[t * 512 + k] = (smo == 0 || (src >= 0 && src < N)) ? auds[(long)src * 512 + k] : 0.f;
    }
    for (int e = threadIdx.x; e < S * 64; e += blockDim.x) {
        const int t = e >> 6, k = e & 63, src = smo > 0 ? f - half + t : f;
        xe[t * 64 + k] = (smo == 0 || (src >= 0 && src < N)) ? exps[(long)src * 64 + k] : 0.f;
    }
    for (int d = threadIdx.x; d < 96; d += blockDim.x) dout_s[d] = d_out[d];
    __syncthreads();
    if constexpr (KEPT) {
        // ---- the forward's activations as dfn_encode_signal_keep left them (same arithmetic, same bits) ----
        for (int e = threadIdx.x; e < S * 256; e += blockDim.x) h1[e] = kept[e];
        for (int e = threadIdx.x; e < S * 128; e += blockDim.x) h2[e] = kept[SIG_MAX_WIN * 256 + e];
        for (int e = threadIdx.x; e < S * 32; e += blockDim.x) e1[e] = kept[SIG_MAX_WIN * 384 + e];
        for (int e = threadIdx.x; e < S * 96; e += blockDim.x) ft[e] = kept[SIG_MAX_WIN * 416 + e];
        __syncthreads();
    } else {
        // ---- forward, every activation kept ----
        linear_rows<512>(PA, PA + 131072, 256, xa, h1, S, true);
        linear_rows<256>(PA + 131328, PA + 131328 + 32768, 128, h1, h2, S, true);
        linear_rows<128>(PA + 164224, PA + 164224 + 8192, 64, h2, a64, S, false);
        linear_rows<64>(PE, PE + 2048, 32, xe, e1, S, true);
        linear_rows<32>(PE + 2080, PE + 2080 + 1024, 32, e1, e32, S, false);
        for (int e = threadIdx.x; e < S * 96; e += blockDim.x) {
            const int t = e / 96, d = e - t * 96;
            ft[t * 96 + d] = d < 64 ? a64[t * 64 + d] : e32[t * 32 + d - 64];
        }
        __syncthreads();
    }
    // ---- backward ----
    if (smo > 0) {
        const float* Pl = stage_att(PT, ps, 96, S);
        attention_fwd_keep(Pl, 96, S, ft, w);
        attention_bwd<SET>(Pl, GT, 96, S, ft, dout_s, w, dft);
    } else {
        for (int d = threadIdx.x; d < 96; d += blockDim.x) dft[d] = dout_s[d];
        __syncthreads();
    }
    // split d feat -> d a64 (g0 as [S][64]) and d e32 (g1 as [S][32])
    for (int e = threadIdx.x; e < S * 96; e += blockDim.x) {
        const int t = e / 96, d = e - t * 96;
        if (d < 64) g0[t * 64 + d] = dft[e];
        else g1[t * 32 + d - 64] = dft[e];
    }
    __syncthreads();
    // ExpressionEnc: e32 = W2 e1 + b2 ; e1 = leaky(W1 xe + b1)      (g1 = d e32; dft reused as d e1)
    linear_rows_bwd<32, SET>(PE + 2080, GE + 2080, GE + 2080 + 1024, 32, e1, g1, dft, S);
    for (int e = threadIdx.x; e < S * 32; e += blockDim.x) dft[e] *= dleaky(e1[e]);
    __syncthreads();
    linear_rows_bwd<64, SET>(PE, GE, GE + 2048, 32, xe, dft, nullptr, S);
    // AudioNet_W2L: a64 = W3 h2 + b3 ; h2 = leaky(W2 h1 + b2) ; h1 = leaky(W1 xa + b1)      (g0 = d a64)
    linear_rows_bwd<128, SET>(PA + 164224, GA + 164224, GA + 164224 + 8192, 64, h2, g0, g1, S);       // g1 = d h2
    for (int e = threadIdx.x; e < S * 128; e += blockDim.x) g1[e] *= dleaky(h2[e]);
    __syncthreads();
    if constexpr (KEPT) {
        // The two big weight gradients (W1: 256 x 512, W2: 128 x 256 - 24 of this kernel's ~60 dependent memory round trips, each
        // 3 us next to the step's big kernels) are left to signal_dw12_kernel, 24 workgroups right behind this launch: here only
        // d h1 = W2^T d h2, and both pre-activation gradients go to the scratch tail of the keep buffer.
        float* dh = const_cast<float*>(kept) + SIG_KEEP_ACT;
        linear_rows_bwd<256, SET, false>(PA + 131328, nullptr, nullptr, 128, h1, g1, g0, S);           // g0 = d h1
        for (int e = threadIdx.x; e < S * 256; e += blockDim.x) dh[e] = g0[e] * dleaky(h1[e]);
        for (int e = threadIdx.x; e < S * 128; e += blockDim.x) dh[SIG_MAX_WIN * 256 + e] = g1[e];
    } else {
        linear_rows_bwd<256, SET>(PA + 131328, GA + 131328, GA + 131328 + 32768, 128, h1, g1, g0, S);     // g0 = d h1
        for (int e = threadIdx.x; e < S * 256; e += blockDim.x) g0[e] *= dleaky(h1[e]);
        __syncthreads();
        linear_rows_bwd<512, SET>(PA, GA, GA + 131072, 256, xa, g0, nullptr, S);
    }
}

// GW1 / Gb1 (256 x 512) and GW2 / Gb2 (128 x 256) of AudioNet_W2L from what the two kernels above left: 384 output rows over
// SIG_DW_BLOCKS x 4 waves, a row per wave and trip (a lane owns columns lane + 64 u), the sums over the S window rows in the
// same order as linear_rows_bwd - the same gradients bit for bit.
constexpr int SIG_DW_BLOCKS = 24;
template <bool SET>
__global__ __launch_bounds__(256) void signal_dw12_kernel(const float* __restrict__ auds, int N, int f, int smo,
                                                          const float* __restrict__ kept, float* GA) {
    __shared__ float xa[SIG_MAX_WIN * 512], h1[SIG_MAX_WIN * 256], d1[SIG_MAX_WIN * 256], d2[SIG_MAX_WIN * 128];
    const int S = smo > 0 ? smo : 1, half = smo / 2;
    for (int e = threadIdx.x; e < S * 512; e += blockDim.x) {
        const int t = e >> 9, k = e & 511, src = smo > 0 ? f - half + t : f;
        xa[t * 512 + k] = (smo == 0 || (src >= 0 && src < N)) ? auds[(long)src * 512 + k] : 0.f;
    }
    for (int e = threadIdx.x; e < S * 256; e += blockDim.x) { h1[e] = kept[e]; d1[e] = kept[SIG_KEEP_ACT + e]; }
    for (int e = threadIdx.x; e < S * 128; e += blockDim.x) d2[e] = kept[SIG_KEEP_ACT + SIG_MAX_WIN * 256 + e];
    __syncthreads();
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int row = gw; row < 384; row += nw) {
        const bool first = row < 256;
        const int o = first ? row : row - 256, K = first ? 512 : 256, M = first ? 256 : 128;
        const float* x = first ? xa : h1;
        const float* dy = first ? d1 : d2;
        float* g = GA + (first ? 0 : 131328) + (long)o * K;
        float* gb = GA + (first ? 131072 : 131328 + 32768);
        float acc[8], sb = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = (!SET && lane + 64 * u < K) ? g[lane + 64 * u] : 0.f;
#pragma unroll
        for (int t = 0; t < SIG_MAX_WIN; ++t)
            if (t < S) {
                const float d = dy[t * M + o];
                sb += d;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (lane + 64 * u < K) acc[u] = fmaf(d, x[t * K + lane + 64 * u], acc[u]);
            }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (lane + 64 * u < K) g[lane + 64 * u] = acc[u];
        if (lane == 0) gb[o] = SET ? sb : gb[o] + sb;
    }
}

template <bool SET>
__global__ __launch_bounds__(SIG_THREADS) void encode_signal_torso_bwd_kernel(const float* __restrict__ PT,
                                                                              const float* __restrict__ poses, int pose_stride,
                                                                              int N, int f, int smo,
                                                                              const float* __restrict__ d_out, float* GT) {
    __shared__ float emb[SIG_MAX_WIN * 42], demb[SIG_MAX_WIN * 42], g0[SIG_MAX_WIN * 42], g1[SIG_MAX_WIN * 42];
    __shared__ float aw[SIG_MAX_WIN * 33], dout_s[42], ps[att_param_count(42, SIG_MAX_WIN)];
    if (smo <= 0) return;                      // before --nosmo_iters no parameter takes part in the torso signal
    const int S = smo, half = smo / 2;
    for (int e = threadIdx.x; e < S * 6; e += blockDim.x) {
        const int t = e / 6, c = e - t * 6, src = f - half + t;
        float v = 0.f;
        if (smo == 0 || (src >= 0 && src < N)) {
            const float* R = poses + (long)src * pose_stride;
            if (c == 0) v = atan2f(R[2 * 4 + 2], R[1 * 4 + 2]);
            else if (c == 1) v = asinf(-R[0 * 4 + 2]);
            else if (c == 2) v = atan2f(R[0 * 4 + 0], -R[0 * 4 + 1]);
            else v = R[(c - 3) * 4 + 3];
        }
        const int g = c / 3, a = c - 3 * g;
        float* o = emb + t * 42 + g * 21;
        o[a] = v;
        o[3 + a] = sinf(v);          o[6 + a] = cosf(v);
        o[9 + a] = sinf(v * 2.0f);   o[12 + a] = cosf(v * 2.0f);
        o[15 + a] = sinf(v * 4.0f);  o[18 + a] = cosf(v * 4.0f);
    }
    for (int d = threadIdx.x; d < 42; d += blockDim.x) dout_s[d] = d_out[d];
    __syncthreads();
    AttWork w;
    w.act[0] = aw; w.act[1] = w.act[0] + SIG_MAX_WIN * 16; w.act[2] = w.act[1] + SIG_MAX_WIN * 8;
    w.act[3] = w.act[2] + SIG_MAX_WIN * 4; w.act[4] = w.act[3] + SIG_MAX_WIN * 2;
    w.att = w.act[4] + SIG_MAX_WIN; w.dz = w.att + SIG_MAX_WIN;
    w.dact[0] = g0; w.dact[1] = g1;
    const float* Pl = stage_att(PT, ps, 42, S);
    attention_fwd_keep(Pl, 42, S, emb, w);
    attention_bwd<SET>(Pl, GT, 42, S, emb, dout_s, w, demb);
}

hipError_t launch_encode_signal_bwd(const float* aud_params, const float* exp_params, const float* att_params,
                                    const float* auds, const float* exps, int N, int frame, int smo, const float* d_out,
                                    float* g_aud, float* g_exp, float* g_att, bool set, const float* kept, hipStream_t st) {
    const size_t lds = sizeof(float) * (SIG_MAX_WIN * (512 + 256 + 128 + 64 + 32 + 96 + 64 + 32 + 96 + 256 + 256 + 33) + ATT_PARAMS_MAX);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)encode_signal_bwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)encode_signal_bwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)encode_signal_bwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)encode_signal_bwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done = true;
    }
#define SIG_BWD_GO(SET_, KEPT_)                                                                                              \
    hipLaunchKernelGGL((encode_signal_bwd_kernel<SET_, KEPT_>), dim3(1), dim3(SIG_THREADS), lds, st, aud_params, exp_params,   \
                       att_params, auds, exps, N, frame, smo, d_out, g_aud, g_exp, g_att, kept)
    if (set && kept) SIG_BWD_GO(true, true);
    else if (set) SIG_BWD_GO(true, false);
    else if (kept) SIG_BWD_GO(false, true);
    else SIG_BWD_GO(false, false);
#undef SIG_BWD_GO
    if (kept) {
        if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
        if (set) hipLaunchKernelGGL(signal_dw12_kernel<true>, dim3(SIG_DW_BLOCKS), dim3(256), 0, st, auds, N, frame, smo, kept, g_aud);
        else hipLaunchKernelGGL(signal_dw12_kernel<false>, dim3(SIG_DW_BLOCKS), dim3(256), 0, st, auds, N, frame, smo, kept, g_aud);
    }
    return hipGetLastError();
}
hipError_t launch_encode_signal_torso_bwd(const float* att_params, const float* poses, int pose_stride, int N, int frame,
                                          int smo, const float* d_out, float* g_att, bool set, hipStream_t st) {
    if (set)
        hipLaunchKernelGGL(encode_signal_torso_bwd_kernel<true>, dim3(1), dim3(SIG_THREADS), 0, st, att_params, poses, pose_stride, N,
                           frame, smo, d_out, g_att);
    else
        hipLaunchKernelGGL(encode_signal_torso_bwd_kernel<false>, dim3(1), dim3(SIG_THREADS), 0, st, att_params, poses, pose_stride, N,
                           frame, smo, d_out, g_att);
    return hipGetLastError();
}

}  // namespace dfn
