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
constexpr int SIG_THREADS = 256;

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.02f * x; }

// y[t][o] = act(b[o] + sum_k W[o][k] x[t][k]),  x: LDS [S][K], y: LDS [S][M]
__device__ void linear_rows(const float* __restrict__ W, const float* __restrict__ b, int M, int K, const float* x,
                            float* y, int S, bool act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int o = wave; o < M; o += nw) {
        float acc[SIG_MAX_WIN];
#pragma unroll
        for (int t = 0; t < SIG_MAX_WIN; ++t) acc[t] = 0.f;
        const float* w = W + (long)o * K;
        for (int k = lane; k < K; k += 64) {
            const float wv = w[k];
#pragma unroll
            for (int t = 0; t < SIG_MAX_WIN; ++t)
                if (t < S) acc[t] = fmaf(wv, x[t * K + k], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < SIG_MAX_WIN; ++t) {
            if (t < S) {
                float a = acc[t];
                for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d);
                if (lane == 0) {
                    a += b[o];
                    y[t * M + o] = act ? leaky(a) : a;
                }
            }
        }
    }
    __syncthreads();
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
                                                                    const int* __restrict__ frame_ids, int smo, float* out) {
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
    const int f = frame_ids[blockIdx.x];
    for (int e = threadIdx.x; e < S * 512; e += blockDim.x) {
        const int t = e >> 9, k = e & 511, src = smo > 0 ? f - half + t : f;
        xa[t * 512 + k] = (src >= 0 && src < N) ? auds[(long)src * 512 + k] : 0.f;      // zero rows outside, MAIN:36-57
    }
    for (int e = threadIdx.x; e < S * 64; e += blockDim.x) {
        const int t = e >> 6, k = e & 63, src = smo > 0 ? f - half + t : f;
        xe[t * 64 + k] = (src >= 0 && src < N) ? exps[(long)src * 64 + k] : 0.f;
    }
    __syncthreads();
    // AudioNet_W2L: 512 -> 256 -> 128 -> 64
    linear_rows(PA, PA + 131072, 256, 512, xa, h1, S, true);
    linear_rows(PA + 131328, PA + 131328 + 32768, 128, 256, h1, h2, S, true);
    linear_rows(PA + 164224, PA + 164224 + 8192, 64, 128, h2, a64, S, false);
    // ExpressionEnc: 64 -> 32 -> 32
    linear_rows(PE, PE + 2048, 32, 64, xe, e1, S, true);
    linear_rows(PE + 2080, PE + 2080 + 1024, 32, 32, e1, e32, S, false);
    for (int e = threadIdx.x; e < S * 96; e += blockDim.x) {
        const int t = e / 96, d = e - t * 96;
        ft[t * 96 + d] = d < 64 ? a64[t * 64 + d] : e32[t * 32 + d - 64];
    }
    __syncthreads();
    float* o = out + (long)blockIdx.x * 96;
    if (smo > 0) attention(PT, 96, S, ft, b0, b1, o);
    else
        for (int d = threadIdx.x; d < 96; d += blockDim.x) o[d] = ft[d];
}

// ---- A8 ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SIG_THREADS) void encode_signal_torso_kernel(const float* __restrict__ PT,
                                                                          const float* __restrict__ poses, int pose_stride,
                                                                          int N, const int* __restrict__ frame_ids, int smo,
                                                                          float* out) {
    __shared__ float emb[SIG_MAX_WIN * 42], b0[SIG_MAX_WIN * 16], b1[SIG_MAX_WIN * 16];
    const int S = smo > 0 ? smo : 1, half = smo / 2;
    const int f = frame_ids[blockIdx.x];
    for (int e = threadIdx.x; e < S * 6; e += blockDim.x) {
        const int t = e / 6, c = e - t * 6, src = smo > 0 ? f - half + t : f;
        float v = 0.f;                                   // zero rows of (euler, trans) outside the sequence, MAIN:96-103
        if (src >= 0 && src < N) {
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
    if (smo > 0) attention(PT, 42, S, emb, b0, b1, o);
    else
        for (int d = threadIdx.x; d < 42; d += blockDim.x) o[d] = emb[d];
}

hipError_t launch_encode_signal(const float* aud_params, const float* exp_params, const float* att_params,
                                const float* auds, const float* exps, int N, const int* frame_ids, int n_frames, int smo,
                                float* out, hipStream_t st) {
    const size_t lds = sizeof(float) * SIG_MAX_WIN * (512 + 256 + 128 + 64 + 32 + 96 + 16 + 16 + 64 + 32);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)encode_signal_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL(encode_signal_kernel, dim3(n_frames), dim3(SIG_THREADS), lds, st, aud_params, exp_params,
                       att_params, auds, exps, N, frame_ids, smo, out);
    return hipGetLastError();
}
hipError_t launch_encode_signal_torso(const float* att_params, const float* poses, int pose_stride, int N,
                                      const int* frame_ids, int n_frames, int smo, float* out, hipStream_t st) {
    hipLaunchKernelGGL(encode_signal_torso_kernel, dim3(n_frames), dim3(SIG_THREADS), 0, st, att_params, poses,
                       pose_stride, N, frame_ids, smo, out);
    return hipGetLastError();
}

}  // namespace dfn
