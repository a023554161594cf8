// Developer probe (GPU box): semantics of the gfx950 instructions an MX-fp4 (e2m1) activation recorder relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/fp4_probe.hip -o tools/fp4_probe.bin && tools/fp4_probe.bin
// (1) v_cvt_scalef32_pk_fp4_bf16: value = x / scale rounded to e2m1 (0 .5 1 1.5 2 3 4 6), two per byte, which nibble first, saturation
// (2) v_mfma_scale_f32_32x32x64_f8f6f4 with A = fp8 (cbsz 0) and B = fp4 (blgp 4): which nibble of a B lane's 16 bytes pairs
//     with which byte of an A lane's 32 bytes, and which lane's scale applies
// (3) ds_read_b64_tr_b4: source nibble of every result nibble
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__global__ void k_cvt(const float* x, unsigned* out, float scale) {
    const int l = threadIdx.x;
    unsigned o = 0;
    for (int k = 0; k < 4; ++k) {
        bf16x2 p = {(__bf16)x[l * 8 + 2 * k], (__bf16)x[l * 8 + 2 * k + 1]};
        switch (k) {
            case 0: o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, p, scale, 0); break;
            case 1: o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, p, scale, 1); break;
            case 2: o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, p, scale, 2); break;
            default: o = __builtin_amdgcn_cvt_scalef32_pk_fp4_bf16(o, p, scale, 3); break;
        }
    }
    out[l] = o;
}
__global__ void k_mfma(const int* a, const int* b, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    i32x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = a[l * 8 + i]; B[i] = b[l * 8 + i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0 /* A: fp8 e4m3 */, 4 /* B: fp4 e2m1 */, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void k_tr4(int stride, int pass, unsigned long long* out) {
    __shared__ unsigned char lds[4096];
    // nibble index q = 2 * byte + (high ? 1 : 0); its value in pass p = bits [4p, 4p + 4) of q
    for (int i = threadIdx.x; i < 4096; i += 64) {
        const int q0 = 2 * i, q1 = 2 * i + 1;
        lds[i] = (unsigned char)(((q0 >> (4 * pass)) & 15) | (((q1 >> (4 * pass)) & 15) << 4));
    }
    __syncthreads();
    const i32x2 v = __builtin_amdgcn_ds_read_tr4_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + threadIdx.x * stride));
    out[threadIdx.x] = (unsigned long long)(unsigned)v[0] | ((unsigned long long)(unsigned)v[1] << 32);
}
template <typename T> T* dev(const std::vector<T>& h) {
    T* d;
    hipMalloc(&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main() {
    // ---- (1)
    {
        std::vector<float> x = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f,  -1.f, 0.25f, 0.75f, 1.25f, 2.5f, 5.f, 7.f, 100.f};
        x.resize(64 * 8, 1.f);
        unsigned* o;
        hipMalloc(&o, 256);
        for (float scale : {1.0f, 2.0f, 0.5f}) {
            hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dev(x), o, scale);
            unsigned h[2];
            hipMemcpy(h, o, 8, hipMemcpyDeviceToHost);
            printf("cvt fp4 scale %4.2f: [0 .5 1 1.5 2 3 4 6] -> %08x   [-1 .25 .75 1.25 2.5 5 7 100] -> %08x   (e2m1 codes 0..7 = 0 .5 1 1.5 2 3 4 6; nibble order: value 0 in bits 3:0 if the first word reads 76543210)\n",
                   scale, h[0], h[1]);
        }
    }
    // ---- (2)
    {
        float* c;
        hipMalloc(&c, 32 * 32 * 4);
        std::vector<float> h(32 * 32);
        const int m0 = 5, n0 = 9;
        // scales: B one-hot nibble (value 1.0 = code 2), A = ones
        for (int kh1 = 0; kh1 < 2; ++kh1)
            for (int q : {0, 1, 15, 16, 31}) {
                std::vector<int> A(64 * 8, 0x38383838), B(64 * 8, 0), SA(64, 127), SB(64, 127);
                ((unsigned char*)&B[(n0 + 32 * kh1) * 8])[q >> 1] = (unsigned char)(2 << (4 * (q & 1)));
                SB[n0] = 128;
                SB[n0 + 32] = 129;
                hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dev(A), dev(B), dev(SA), dev(SB), c);
                hipMemcpy(h.data(), c, 32 * 32 * 4, hipMemcpyDeviceToHost);
                printf("B one-hot (1.0) at lane (n0, kh %d) nibble %2d, A = ones: C[m0][n0] = %g  (2 = scale of lane kh 0, 4 = of lane kh 1, 0 = nibble not read)\n",
                       kh1, q, h[m0 * 32 + n0]);
            }
        // nibbles beyond 16 bytes must be ignored
        {
            std::vector<int> A(64 * 8, 0x38383838), B(64 * 8, 0), SA(64, 127), SB(64, 127);
            for (int w = 4; w < 8; ++w) B[n0 * 8 + w] = 0x22222222;
            hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dev(A), dev(B), dev(SA), dev(SB), c);
            hipMemcpy(h.data(), c, 32 * 32 * 4, hipMemcpyDeviceToHost);
            printf("B dwords 4..7 = ones, 0..3 = 0: C[m0][n0] = %g (0 = the upper 16 bytes are ignored)\n", h[m0 * 32 + n0]);
        }
        // pairing: A one-hot (m0, kh0, byte j0) x B one-hot (n0, kh1, nibble q)
        for (int kh0 = 0; kh0 < 2; ++kh0)
            for (int j0 : {0, 1, 2, 7, 8, 15, 16, 17, 31}) {
                int found = 0;
                for (int kh1 = 0; kh1 < 2; ++kh1)
                    for (int q = 0; q < 32; ++q) {
                        std::vector<int> A(64 * 8, 0), B(64 * 8, 0), SA(64, 127), SB(64, 127);
                        ((unsigned char*)&A[(m0 + 32 * kh0) * 8])[j0] = 0x38;
                        ((unsigned char*)&B[(n0 + 32 * kh1) * 8])[q >> 1] = (unsigned char)(2 << (4 * (q & 1)));
                        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dev(A), dev(B), dev(SA), dev(SB), c);
                        hipMemcpy(h.data(), c, 32 * 32 * 4, hipMemcpyDeviceToHost);
                        if (h[m0 * 32 + n0] != 0.f) { printf("pairing: A lane kh %d byte %2d  <->  B lane kh %d nibble %2d  (C = %g)\n", kh0, j0, kh1, q, h[m0 * 32 + n0]); ++found; }
                    }
                if (!found) printf("pairing: A lane kh %d byte %2d  <->  nothing\n", kh0, j0);
            }
    }
    // ---- (3)
    {
        unsigned long long *d, r[4][64];
        hipMalloc(&d, 512);
        for (int stride : {8, 16}) {
            for (int p = 0; p < 4; ++p) {
                hipLaunchKernelGGL(k_tr4, dim3(1), dim3(64), 0, 0, stride, p, d);
                hipMemcpy(r[p], d, 512, hipMemcpyDeviceToHost);
            }
            printf("ds_read_b64_tr_b4: lane l names byte address l * %d; source NIBBLE index (2 * byte + high) of its 16 result nibbles:\n", stride);
            for (int l = 0; l < 64; ++l) {
                if (l >= 18 && l < 32) continue;
                if (l >= 34) continue;
                printf("  lane %2d:", l);
                for (int j = 0; j < 16; ++j) {
                    int q = 0;
                    for (int p = 0; p < 4; ++p) q |= (int)((r[p][l] >> (4 * j)) & 15) << (4 * p);
                    printf(" %4d", q);
                }
                printf("\n");
            }
        }
    }
    return 0;
}
