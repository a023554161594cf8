// Developer probe (GPU box): semantics of the gfx950 MX instructions the fp8 recorder relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/mx_probe.hip -o tools/mx_probe.bin && tools/mx_probe.bin
// (1) v_cvt_scalef32_pk_fp8_bf16 / _f32: does `scale` divide?  (2) v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 operands:
// lane (m = l & 31, kh = l >> 5) byte j of A pairs with lane (n, kh) byte j of B; each lane's scale byte (E8M0, op_sel 0 =
// byte 0 of the scale register) applies to ITS 32 values; C/D in the standard 32x32 map.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__global__ void k_mfma(const int* a, const int* b, const int* sa, const int* sb, float* c, int opsel) {
    const int l = threadIdx.x;
    i32x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = a[l * 8 + i]; B[i] = b[l * 8 + i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (opsel == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 0, sa[l], 0, sb[l]);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0, 0, 1, sa[l], 2, sb[l]);
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void k_cvt(const float* x, int* out_f32, int* out_bf16, float scale) {
    const int l = threadIdx.x;
    s16x2 o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, x[l * 4], x[l * 4 + 1], scale, false);
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, x[l * 4 + 2], x[l * 4 + 3], scale, true);
    out_f32[l] = __builtin_bit_cast(int, o);
    bf16x2 p = {(__bf16)x[l * 4], (__bf16)x[l * 4 + 1]}, q = {(__bf16)x[l * 4 + 2], (__bf16)x[l * 4 + 3]};
    s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, p, scale, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, q, scale, true);
    out_bf16[l] = __builtin_bit_cast(int, w);
}
static unsigned char enc(int v) {       // e4m3fn of small integers
    const unsigned char t[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return (unsigned char)((v < 0 ? 0x80 : 0) | t[abs(v)]);
}
template <typename T> T* dev(const std::vector<T>& h) {
    T* d;
    hipMalloc(&d, h.size() * sizeof(T));
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main() {
    // ---- (1) conversions
    std::vector<float> x = {1.f, 2.f, 0.5f, -4.f, 448.f, 1000.f, 0.001953125f, 3.f};
    x.resize(256, 1.f);
    int *of, *ob;
    hipMalloc(&of, 256);
    hipMalloc(&ob, 256);
    for (float scale : {1.0f, 4.0f, 0.25f}) {
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dev(x), of, ob, scale);
        int hf[2], hb[2];
        hipMemcpy(hf, of, 8, hipMemcpyDeviceToHost);
        hipMemcpy(hb, ob, 8, hipMemcpyDeviceToHost);
        printf("cvt scale %5.2f: f32 -> %08x %08x   bf16 -> %08x %08x   (x = 1 2 .5 -4 | 448 1000 2^-9 3; e4m3: 1=38 2=40 .5=30 4=48 .25=28 8=50 16=58)\n",
               scale, hf[0], hf[1], hb[0], hb[1]);
    }
    // ---- (2) the scaled MFMA
    for (int opsel = 0; opsel < 2; ++opsel) {
        std::vector<int> A(64 * 8), B(64 * 8), SA(64), SB(64);
        std::vector<double> ref(32 * 32, 0.0);
        auto av = [](int m, int kh, int j) { return ((m + 2 * j + 3 * kh) % 5) - 2; };
        auto bv = [](int n, int kh, int j) { return ((n * 3 + j + kh) % 7) - 3; };
        auto ea = [](int m, int kh) { return (m + kh) % 3 - 1; };
        auto eb = [](int n, int kh) { return (n + 2 * kh) % 2; };
        for (int l = 0; l < 64; ++l) {
            const int m = l & 31, kh = l >> 5;
            unsigned char* pa = (unsigned char*)&A[l * 8];
            unsigned char* pb = (unsigned char*)&B[l * 8];
            for (int j = 0; j < 32; ++j) { pa[j] = enc(av(m, kh, j)); pb[j] = enc(bv(m, kh, j)); }
            const int sa = 127 + ea(m, kh), sb = 127 + eb(m, kh);
            // op_sel 0: byte 0 carries the scale; the second launch puts it in byte 1 (A) / byte 2 (B) with junk elsewhere
            SA[l] = opsel ? (0x7f00007f & 0) | (sa << 8) | 0x00990011 : sa | 0x55443300;
            SB[l] = opsel ? (sb << 16) | 0x11000022 : sb | 0x66778800;
        }
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n)
                for (int kh = 0; kh < 2; ++kh)
                    for (int j = 0; j < 32; ++j)
                        ref[m * 32 + n] += av(m, kh, j) * std::ldexp(1.0, ea(m, kh)) * bv(n, kh, j) * std::ldexp(1.0, eb(n, kh));
        float* c;
        hipMalloc(&c, 32 * 32 * 4);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dev(A), dev(B), dev(SA), dev(SB), c, opsel);
        std::vector<float> h(32 * 32);
        hipMemcpy(h.data(), c, 32 * 32 * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int i = 0; i < 32 * 32; ++i) worst = std::fmax(worst, std::fabs(h[i] - ref[i]));
        printf("mfma_scale 32x32x64 fp8, scale byte via op_sel %d: max |C - model| = %g  (C[0][0] %g vs %g, C[5][9] %g vs %g) -> %s\n", opsel,
               worst, h[0], ref[0], h[5 * 32 + 9], ref[5 * 32 + 9], worst == 0 ? "MODEL HOLDS" : "MODEL WRONG");
    }
    // ---- (3) which lane's scale applies to which byte, and which A byte pairs with which B byte: one-hot probes
    {
        float* c;
        hipMalloc(&c, 32 * 32 * 4);
        std::vector<float> h(32 * 32);
        const int m0 = 5, n0 = 9;
        for (int kh0 = 0; kh0 < 2; ++kh0)
            for (int j0 : {0, 3, 4, 15, 16, 31}) {
                std::vector<int> A(64 * 8, 0), B(64 * 8, 0x38383838), SA(64, 127), SB(64, 127);
                ((unsigned char*)&A[(m0 + 32 * kh0) * 8])[j0] = 0x38;
                SA[m0] = 128;            // lane (m0, kh 0): x2
                SA[m0 + 32] = 129;       // lane (m0, kh 1): x4
                hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dev(A), dev(B), dev(SA), dev(SB), c, 0);
                hipMemcpy(h.data(), c, 32 * 32 * 4, hipMemcpyDeviceToHost);
                printf("A one-hot at lane (m0, kh %d) byte %2d, B = ones: C[m0][n0] = %g  (2 = scale of lane kh 0, 4 = of lane kh 1)\n", kh0, j0,
                       h[m0 * 32 + n0]);
            }
        // pairing: A one-hot (m0, kh0, j0) x B one-hot (n0, kh1, j1)
        int bad = 0;
        for (int kh0 = 0; kh0 < 2; ++kh0)
            for (int j0 = 0; j0 < 32; j0 += 5)
                for (int kh1 = 0; kh1 < 2; ++kh1)
                    for (int j1 = 0; j1 < 32; ++j1) {
                        std::vector<int> A(64 * 8, 0), B(64 * 8, 0), SA(64, 127), SB(64, 127);
                        ((unsigned char*)&A[(m0 + 32 * kh0) * 8])[j0] = 0x38;
                        ((unsigned char*)&B[(n0 + 32 * kh1) * 8])[j1] = 0x38;
                        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dev(A), dev(B), dev(SA), dev(SB), c, 0);
                        hipMemcpy(h.data(), c, 32 * 32 * 4, hipMemcpyDeviceToHost);
                        const bool hit = h[m0 * 32 + n0] != 0.f, want = kh0 == kh1 && j0 == j1;
                        if (hit != want) { if (bad < 10) printf("pairing: A(kh %d, j %d) x B(kh %d, j %d) -> %g\n", kh0, j0, kh1, j1, h[m0 * 32 + n0]); ++bad; }
                    }
        printf("pairing model (same kh, same byte): %d mismatches\n", bad);
    }
    return 0;
}
