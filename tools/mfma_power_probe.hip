// Developer probe (GPU box): what bounds a register-resident MFMA chain on MI355X - the matrix pipe, or the power budget?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/mfma_power_probe.bin && tools/mfma_power_probe.bin
//
// "a clock-vs-(LDS bytes per MFMA, VALU per MFMA) microbenchmark that predicts what pipe-busy / clock pair
// 0.65 would need".  The kernel is render_kernel's inner loop reduced to its cost drivers: every wave (8 per workgroup, one
// workgroup per compute unit and round, like the renderer) issues v_mfma_f32_32x32x16_{f16,bf16} on two alternating
// accumulator sets with
//     LDS  = how many 1-KiB A fragments it reads from LDS per MFMA (the renderer: 1 - every wave re-reads the shared weight
//            fragment; 0 = operands stay in registers; 0.5 = a fragment feeds two MFMAs, what a 64-point wave would do),
//     VALU = how many vector ALU instructions (v_cvt_pk + v_pk_max, the convert / ReLU epilogue's mix) per MFMA (the
//            renderer: 2.15),
//     DATA = random operands (toggling bits: what an MLP sees) or zeros (the switching-energy floor).
// Per configuration: TFLOP/s, the shader clock under load (s_memtime cycles / s_memrealtime 100-MHz ticks, as the renderer's
// own probe reads it), the matrix-pipe duty = TFLOP/s / (2500 x clock / 2.4 GHz), and
// frac = TFLOP/s / 2500.  Reading the table: if the chain with LDS = 0, VALU = 0 already clocks well below 2.4 GHz the MFMAs
// themselves hit the power limit; the rows LDS = 1 / VALU = 2 give the operating point of the renderer, and the gap between
// the two is what ANY rescheduling of fragment traffic and epilogue can return.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int WAVES = 8, FRAGS = 32;            // a 32-KiB slab of fragments in LDS, re-read round robin

// LDS2: fragment reads per TWO MFMAs (0, 1, 2); VALU2: vector instructions per two MFMAs
// MODE (round 3, second series - why do LDS reads and VALU cost more TOGETHER than the sum of each alone?):
//   0 as above; 1 the LDS reads land in registers the MFMAs do NOT use (operands stay constant); 2 the VALU are plain v_mov_b32;
//   3 the fragment reads of an iteration are issued as one burst in front of its 32 MFMAs; 4 VALU = v_pk_max_i16 only
template <bool F16, int LDS2, int VALU2, int MODE = 0>
__global__ __launch_bounds__(64 * WAVES) void chain(const u32x4* __restrict__ frag_src, const u32x4* __restrict__ b_src,
                                                       int iters, float* out, unsigned long long* clk) {
    extern __shared__ u32x4 slab[];          // 150 KiB requested at launch: ONE workgroup per compute unit, 2 waves per SIMD, like the renderer
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < FRAGS * 64; i += 64 * WAVES) slab[i] = frag_src[i];
    __syncthreads();
    u32x4 b[4];
    for (int k = 0; k < 4; ++k) b[k] = b_src[(threadIdx.x * 4 + k) & 4095];
    f32x16 acc0 = {}, acc1 = {};
    u32x4 a0 = slab[lane], a1 = slab[64 + lane];
    unsigned e0 = b[0][0], e1 = b[1][1], e2 = b[2][2], e3 = b[3][3];
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    u32x4 d0 = a0, d1 = a1;
    u32x4 burst[16];
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 16; ++q) burst[q] = slab[((q + it) % FRAGS) * 64 + lane];
        }
        if constexpr (MODE == 5 || MODE == 6) {          // the iteration's VALU as ONE burst (5: in front of its 32 MFMAs; 6: two bursts of half)
#pragma unroll
            for (int v = 0; v < 16 * VALU2 / (MODE == 6 ? 2 : 1); ++v) {
                unsigned& x = (v & 3) == 0 ? e0 : (v & 3) == 1 ? e1 : (v & 3) == 2 ? e2 : e3;
                if (v & 1) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(e0 ^ e2));
                else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(__builtin_bit_cast(float, x)), "v"(__builtin_bit_cast(float, x ^ 0x3f800000u)));
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {          // 32 MFMAs per iteration
            if constexpr (MODE == 6) {
                if (k == 8) {
#pragma unroll
                    for (int v = 0; v < 8 * VALU2; ++v) {
                        unsigned& x = (v & 3) == 0 ? e0 : (v & 3) == 1 ? e1 : (v & 3) == 2 ? e2 : e3;
                        if (v & 1) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(e0 ^ e2));
                        else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(__builtin_bit_cast(float, x)), "v"(__builtin_bit_cast(float, x ^ 0x3f800000u)));
                    }
                }
            }
            const int f = (2 * k) % FRAGS;
            if constexpr (MODE == 1) {
                if (LDS2 >= 1) { d0 = slab[f * 64 + lane]; asm volatile("" : "+v"(d0)); }
                if (LDS2 >= 2) { d1 = slab[(f + 1) * 64 + lane]; asm volatile("" : "+v"(d1)); }
            } else if constexpr (MODE == 3) {
                a0 = burst[(2 * k) % 16];
                a1 = burst[(2 * k + 1) % 16];
            } else {
                if (LDS2 >= 1) a0 = slab[f * 64 + lane];
                if (LDS2 >= 2) a1 = slab[(f + 1) * 64 + lane];
            }
            if constexpr (F16) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b[k & 3]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b[(k + 1) & 3]), acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b[k & 3]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b[(k + 1) & 3]), acc1, 0, 0, 0);
            }
            // the epilogue's instruction mix on registers the MFMAs do not wait for: packed convert + packed max
#pragma unroll
            for (int v = 0; v < (MODE == 5 || MODE == 6 ? 0 : VALU2); ++v) {
                unsigned& x = (v & 3) == 0 ? e0 : (v & 3) == 1 ? e1 : (v & 3) == 2 ? e2 : e3;
                if constexpr (MODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(x ^ (unsigned)k));
                else if ((v & 1) || MODE == 4) asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(x) : "v"(x), "v"(e0 ^ e2));
                else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(__builtin_bit_cast(float, x)), "v"(__builtin_bit_cast(float, x ^ 0x3f800000u)));
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(e0 ^ e1 ^ e2 ^ e3 ^ d0[0] ^ d1[1]);
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <bool F16, int LDS2, int VALU2, int MODE = 0>
void run(const char* label, const u32x4* frag, const u32x4* bsrc, float* out, unsigned long long* clk, int cus, const char* data) {
    const int iters = 4000, blocks = cus * 4;            // 4 rounds of one workgroup per compute unit
    const int lds = 150 * 1024;
    hipFuncSetAttribute((const void*)chain<F16, LDS2, VALU2, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<F16, LDS2, VALU2, MODE>), dim3(blocks), dim3(64 * WAVES), lds, 0, frag, bsrc, 200, out, clk);     // warm-up
    hipDeviceSynchronize();
    float best = 1e30f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((chain<F16, LDS2, VALU2, MODE>), dim3(blocks), dim3(64 * WAVES), lds, 0, frag, bsrc, iters, out, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
    }
    const double mfma = (double)blocks * WAVES * iters * 32.0;           // wave-level instructions
    const double tf = mfma * 32768.0 / (best * 1e-3) / 1e12;
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    // matrix-pipe duty from the measured rate and the measured clock: at duty 1 and 2.4 GHz the chip does 2500 TFLOP/s
    const double duty = ghz > 0 ? tf / (2500.0 * ghz / 2.4) : 0.0;
    printf("%-5s %-6s LDS %.1f KiB/MFMA  VALU %.1f/MFMA mode %d : %7.1f TFLOP/s  frac %.3f  clock %.2f GHz  matrix-pipe duty %.2f\n", label, data,
           LDS2 / 2.0, VALU2 / 2.0, MODE, tf, tf / 2500.0, ghz, duty);
}

int main() {
    int dev = 0, cus = 256;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    std::vector<unsigned> h(FRAGS * 64 * 4), hb(4096 * 4);
    unsigned* d_frag;
    unsigned* d_b;
    float* out;
    unsigned long long* clk;
    hipMalloc(&d_frag, h.size() * 4);
    hipMalloc(&d_b, hb.size() * 4);
    hipMalloc(&out, (size_t)cus * 4 * 64 * WAVES * 4);
    hipMalloc(&clk, 16);
    for (int pass = 0; pass < 2; ++pass)
        for (int f16 = 1; f16 >= 0; --f16) {
            // random: values of magnitude 2^-3 .. 2^0 with random mantissas and signs in the operand type's own encoding
            // (f16: exponent field 0x30-0x3b.., bf16: 0x3e00-0x3f7f); zeros: the switching-energy floor
            unsigned seed = 12345u;
            auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
            auto half = [&]() -> unsigned {
                if (pass) return 0u;
                const unsigned sign = (rnd() & 1u) << 15;
                return sign | (f16 ? 0x3000u + (rnd() % 0x0c00u) : 0x3e00u + (rnd() % 0x0180u));
            };
            for (auto& w : h) w = half() | (half() << 16);
            for (auto& w : hb) w = half() | (half() << 16);
            hipMemcpy(d_frag, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(d_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
            const char* data = pass ? "zeros" : "random";
            const u32x4* F = (const u32x4*)d_frag;
            const u32x4* B = (const u32x4*)d_b;
#define RUN(T, L, V) run<T, L, V>(T ? "f16" : "bf16", F, B, out, clk, cus, data)
            if (f16) {
                RUN(true, 0, 0); RUN(true, 1, 0); RUN(true, 2, 0);
                RUN(true, 0, 4); RUN(true, 2, 2); RUN(true, 2, 4); RUN(true, 2, 6); RUN(true, 1, 4);
                if (getenv("PROBE_SERIES2")) {
#define RUNM(L, V, M) run<true, L, V, M>("f16", F, B, out, clk, cus, data)
                    RUNM(2, 4, 2); RUNM(0, 4, 2); RUNM(2, 4, 4); RUNM(0, 4, 4);
                    RUNM(2, 2, 5); RUNM(2, 4, 5); RUNM(2, 6, 5); RUNM(0, 4, 5); RUNM(2, 4, 6); RUNM(2, 2, 0); RUNM(2, 4, 0);
#undef RUNM
                }
            } else {
                RUN(false, 0, 0); RUN(false, 2, 0); RUN(false, 2, 4); RUN(false, 1, 4);
            }
#undef RUN
        }
    return 0;
}
