// Developer probe (GPU box): does the CLAMP bit of the packed f32 -> f16 / bf16 conversions saturate the RESULT to [0, 1] on gfx950?
// (If it does, conversion + ReLU of activations pre-scaled into [0, 1] is ONE instruction per two values instead of two:
// LABNOTES.md 8 "Open (0)", tools/power_mix_sweep.py.)   hipcc --offload-arch=gfx950 tools/clamp_cvt_probe.hip -o tools/clamp_cvt_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* a, unsigned* o) {
    const float x = a[2 * threadIdx.x], y = a[2 * threadIdx.x + 1];
    unsigned r, r2;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(x), "v"(y));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r2) : "v"(x), "v"(y));
    o[2 * threadIdx.x] = r;
    o[2 * threadIdx.x + 1] = r2;
}
static float h2f(unsigned short h) {
    const unsigned s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = m * 5.9604644775390625e-08f;
    else if (e == 31) v = m ? __builtin_nanf("") : __builtin_inff();
    else { unsigned u = ((e + 112) << 23) | (m << 13); memcpy(&v, &u, 4); }
    return s ? -v : v;
}
static float b2f(unsigned short b) { unsigned u = (unsigned)b << 16; float v; memcpy(&v, &u, 4); return v; }
int main() {
    const float in[16] = {-2.f, -0.5f, -0.f, 0.f, 1e-7f, 0.3f, 0.999f, 1.f, 1.0005f, 1.7f, 100.f, 70000.f, -1e-8f, 0.5f, 6.1e-5f, 3e-5f};
    float* d; unsigned* o; unsigned ho[16];
    hipMalloc(&d, sizeof in); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, in, sizeof in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int t = 0; t < 8; ++t)
        printf("in (%g, %g): f16 clamp -> (%g, %g)   bf16 clamp -> (%g, %g)\n", in[2 * t], in[2 * t + 1], h2f(ho[2 * t] & 0xffff),
               h2f(ho[2 * t] >> 16), b2f(ho[2 * t + 1] & 0xffff), b2f(ho[2 * t + 1] >> 16));
    return 0;
}
