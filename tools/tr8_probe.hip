// Developer probe (GPU box): semantics of ds_read_b64_tr_b8 (gfx950 LDS transpose read, 8-bit elements).
//   hipcc --offload-arch=gfx950 -O2 tools/tr8_probe.hip -o tools/tr8_probe.bin && tools/tr8_probe.bin
// LDS holds byte i = i & 255 in one pass and i >> 8 in a second: each lane names an address, and the 8 bytes it receives
// identify their source addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void k(int stride, int hi, unsigned long long* out) {
    __shared__ unsigned char lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = hi ? (unsigned char)(i >> 8) : (unsigned char)(i & 255);
    __syncthreads();
    const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + threadIdx.x * stride));
    out[threadIdx.x] = (unsigned long long)(unsigned)v[0] | ((unsigned long long)(unsigned)v[1] << 32);
}
int main() {
    unsigned long long *d, lo[64], hi[64];
    hipMalloc(&d, 512);
    for (int stride : {8, 16, 32}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, stride, 0, d);
        hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, stride, 1, d);
        hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
        printf("lane l names address l * %d; source addresses of its 8 result bytes:\n", stride);
        for (int l = 0; l < 64; ++l) {
            if (l >= 20 && l < 32) continue;
            if (l >= 36) continue;
            printf("  lane %2d:", l);
            for (int j = 0; j < 8; ++j) printf(" %4d", (int)(((hi[l] >> (8 * j)) & 255) << 8 | ((lo[l] >> (8 * j)) & 255)));
            printf("\n");
        }
    }
    return 0;
}
