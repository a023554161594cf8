// Developer probe (GPU box): which SIMD of a compute unit does wave w of a 512-thread workgroup run on?
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o tools/hwid_probe.bin && tools/hwid_probe.bin
// Reads HW_REG_HW_ID (gfx9 layout: WAVE_ID [3:0], SIMD_ID [5:4], PIPE_ID [7:6], CU_ID [11:8], SH_ID [12], SE_ID [15:13]) in
// every wave of workgroups that hold a compute unit alone (96 KiB of LDS, like the training kernels): the question behind
// the de-phased hand-over (LABNOTES.md 7, round 5) - are waves w and w + 4 the two waves of one SIMD, or w and w + 1?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void k(unsigned* out) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = 0;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    const int blocks = 512;
    unsigned *d, h[blocks * 8];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 96 * 1024, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {}, pair4 = 0, pair1 = 0;
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
        int same4 = 1, same1 = 1;
        for (int w = 0; w < 4; ++w) same4 &= ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3);
        for (int w = 0; w < 8; w += 2) same1 &= ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 1] >> 4) & 3);
        pair4 += same4;
        pair1 += same1;
    }
    printf("workgroup 0: ");
    for (int w = 0; w < 8; ++w) printf("wave %d -> SIMD %u (wave slot %u, CU %u)  ", w, (h[w] >> 4) & 3, h[w] & 15, (h[w] >> 8) & 15);
    printf("\nSIMD histogram per wave index over %d workgroups:\n", blocks);
    for (int w = 0; w < 8; ++w) printf("  wave %d: SIMD0 %d  SIMD1 %d  SIMD2 %d  SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("workgroups whose waves (w, w + 4) share a SIMD for every w: %d of %d; whose waves (2j, 2j + 1) do: %d of %d\n", pair4, blocks,
           pair1, blocks);
    return 0;
}
