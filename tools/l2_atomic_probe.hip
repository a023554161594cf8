// l2_atomic_probe.hip - what does it cost to accumulate weight-gradient partials in the XCD's own L2?
//
// Design question behind it (LABNOTES.md §7): a backward kernel that forms dW inside the dX chain has to flush a
// 256 x 256 f32 accumulator tile (256 KB) per 256 sample points and layer.  If every XCD owns one private copy of
// the gradient buffer (2.3 MB per field: L2-resident) and its workgroups add into it with atomics that are resolved
// in that L2, no partial ever goes to HBM.  This probe measures the rate of exactly that flush pattern:
//   f32  : global_atomic_add_f32, no return, workgroup scope (resolved in the local L2)
//   f32a : the same at agent scope (sc1: resolved memory-side) - what a single shared buffer would cost
//   u64  : global_atomic_add_x2 (64-bit integer, fixed point -> order-independent, i.e. bit-reproducible sums)
//   st   : plain stores of the same bytes (the flush without the read-modify-write), for scale
// and, beside it, the streaming read rate of HBM for the operand arrays (dwordx4 loads, grid sweep).
//
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/l2_atomic_probe tools/l2_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID[3:0]

// every workgroup (8 waves) flushes `rounds` tiles of 256 x 256 elements; wave w owns rows 32 w .. 32 w + 31, and one
// instruction covers two 128-byte row segments (lanes 0..31 one row, 32..63 another), like the MFMA C layout
template <int MODE>
__global__ __launch_bounds__(512) void flush_kernel(void* bufs, long buf_elems, int tiles_per_buf, int rounds, int private_xcd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = private_xcd ? xcc_id() : 0;
    for (int r = 0; r < rounds; ++r) {
        const int tile = (blockIdx.x / 8 + r) % tiles_per_buf;
        const long base = (long)x * buf_elems + (long)tile * 65536;
#pragma unroll 4
        for (int i = 0; i < 128; ++i) {
            const int reg = i & 15, ct = i >> 4;                       // 8 column tiles x 16 registers
            const int row = 32 * wave + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            const long e = base + (long)row * 256 + 32 * ct + (lane & 31);
            if (MODE == 0) {
                __hip_atomic_fetch_add((float*)bufs + e, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 1) {
                __hip_atomic_fetch_add((float*)bufs + e, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (MODE == 2) {
                __hip_atomic_fetch_add((unsigned long long*)bufs + e, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 3) {
                __hip_atomic_fetch_add((unsigned long long*)bufs + e, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                ((float*)bufs)[e] = 1.0f;
            }
        }
    }
}

__global__ void xcc_hist_kernel(int* hist, int* mism) {
    if (threadIdx.x == 0) {
        const int x = xcc_id();
        atomicAdd(hist + x, 1);
        if (x != (int)(blockIdx.x % 8)) atomicAdd(mism, 1);
    }
}

// streaming read: every lane 16 bytes per load, UNR loads in flight, grid-stride
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int UNR>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* p, long n, unsigned* sink) {
    unsigned acc = 0;
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// streaming write: every lane 16 bytes per store, grid-stride; NT = non-temporal
template <int NT>
__global__ __launch_bounds__(256) void write_kernel(u32x4* p, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, p + i);
        else p[i] = v;
    }
}
// the recorder's pattern: a wave owns a contiguous 16 KiB chunk and writes it with 64 dword stores, each covering two
// full 128-byte lines (lanes 0..31 one line, 32..63 the line 256 bytes further on)
template <int NT>
__global__ __launch_bounds__(512) void write_rec_kernel(unsigned* p, long n_chunks) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long c = wave; c < n_chunks; c += waves) {
        unsigned* base = p + c * 4096;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) {
            unsigned* a = base + (i >> 1) * 128 + (i & 1) * 32 + (lane >> 5) * 64 + (lane & 31);
            if (NT) __builtin_nontemporal_store((unsigned)i, a);
            else *a = (unsigned)i;
        }
    }
}

// Stores next to arithmetic: every wave alternates ~`work` dependent FMAs with stores of one 16-KiB chunk per 64 iterations.
// MODE 0: no stores; 1: one dword store per iteration (the recorder's pattern); 2: one dwordx4 store every 4th iteration
// (the same bytes in a quarter of the instructions).  Does the store STREAM hide under the arithmetic, or add to it?
template <int MODE>
__global__ __launch_bounds__(512) void mix_kernel(unsigned* p, long n_chunks, int work, float* sink) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((long)gridDim.x * blockDim.x) >> 6;
    float a = (float)lane, b = 1.0001f;
    for (long c = wave; c < n_chunks; c += waves) {
        unsigned* base = p + c * 4096;
#pragma unroll 4
        for (int i = 0; i < 64; ++i) {
            for (int k = 0; k < work; ++k) a = __builtin_fmaf(a, b, 0.5f);
            if (MODE == 1) {
                __builtin_nontemporal_store(__float_as_uint(a), base + (i >> 1) * 128 + (i & 1) * 32 + (lane >> 5) * 64 + (lane & 31));
            } else if (MODE == 2) {
                if ((i & 3) == 3) {
                    const u32x4 v = {__float_as_uint(a), 1u, 2u, 3u};
                    __builtin_nontemporal_store(v, (u32x4*)(base + (i >> 2) * 256) + lane);
                }
            }
        }
    }
    if (a == 12345.678f) *sink = a;
}

template <class F> static float time_ms(F f, int n) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / n;
}

int main() {
    // ---- block -> XCD map -------------------------------------------------------------------------------------
    int *hist, *mism;
    CK(hipMalloc(&hist, 64)); CK(hipMalloc(&mism, 4));
    CK(hipMemset(hist, 0, 64)); CK(hipMemset(mism, 0, 4));
    hipLaunchKernelGGL(xcc_hist_kernel, dim3(512), dim3(64), 0, 0, hist, mism);
    int h[16], mm;
    CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&mm, mism, 4, hipMemcpyDeviceToHost));
    printf("xcc histogram of 512 blocks:");
    for (int i = 0; i < 8; ++i) printf(" %d", h[i]);
    printf("  (blocks with xcc != block %% 8: %d)\n", mm);

    // ---- flush patterns ---------------------------------------------------------------------------------------
    const int tiles_per_buf = 9;                              // 9 tiles of 256 x 256 = 2.36 MB f32 per XCD
    const long buf_elems = (long)tiles_per_buf * 65536;
    void* bufs;
    CK(hipMalloc(&bufs, 8 * buf_elems * 8));
    CK(hipMemset(bufs, 0, 8 * buf_elems * 8));
    const int rounds = 9;                                     // one workgroup pass = 9 layer flushes
    const char* names[] = {"f32 wg-scope, per-XCD buffers", "f32 agent-scope, per-XCD buffers", "u64 wg-scope, per-XCD buffers",
                           "u64 agent-scope, per-XCD buffers", "plain f32 stores, per-XCD buffers"};
    for (int grid : {256, 512, 1024}) {
        for (int mode = 0; mode < 5; ++mode) {
            for (int priv = 1; priv >= 0; --priv) {
                if (!priv && (mode == 0 || mode == 2 || mode == 4)) continue;     // one shared buffer needs agent scope
                auto f = [&]() {
                    switch (mode) {
                        case 0: hipLaunchKernelGGL(flush_kernel<0>, dim3(grid), dim3(512), 0, 0, bufs, buf_elems, tiles_per_buf, rounds, priv); break;
                        case 1: hipLaunchKernelGGL(flush_kernel<1>, dim3(grid), dim3(512), 0, 0, bufs, buf_elems, tiles_per_buf, rounds, priv); break;
                        case 2: hipLaunchKernelGGL(flush_kernel<2>, dim3(grid), dim3(512), 0, 0, bufs, buf_elems, tiles_per_buf, rounds, priv); break;
                        case 3: hipLaunchKernelGGL(flush_kernel<3>, dim3(grid), dim3(512), 0, 0, bufs, buf_elems, tiles_per_buf, rounds, priv); break;
                        default: hipLaunchKernelGGL(flush_kernel<4>, dim3(grid), dim3(512), 0, 0, bufs, buf_elems, tiles_per_buf, rounds, priv); break;
                    }
                };
                const float ms = time_ms(f, 10);
                const double elems = (double)grid * rounds * 65536;
                printf("grid %4d  %-36s %s: %8.3f ms  %7.1f G elem/s  (a field's 512 passes x 9 tiles = 302 M elems: %6.3f ms)\n", grid,
                       names[mode], priv ? "" : "[ONE shared buffer]", ms, elems / ms / 1e6, 302e6 / (elems / ms) );
            }
        }
    }
    // correctness of the wg-scope f32 path: after a known number of adds every element of XCD x holds an integer
    CK(hipMemset(bufs, 0, 8 * buf_elems * 8));
    hipLaunchKernelGGL(flush_kernel<0>, dim3(512), dim3(512), 0, 0, bufs, buf_elems, tiles_per_buf, rounds, 1);
    CK(hipDeviceSynchronize());
    {
        std::vector<float> hb(8 * buf_elems);
        CK(hipMemcpy(hb.data(), bufs, 8 * buf_elems * 4, hipMemcpyDeviceToHost));
        double tot = 0;
        for (float v : hb) tot += v;
        printf("wg-scope f32 sum check: %.0f adds landed of %.0f issued\n", tot, 512.0 * rounds * 65536);
    }

    // ---- streaming reads ----------------------------------------------------------------------------------------
    const long bytes = 3l << 30;
    u32x4* big;
    unsigned* sink;
    CK(hipMalloc(&big, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(big, 1, bytes));
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        const int grid = 256 * wgs_per_cu;
        float ms4 = time_ms([&]() { hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, big, bytes / 16, sink); }, 5);
        float ms8 = time_ms([&]() { hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, big, bytes / 16, sink); }, 5);
        printf("read 3 GiB, %d x 256-thread blocks per CU: 4 loads in flight %.2f TB/s, 8 in flight %.2f TB/s\n", wgs_per_cu,
               bytes / ms4 / 1e9, bytes / ms8 / 1e9);
    }
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        const int grid = 256 * wgs_per_cu;
        float a = time_ms([&]() { hipLaunchKernelGGL(write_kernel<0>, dim3(grid), dim3(256), 0, 0, big, bytes / 16); }, 5);
        float b = time_ms([&]() { hipLaunchKernelGGL(write_kernel<1>, dim3(grid), dim3(256), 0, 0, big, bytes / 16); }, 5);
        printf("write 3 GiB, %d x 256-thread blocks per CU, 16 B per lane: plain %.2f TB/s, non-temporal %.2f TB/s\n", wgs_per_cu,
               bytes / a / 1e9, bytes / b / 1e9);
    }
    for (int wgs_per_cu : {1, 2}) {
        const int grid = 256 * wgs_per_cu;
        float a = time_ms([&]() { hipLaunchKernelGGL(write_rec_kernel<0>, dim3(grid), dim3(512), 0, 0, (unsigned*)big, bytes / 16384); }, 5);
        float b = time_ms([&]() { hipLaunchKernelGGL(write_rec_kernel<1>, dim3(grid), dim3(512), 0, 0, (unsigned*)big, bytes / 16384); }, 5);
        printf("write 3 GiB in the recorder's pattern (16 KiB chunk per wave, dword stores, two full lines each), %d x 512 threads per CU: plain %.2f TB/s, non-temporal %.2f TB/s\n",
               wgs_per_cu, bytes / a / 1e9, bytes / b / 1e9);
    }
    {
        float* fs;
        CK(hipMalloc(&fs, 4));
        const long chunks = (1l << 30) / 16384;            // 1 GiB of stores per launch: 16 chunks per wave at 512 blocks x 8 waves
        for (int work : {16, 32, 64}) {
            float t0 = time_ms([&]() { hipLaunchKernelGGL(mix_kernel<0>, dim3(512), dim3(512), 0, 0, (unsigned*)big, chunks, work, fs); }, 5);
            float t1 = time_ms([&]() { hipLaunchKernelGGL(mix_kernel<1>, dim3(512), dim3(512), 0, 0, (unsigned*)big, chunks, work, fs); }, 5);
            float t2 = time_ms([&]() { hipLaunchKernelGGL(mix_kernel<2>, dim3(512), dim3(512), 0, 0, (unsigned*)big, chunks, work, fs); }, 5);
            printf("mix: %2d FMAs per iteration, 1 GiB of stores: arithmetic alone %.1f us, + dword stores %.1f us (%.2f TB/s), + dwordx4 stores %.1f us (%.2f TB/s)\n",
                   work, t0 * 1e3, t1 * 1e3, 1.0737 / t1, t2 * 1e3, 1.0737 / t2);
        }
    }
    return 0;
}
