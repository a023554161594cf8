// dfn_train.h - argument blocks and launchers of the training kernels (dfn_train.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "dfanerf.h"
#include "dfn_layout.h"

namespace dfn {

struct MlpBwdArgs {
    const char* wblob_T;        // transposed packed stream of the field
    int nslab;
    const float* samples;       // [NP][8] forward outputs (sigma_h, rgb_h, sigma_t, rgb_t)
    const float* dsamples;      // [NP][8] gradients w.r.t. them
    const unsigned* masks;      // forward ReLU bits of the field
    void* dy_T;                 // out: tile-major [NP/32][rows][32] pre-activation gradients
    long NP;
};

struct CompositeBwdArgs {
    DfnFrame frame;
    const int* pix_index;
    const float* bg_f32;
    const unsigned char* bg_u8;
    const float* samples;       // [rays][64][8]
    const float* d_rgb_head;    // [rays][3]
    const float* d_rgb_com;     // [rays][3] or null
    float* dsamples;            // out [rays][64][8]
};

struct WOp {                    // one weight-gradient GEMM: C[M x N] = dy_T[a_row.., :] * act_T[b_row.., :]^T
    int a_row, M, b_row, N, c_off;
    int bias_owner;             // this GEMM also accumulates the row sums of its dy_T rows (bias gradients)
};

hipError_t launch_mlp_bwd(int tier, int field, const MlpBwdArgs& A, hipStream_t st);
hipError_t launch_mlp_bwd_bf16(bool torso, const MlpBwdArgs& A, hipStream_t st);     // dfn_bwd_bf16.hip
void bwd_program_info(int tier, int field, ProgramInfo* out);
hipError_t launch_composite_bwd(const CompositeBwdArgs& A, hipStream_t st);
hipError_t launch_wgrad(int tier, int field, const WOp* ops_dev, int n_ops, const int* prefix_dev, int total_items,
                        const void* dy_T, const void* act_T, long NP, int ksplit, float* C, const int* e_of, float* dbias,
                        hipStream_t st);
// bf16 tier: one workgroup per (GEMM, slice of the points), operands through LDS (dfn_wgrad_bf16.hip); order = GEMMs by
// decreasing size
hipError_t launch_wgrad_bf16(int field, const WOp* ops_dev, const int* order_dev, int n_ops, const void* dy_T,
                             const void* act_T, long NP, int ksplit, float* C, const int* e_of, float* dbias,
                             hipStream_t st);
hipError_t launch_scatter_add(const int* map, const float* dense, long n, float* grad_flat, hipStream_t st);
// macro-tile of one wgrad wave, in 32x32 output tiles (dfn_api.hip sizes the work list with the same numbers)
constexpr int WG_MT = 2, WG_NT = 4;
constexpr int WG_PF = 4;      // register prefetch depth of wgrad_kernel (f32 tier), in 8-point steps
hipError_t launch_bias_grad(int tier, int field, const int* row_of, int n, const void* dy_T, long NP, float* dbias,
                            hipStream_t st);

}  // namespace dfn
