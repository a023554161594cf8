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
    const float* samples;       // [rays][S][8], S = 64 + n_fine, evaluation order
    const float* d_rgb_head;    // [rays][3]
    const float* d_rgb_com;     // [rays][3] or null
    float* dsamples;            // out [rays][S][8]
    // hierarchical step only (launch_composite_bwd_hier): what the forward left about the merge
    const float* z_all;         // [rays][S] merged, sorted depths
    const unsigned char* ranks; // [rays][S] merged rank of evaluated point i
    // optional: a buffer this launch also fills with zeros (the step's gradient buffer: its own fill launch was 9 us + a gap
    // between the compositing backward and the dX chain)
    float* zero_buf;
    long zero_floats;
};

struct WOp {                    // one weight-gradient GEMM: C[M x N] = dy_T[a_row.., :] * act_T[b_row.., :]^T
    int a_row, M, b_row, N, c_off;
    int bias_owner;             // this GEMM also accumulates the row sums of its dy_T rows (bias gradients)
};

hipError_t launch_mlp_bwd(int tier, int field, const MlpBwdArgs& A, hipStream_t st);
hipError_t launch_mlp_bwd_bf16(bool torso, const MlpBwdArgs& A, hipStream_t st);     // dfn_bwd_bf16.hip
void bwd_program_info(int tier, int field, ProgramInfo* out);
hipError_t launch_composite_bwd(const CompositeBwdArgs& A, hipStream_t st);
hipError_t launch_composite_bwd_hier(const CompositeBwdArgs& A, hipStream_t st);      // n_fine in {64, 128}
// Split-K partials: C [ksplit][c_stride] and dbias [ksplit][n_bias], one writer per element and slice (no atomics);
// launch_reduce_scatter / launch_reduce_bias add the slices in index order (bit-reproducible gradients).
// f32 tier: two launches per field, both with the operands through LDS (dfn_train.hip) -
//   wgrad_full_kernel    the 256 x 256 GEMMs (`full_ops_dev`: their indices into ops_dev), one workgroup per (GEMM, slice);
//   wgrad_narrow_kernel  every other GEMM, one workgroup per WNItem (below), all shapes side by side in ONE launch.
struct WNItem {                 // slice `ks` of row tiles [m_tile0, m_tile0 + the shape's MT) of GEMM `op` (f32 tier, narrow shapes)
    int op, ks, m_tile0, shape; // shape: WN_* (dfn_train.hip: the (MT, NT) instantiations of wgrad_lds_part)
};
enum WNShape : int { WN_4x4 = 0, WN_4x2, WN_1x8, WN_4x1, WN_2x2, WN_ROWS, WN_COUNT };
// classification of a GEMM M x N (dy_T rows x act_T rows) for the f32 tier: WN_* and the number of row parts it is cut into
// (-1: the 256 x 256 shape of wgrad_full_kernel; -2: a shape no kernel is instantiated for)
DFN_HD constexpr int wn_shape_of(int M, int N) {
    return (M == 256 && N == 256) ? -1 : (M == 256 && N == 128) ? WN_4x4 : (M == 256 && N == 64) ? WN_4x2 : (M == 32 && N == 256) ? WN_1x8
         : (M == 256 && N == 32) ? WN_4x1 : (M == 64 && N == 64) ? WN_2x2 : (N == 0 && M > 0 && M % 32 == 0) ? WN_ROWS : -2;
}
DFN_HD constexpr int wn_shape_mt(int shape) { return shape == WN_1x8 ? 1 : (shape == WN_2x2 || shape == WN_ROWS) ? 2 : 4; }
hipError_t launch_wgrad(int tier, int field, const WOp* ops_dev, const int* full_ops_dev, int n_full, const WNItem* nitems_dev,
                        int n_nitems, const void* dy_T, const void* act_T, long NP, int ksplit, float* C, long c_stride,
                        const int* e_of, float* dbias, int n_bias, hipStream_t st);
// bf16 tier: one workgroup per (GEMM, slice of the points), operands through LDS (dfn_wgrad_bf16.hip); order = GEMMs by
// decreasing size
struct WItem {                  // one workgroup of the 16-bit tier's weight-gradient launch: slice `ks` of `n` of GEMM `op`
    int op, ks, n, pad;
};
// act_fp4: act_T is MX-fp4 (recorded by the fused training step) / MX-fp8 e4m3 (by the decoder-on-points recorder)
hipError_t launch_wgrad_bf16(int field, bool act_fp4, const WOp* ops_dev, const WItem* items_dev, int n_items, const void* dy_T,
                             const void* act_T, long NP, float* C, long c_stride, const int* e_of, float* dbias, int n_bias,
                             hipStream_t st);
// grad_flat[map[i]] += sum over the first `slices` slices of parts[.][i]   (i < n; map[i] < 0: structural padding)
hipError_t launch_reduce_scatter(const int* map, const float* parts, long n, long stride, int slices, float* grad_flat,
                                 hipStream_t st);
// launch_reduce_bias + launch_reduce_scatter in one launch (same sums, same order)
// blk_n / bias_n (may be NULL: `slices` everywhere): slices the GEMM owning a 256-element block of C / a bias element was
// split into; `units` = tile pairs of the call - a GEMM split n ways over them fills ceil(units / ceil(units / n)) slices
hipError_t launch_reduce_both(const int* map, const float* parts, long n, long stride, int slices, float* grad_flat,
                              const int* rows, const float* bparts, int n_bias, float* dbias, const unsigned char* blk_n,
                              const unsigned char* bias_n, long units, hipStream_t st);
// dbias[e] = sum over slices of parts[.][e] for the elements that have a gradient row (rows[e] >= 0), 0 otherwise
hipError_t launch_reduce_bias(const int* rows, const float* parts, int n_bias, int slices, float* dbias, hipStream_t st);
constexpr int SAMPLE_PIXELS_CANDIDATES = 8192;
hipError_t launch_sample_pixels(int H, int W, int n, int rect_num, const int* rect, unsigned long long seed,
                                unsigned long long counter, int* out, int* status, hipStream_t st);
hipError_t launch_mse_loss(const float* rgb_head, const float* rgb_com, const unsigned char* img_head,
                           const unsigned char* img_com, const int* pix, int n, float* losses, float* d_head, float* d_com,
                           hipStream_t st);
constexpr int BIAS_GRAD_SLICES = 128;      // slices of the points in the streaming bias_grad_kernel
// streaming row sums: parts [BIAS_GRAD_SLICES][n] (workspace), then launch_reduce_bias
hipError_t launch_bias_grad(int tier, int field, const int* e_of, const int* rows, int n, const void* dy_T, long NP,
                            float* parts, float* dbias, hipStream_t st);

// row sums of the dy_T rows behind d(signal) only (dfn_signal_grad): parts [SIG_ROW_SLICES][n_sig], dbias[elem_of[i]] = sum
constexpr int SIG_ROW_SLICES = 128;
hipError_t launch_signal_rows(int tier, int field, const int* row_of, const int* elem_of, int n_sig, const void* dy_T, long NP,
                              float* parts, float* dbias, hipStream_t st);

}  // namespace dfn
