// dfn_misc.h - launchers of the small kernels (dfn_misc.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "dfanerf.h"

namespace dfn {
struct PrepareJobs {            // dfn_train_prepare: 4 pack jobs (head, torso, head^T, torso^T) + 2 bias folds (head, torso)
    const float* params;
    int tier;
    const int* plan[4]; void* out[4]; long n[4]; int pack_blocks[4];
    const float* sig[2]; const float* zs[2]; const float* za[2]; float* bias[2]; int nb[2]; int fold_blocks[2];
};
hipError_t launch_prepare(PrepareJobs J, hipStream_t st);
hipError_t launch_pack(const int* plan, const float* params, void* out, long n, int tier, hipStream_t st);
hipError_t launch_fold_bwd(int field, const float* params, const float* sig, const float* zs, const float* za,
                           const float* dbias, float* grad_flat, float* dsig, int n, hipStream_t st);
hipError_t launch_fold_bwd_sig(int field, const float* params, const float* dbias, float* dsig, bool overwrite,
                               hipStream_t st);
int sig_term_elements(int field, int* out);      // <= 512 bias-blob elements whose fold carries a signal term
hipError_t launch_fold(int field, const float* params, const float* sig, const float* zs, const float* za,
                       float* out, int n, hipStream_t st);
hipError_t launch_get_rays(int H, int W, int stride, float focal, float cx, float cy, const float* c2w_host, float* ro,
                           float* rd, hipStream_t st);
hipError_t launch_ndc_rays(int H, int W, float focal, float z_near, const float* ro, const float* rd, long n,
                           float* oo, float* od, hipStream_t st);
hipError_t launch_sample_pdf(const float* bins, const float* weights, long R, int nb, int ns, const float* u,
                             float* out, hipStream_t st);
hipError_t launch_composite(const float* sigma, const float* feat, int K, long N, float* ssum, float* fw,
                            hipStream_t st);
hipError_t launch_volume_weights(const float* z, const float* ray, const float* sigma, long R, int S,
                                 float last_dist, float* w, hipStream_t st);
hipError_t launch_composite_grad(const float* sigma, const float* feat, int K, long N, const float* d_ssum, const float* d_fw,
                                 float* d_sigma, float* d_feat, hipStream_t st);
hipError_t launch_volume_weights_grad(const float* z, const float* ray, const float* sigma, long R, int S, float last_dist,
                                      const float* d_w, float* d_sigma, hipStream_t st);
hipError_t launch_zero_words(unsigned* p, long n, hipStream_t st);
hipError_t launch_mfma_chain(bool f16, int lds2, int valu2, const void* frags, const void* b, int iters, int blocks, float* out,
                             unsigned long long* clk, hipStream_t st);
hipError_t launch_to8b(const float* x, long n, unsigned char* out, hipStream_t st);
hipError_t launch_mfma_probe(float* out, hipStream_t st);
hipError_t launch_adam_multi(const DfnAdamItem* items, const void* chunks, int n_chunks, float lr, double beta1, double beta2,
                             float eps, float bias_c1, float bias_c2_sqrt, hipStream_t st);
}  // namespace dfn
