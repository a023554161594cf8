// dfn_misc.h - launchers of the small kernels (dfn_misc.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "dfanerf.h"

namespace dfn {
hipError_t launch_pack(const int* plan, const float* params, void* out, long n, int tier, hipStream_t st);
hipError_t launch_fold_bwd(int field, const float* params, const float* sig, const float* zs, const float* za,
                           const float* dbias, float* grad_flat, float* dsig, int n, hipStream_t st);
hipError_t launch_fold(int field, const float* params, const float* sig, const float* zs, const float* za,
                       float* out, int n, hipStream_t st);
hipError_t launch_get_rays(int H, int W, float focal, float cx, float cy, const float* c2w_host, float* ro,
                           float* rd, hipStream_t st);
hipError_t launch_ndc_rays(int H, int W, float focal, float z_near, const float* ro, const float* rd, long n,
                           float* oo, float* od, hipStream_t st);
hipError_t launch_sample_pdf(const float* bins, const float* weights, long R, int nb, int ns, const float* u,
                             float* out, hipStream_t st);
hipError_t launch_composite(const float* sigma, const float* feat, int K, long N, float* ssum, float* fw,
                            hipStream_t st);
hipError_t launch_volume_weights(const float* z, const float* ray, const float* sigma, long R, int S,
                                 float last_dist, float* w, hipStream_t st);
hipError_t launch_to8b(const float* x, long n, unsigned char* out, hipStream_t st);
hipError_t launch_mfma_probe(float* out, hipStream_t st);
hipError_t launch_adam_multi(const DfnAdamItem* items, const void* chunks, int n_chunks, float lr, double beta1, double beta2,
                             float eps, float bias_c1, float bias_c2_sqrt, hipStream_t st);
}  // namespace dfn
