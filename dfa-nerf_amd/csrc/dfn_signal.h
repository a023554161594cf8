// dfn_signal.h - launchers of the conditioning-signal kernels (dfn_signal.hip)
#pragma once
#include <hip/hip_runtime.h>
namespace dfn {
hipError_t launch_encode_signal(const float* aud_params, const float* exp_params, const float* att_params,
                                const float* auds, const float* exps, int N, const int* frame_ids, int n_frames, int smo,
                                float* out, float* keep, hipStream_t st);
// dfn_encode_signal_keep's buffer (window <= 8 rows): activations h1 | h2 | e1 | feat, then the backward's scratch d h1 | d h2
constexpr int SIG_KEEP_FLOATS = 8 * (256 + 128 + 32 + 96) + 8 * (256 + 128);
hipError_t launch_encode_signal_torso(const float* att_params, const float* poses, int pose_stride, int N,
                                      const int* frame_ids, int n_frames, int smo, float* out, hipStream_t st);
}  // namespace dfn
namespace dfn {
hipError_t launch_encode_signal_bwd(const float* aud_params, const float* exp_params, const float* att_params,
                                    const float* auds, const float* exps, int N, int frame, int smo, const float* d_out,
                                    float* g_aud, float* g_exp, float* g_att, bool set, const float* kept, hipStream_t st);
hipError_t launch_encode_signal_torso_bwd(const float* att_params, const float* poses, int pose_stride, int N, int frame,
                                          int smo, const float* d_out, float* g_att, bool set, hipStream_t st);
}  // namespace dfn
