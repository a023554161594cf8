// dfn_signal.h - launchers of the conditioning-signal kernels (dfn_signal.hip)
#pragma once
#include <hip/hip_runtime.h>
namespace dfn {
hipError_t launch_encode_signal(const float* aud_params, const float* exp_params, const float* att_params,
                                const float* auds, const float* exps, int N, const int* frame_ids, int n_frames, int smo,
                                float* out, hipStream_t st);
hipError_t launch_encode_signal_torso(const float* att_params, const float* poses, int pose_stride, int N,
                                      const int* frame_ids, int n_frames, int smo, float* out, hipStream_t st);
}  // namespace dfn
