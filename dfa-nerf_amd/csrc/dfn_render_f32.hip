// dfn_render_f32.hip - the render / decoder kernels of the f32 tier (templates: dfn_render_kernels.h)
#include "dfn_render_kernels.h"

namespace dfn {
hipError_t launch_render_f32(const RenderArgs& A, hipStream_t st) { return launch_render_tier<TIER_F32, true>(A, st); }
hipError_t launch_decoder_f32(const DecoderArgs& A, hipStream_t st) { return launch_decoder_tier<TIER_F32, true>(A, st); }
}  // namespace dfn
