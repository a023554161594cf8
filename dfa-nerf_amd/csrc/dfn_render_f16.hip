// dfn_render_f16.hip - the render / decoder kernels of the f16 tier (templates: dfn_render_kernels.h)
#include "dfn_render_kernels.h"

namespace dfn {
hipError_t launch_render_f16(const RenderArgs& A, hipStream_t st) { return launch_render_tier<TIER_F16, false>(A, st); }
hipError_t launch_decoder_f16(const DecoderArgs& A, hipStream_t st) { return launch_decoder_tier<TIER_F16, false>(A, st); }
}  // namespace dfn
