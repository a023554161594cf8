// dfn_render_bf16.hip - the render / decoder kernels of the bf16 tier (templates: dfn_render_kernels.h)
#include "dfn_render_kernels.h"

namespace dfn {
hipError_t launch_render_bf16(const RenderArgs& A, hipStream_t st) { return launch_render_tier<TIER_BF16, true>(A, st); }
hipError_t launch_decoder_bf16(const DecoderArgs& A, hipStream_t st) { return launch_decoder_tier<TIER_BF16, true>(A, st); }
}  // namespace dfn
