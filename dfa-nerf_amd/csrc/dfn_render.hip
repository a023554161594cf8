// dfn_render.hip - tier dispatch of the fused frame renderer and the fused decoder.
// The kernels are templates in dfn_render_kernels.h, instantiated per precision tier in dfn_render_{f32,bf16,f16}.hip.
#include <hip/hip_runtime.h>
#include "dfn_layout.h"
#include "dfn_mlp.h"
#include "dfn_params.h"

namespace dfn {

hipError_t launch_render_f32(const RenderArgs& A, hipStream_t st);
hipError_t launch_render_bf16(const RenderArgs& A, hipStream_t st);
hipError_t launch_render_f16(const RenderArgs& A, hipStream_t st);
hipError_t launch_decoder_f32(const DecoderArgs& A, hipStream_t st);
hipError_t launch_decoder_bf16(const DecoderArgs& A, hipStream_t st);
hipError_t launch_decoder_f16(const DecoderArgs& A, hipStream_t st);

hipError_t launch_render(int tier, const RenderArgs& A, hipStream_t st) {
    switch (tier) {
    case TIER_BF16: return launch_render_bf16(A, st);
    case TIER_F16: return launch_render_f16(A, st);
    default: return launch_render_f32(A, st);
    }
}
hipError_t launch_decoder(int tier, const DecoderArgs& A, hipStream_t st) {
    switch (tier) {
    case TIER_BF16: return launch_decoder_bf16(A, st);
    case TIER_F16: return launch_decoder_f16(A, st);
    default: return launch_decoder_f32(A, st);
    }
}

// the 16-bit tiers share one program (same fragment counts and bias blob)
void program_info(int tier, int field, ProgramInfo* out) {
    if (tier != TIER_F32) {
        using P = Prog<TIER_BF16>;
        static_assert(Prog<TIER_F16>::H_FRAGS == P::H_FRAGS && Prog<TIER_F16>::T_FRAGS == P::T_FRAGS, "16-bit tiers");
        *out = field == FIELD_TORSO ? ProgramInfo{P::T_FRAGS, P::T_SLABS, P::T_NBIAS}
                                    : ProgramInfo{P::H_FRAGS, P::H_SLABS, P::H_NBIAS};
    } else {
        using P = Prog<TIER_F32>;
        *out = field == FIELD_TORSO ? ProgramInfo{P::T_FRAGS, P::T_SLABS, P::T_NBIAS}
                                    : ProgramInfo{P::H_FRAGS, P::H_SLABS, P::H_NBIAS};
    }
}

}  // namespace dfn
