// dfn_bwd_bf16.hip - bf16 instantiation of the MLP backward kernel (dfn_bwd_kernel.h).
// A translation unit of its own so build.sh can give it the minimum-register instruction scheduler: under the
// default one these two kernels spill 184-213 registers, under that one 17-89 (measured 447 -> 362 us and
// 391 -> 260 us per training step); the f32 tier and the weight-gradient kernels are faster with the default.
#include "dfn_bwd_kernel.h"

namespace dfn {

hipError_t launch_mlp_bwd_bf16(bool torso, const MlpBwdArgs& A, hipStream_t st) {
    return torso ? launch_mlp_bwd_t<TIER_BF16, true>(A, st) : launch_mlp_bwd_t<TIER_BF16, false>(A, st);
}

}  // namespace dfn
