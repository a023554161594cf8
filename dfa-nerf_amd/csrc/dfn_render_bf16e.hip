// dfn_render_bf16e.hip - the 16-bit training forwards whose recorder writes act_T as MX-fp8 e4m3 instead of MX-fp4
// (DFN_TRAIN_ACT_E4M3 or'ed into the tier of dfn_train_fwd*: the run-time opt-out of the narrow activation format, for A/B
// runs of the two formats on real data in one process).  Templates: dfn_render_kernels.h; a unit of its own so that the two
// kernels compile next to the others.
#include "dfn_render_kernels.h"

namespace dfn {
hipError_t launch_train_bf16_e4m3(const RenderArgs& A, hipStream_t st) {
    return A.frame.n_fine > 0 ? launch_render_t<TIER_BF16, true, 2, false>(A, st) : launch_render_t<TIER_BF16, true, 1, false>(A, st);
}
}  // namespace dfn
