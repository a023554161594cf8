// dfn_bwd_bf16.hip - bf16 instantiation of the MLP backward kernel (dfn_bwd_kernel.h).
// A translation unit of its own: the two kernels are the longest compile of the library.  (For a while this unit was
// built with LLVM's minimum-register scheduler, which cut their spills from ~200 to 17-89 registers; since the
// operand vectors are converted in register pairs and the ReLU bits applied with bfe + and (dfn_mlp.h: acc_to_vec,
// dfn_bwd.h: apply_mask) the default scheduler needs no spill at all and is 8 % faster than that build.)
// Round 3 (MX-fp8 recording, packed mask application): 16 spilled registers in the torso kernel, none in the head kernel.
#define DFN_DPP_ASM 0           // q8_of_tiles: the builtin DPP steps (dfn_mlp.h)
#include "dfn_bwd_kernel.h"

namespace dfn {

hipError_t launch_mlp_bwd_bf16(bool torso, const MlpBwdArgs& A, hipStream_t st) {
    return torso ? launch_mlp_bwd_t<TIER_BF16, true>(A, st) : launch_mlp_bwd_t<TIER_BF16, false>(A, st);
}

}  // namespace dfn

#ifdef DFN_TIMING
// developer build only (tools/time_dx.py): copy the per-wave cycle counters of the last dX launch to the host
extern "C" int dfn_debug_bwd_timing(unsigned long long* out, long n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dfn::g_bwd_timing), sizeof(unsigned long long) * (size_t)n, 0,
                                    hipMemcpyDeviceToHost);
}
#endif
