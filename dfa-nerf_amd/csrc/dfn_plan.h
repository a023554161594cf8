// dfn_plan.h - host-side pack planner (see dfn_plan.cpp)
#pragma once
#include <cstdint>
#include <vector>

namespace dfn {
// Fills `plan` (one int32 per packed element, padded to whole slabs) and returns the number of
// fragments one pass consumes.  field: 0 head, 1 torso, 2 listener (head program, listener weights).
long build_pack_plan(int tier, int field, std::vector<int32_t>& plan);
struct WOpHost {
    int a_row, M, b_row, N, c_off;
    int bias_owner;      // 1: the first GEMM that reads dy_T rows [a_row, a_row + M): it also produces their row sums
};
// Weight-gradient GEMM list of a field, the map dense-C element -> flat parameter index (or -1), and for every
// element of the field's bias blob the row of dy_T whose sum over the sample points is its gradient.
void build_wgrad_plan(int field, std::vector<WOpHost>& ops, std::vector<int32_t>& map, std::vector<int32_t>& bias_rows);
// Transposed (backward) stream of the same field, op order of dfn_bwd.h.  field: 0 head, 1 torso.
long build_bwd_plan(int tier, int field, std::vector<int32_t>& plan);
}  // namespace dfn
