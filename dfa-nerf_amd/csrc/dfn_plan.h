// dfn_plan.h - host-side pack planner (see dfn_plan.cpp)
#pragma once
#include <cstdint>
#include <vector>

namespace dfn {
// Fills `plan` (one int32 per packed element, padded to whole slabs) and returns the number of
// fragments one pass consumes.  field: 0 head, 1 torso, 2 listener (head program, listener weights).
long build_pack_plan(int tier, int field, std::vector<int32_t>& plan);
}  // namespace dfn
