// Host-side view of the traversal layout (no HIP headers): what the host tree builders fill.
#ifndef PT_TRAVERSE_HOST_H
#define PT_TRAVERSE_HOST_H

#include <vector>

#include "../../include/pathtracer_amd.h"

namespace ptd {

// Same 128 bytes as the device's PairNode: both children's boxes, then both child
// references (>= 0: pair index, < 0: ~primitive index).
struct PairNodeHost {
    PtAabb lbox;
    PtAabb rbox;
    int32_t left, right;
    int32_t pad_[6];
};
static_assert(sizeof(PairNodeHost) == 128, "PairNodeHost mirrors PairNode");

}  // namespace ptd

namespace pt {

// Surface-area-heuristic hierarchy over `boxes` (one primitive per leaf, pre-order pair
// numbering, height <= max_height).  *root_ref is a child reference (~0 for a single box).
int build_sah_pairs(const std::vector<PtAabb>& boxes, int max_height, std::vector<ptd::PairNodeHost>* pairs, int* root_ref,
                    int* height_out);

}  // namespace pt

#endif
