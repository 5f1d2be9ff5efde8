// Host-side view of the traversal layout (no HIP headers): what the host tree builders fill.
#ifndef PT_TRAVERSE_HOST_H
#define PT_TRAVERSE_HOST_H

#include <vector>

#include "pt_internal.h"

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

// Same 128 bytes as the device's QuadNode32 (four children: six bounds each, single precision, rounded
// outward; an empty slot holds NaN bounds and child 0), and the exact boxes beside it (QuadBoxes64).  The lowest mantissa
// bytes of the x and y planes carry the child references a second time (collapse_to_quads puts them there).
struct QuadNode32Host {
    float plane[6][4];
    int32_t child[4];      // >= 0: quad index, < 0: ~primitive
    int32_t pad_[4];
};
static_assert(sizeof(QuadNode32Host) == 128, "QuadNode32Host mirrors QuadNode32");
struct QuadBoxes64Host { double box[4][6]; };

struct QuadHierarchy {
    pt::UnfilledVector<QuadNode32Host> nodes;      // node 0 is the root
    pt::UnfilledVector<QuadBoxes64Host> exact;
    int deepest_pending = 0;                // the most stack entries that can be waiting anywhere in a search
};

}  // namespace ptd

namespace pt {

// Surface-area-heuristic hierarchy over `boxes` (one primitive per leaf, pre-order pair
// numbering, height <= max_height).  *root_ref is a child reference (~0 for a single box).
int build_sah_pairs(const PtAabb* boxes, int count, int max_height, pt::UnfilledVector<ptd::PairNodeHost>* pairs, int* root_ref,
                    int* height_out);


// References for the search hierarchy (presplit.cpp): boxes[i] bounds a part of triangle primitive[i]; large triangles that
// fill little of their boxes are cut into several references.  Returns how many were added to the `count` uncut ones.
long long presplit_triangles(const PtTriangle* triangles, long long count, const PtAabb* own_boxes, double floor_factor, double extra_fraction,
                             pt::UnfilledVector<PtAabb>* boxes, std::vector<int32_t>* primitive);

// The binary hierarchy `pairs` (root = pair 0) collapsed to four children per node: a node takes the two
// children of a pair and then, while a slot is free, replaces its largest inner child by that child's two.
// Built by several threads; the result does not depend on how many.
void collapse_to_quads(const ptd::PairNodeHost* pairs, size_t pair_count, ptd::QuadHierarchy* out);

}  // namespace pt

#endif
