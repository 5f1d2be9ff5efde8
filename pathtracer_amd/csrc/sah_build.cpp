// The GPU's own hierarchy over the leaf boxes of a reference BVH.
//
// The reference's tree (src/bvh.cpp:20-58) is a median split on a cycling axis over boxes
// sorted by their minimum: cheap to build, expensive to traverse.  Which leaves the reference
// reaches does not depend on that topology: for a ray with finite reciprocals a box passes
// the slab test only if every box containing it passes, so a leaf is reached exactly when its
// own box passes.  Any hierarchy whose inner boxes are exact unions of the same leaf boxes
// therefore finds the same leaves, and the closest valid hit among them is the reference's
// answer whenever it is unique (ties and non-finite rays go through the reference tree itself).
//
// This file builds that hierarchy top-down with a binned surface-area heuristic, one
// primitive per leaf, numbered in pre-order, with its height capped so that the traversal
// stack in LDS stays small: a split is only accepted if both sides can still be finished as
// balanced subtrees within the remaining levels.
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#include "pt_internal.h"
#include "pt_traverse_host.h"

namespace pt {

namespace {

constexpr int BINS = 16;

inline PtAabb empty_box() {
    const double inf = INFINITY;
    return PtAabb{PtVec3{inf, inf, inf}, PtVec3{-inf, -inf, -inf}};
}
inline void grow(PtAabb& a, const PtAabb& b) {
    a.min.x = std::min(a.min.x, b.min.x); a.min.y = std::min(a.min.y, b.min.y); a.min.z = std::min(a.min.z, b.min.z);
    a.max.x = std::max(a.max.x, b.max.x); a.max.y = std::max(a.max.y, b.max.y); a.max.z = std::max(a.max.z, b.max.z);
}
inline double half_area(const PtAabb& b) {
    const double dx = b.max.x - b.min.x, dy = b.max.y - b.min.y, dz = b.max.z - b.min.z;
    return dx * dy + dy * dz + dz * dx;
}
inline double axis_of(const PtVec3& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
inline int levels_for(int count) {      // height of a balanced tree over `count` leaves
    int levels = 1;
    while ((1ll << (levels - 1)) < count) ++levels;
    return levels;
}

struct Builder {
    const std::vector<PtAabb>& boxes;
    std::vector<PtVec3> centroids;
    std::vector<int> order;
    ptd::PairNodeHost* pairs;
    int max_height;

    // Builds the subtree over order[begin, end) whose root pair (if any) is pairs[node];
    // `level` is the depth of that root (tree root = 1).  Returns the child reference and box.
    int build(int begin, int end, int node, int level, int spawn_budget, PtAabb* box_out) {
        const int count = end - begin;
        if (count == 1) {
            *box_out = boxes[order[begin]];
            return ~order[begin];
        }
        PtAabb centroid_bounds = empty_box();
        for (int i = begin; i < end; ++i) {
            const PtVec3& c = centroids[order[i]];
            grow(centroid_bounds, PtAabb{c, c});
        }
        const int mid = choose_split(begin, end, level, centroid_bounds);
        const int left_count = mid - begin;
        PtAabb left_box, right_box;
        int left_ref, right_ref;
        if (spawn_budget > 1 && count > 65536) {
            std::thread other([&]() { left_ref = build(begin, mid, node + 1, level + 1, spawn_budget / 2, &left_box); });
            right_ref = build(mid, end, node + left_count, level + 1, spawn_budget - spawn_budget / 2, &right_box);
            other.join();
        } else {
            left_ref = build(begin, mid, node + 1, level + 1, 1, &left_box);
            right_ref = build(mid, end, node + left_count, level + 1, 1, &right_box);
        }
        ptd::PairNodeHost& p = pairs[node];
        p.lbox = left_box;
        p.rbox = right_box;
        p.left = left_ref;
        p.right = right_ref;
        *box_out = left_box;
        grow(*box_out, right_box);
        return node;
    }

    // Partitions order[begin, end) and returns the split position (begin < mid < end).
    int choose_split(int begin, int end, int level, const PtAabb& cb) {
        const int count = end - begin;
        const int levels_left = max_height - level;     // levels available below this node's children level
        // a side of m leaves needs levels_for(m) levels starting at level + 1
        auto fits = [&](int m) { return levels_for(m) <= levels_left; };

        int best_axis = -1, best_bin = -1;
        double best_cost = INFINITY;
        if (count > 2) {
            for (int axis = 0; axis < 3; ++axis) {
                const double lo = axis_of(cb.min, axis), hi = axis_of(cb.max, axis);
                if (!(hi > lo)) continue;
                const double scale = BINS / (hi - lo);
                PtAabb bin_box[BINS];
                int bin_count[BINS] = {};
                for (PtAabb& b : bin_box) b = empty_box();
                for (int i = begin; i < end; ++i) {
                    const int id = order[i];
                    int b = static_cast<int>((axis_of(centroids[id], axis) - lo) * scale);
                    b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    ++bin_count[b];
                    grow(bin_box[b], boxes[id]);
                }
                double right_area[BINS];
                int right_count[BINS];
                PtAabb acc = empty_box();
                int n = 0;
                for (int b = BINS - 1; b > 0; --b) {
                    grow(acc, bin_box[b]);
                    n += bin_count[b];
                    right_area[b] = half_area(acc);
                    right_count[b] = n;
                }
                acc = empty_box();
                n = 0;
                for (int b = 0; b < BINS - 1; ++b) {
                    grow(acc, bin_box[b]);
                    n += bin_count[b];
                    const int nr = right_count[b + 1];
                    if (n == 0 || nr == 0 || !fits(n) || !fits(nr)) continue;
                    const double cost = half_area(acc) * n + right_area[b + 1] * nr;
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }
        int* first = order.data() + begin;
        int* last = order.data() + end;
        if (best_axis >= 0) {
            const double lo = axis_of(cb.min, best_axis), hi = axis_of(cb.max, best_axis);
            const double scale = BINS / (hi - lo);
            int* middle = std::partition(first, last, [&](int id) {
                int b = static_cast<int>((axis_of(centroids[id], best_axis) - lo) * scale);
                b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                return b <= best_bin;
            });
            return begin + static_cast<int>(middle - first);
        }
        // no acceptable binned split (coincident centroids, or the height cap): halve on the widest axis
        int axis = 0;
        const double ex = cb.max.x - cb.min.x, ey = cb.max.y - cb.min.y, ez = cb.max.z - cb.min.z;
        if (ey > ex && ey >= ez) axis = 1; else if (ez > ex && ez > ey) axis = 2;
        int* middle = first + count / 2;
        std::nth_element(first, middle, last, [&](int l, int r) {
            const double a = axis_of(centroids[l], axis), b = axis_of(centroids[r], axis);
            return a < b || (a == b && l < r);
        });
        return begin + count / 2;
    }
};

}  // namespace

int build_sah_pairs(const std::vector<PtAabb>& boxes, int max_height, std::vector<ptd::PairNodeHost>* pairs, int* root_ref,
                    int* height_out) {
    const int count = static_cast<int>(boxes.size());
    if (count == 0) return fail(PT_ERR_INVALID_ARGUMENT, "build_sah_pairs: no boxes");
    if (levels_for(count) > max_height) return fail(PT_ERR_UNSUPPORTED, "%d leaves cannot fit a tree of height %d", count, max_height);
    pairs->assign(count > 1 ? count - 1 : 0, ptd::PairNodeHost{});
    Builder b{boxes, {}, {}, pairs->data(), max_height};
    b.centroids.resize(count);
    b.order.resize(count);
    for (int i = 0; i < count; ++i) {
        b.centroids[i] = PtVec3{0.5 * (boxes[i].min.x + boxes[i].max.x), 0.5 * (boxes[i].min.y + boxes[i].max.y),
                                0.5 * (boxes[i].min.z + boxes[i].max.z)};
        b.order[i] = i;
    }
    const int threads = static_cast<int>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
    PtAabb root_box;
    *root_ref = b.build(0, count, 0, 1, threads, &root_box);
    // measured height (root pair = level 1, its leaf children = level 2, ...)
    int height = 1;
    if (count > 1) {
        std::vector<std::pair<int, int>> todo{{0, 1}};
        while (!todo.empty()) {
            const auto [node, level] = todo.back();
            todo.pop_back();
            const ptd::PairNodeHost& p = (*pairs)[node];
            for (int ref : {p.left, p.right}) {
                height = std::max(height, level + 1);
                if (ref >= 0) todo.emplace_back(ref, level + 1);
            }
        }
    }
    *height_out = height;
    return PT_OK;
}

}  // namespace pt
