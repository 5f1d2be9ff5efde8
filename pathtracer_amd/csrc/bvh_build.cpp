// Host BVH build: the reference's median-split tree (src/bvh.cpp:20-100), node for node.
//
// The reference numbers nodes in pre-order while recursing.  A subtree over c primitives
// always has 2c-1 nodes, so the index of every node is known before its subtree is built:
// the left child of node i is i+1 and the right child is i + 2*left_count.  That lets
// disjoint subtrees be built by different threads straight into their final slots.
//
// Ties matter: many triangles share a box minimum, and the order std::sort leaves equal
// keys in decides the topology.  Each range is therefore sorted with std::sort and the
// reference's comparator (strict < on the box minimum of one axis), exactly once per node
// as in src/bvh.cpp:33-38.  The reference's second sort on the way out (:49-55) orders a
// range nobody reads again and is not reproduced.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "pt_internal.h"

namespace pt {

PtAabb box_of(const PtTriangle& t) {   // construct_aabb, src/geometry.cpp:92-104
    PtAabb box;
    box.min.x = min_of(t.a.x, min_of(t.b.x, t.c.x));
    box.min.y = min_of(t.a.y, min_of(t.b.y, t.c.y));
    box.min.z = min_of(t.a.z, min_of(t.b.z, t.c.z));
    box.max.x = max_of(t.a.x, max_of(t.b.x, t.c.x));
    box.max.y = max_of(t.a.y, max_of(t.b.y, t.c.y));
    box.max.z = max_of(t.a.z, max_of(t.b.z, t.c.z));
    return box;
}

PtAabb box_of(const PtSphere& s) {     // construct_aabb, src/geometry.cpp:51-57
    const PtVec3 r{s.radius, s.radius, s.radius};
    return PtAabb{s.centre - r, s.centre + r};
}

namespace {

struct Job { int node, axis, begin, end; };

struct KeyedIndex { double key; int index; };

struct Build {
    const std::vector<PtAabb>& boxes;
    std::vector<int>& order;
    PtBvhNode* nodes;
    std::vector<KeyedIndex>& keyed;      // as long as `order`: a range is sorted in its own stretch of it

    // std::sort of order[begin, end) by the box minimum on `axis` (src/bvh.cpp:33-38).  The keys are copied next to
    // the indices first: the comparisons then read consecutive memory instead of boxes all over the array.  What
    // std::sort does depends only on the outcomes of the comparisons it makes, and those are the same, so the
    // permutation is the one the reference's call produces (equal keys included).
    void sort_range(int axis, int begin, int end) const {
        const PtAabb* b = boxes.data();
        KeyedIndex* first = keyed.data() + begin;
        KeyedIndex* last = keyed.data() + end;
        const int* from = order.data() + begin;
        for (KeyedIndex* k = first; k != last; ++k, ++from) {
            const PtVec3& m = b[*from].min;
            *k = KeyedIndex{axis == 0 ? m.x : (axis == 1 ? m.y : m.z), *from};
        }
        std::sort(first, last, [](const KeyedIndex& l, const KeyedIndex& r) { return l.key < r.key; });
        int* to = order.data() + begin;
        for (KeyedIndex* k = first; k != last; ++k, ++to) *to = k->index;
    }

    // Fills node `self` and its whole subtree; returns nothing, the box is read back from nodes[self].
    // With `deferred` set, subtrees smaller than `defer_below` are queued instead of built, and the two
    // halves of a node above them are handled by two threads down to `fork_levels` levels (each
    // std::sort call still sees exactly the sequence the serial build would give it, so ties fall the
    // same way).
    void fill(int self, int axis, int begin, int end, std::vector<Job>* deferred, int defer_below, int fork_levels = 0,
              std::mutex* deferred_lock = nullptr) const {
        PtBvhNode& node = nodes[self];
        const int count = end - begin;
        if (count == 1) {
            const int primitive = order[begin];
            node.aabb = boxes[primitive];
            node.index = primitive;
            node.left = node.right = 0;
            node.pad_ = 0;
            return;
        }
        if (deferred != nullptr && count < defer_below) {
            if (deferred_lock != nullptr) deferred_lock->lock();
            deferred->push_back(Job{self, axis, begin, end});
            if (deferred_lock != nullptr) deferred_lock->unlock();
            return;
        }
        sort_range(axis, begin, end);
        const int next_axis = (axis + 1) % 3;
        const int mid = begin + count / 2;
        const int left = self + 1;
        const int right = self + 2 * (mid - begin);
        if (deferred != nullptr && fork_levels > 0) {
            std::thread other([&]() { fill(left, next_axis, begin, mid, deferred, defer_below, fork_levels - 1, deferred_lock); });
            fill(right, next_axis, mid, end, deferred, defer_below, fork_levels - 1, deferred_lock);
            other.join();
        } else {
            fill(left, next_axis, begin, mid, deferred, defer_below, 0, deferred_lock);
            fill(right, next_axis, mid, end, deferred, defer_below, 0, deferred_lock);
        }
        node.index = 0;
        node.left = left;
        node.right = right;
        node.pad_ = 0;
        if (deferred == nullptr) join(self);
    }

    void join(int self) const {            // operator+=, src/geometry.cpp:39-49
        PtBvhNode& node = nodes[self];
        const PtAabb& l = nodes[node.left].aabb;
        const PtAabb& r = nodes[node.right].aabb;
        node.aabb.min = PtVec3{min_of(l.min.x, r.min.x), min_of(l.min.y, r.min.y), min_of(l.min.z, r.min.z)};
        node.aabb.max = PtVec3{max_of(l.max.x, r.max.x), max_of(l.max.y, r.max.y), max_of(l.max.z, r.max.z)};
    }

    // Boxes of the nodes above the deferred subtrees, children before parents.
    void join_above(int self, int count, int defer_below) const {
        if (count == 1 || count < defer_below) return;
        const int left_count = count / 2;
        join_above(nodes[self].left, left_count, defer_below);
        join_above(nodes[self].right, count - left_count, defer_below);
        join(self);
    }
};

}  // namespace

void build_bvh(const std::vector<PtAabb>& boxes, PtBvhNode* nodes, int threads) {
    const int count = static_cast<int>(boxes.size());
    if (count == 0) return;
    std::vector<int> order(count);
    for (int i = 0; i < count; ++i) order[i] = i;
    std::vector<KeyedIndex> keyed(count);
    Build build{boxes, order, nodes, keyed};
    if (threads <= 1 || count < 1 << 15) {
        build.fill(0, 0, 0, count, nullptr, 0);
        return;
    }
    // The top of the tree down to subtrees of < count/(4*threads) primitives (its first levels forked
    // over threads: the root's sort is the one serial step), then one queue of independent subtrees
    // shared by the workers.
    const int defer_below = std::max(2, count / (4 * threads));
    std::vector<Job> jobs;
    std::mutex jobs_lock;
    int fork_levels = 0;
    while ((1 << (fork_levels + 1)) <= threads) ++fork_levels;
    build.fill(0, 0, 0, count, &jobs, defer_below, fork_levels, &jobs_lock);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= jobs.size()) break;
            build.fill(jobs[i].node, jobs[i].axis, jobs[i].begin, jobs[i].end, nullptr, 0);
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
    for (std::thread& t : pool) t.join();
    build.join_above(0, count, defer_below);
}

}  // namespace pt

using namespace pt;

static int default_build_threads() {
    const unsigned n = std::thread::hardware_concurrency();
    return static_cast<int>(n == 0 ? 1 : (n > 16 ? 16 : n));
}

extern "C" {

int pt_construct_triangle_bvh(const PtTriangle* triangles, int32_t count, PtBvhNode* nodes) {
    if (triangles == nullptr || nodes == nullptr || count <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_construct_triangle_bvh: need count > 0 and non-null arrays");
    std::vector<PtAabb> boxes(count);
    for (int32_t i = 0; i < count; ++i) boxes[i] = box_of(triangles[i]);
    build_bvh(boxes, nodes, default_build_threads());
    return PT_OK;
}

int pt_construct_sphere_bvh(const PtSphere* spheres, int32_t count, PtBvhNode* nodes) {
    if (spheres == nullptr || nodes == nullptr || count <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_construct_sphere_bvh: need count > 0 and non-null arrays");
    std::vector<PtAabb> boxes(count);
    for (int32_t i = 0; i < count; ++i) boxes[i] = box_of(spheres[i]);
    build_bvh(boxes, nodes, default_build_threads());
    return PT_OK;
}

}  // extern "C"
