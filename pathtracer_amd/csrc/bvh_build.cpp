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
#include <condition_variable>
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

inline bool key_less(const KeyedIndex& l, const KeyedIndex& r) { return l.key < r.key; }

// std::sort(first, last, key_less) by several threads, with the permutation std::sort itself leaves (equal keys
// included).  libstdc++'s std::sort is an introsort: partition steps about a median-of-three pivot until a range is
// 16 long or the depth budget (2 log2 n) is spent (then a heap sort of that range), and one insertion-sort pass over
// everything at the end.  After a partition step the two sides never exchange elements again (left <= pivot <= right, and the
// final pass moves an element left only past strictly greater ones), so the sides can go to different threads, each
// finishing with the insertion pass over its own range.  The steps themselves are the library's own functions, called in
// the library's own order per range; with another standard library this is a plain std::sort.
class SortTeam {
public:
    static void sort(KeyedIndex* first, KeyedIndex* last, int threads) {
#if defined(__GLIBCXX__)
        if (threads > 1 && last - first > 2 * GRAIN && same_as_the_library()) {
            SortTeam team;
            team.tasks.push_back(Task{first, last, static_cast<long>(std::__lg(last - first)) * 2});
            std::vector<std::thread> others;
            for (int t = 1; t < threads; ++t) others.emplace_back([&team]() { team.work(); });
            team.work();
            for (std::thread& t : others) t.join();
            return;
        }
#endif
        (void)threads;
        std::sort(first, last, key_less);
    }

private:
    static constexpr long GRAIN = 1 << 14;       // ranges up to this long are finished by the thread that has them
#if defined(__GLIBCXX__)
    // The team's steps are the internals of this libstdc++'s std::sort.  Should another release change them (the
    // threshold, the pivot rule, the final pass), the team's permutation of equal keys could differ from std::sort's
    // without any error: so, once per process, both sort one array of 100,000 keys of which almost all tie, and the
    // team is used only if the two orders are the same.
    static bool same_as_the_library() {
        static const bool same = []() {
            const int n = 100000;
            std::vector<KeyedIndex> plain(n), shared(n);
            unsigned state = 12345u;
            for (int i = 0; i < n; ++i) {
                state = state * 1664525u + 1013904223u;
                plain[i] = shared[i] = KeyedIndex{static_cast<double>((state >> 16) % 37u), i};
            }
            std::sort(plain.begin(), plain.end(), key_less);
            SortTeam team;
            team.tasks.push_back(Task{shared.data(), shared.data() + n, static_cast<long>(std::__lg(n)) * 2});
            std::vector<std::thread> others;
            for (int t = 1; t < 3; ++t) others.emplace_back([&team]() { team.work(); });
            team.work();
            for (std::thread& t : others) t.join();
            for (int i = 0; i < n; ++i) if (plain[i].index != shared[i].index) return false;
            return true;
        }();
        return same;
    }
#endif
    struct Task { KeyedIndex* first; KeyedIndex* last; long depth; };
    std::mutex lock;
    std::condition_variable changed;
    std::vector<Task> tasks;
    int running = 0;

#if defined(__GLIBCXX__)
    void work() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> hold(lock);
                changed.wait(hold, [this]() { return !tasks.empty() || running == 0; });
                if (tasks.empty()) return;            // nothing queued and nobody left who could queue more
                t = tasks.back();
                tasks.pop_back();
                ++running;
            }
            auto less = __gnu_cxx::__ops::__iter_comp_iter(key_less);
            while (t.last - t.first > GRAIN && t.depth > 0) {
                --t.depth;
                KeyedIndex* cut = std::__unguarded_partition_pivot(t.first, t.last, less);
                {
                    std::lock_guard<std::mutex> hold(lock);
                    tasks.push_back(Task{cut, t.last, t.depth});
                }
                changed.notify_one();
                t.last = cut;
            }
            std::__introsort_loop(t.first, t.last, t.depth, less);
            std::__insertion_sort(t.first, t.last, less);
            {
                std::lock_guard<std::mutex> hold(lock);
                --running;
            }
            changed.notify_all();
        }
    }
#endif
};

struct Build {
    const std::vector<PtAabb>& boxes;
    std::vector<int>& order;
    PtBvhNode* nodes;
    std::vector<KeyedIndex>& keyed;      // as long as `order`: a range is sorted in its own stretch of it

    // std::sort of order[begin, end) by the box minimum on `axis` (src/bvh.cpp:33-38).  The keys are copied next to
    // the indices first: the comparisons then read consecutive memory instead of boxes all over the array.  What
    // std::sort does depends only on the outcomes of the comparisons it makes, and those are the same, so the
    // permutation is the one the reference's call produces (equal keys included).
    void sort_range(int axis, int begin, int end, int threads = 1) const {
        const PtAabb* b = boxes.data();
        KeyedIndex* keys = keyed.data();
        int* ids = order.data();
        const int team = (end - begin >= (1 << 16)) ? threads : 1;
        in_parts(begin, end, team, [=](long long from, long long to, int) {
            for (long long i = from; i < to; ++i) {
                const PtVec3& m = b[ids[i]].min;
                keys[i] = KeyedIndex{axis == 0 ? m.x : (axis == 1 ? m.y : m.z), ids[i]};
            }
        });
        SortTeam::sort(keys + begin, keys + end, team);
        in_parts(begin, end, team, [=](long long from, long long to, int) {
            for (long long i = from; i < to; ++i) ids[i] = keys[i].index;
        });
    }

    // Fills node `self` and its whole subtree.
    void fill(int self, int axis, int begin, int end) const {
        const int count = end - begin;
        if (count == 1) {
            set_leaf(self, order[begin]);
            return;
        }
        sort_range(axis, begin, end);
        const int next_axis = (axis + 1) % 3;
        const int mid = begin + count / 2;
        const int left = self + 1;
        const int right = self + 2 * (mid - begin);
        fill(left, next_axis, begin, mid);
        fill(right, next_axis, mid, end);
        set_inner(self, left, right);
        join(self);
    }

    void set_leaf(int self, int primitive) const {
        PtBvhNode& node = nodes[self];
        node.aabb = boxes[primitive];
        node.index = primitive;
        node.left = node.right = 0;
        node.pad_ = 0;
    }

    void set_inner(int self, int left, int right) const {
        PtBvhNode& node = nodes[self];
        node.index = 0;
        node.left = left;
        node.right = right;
        node.pad_ = 0;
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
    std::vector<KeyedIndex> keyed(count);
    Build build{boxes, order, nodes, keyed};
    if (threads <= 1 || count < 1 << 15) {
        for (int i = 0; i < count; ++i) order[i] = i;
        build.fill(0, 0, 0, count);
        return;
    }
    in_parts(0, count, threads, [&](long long from, long long to, int) {
        for (long long i = from; i < to; ++i) order[i] = static_cast<int>(i);
    });
    // The top of the tree level by level, down to subtrees of < count/(4*threads) primitives: the ranges of a level are
    // sorted side by side, each by its share of the threads (every sort is the std::sort call of the serial build on the
    // same sequence, so ties fall the same way).  Then one queue of independent subtrees shared by the threads.
    const int defer_below = std::max(2, count / (4 * threads));
    std::vector<Job> level{Job{0, 0, 0, count}}, next, jobs;
    StageClock clock;
    while (!level.empty()) {
        const int ranges = static_cast<int>(level.size());
        const int side_by_side = std::min(ranges, threads);
        in_parts(0, ranges, side_by_side, [&](long long from, long long to, int) {
            for (long long r = from; r < to; ++r) build.sort_range(level[r].axis, level[r].begin, level[r].end, std::max(1, threads / ranges));
        });
        next.clear();
        for (const Job& j : level) {
            const int mid = j.begin + (j.end - j.begin) / 2;
            const int left = j.node + 1, right = j.node + 2 * (mid - j.begin);
            build.set_inner(j.node, left, right);
            const Job halves[2] = {Job{left, (j.axis + 1) % 3, j.begin, mid}, Job{right, (j.axis + 1) % 3, mid, j.end}};
            for (const Job& h : halves) (h.end - h.begin < defer_below ? jobs : next).push_back(h);
        }
        level.swap(next);
    }
    clock.done("  upper levels sorted");
    std::atomic<size_t> next_job{0};
    in_parts(0, threads, threads, [&](long long, long long, int) {
        for (;;) {
            const size_t i = next_job.fetch_add(1);
            if (i >= jobs.size()) break;
            build.fill(jobs[i].node, jobs[i].axis, jobs[i].begin, jobs[i].end);
        }
    });
    clock.done("  subtrees built");
    build.join_above(0, count, defer_below);
}

bool sortable(const std::vector<PtAabb>& boxes) {
    for (const PtAabb& b : boxes) {
        if (b.min.x != b.min.x || b.min.y != b.min.y || b.min.z != b.min.z) return false;
    }
    return true;
}

}  // namespace pt

using namespace pt;

static int default_build_threads() { return host_threads(); }


extern "C" {

int pt_construct_triangle_bvh(const PtTriangle* triangles, int32_t count, PtBvhNode* nodes) {
    if (triangles == nullptr || nodes == nullptr || count <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_construct_triangle_bvh: need count > 0 and non-null arrays");
    std::vector<PtAabb> boxes(count);
    for (int32_t i = 0; i < count; ++i) boxes[i] = box_of(triangles[i]);
    if (!sortable(boxes)) return fail(PT_ERR_INVALID_ARGUMENT, "pt_construct_triangle_bvh: a triangle has a NaN coordinate (the reference sorts box minima with std::sort, which is undefined for unordered keys)");
    build_bvh(boxes, nodes, default_build_threads());
    return PT_OK;
}

int pt_construct_sphere_bvh(const PtSphere* spheres, int32_t count, PtBvhNode* nodes) {
    if (spheres == nullptr || nodes == nullptr || count <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_construct_sphere_bvh: need count > 0 and non-null arrays");
    std::vector<PtAabb> boxes(count);
    for (int32_t i = 0; i < count; ++i) boxes[i] = box_of(spheres[i]);
    if (!sortable(boxes)) return fail(PT_ERR_INVALID_ARGUMENT, "pt_construct_sphere_bvh: a sphere has a NaN centre or radius (the reference sorts box minima with std::sort, which is undefined for unordered keys)");
    build_bvh(boxes, nodes, default_build_threads());
    return PT_OK;
}

}  // extern "C"
