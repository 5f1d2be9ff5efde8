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
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pt_internal.h"
#include "pt_traverse_host.h"

namespace pt {

namespace {

constexpr int MAX_BINS = 64;
// Both are read from the environment once, when the library is loaded: threads that create scenes at the same time all
// see the same values.
// bins per axis of a split (PT_SAH_BINS overrides, at most MAX_BINS; 16 -> 32: 0.6 % fewer node steps and triangle tests)
const int BINS = []() { const char* v = std::getenv("PT_SAH_BINS"); return v != nullptr ? std::min(MAX_BINS, std::max(2, std::atoi(v))) : 32; }();
// a side of n leaves weighs n^COUNT_POWER in a split's cost (PT_SAH_POWER; 1 = the usual surface-area heuristic)
const double COUNT_POWER = []() { const char* v = std::getenv("PT_SAH_POWER"); return v != nullptr ? std::atof(v) : 1.0; }();
inline double weight(int n) { return COUNT_POWER == 1.0 ? static_cast<double>(n) : std::pow(static_cast<double>(n), COUNT_POWER); }

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

constexpr int SPARSE_MAXIMUM = 64;        // ranges up to this long are swept over their occupied bins only
constexpr int TEAM_MINIMUM = 1 << 18;     // ranges at least this long are binned and partitioned by several threads

struct Bins {
    PtAabb box[3][MAX_BINS];
    int count[3][MAX_BINS];
    Bins() {
        for (int a = 0; a < 3; ++a) for (int b = 0; b < BINS; ++b) { box[a][b] = empty_box(); count[a][b] = 0; }
    }
};

// A primitive as the builder moves it about: its box and its number.  The ranges of the recursion are contiguous
// stretches of one array of these, so that binning and partitioning read and write memory in order.
struct Item {
    PtAabb box;
    int id;
    int pad_;
};
inline PtVec3 centroid_of(const PtAabb& b) { return PtVec3{0.5 * (b.min.x + b.max.x), 0.5 * (b.min.y + b.max.y), 0.5 * (b.min.z + b.max.z)}; }

struct Builder {
    UnfilledVector<Item> items;
    UnfilledVector<Item> spare;          // partition buffer of the ranges that several threads split together
    ptd::PairNodeHost* pairs;
    int max_height;

    // Builds the subtree over items[begin, end) whose root pair (if any) is pairs[node];
    // `level` is the depth of that root (tree root = 1).  Returns the child reference, the box and the subtree's height.
    // `threads` = how many threads this subtree may use: for its own split while it is long, then shared out
    // between its two sides.  The tree does not depend on it: every split is a function of the range as a set.
    int build(int begin, int end, int node, int level, int threads, PtAabb* box_out, int* height_out) {
        const int count = end - begin;
        if (count == 1) {
            *box_out = items[begin].box;
            *height_out = 1;
            return ~items[begin].id;
        }
        const int team = (count >= TEAM_MINIMUM) ? threads : 1;
        PtAabb part_bounds[16];
        in_parts(begin, end, team, [&](long long lo, long long hi, int part) {
            PtAabb b = empty_box();
            for (long long i = lo; i < hi; ++i) {
                const PtVec3 c = centroid_of(items[i].box);
                grow(b, PtAabb{c, c});
            }
            part_bounds[part] = b;
        });
        PtAabb centroid_bounds = empty_box();
        for (int part = 0; part < team; ++part) grow(centroid_bounds, part_bounds[part]);
        const int mid = choose_split(begin, end, level, centroid_bounds, team);
        const int left_count = mid - begin;
        PtAabb left_box, right_box;
        int left_ref, right_ref, left_height = 0, right_height = 0;
        if (threads > 1 && count > 65536) {
            const int left_threads = std::max(1, std::min(threads - 1, static_cast<int>(static_cast<long long>(threads) * left_count / count)));
            std::thread other([&]() { left_ref = build(begin, mid, node + 1, level + 1, left_threads, &left_box, &left_height); });
            right_ref = build(mid, end, node + left_count, level + 1, threads - left_threads, &right_box, &right_height);
            other.join();
        } else {
            left_ref = build(begin, mid, node + 1, level + 1, 1, &left_box, &left_height);
            right_ref = build(mid, end, node + left_count, level + 1, 1, &right_box, &right_height);
        }
        ptd::PairNodeHost& p = pairs[node];
        p.lbox = left_box;
        p.rbox = right_box;
        p.left = left_ref;
        p.right = right_ref;
        for (int32_t& unused : p.pad_) unused = 0;
        *box_out = left_box;
        grow(*box_out, right_box);
        *height_out = 1 + std::max(left_height, right_height);
        return node;
    }

    static int bin_of(double centroid, double lo, double scale) {
        const int b = static_cast<int>((centroid - lo) * scale);
        return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
    }

    // Partitions items[begin, end) and returns the split position (begin < mid < end).
    int choose_split(int begin, int end, int level, const PtAabb& cb, int team) {
        const int count = end - begin;
        const int levels_left = max_height - level;     // levels available below this node's children level
        // a side of m leaves needs levels_for(m) levels starting at level + 1
        auto fits = [&](int m) { return levels_for(m) <= levels_left; };

        int best_axis = -1, best_bin = -1;
        double best_cost = INFINITY;
        double lo[3], scale[3];
        bool open[3];
        for (int axis = 0; axis < 3; ++axis) {
            lo[axis] = axis_of(cb.min, axis);
            const double hi = axis_of(cb.max, axis);
            open[axis] = hi > lo[axis];
            scale[axis] = open[axis] ? BINS / (hi - lo[axis]) : 0.0;
        }
        if (count > 2 && count <= SPARSE_MAXIMUM) {
            // A short range fills few of the bins.  Only a bin that holds something can be a new best split: at an empty
            // bin the two sides are those of the bin before it, and an equal cost never replaces the best.  So the items
            // are ordered by bin and the sweep visits the occupied bins only; sides, unions, areas and costs are the ones
            // the dense sweep below forms (unions of boxes do not depend on the order they are formed in), at a cost
            // that grows with the range instead of with the number of bins.
            const Item* item = items.data() + begin;
            for (int axis = 0; axis < 3; ++axis) {
                if (!open[axis]) continue;
                int bin[SPARSE_MAXIMUM], by_bin[SPARSE_MAXIMUM];
                double right_area[SPARSE_MAXIMUM + 1];
                for (int i = 0; i < count; ++i) {
                    bin[i] = bin_of(0.5 * (axis_of(item[i].box.min, axis) + axis_of(item[i].box.max, axis)), lo[axis], scale[axis]);
                    int at = i;
                    while (at > 0 && bin[by_bin[at - 1]] > bin[i]) { by_bin[at] = by_bin[at - 1]; --at; }
                    by_bin[at] = i;
                }
                PtAabb acc = empty_box();
                for (int k = count - 1; k >= 0; --k) {
                    grow(acc, item[by_bin[k]].box);
                    right_area[k] = half_area(acc);
                }
                acc = empty_box();
                int k = 0;
                while (k < count) {
                    const int b = bin[by_bin[k]];
                    while (k < count && bin[by_bin[k]] == b) grow(acc, item[by_bin[k++]].box);
                    const int n = k, nr = count - k;
                    if (nr == 0 || b >= BINS - 1 || !fits(n) || !fits(nr)) continue;
                    const double cost = half_area(acc) * weight(n) + right_area[k] * weight(nr);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        } else if (count > 2) {
            // all three axes in one pass over the range (bin boxes and counts are unions and sums: whoever
            // forms them, in whatever order, gets the same)
            std::vector<Bins> part_bins(team);
            in_parts(begin, end, team, [&](long long from, long long to, int part) {
                Bins& mine = part_bins[part];
                for (long long i = from; i < to; ++i) {
                    const PtAabb& box = items[i].box;
                    const PtVec3 c = centroid_of(box);
                    const int bx = bin_of(c.x, lo[0], scale[0]), by = bin_of(c.y, lo[1], scale[1]), bz = bin_of(c.z, lo[2], scale[2]);
                    ++mine.count[0][bx]; grow(mine.box[0][bx], box);
                    ++mine.count[1][by]; grow(mine.box[1][by], box);
                    ++mine.count[2][bz]; grow(mine.box[2][bz], box);
                }
            });
            Bins& all = part_bins[0];
            for (int part = 1; part < team; ++part) {
                for (int a = 0; a < 3; ++a) for (int b = 0; b < BINS; ++b) {
                    all.count[a][b] += part_bins[part].count[a][b];
                    grow(all.box[a][b], part_bins[part].box[a][b]);
                }
            }
            for (int axis = 0; axis < 3; ++axis) {
                if (!open[axis]) continue;
                const PtAabb* bin_box = all.box[axis];
                const int* bin_count = all.count[axis];
                double right_area[MAX_BINS];
                int right_count[MAX_BINS];
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
                    const double cost = half_area(acc) * weight(n) + right_area[b + 1] * weight(nr);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }
        Item* first = items.data() + begin;
        Item* last = items.data() + end;
        if (best_axis >= 0) {
            const double axis_lo = lo[best_axis], axis_scale = scale[best_axis];
            const int axis = best_axis;
            auto goes_left = [=](const Item& item) {
                const double c = 0.5 * (axis_of(item.box.min, axis) + axis_of(item.box.max, axis));
                return bin_of(c, axis_lo, axis_scale) <= best_bin;
            };
            if (team <= 1) {
                Item* middle = std::partition(first, last, goes_left);
                return begin + static_cast<int>(middle - first);
            }
            // several threads: each counts its piece's left side, then copies its piece to its places in `spare`
            std::vector<int> lefts(team + 1, 0);
            in_parts(begin, end, team, [&](long long from, long long to, int part) {
                int n = 0;
                for (long long i = from; i < to; ++i) n += goes_left(items[i]) ? 1 : 0;
                lefts[part + 1] = n;
            });
            for (int part = 0; part < team; ++part) lefts[part + 1] += lefts[part];
            const int total_left = lefts[team];
            in_parts(begin, end, team, [&](long long from, long long to, int part) {
                long long l = begin + lefts[part];
                long long r = begin + total_left + (from - begin) - lefts[part];
                for (long long i = from; i < to; ++i) {
                    if (goes_left(items[i])) spare[l++] = items[i]; else spare[r++] = items[i];
                }
            });
            in_parts(begin, end, team, [&](long long from, long long to, int) { std::copy(spare.begin() + from, spare.begin() + to, items.begin() + from); });
            return begin + total_left;
        }
        // no acceptable binned split (coincident centroids, or the height cap): halve on the widest axis
        int axis = 0;
        const double ex = cb.max.x - cb.min.x, ey = cb.max.y - cb.min.y, ez = cb.max.z - cb.min.z;
        if (ey > ex && ey >= ez) axis = 1; else if (ez > ex && ez > ey) axis = 2;
        Item* middle = first + count / 2;
        std::nth_element(first, middle, last, [&](const Item& l, const Item& r) {
            const double a = 0.5 * (axis_of(l.box.min, axis) + axis_of(l.box.max, axis)), b = 0.5 * (axis_of(r.box.min, axis) + axis_of(r.box.max, axis));
            return a < b || (a == b && l.id < r.id);
        });
        return begin + count / 2;
    }
};

}  // namespace

int host_threads() {
    if (const char* v = std::getenv("PT_BUILD_THREADS")) return std::min(16, std::max(1, std::atoi(v)));
    return static_cast<int>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
}

int build_sah_pairs(const PtAabb* boxes, int count, int max_height, UnfilledVector<ptd::PairNodeHost>* pairs, int* root_ref,
                    int* height_out) {
    if (count <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "build_sah_pairs: no boxes");
    if (levels_for(count) > max_height) return fail(PT_ERR_UNSUPPORTED, "%d leaves cannot fit a tree of height %d", count, max_height);
    pairs->resize(count > 1 ? count - 1 : 0);      // every pair is written in full by build()
    Builder b{{}, {}, pairs->data(), max_height};
    b.items.resize(count);
    if (count >= TEAM_MINIMUM) b.spare.resize(count);
    const int threads = host_threads();
    in_parts(0, count, count >= TEAM_MINIMUM ? threads : 1, [&](long long from, long long to, int) {
        for (long long i = from; i < to; ++i) b.items[i] = Item{boxes[i], static_cast<int>(i), 0};
    });
    PtAabb root_box;
    int height = 1;     // root pair = level 1, its leaf children = level 2, ...; a lone leaf counts as 1
    *root_ref = b.build(0, count, 0, 1, threads, &root_box, &height);
    *height_out = height;
    return PT_OK;
}

namespace {

float outward_down(double x) { const float f = static_cast<float>(x); return (static_cast<double>(f) > x) ? std::nextafterf(f, -INFINITY) : f; }
float outward_up(double x) { const float f = static_cast<float>(x); return (static_cast<double>(f) < x) ? std::nextafterf(f, INFINITY) : f; }
const double* bounds(const PtAabb& b) { return &b.min.x; }
double half_area_of(const double* b) {
    const double dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}

// A subtree left for a worker thread: its pair, the entries waiting above it, and where its root's index goes.
struct QuadTask { int pair; int pending; int parent; int slot; };

struct Collapser {
    const ptd::PairNodeHost* pairs;
    const std::vector<int>* leaves_below;     // per pair; null while a worker collapses its whole subtree
    int task_leaves;                          // subtrees with at most this many leaves become tasks
    std::vector<QuadTask>* tasks;

    // Appends the four-wide node of `pair` (and, recursively, of its inner children) to `out`; returns its index there.
    // `pending` = stack entries that can be waiting when the node is entered (each ancestor leaves at most its other
    // children behind).
    int collapse(int pair, int pending, ptd::QuadHierarchy* out) const {
        int ref[4];
        const double* box[4];
        int n = 2;
        ref[0] = pairs[pair].left;  box[0] = bounds(pairs[pair].lbox);
        ref[1] = pairs[pair].right; box[1] = bounds(pairs[pair].rbox);
        while (n < 4) {
            int widest = -1;
            for (int i = 0; i < n; ++i) {
                if (ref[i] >= 0 && (widest < 0 || half_area_of(box[i]) > half_area_of(box[widest]))) widest = i;
            }
            if (widest < 0) break;
            const ptd::PairNodeHost& inner = pairs[ref[widest]];
            ref[widest] = inner.left;  box[widest] = bounds(inner.lbox);
            ref[n] = inner.right;      box[n] = bounds(inner.rbox);
            ++n;
        }
        const int index = static_cast<int>(out->nodes.size());
        out->nodes.emplace_back();
        out->exact.emplace_back();
        out->deepest_pending = std::max(out->deepest_pending, pending + n - 1);   // entries left behind while a child of this node is searched
        {
            ptd::QuadNode32Host& q = out->nodes[index];
            ptd::QuadBoxes64Host& e = out->exact[index];
            for (int i = 0; i < 4; ++i) {
                for (int k = 0; k < 3; ++k) {
                    // an empty slot is all NaN: the kernel's entry test is false for NaN bounds
                    q.plane[k][i] = (i < n) ? outward_down(box[i][k]) : std::nanf("");
                    q.plane[3 + k][i] = (i < n) ? outward_up(box[i][3 + k]) : std::nanf("");
                    e.box[i][k] = (i < n) ? box[i][k] : std::nan("");
                    e.box[i][3 + k] = (i < n) ? box[i][3 + k] : std::nan("");
                }
                q.child[i] = (i < n) ? ref[i] : 0;
                q.pad_[i] = 0;
            }
        }
        for (int i = 0; i < n; ++i) {
            if (ref[i] < 0) continue;
            if (leaves_below != nullptr && (*leaves_below)[ref[i]] <= task_leaves) {
                tasks->push_back(QuadTask{ref[i], pending + n - 1, index, i});
                continue;
            }
            const int child = collapse(ref[i], pending + n - 1, out);
            out->nodes[index].child[i] = child;      // the vector may have grown: index, not a reference
        }
        return index;
    }
};

float float_of_bits(uint32_t bits) { float f; std::memcpy(&f, &bits, sizeof f); return f; }

// The float nearest to `plane` on its outer side (at or below a lower bound, at or above an upper bound) whose lowest
// mantissa byte is `byte`: at most 255 units in the last place away, 2^-15 relative.  NaN (an empty slot) stays NaN.
float carrying(float plane, uint32_t byte, bool lower_bound) {
    if (plane != plane) return float_of_bits(0x7fc00000u | byte);
    uint32_t bits;
    std::memcpy(&bits, &plane, sizeof bits);
    const bool negative = (bits >> 31) != 0;
    const uint32_t sign = bits & 0x80000000u, magnitude = bits & 0x7fffffffu;
    uint32_t moved = (magnitude & ~0xffu) | byte;
    if (negative == lower_bound) {                   // outward is away from zero
        if (moved < magnitude) moved += 0x100u;
        return float_of_bits(sign | moved);
    }
    if (moved > magnitude) {                         // outward is towards zero, and past it if the magnitude runs out
        if (moved < 0x100u) return float_of_bits((sign ^ 0x80000000u) | byte);
        moved -= 0x100u;
    }
    return float_of_bits(sign | moved);
}

// The kernel reads a node's child references out of its planes instead of fetching them: byte k of child i's reference
// is the lowest mantissa byte of that child's plane in row CARRIER_ROW[k] (lower bound of x, of y, upper bound of x, of y),
// each plane moved outward to the nearest float that carries its byte.  The references stay in `child` as well, for the
// host, the cross-checking build and the kernel that fetches them.
constexpr int CARRIER_ROW[4] = {0, 1, 3, 4};
void carry_child_references(ptd::QuadNode32Host* nodes, size_t count, int threads) {
    in_parts(0, static_cast<long long>(count), count >= (size_t(1) << 16) ? threads : 1, [&](long long from, long long to, int) {
        for (long long n = from; n < to; ++n) {
            ptd::QuadNode32Host& q = nodes[n];
            for (int i = 0; i < 4; ++i) {
                const uint32_t reference = static_cast<uint32_t>(q.child[i]);
                for (int k = 0; k < 4; ++k) {
                    const int row = CARRIER_ROW[k];
                    q.plane[row][i] = carrying(q.plane[row][i], (reference >> (8 * k)) & 0xffu, row < 3);
                }
            }
        }
    });
}

}  // namespace

void collapse_to_quads(const ptd::PairNodeHost* pairs, size_t pair_count, ptd::QuadHierarchy* out) {
    out->nodes.clear();
    out->exact.clear();
    out->deepest_pending = 0;
    if (pair_count == 0) return;
    const int threads = host_threads();
    if (pair_count < (size_t(1) << 16)) {
        Collapser{pairs, nullptr, 0, nullptr}.collapse(0, 0, out);
        carry_child_references(out->nodes.data(), out->nodes.size(), 1);
        return;
    }
    StageClock clock;
    // leaves below every pair: children are numbered after their parent
    std::vector<int> leaves_below(pair_count);
    for (size_t i = pair_count; i-- > 0;) {
        const ptd::PairNodeHost& p = pairs[i];
        leaves_below[i] = (p.left < 0 ? 1 : leaves_below[p.left]) + (p.right < 0 ? 1 : leaves_below[p.right]);
    }
    // The top of the tree here; subtrees of at most 1/64 of the leaves are left as tasks, collapsed by the threads
    // into arrays of their own (indices from 0) and appended in the order they were met, so that the layout is the
    // same whatever the number of threads.
    clock.done("  leaves counted");
    std::vector<QuadTask> tasks;
    Collapser{pairs, &leaves_below, std::max(1024, static_cast<int>((pair_count + 1) / 64)), &tasks}.collapse(0, 0, out);
    std::vector<ptd::QuadHierarchy> parts(tasks.size());
    in_parts(0, static_cast<int>(tasks.size()), std::min<int>(threads, static_cast<int>(tasks.size())), [&](int, int, int part) {
        // tasks are dealt round-robin: neighbouring subtrees differ in size less than far ones
        const int stride = std::min<int>(threads, static_cast<int>(tasks.size()));
        for (size_t t = part; t < tasks.size(); t += stride) Collapser{pairs, nullptr, 0, nullptr}.collapse(tasks[t].pair, tasks[t].pending, &parts[t]);
    });
    clock.done("  subtrees collapsed");
    std::vector<size_t> first(tasks.size() + 1, out->nodes.size());
    for (size_t t = 0; t < tasks.size(); ++t) first[t + 1] = first[t] + parts[t].nodes.size();
    out->nodes.resize(first[tasks.size()]);
    out->exact.resize(first[tasks.size()]);
    for (size_t t = 0; t < tasks.size(); ++t) {
        out->nodes[tasks[t].parent].child[tasks[t].slot] = static_cast<int32_t>(first[t]);
        out->deepest_pending = std::max(out->deepest_pending, parts[t].deepest_pending);
    }
    in_parts(0, static_cast<int>(tasks.size()), std::min<int>(threads, static_cast<int>(tasks.size())), [&](int, int, int part) {
        const int stride = std::min<int>(threads, static_cast<int>(tasks.size()));
        for (size_t t = part; t < tasks.size(); t += stride) {
            const int32_t offset = static_cast<int32_t>(first[t]);
            for (size_t i = 0; i < parts[t].nodes.size(); ++i) {
                ptd::QuadNode32Host q = parts[t].nodes[i];
                for (int32_t& c : q.child) c = (c > 0) ? c + offset : c;      // a subtree's own root (0) is never a child
                out->nodes[first[t] + i] = q;
                out->exact[first[t] + i] = parts[t].exact[i];
            }
        }
    });
    clock.done("  subtrees put in place");
    carry_child_references(out->nodes.data(), out->nodes.size(), threads);
    clock.done("  child references put into the planes");
}

}  // namespace pt
