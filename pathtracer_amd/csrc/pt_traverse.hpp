// HBM scene layout and the per-lane tree searches shared by every kernel.
#ifndef PT_TRAVERSE_HPP
#define PT_TRAVERSE_HPP

#include "pt_device.hpp"

namespace ptd {

// PairNode (128 B, one per inner node of the reference tree): both children's boxes and
// both child references, so one aligned 128-byte fetch decides the two subtrees and lets
// the nearer one be entered first.  A child reference >= 0 is a PairNode index; < 0 is
// ~primitive_index of a leaf.  The root node's own box lives in the TreeRef.
struct PairNode {
    double lbox[6];
    double rbox[6];
    int32_t left, right;
    int32_t pad_[6];
};
static_assert(sizeof(PairNode) == 128, "PairNode is one 128-byte line");

struct TreeRef {
    const PairNode* pairs;
    const PtBvhNode* ref_nodes;   // the reference's own node array, for the exhaustive path
    int32_t root;                 // child-reference encoding; `count == 0` means "no tree"
    int32_t count;
    int32_t nested;               // every box has min <= max and every inner box contains its children's: only then may a
                                  // subtree be skipped because its box is entered beyond the best hit.  The reference's own
                                  // build gives such a tree unless a sphere is hollow (negative radius: inverted box,
                                  // src/geometry.cpp:51-57); a caller's tree is taken as it is.
    int32_t bounded;              // no box holds an infinity or a NaN: only then is the short form of the slab test
                                  // (hardware min/max) sure to meet no NaN product and to equal the reference's selections
    double root_box[6];
};

// ---------------------------------------------------------------------------------------------
// Single-precision inner nodes (128 B, four children) for the wavefront kernel.
//
// For a ray whose reciprocals are finite, a box that lies inside another box passes the
// reference's slab test only if the outer box passes it too, and its entry distance is no
// smaller (rounded subtraction and rounded multiplication by a fixed factor are monotone).
// So which leaves the reference reaches depends only on each leaf's OWN box, and inner boxes
// serve purely to skip subtrees.  They may therefore be tested conservatively in f32, as long
// as no subtree is skipped that holds a leaf the exact test would accept within the pruning
// distance; the exact f64 test of the leaf's own box decides (LeafBox).
//
// Conservative test: boxes are stored rounded outward.  Per axis the entry distance is
// (near plane - origin) * reciprocal = near plane * reciprocal - origin * reciprocal, and the f32 code
// evaluates the right-hand form with one fused multiply-add: the product is exact inside it, and
// `origin * reciprocal` (taken with the f32 reciprocal, so that the reciprocal's rounding is a
// common factor of the whole difference and nothing cancels) enters as a float at or ABOVE its value
// for entry distances and at or BELOW it for exit distances.  Before its single rounding the f32 slab
// interval therefore contains the exact one scaled by (1 + e), |e| <= 2^-24; that factor and the
// rounding of the fused operation are covered by widening the interval by RHO = 2^-20 relative
// plus EPS32 absolute (denormal range).  Which plane of a box is the near one follows from the
// reciprocal's sign, so it is chosen when the plane is loaded and costs no min/max.
//
// Four children per node: a node of the binary hierarchy with its children (and, while slots are
// free, the largest grandchildren) pulled up.  plane[k] holds the k-th bound (min x, y, z, max x, y,
// z) of the four child boxes side by side; an empty slot holds NaN and is never entered.
struct alignas(128) QuadNode32 {
    float plane[6][4];
    int32_t child[4];      // >= 0: quad index, < 0: ~primitive (the planes of x and y carry them a second time: see slab32_quad)
    int32_t pad_[4];
};
static_assert(sizeof(QuadNode32) == 128, "QuadNode32 is one 128-byte line");
struct QuadBoxes64 { double box[4][6]; };   // the exact child boxes, read only by the cross-checking build

// A leaf's own box exactly as the reference's BVH stores it: the reference tests it before the
// triangle (src/path_tracing.cpp:113-127).  The wavefront kernel tests the triangle alone and then
// checks the box of the one triangle that won, when the path is shaded.
struct alignas(16) LeafBox { double box[6]; };
// The 48 bytes as three aligned 16-byte loads.
typedef double Double2 __attribute__((ext_vector_type(2)));
PT_DEV LeafBox fetch_leaf_box(const LeafBox* __restrict__ b) {
    const Double2* v = reinterpret_cast<const Double2*>(b);
    const Double2 lo = v[0], mid = v[1], hi = v[2];
    return LeafBox{{lo.x, lo.y, mid.x, mid.y, hi.x, hi.y}};
}

constexpr float RHO32 = 9.5367431640625e-07f;    // 2^-20
constexpr float EPS32 = 1.0e-30f;
constexpr double MAX_COORD32 = 1048576.0;        // |coordinate| <= 2^20 keeps every f32 product finite
constexpr double MAX_INV32 = 1.2676506002282294e30;   // 2^100

struct Ray32 {
    float cnx, cny, cnz;     // a float at or above origin * reciprocal: taken off the near plane's product
    float cfx, cfy, cfz;     // a float at or below it: taken off the far plane's product
    float ix, iy, iz;        // reciprocals, rounded to nearest
    unsigned kx, ky, kz;     // bytes from a node's lower plane of the axis to its near plane: 48 if the reciprocal is negative
};

// A float >= x and a float <= x for |x| <= 2^121: the nearest float moved outward by |f| * 2^-23, which
// is at least one unit in its last place (so it passes x, at most half a unit away), plus 1e-37 for
// the denormal range, where that product vanishes.
PT_DEV void bracket32(double x, float& above, float& below) {
    const float f = (float)x;
    above = fmaf(fabsf(f), 1.1920928955078125e-07f, f) + 1.0e-37f;
    below = fmaf(fabsf(f), -1.1920928955078125e-07f, f) - 1.0e-37f;
}

// True if the ray may use the f32 inner nodes (otherwise it takes the exact traversal).
PT_DEV bool ray32_allowed(const Ray& ray, D3 inv) {
    return fabs(inv.x) <= MAX_INV32 && fabs(inv.y) <= MAX_INV32 && fabs(inv.z) <= MAX_INV32 &&
           fabs(ray.o.x) <= MAX_COORD32 && fabs(ray.o.y) <= MAX_COORD32 && fabs(ray.o.z) <= MAX_COORD32;
}
PT_DEV Ray32 make_ray32(const Ray& ray, D3 inv) {
    Ray32 r;
    r.ix = (float)inv.x;
    r.iy = (float)inv.y;
    r.iz = (float)inv.z;
    // |origin| <= 2^20 and |reciprocal| <= 2^100: the products (rounded once more, in f64) stay far inside the f32 range
    bracket32(ray.o.x * (double)r.ix, r.cnx, r.cfx);
    bracket32(ray.o.y * (double)r.iy, r.cny, r.cfy);
    bracket32(ray.o.z * (double)r.iz, r.cnz, r.cfz);
    r.kx = (inv.x < 0.0f) ? 48u : 0u;
    r.ky = (inv.y < 0.0f) ? 48u : 0u;
    r.kz = (inv.z < 0.0f) ? 48u : 0u;
    return r;
}
// Upper bound, as a float, of the pruning distance.
PT_DEV float cut32_of(double cut) { return (float)((cut + 1.0e-30) * (1.0 + 4.76837158203125e-07)); }
// The same for the pruning distance that follows a hit at `distance` >= 0 (prune_cut: distance * (1 + 1e-7) + 1e-7), in two
// single-precision instructions: the conversion loses at most 2^-24 relative, the constant is 1 + 7 * 2^-23 = 1 + 8.3e-7 as
// a float, the fused operation rounds once more, so the result is at least distance * (1 + 5.9e-7) + 1.09e-7.  Any float at
// or above the pruning distance serves (a larger one only prunes less).
PT_DEV float cut32_after_hit(double distance) { return fmaf((float)distance, 1.0f + 8.0e-7f, 1.1e-7f); }

// Conservative bounds of the four child boxes at once, on 4-wide vectors: `lower` <= exact entry
// distance, `upper` >= exact exit distance.
typedef float Float4 __attribute__((ext_vector_type(4)));
typedef int Int4 __attribute__((ext_vector_type(4)));
typedef float Float2 __attribute__((ext_vector_type(2)));
PT_DEV Float4 fma4(Float4 p, float i, float c) {       // two packed fused multiply-adds
    const Float2 lo = __builtin_elementwise_fma(Float2{p.x, p.y}, Float2{i, i}, Float2{c, c});
    const Float2 hi = __builtin_elementwise_fma(Float2{p.z, p.w}, Float2{i, i}, Float2{c, c});
    return Float4{lo.x, lo.y, hi.x, hi.y};
}
PT_DEV Float4 max4_3(Float4 a, Float4 b, Float4 c) {
    return Float4{fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z), fmaxf(fmaxf(a.w, b.w), c.w)};
}
PT_DEV Float4 min4_3(Float4 a, Float4 b, Float4 c) {
    return Float4{fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z), fminf(fminf(a.w, b.w), c.w)};
}
// Node `node` of `nodes`: every field is addressed as the array's base plus a 32-bit byte offset (the
// host keeps the array below 4 GB), the form the hardware adds by itself.
// CARRIED: the child references are put together from the planes (six 16-byte pieces per visit); otherwise they are
// fetched as the node's seventh piece.  The first is what a kernel bound by the vector L1's access rate wants (the triangle
// scenes), the second what one bound by vector issue wants (triangles and spheres in one hierarchy): 16 instructions
// against one piece.
template <bool CARRIED>
PT_DEV void slab32_quad(const Ray32 r, const QuadNode32* __restrict__ nodes, int node, Float4& lower, Float4& upper, Int4& children) {
    const char* base = reinterpret_cast<const char*>(nodes);
    const unsigned at = (unsigned)node << 7;
    auto field = [&](unsigned offset, int more) { return *reinterpret_cast<const Float4*>(base + (size_t)offset + more); };
    // every load is issued before the first result is used; the far plane lies 48 bytes from the near one, on the other
    // side of it (k ^ 48 is 48 - k for k = 0 and k = 48, and an exclusive-or and an addition are one instruction)
    const Float4 px0 = field(at + r.kx, 0), px1 = field(at + (r.kx ^ 48u), 0);
    const Float4 py0 = field(at + r.ky, 16), py1 = field(at + (r.ky ^ 48u), 16);
    const Float4 pz0 = field(at + r.kz, 32), pz1 = field(at + (r.kz ^ 48u), 32);
    // CARRIED: the child references ride in the planes the step reads anyway.  Byte k of child i's reference is the lowest
    // mantissa byte of that child's plane in row {0, 1, 3, 4}[k] (x lower, y lower, x upper, y upper); the host has moved each
    // of those planes outward to the nearest float that carries its byte (at most 2^-15 relative), so the planes are used as
    // they are.  v_perm_b32 picks each byte of its result out of the eight bytes of its two sources (4..7: the first, 0..3:
    // the second; 12: zero): bytes 0 and 2 come from the two x planes, bytes 1 and 3 from the two y planes.  Which of the
    // two x (y) rows a lane has in `px0` (`py0`) depends on its ray's direction, and the two selectors that go with the two
    // cases are each other turned by 16 bits, which is what the offset k (0 or 48) is as a five-bit shift count.
    if (CARRIED) {
        const unsigned from_x = __builtin_amdgcn_alignbit(0x0c040c00u, 0x0c040c00u, r.kx);
        const unsigned from_y = __builtin_amdgcn_alignbit(0x040c000cu, 0x040c000cu, r.ky);
        auto reference = [&](float x_first, float x_second, float y_first, float y_second) {
            return (int)(__builtin_amdgcn_perm(__float_as_uint(x_second), __float_as_uint(x_first), from_x) |
                         __builtin_amdgcn_perm(__float_as_uint(y_second), __float_as_uint(y_first), from_y));
        };
        children = Int4{reference(px0.x, px1.x, py0.x, py1.x), reference(px0.y, px1.y, py0.y, py1.y),
                        reference(px0.z, px1.z, py0.z, py1.z), reference(px0.w, px1.w, py0.w, py1.w)};
    } else {
        children = *reinterpret_cast<const Int4*>(base + (size_t)at + 96);
    }
    const Float4 x0 = fma4(px0, r.ix, -r.cnx), x1 = fma4(px1, r.ix, -r.cfx);
    const Float4 y0 = fma4(py0, r.iy, -r.cny), y1 = fma4(py1, r.iy, -r.cfy);
    const Float4 z0 = fma4(pz0, r.iz, -r.cnz), z1 = fma4(pz1, r.iz, -r.cfz);
    // an empty slot is NaN on every axis, so it stays NaN through max/min
    const Float4 tmin = max4_3(x0, y0, z0);
    const Float4 tmax = min4_3(x1, y1, z1);
    // Widening.  A non-negative entry distance is lowered by RHO relative; a negative one passes both of
    // its comparisons whatever its size, so it needs none.  The exit distance matters only when the exact
    // one is >= 0, and then the f32 one is >= 0 too, up to an error in the denormal range: it is raised by
    // RHO relative, and EPS32 is added once here instead of in every comparison (`upper` includes it).
    lower = tmin * (1.0f - RHO32);
    upper = fma4(tmax, 1.0f + RHO32, EPS32);
}
// May the child be entered?  True for every box the exact test accepts within `cut`; false for NaN
// bounds (empty slots).  `upper` is slab32_quad's: EPS32 above the widened exit distance.
PT_DEV bool enter32(float lower, float upper, float cut32) {
    return (upper >= 0.0f) & (lower <= upper) & (lower <= cut32);
}
struct DeviceScene {
    TreeRef tri_tree;
    const TriRecord* tris;
    // The wavefront kernel's search hierarchy: 4-wide single-precision nodes over the triangles and, when
    // the scene has more than one sphere, over the spheres too (one search finds both nearest hits).
    // Primitive p of it is triangle p if p < sphere_base, else sphere p - sphere_base.
    const QuadNode32* tri_quads;     // null if the scene is out of f32 range
    const QuadBoxes64* tri_quad_boxes;
    const LeafBox* tri_boxes;        // own box of every primitive of the hierarchy
    int32_t sphere_base;             // INT_MAX when the spheres are not part of the hierarchy
    int32_t search_root;             // child reference of the hierarchy's root (0, or ~primitive if there is one primitive)
    int32_t search_count;            // primitives in the hierarchy
    int32_t with_metals;             // some material is a metal (scatter then treats Lambertian and metal lanes together)
    double search_box[6];            // exact box of everything in it
    TreeRef sph_tree;
    const double* spheres;        // 4 doubles each
    const int32_t* sph_material;
    const PtMaterial* materials;
    D3 bg0, bg1;
};

struct Hit { int index; double distance; };

struct TriLeaf {
    const TriRecord* tris;
    PT_DEV bool test(const Ray& ray, int primitive, double& d) const { return hit_triangle(ray, tris + primitive, d); }
};
struct SphLeaf {
    const double* spheres;
    PT_DEV bool test(const Ray& ray, int primitive, double& d) const { return hit_sphere(ray, spheres + 4 * primitive, d); }
};

// Margin added to the best distance before a subtree is skipped: far above any rounding
// disagreement between a box-entry distance and a primitive distance, far below anything
// that would cost visits.
PT_DEV double prune_cut(double limit) { return limit + (1.0e-7 * fabs(limit) + 1.0e-7); }

// The reference's traversal (src/path_tracing.cpp:111-169) evaluated without recursion:
// every node whose box passes is entered, both children always, and an inner node keeps the
// strictly smaller child distance; equal distances (two misses included) give a miss.
// One lane of a wave runs this at a time, with its frames in that wave's LDS slice.
struct TourFrame { int right; int state; int left_index; double left_distance; };

// Arguments by value: they travel in registers, so that no caller has to keep its ray in memory for this call's sake.
template <class Leaf>
__device__ __noinline__ Hit tournament(const Ray ray, const TreeRef tree, const Leaf leaf, TourFrame* frames) {
    const D3 inv{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
    Hit result{-1, REAL_MAX};
    int depth = 0;
    int node = 0;
    for (;;) {
        // ---- evaluate `node` downwards until a result exists
        for (;;) {
            const PtBvhNode* n = tree.ref_nodes + node;
            double tmin, tmax;
            slab_exact(ray.o, inv, &n->aabb.min.x, tmin, tmax);
            result = Hit{-1, REAL_MAX};
            if (tmax < 0.0f || tmin > tmax) break;
            if (n->left == 0 && n->right == 0) {
                double d;
                if (leaf.test(ray, n->index, d)) result = Hit{n->index, d};
                break;
            }
            if (depth >= STACK_DEPTH) return Hit{-1, REAL_MAX};   // unreachable: depth is validated at scene creation
            frames[depth++] = TourFrame{n->right, 0, -1, REAL_MAX};
            node = n->left;
        }
        // ---- hand the result upwards
        for (;;) {
            if (depth == 0) return result;
            TourFrame& f = frames[depth - 1];
            if (f.state == 0) {
                f.state = 1;
                f.left_index = result.index;
                f.left_distance = result.distance;
                node = f.right;
                break;
            }
            const Hit l{f.left_index, f.left_distance};
            if (l.distance < result.distance) result = l;
            else if (!(result.distance < l.distance)) result = Hit{-1, REAL_MAX};
            --depth;
        }
    }
}

// Lanes for which the pruned search cannot vouch (`slow`) take turns through the exact traversal.
template <class Leaf>
PT_DEV void resolve_slow_lanes(bool slow, const Ray& ray, const TreeRef& tree, const Leaf& leaf, TourFrame* wave_frames, Hit& hit) {
    unsigned long long pending = __ballot(slow);
    while (pending != 0ull) {
        const int turn = __ffsll((long long)pending) - 1;
        pending &= pending - 1ull;
        if ((int)(threadIdx.x & 63) == turn) hit = tournament(ray, tree, leaf, wave_frames);
    }
}

// Closest-hit search for rays with finite reciprocals: depth-first, nearer child first,
// subtrees entered beyond `limit` (plus the margin) skipped.  `tie` reports a second leaf at
// exactly the winning distance; only then can the answer differ from the tournament's.
template <class Leaf>
PT_DEV Hit closest_pruned(const Ray& ray, D3 inv, const TreeRef& tree, const Leaf& leaf, double limit, int* stack /* [level*BLOCK_THREADS] */, bool& tie) {
    Hit best{-1, REAL_MAX};
    tie = false;
    // In a tree whose boxes do not nest, a leaf can lie outside the boxes above it: what the reference reaches is then
    // decided by those boxes alone, whatever the distances, and nothing may be skipped for being far.
    const bool prune = tree.nested != 0;
    if (!prune) limit = REAL_MAX;
    double cut = prune_cut(limit);
    // A box with an infinity or a NaN can make a slab product NaN, and then the order in which the reference's std::min /
    // std::max take their arguments decides what the test returns.
    const bool short_form = tree.bounded != 0;
    auto slab = [&](const double* box, double& tmin, double& tmax) {
        if (short_form) slab_finite(ray.o, inv, box[0], box[1], box[2], box[3], box[4], box[5], tmin, tmax);
        else slab_exact(ray.o, inv, box, tmin, tmax);
    };
    {
        double tmin, tmax;
        slab(tree.root_box, tmin, tmax);
        if (tmax < 0.0f || tmin > tmax || tmin > cut) return best;
    }
    auto visit_leaf = [&](int primitive) {
        double d;
        if (leaf.test(ray, primitive, d)) {
            if (d < best.distance) {
                best = Hit{primitive, d};
                tie = false;
                if (prune && d < limit) {
                    limit = d;
                    cut = prune_cut(limit);
                }
            } else if (d == best.distance) {
                tie = true;
            }
        }
    };
    if (tree.root < 0) {
        // one primitive, no inner node: the reference returns the leaf's own result, with no comparison in which a hit at
        // an infinite distance (a sphere whose r * r overflows) would lose to a miss (src/path_tracing.cpp:113-127)
        double d;
        if (leaf.test(ray, ~tree.root, d)) best = Hit{~tree.root, d};
        return best;
    }
    int sp = 0;
    int cur = tree.root;
    for (;;) {
        const PairNode* p = tree.pairs + cur;
        const int lref = p->left, rref = p->right;
        double lmin, lmax, rmin, rmax;
        slab(p->lbox, lmin, lmax);
        slab(p->rbox, rmin, rmax);
        bool go_l = !(lmax < 0.0f || lmin > lmax) && !(lmin > cut);
        bool go_r = !(rmax < 0.0f || rmin > rmax) && !(rmin > cut);
        if (go_l && lref < 0) { visit_leaf(~lref); go_l = false; }
        if (go_r && rref < 0) { visit_leaf(~rref); go_r = false; }
        go_l = go_l && !(lmin > cut);
        go_r = go_r && !(rmin > cut);
        if (go_l && go_r) {
            const bool left_first = !(rmin < lmin);
            stack[sp * BLOCK_THREADS] = left_first ? rref : lref;
            ++sp;
            cur = left_first ? lref : rref;
        } else if (go_l) {
            cur = lref;
        } else if (go_r) {
            cur = rref;
        } else {
            if (sp == 0) break;
            --sp;
            cur = stack[sp * BLOCK_THREADS];
        }
    }
    return best;
}

struct BlockShared {
    int stack[STACK_DEPTH * BLOCK_THREADS];                       // [level][thread]
    TourFrame frames[(BLOCK_THREADS / 64) * (STACK_DEPTH + 1)];   // one slice per wave
};

PT_DEV TourFrame* wave_frames(BlockShared& sh) { return sh.frames + (threadIdx.x >> 6) * (STACK_DEPTH + 1); }

// How a ray may be searched.  A NaN anywhere in the ray makes every primitive test fail (the
// distance is NaN) whatever boxes were entered: the reference walks the tree and reports a
// miss.  An infinite reciprocal can make slab products NaN, where the exact min/max order
// of the reference matters: such rays go through the exact traversal.
enum RayClass { RAY_FINITE = 0, RAY_EXACT_ONLY = 1, RAY_POISONED = 2 };
// The same for a caller that goes on to demand |origin| <= 2^20 of a RAY_FINITE ray (ray32_allowed): the origin's
// finiteness is then implied and not tested twice.
PT_DEV int classify_ray_of_bounded_origin(const Ray& ray, D3 inv) {
    if (any_nan(ray.o) || any_nan(ray.d)) return RAY_POISONED;
    return (isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z)) ? RAY_FINITE : RAY_EXACT_ONLY;
}
PT_DEV int classify_ray(const Ray& ray, D3 inv) {
    if (any_nan(ray.o) || any_nan(ray.d)) return RAY_POISONED;
    // an infinite origin makes `plane - origin` infinite or NaN: exact traversal as well
    return (isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z) && isfinite(ray.o.x) && isfinite(ray.o.y) && isfinite(ray.o.z)) ? RAY_FINITE : RAY_EXACT_ONLY;
}

// Closest hit in one tree with the reference's semantics (whole search by the calling lane).
// `lane_stack` is the calling lane's column of an LDS stack laid out [level][BLOCK_THREADS];
// `frames` is the calling wave's slice of tournament frames.
template <class Leaf>
PT_DEV Hit trace_tree(const Ray& ray, const TreeRef& tree, const Leaf& leaf, double limit, int* lane_stack, TourFrame* frames) {
    Hit hit{-1, REAL_MAX};
    if (tree.count == 0) return hit;
    const D3 inv{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
    const int kind = classify_ray(ray, inv);
    bool slow = (kind == RAY_EXACT_ONLY);
    if (kind == RAY_FINITE) {
        bool tie;
        hit = closest_pruned(ray, inv, tree, leaf, limit, lane_stack, tie);
        slow = tie;
    }
    resolve_slow_lanes(slow, ray, tree, leaf, frames, hit);
    return hit;
}
template <class Leaf>
PT_DEV Hit trace_tree(const Ray& ray, const TreeRef& tree, const Leaf& leaf, double limit, BlockShared& sh) {
    return trace_tree(ray, tree, leaf, limit, sh.stack + threadIdx.x, wave_frames(sh));
}

PT_DEV Mat load_material(const PtMaterial* m) { return Mat{m->v[0], m->v[1], m->v[2], m->v[3], m->type}; }

// What happens at a hit / miss (src/path_tracing.cpp:179-223).  Returns true if the path
// goes on with `ray` replaced by the scattered ray; `att` is updated, and `gain` receives the
// radiance factor picked up at this vertex (emission or background), to be multiplied by the
// attenuation the path had on arrival.
PT_DEV bool shade_vertex(const DeviceScene& s, Ray& ray, const Hit& hs, const Hit& ht, D3& att, D3& gain) {
    if (hs.index != -1 || ht.index != -1) {
        D3 p, n;
        int material_index;
        // A sphere can be "hit" at an infinite distance (r * r overflows from |r| = 1.4e154 on) while every triangle is
        // missed: the comparison of :183 then sends the reference to triangle -1, where its assert stops it (:196).
        // Here such a vertex is the sphere's: the one index that exists.  Nothing else is changed by the second clause.
        if (hs.distance < ht.distance || ht.index == -1) {                 // :183-194
            const double* sp = s.spheres + 4 * hs.index;
            p = add3(ray.o, scale3(hs.distance, ray.d));
            const double sign = (sp[3] < 0.0f) ? -1.0f : 1.0f;
            n = scale3(sign, unit3(sub3(p, D3{sp[0], sp[1], sp[2]})));
            material_index = s.sph_material[hs.index];
        } else {                                                           // :195-206
            const TriRecord* t = s.tris + ht.index;
            p = add3(ray.o, scale3(ht.distance, ray.d));
            n = D3{t->un[0], t->un[1], t->un[2]};
            material_index = t->material;
        }
        const Mat m = load_material(s.materials + material_index);
        gain = D3{0.0f, 0.0f, 0.0f};                                       // get_emission, src/material.cpp:64-81
        if (m.type == PT_DIFFUSE_LIGHT) gain = scale3(m.v3, D3{m.v0, m.v1, m.v2});
        Ray next;
        const bool valid = scatter(ray, m, p, n, next, s.with_metals != 0);
        if (valid) {
            ray = next;
            D3 albedo;                                                     // get_colour, src/material.cpp:39-62
            if (m.type == PT_LAMBERTIAN || m.type == PT_METAL) albedo = D3{m.v0, m.v1, m.v2};
            else if (m.type == PT_DIELECTRIC) albedo = D3{1.0f, 1.0f, 1.0f};
            else albedo = D3{0.0f, 0.0f, 0.0f};
            gain = mul3(att, gain);                                        // :214, with the arrival attenuation
            att = mul3(att, albedo);                                       // :215
            return true;
        }
        gain = mul3(att, gain);                                            // :217
        return false;
    }
    const double t = 0.5f * (ray.d.y + 1.0f);                              // background_gradient, :99-102
    const D3 bg = add3(scale3(1.0f - t, s.bg0), scale3(t, s.bg1));
    gain = mul3(att, bg);                                                  // :221
    return false;
}

}  // namespace ptd

#endif
