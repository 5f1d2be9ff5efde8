// TEST INFRASTRUCTURE — not part of the product.
//
// CPU restatement of the reference renderer's per-pixel bounce loop
// (LewisBray/PathTracer, src/path_tracing.cpp and the files it calls).  It is the
// checker the HIP path is compared with in tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg; nothing in the product links, loads or calls it.
//
// Parity pin: every function here is checked bit-for-bit against the reference's
// own sources compiled in place (oracle/_ref/libptref.so, see ref_harness.cpp) by
// tests/test_oracle.py, and against the committed vectors in
// tests/golden/ that were produced by that build (oracle/gen_golden.py).
//
// Arithmetic rules that make the restatement exact:
//   * every operation is IEEE binary64, evaluated left to right as the reference
//     writes it; no FMA (-ffp-contract=off), no reassociation;
//   * float literals of the reference (0.99f, 0.001f, 1.0e-6f ...) are spelt with
//     the f suffix here too, so they widen to the same doubles;
//   * min/max follow std::min/std::max, i.e. min(a,b) = (b < a) ? b : a and
//     max(a,b) = (a < b) ? b : a, which fixes the result when an operand is NaN;
//   * the integer hash runs in uint32 (the reference's signed overflow in
//     src/rng.cpp:25 wraps in practice and under -fwrapv).

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/pathtracer_amd.h"

namespace {

typedef uint32_t u32;

struct V3 { double x, y, z; };

inline V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }          // linear_algebra.cpp:5-7
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }          // :13-15
inline V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }                               // :9-11
inline V3 scale(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }          // :17-19
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // :21-23
inline V3 cross(V3 a, V3 b) {                                                      // :25-31
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(V3 a) { return std::sqrt(dot(a, a)); }                        // :33-35
inline V3 unit(V3 a) {                                                             // :37-40 one reciprocal, three products
    const double inv = 1.0f / length(a);
    return scale(inv, a);
}
inline double stdmin(double a, double b) { return (b < a) ? b : a; }
inline double stdmax(double a, double b) { return (a < b) ? b : a; }

inline V3 v3(const PtVec3& p) { return V3{p.x, p.y, p.z}; }

// ---------------------------------------------------------------- src/rng.cpp
inline u32 hash1(u32 m) {                       // noise_1d, :5-20 (NOISE_SEED = 1, :3)
    m *= 0xB5297A4Du;
    m += 1u;
    m ^= (m >> 8);
    m += 0x68E31DA4u;
    m ^= (m << 8);
    m *= 0x1B56C4E9u;
    m ^= (m >> 8);
    return m;
}
inline u32 hash3(int x, int y, int z) {         // noise_3d, :22-28, wrapping
    return hash1((u32)x + 198491317u * (u32)y + 6542989u * (u32)z);
}
inline u32 xorshift(u32 s) {                    // random_number, :63-69
    s ^= (s << 13);
    s ^= (s >> 17);
    s ^= (s << 5);
    return s;
}
inline double unit_real(u32 r) { return (double)r / (double)UINT_MAX; }   // real_from_rng, :30-32

V3 random_direction(V3 seed) {                  // random_unit_vector, :34-61
    const int sx = (int)(1000000000.0f * seed.x);
    const int sy = (int)(1000000000.0f * seed.y);
    const int sz = (int)(1000000000.0f * seed.z);
    u32 r = hash3(sx, sy, sz);

    r = xorshift(r);
    const double rx = 2.0f * unit_real(r) - 1.0f;
    const double ymax = std::sqrt(1.0f - rx * rx);
    const double ymin = -ymax;

    r = xorshift(r);
    const double ry = (ymax - ymin) * unit_real(r) + ymin;
    const double zmax = std::sqrt(1.0f - rx * rx - ry * ry);
    const double zmin = -zmax;

    r = xorshift(r);
    const double rz = (zmax - zmin) * unit_real(r) + zmin;
    return unit(V3{rx, ry, rz});
}

// ---------------------------------------------------------------- src/geometry.cpp
struct Ray { V3 o, d; };

// intersect(ray, aabb), :7-37.  The three reciprocals depend on the ray only, so callers
// pass them in (same bits as recomputing them at every node).
inline void slab(const Ray& ray, V3 inv, const double* box /* min xyz, max xyz */, double* tmin, double* tmax) {
    const double x0 = (box[0] - ray.o.x) * inv.x, x1 = (box[3] - ray.o.x) * inv.x;
    const double xl = stdmin(x0, x1), xh = stdmax(x0, x1);
    const double y0 = (box[1] - ray.o.y) * inv.y, y1 = (box[4] - ray.o.y) * inv.y;
    const double yl = stdmin(y0, y1), yh = stdmax(y0, y1);
    const double z0 = (box[2] - ray.o.z) * inv.z, z1 = (box[5] - ray.o.z) * inv.z;
    const double zl = stdmin(z0, z1), zh = stdmax(z0, z1);
    *tmin = stdmax(zl, stdmax(yl, xl));        // :33
    *tmax = stdmin(zh, stdmin(yh, xh));        // :34
}
inline V3 reciprocal(V3 d) { return V3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z}; }   // :11,18,25

// intersect(ray, sphere), :59-82
inline bool hit_sphere(const Ray& ray, const double* s /* centre xyz, radius */, double* lo, double* hi) {
    const V3 oc = sub(V3{s[0], s[1], s[2]}, ray.o);
    const double tm = dot(oc, ray.d);
    const double oc2 = dot(oc, oc);
    const double tm2 = tm * tm;
    const double off2 = oc2 - tm2;
    const double r2 = s[3] * s[3];
    const double h2 = r2 - off2;
    if (h2 < 0.0f) return false;
    const double h = std::sqrt(h2);
    *lo = tm - h;
    *hi = tm + h;
    return true;
}

// intersect(ray, triangle), :106-133
inline bool hit_triangle(const Ray& ray, const double* t /* a, b, c */, double* distance) {
    const V3 a{t[0], t[1], t[2]}, b{t[3], t[4], t[5]}, c{t[6], t[7], t[8]};
    const V3 ab = sub(b, a), ac = sub(c, a);
    const V3 n = cross(ab, ac);
    const double dn = dot(ray.d, n);
    const bool parallel = std::fabs(dn) < 1.0e-6f;
    const double dist = dot(sub(a, ray.o), n) / dn;
    const V3 p = add(ray.o, scale(dist, ray.d));
    const V3 ap = sub(p, a);
    const double nn = dot(n, n);
    const double alpha = dot(cross(ab, ap), n) / nn;
    const double beta = dot(cross(ap, ac), n) / nn;
    const bool inside = (alpha >= 0.0f) && (beta >= 0.0f) && (alpha + beta <= 1.0f);
    *distance = dist;
    return !parallel && (dist >= 1.0e-6f) && inside;
}

inline V3 triangle_normal(const double* t) {    // unit_normal, :84-90
    const V3 a{t[0], t[1], t[2]}, b{t[3], t[4], t[5]}, c{t[6], t[7], t[8]};
    return unit(cross(sub(b, a), sub(c, a)));
}

// ---------------------------------------------------------------- src/bvh.cpp
inline void box_of_triangle(const double* t, double* box) {   // construct_aabb(triangle), geometry.cpp:92-104
    for (int k = 0; k < 3; ++k) {
        box[k] = stdmin(t[k], stdmin(t[3 + k], t[6 + k]));
        box[3 + k] = stdmax(t[k], stdmax(t[3 + k], t[6 + k]));
    }
}
inline void box_of_sphere(const double* s, double* box) {     // construct_aabb(sphere), geometry.cpp:51-57
    for (int k = 0; k < 3; ++k) {
        box[k] = s[k] - s[3];
        box[3 + k] = s[k] + s[3];
    }
}

struct Builder {
    std::vector<PtBvhNode>& nodes;
    const std::vector<PtAabb>& boxes;
    std::vector<int>& order;

    static double key(const PtAabb& b, int axis) { return axis == 0 ? b.min.x : (axis == 1 ? b.min.y : b.min.z); }

    // add_node, bvh.cpp:20-58: pre-order numbering, median split of the range sorted by the
    // box minimum on a cycling axis.  The reference sorts the range a second time on the
    // way out (:49-55); that order is never read again but is kept here so the index array
    // goes through the same states.
    int add(int axis, int begin, int end) {
        const int self = (int)nodes.size();
        nodes.push_back(PtBvhNode{});
        const int count = end - begin;
        if (count == 1) {
            nodes[self].aabb = boxes[order[begin]];
            nodes[self].index = order[begin];
            return self;
        }
        std::sort(order.begin() + begin, order.begin() + end,
                  [this, axis](int l, int r) { return key(boxes[l], axis) < key(boxes[r], axis); });
        const int next = (axis + 1) % 3;
        const int mid = begin + count / 2;
        const int left = add(next, begin, mid);
        nodes[self].left = left;
        const int right = add(next, mid, end);
        nodes[self].right = right;
        PtAabb u = nodes[left].aabb;                      // operator+=, geometry.cpp:39-49
        const PtAabb& v = nodes[right].aabb;
        u.min.x = stdmin(u.min.x, v.min.x); u.min.y = stdmin(u.min.y, v.min.y); u.min.z = stdmin(u.min.z, v.min.z);
        u.max.x = stdmax(u.max.x, v.max.x); u.max.y = stdmax(u.max.y, v.max.y); u.max.z = stdmax(u.max.z, v.max.z);
        nodes[self].aabb = u;
        const int prev = (axis + 2) % 3;
        std::sort(order.begin() + begin, order.begin() + end,
                  [this, prev](int l, int r) { return key(boxes[l], prev) < key(boxes[r], prev); });
        return self;
    }
};

std::vector<PtBvhNode> build_bvh(const std::vector<PtAabb>& boxes) {   // construct_*_bvh, bvh.cpp:60-100
    std::vector<PtBvhNode> nodes;
    if (boxes.empty()) return nodes;
    nodes.reserve(2 * boxes.size() - 1);
    std::vector<int> order(boxes.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    Builder b{nodes, boxes, order};
    b.add(0, 0, (int)boxes.size());
    return nodes;
}

// ---------------------------------------------------------------- scene + counters
struct Counters {
    uint64_t samples, bounces, triangle_nodes, triangle_tests, sphere_nodes, sphere_tests;
    uint64_t ties;                 // inner nodes whose two children reported equal finite distances
    uint64_t pruned_mismatch;      // rays where the pruned closest-hit search disagreed with the tournament
    uint64_t ended_at[64];         // histogram of the bounce index at which a path ended (>= 63 clamps)
    void merge(const Counters& o) {
        samples += o.samples; bounces += o.bounces; triangle_nodes += o.triangle_nodes; triangle_tests += o.triangle_tests;
        sphere_nodes += o.sphere_nodes; sphere_tests += o.sphere_tests; ties += o.ties; pruned_mismatch += o.pruned_mismatch;
        for (int i = 0; i < 64; ++i) ended_at[i] += o.ended_at[i];
    }
};

struct OrcScene {
    std::vector<PtMaterial> materials;
    std::vector<PtSphere> spheres;
    std::vector<int> sphere_material;
    std::vector<PtBvhNode> sphere_nodes;
    std::vector<PtTriangle> triangles;
    std::vector<int> triangle_material;
    std::vector<PtBvhNode> triangle_nodes;
    V3 bg0, bg1;
};

struct Hit { int index; double distance; };
const Hit NO_HIT = {-1, DBL_MAX};                 // MISS, path_tracing.cpp:109

// intersect(ray, bvh, node, triangles), path_tracing.cpp:141-169: both children are always
// visited and the smaller distance is kept; equal distances (two misses included) give a miss.
Hit tournament_triangles(const OrcScene& s, const Ray& ray, V3 inv, int node, Counters* k) {
    const PtBvhNode& n = s.triangle_nodes[node];
    if (k) ++k->triangle_nodes;
    double tmin, tmax;
    slab(ray, inv, &n.aabb.min.x, &tmin, &tmax);
    if (tmax < 0.0f || tmin > tmax) return NO_HIT;
    if (n.left == 0 && n.right == 0) {
        if (k) ++k->triangle_tests;
        double d;
        Hit h = NO_HIT;
        if (hit_triangle(ray, &s.triangles[n.index].a.x, &d)) { h.distance = d; h.index = n.index; }
        return h;
    }
    const Hit l = tournament_triangles(s, ray, inv, n.left, k);
    const Hit r = tournament_triangles(s, ray, inv, n.right, k);
    if (l.distance < r.distance) return l;
    if (r.distance < l.distance) return r;
    if (k && l.index != -1) ++k->ties;
    return NO_HIT;
}

// intersect(ray, bvh, node, spheres), path_tracing.cpp:111-139
Hit tournament_spheres(const OrcScene& s, const Ray& ray, V3 inv, int node, Counters* k) {
    const PtBvhNode& n = s.sphere_nodes[node];
    if (k) ++k->sphere_nodes;
    double tmin, tmax;
    slab(ray, inv, &n.aabb.min.x, &tmin, &tmax);
    if (tmax < 0.0f || tmin > tmax) return NO_HIT;
    if (n.left == 0 && n.right == 0) {
        if (k) ++k->sphere_tests;
        double lo, hi;
        Hit h = NO_HIT;
        if (hit_sphere(ray, &s.spheres[n.index].centre.x, &lo, &hi) && (lo > 0.0f || hi > 0.0f)) {
            h.distance = (lo > 0.0f) ? lo : hi;
            h.index = n.index;
        }
        return h;
    }
    const Hit l = tournament_spheres(s, ray, inv, n.left, k);
    const Hit r = tournament_spheres(s, ray, inv, n.right, k);
    if (l.distance < r.distance) return l;
    if (r.distance < l.distance) return r;
    if (k && l.index != -1) ++k->ties;
    return NO_HIT;
}

// A closest-hit search that skips subtrees entered beyond the best hit so far.  Not in the
// reference: it models what a GPU traversal does, and `exact` reports whether its answer is
// guaranteed to equal the tournament's (no second leaf at exactly the best distance).
Hit pruned_triangles(const OrcScene& s, const Ray& ray, V3 inv, bool* exact) {
    Hit best = NO_HIT;
    bool tie = false;
    int stack[64];
    int top = 0;
    stack[top++] = 0;
    while (top > 0) {
        const PtBvhNode& n = s.triangle_nodes[stack[--top]];
        double tmin, tmax;
        slab(ray, inv, &n.aabb.min.x, &tmin, &tmax);
        if (tmax < 0.0f || tmin > tmax) continue;
        if (tmin > best.distance + (1.0e-7 * std::fabs(best.distance) + 1.0e-7)) continue;
        if (n.left == 0 && n.right == 0) {
            double d;
            if (hit_triangle(ray, &s.triangles[n.index].a.x, &d)) {
                if (d < best.distance) { best.distance = d; best.index = n.index; tie = false; }
                else if (d == best.distance) tie = true;
            }
            continue;
        }
        stack[top++] = n.right;
        stack[top++] = n.left;
    }
    *exact = !tie;
    return best;
}

// scatter + reflect/refract/reflectance, path_tracing.cpp:7-97
bool scatter(const Ray& in, const PtMaterial& m, V3 p, V3 n, Ray* out) {
    const double nudge = 0.001f;                                       // :39
    switch (m.type) {
    case PT_LAMBERTIAN: {                                              // :42-50
        const V3 rnd = random_direction(in.d);
        out->o = add(p, scale(nudge, n));
        out->d = unit(add(n, scale(0.99f, rnd)));
        return true;
    }
    case PT_METAL: {                                                   // :52-62
        const V3 rnd = random_direction(in.d);
        const double fuzz = m.v[3];
        const V3 refl = sub(in.d, scale(2.0f * dot(in.d, n), n));      // reflect, :11
        out->o = add(p, scale(nudge, n));
        out->d = unit(add(refl, scale(fuzz, rnd)));
        return dot(out->d, n) > 0.0f;
    }
    case PT_DIELECTRIC: {                                              // :64-85
        const bool front = dot(in.d, n) < 0.0f;
        const double ratio = front ? 1.0f / m.v[0] : m.v[0];
        const V3 nn = front ? n : neg(n);
        const double cos_t = dot(neg(in.d), nn);
        const double sin_t = std::sqrt(1.0f - cos_t * cos_t);
        const u32 h = hash1((u32)(1000000000.0f * sin_t));             // :72, no xorshift step
        const double threshold = unit_real(h);
        const double r = (1.0f - ratio) / (1.0f + ratio);              // reflectance, :27-33
        const double r0 = r * r;
        const double q = 1.0f - cos_t;
        const double schlick = r0 + (1.0f - r0) * q * q * q * q * q;
        if (ratio * sin_t > 1.0f || schlick > threshold) {
            out->o = add(p, scale(nudge, nn));
            out->d = sub(in.d, scale(2.0f * dot(in.d, nn), nn));       // reflect
        } else {
            out->o = sub(p, scale(nudge, nn));
            const double c = dot(neg(in.d), nn);                       // refract, :14-24
            const V3 perp = scale(ratio, add(in.d, scale(c, nn)));
            const V3 para = scale(-std::sqrt(std::fabs(1.0f - dot(perp, perp))), nn);
            out->d = unit(add(perp, para));
        }
        return true;
    }
    default:                                                           // DIFFUSE_LIGHT, :87-90
        *out = Ray{V3{0, 0, 0}, V3{0, 0, 0}};
        return false;
    }
}

// intersect(ray, scene), path_tracing.cpp:171-227, with MAX_BOUNCE_COUNT (:172) as a parameter.
V3 trace_path(const OrcScene& s, Ray ray, int max_bounces, Counters* k, bool check_pruned) {
    V3 colour{0.0f, 0.0f, 0.0f};
    V3 att{1.0f, 1.0f, 1.0f};
    int bounce = 0;
    for (; bounce < max_bounces; ++bounce) {
        if (k) ++k->bounces;
        const V3 inv = reciprocal(ray.d);
        const Hit hs = s.sphere_nodes.empty() ? NO_HIT : tournament_spheres(s, ray, inv, 0, k);
        const Hit ht = s.triangle_nodes.empty() ? NO_HIT : tournament_triangles(s, ray, inv, 0, k);
        if (check_pruned && k && !s.triangle_nodes.empty()) {
            bool exact;
            const Hit hp = pruned_triangles(s, ray, inv, &exact);
            if (exact && (hp.index != ht.index || hp.distance != ht.distance)) ++k->pruned_mismatch;
        }
        if (hs.index != -1 || ht.index != -1) {
            V3 p, n;
            const PtMaterial* m;
            // ht.index == -1 with hs.distance not below REAL_MAX (a sphere "hit" at infinity: r * r overflowed): the reference's
            // assert stops it at :196; the checker, like the library, lets the sphere have the vertex
            if (hs.distance < ht.distance || ht.index == -1) {                       // :183
                const PtSphere& sp = s.spheres[hs.index];
                p = add(ray.o, scale(hs.distance, ray.d));
                const double sign = (sp.radius < 0.0f) ? -1.0f : 1.0f;               // :190
                n = scale(sign, unit(sub(p, v3(sp.centre))));
                m = &s.materials[s.sphere_material[hs.index]];
            } else {
                p = add(ray.o, scale(ht.distance, ray.d));
                n = triangle_normal(&s.triangles[ht.index].a.x);                     // :202
                m = &s.materials[s.triangle_material[ht.index]];
            }
            V3 emission{0.0f, 0.0f, 0.0f};                                           // get_emission, material.cpp:64-81
            if (m->type == PT_DIFFUSE_LIGHT) emission = scale(m->v[3], V3{m->v[0], m->v[1], m->v[2]});
            Ray next;
            const bool valid = scatter(ray, *m, p, n, &next);
            colour = add(colour, V3{att.x * emission.x, att.y * emission.y, att.z * emission.z});   // :214 / :217
            if (!valid) break;
            ray = next;
            V3 albedo;                                                               // get_colour, material.cpp:39-62
            if (m->type == PT_LAMBERTIAN || m->type == PT_METAL) albedo = V3{m->v[0], m->v[1], m->v[2]};
            else if (m->type == PT_DIELECTRIC) albedo = V3{1.0f, 1.0f, 1.0f};
            else albedo = V3{0.0f, 0.0f, 0.0f};
            att = V3{att.x * albedo.x, att.y * albedo.y, att.z * albedo.z};          // :215
        } else {
            const double t = 0.5f * (ray.d.y + 1.0f);                                // background_gradient, :99-102
            const V3 bg = add(scale(1.0f - t, s.bg0), scale(t, s.bg1));
            colour = add(colour, V3{att.x * bg.x, att.y * bg.y, att.z * bg.z});      // :221
            break;
        }
    }
    if (k) { ++k->samples; ++k->ended_at[bounce < 63 ? bounce : 63]; }
    return colour;
}

// per-pixel body of render_scanline, src/main.cpp:87-113
Ray camera_ray(const PtFrame& f, int row, int column, int sample) {
    const V3 pos = v3(f.camera_position), cx = v3(f.camera_x), cy = v3(f.camera_y);
    const V3 bl = v3(f.bottom_left), sx = v3(f.step_x), sy = v3(f.step_y);
    const double lens = 0.5f * f.aperture;
    u32 r = hash3(row, column, sample);
    r = xorshift(r);
    const double u = ((double)column + unit_real(r)) / (double)(f.width - 1);
    r = xorshift(r);
    const double v = ((double)row + unit_real(r)) / (double)(f.height - 1);
    r = xorshift(r);
    const double a = f.aperture * unit_real(r) - lens;
    const double bmax = std::sqrt(lens * lens - a * a);
    const double bmin = -bmax;
    r = xorshift(r);
    const double b = (bmax - bmin) * unit_real(r) + bmin;
    const V3 off = add(scale(a, cx), scale(b, cy));
    const V3 d = unit(sub(sub(add(add(bl, scale(u, sx)), scale(v, sy)), pos), off));
    return Ray{add(pos, off), d};
}

void render_rows(const OrcScene& s, const PtFrame& f, double* accum, int threads, Counters* totals, bool check_pruned) {
    std::atomic<int> next_row{f.row_begin};
    std::vector<Counters> per_thread(threads > 1 ? threads : 1);
    for (Counters& c : per_thread) std::memset(&c, 0, sizeof(c));
    auto worker = [&](int tid) {
        Counters* k = totals ? &per_thread[tid] : nullptr;
        for (;;) {
            const int row = next_row.fetch_add(1);
            if (row >= f.row_end) break;
            if (f.shard_count > 1 && ((row - f.row_begin) / PT_ROW_BLOCK) % f.shard_count != f.shard_index) continue;
            // sample-major inside a row: a pixel still receives its samples in ascending order,
            // which is all the frame-by-frame accumulation of src/main.cpp:1001-1036 fixes.
            for (int sample = f.sample_begin; sample < f.sample_begin + f.sample_count; ++sample) {
                for (int column = 0; column < f.width; ++column) {
                    const V3 c = trace_path(s, camera_ray(f, row, column, sample), f.max_bounces, k, check_pruned);
                    double* px = accum + 4ll * ((long long)row * f.width + column);  // main.cpp:117-121
                    px[0] += c.z;
                    px[1] += c.y;
                    px[2] += c.x;
                    px[3] += 1.0f;
                }
            }
        }
    };
    if (threads <= 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i) pool.emplace_back(worker, i);
        for (std::thread& t : pool) t.join();
    }
    if (totals) for (const Counters& c : per_thread) totals->merge(c);
}

Ray ray6(const double* p) { return Ray{V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}}; }

}  // namespace

extern "C" {

// ---------------------------------------------------------------- known-answer hooks
unsigned orc_noise_1d(int x) { return hash1((u32)x); }
unsigned orc_noise_3d(int x, int y, int z) { return hash3(x, y, z); }
unsigned orc_random_number(unsigned seed) { return xorshift(seed); }
double orc_real_from_rng(unsigned rng) { return unit_real(rng); }

// Checks, for every r in [begin, end), that the GPU library's division-free form of real_from_rng
// (pathtracer_amd/csrc/pt_device.hpp) gives the bits of the reference's r / UINT_MAX; returns the
// number of values that differ.  std::fma is exact (one rounding) whether or not the CPU has the instruction.
unsigned long long orc_check_fused_unit_real(unsigned long long begin, unsigned long long end, int threads) {
    std::vector<std::thread> pool;
    std::vector<unsigned long long> wrong(threads > 0 ? threads : 1, 0);
    const int n = (int)wrong.size();
    for (int t = 0; t < n; ++t) {
        pool.emplace_back([&, t]() {
            const unsigned long long lo = begin + (end - begin) * t / n, hi = begin + (end - begin) * (t + 1) / n;
            unsigned long long bad = 0;
            for (unsigned long long i = lo; i < hi; ++i) {
                const u32 r = (u32)i;
                const double d = (double)r;
                const double fused = std::fma(d, 0x1p-32, d * (0x1p-64 + 0x1p-96));
                const double divided = unit_real(r);
                bad += std::memcmp(&fused, &divided, sizeof(double)) != 0;
            }
            wrong[t] = bad;
        });
    }
    for (std::thread& th : pool) th.join();
    unsigned long long total = 0;
    for (unsigned long long w : wrong) total += w;
    return total;
}
void orc_random_unit_vector(const double* seed, double* out) {
    const V3 r = random_direction(V3{seed[0], seed[1], seed[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_ray_aabb(const double* ray, const double* aabb, double* out) {
    const Ray r = ray6(ray);
    slab(r, reciprocal(r.d), aabb, &out[0], &out[1]);
}
void orc_ray_sphere(const double* ray, const double* sphere, double* out) {
    double lo = 0.0, hi = 0.0;
    const bool ok = hit_sphere(ray6(ray), sphere, &lo, &hi);
    out[0] = ok ? 1.0 : 0.0; out[1] = lo; out[2] = hi;
}
void orc_ray_triangle(const double* ray, const double* tri, double* out) {
    double d;
    const bool ok = hit_triangle(ray6(ray), tri, &d);
    out[0] = ok ? 1.0 : 0.0; out[1] = d;
}
void orc_unit_normal(const double* tri, double* out) {
    const V3 n = triangle_normal(tri);
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
void orc_scatter(const double* ray, const double* material, const double* point, const double* normal, double* out) {
    PtMaterial m;
    std::memset(&m, 0, sizeof(m));
    std::memcpy(m.v, material, 32);
    m.type = (int)material[4];
    Ray next;
    const bool ok = scatter(ray6(ray), m, V3{point[0], point[1], point[2]}, V3{normal[0], normal[1], normal[2]}, &next);
    out[0] = ok ? 1.0 : 0.0;
    out[1] = next.o.x; out[2] = next.o.y; out[3] = next.o.z;
    out[4] = next.d.x; out[5] = next.d.y; out[6] = next.d.z;
}

// ---------------------------------------------------------------- host entry points
int orc_construct_triangle_bvh(const PtTriangle* triangles, int count, PtBvhNode* nodes) {
    std::vector<PtAabb> boxes(count);
    for (int i = 0; i < count; ++i) box_of_triangle(&triangles[i].a.x, &boxes[i].min.x);
    const std::vector<PtBvhNode> out = build_bvh(boxes);
    std::memcpy(nodes, out.data(), out.size() * sizeof(PtBvhNode));
    return (int)out.size();
}
int orc_construct_sphere_bvh(const PtSphere* spheres, int count, PtBvhNode* nodes) {
    std::vector<PtAabb> boxes(count);
    for (int i = 0; i < count; ++i) box_of_sphere(&spheres[i].centre.x, &boxes[i].min.x);
    const std::vector<PtBvhNode> out = build_bvh(boxes);
    std::memcpy(nodes, out.data(), out.size() * sizeof(PtBvhNode));
    return (int)out.size();
}

// rotation_matrix, linear_algebra.cpp:81-101 (Rodrigues form, axis normalised first)
void orc_rotation_matrix(double angle, double ax, double ay, double az, PtMat3* out) {
    const V3 k = unit(V3{ax, ay, az});
    const double s = std::sin(angle), c = std::cos(angle);
    double (*m)[3] = out->rows;
    m[0][0] = (1.0f - c) * k.x * k.x + c;
    m[0][1] = (1.0f - c) * k.x * k.y - s * k.z;
    m[0][2] = (1.0f - c) * k.x * k.z + s * k.y;
    m[1][0] = (1.0f - c) * k.x * k.y + s * k.z;
    m[1][1] = (1.0f - c) * k.y * k.y + c;
    m[1][2] = (1.0f - c) * k.y * k.z - s * k.x;
    m[2][0] = (1.0f - c) * k.x * k.z - s * k.y;
    m[2][1] = (1.0f - c) * k.y * k.z + s * k.x;
    m[2][2] = (1.0f - c) * k.z * k.z + c;
}

// look_at_matrix, linear_algebra.cpp:103-122: columns are the camera's x, y, z axes
void orc_look_at_matrix(const PtVec3* position, const PtVec3* target, PtMat3* out) {
    const V3 z = unit(sub(v3(*position), v3(*target)));
    const V3 x = unit(cross(V3{0.0f, 1.0f, 0.0f}, z));
    const V3 y = cross(z, x);
    double (*m)[3] = out->rows;
    m[0][0] = x.x; m[1][0] = x.y; m[2][0] = x.z;
    m[0][1] = y.x; m[1][1] = y.y; m[2][1] = y.z;
    m[0][2] = z.x; m[1][2] = z.y; m[2][2] = z.z;
}

void orc_camera_look_at(const PtVec3* position, const PtVec3* target, double fov_y, double aperture, double focus_distance, PtCamera* out) {
    std::memset(out, 0, sizeof(*out));
    out->target = *target;
    orc_look_at_matrix(position, target, &out->orientation);
    out->distance = length(sub(v3(*position), v3(*target)));
    out->fov_y = fov_y;
    out->aperture = aperture;
    out->focus_distance = (focus_distance > 0.0) ? focus_distance : out->distance;
}

double orc_degrees_to_radians(double degrees) {               // src/main.cpp:27-30, float pi
    const double pi = 3.14159265358979323846264f;
    return degrees / 180.0f * pi;
}

// src/main.cpp:928-930, 987-994; get_position, path_tracing.cpp:229-231
void orc_camera_frame(const PtCamera* cam, double aspect_ratio, int width, int height, PtFrame* frame) {
    const double (*m)[3] = cam->orientation.rows;
    const V3 cx{m[0][0], m[1][0], m[2][0]}, cy{m[0][1], m[1][1], m[2][1]}, cz{m[0][2], m[1][2], m[2][2]};
    const V3 back{0.0f, 0.0f, cam->distance};
    const V3 rotated{m[0][0] * back.x + m[0][1] * back.y + m[0][2] * back.z,
                     m[1][0] * back.x + m[1][1] * back.y + m[1][2] * back.z,
                     m[2][0] * back.x + m[2][1] * back.y + m[2][2] * back.z};
    const V3 pos = add(rotated, v3(cam->target));
    const double vh = 2.0f * std::tan(0.5f * cam->fov_y);
    const double vw = aspect_ratio * vh;
    const V3 sx = scale(cam->focus_distance * vw, cx);
    const V3 sy = scale(cam->focus_distance * vh, cy);
    const V3 bl = sub(sub(sub(pos, scale(0.5f, sx)), scale(0.5f, sy)), scale(cam->focus_distance, cz));
    std::memset(frame, 0, sizeof(*frame));
    frame->width = width; frame->height = height;
    frame->row_begin = 0; frame->row_end = height;
    frame->sample_begin = 0; frame->sample_count = 1;
    frame->shard_index = 0; frame->shard_count = 1;
    frame->max_bounces = 50;
    frame->aperture = cam->aperture;
    frame->camera_position = PtVec3{pos.x, pos.y, pos.z};
    frame->camera_x = PtVec3{cx.x, cx.y, cx.z};
    frame->camera_y = PtVec3{cy.x, cy.y, cy.z};
    frame->bottom_left = PtVec3{bl.x, bl.y, bl.z};
    frame->step_x = PtVec3{sx.x, sx.y, sx.z};
    frame->step_y = PtVec3{sy.x, sy.y, sy.z};
}

// ---------------------------------------------------------------- scenes
OrcScene* orc_scene_create(const PtSceneDesc* d) {
    OrcScene* s = new OrcScene();
    s->materials.assign(d->materials, d->materials + d->material_count);
    if (d->sphere_count > 0) {
        s->spheres.assign(d->spheres, d->spheres + d->sphere_count);
        s->sphere_material.assign(d->sphere_material_indices, d->sphere_material_indices + d->sphere_count);
        s->sphere_nodes.resize(2 * d->sphere_count - 1);
        // a tree that comes with the description is traversed as it is (the reference's Scene only points at one: path_tracing.h:9-22)
        if (d->sphere_bvh != nullptr && d->sphere_bvh_node_count == 2 * d->sphere_count - 1) std::memcpy(s->sphere_nodes.data(), d->sphere_bvh, s->sphere_nodes.size() * sizeof(PtBvhNode));
        else orc_construct_sphere_bvh(s->spheres.data(), d->sphere_count, s->sphere_nodes.data());
    }
    if (d->triangle_count > 0) {
        s->triangles.assign(d->triangles, d->triangles + d->triangle_count);
        s->triangle_material.assign(d->triangle_material_indices, d->triangle_material_indices + d->triangle_count);
        s->triangle_nodes.resize(2 * d->triangle_count - 1);
        if (d->triangle_bvh != nullptr && d->triangle_bvh_node_count == 2 * d->triangle_count - 1) std::memcpy(s->triangle_nodes.data(), d->triangle_bvh, s->triangle_nodes.size() * sizeof(PtBvhNode));
        else orc_construct_triangle_bvh(s->triangles.data(), d->triangle_count, s->triangle_nodes.data());
    }
    s->bg0 = V3{d->background_gradient_start.r, d->background_gradient_start.g, d->background_gradient_start.b};
    s->bg1 = V3{d->background_gradient_end.r, d->background_gradient_end.g, d->background_gradient_end.b};
    return s;
}

// The reference's active scene, src/main.cpp:800-840, with the model file as a parameter.
OrcScene* orc_model_scene(const char* triangles_path, PtCamera* camera_out, double* aspect_out) {
    FILE* f = std::fopen(triangles_path, "rb");                     // load_triangles_file, model_loading.cpp:10-39
    if (!f) return nullptr;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (bytes <= 0 || bytes % (long)sizeof(PtTriangle) != 0) { std::fclose(f); return nullptr; }
    std::vector<PtTriangle> tris(bytes / sizeof(PtTriangle));
    const size_t got = std::fread(tris.data(), 1, bytes, f);
    std::fclose(f);
    if (got != (size_t)bytes) return nullptr;

    const double pi = 3.14159265358979323846264f;
    PtMat3 rot;
    orc_rotation_matrix(-pi / 2.0f, 1.0f, 0.0f, 0.0f, &rot);        // main.cpp:808
    for (PtTriangle& t : tris) {                                    // :809-813
        PtVec3* v[3] = {&t.a, &t.b, &t.c};
        for (PtVec3* p : v) {
            const PtVec3 q = *p;
            p->x = rot.rows[0][0] * q.x + rot.rows[0][1] * q.y + rot.rows[0][2] * q.z;
            p->y = rot.rows[1][0] * q.x + rot.rows[1][1] * q.y + rot.rows[1][2] * q.z;
            p->z = rot.rows[2][0] * q.x + rot.rows[2][1] * q.y + rot.rows[2][2] * q.z;
        }
    }
    PtMaterial mats[2];
    std::memset(mats, 0, sizeof(mats));
    mats[0].type = PT_LAMBERTIAN;                                   // :801 (float divisions, then widened)
    mats[0].v[0] = 6.0f / 255.0f; mats[0].v[1] = 4.0f / 255.0f; mats[0].v[2] = 3.0f / 255.0f;
    mats[1].type = PT_DIFFUSE_LIGHT;                                // :802
    mats[1].v[0] = 1.0f; mats[1].v[1] = 1.0f; mats[1].v[2] = 1.0f; mats[1].v[3] = 10.0f;
    const PtSphere light = {PtVec3{20.0f, 80.0f, 10.0f}, 20.0f};    // :817
    const int light_material = 1;                                   // :818
    std::vector<int> tri_material(tris.size(), 0);                  // :806

    PtSceneDesc d;
    std::memset(&d, 0, sizeof(d));
    d.materials = mats; d.material_count = 2;
    d.spheres = &light; d.sphere_count = 1; d.sphere_material_indices = &light_material;
    d.triangles = tris.data(); d.triangle_count = (int)tris.size(); d.triangle_material_indices = tri_material.data();
    d.background_gradient_start = PtColour{0.01f, 0.01f, 0.01f};    // :829-830
    d.background_gradient_end = PtColour{0.01f, 0.01f, 0.01f};
    OrcScene* s = orc_scene_create(&d);

    if (camera_out) {                                               // :833-840
        const PtVec3 start{0.0f, 150.0f, 150.0f}, target{0.0f, 0.0f, 0.0f};
        orc_camera_look_at(&start, &target, orc_degrees_to_radians(40.0f), 0.1f, 0.0, camera_out);
    }
    if (aspect_out) *aspect_out = 1.0f;
    return s;
}

namespace {
PtMaterial make_material(int type, double a, double b, double c, double d) {
    PtMaterial m;
    std::memset(&m, 0, sizeof(m));
    m.v[0] = a; m.v[1] = b; m.v[2] = c; m.v[3] = d;
    m.type = type;
    return m;
}

// Mat3 * Mat3, linear_algebra.cpp:50-66 (every product and sum kept, zeros included)
void mat_mul(const PtMat3& a, const PtMat3& b, PtMat3* out) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out->rows[i][j] = a.rows[i][0] * b.rows[0][j] + a.rows[i][1] * b.rows[1][j] + a.rows[i][2] * b.rows[2][j];
}

PtVec3 mat_apply(const PtMat3& m, const PtVec3& v, const PtVec3& shift) {       // linear_algebra.cpp:42-48, then + shift
    const double x = m.rows[0][0] * v.x + m.rows[0][1] * v.y + m.rows[0][2] * v.z;
    const double y = m.rows[1][0] * v.x + m.rows[1][1] * v.y + m.rows[1][2] * v.z;
    const double z = m.rows[2][0] * v.x + m.rows[2][1] * v.y + m.rows[2][2] * v.z;
    return PtVec3{x + shift.x, y + shift.y, z + shift.z};
}

// the 12 triangles of the unit cube in the order +z, +x, -z, -x, +y, -y (src/main.cpp:643-667):
// each face is a quad p0 p1 p2 p3 split as (p0 p1 p2), (p2 p3 p0); corners as bit masks x | y<<1 | z<<2
const unsigned char CUBE_FACES[6][4] = {
    {6, 4, 5, 7}, {7, 5, 1, 3}, {3, 1, 0, 2}, {2, 0, 4, 6}, {7, 3, 2, 6}, {4, 0, 1, 5}};

void append_box(std::vector<PtTriangle>* out, double sx, double sy, double sz, double angle, const PtVec3& shift) {
    PtMat3 scale_m, rot, both;
    std::memset(&scale_m, 0, sizeof(scale_m));                                  // scaling_matrix, linear_algebra.cpp:72-79
    scale_m.rows[0][0] = sx; scale_m.rows[1][1] = sy; scale_m.rows[2][2] = sz;
    orc_rotation_matrix(angle, 0.0f, 1.0f, 0.0f, &rot);
    mat_mul(scale_m, rot, &both);                                               // :669, :679
    for (const unsigned char* face : {CUBE_FACES[0], CUBE_FACES[1], CUBE_FACES[2], CUBE_FACES[3], CUBE_FACES[4], CUBE_FACES[5]}) {
        PtVec3 p[4];
        for (int k = 0; k < 4; ++k) {
            const PtVec3 corner{(double)(face[k] & 1), (double)((face[k] >> 1) & 1), (double)((face[k] >> 2) & 1)};
            p[k] = mat_apply(both, corner, shift);                              // :671-673, :681-683
        }
        out->push_back(PtTriangle{p[0], p[1], p[2]});
        out->push_back(PtTriangle{p[2], p[3], p[0]});
    }
}
}  // namespace

// The reference's two other built-in scenes: kind 0 = random spheres (src/main.cpp:516-625),
// kind 1 = Cornell box (src/main.cpp:627-798).
OrcScene* orc_builtin_scene(int kind, PtCamera* camera_out, double* aspect_out) {
    std::vector<PtMaterial> mats;
    std::vector<PtSphere> spheres;
    std::vector<int> sphere_material;
    std::vector<PtTriangle> tris;
    std::vector<int> tri_material;
    PtSceneDesc d;
    std::memset(&d, 0, sizeof(d));
    PtVec3 eye, target;
    double fov_degrees, focus, aspect;
    if (kind == 0) {
        auto add = [&](const PtMaterial& m, double x, double y, double z, double r) {
            sphere_material.push_back((int)mats.size());                        // sphere k uses material k
            mats.push_back(m);
            spheres.push_back(PtSphere{PtVec3{x, y, z}, r});
        };
        add(make_material(PT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0), 0.0f, -1000.0f, 0.0f, 1000.0f);   // :522-523
        u32 r = 479001599u;                                                     // :527
        auto next = [&]() { r = xorshift(r); return unit_real(r); };
        for (int a = -11; a < 11; ++a) {
            for (int b = -11; b < 11; ++b) {
                const double choice = next();                                   // :530-537
                const double dx = 0.9f * next();
                const double dz = 0.9f * next();
                PtMaterial m;
                if (choice < 0.8f) {                                            // :542-562: product of two random colours
                    double c1[3], c2[3];
                    for (double& c : c1) c = next();
                    for (double& c : c2) c = next();
                    m = make_material(PT_LAMBERTIAN, c1[0] * c2[0], c1[1] * c2[1], c1[2] * c2[2], 0.0);
                } else if (choice < 0.95f) {                                    // :563-576
                    double c[3];
                    for (double& x : c) x = 0.5f * next() + 0.5f;
                    const double fuzz = next();
                    m = make_material(PT_METAL, c[0], c[1], c[2], fuzz);
                } else {
                    m = make_material(PT_DIELECTRIC, 1.5f, 0.0, 0.0, 0.0);      // :578
                }
                add(m, (double)a + dx, 0.2f, (double)b + dz, 0.2f);             // :539-540
            }
        }
        add(make_material(PT_DIELECTRIC, 1.5f, 0.0, 0.0, 0.0), 0.0f, 1.0f, 0.0f, 1.0f);             // :588-589
        add(make_material(PT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0), -4.0f, 1.0f, 0.0f, 1.0f);          // :593-594
        add(make_material(PT_METAL, 0.7f, 0.6f, 0.5f, 0.0f), 4.0f, 1.0f, 0.0f, 1.0f);               // :598-599
        d.background_gradient_start = PtColour{1.0f, 1.0f, 1.0f};               // :614-615
        d.background_gradient_end = PtColour{0.5f, 0.7f, 1.0f};
        eye = PtVec3{13.0f, 2.0f, 3.0f}; target = PtVec3{0.0f, 0.0f, 0.0f};     // :618-625
        fov_degrees = 20.0f; focus = 10.0f; aspect = 3.0f / 2.0f;               // :33
    } else if (kind == 1) {
        mats = {make_material(PT_LAMBERTIAN, 0.65f, 0.05f, 0.05f, 0.0),         // :627-633
                make_material(PT_LAMBERTIAN, 0.73f, 0.73f, 0.73f, 0.0),
                make_material(PT_LAMBERTIAN, 0.12f, 0.45f, 0.15f, 0.0),
                make_material(PT_DIFFUSE_LIGHT, 1.0f, 1.0f, 1.0f, 15.0f),
                make_material(PT_DIELECTRIC, 1.5f, 0.0, 0.0, 0.0)};
        spheres = {PtSphere{PtVec3{183.0f, 240.0f, 169.0f}, 75.0f}};            // :635-639
        sphere_material = {4};
        // walls, floor, ceiling, back and light as quads p0 p1 p2 p3 -> (p0 p1 p2), (p2 p3 p0), :687-710
        const double L = 555.0f;
        const struct { double p[4][3]; int material; } quads[6] = {
            {{{L, 0, 0}, {L, 0, L}, {L, L, L}, {L, L, 0}}, 2},
            {{{0, 0, 0}, {0, L, 0}, {0, L, L}, {0, 0, L}}, 0},
            {{{0, 0, 0}, {0, 0, L}, {L, 0, L}, {L, 0, 0}}, 1},
            {{{0, L, 0}, {L, L, 0}, {L, L, L}, {0, L, L}}, 1},
            {{{0, 0, L}, {0, L, L}, {L, L, L}, {L, 0, L}}, 1},
            {{{213.0f, 554.0f, 227.0f}, {343.0f, 554.0f, 227.0f}, {343.0f, 554.0f, 332.0f}, {213.0f, 554.0f, 332.0f}}, 3}};
        for (const auto& q : quads) {
            PtVec3 p[4];
            for (int k = 0; k < 4; ++k) p[k] = PtVec3{q.p[k][0], q.p[k][1], q.p[k][2]};
            tris.push_back(PtTriangle{p[0], p[1], p[2]});
            tris.push_back(PtTriangle{p[2], p[3], p[0]});
            tri_material.push_back(q.material);                                 // :751-768
            tri_material.push_back(q.material);
        }
        const double pi = 3.14159265358979323846264f;
        append_box(&tris, 165.0f, 165.0f, 165.0f, -pi / 10.0f, PtVec3{130.0f, 0.0f, 65.0f});        // :669-673
        append_box(&tris, 165.0f, 330.0f, 165.0f, pi / 12.0f, PtVec3{265.0f, 0.0f, 295.0f});        // :679-683
        tri_material.resize(tris.size(), 1);                                    // :770-774
        d.background_gradient_start = PtColour{0.0f, 0.0f, 0.0f};               // :787-788
        d.background_gradient_end = PtColour{0.0f, 0.0f, 0.0f};
        eye = PtVec3{278.0f, 278.0f, -800.0f}; target = PtVec3{278.0f, 278.0f, 0.0f};               // :791-798
        fov_degrees = 40.0f; focus = 0.0; aspect = 1.0f;                        // focus = camera distance; :34
    } else {
        return nullptr;
    }
    d.materials = mats.data(); d.material_count = (int)mats.size();
    d.spheres = spheres.data(); d.sphere_count = (int)spheres.size(); d.sphere_material_indices = sphere_material.data();
    d.triangles = tris.empty() ? nullptr : tris.data(); d.triangle_count = (int)tris.size();
    d.triangle_material_indices = tris.empty() ? nullptr : tri_material.data();
    OrcScene* s = orc_scene_create(&d);
    if (camera_out) orc_camera_look_at(&eye, &target, orc_degrees_to_radians(fov_degrees), 0.1f, focus, camera_out);
    if (aspect_out) *aspect_out = aspect;
    return s;
}

void orc_scene_destroy(OrcScene* s) { delete s; }

int orc_scene_counts(const OrcScene* s, int* out) {
    out[0] = (int)s->materials.size(); out[1] = (int)s->spheres.size(); out[2] = (int)s->sphere_nodes.size();
    out[3] = (int)s->triangles.size(); out[4] = (int)s->triangle_nodes.size();
    return 0;
}

int orc_scene_export(const OrcScene* s, PtMaterial* materials, PtSphere* spheres, int* sphere_mat, PtBvhNode* sphere_nodes,
                     PtTriangle* triangles, int* triangle_mat, PtBvhNode* triangle_nodes, PtColour* background) {
    if (materials) std::memcpy(materials, s->materials.data(), s->materials.size() * sizeof(PtMaterial));
    if (spheres) std::memcpy(spheres, s->spheres.data(), s->spheres.size() * sizeof(PtSphere));
    if (sphere_mat) std::memcpy(sphere_mat, s->sphere_material.data(), s->sphere_material.size() * sizeof(int));
    if (sphere_nodes) std::memcpy(sphere_nodes, s->sphere_nodes.data(), s->sphere_nodes.size() * sizeof(PtBvhNode));
    if (triangles) std::memcpy(triangles, s->triangles.data(), s->triangles.size() * sizeof(PtTriangle));
    if (triangle_mat) std::memcpy(triangle_mat, s->triangle_material.data(), s->triangle_material.size() * sizeof(int));
    if (triangle_nodes) std::memcpy(triangle_nodes, s->triangle_nodes.data(), s->triangle_nodes.size() * sizeof(PtBvhNode));
    if (background) {
        background[0] = PtColour{s->bg0.x, s->bg0.y, s->bg0.z};
        background[1] = PtColour{s->bg1.x, s->bg1.y, s->bg1.z};
    }
    return 0;
}

// ---------------------------------------------------------------- the hot path
int orc_render(const OrcScene* s, const PtFrame* f, double* accum, int threads) {
    render_rows(*s, *f, accum, threads, nullptr, false);
    return 0;
}

// Same render, also filling the work counters used for the algorithmic bytes / flops per
// sample (64 B per node visit, 72 B per triangle test ...).  counters_out: 8 + 64 uint64.
int orc_render_counted(const OrcScene* s, const PtFrame* f, double* accum, int threads, int check_pruned, uint64_t* counters_out) {
    Counters k;
    std::memset(&k, 0, sizeof(k));
    render_rows(*s, *f, accum, threads, &k, check_pruned != 0);
    std::memcpy(counters_out, &k, sizeof(k));
    return 0;
}

int orc_sample_colours(const OrcScene* s, const PtFrame* f, int sample, double* rgb) {
    for (int row = f->row_begin; row < f->row_end; ++row) {
        for (int column = 0; column < f->width; ++column) {
            const V3 c = trace_path(*s, camera_ray(*f, row, column, sample), f->max_bounces, nullptr, false);
            double* out = rgb + 3ll * ((long long)row * f->width + column);
            out[0] = c.x; out[1] = c.y; out[2] = c.z;
        }
    }
    return 0;
}

int orc_primary_rays(const PtFrame* f, int sample, double* rays) {
    for (int row = f->row_begin; row < f->row_end; ++row) {
        for (int column = 0; column < f->width; ++column) {
            const Ray r = camera_ray(*f, row, column, sample);
            std::memcpy(rays + 6ll * ((long long)row * f->width + column), &r, sizeof(r));
        }
    }
    return 0;
}

int orc_primary_hits(const OrcScene* s, const PtFrame* f, int* triangle_index, int* sphere_index) {
    for (int row = f->row_begin; row < f->row_end; ++row) {
        for (int column = 0; column < f->width; ++column) {
            const Ray ray = camera_ray(*f, row, column, f->sample_begin);
            const V3 inv = reciprocal(ray.d);
            const long long i = (long long)row * f->width + column;
            if (triangle_index) triangle_index[i] = s->triangle_nodes.empty() ? -1 : tournament_triangles(*s, ray, inv, 0, nullptr).index;
            if (sphere_index) sphere_index[i] = s->sphere_nodes.empty() ? -1 : tournament_spheres(*s, ray, inv, 0, nullptr).index;
        }
    }
    return 0;
}

int orc_trace_triangles(const OrcScene* s, const double* rays, long long n, int* index, double* distance) {
    for (long long i = 0; i < n; ++i) {
        const Ray ray = ray6(rays + 6 * i);
        const Hit h = s->triangle_nodes.empty() ? NO_HIT : tournament_triangles(*s, ray, reciprocal(ray.d), 0, nullptr);
        index[i] = h.index; distance[i] = h.distance;
    }
    return 0;
}

// src/main.cpp:1024-1034, sample_count standing for (sample + 1)
int orc_tonemap_bgra8(const double* accum, long long pixel_count, int sample_count, unsigned char* out) {
    for (long long i = 0; i < pixel_count; ++i) {
        for (int c = 0; c < 3; ++c) {
            const double v = stdmin(std::sqrt(accum[4 * i + c] / (double)sample_count), 1.0f);
            out[4 * i + c] = (unsigned char)(255.0f * v);
        }
        out[4 * i + 3] = 255;
    }
    return 0;
}

// The camera controls of the frame loop, src/main.cpp:890-985 (PtViewerInput holds what MouseInput / KeyboardInput and
// frame_duration hold there).  Returns 1 if the camera changed (camera_modified).
int orc_camera_apply_input(PtCamera* cam, const PtViewerInput* in) {
    int modified = 0;
    const double dt = in->frame_duration;
    if (in->left_button_held && (in->mouse_delta_x != 0 || in->mouse_delta_y != 0)) {              // :898-908
        const double mouse_sensitivity = 4.0f;
        const double yaw_delta = (double)in->mouse_delta_x * mouse_sensitivity * dt;
        const double pitch_delta = -(double)in->mouse_delta_y * mouse_sensitivity * dt;
        PtMat3 yaw, pitch, left, both;
        orc_rotation_matrix(orc_degrees_to_radians(yaw_delta), 0.0f, 0.0f, 1.0f, &yaw);
        orc_rotation_matrix(orc_degrees_to_radians(pitch_delta), 1.0f, 0.0f, 0.0f, &pitch);
        mat_mul(yaw, cam->orientation, &left);
        mat_mul(left, pitch, &both);
        cam->orientation = both;
        modified = 1;
    }
    const float scroll_sensitivity = 0.95f;                                                        // :910-924
    if (in->scroll > 0) {
        for (int i = 0; i < in->scroll; ++i) cam->distance *= scroll_sensitivity;
        modified = 1;
    } else if (in->scroll < 0) {
        for (int i = in->scroll; i < 0; ++i) cam->distance /= scroll_sensitivity;
        modified = 1;
    }
    const double (*m)[3] = cam->orientation.rows;                                                  // :928-929
    const V3 cx{m[0][0], m[1][0], m[2][0]}, cy{m[0][1], m[1][1], m[2][1]};
    const float camera_speed = 0.6f;                                                               // :932-952
    auto move = [&](double sign, const V3& axis) {
        const V3 step = scale(cam->distance * camera_speed * dt, axis);
        const V3 t = (sign < 0.0) ? sub(v3(cam->target), step) : add(v3(cam->target), step);
        cam->target = PtVec3{t.x, t.y, t.z};
        modified = 1;
    };
    if (in->key_a) move(-1.0, cx);
    if (in->key_d) move(+1.0, cx);
    if (in->key_w) move(+1.0, cy);
    if (in->key_s) move(-1.0, cy);
    const double fov_sensitivity = 0.1f, focus_sensitivity = 1.0f, aperture_sensitivity = 0.1f;    // :954-985
    if (in->key_q) { cam->fov_y -= fov_sensitivity * dt; modified = 1; }
    if (in->key_e) { cam->fov_y += fov_sensitivity * dt; modified = 1; }
    if (in->key_up) { cam->focus_distance += focus_sensitivity * dt * cam->distance; modified = 1; }
    if (in->key_down) { cam->focus_distance -= focus_sensitivity * dt * cam->distance; modified = 1; }
    if (in->key_left) { cam->aperture -= aperture_sensitivity * dt; modified = 1; }
    if (in->key_right) { cam->aperture += aperture_sensitivity * dt; modified = 1; }
    return modified;
}

}  // extern "C"
