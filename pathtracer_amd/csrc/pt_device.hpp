// Device-side building blocks of the bounce loop (gfx950, FP64 throughout).
//
// Arithmetic contract: every value that reaches a comparison, a hash or the output is
// computed with the same IEEE binary64 operations, in the same order, as the reference
// renderer computes it on the CPU.  The file is compiled with -ffp-contract=off so no
// multiply-add is ever fused; f64 division and sqrt on gfx950 are correctly rounded.
#ifndef PT_DEVICE_HPP
#define PT_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pathtracer_amd.h"

#define PT_DEV __device__ __forceinline__

namespace ptd {

constexpr int STACK_DEPTH = 32;        // deepest tree the traversal stacks can hold
constexpr int BLOCK_THREADS = 256;
constexpr double REAL_MAX = 1.7976931348623157e308;   // DBL_MAX: MISS distance, src/path_tracing.cpp:109

struct D3 { double x, y, z; };

PT_DEV D3 add3(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV D3 sub3(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV D3 neg3(D3 a) { return D3{-a.x, -a.y, -a.z}; }
PT_DEV D3 scale3(double s, D3 a) { return D3{s * a.x, s * a.y, s * a.z}; }
PT_DEV D3 mul3(D3 a, D3 b) { return D3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }               // src/linear_algebra.cpp:21-23
PT_DEV D3 cross3(D3 a, D3 b) {                                                             // :25-31
    return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PT_DEV D3 unit3(D3 a) {                                                                    // :37-40
    const double inv = 1.0f / sqrt(dot3(a, a));
    return scale3(inv, a);
}
// std::min / std::max as the reference calls them: the first argument survives a NaN in the second.
PT_DEV double std_min(double a, double b) { return (b < a) ? b : a; }
PT_DEV double std_max(double a, double b) { return (a < b) ? b : a; }
PT_DEV bool any_nan(D3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }

struct Ray { D3 o, d; };

// ---------------------------------------------------------------- hashing, src/rng.cpp
PT_DEV uint32_t noise_1d(uint32_t m) {                 // :5-20, NOISE_SEED = 1 (:3)
    m *= 0xB5297A4Du;
    m += 1u;
    m ^= (m >> 8);
    m += 0x68E31DA4u;
    m ^= (m << 8);
    m *= 0x1B56C4E9u;
    m ^= (m >> 8);
    return m;
}
PT_DEV uint32_t noise_3d(int x, int y, int z) {        // :22-28; the reference's int arithmetic wraps
    return noise_1d((uint32_t)x + 198491317u * (uint32_t)y + 6542989u * (uint32_t)z);
}
PT_DEV uint32_t xorshift(uint32_t s) {                 // random_number, :63-69
    s ^= (s << 13);
    s ^= (s >> 17);
    s ^= (s << 5);
    return s;
}
// real_from_rng, :30-32: r / UINT_MAX, correctly rounded.  r / (2^32 - 1) = r 2^-32 + r 2^-64 + r 2^-96 + ...;
// the first term exactly plus the rounded next two, in one fused multiply-add, rounds to the same double
// as the division for every one of the 2^32 values of r (checked exhaustively on the CPU by the tests),
// in 3 instructions instead of the 13 of a double-precision division.
PT_DEV double real_from_rng(uint32_t r) {
    const double d = (double)r;
    return __builtin_fma(d, 0x1p-32, d * (0x1p-64 + 0x1p-96));
}
// static_cast<int>(double) as x86-64 performs it (cvttsd2si): out-of-range and NaN give INT_MIN,
// where the GPU's own conversion would saturate or give 0.
PT_DEV int to_int_x86(double v) {
    return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000;
}
PT_DEV D3 random_unit_vector(D3 seed) {                // :34-61
    const int sx = to_int_x86(1000000000.0f * seed.x);
    const int sy = to_int_x86(1000000000.0f * seed.y);
    const int sz = to_int_x86(1000000000.0f * seed.z);
    uint32_t r = noise_3d(sx, sy, sz);
    r = xorshift(r);
    const double rx = 2.0f * real_from_rng(r) - 1.0f;
    const double ymax = sqrt(1.0f - rx * rx);
    const double ymin = -ymax;
    r = xorshift(r);
    const double ry = (ymax - ymin) * real_from_rng(r) + ymin;
    const double zmax = sqrt(1.0f - rx * rx - ry * ry);
    const double zmin = -zmax;
    r = xorshift(r);
    const double rz = (zmax - zmin) * real_from_rng(r) + zmin;
    return unit3(D3{rx, ry, rz});
}

// ---------------------------------------------------------------- primitives, src/geometry.cpp
// Slab test (:7-37) with the reference's exact min/max selection; `inv` = 1.0f / direction.
PT_DEV void slab_exact(D3 o, D3 inv, const double* __restrict__ box, double& tmin, double& tmax) {
    const double x0 = (box[0] - o.x) * inv.x, x1 = (box[3] - o.x) * inv.x;
    const double y0 = (box[1] - o.y) * inv.y, y1 = (box[4] - o.y) * inv.y;
    const double z0 = (box[2] - o.z) * inv.z, z1 = (box[5] - o.z) * inv.z;
    const double xl = std_min(x0, x1), xh = std_max(x0, x1);
    const double yl = std_min(y0, y1), yh = std_max(y0, y1);
    const double zl = std_min(z0, z1), zh = std_max(z0, z1);
    tmin = std_max(zl, std_max(yl, xl));
    tmax = std_min(zh, std_min(yh, xh));
}
// Same test for rays whose three reciprocals are finite: no product can be NaN then, and on
// NaN-free operands the hardware min/max return the same values as the selections above (a
// zero may differ in sign, which no later comparison can see).
PT_DEV void slab_finite(D3 o, D3 inv, double bx0, double by0, double bz0, double bx1, double by1, double bz1,
                        double& tmin, double& tmax) {
    const double x0 = (bx0 - o.x) * inv.x, x1 = (bx1 - o.x) * inv.x;
    const double y0 = (by0 - o.y) * inv.y, y1 = (by1 - o.y) * inv.y;
    const double z0 = (bz0 - o.z) * inv.z, z1 = (bz1 - o.z) * inv.z;
    tmin = fmax(fmin(z0, z1), fmax(fmin(y0, y1), fmin(x0, x1)));
    tmax = fmin(fmax(z0, z1), fmin(fmax(y0, y1), fmax(x0, x1)));
}

// Triangle record in HBM, 128 B: what intersect(ray, triangle) (:106-133) and unit_normal (:84-90)
// need that does not depend on the ray, precomputed on the host with the same operations:
// a, ab = b-a, ac = c-a, nn = n.n and un = normalise(n) for n = ab^ac.  n itself is formed again on
// the device (nine operations, the same as on the host): the test then reads 80 bytes, five 16-byte
// loads, instead of seven, and the load path, not arithmetic, is what the kernel runs out of first.
struct alignas(128) TriRecord {
    double a[3], ab[3], ac[3], nn, un[3];
    int32_t material;
    int32_t pad_[5];
};
static_assert(sizeof(TriRecord) == 128, "TriRecord is one 128-byte line");

// `Record` is any struct with the fields a, ab, ac (double[3]) and nn.
template <class Record>
PT_DEV bool hit_triangle(const Ray& ray, const Record* __restrict__ t, double& distance) {
    const D3 a{t->a[0], t->a[1], t->a[2]};
    const D3 ab{t->ab[0], t->ab[1], t->ab[2]};
    const D3 ac{t->ac[0], t->ac[1], t->ac[2]};
    const double nn = t->nn;
    const D3 n = cross3(ab, ac);
    const double dn = dot3(ray.d, n);
    const bool parallel = fabs(dn) < 1.0e-6f;
    const double dist = dot3(sub3(a, ray.o), n) / dn;
    const D3 p = add3(ray.o, scale3(dist, ray.d));
    const D3 ap = sub3(p, a);
    const double alpha = dot3(cross3(ab, ap), n) / nn;
    const double beta = dot3(cross3(ap, ac), n) / nn;
    const bool inside = (alpha >= 0.0f) & (beta >= 0.0f) & (alpha + beta <= 1.0f);
    distance = dist;
    return !parallel & (dist >= 1.0e-6f) & inside;
}

// The same test made by the active lanes of a wave together (all of them call it at once), without its two barycentric
// divisions wherever their outcome is certain.  With A and B the numerators and N = n.n > 0, the reference asks for
// fl(A/N) >= 0, fl(B/N) >= 0 and fl(fl(A/N) + fl(B/N)) <= 1.  For 1e-150 <= N <= 1e150:
//   * A < -1e-170 gives a quotient of magnitude >= 1e-320, which does not round to zero: negative, so outside (B alike);
//   * A >= 0 and B >= 0 (a zero of either sign) with fl(A + B) <= fl(N * (1 - 2^-49)): the three divisions and additions
//     involved err by 2^-53 relative each, far inside the 2^-49, so the sum of the quotients is <= 1: inside;
//   * fl(A + B) >= fl(N * (1 + 2^-49)): the sum of the quotients exceeds 1 if both are >= 0, and one of them is negative
//     otherwise: outside.
// A lane none of these settles (a point within 2^-49 of the far edge, a numerator between -1e-170 and 0, N out of range, a
// NaN, for which every comparison is false) makes the whole wave divide; so does nothing else, and the result is the same
// bits as hit_triangle's either way.
template <class Record>
PT_DEV bool hit_triangle_together(const Ray& ray, const Record* __restrict__ t, double& distance) {
    const D3 a{t->a[0], t->a[1], t->a[2]};
    const D3 ab{t->ab[0], t->ab[1], t->ab[2]};
    const D3 ac{t->ac[0], t->ac[1], t->ac[2]};
    const double nn = t->nn;
    const D3 n = cross3(ab, ac);
    const double dn = dot3(ray.d, n);
    const bool parallel = fabs(dn) < 1.0e-6f;
    const double dist = dot3(sub3(a, ray.o), n) / dn;
    const D3 p = add3(ray.o, scale3(dist, ray.d));
    const D3 ap = sub3(p, a);
    const double num_a = dot3(cross3(ab, ap), n), num_b = dot3(cross3(ap, ac), n);
    const bool in_front = dist >= 1.0e-6f;
    const bool in_range = (nn >= 1.0e-150) & (nn <= 1.0e150);
    const bool behind_a = num_a < 0.0, behind_b = num_b < 0.0;
    const bool far_behind = (num_a < -1.0e-170) | (num_b < -1.0e-170);
    const double sum = num_a + num_b;
    const bool below = sum <= nn * (1.0 - 0x1p-49), above = sum >= nn * (1.0 + 0x1p-49);
    const bool surely_inside = in_range & !behind_a & !behind_b & below;
    const bool surely_outside = in_range & (far_behind | above);
    // (a lane whose plane distance or direction rules the triangle out needs no inside test at all)
    const bool settled = surely_inside | surely_outside | parallel | !in_front;
    bool inside = surely_inside;
    if (__ballot(!settled) != 0ull) {
        const double alpha = num_a / nn;
        const double beta = num_b / nn;
        inside = (alpha >= 0.0f) & (beta >= 0.0f) & (alpha + beta <= 1.0f);
    }
    distance = dist;
    return !parallel & in_front & inside;
}

// intersect(ray, sphere), :59-82, followed by the leaf rule of src/path_tracing.cpp:122-125.
PT_DEV bool hit_sphere_raw(const Ray& ray, const double* __restrict__ s, double& lo, double& hi) {
    const D3 oc = sub3(D3{s[0], s[1], s[2]}, ray.o);
    const double tm = dot3(oc, ray.d);
    const double oc2 = dot3(oc, oc);
    const double tm2 = tm * tm;
    const double off2 = oc2 - tm2;
    const double r2 = s[3] * s[3];
    const double h2 = r2 - off2;
    lo = 0.0;
    hi = 0.0;
    if (h2 < 0.0f) return false;
    const double h = sqrt(h2);
    lo = tm - h;
    hi = tm + h;
    return true;
}
PT_DEV bool hit_sphere(const Ray& ray, const double* __restrict__ s, double& distance) {
    double lo, hi;
    if (!hit_sphere_raw(ray, s, lo, hi)) return false;
    if (!(lo > 0.0f || hi > 0.0f)) return false;
    distance = (lo > 0.0f) ? lo : hi;
    return true;
}

// ---------------------------------------------------------------- materials + scatter
struct Mat { double v0, v1, v2, v3; int type; };

// scatter (src/path_tracing.cpp:35-97) with reflect (:7-12), refract (:14-24), reflectance (:27-33).
// `with_metals`, the same for every lane: the scene has a metal among its materials.  Lambertian and metal are then one
// body, `unit(base + k * random_unit_vector(in.d))` from the nudged point, so that lanes of the two kinds run its square
// roots and divisions (most of a shade phase) together instead of one kind after the other.  Same operations on the
// same operands either way.
PT_DEV bool scatter(const Ray& in, const Mat& m, D3 p, D3 n, Ray& out, bool with_metals) {
    const double nudge = 0.001f;
    if (with_metals) {
        if (m.type == PT_LAMBERTIAN || m.type == PT_METAL) {
            const bool metal = m.type == PT_METAL;
            const D3 rnd = random_unit_vector(in.d);
            const D3 refl = sub3(in.d, scale3(2.0f * dot3(in.d, n), n));
            const D3 base = metal ? refl : n;
            const double k = metal ? m.v3 : 0.99f;
            out.o = add3(p, scale3(nudge, n));
            out.d = unit3(add3(base, scale3(k, rnd)));
            return !metal || dot3(out.d, n) > 0.0f;
        }
    } else if (m.type == PT_LAMBERTIAN) {
        const D3 rnd = random_unit_vector(in.d);
        out.o = add3(p, scale3(nudge, n));
        out.d = unit3(add3(n, scale3(0.99f, rnd)));
        return true;
    }
    if (m.type == PT_DIELECTRIC) {
        const bool front = dot3(in.d, n) < 0.0f;
        const double ratio = front ? 1.0f / m.v0 : m.v0;
        const D3 nn = front ? n : neg3(n);
        const double cos_t = dot3(neg3(in.d), nn);
        const double sin_t = sqrt(1.0f - cos_t * cos_t);
        const uint32_t h = noise_1d((uint32_t)(1000000000.0f * sin_t));
        const double threshold = real_from_rng(h);
        const double r = (1.0f - ratio) / (1.0f + ratio);
        const double r0 = r * r;
        const double q = 1.0f - cos_t;
        const double schlick = r0 + (1.0f - r0) * q * q * q * q * q;
        if (ratio * sin_t > 1.0f || schlick > threshold) {
            out.o = add3(p, scale3(nudge, nn));
            out.d = sub3(in.d, scale3(2.0f * dot3(in.d, nn), nn));
        } else {
            out.o = sub3(p, scale3(nudge, nn));
            const D3 perp = scale3(ratio, add3(in.d, scale3(cos_t, nn)));
            const D3 para = scale3(-sqrt(fabs(1.0f - dot3(perp, perp))), nn);
            out.d = unit3(add3(perp, para));
        }
        return true;
    }
    out.o = D3{0.0, 0.0, 0.0};          // DIFFUSE_LIGHT: Maybe<Ray>{} is invalid and zeroed
    out.d = D3{0.0, 0.0, 0.0};
    return false;
}

// ---------------------------------------------------------------- primary rays, src/main.cpp:87-113
struct FrameParams {
    int width, height, row_begin, row_end, sample_begin, sample_count, max_bounces, pad_;
    int shard_index, shard_count;      // row blocks of 8 dealt round-robin; count >= 1
    double aperture;
    D3 pos, cx, cy, bl, sx, sy;
};

PT_DEV Ray camera_ray(const FrameParams& f, int row, int column, int sample) {
    const double lens = 0.5f * f.aperture;
    uint32_t r = noise_3d(row, column, sample);
    r = xorshift(r);
    const double u = ((double)column + real_from_rng(r)) / (double)(f.width - 1);
    r = xorshift(r);
    const double v = ((double)row + real_from_rng(r)) / (double)(f.height - 1);
    r = xorshift(r);
    const double a = f.aperture * real_from_rng(r) - lens;
    const double bmax = sqrt(lens * lens - a * a);
    const double bmin = -bmax;
    r = xorshift(r);
    const double b = (bmax - bmin) * real_from_rng(r) + bmin;
    const D3 off = add3(scale3(a, f.cx), scale3(b, f.cy));
    const D3 d = unit3(sub3(sub3(add3(add3(f.bl, scale3(u, f.sx)), scale3(v, f.sy)), f.pos), off));
    return Ray{add3(f.pos, off), d};
}

}  // namespace ptd

#endif
