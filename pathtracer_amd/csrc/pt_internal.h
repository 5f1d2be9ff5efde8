// Shared by the host-side sources of libpathtracer_amd: error reporting, small
// double-precision vector helpers and the host scene container.
#ifndef PT_INTERNAL_H
#define PT_INTERNAL_H

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/pathtracer_amd.h"

namespace pt {

// Records `code` with a printf-style message for pt_last_error() and returns `code`.
int fail(int code, const char* format, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// The reference spells most constants as float literals that widen to double
// (src/main.cpp:27: PI = 3.14159265358979323846264f).
constexpr double PI_F = 3.14159265358979323846264f;

inline PtVec3 operator+(const PtVec3& a, const PtVec3& b) { return PtVec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline PtVec3 operator-(const PtVec3& a, const PtVec3& b) { return PtVec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline PtVec3 operator*(double s, const PtVec3& a) { return PtVec3{s * a.x, s * a.y, s * a.z}; }
inline double dot(const PtVec3& a, const PtVec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline PtVec3 cross(const PtVec3& a, const PtVec3& b) {
    return PtVec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double magnitude(const PtVec3& a) { return std::sqrt(dot(a, a)); }
// src/linear_algebra.cpp:37-40: one reciprocal, then three products.
inline PtVec3 normalise(const PtVec3& a) { return (1.0f / magnitude(a)) * a; }
inline PtVec3 transform(const PtMat3& m, const PtVec3& v) {   // src/linear_algebra.cpp:42-48
    return PtVec3{m.rows[0][0] * v.x + m.rows[0][1] * v.y + m.rows[0][2] * v.z,
                  m.rows[1][0] * v.x + m.rows[1][1] * v.y + m.rows[1][2] * v.z,
                  m.rows[2][0] * v.x + m.rows[2][1] * v.y + m.rows[2][2] * v.z};
}
PtMat3 multiply(const PtMat3& a, const PtMat3& b);            // src/linear_algebra.cpp:50-66
PtMat3 scaling(double x, double y, double z);                 // src/linear_algebra.cpp:72-79
PtMat3 rotation(double angle, double ax, double ay, double az);
inline double degrees_to_radians(double degrees) { return degrees / 180.0f * PI_F; }   // src/main.cpp:28-30

// std::min / std::max argument order as used by the reference for boxes (src/geometry.cpp:39-49, :92-104).
inline double min_of(double a, double b) { return (b < a) ? b : a; }
inline double max_of(double a, double b) { return (a < b) ? b : a; }

PtAabb box_of(const PtTriangle& t);
PtAabb box_of(const PtSphere& s);
// Builds the 2N-1 node tree over `boxes` (src/bvh.cpp:20-58); `threads` > 1 builds disjoint subtrees concurrently.
void build_bvh(const std::vector<PtAabb>& boxes, PtBvhNode* nodes, int threads);
// The sort keys of that build (src/bvh.cpp:33-38) are the box minima: false if one is a NaN, which has no place in any
// order (std::sort is undefined then, in the reference as here).
bool sortable(const std::vector<PtAabb>& boxes);

// Threads the host-side builders use: the machine's, at most 16 (PT_BUILD_THREADS overrides; results do not depend on it).
int host_threads();

// f(lo, hi, part) over [begin, end) cut into `parts` contiguous pieces, one thread each (the caller's included).
template <class F>
void in_parts(long long begin, long long end, int parts, F f) {
    if (parts <= 1) { f(begin, end, 0); return; }
    std::vector<std::thread> team;
    const long long count = end - begin;
    for (int t = 1; t < parts; ++t) {
        team.emplace_back([=, &f]() { f(begin + count * t / parts, begin + count * (t + 1) / parts, t); });
    }
    f(begin, begin + count / parts, 0);
    for (std::thread& t : team) t.join();
}

// std::allocator, except that resize() leaves trivially constructible elements unwritten: a large array that is about to
// be filled by several threads then costs no pass of its own, and its pages are first touched by the threads that fill them.
template <class T>
struct UnfilledAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = UnfilledAllocator<U>; };
    UnfilledAllocator() = default;
    template <class U> UnfilledAllocator(const UnfilledAllocator<U>&) {}
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
template <class T> using UnfilledVector = std::vector<T, UnfilledAllocator<T>>;

// PT_VERBOSE=1: seconds since the previous call, printed to stderr with what was done in them.
struct StageClock {
    const bool on = std::getenv("PT_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void done(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "pathtracer_amd: %-46s %.3f s\n", what, std::chrono::duration<double>(now - last).count());
        last = now;
    }
};

PtMaterial lambertian(double r, double g, double b);
PtMaterial metal(double r, double g, double b, double fuzziness);
PtMaterial dielectric(double refraction_index);
PtMaterial diffuse_light(double r, double g, double b, double power);

}  // namespace pt

// Host-side scene: the arrays a reference `Scene` (src/path_tracing.h:9-22) points into.
struct PtHostScene {
    std::vector<PtMaterial> materials;
    std::vector<PtSphere> spheres;
    std::vector<int32_t> sphere_material_indices;
    std::vector<PtBvhNode> sphere_bvh;
    pt::UnfilledVector<PtTriangle> triangles;
    pt::UnfilledVector<int32_t> triangle_material_indices;
    pt::UnfilledVector<PtBvhNode> triangle_bvh;
    PtColour background_start{0, 0, 0}, background_end{0, 0, 0};
    PtCamera camera{};
    double aspect_ratio = 1.0;
};

#endif
