// Tighter leaf boxes for the search hierarchy: large triangles whose boxes are mostly empty are cut into pieces.
//
// The search hierarchy (sah_build.cpp, the 4-wide nodes made from it) is this library's own: which leaves the
// reference reaches depends only on each leaf's OWN box (checked exactly, once, on the triangle that won), so the
// hierarchy may bound a triangle by any set of boxes that together contain it.  A long thin triangle that lies
// diagonally in its box - the fans that close the chess pieces' bases and tops are the extreme case - fills a few
// percent of that box, and almost every ray that enters the box pays a triangle test (double precision, three
// divisions) for nothing.  Here such a triangle is clipped at the middle of its box's longest axis, again and again,
// each piece bounded by its own box ("early split clipping"): the hierarchy then holds several REFERENCES to one
// triangle.  Boxes of pieces are supersets of the clipped polygon, so every point of the triangle is in one of them.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <queue>
#include <vector>

#include "pt_internal.h"
#include "pt_traverse_host.h"

namespace pt {

namespace {

struct P3 { double x, y, z; };
inline double coord(const P3& p, int axis) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }

struct Polygon {
    P3 v[10];         // a triangle cut by up to six planes... by one plane per split: a convex polygon gains at most one vertex per cut
    int n = 0;
};

PtAabb box_of(const Polygon& p) {
    PtAabb b{{p.v[0].x, p.v[0].y, p.v[0].z}, {p.v[0].x, p.v[0].y, p.v[0].z}};
    for (int i = 1; i < p.n; ++i) {
        b.min.x = std::min(b.min.x, p.v[i].x); b.min.y = std::min(b.min.y, p.v[i].y); b.min.z = std::min(b.min.z, p.v[i].z);
        b.max.x = std::max(b.max.x, p.v[i].x); b.max.y = std::max(b.max.y, p.v[i].y); b.max.z = std::max(b.max.z, p.v[i].z);
    }
    return b;
}

double surface(const PtAabb& b) {
    const double ex = b.max.x - b.min.x, ey = b.max.y - b.min.y, ez = b.max.z - b.min.z;
    return 2.0 * (ex * ey + ey * ez + ez * ex);
}

double area(const Polygon& p) {        // of a planar convex polygon: half the length of the summed cross products of a fan
    double sx = 0, sy = 0, sz = 0;
    for (int i = 1; i + 1 < p.n; ++i) {
        const P3 a{p.v[i].x - p.v[0].x, p.v[i].y - p.v[0].y, p.v[i].z - p.v[0].z};
        const P3 b{p.v[i + 1].x - p.v[0].x, p.v[i + 1].y - p.v[0].y, p.v[i + 1].z - p.v[0].z};
        sx += a.y * b.z - a.z * b.y; sy += a.z * b.x - a.x * b.z; sz += a.x * b.y - a.y * b.x;
    }
    return 0.5 * std::sqrt(sx * sx + sy * sy + sz * sz);
}

// The part of `in` with coordinate <= at (keep_low) or >= at on `axis` (Sutherland-Hodgman, one plane).
Polygon clip(const Polygon& in, int axis, double at, bool keep_low) {
    Polygon out;
    for (int i = 0; i < in.n; ++i) {
        const P3& a = in.v[i];
        const P3& b = in.v[(i + 1) % in.n];
        const double ca = coord(a, axis), cb = coord(b, axis);
        const bool a_in = keep_low ? ca <= at : ca >= at, b_in = keep_low ? cb <= at : cb >= at;
        if (a_in && out.n < 10) out.v[out.n++] = a;
        if (a_in != b_in && out.n < 10) {
            const double t = (at - ca) / (cb - ca);
            P3 c{a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)};
            (axis == 0 ? c.x : (axis == 1 ? c.y : c.z)) = at;      // on the plane exactly: the two sides' boxes meet there
            out.v[out.n++] = c;
        }
    }
    return out;
}

// What a reference's box has beyond the flat piece it bounds: its surface minus the piece's two faces.
struct Candidate {
    double excess;
    int32_t primitive;
    int32_t pieces_left;       // how many more references this triangle may still be cut into
    Polygon shape;
    bool operator<(const Candidate& o) const { return excess < o.excess; }
};

}  // namespace

// References for the search hierarchy: `boxes[i]` bounds a part of triangle `primitive[i]`; every triangle has at least
// one, and the boxes of a triangle's references together contain it.  At most `extra_fraction` * count references are
// added to the `count` there are without cutting.  Returns the number of references added.
long long presplit_triangles(const PtTriangle* triangles, long long count, const PtAabb* own_boxes, double extra_fraction,
                             UnfilledVector<PtAabb>* boxes, std::vector<int32_t>* primitive) {
    boxes->resize(count);
    primitive->resize(count);
    double total = 0.0;
    for (long long i = 0; i < count; ++i) {
        (*boxes)[i] = own_boxes[i];
        (*primitive)[i] = static_cast<int32_t>(i);
        const double s = surface(own_boxes[i]);
        if (std::isfinite(s)) total += s;
    }
    const long long budget = static_cast<long long>(extra_fraction * static_cast<double>(count));
    if (budget <= 0 || count == 0 || !(total > 0.0)) return 0;
    // worth cutting: boxes with more spare surface than a typical box has surface at all
    double floor_factor = 2.0;
    if (const char* v = std::getenv("PT_PRESPLIT_FLOOR")) floor_factor = std::atof(v);
    const double floor_excess = floor_factor * total / static_cast<double>(count);
    std::priority_queue<Candidate> queue;
    for (long long i = 0; i < count; ++i) {
        const PtTriangle& t = triangles[i];
        Polygon p;
        p.n = 3;
        p.v[0] = P3{t.a.x, t.a.y, t.a.z}; p.v[1] = P3{t.b.x, t.b.y, t.b.z}; p.v[2] = P3{t.c.x, t.c.y, t.c.z};
        const double excess = surface(own_boxes[i]) - 2.0 * area(p);
        if (excess > floor_excess && std::isfinite(excess)) queue.push(Candidate{excess, static_cast<int32_t>(i), 63, p});
    }
    // A triangle's first reference keeps the triangle's place in the arrays (and is narrowed in place); later pieces are appended.
    std::vector<char> placed(static_cast<size_t>(count), 0);
    long long added = 0;
    std::vector<Candidate> finished;
    while (!queue.empty()) {
        Candidate c = queue.top();
        queue.pop();
        const PtAabb b = box_of(c.shape);
        const double ex = b.max.x - b.min.x, ey = b.max.y - b.min.y, ez = b.max.z - b.min.z;
        const int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
        const double lo = axis == 0 ? b.min.x : (axis == 1 ? b.min.y : b.min.z), hi = axis == 0 ? b.max.x : (axis == 1 ? b.max.y : b.max.z);
        const double at = 0.5 * (lo + hi);
        bool cut = added < budget && c.pieces_left > 0 && c.excess > floor_excess && at > lo && at < hi;
        Polygon low, high;
        if (cut) {
            low = clip(c.shape, axis, at, true);
            high = clip(c.shape, axis, at, false);
            cut = low.n >= 3 && high.n >= 3;
        }
        if (!cut) {
            finished.push_back(c);
            continue;
        }
        ++added;
        const int32_t share = c.pieces_left - 1;
        queue.push(Candidate{surface(box_of(low)) - 2.0 * area(low), c.primitive, share / 2, low});
        queue.push(Candidate{surface(box_of(high)) - 2.0 * area(high), c.primitive, share - share / 2, high});
    }
    for (const Candidate& c : finished) {
        PtAabb b = box_of(c.shape);
        // the cut points are rounded: a margin far above that rounding, far below anything that would cost visits
        const double mx = 1e-9 * (std::fabs(b.min.x) + std::fabs(b.max.x)) + 1e-30, my = 1e-9 * (std::fabs(b.min.y) + std::fabs(b.max.y)) + 1e-30,
                     mz = 1e-9 * (std::fabs(b.min.z) + std::fabs(b.max.z)) + 1e-30;
        b.min.x -= mx; b.min.y -= my; b.min.z -= mz; b.max.x += mx; b.max.y += my; b.max.z += mz;
        // never beyond the triangle's own box: that one is exact
        const PtAabb& own = own_boxes[c.primitive];
        b.min.x = std::max(b.min.x, own.min.x); b.min.y = std::max(b.min.y, own.min.y); b.min.z = std::max(b.min.z, own.min.z);
        b.max.x = std::min(b.max.x, own.max.x); b.max.y = std::min(b.max.y, own.max.y); b.max.z = std::min(b.max.z, own.max.z);
        if (!placed[c.primitive]) {
            placed[c.primitive] = 1;
            (*boxes)[c.primitive] = b;
        } else {
            boxes->push_back(b);
            primitive->push_back(c.primitive);
        }
    }
    return added;
}

}  // namespace pt
