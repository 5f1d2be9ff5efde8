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
double excess_of(const Polygon& p) { return surface(box_of(p)) - 2.0 * area(p); }

struct Piece { int32_t primitive; PtAabb box; };

// The references of one triangle: the triangle is halved at the middle of its box's longest axis while a piece's box has
// more than `floor_excess` of spare surface, into at most MAX_PIECES pieces.  The first piece goes to *first, the rest
// are appended to *more.  Returns the number of pieces.
constexpr int MAX_PIECES = 64;
int cut_triangle(const PtTriangle& t, int32_t primitive, const PtAabb& own, double floor_excess, double least_gain, PtAabb* first, std::vector<Piece>* more) {
    struct Work { Polygon shape; int pieces; };     // `pieces`: how many references this part may still become
    Work stack[2 * MAX_PIECES];
    int top = 0, made = 0;
    stack[top].shape.n = 3;
    stack[top].shape.v[0] = P3{t.a.x, t.a.y, t.a.z}; stack[top].shape.v[1] = P3{t.b.x, t.b.y, t.b.z}; stack[top].shape.v[2] = P3{t.c.x, t.c.y, t.c.z};
    stack[top++].pieces = MAX_PIECES;
    while (top > 0) {
        const Work w = stack[--top];
        const PtAabb b = box_of(w.shape);
        const double ex = b.max.x - b.min.x, ey = b.max.y - b.min.y, ez = b.max.z - b.min.z;
        const int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
        const double lo = axis == 0 ? b.min.x : (axis == 1 ? b.min.y : b.min.z), hi = axis == 0 ? b.max.x : (axis == 1 ? b.max.y : b.max.z);
        const double at = 0.5 * (lo + hi);
        const double excess = excess_of(w.shape);
        bool cut = w.pieces >= 2 && at > lo && at < hi && excess > floor_excess;
        Polygon low, high;
        if (cut) {
            low = clip(w.shape, axis, at, true);
            high = clip(w.shape, axis, at, false);
            // A cut adds a node to the hierarchy, which costs rays in proportion to the box's surface; it must take away spare
            // surface (rays that pay a triangle test for nothing) in some proportion to that.  Half a diagonal sliver's box is
            // spare and a cut removes half of that; a cut of an axis-parallel right triangle removes a quarter of the box's surface.
            cut = low.n >= 3 && high.n >= 3 && excess - excess_of(low) - excess_of(high) > least_gain * surface(b);
        }
        if (cut) {
            stack[top++] = Work{high, w.pieces - w.pieces / 2};
            stack[top++] = Work{low, w.pieces / 2};
            continue;
        }
        PtAabb piece = b;
        // the cut points are rounded: a margin far above that rounding, far below anything that would cost visits
        const double mx = 1e-9 * (std::fabs(b.min.x) + std::fabs(b.max.x)) + 1e-30, my = 1e-9 * (std::fabs(b.min.y) + std::fabs(b.max.y)) + 1e-30,
                     mz = 1e-9 * (std::fabs(b.min.z) + std::fabs(b.max.z)) + 1e-30;
        piece.min.x -= mx; piece.min.y -= my; piece.min.z -= mz; piece.max.x += mx; piece.max.y += my; piece.max.z += mz;
        // never beyond the triangle's own box: that one is exact
        piece.min.x = std::max(piece.min.x, own.min.x); piece.min.y = std::max(piece.min.y, own.min.y); piece.min.z = std::max(piece.min.z, own.min.z);
        piece.max.x = std::min(piece.max.x, own.max.x); piece.max.y = std::min(piece.max.y, own.max.y); piece.max.z = std::min(piece.max.z, own.max.z);
        if (made++ == 0) *first = piece;
        else more->push_back(Piece{primitive, piece});
    }
    return made;
}

}  // namespace

// References for the search hierarchy: `boxes[i]` bounds a part of triangle `primitive[i]`; every triangle has at least
// one (reference i < count belongs to triangle i; the pieces beyond a triangle's first follow, by triangle), and the boxes
// of a triangle's references together contain it.  A piece is cut while its box has more spare surface than
// `floor_factor` times the mean surface of all triangles' boxes; if that adds more than `extra_fraction` * count
// references, the bar is raised until it does not.  The result does not depend on the number of threads.
// Returns the number of references added.
long long presplit_triangles(const PtTriangle* triangles, long long count, const PtAabb* own_boxes, double floor_factor, double extra_fraction,
                             UnfilledVector<PtAabb>* boxes, std::vector<int32_t>* primitive) {
    boxes->resize(count);
    primitive->resize(count);
    if (count == 0) return 0;
    const int team = count >= (1 << 16) ? host_threads() : 1;
    // summed in one fixed order, so that the bar below (and with it every cut) is the same for any number of threads
    double total = 0.0;
    for (long long i = 0; i < count; ++i) {
        const double s = surface(own_boxes[i]);
        if (std::isfinite(s)) total += s;
    }
    const long long budget = static_cast<long long>(extra_fraction * static_cast<double>(count));
    if (budget <= 0 || !(total > 0.0) || !(floor_factor > 0.0)) {
        for (long long i = 0; i < count; ++i) { (*boxes)[i] = own_boxes[i]; (*primitive)[i] = static_cast<int32_t>(i); }
        return 0;
    }
    double floor_excess = floor_factor * total / static_cast<double>(count);
    // PT_PRESPLIT_GAIN: a cut must also remove this fraction of the piece's box surface in spare surface (measured: 0.3 keeps
    // axis-parallel right triangles whole, which costs the chess set 3 %; 0 = no such condition)
    double least_gain = 0.0;
    if (const char* v = std::getenv("PT_PRESPLIT_GAIN")) least_gain = std::atof(v);
    std::vector<std::vector<Piece>> more(team);
    for (;;) {
        in_parts(0, count, team, [&](long long from, long long to, int part) {
            std::vector<Piece>& mine = more[part];
            mine.clear();
            for (long long i = from; i < to; ++i) {
                (*boxes)[i] = own_boxes[i];
                (*primitive)[i] = static_cast<int32_t>(i);
                const double s = surface(own_boxes[i]);
                if (!(s > floor_excess) || !std::isfinite(s)) continue;        // the spare surface is less than the surface
                cut_triangle(triangles[i], static_cast<int32_t>(i), own_boxes[i], floor_excess, least_gain, &(*boxes)[i], &mine);
            }
        });
        long long added = 0;
        for (const std::vector<Piece>& m : more) added += static_cast<long long>(m.size());
        if (added <= budget) break;
        floor_excess *= 2.0;
    }
    long long added = 0;
    for (const std::vector<Piece>& m : more) {          // in the order of the parts: by triangle, whatever the number of threads
        for (const Piece& p : m) {
            boxes->push_back(p.box);
            primitive->push_back(p.primitive);
            ++added;
        }
    }
    return added;
}

}  // namespace pt
