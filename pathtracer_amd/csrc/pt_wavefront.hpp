// The persistent wavefront kernel: paths are decoupled from pixels.
//
// Why: the reference sums a pixel's samples in ascending order (src/main.cpp:117-121 run once
// per frame, src/main.cpp:1001-1036), and f64 addition does not commute with reordering, so a
// lane that owns a pixel must walk its samples one after another; lanes of one wave then run
// out of step (most primary rays leave the scene at once, a few paths bounce 50 times) and the
// wave idles.  A path's colour, however, is a pure function of (row, column, sample): it can
// be computed by any lane at any time.  So:
//
//   trace_paths_kernel   persistent waves draw 1024 paths at a time (one sample each of 16
//                        neighbouring 8x8 tiles) from a counter in HBM and write each path's colour
//                        to `rgb`; a lane that finishes a path takes the next one, so all 64
//                        lanes stay busy until the frame runs out of work.  Inside the wave,
//                        lanes are regrouped with ballots: node steps, leaf tests and
//                        shading each run when enough lanes want them, never one lane at a time.
//   sum_samples_kernel   one lane per pixel adds that pixel's colours in ascending sample
//                        order into the accumulation buffer: the reference's order, bit for bit.
//
// Requires every albedo finite and |albedo| <= 1: then the attenuation stays finite, each
// vertex before the last adds exactly +-0 to the path's colour, and the colour is 0.0 plus
// the last vertex's term.  Other scenes use the one-lane-per-pixel kernel.
#ifndef PT_WAVEFRONT_HPP
#define PT_WAVEFRONT_HPP

#include "pt_traverse.hpp"

namespace ptd {

#ifndef PT_SAMPLES_PER_CHUNK
#define PT_SAMPLES_PER_CHUNK 16
#endif
constexpr int SAMPLES_PER_CHUNK = PT_SAMPLES_PER_CHUNK;
constexpr int CHUNK_PATHS = 64 * SAMPLES_PER_CHUNK;
constexpr int MAX_SEARCH_HEIGHT = 32;  // levels the rebuilt binary hierarchy is kept to (= STACK_DEPTH: a height cap near the minimum forces balanced splits and costs large scenes a third of their speed)
constexpr int MAX_WF_STACK_SLOTS = 36; // per-lane stack slots a block is given at most (1 KB of LDS each): three blocks per CU still fit
constexpr int QUAD_STACK_SLOTS = 24;   // what the search through four-wide nodes is given when the launch has a backing store for
                                       // its stacks: four blocks per CU fit
constexpr int PARK_UNIT = 12;          // entries a lane whose stack is full moves to the backing store at a time (the oldest ones)
constexpr int PARKED_SHIFT = 24;       // a lane's count of parked units lives in its bounce counter above this bit

struct WorkParams {
    int tiles_x;                 // 8x8 tiles across the image
    int groups_per_tile;         // ceil(sample_count / SAMPLES_PER_CHUNK)
    unsigned int tile_count;     // tiles of this launch (this GPU's shard)
    unsigned int unit_count;     // blocks of 4x4 tiles * groups_per_tile * 256: units of 64 paths on the counter
    int shade_threshold;         // lanes that must be waiting before a shade phase runs while others still search
    int leaf_threshold;          // same for a leaf phase while others still have inner nodes to visit
    int stack_slots;             // per-lane stack slots the launch provides in LDS
    int parking;                 // 1: a lane whose stack is full moves its oldest entries to TraceArgs::park and carries on;
                                 // 0: it gives the ordered search up and is resolved by the exhaustive traversal
    int distrust_mask;           // test aid: a winner whose index has none of these bits set is sent to the exhaustive
                                 // traversal as if its box test had failed (negative = never)
    int export_hits;             // the launch reports each path's first-vertex triangle and sphere index (TraceArgs::hit_*):
                                 // the two searches then may not shorten each other (the reference runs both in full,
                                 // src/path_tracing.cpp:177-178, and only then compares, :183)
};

// rgb layout: [chunk][CHUNK_PATHS][r, g, b, 0] doubles, one 32-byte record per path, written whole
// (two 16-byte stores that fill one 32-byte sector); chunk = tile * groups + sample group, and path j
// of a chunk is sample (j / 64) of that group at pixel (j % 64) of the tile.  (The order in which the
// paths are handed out is another matter: see the refill of the kernel.)
PT_DEV long long rgb_index(long long chunk, int channel, int j) { return (chunk * CHUNK_PATHS + j) * 4 + channel; }

// Everything the kernel is given, as one argument: the kernel-argument segment then holds exactly this
// struct, and code that runs once in a while can fetch what it needs from there when it needs it.
struct TraceArgs {
    DeviceScene s;
    FrameParams f;
    WorkParams w;
    double* rgb;
    unsigned int* next_unit;
    unsigned long long* stats;
    int* hit_triangle;           // per pixel (row * width + column), written at a path's first vertex; null unless w.export_hits
    int* hit_sphere;
    int* park;                   // backing store of the lanes' stacks, [entry][thread of the grid]; null unless w.parking
};
// The argument struct behind a pointer the compiler must treat as new at this point: what is read through
// it is fetched here, by scalar loads, instead of being kept in scalar registers across the whole work
// loop, where the node and leaf steps need those registers.
typedef const __attribute__((address_space(4))) TraceArgs* ArgsInMemory;
PT_DEV ArgsInMemory args_fetched_here() {
    ArgsInMemory p = (ArgsInMemory)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

enum LaneState { ST_NEED = 0, ST_LEAF = 1, ST_SHADE = 2, ST_NEWRAY = 3, ST_NODE = 4 };

struct WaveStats {   // wave-uniform counters of a profiling build
    unsigned long long node_phases, node_lanes, leaf_phases, leaf_lanes, shade_phases, shade_lanes, loops, paths;
    unsigned long long unsafe_skips;   // f32 test skipped a box the exact test accepts: must stay 0
    unsigned long long t_node, t_leaf, t_shade, t_refill, t_setup, t_total;   // shader-clock ticks spent per section
};

// LDS: the tournament frames of the exhaustive fallback (static) and the lanes' stacks of pending
// subtrees, [slot][thread], sized at launch to what the scene's hierarchy can need (dynamic).
struct WaveShared {
    TourFrame frames[(BLOCK_THREADS / 64) * (STACK_DEPTH + 1)];
    double pool[BLOCK_THREADS / 64][6][64];     // per wave: the camera rays of the next 64 paths (origin, direction)
};

// STATS: 0 = none; 1 = phase/lane counters and the f32-versus-f64 box-test cross-check; 2 = the
// counters and shader-clock time per section (no cross-check, so the times are the product's).
// MERGED (with NODE32 only): the hierarchy holds the spheres as well as the triangles.
// NODE32: inner nodes are the 128-byte single-precision nodes with four children (conservative
// test; the exact test of the winning primitive's own box is made before the path is shaded);
// otherwise the 128-byte double-precision pairs, whose test is the reference's own.
// Waves per SIMD the registers are budgeted for: four (128 registers) with the four-wide nodes, whose stacks leave room for
// four blocks per CU in LDS; the double-precision pairs need their tree's full height in stack slots, so three at most, and
// the counting and timing builds carry more state than 128 registers hold.
#ifndef PT_REPEAT_LANES
#define PT_REPEAT_LANES 16      // lanes that must still stand on a node for an iteration's second node step
#endif
#ifndef PT_WAVES_PER_SIMD
#define PT_WAVES_PER_SIMD 4
#endif
template <int STATS, bool NODE32, bool MERGED>
__global__ __launch_bounds__(BLOCK_THREADS, (NODE32 && STATS == 0) ? PT_WAVES_PER_SIMD : 3) void trace_paths_kernel(TraceArgs args) {
    const DeviceScene& s = args.s;
    const WorkParams& w = args.w;
    unsigned long long* const __restrict__ stats = args.stats;
    __shared__ WaveShared sh;
    extern __shared__ int lane_stacks[];
    const int lane = threadIdx.x & 63;
    int* const stack_ref = lane_stacks + threadIdx.x;
    TourFrame* const frames = sh.frames + (threadIdx.x >> 6) * (STACK_DEPTH + 1);
    double (*const pool)[64] = sh.pool[threadIdx.x >> 6];
    const TriLeaf tri_leaf{s.tris};
    const SphLeaf sph_leaf{s.spheres};

    // wave-uniform work state: the units of 64 paths drawn and where the current one's tile lies
    unsigned int unit_next = 0, unit_end = 0, draw_tile_row0 = 0, draw_tile_col0 = 0;   // the units drawn and not yet started; the draw's first tile
    int unit_row0 = 0, unit_col0 = 0, draw_group = 0, draw_shift = 0;
    bool work_left = true;          // units may remain on the counter
    long long pool_first = 0;       // colour-scratch slot of the pool's first path
    int pool_taken = 64;            // entries of the pool below this are gone
    WaveStats ws{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = (STATS == 2) ? clock64() : 0;
    long long t0 = 0;

    // per-lane path state
    int st = ST_NEED;
    int cur = 0, sp = 0, bounce = 0, best_i = -1;
    long long slot = 0;                 // place of the lane's path in the colour scratch: unit * 64 + pixel of the tile
    bool tie = false, tie_s = false;    // a second triangle / sphere at exactly the best triangle / sphere distance
    double best_t = REAL_MAX, cut = REAL_MAX;
    float cut32 = 0.0f;
    Hit hs{-1, REAL_MAX};
    Ray ray{D3{0, 0, 0}, D3{0, 0, 1}};
    D3 inv{0, 0, 1}, att{1.0f, 1.0f, 1.0f};
    Ray32 r32{0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0};

    // The reference's box test (src/geometry.cpp:7-37).  Its min/max argument order matters only when a
    // slab product is NaN, which needs an infinite reciprocal or a NaN in the ray or the box.  A scene
    // that uses the single-precision hierarchy has finite boxes only (checked when it is built): there,
    // when no active lane's ray has such a value, the hardware min/max give the same values with half
    // the instructions.
    auto box_test = [&](D3 origin, D3 reciprocal, const double* box, double& tmin, double& tmax) {
        const bool plain = isfinite(reciprocal.x + reciprocal.y + reciprocal.z) && isfinite(origin.x + origin.y + origin.z);
        if (NODE32 && __ballot(!plain) == 0ull) slab_finite(origin, reciprocal, box[0], box[1], box[2], box[3], box[4], box[5], tmin, tmax);
        else slab_exact(origin, reciprocal, box, tmin, tmax);
    };

    // Leaves the current node: takes the pending subtree on top of the stack, or ends the search.
    auto pop_once = [&]() {
        const int top = max(sp - 1, 0);
        const int pending = stack_ref[top * BLOCK_THREADS];
        st = (sp == 0) ? ST_SHADE : ((pending < 0) ? ST_LEAF : ST_NODE);
        sp = top;
        cur = pending;
    };

    // One visit of a four-wide node by the lanes that are active here: four conservative box tests, the
    // entered children pushed nearest on top, the top taken as the lane's next stop.
    auto quad_step = [&](const int node) {
        Float4 low, up;
        Int4 children;
        slab32_quad<!MERGED>(r32, s.tri_quads, node, low, up, children);
        const int c0 = children.x, c1 = children.y, c2 = children.z, c3 = children.w;
        // an empty slot holds NaN planes: every comparison below is false for it
        const bool go0 = enter32(low.x, up.x, cut32), go1 = enter32(low.y, up.y, cut32);
        const bool go2 = enter32(low.z, up.z, cut32), go3 = enter32(low.w, up.w, cut32);
        if (STATS == 1) {
            const QuadBoxes64* e = s.tri_quad_boxes + node;
            const bool go[4] = {go0, go1, go2, go3};
            bool unsafe = false;
            for (int i = 0; i < 4; ++i) {
                double tmin, tmax;
                slab_finite(ray.o, inv, e->box[i][0], e->box[i][1], e->box[i][2], e->box[i][3], e->box[i][4], e->box[i][5], tmin, tmax);
                const bool exact = (s.tri_quads[node].child[i] != 0) && !(tmax < 0.0f || tmin > tmax) && !(tmin > cut);
                unsafe = unsafe || (exact && !go[i]);
            }
            ws.unsafe_skips += __popcll(__ballot(unsafe));
        }
        // The four children are put in descending order of entry distance, skipped ones last, by a five-comparator network
        // on (distance, reference) pairs, and written to the four slots above the pending entries as they then stand: the
        // farthest entered child lowest, the nearest on top, skipped ones above the new top (where nothing is read).  The
        // slots are fixed distances from one address, so the four writes need no address arithmetic of their own.
        // Entry distances of entered children are finite (finite planes and reciprocals).
        const float none = -__builtin_inff();
        float k0 = go0 ? low.x : none, k1 = go1 ? low.y : none, k2 = go2 ? low.z : none, k3 = go3 ? low.w : none;
        int v0 = c0, v1 = c1, v2 = c2, v3 = c3;
        auto farther_first = [](float& ka, int& va, float& kb, int& vb) {
            const bool swap = kb >= ka;      // (of two children entered at the same distance the one in the lower slot is visited first)
            const float larger = fmaxf(ka, kb), smaller = fminf(ka, kb);
            const int of_larger = swap ? vb : va, of_smaller = swap ? va : vb;
            ka = larger; kb = smaller; va = of_larger; vb = of_smaller;
        };
        farther_first(k0, v0, k1, v1);
        farther_first(k2, v2, k3, v3);
        farther_first(k0, v0, k2, v2);
        farther_first(k1, v1, k3, v3);
        farther_first(k1, v1, k2, v2);
        const int count = (go0 ? 1 : 0) + (go1 ? 1 : 0) + (go2 ? 1 : 0) + (go3 ? 1 : 0);
        typedef __attribute__((address_space(3))) int* LdsInts;
        static_assert(BLOCK_THREADS * 4 == 1 << 10, "a stack slot of the block is 1 KB");
        const LdsInts frame = (LdsInts)(unsigned long long)((unsigned)(unsigned long long)(LdsInts)stack_ref + (unsigned)sp * (BLOCK_THREADS * 4u));
        frame[0] = v0;
        frame[BLOCK_THREADS] = v1;
        frame[2 * BLOCK_THREADS] = v2;
        frame[3 * BLOCK_THREADS] = v3;
        sp += count;
        const int top = max(sp - 1, 0);
        const int next = stack_ref[top * BLOCK_THREADS];
        // The launch may give a lane fewer slots than the deepest search needs (LDS is what limits the
        // waves per CU).  A lane whose next visit could outgrow its stack waits for the next shade phase with
        // entries still pending (sp > 0, which no finished search has).  That phase moves its oldest entries to the
        // backing store (w.parking); without one the lane gives up the ordered search there and is resolved like a
        // tie, by the exhaustive traversal.
        const bool overflow = sp + 3 > w.stack_slots;
        st = ((sp == 0) | overflow) ? ST_SHADE : ((next < 0) ? ST_LEAF : ST_NODE);
        sp = top;
        cur = next;
    };

    for (;;) {
        const unsigned long long m_node = __ballot(st == ST_NODE);
        const unsigned long long m_leaf = __ballot(st == ST_LEAF);
        const bool paths_left = work_left || pool_taken < 64;
        // no lane is in ST_NEWRAY here (the set-up below resolves it at once): whoever is neither on a node nor on a leaf is
        // idle or waits to be shaded, and an idle lane is as good as a waiting one while paths are left to hand out
        unsigned long long m_ready = ~(m_node | m_leaf);
        if (!paths_left) m_ready = __ballot(st == ST_SHADE);
        const int n_ready = __popcll(m_ready);
        if (STATS) ++ws.loops;

        const bool shade_now = (m_node | m_leaf) == 0ull || n_ready >= w.shade_threshold;
        if (shade_now) {
            if (n_ready == 0) break;      // nothing in flight, nothing to shade, no work left
            if (STATS) { ++ws.shade_phases; ws.shade_lanes += n_ready; }
            if (STATS == 2) t0 = clock64();
            const ArgsInMemory fetched = args_fetched_here();
            const FrameParams& f = *(const FrameParams*)&fetched->f;      // camera and frame: used by this branch only
            const DeviceScene& scene_for_shading = *(const DeviceScene*)&fetched->s;   // materials, normals, background
            // ---------------- lanes whose stack ran full, or ran empty with entries in the backing store, search on
            if (NODE32 && fetched->w.parking == 0) tie = tie | ((st == ST_SHADE) & (sp > 0));      // no backing store: see quad_step
            if (NODE32 && fetched->w.parking != 0) {
                int parked = bounce >> PARKED_SHIFT;
                if (st == ST_SHADE && (sp > 0 || parked > 0)) {
                    const long long threads = (long long)gridDim.x * BLOCK_THREADS;
                    int* const mine = fetched->park + ((long long)blockIdx.x * BLOCK_THREADS + threadIdx.x);
                    if (sp > 0) {
                        // full (quad_step left sp within three of the end): the oldest PARK_UNIT entries go out, the rest move down,
                        // and the lane carries on with the stop it had already taken
                        for (int k = 0; k < PARK_UNIT; ++k) mine[(parked * PARK_UNIT + k) * threads] = stack_ref[k * BLOCK_THREADS];
                        for (int k = PARK_UNIT; k < sp; ++k) stack_ref[(k - PARK_UNIT) * BLOCK_THREADS] = stack_ref[k * BLOCK_THREADS];
                        sp -= PARK_UNIT;
                        ++parked;
                        st = (cur < 0) ? ST_LEAF : ST_NODE;
                    } else {
                        // empty, but older entries wait outside: the unit parked last comes back
                        --parked;
                        for (int k = 0; k < PARK_UNIT; ++k) stack_ref[k * BLOCK_THREADS] = mine[(parked * PARK_UNIT + k) * threads];
                        sp = PARK_UNIT;
                        pop_once();
                    }
                    bounce = (bounce & ((1 << PARKED_SHIFT) - 1)) | (parked << PARKED_SHIFT);
                }
            }
            // ---------------- shade the lanes whose search is over (src/path_tracing.cpp:179-223)
            if (st == ST_SHADE) {
                Hit ht{best_i, best_t};
                constexpr bool merged = NODE32 && MERGED;
                // (with the single-precision nodes the reciprocals are not kept while a ray is searched: see the set-up below)
                const D3 inv_now = NODE32 ? D3{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z} : inv;
                if (NODE32 && best_i >= 0) {
                    // The reference reaches a triangle only through its leaf's box (src/path_tracing.cpp:113-127).
                    // If the winner's box fails the exact test, some other triangle (or none) is the answer: the
                    // exhaustive traversal finds it.  A ray that traversal has handled already (infinite
                    // reciprocals) arrives here with a box it has tested itself.
                    const LeafBox own = fetch_leaf_box(s.tri_boxes + best_i);
                    double tmin, tmax;
                    box_test(ray.o, inv_now, own.box, tmin, tmax);
                    tie = tie | (tmax < 0.0f) | (tmin > tmax) | ((w.distrust_mask >= 0) & ((best_i & w.distrust_mask) == 0));
                }
                if (merged) {
                    // Spheres found by the same search: the same check on the winning sphere's box.  Either
                    // search was pruned with the other's hits, so a doubt about one sends both to the
                    // exhaustive traversals.
                    if (hs.index >= 0) {
                        const LeafBox own = fetch_leaf_box(s.tri_boxes + s.sphere_base + hs.index);
                        double tmin, tmax;
                        box_test(ray.o, inv_now, own.box, tmin, tmax);
                        tie_s = tie_s | (tmax < 0.0f) | (tmin > tmax) | ((w.distrust_mask >= 0) & ((hs.index & w.distrust_mask) == 0));
                    }
                    tie = tie | tie_s;
                    if (__ballot(tie) != 0ull) {
                        resolve_slow_lanes(tie, ray, s.sph_tree, sph_leaf, frames, hs);
                        if (s.tri_tree.count > 0) resolve_slow_lanes(tie, ray, s.tri_tree, tri_leaf, frames, ht);
                    }
                } else {
                    resolve_slow_lanes(tie, ray, s.tri_tree, tri_leaf, frames, ht);
                }
                if (fetched->w.export_hits != 0 && bounce == 0) {
                    // `ClosestShapeIntersection.index` of both traversals at the camera ray's vertex (src/path_tracing.cpp:177-178)
                    const int groups = fetched->w.groups_per_tile, tiles_x = fetched->w.tiles_x;
                    const int tile = (int)((slot >> 6) / (groups * SAMPLES_PER_CHUNK));
                    const int tile_row = tile / tiles_x, pixel = (int)(slot & 63);
                    const int row = f.row_begin + (tile_row * f.shard_count + f.shard_index) * 8 + (pixel >> 3);
                    const int column = (tile - tile_row * tiles_x) * 8 + (pixel & 7);
                    const long long at = (long long)row * f.width + column;
                    fetched->hit_triangle[at] = ht.index;
                    fetched->hit_sphere[at] = hs.index;
                }
                D3 gain;
                const bool go_on = shade_vertex(scene_for_shading, ray, hs, ht, att, gain);
                ++bounce;
                if (!go_on || bounce >= f.max_bounces) {
                    // every earlier vertex added +-0 (see the file header): colour = 0 + last term
                    double* record = fetched->rgb + 4 * slot;
                    // written once, read once by the other kernel: kept from displacing the scene in the caches
                    Double2* out = reinterpret_cast<Double2*>(record);
                    __builtin_nontemporal_store(Double2{0.0f + gain.x, 0.0f + gain.y}, out);
                    __builtin_nontemporal_store(Double2{0.0f + gain.z, 0.0}, out + 1);
                    st = ST_NEED;
                } else {
                    st = ST_NEWRAY;
                }
            }
            if (STATS == 2) { const long long t1 = clock64(); ws.t_shade += t1 - t0; t0 = t1; }
            // ---------------- hand new paths to idle lanes
            // Camera rays are made 64 at a time by the whole wave (all lanes busy, whatever their own path is
            // doing) and parked in LDS; an idle lane then only picks one up.
            const WorkParams& cold = *(const WorkParams*)&fetched->w;       // how work is dealt: read when a unit is drawn
            for (;;) {
                const unsigned long long m_need = __ballot(st == ST_NEED);
                if (m_need == 0ull) break;
                if (pool_taken == 64) {
                    // Work is dealt in units of 64 paths (one sample of one 8x8 tile), 16 units per draw from the
                    // counter.  Tiles differ a lot in cost (a pixel whose paths get inside a closed mesh bounces to
                    // the limit every time), so the 16 units of a draw are 16 DIFFERENT tiles, one sample each:
                    // a block of 4x4 neighbouring tiles x the 16 samples of a group form a square of 256 units, and
                    // draw d of the square takes sample (k + d) % 16 of its tile k.  A costly tile then weighs on 16 draws
                    // by a sixteenth each instead of on one draw entirely.
                    if (unit_next == unit_end) {
                        if (!work_left) break;
                        unsigned int first = 0;
                        if (lane == 0) first = atomicAdd(fetched->next_unit, (unsigned int)SAMPLES_PER_CHUNK);
                        first = __builtin_amdgcn_readfirstlane(first);
                        if (first >= cold.unit_count) { work_left = false; break; }
                        unit_next = first;
                        unit_end = first + (unsigned int)SAMPLES_PER_CHUNK;
                        const unsigned int draw = first / SAMPLES_PER_CHUNK;
                        const unsigned int square = draw / SAMPLES_PER_CHUNK;
                        draw_shift = (int)(draw % SAMPLES_PER_CHUNK);
                        const unsigned int tile_block = square / (unsigned int)cold.groups_per_tile;
                        draw_group = (int)(square - tile_block * (unsigned int)cold.groups_per_tile);
                        const unsigned int blocks_x = ((unsigned int)cold.tiles_x + 3u) / 4u;
                        const unsigned int block_row = tile_block / blocks_x;
                        draw_tile_row0 = block_row * 4u;
                        draw_tile_col0 = (tile_block - block_row * blocks_x) * 4u;
                    }
                    const unsigned int k = unit_next % SAMPLES_PER_CHUNK;
                    ++unit_next;
                    const unsigned int tile_row = draw_tile_row0 + (k >> 2), tile_col = draw_tile_col0 + (k & 3u);
                    const unsigned int tile = tile_row * (unsigned int)cold.tiles_x + tile_col;
                    const int in_group = (int)((k + (unsigned int)draw_shift) % SAMPLES_PER_CHUNK);
                    const int sample = draw_group * SAMPLES_PER_CHUNK + in_group;
                    // the squares at the right and upper edge and the last group of a sample range may be partial
                    if (tile_col >= (unsigned int)cold.tiles_x || tile >= cold.tile_count || sample >= f.sample_count) continue;
                    unit_col0 = (int)tile_col * 8;
                    unit_row0 = f.row_begin + ((int)tile_row * f.shard_count + f.shard_index) * 8;
                    const unsigned int unit_in_scratch = (tile * (unsigned int)cold.groups_per_tile + (unsigned int)draw_group) * SAMPLES_PER_CHUNK + (unsigned int)in_group;
                    // lane l makes the ray of the tile's pixel l (pixels of a partial tile that lie outside the
                    // image get a ray too; nobody will pick it up)
                    const int row = min(unit_row0 + (lane >> 3), f.height - 1);
                    const int column = min(unit_col0 + (lane & 7), f.width - 1);
                    const Ray made = camera_ray(f, row, column, f.sample_begin + sample);
                    pool[0][lane] = made.o.x; pool[1][lane] = made.o.y; pool[2][lane] = made.o.z;
                    pool[3][lane] = made.d.x; pool[4][lane] = made.d.y; pool[5][lane] = made.d.z;
                    pool_first = (long long)unit_in_scratch * 64;
                    pool_taken = 0;
                }
                const int rank = __popcll(m_need & ((1ull << lane) - 1ull));
                const int entry = pool_taken + rank;
                if (st == ST_NEED && entry < 64) {
                    const int row = unit_row0 + (entry >> 3);
                    const int column = unit_col0 + (entry & 7);
                    if (row < f.row_end && column < f.width) {
                        slot = pool_first + entry;
                        ray = Ray{D3{pool[0][entry], pool[1][entry], pool[2][entry]}, D3{pool[3][entry], pool[4][entry], pool[5][entry]}};
                        att = D3{1.0f, 1.0f, 1.0f};
                        bounce = 0;
                        st = ST_NEWRAY;
                    }
                    // a pixel outside the image (partial tile) is skipped; the lane asks again
                }
                const int taken = min(__popcll(m_need), 64 - pool_taken);
                if (STATS) ws.paths += taken;
                pool_taken += taken;
            }
            if (STATS == 2) { const long long t1 = clock64(); ws.t_refill += t1 - t0; t0 = t1; }
            // ---------------- start the search of every new ray (src/path_tracing.cpp:177-178)
            // With the single-precision nodes a search needs the ray's reciprocals and the single-precision ray only in
            // the form `r32`, and shading needs neither; both are pure functions of the ray.  They are made again here
            // for EVERY lane, whatever its state (same values for a lane in mid-search, unused ones for an idle lane),
            // so that neither occupies registers while the code above runs: that is where the kernel needs most.
            if (NODE32) {
                inv = D3{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
                r32 = make_ray32(ray, inv);
            }
            if (st == ST_NEWRAY) {
                if (!NODE32) inv = D3{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
                constexpr bool merged = NODE32 && MERGED;
                int kind = NODE32 ? classify_ray_of_bounded_origin(ray, inv) : classify_ray(ray, inv);
                if (NODE32 && kind == RAY_FINITE && !ray32_allowed(ray, inv)) kind = RAY_EXACT_ONLY;
                hs = Hit{-1, REAL_MAX};
                if (s.sph_tree.count == 1) {
                    // one sphere: its node is the root and a leaf; the reference tests the box, then the sphere
                    // (src/path_tracing.cpp:113-127); min/max order matters only if a slab product is NaN
                    // (both tests are pure: the sphere's comes first here because most rays fail it, and the box's is then
                    // made by the lanes that passed, if any)
                    double d;
                    if (hit_sphere(ray, s.spheres, d)) {
                        double tmin, tmax;
                        box_test(ray.o, inv, s.sph_tree.root_box, tmin, tmax);
                        if (!(tmax < 0.0f || tmin > tmax)) hs = Hit{0, d};
                    }
                } else if (!merged) {
                    // spheres outside the search hierarchy: the lane walks the sphere tree on its own
                    hs = trace_tree(ray, s.sph_tree, sph_leaf, REAL_MAX, stack_ref, frames);
                } else if (__ballot(kind == RAY_EXACT_ONLY) != 0ull) {
                    // a ray the hierarchy cannot take: the reference's own traversal of the sphere tree (a ray with a NaN
                    // hits nothing, src/geometry.cpp:59-82)
                    resolve_slow_lanes(kind == RAY_EXACT_ONLY, ray, s.sph_tree, sph_leaf, frames, hs);
                }
                best_i = -1;
                best_t = REAL_MAX;
                tie = false;
                tie_s = false;
                sp = 0;
                st = ST_SHADE;                       // unless a search is started below
                if (kind == RAY_EXACT_ONLY && s.tri_tree.count > 0) {
                    Hit ht{-1, REAL_MAX};
                    resolve_slow_lanes(true, ray, s.tri_tree, tri_leaf, frames, ht);
                    best_i = ht.index;
                    best_t = ht.distance;
                } else if (kind == RAY_FINITE && (NODE32 ? s.search_count : s.tri_tree.count) > 0) {
                    // a triangle beyond the sphere hit cannot win (:183): the search may stop there (unless the
                    // launch reports both indices: then the triangle search runs as if there were no sphere)
                    cut = prune_cut(fetched->w.export_hits != 0 ? REAL_MAX : hs.distance);
                    if (NODE32) {
                        // Every new ray starts at the same node, so it is visited here and now: the lanes all read
                        // the same 128 bytes (one access per load instead of one per lane), and the search needs
                        // one node phase less.  The hierarchy's own box is not tested: its children's are.
                        cut32 = cut32_of(cut);
                        quad_step(0);
                    } else {
                        double tmin, tmax;
                        const double* rb = s.tri_tree.root_box;
                        slab_finite(ray.o, inv, rb[0], rb[1], rb[2], rb[3], rb[4], rb[5], tmin, tmax);
                        if (!(tmax < 0.0f || tmin > tmax || tmin > cut)) {
                            cur = s.tri_tree.root;
                            st = (cur < 0) ? ST_LEAF : ST_NODE;
                        }
                    }
                }
            }
            if (STATS == 2) ws.t_setup += clock64() - t0;
        }
        // (a node step follows a shade phase in the same iteration: the rays it set up have visited the first node and mostly stand on one)
        {
            // ---------------- one inner-node step: both children's boxes, nearer child first
            if (STATS == 2) t0 = clock64();
            // Straight-line code: all four children are written to the stack, skipped ones above its new top
            // (the launch sizes the stack for it), and every decision is a select.
            if (STATS) { const int on_a_node = __popcll(__ballot(st == ST_NODE)); ws.node_phases += on_a_node > 0 ? 1 : 0; ws.node_lanes += on_a_node; }
            if (st == ST_NODE) {
                if (NODE32) {
                    quad_step(cur);
                } else {
                    const PairNode* p = s.tri_tree.pairs + cur;
                    const int lref = p->left, rref = p->right;
                    double lmin, lmax, rmin, rmax;
                    slab_finite(ray.o, inv, p->lbox[0], p->lbox[1], p->lbox[2], p->lbox[3], p->lbox[4], p->lbox[5], lmin, lmax);
                    slab_finite(ray.o, inv, p->rbox[0], p->rbox[1], p->rbox[2], p->rbox[3], p->rbox[4], p->rbox[5], rmin, rmax);
                    const bool go_l = !(lmax < 0.0f || lmin > lmax) && !(lmin > cut);
                    const bool go_r = !(rmax < 0.0f || rmin > rmax) && !(rmin > cut);
                    const bool left_first = !(rmin < lmin);
                    const bool both = go_l && go_r;
                    const int near = (both ? left_first : go_l) ? lref : rref;
                    stack_ref[sp * BLOCK_THREADS] = left_first ? rref : lref;
                    sp += both ? 1 : 0;
                    const bool stuck = !(go_l || go_r);
                    const int top = max(sp - 1, 0);
                    const int pending = stack_ref[top * BLOCK_THREADS];
                    const int next = stuck ? pending : near;
                    st = (stuck && sp == 0) ? ST_SHADE : ((next < 0) ? ST_LEAF : ST_NODE);
                    sp = stuck ? top : sp;
                    cur = next;
                }
            }
            if (STATS == 2) ws.t_node += clock64() - t0;
        }
        // (and a second one, when enough lanes are still on a node)
        if (__popcll(__ballot(st == ST_NODE)) >= PT_REPEAT_LANES) {
            // ---------------- one inner-node step: both children's boxes, nearer child first
            if (STATS == 2) t0 = clock64();
            // Straight-line code: all four children are written to the stack, skipped ones above its new top
            // (the launch sizes the stack for it), and every decision is a select.
            if (STATS) { const int on_a_node = __popcll(__ballot(st == ST_NODE)); ws.node_phases += on_a_node > 0 ? 1 : 0; ws.node_lanes += on_a_node; }
            if (st == ST_NODE) {
                if (NODE32) {
                    quad_step(cur);
                } else {
                    const PairNode* p = s.tri_tree.pairs + cur;
                    const int lref = p->left, rref = p->right;
                    double lmin, lmax, rmin, rmax;
                    slab_finite(ray.o, inv, p->lbox[0], p->lbox[1], p->lbox[2], p->lbox[3], p->lbox[4], p->lbox[5], lmin, lmax);
                    slab_finite(ray.o, inv, p->rbox[0], p->rbox[1], p->rbox[2], p->rbox[3], p->rbox[4], p->rbox[5], rmin, rmax);
                    const bool go_l = !(lmax < 0.0f || lmin > lmax) && !(lmin > cut);
                    const bool go_r = !(rmax < 0.0f || rmin > rmax) && !(rmin > cut);
                    const bool left_first = !(rmin < lmin);
                    const bool both = go_l && go_r;
                    const int near = (both ? left_first : go_l) ? lref : rref;
                    stack_ref[sp * BLOCK_THREADS] = left_first ? rref : lref;
                    sp += both ? 1 : 0;
                    const bool stuck = !(go_l || go_r);
                    const int top = max(sp - 1, 0);
                    const int pending = stack_ref[top * BLOCK_THREADS];
                    const int next = stuck ? pending : near;
                    st = (stuck && sp == 0) ? ST_SHADE : ((next < 0) ? ST_LEAF : ST_NODE);
                    sp = stuck ? top : sp;
                    cur = next;
                }
            }
            if (STATS == 2) ws.t_node += clock64() - t0;
        }
        // ---------------- leaf visits, when enough lanes stand on a leaf by now or none is left on a node
        const unsigned long long on_a_leaf = __ballot(st == ST_LEAF);
        if (__popcll(on_a_leaf) >= w.leaf_threshold || (on_a_leaf != 0ull && __ballot(st == ST_NODE) == 0ull)) {
            // ---------------- leaf visits: the reference's box test on the leaf's own box, then the triangle
            if (STATS) { ++ws.leaf_phases; ws.leaf_lanes += __popcll(__ballot(st == ST_LEAF)); }
            if (STATS == 2) t0 = clock64();
            // Straight-line code: the whole record is fetched at once and the outcome applied with selects.
            if (st == ST_LEAF) {
                const int primitive = ~cur;
                // what the lane visits next is read from its stack before the record is fetched, not after the test
                const int below = max(sp - 1, 0);
                const int pending = stack_ref[below * BLOCK_THREADS];
                // a record is addressed as the array's base plus a 32-bit byte offset (the host keeps the four-wide
                // hierarchy to scenes of at most 2^25 triangles), the form the hardware adds by itself
                const TriRecord* const record = reinterpret_cast<const TriRecord*>(reinterpret_cast<const char*>(s.tris) + (size_t)((unsigned)primitive << 7));
                // With the conservative nodes the leaf's own box has not been tested exactly yet; that test is
                // made once, on the winning primitive, before the path is shaded.
                if (NODE32 && MERGED) {
                    const bool is_sphere = primitive >= s.sphere_base;
                    double d = 0.0;
                    bool hit = false;
                    if (is_sphere) hit = hit_sphere(ray, s.spheres + 4 * (primitive - s.sphere_base), d);
                    else hit = hit_triangle_together(ray, NODE32 ? record : s.tris + primitive, d);
                    // nearest triangle and nearest sphere are kept apart (the reference compares them only at
                    // the end, src/path_tracing.cpp:183); both prune the one search
                    const double own = is_sphere ? hs.distance : best_t;
                    const double other = is_sphere ? best_t : hs.distance;
                    const bool better = hit & (d < own);
                    // (a large triangle is in the hierarchy under several references: meeting the winner again is no tie)
                    const bool equal = hit & (d == own) & (is_sphere | (primitive != best_i));
                    const bool closer = better & (d < other) & (w.export_hits == 0);    // reporting both indices: nothing is pruned by a hit
                    const double cut_d = prune_cut(d);
                    tie = is_sphere ? tie : ((tie & !better) | equal);
                    tie_s = is_sphere ? ((tie_s & !better) | equal) : tie_s;
                    best_t = (better & !is_sphere) ? d : best_t;
                    best_i = (better & !is_sphere) ? primitive : best_i;
                    hs.distance = (better & is_sphere) ? d : hs.distance;
                    hs.index = (better & is_sphere) ? primitive - s.sphere_base : hs.index;
                    cut = closer ? cut_d : cut;
                    cut32 = closer ? cut32_after_hit(d) : cut32;
                } else {
                    double d;
                    const bool hit = hit_triangle_together(ray, NODE32 ? record : s.tris + primitive, d);
                    const bool better = hit & (d < best_t);
                    // (a large triangle is in the hierarchy under several references: meeting the winner again is no tie)
                    const bool equal = hit & (d == best_t) & (primitive != best_i);
                    const bool closer = better & (d < hs.distance);
                    const double cut_d = prune_cut(d);
                    tie = (tie & !better) | equal;
                    best_t = better ? d : best_t;
                    best_i = better ? primitive : best_i;
                    cut = closer ? cut_d : cut;
                    if (NODE32) cut32 = closer ? cut32_after_hit(d) : cut32;
                }
                st = (sp == 0) ? ST_SHADE : ((pending < 0) ? ST_LEAF : ST_NODE);
                sp = below;
                cur = pending;
            }
            if (STATS == 2) ws.t_leaf += clock64() - t0;
        }
    }
    if (STATS && lane == 0) {
        if (STATS == 2) atomicAdd(stats + 15, 1ull);   // waves
        atomicAdd(stats + 9, ws.t_node); atomicAdd(stats + 10, ws.t_leaf); atomicAdd(stats + 11, ws.t_shade);
        atomicAdd(stats + 12, ws.t_refill); atomicAdd(stats + 13, ws.t_setup);
        atomicAdd(stats + 14, (unsigned long long)(clock64() - t_begin));
        atomicAdd(stats + 0, ws.node_phases); atomicAdd(stats + 1, ws.node_lanes);
        atomicAdd(stats + 2, ws.leaf_phases); atomicAdd(stats + 3, ws.leaf_lanes);
        atomicAdd(stats + 4, ws.shade_phases); atomicAdd(stats + 5, ws.shade_lanes);
        atomicAdd(stats + 6, ws.loops); atomicAdd(stats + 7, ws.paths);
        atomicAdd(stats + 8, ws.unsafe_skips);
    }
}

// One lane per pixel: the pixel's colours are added in ascending sample order, the reference's
// `pixels[i] += colour` once per frame (src/main.cpp:117-121).
__global__ __launch_bounds__(BLOCK_THREADS) void sum_samples_kernel(FrameParams f, WorkParams w, const double* __restrict__ rgb,
                                                                    double* __restrict__ accum) {
    const int lane = threadIdx.x & 63;
    const long long tile = (long long)blockIdx.x * (BLOCK_THREADS / 64) + (threadIdx.x >> 6);
    const int tile_row = (int)(tile / w.tiles_x) * f.shard_count + f.shard_index, tile_col = (int)(tile % w.tiles_x);
    const int row = f.row_begin + tile_row * 8 + (lane >> 3);
    const int column = tile_col * 8 + (lane & 7);
    if (!(row < f.row_end && column < f.width)) return;
    double* px = accum + 4ll * ((long long)row * f.width + column);
    double sum_b = px[0], sum_g = px[1], sum_r = px[2], sum_n = px[3];
    for (int i = 0; i < f.sample_count; ++i) {
        const long long chunk = tile * w.groups_per_tile + i / SAMPLES_PER_CHUNK;
        const int j = (i % SAMPLES_PER_CHUNK) * 64 + lane;
        sum_r += rgb[rgb_index(chunk, 0, j)];
        sum_g += rgb[rgb_index(chunk, 1, j)];
        sum_b += rgb[rgb_index(chunk, 2, j)];
        sum_n += 1.0f;
    }
    px[0] = sum_b; px[1] = sum_g; px[2] = sum_r; px[3] = sum_n;
}

}  // namespace ptd

#endif
