// The persistent wavefront kernel: paths are decoupled from pixels.
//
// Why: the reference sums a pixel's samples in ascending order (src/main.cpp:117-121 run once
// per frame, src/main.cpp:1001-1036), and f64 addition does not commute with reordering, so a
// lane that owns a pixel must walk its samples one after another; lanes of one wave then run
// out of step (most primary rays leave the scene at once, a few paths bounce 50 times) and the
// wave idles.  A path's colour, however, is a pure function of (row, column, sample): it can
// be computed by any lane at any time.  So:
//
//   trace_paths_kernel   persistent waves pull chunks of 1024 paths (one 8x8 tile x 16
//                        samples) from a counter in HBM and write each path's colour to
//                        `rgb`; a lane that finishes a path takes the next one, so all 64
//                        lanes stay busy until the frame runs out of work.  Inside the wave,
//                        lanes are regrouped with ballots: node steps, triangle tests and
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

constexpr int SAMPLES_PER_CHUNK = 16;
constexpr int CHUNK_PATHS = 64 * SAMPLES_PER_CHUNK;

struct WorkParams {
    int tiles_x;                 // 8x8 tiles across the image
    int groups_per_tile;         // ceil(sample_count / SAMPLES_PER_CHUNK)
    long long chunk_count;       // tiles_x * local row blocks * groups_per_tile
};

// rgb layout: [chunk][channel r,g,b][CHUNK_PATHS] doubles; path j of a chunk is sample
// (j / 64) of the chunk's sample group at pixel (j % 64) of the chunk's tile.
PT_DEV long long rgb_index(long long chunk, int channel, int j) { return (chunk * 3 + channel) * CHUNK_PATHS + j; }

enum LaneState { ST_NEED = 0, ST_NODE = 1, ST_LEAF = 2, ST_SHADE = 3, ST_NEWRAY = 4 };

struct WaveStats {   // wave-uniform counters of a profiling build
    unsigned long long node_phases, node_lanes, leaf_phases, leaf_lanes, shade_phases, shade_lanes, loops, paths;
};

template <bool STATS, int S_THRESHOLD, int L_THRESHOLD>
__global__ __launch_bounds__(BLOCK_THREADS) void trace_paths_kernel(DeviceScene s, FrameParams f, WorkParams w,
                                                                    double* __restrict__ rgb,
                                                                    unsigned int* __restrict__ next_chunk,
                                                                    unsigned long long* __restrict__ stats) {
    __shared__ BlockShared sh;
    const int lane = threadIdx.x & 63;
    int* const stack = sh.stack + threadIdx.x;
    const TriLeaf tri_leaf{s.tris};
    const SphLeaf sph_leaf{s.spheres};

    // wave-uniform work state
    long long chunk = -1;
    int chunk_next = 0, chunk_size = 0;
    bool work_left = true;
    WaveStats ws{0, 0, 0, 0, 0, 0, 0, 0};

    // per-lane path state
    int st = ST_NEED;
    int cur = 0, sp = 0, bounce = 0, best_i = -1;
    long long slot = 0;                 // chunk * CHUNK_PATHS + j of the lane's path
    bool tie = false;
    double best_t = REAL_MAX, cut = REAL_MAX;
    Hit hs{-1, REAL_MAX};
    Ray ray{D3{0, 0, 0}, D3{0, 0, 1}};
    D3 inv{0, 0, 1}, att{1.0f, 1.0f, 1.0f};

    for (;;) {
        const unsigned long long m_node = __ballot(st == ST_NODE);
        const unsigned long long m_leaf = __ballot(st == ST_LEAF);
        const unsigned long long m_ready = __ballot(st == ST_SHADE || (st == ST_NEED && work_left));
        const int n_ready = __popcll(m_ready);
        if (STATS) ++ws.loops;

        if ((m_node | m_leaf) == 0ull || n_ready >= S_THRESHOLD) {
            if (n_ready == 0) break;      // nothing in flight, nothing to shade, no work left
            if (STATS) { ++ws.shade_phases; ws.shade_lanes += n_ready; }
            // ---------------- shade the lanes whose search is over (src/path_tracing.cpp:179-223)
            if (st == ST_SHADE) {
                Hit ht{best_i, best_t};
                resolve_slow_lanes(tie, ray, s.tri_tree, tri_leaf, wave_frames(sh), ht);
                D3 gain;
                const bool go_on = shade_vertex(s, ray, hs, ht, att, gain);
                ++bounce;
                if (!go_on || bounce >= f.max_bounces) {
                    // every earlier vertex added +-0 (see the file header): colour = 0 + last term
                    const long long c = slot / CHUNK_PATHS;
                    const int j = (int)(slot % CHUNK_PATHS);
                    rgb[rgb_index(c, 0, j)] = 0.0f + gain.x;
                    rgb[rgb_index(c, 1, j)] = 0.0f + gain.y;
                    rgb[rgb_index(c, 2, j)] = 0.0f + gain.z;
                    st = ST_NEED;
                } else {
                    st = ST_NEWRAY;
                }
            }
            // ---------------- hand new paths to idle lanes
            for (;;) {
                const unsigned long long m_need = __ballot(st == ST_NEED);
                if (m_need == 0ull || !work_left) break;
                if (chunk_next >= chunk_size) {
                    unsigned int c = 0;
                    if (lane == 0) c = atomicAdd(next_chunk, 1u);
                    c = __builtin_amdgcn_readfirstlane(c);
                    if ((long long)c >= w.chunk_count) { work_left = false; break; }
                    chunk = (long long)c;
                    chunk_next = 0;
                    const int group = (int)(chunk % w.groups_per_tile);
                    const int samples = min(SAMPLES_PER_CHUNK, f.sample_count - group * SAMPLES_PER_CHUNK);
                    chunk_size = 64 * samples;
                }
                const int rank = __popcll(m_need & ((1ull << lane) - 1ull));
                const int available = chunk_size - chunk_next;
                if (st == ST_NEED && rank < available) {
                    const int j = chunk_next + rank;
                    const long long tile = chunk / w.groups_per_tile;
                    const int group = (int)(chunk % w.groups_per_tile);
                    const int tile_row = (int)(tile / w.tiles_x) * f.shard_count + f.shard_index;
                    const int tile_col = (int)(tile % w.tiles_x);
                    const int pixel = j & 63;
                    const int row = f.row_begin + tile_row * 8 + (pixel >> 3);
                    const int column = tile_col * 8 + (pixel & 7);
                    const int sample = f.sample_begin + group * SAMPLES_PER_CHUNK + (j >> 6);
                    if (row < f.row_end && column < f.width) {
                        slot = chunk * CHUNK_PATHS + j;
                        ray = camera_ray(f, row, column, sample);
                        att = D3{1.0f, 1.0f, 1.0f};
                        bounce = 0;
                        st = ST_NEWRAY;
                    }
                    // a pixel outside the image (partial tile) is skipped; the lane asks again
                }
                const int taken = min(__popcll(m_need), available);
                if (STATS) ws.paths += taken;
                chunk_next += taken;
            }
            // ---------------- start the search of every new ray (src/path_tracing.cpp:177-178)
            if (st == ST_NEWRAY) {
                inv = D3{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
                hs = trace_tree(ray, s.sph_tree, sph_leaf, REAL_MAX, sh);
                best_i = -1;
                best_t = REAL_MAX;
                tie = false;
                sp = 0;
                st = ST_SHADE;                       // unless a search is started below
                if (s.tri_tree.count > 0) {
                    const int kind = classify_ray(ray, inv);
                    if (kind == RAY_EXACT_ONLY) {
                        Hit ht{-1, REAL_MAX};
                        resolve_slow_lanes(true, ray, s.tri_tree, tri_leaf, wave_frames(sh), ht);
                        best_i = ht.index;
                        best_t = ht.distance;
                    } else if (kind == RAY_FINITE) {
                        // a triangle beyond the sphere hit cannot win (:183): the search may stop there
                        cut = prune_cut(hs.distance);
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
        } else if (m_node == 0ull || __popcll(m_leaf) >= L_THRESHOLD) {
            // ---------------- triangle tests for the lanes standing on a leaf
            if (STATS) { ++ws.leaf_phases; ws.leaf_lanes += __popcll(m_leaf); }
            if (st == ST_LEAF) {
                const int primitive = ~cur;
                double d;
                if (hit_triangle(ray, s.tris + primitive, d)) {
                    if (d < best_t) {
                        best_t = d;
                        best_i = primitive;
                        tie = false;
                        if (d < hs.distance) cut = prune_cut(d);
                    } else if (d == best_t) {
                        tie = true;
                    }
                }
                if (sp == 0) {
                    st = ST_SHADE;
                } else {
                    --sp;
                    cur = stack[sp * BLOCK_THREADS];
                    st = (cur < 0) ? ST_LEAF : ST_NODE;
                }
            }
        } else {
            // ---------------- one inner-node step: both children's boxes, nearer child first
            if (STATS) { ++ws.node_phases; ws.node_lanes += __popcll(m_node); }
            if (st == ST_NODE) {
                const PairNode* p = s.tri_tree.pairs + cur;
                const int lref = p->left, rref = p->right;
                double lmin, lmax, rmin, rmax;
                slab_finite(ray.o, inv, p->lbox[0], p->lbox[1], p->lbox[2], p->lbox[3], p->lbox[4], p->lbox[5], lmin, lmax);
                slab_finite(ray.o, inv, p->rbox[0], p->rbox[1], p->rbox[2], p->rbox[3], p->rbox[4], p->rbox[5], rmin, rmax);
                const bool go_l = !(lmax < 0.0f || lmin > lmax) && !(lmin > cut);
                const bool go_r = !(rmax < 0.0f || rmin > rmax) && !(rmin > cut);
                if (go_l && go_r) {
                    const bool left_first = !(rmin < lmin);
                    stack[sp * BLOCK_THREADS] = left_first ? rref : lref;
                    ++sp;
                    cur = left_first ? lref : rref;
                } else if (go_l) {
                    cur = lref;
                } else if (go_r) {
                    cur = rref;
                } else if (sp == 0) {
                    st = ST_SHADE;
                } else {
                    --sp;
                    cur = stack[sp * BLOCK_THREADS];
                }
                if (st != ST_SHADE) st = (cur < 0) ? ST_LEAF : ST_NODE;
            }
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(stats + 0, ws.node_phases); atomicAdd(stats + 1, ws.node_lanes);
        atomicAdd(stats + 2, ws.leaf_phases); atomicAdd(stats + 3, ws.leaf_lanes);
        atomicAdd(stats + 4, ws.shade_phases); atomicAdd(stats + 5, ws.shade_lanes);
        atomicAdd(stats + 6, ws.loops); atomicAdd(stats + 7, ws.paths);
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
