// The GPU path: scene flattening into HBM, the render / primary-hit / self-test kernels
// and their C-ABI entry points (pt_scene_create, pt_render, pt_render_device, pt_primary_hits,
// pt_device_eval).  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "pt_wavefront.hpp"
#include "pt_internal.h"
#include "pt_traverse_host.h"

using namespace ptd;

// =====================================================================================
// Kernels
// =====================================================================================


// Wave -> 8x8 pixel tile, lane -> pixel inside the tile.
PT_DEV bool pixel_of_thread(const FrameParams& f, int& row, int& column) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (BLOCK_THREADS / 64) + (threadIdx.x >> 6);
    const int tiles_x = (f.width + 7) >> 3;
    const int tile_row = (int)(wave / tiles_x) * f.shard_count + f.shard_index, tile_col = (int)(wave % tiles_x);
    row = f.row_begin + tile_row * 8 + (lane >> 3);
    column = tile_col * 8 + (lane & 7);
    return row < f.row_end && column < f.width;
}

// One lane owns one pixel and walks its samples in ascending order, so the per-pixel
// sum is formed in the order the reference's frame loop forms it (src/main.cpp:1001-1036).
// Lanes regenerate independently: a lane whose path ended starts its next sample while its
// neighbours are still bouncing.
__global__ __launch_bounds__(BLOCK_THREADS) void render_pixels_kernel(DeviceScene s, FrameParams f, double* __restrict__ accum) {
    __shared__ BlockShared sh;
    int row, column;
    const bool in_image = pixel_of_thread(f, row, column);
    bool active = in_image && f.sample_count > 0;

    double* px = accum + 4ll * ((long long)row * f.width + column);
    double sum_b = 0.0, sum_g = 0.0, sum_r = 0.0, sum_n = 0.0;
    if (active) { sum_b = px[0]; sum_g = px[1]; sum_r = px[2]; sum_n = px[3]; }

    int sample = f.sample_begin;
    const int sample_end = f.sample_begin + f.sample_count;
    int bounce = 0;
    Ray ray{D3{0, 0, 0}, D3{0, 0, 1}};
    D3 colour{0.0f, 0.0f, 0.0f}, att{1.0f, 1.0f, 1.0f};
    if (active && f.max_bounces <= 0) {
        // the reference's bounce loop would not run at all: each sample adds a zero colour and one count
        for (; sample < sample_end; ++sample) { sum_b += 0.0f; sum_g += 0.0f; sum_r += 0.0f; sum_n += 1.0f; }
        active = false;
    }
    if (active) ray = camera_ray(f, row, column, sample);

    const TriLeaf tri_leaf{s.tris};
    const SphLeaf sph_leaf{s.spheres};

    while (__any(active)) {
        if (active) {
            // src/path_tracing.cpp:177-178
            const Hit hs = trace_tree(ray, s.sph_tree, sph_leaf, REAL_MAX, sh);
            // a triangle beyond the sphere hit cannot win (:183), so the search may stop there
            const Hit ht = trace_tree(ray, s.tri_tree, tri_leaf, hs.distance, sh);
            D3 gain;
            const bool go_on = shade_vertex(s, ray, hs, ht, att, gain);
            colour = add3(colour, gain);                                           // :214 / :217 / :221
            if (go_on) ++bounce;
            const bool ended = !go_on || bounce >= f.max_bounces;                  // :218, :222, loop exhaustion :176
            if (ended) {
                sum_b += colour.z;                                                 // src/main.cpp:117-121
                sum_g += colour.y;
                sum_r += colour.x;
                sum_n += 1.0f;
                ++sample;
                if (sample < sample_end) {
                    ray = camera_ray(f, row, column, sample);
                    colour = D3{0.0f, 0.0f, 0.0f};
                    att = D3{1.0f, 1.0f, 1.0f};
                    bounce = 0;
                } else {
                    active = false;
                }
            }
        }
    }
    if (in_image && f.sample_count > 0) { px[0] = sum_b; px[1] = sum_g; px[2] = sum_r; px[3] = sum_n; }
}

__global__ __launch_bounds__(BLOCK_THREADS) void primary_hits_kernel(DeviceScene s, FrameParams f, int* __restrict__ tri_index, int* __restrict__ sph_index) {
    __shared__ BlockShared sh;
    int row, column;
    const bool in_image = pixel_of_thread(f, row, column);
    Ray ray{D3{0, 0, 0}, D3{0, 0, 1}};
    if (in_image) ray = camera_ray(f, row, column, f.sample_begin);
    Hit hs{-1, REAL_MAX}, ht{-1, REAL_MAX};
    if (in_image) {
        hs = trace_tree(ray, s.sph_tree, SphLeaf{s.spheres}, REAL_MAX, sh);
        ht = trace_tree(ray, s.tri_tree, TriLeaf{s.tris}, REAL_MAX, sh);
    }
    if (in_image) {
        const long long i = (long long)row * f.width + column;
        if (tri_index != nullptr) tri_index[i] = ht.index;
        if (sph_index != nullptr) sph_index[i] = hs.index;
    }
}

// One building block per record; see PT_EVAL_* in the public header for the record shapes.
__global__ void eval_kernel(int op, const void* __restrict__ in, long long n, void* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* din = (const double*)in;
    double* dout = (double*)out;
    switch (op) {
    case PT_EVAL_NOISE_1D:
        ((uint32_t*)out)[i] = noise_1d((uint32_t)((const int*)in)[i]);
        break;
    case PT_EVAL_NOISE_3D: {
        const int* v = (const int*)in + 3 * i;
        ((uint32_t*)out)[i] = noise_3d(v[0], v[1], v[2]);
        break;
    }
    case PT_EVAL_XORSHIFT_REAL: {
        const uint32_t r = xorshift(((const uint32_t*)in)[i]);
        ((uint32_t*)out)[4 * i] = r;
        ((uint32_t*)out)[4 * i + 1] = 0u;
        dout[2 * i + 1] = real_from_rng(r);
        break;
    }
    case PT_EVAL_RANDOM_UNIT_VECTOR: {
        const D3 v = random_unit_vector(D3{din[3 * i], din[3 * i + 1], din[3 * i + 2]});
        dout[3 * i] = v.x; dout[3 * i + 1] = v.y; dout[3 * i + 2] = v.z;
        break;
    }
    case PT_EVAL_RAY_AABB: {
        const double* r = din + 12 * i;
        const D3 inv{1.0f / r[3], 1.0f / r[4], 1.0f / r[5]};
        double tmin, tmax;
        slab_exact(D3{r[0], r[1], r[2]}, inv, r + 6, tmin, tmax);
        bool bounded = true;
        for (int k = 0; k < 12; ++k) bounded = bounded && isfinite(r[k]);
        if (bounded && isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z)) {
            // the short form must agree wherever it is allowed to run: finite reciprocals, origin and box (zeros may differ in sign)
            double fmin_, fmax_;
            slab_finite(D3{r[0], r[1], r[2]}, inv, r[6], r[7], r[8], r[9], r[10], r[11], fmin_, fmax_);
            if (!(fmin_ == tmin && fmax_ == tmax)) { tmin = __builtin_nan(""); tmax = __builtin_nan(""); }
        }
        dout[2 * i] = tmin; dout[2 * i + 1] = tmax;
        break;
    }
    case PT_EVAL_RAY_SPHERE: {
        const double* r = din + 10 * i;
        double lo, hi;
        const bool ok = hit_sphere_raw(Ray{D3{r[0], r[1], r[2]}, D3{r[3], r[4], r[5]}}, r + 6, lo, hi);
        dout[3 * i] = ok ? 1.0 : 0.0; dout[3 * i + 1] = lo; dout[3 * i + 2] = hi;
        break;
    }
    case PT_EVAL_RAY_TRIANGLE: {
        const double* r = din + 15 * i;
        // the record is normally precomputed on the host; here it is formed on the device
        const D3 a{r[6], r[7], r[8]}, b{r[9], r[10], r[11]}, c{r[12], r[13], r[14]};
        const D3 ab = sub3(b, a), ac = sub3(c, a), nrm = cross3(ab, ac);
        TriRecord t;
        t.a[0] = a.x; t.a[1] = a.y; t.a[2] = a.z;
        t.ab[0] = ab.x; t.ab[1] = ab.y; t.ab[2] = ab.z;
        t.ac[0] = ac.x; t.ac[1] = ac.y; t.ac[2] = ac.z;
        t.nn = dot3(nrm, nrm);
        double d;
        const bool ok = hit_triangle(Ray{D3{r[0], r[1], r[2]}, D3{r[3], r[4], r[5]}}, &t, d);
        // the form the wavefront kernel's lanes run together must agree on every input
        double d_together;
        const bool ok_together = hit_triangle_together(Ray{D3{r[0], r[1], r[2]}, D3{r[3], r[4], r[5]}}, &t, d_together);
        const bool same = ok_together == ok && (d_together == d || (d_together != d_together && d != d));
        dout[2 * i] = same ? (ok ? 1.0 : 0.0) : __builtin_nan(""); dout[2 * i + 1] = d;
        break;
    }
    case PT_EVAL_UNIT_NORMAL: {
        const double* r = din + 9 * i;
        const D3 a{r[0], r[1], r[2]}, b{r[3], r[4], r[5]}, c{r[6], r[7], r[8]};
        const D3 un = unit3(cross3(sub3(b, a), sub3(c, a)));
        dout[3 * i] = un.x; dout[3 * i + 1] = un.y; dout[3 * i + 2] = un.z;
        break;
    }
    case PT_EVAL_SCATTER: {
        const double* r = din + 17 * i;
        const Mat m{r[6], r[7], r[8], r[9], (int)r[10]};
        Ray next;
        const Ray in{D3{r[0], r[1], r[2]}, D3{r[3], r[4], r[5]}};
        const D3 point{r[11], r[12], r[13]}, normal{r[14], r[15], r[16]};
        bool ok = scatter(in, m, point, normal, next, true);
        if (m.type != PT_METAL) {
            // a scene without metals takes the other form of the same arithmetic: any difference poisons the answer
            Ray other;
            const bool other_ok = scatter(in, m, point, normal, other, false);
            auto bits = [](double v) { return __double_as_longlong(v); };
            const bool same = other_ok == ok && bits(other.o.x) == bits(next.o.x) && bits(other.o.y) == bits(next.o.y) && bits(other.o.z) == bits(next.o.z) &&
                              bits(other.d.x) == bits(next.d.x) && bits(other.d.y) == bits(next.d.y) && bits(other.d.z) == bits(next.d.z);
            if (!same) { next.o = D3{__builtin_nan(""), __builtin_nan(""), __builtin_nan("")}; next.d = next.o; ok = !ok; }
        }
        double* o = dout + 7 * i;
        o[0] = ok ? 1.0 : 0.0;
        o[1] = next.o.x; o[2] = next.o.y; o[3] = next.o.z;
        o[4] = next.d.x; o[5] = next.d.y; o[6] = next.d.z;
        break;
    }
    case PT_EVAL_SQRT_DIV:
        dout[2 * i] = sqrt(din[2 * i]);
        dout[2 * i + 1] = din[2 * i] / din[2 * i + 1];
        break;
    default:
        break;
    }
}

// =====================================================================================
// Host side: flattening, upload, launches
// =====================================================================================

// Workspace of pt_render_device_noalloc: the unit counter and profiling counters (first
// STATIC_COUNTER_BYTES) and the per-path colour records behind them.  Every GPU the library runs on holds
// its own copy of the array, at its own address, until the process ends.  In host code the array's name is
// only a handle for hipGetSymbolAddress: kernels, memsets and copies are given the address that call returns.
constexpr size_t STATIC_COUNTER_BYTES = 1024;
constexpr size_t STATIC_COLOUR_BYTES = size_t(PT_NOALLOC_MAX_TILES) * 4 * CHUNK_PATHS * sizeof(double);   // one sample group of that many tiles
static __device__ __align__(256) unsigned char scratch[STATIC_COUNTER_BYTES + STATIC_COLOUR_BYTES];

namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        const hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return pt::fail(PT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

typedef pt::UnfilledVector<PairNodeHost> PairNodes;

struct DeviceBuffer {
    void* p = nullptr;
    ~DeviceBuffer() { if (p != nullptr) (void)hipFree(p); }
    template <class T> const T* as() const { return static_cast<const T*>(p); }
};

constexpr int MAX_DEVICES = 64;
constexpr double DEFAULT_PRESPLIT = 1.5, DEFAULT_PRESPLIT_FLOOR = 0.2;
constexpr long long PRESPLIT_LEAST_TRIANGLES = 4096;
// What a scene handle may keep of device memory for its per-path colour records unless its owner says otherwise: the
// benchmark frame (1920 x 1080 x 256 samples, 17 GB) fits; longer sample ranges run in batches (3840 x 2160 x 1024: 11 of them,
// 2 % slower than in 3 through 96 GB).
constexpr size_t DEFAULT_WORKSPACE_BYTES = size_t(24) << 30;
std::mutex static_scratch_lock;
void* static_scratch_address[MAX_DEVICES] = {};     // per device, resolved once with that device current
// Every scene handle of a device shares that device's static workspace: each launch that uses it waits for the end of the
// one before (this event, recorded behind every such launch), whatever handles and streams the two came from.
hipEvent_t static_scratch_free[MAX_DEVICES] = {};
bool static_scratch_used[MAX_DEVICES] = {};

// Address of `device`'s copy of the static workspace (the device must be current).
int static_scratch_of(int device, void** out) {
    if (device < 0 || device >= MAX_DEVICES) return pt::fail(PT_ERR_UNSUPPORTED, "device %d: the static workspace is tracked for %d devices", device, MAX_DEVICES);
    std::lock_guard<std::mutex> guard(static_scratch_lock);
    if (static_scratch_address[device] == nullptr) {
        void* address = nullptr;
        HIP_TRY(hipGetSymbolAddress(&address, HIP_SYMBOL(scratch)));
        if (address == nullptr) return pt::fail(PT_ERR_DEVICE, "hipGetSymbolAddress gave no address for the static workspace on device %d", device);
        HIP_TRY(hipEventCreateWithFlags(&static_scratch_free[device], hipEventDisableTiming));
        static_scratch_address[device] = address;
    }
    *out = static_scratch_address[device];
    return PT_OK;
}

int upload(DeviceBuffer& buffer, const void* host, size_t bytes) {
    if (bytes == 0) return PT_OK;
    HIP_TRY(hipMalloc(&buffer.p, bytes));
    HIP_TRY(hipMemcpy(buffer.p, host, bytes, hipMemcpyHostToDevice));
    return PT_OK;
}

// Host arrays on their way to the device while the caller builds the next one: device memory is allocated at once (the
// caller may store the address), the copy is made by a thread of the uploader's own, in the order of the calls.  An array
// must stay as it is until finish() has returned.
class Uploader {
public:
    explicit Uploader(int device) : device_(device) {}
    ~Uploader() { (void)finish(); }
    int send(DeviceBuffer& buffer, const void* host, size_t bytes) {
        if (bytes == 0) return PT_OK;
        HIP_TRY(hipMalloc(&buffer.p, bytes));
        {
            std::lock_guard<std::mutex> hold(lock_);
            queue_.push_back(Copy{buffer.p, host, bytes});
        }
        if (!worker_.joinable()) worker_ = std::thread([this]() { run(); });
        changed_.notify_one();
        return PT_OK;
    }
    // Waits for every copy; reports the first one that failed.
    int finish() {
        if (worker_.joinable()) {
            {
                std::lock_guard<std::mutex> hold(lock_);
                closed_ = true;
            }
            changed_.notify_one();
            worker_.join();
        }
        const hipError_t e = failed_;
        failed_ = hipSuccess;
        closed_ = false;
        if (e != hipSuccess) return pt::fail(PT_ERR_DEVICE, "copying a scene array to the device: %s", hipGetErrorString(e));
        return PT_OK;
    }

private:
    struct Copy { void* to; const void* from; size_t bytes; };
    void run() {
        hipError_t e = hipSetDevice(device_);
        for (;;) {
            Copy c;
            {
                std::unique_lock<std::mutex> hold(lock_);
                changed_.wait(hold, [this]() { return !queue_.empty() || closed_; });
                if (queue_.empty()) break;
                c = queue_.front();
                queue_.pop_front();
            }
            if (e == hipSuccess) e = hipMemcpy(c.to, c.from, c.bytes, hipMemcpyHostToDevice);
        }
        failed_ = e;
    }
    int device_;
    std::mutex lock_;
    std::condition_variable changed_;
    std::deque<Copy> queue_;
    bool closed_ = false;
    hipError_t failed_ = hipSuccess;
    std::thread worker_;
};

// Checks that `nodes` is a well-formed tree in the reference's layout over `primitive_count`
// leaves (every primitive exactly once) and returns its height (leaf-only tree = 1).
// *nested_out: every box has min <= max (false for NaN) and every inner box contains both children's boxes.
// *finite_out: no box holds an infinity or a NaN.
int validate_tree(const PtBvhNode* nodes, int node_count, int primitive_count, const char* what, int* height_out, bool* nested_out, bool* finite_out) {
    if (node_count != 2 * primitive_count - 1) return pt::fail(PT_ERR_INVALID_ARGUMENT, "%s BVH: %d nodes for %d primitives, expected %d", what, node_count, primitive_count, 2 * primitive_count - 1);
    std::vector<char> seen_node(node_count, 0), seen_primitive(primitive_count, 0);
    std::vector<std::pair<int, int>> todo;   // node, depth
    todo.emplace_back(0, 1);
    int height = 0;
    bool nested = true, finite = true;
    auto bounded = [](const PtAabb& b) {
        return std::isfinite(b.min.x) && std::isfinite(b.min.y) && std::isfinite(b.min.z) && std::isfinite(b.max.x) && std::isfinite(b.max.y) && std::isfinite(b.max.z);
    };
    auto upright = [](const PtAabb& b) { return b.min.x <= b.max.x && b.min.y <= b.max.y && b.min.z <= b.max.z; };
    auto holds = [](const PtAabb& outer, const PtAabb& inner) {
        return outer.min.x <= inner.min.x && outer.min.y <= inner.min.y && outer.min.z <= inner.min.z &&
               outer.max.x >= inner.max.x && outer.max.y >= inner.max.y && outer.max.z >= inner.max.z;
    };
    while (!todo.empty()) {
        const auto [i, depth] = todo.back();
        todo.pop_back();
        if (i < 0 || i >= node_count || seen_node[i]) return pt::fail(PT_ERR_INVALID_ARGUMENT, "%s BVH: node %d is out of range or reached twice", what, i);
        seen_node[i] = 1;
        height = std::max(height, depth);
        const PtBvhNode& n = nodes[i];
        nested = nested && upright(n.aabb);
        finite = finite && bounded(n.aabb);
        if (n.left == 0 && n.right == 0) {
            if (n.index < 0 || n.index >= primitive_count || seen_primitive[n.index]) return pt::fail(PT_ERR_INVALID_ARGUMENT, "%s BVH: leaf %d names primitive %d (out of range or repeated)", what, i, n.index);
            seen_primitive[n.index] = 1;
        } else {
            if (n.left <= 0 || n.right <= 0) return pt::fail(PT_ERR_INVALID_ARGUMENT, "%s BVH: node %d has one child only", what, i);
            if (n.left < node_count && n.right < node_count) nested = nested && holds(n.aabb, nodes[n.left].aabb) && holds(n.aabb, nodes[n.right].aabb);
            todo.emplace_back(n.right, depth + 1);
            todo.emplace_back(n.left, depth + 1);
        }
    }
    *height_out = height;
    *nested_out = nested;
    *finite_out = finite;
    return PT_OK;
}

// Reference tree -> PairNode array (inner nodes numbered in pre-order).
void flatten_tree(const PtBvhNode* nodes, int node_count, PairNodes* pairs, int32_t* root_ref) {
    std::vector<int32_t> pair_of(node_count, -1);
    int32_t next = 0;
    for (int i = 0; i < node_count; ++i) {   // the reference numbers nodes in pre-order already
        if (!(nodes[i].left == 0 && nodes[i].right == 0)) pair_of[i] = next++;
    }
    pairs->assign(next, PairNodeHost{});
    auto ref_of = [&](int node) -> int32_t {
        return (nodes[node].left == 0 && nodes[node].right == 0) ? ~nodes[node].index : pair_of[node];
    };
    for (int i = 0; i < node_count; ++i) {
        if (pair_of[i] < 0) continue;
        PairNodeHost& p = (*pairs)[pair_of[i]];
        p.lbox = nodes[nodes[i].left].aabb;
        p.rbox = nodes[nodes[i].right].aabb;
        p.left = ref_of(nodes[i].left);
        p.right = ref_of(nodes[i].right);
    }
    *root_ref = ref_of(0);
}

D3 d3(const PtVec3& v) { return D3{v.x, v.y, v.z}; }

// Levels the rebuilt binary hierarchy is kept to (PT_SEARCH_HEIGHT overrides; at most what the traversal stacks hold).
int search_height_cap() {
    if (const char* v = std::getenv("PT_SEARCH_HEIGHT")) return std::min(static_cast<int>(STACK_DEPTH), std::max(2, std::atoi(v)));
    return MAX_SEARCH_HEIGHT;
}

using pt::StageClock;

// Cutting large triangles into several references (presplit.cpp): a piece is cut while its box has more spare surface than
// this many mean box surfaces (PT_PRESPLIT_FLOOR overrides), for at most this fraction of the triangle count in added
// references (PT_PRESPLIT overrides; 0 = no cutting).
double presplit_floor() {
    if (const char* v = std::getenv("PT_PRESPLIT_FLOOR")) return std::max(0.0, std::atof(v));
    return DEFAULT_PRESPLIT_FLOOR;
}
double presplit_fraction() {
    if (const char* v = std::getenv("PT_PRESPLIT")) return std::max(0.0, std::atof(v));
    return DEFAULT_PRESPLIT;
}
// Scenes of fewer triangles than this are not cut (PT_PRESPLIT_LEAST overrides: the tests cut scenes of a few triangles).
long long presplit_least_triangles() {
    if (const char* v = std::getenv("PT_PRESPLIT_LEAST")) return std::max(0ll, std::atoll(v));
    return PRESPLIT_LEAST_TRIANGLES;
}

}  // namespace

struct PtScene {
    int device = 0;
    DeviceScene dev{};
    DeviceBuffer tri_pairs, tri_ref_nodes, tri_records, tri_quads, tri_quad_boxes, tri_leaf_boxes;
    int tri_height = 0, sph_height = 0;
    int tri_stack_need = 1;             // slots of the wavefront kernel's stack a search of the 4-wide hierarchy can touch
    DeviceBuffer sph_pairs, sph_ref_nodes, spheres, sph_material, materials;
    hipEvent_t start = nullptr, stop = nullptr;
    hipStream_t last_stream = nullptr;
    bool timed = false;
    // wavefront kernel: per-path colours, the chunk counter and the profiling counters
    DeviceBuffer rgb, counters, park;   // (park: backing store of the lanes' stacks, see WorkParams::parking)
    size_t rgb_bytes = 0, park_bytes = 0;
    size_t workspace_limit = DEFAULT_WORKSPACE_BYTES;   // cap of `rgb` (pt_scene_set_workspace_limit)
    void* last_counters = nullptr;      // counters of the most recent wavefront launch (the scene's own or the static workspace's)
    void* static_workspace = nullptr;   // this device's copy of the static workspace (pt_render_device_noalloc)
    bool albedo_bounded = false;        // every albedo finite with |albedo| <= 1
    int compute_units = 0, blocks_per_cu = 0, occupancy_variant = -1;
    unsigned long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool has_stats = false;
    ~PtScene() {
        if (start != nullptr) (void)hipEventDestroy(start);
        if (stop != nullptr) (void)hipEventDestroy(stop);
    }
};

namespace {

// `pairs_out` (if given) receives the binary hierarchy and must outlive `up`'s copies; so must `nodes`.
int build_tree(const PtBvhNode* nodes, int node_count, int primitive_count, const char* what, DeviceBuffer& pairs_buffer,
               DeviceBuffer& ref_buffer, TreeRef* tree, int* height_out, PairNodes* pairs_out, bool rebuild, Uploader& up) {
    int height = 0;
    StageClock clock;
    bool nested = false, finite = false;
    int rc = validate_tree(nodes, node_count, primitive_count, what, &height, &nested, &finite);
    clock.done("reference tree checked");
    if (rc != PT_OK) return rc;
    // Boxes that do not nest: which leaves the reference reaches depends on this very tree, so it is searched as it is.
    // Unbounded boxes have no centroid to sort by: their tree is kept too.
    rebuild = rebuild && nested && finite;
    if (height > STACK_DEPTH) return pt::fail(PT_ERR_UNSUPPORTED, "%s BVH is %d levels deep; the traversal stack holds %d", what, height, STACK_DEPTH);
    rc = up.send(ref_buffer, nodes, static_cast<size_t>(node_count) * sizeof(PtBvhNode));
    if (rc != PT_OK) return rc;
    static_assert(sizeof(PairNodeHost) == sizeof(PairNode), "host and device pair nodes must match");
    PairNodes local_pairs;
    PairNodes& pairs = pairs_out != nullptr ? *pairs_out : local_pairs;
    int32_t root = 0;
    bool rebuilt = false;
    if (rebuild && primitive_count > 2) {
        // The search hierarchy is this library's own (see sah_build.cpp); the reference's tree is
        // kept beside it for the exact traversal.  Leaf boxes are taken from the reference's leaves.
        pt::UnfilledVector<PtAabb> boxes(primitive_count);
        pt::in_parts(0, node_count, node_count >= (1 << 16) ? pt::host_threads() : 1, [&](long long from, long long to, int) {
            for (long long i = from; i < to; ++i) {
                if (nodes[i].left == 0 && nodes[i].right == 0) boxes[nodes[i].index] = nodes[i].aabb;
            }
        });
        int sah_root = 0, sah_height = 0;
        if (pt::build_sah_pairs(boxes.data(), primitive_count, search_height_cap(), &pairs, &sah_root, &sah_height) == PT_OK) {
            root = sah_root;
            height = sah_height;
            rebuilt = true;
        } else {
            pt::clear_error();
        }
    }
    if (!rebuilt) flatten_tree(nodes, node_count, &pairs, &root);
    clock.done(rebuilt ? "binary search hierarchy built (SAH)" : "reference tree flattened");
    *height_out = height;
    rc = up.send(pairs_buffer, pairs.data(), pairs.size() * sizeof(PairNode));
    if (rc != PT_OK) return rc;
    if (pairs_out == nullptr) {
        rc = up.finish();           // the pairs are this function's own
        if (rc != PT_OK) return rc;
    }
    tree->pairs = pairs_buffer.as<PairNode>();
    tree->ref_nodes = ref_buffer.as<PtBvhNode>();
    tree->root = root;
    tree->count = primitive_count;
    tree->nested = nested ? 1 : 0;
    tree->bounded = finite ? 1 : 0;
    std::memcpy(tree->root_box, &nodes[0].aabb, 48);
    return PT_OK;
}

// The wavefront kernel's search hierarchy: 4-wide single-precision nodes (boxes rounded outward)
// and every primitive's own exact box.  `tri_pairs` is the binary hierarchy over the triangles; when
// the scene has more than one sphere, a binary SAH hierarchy over triangles and spheres together is
// built here instead, so that one search finds both nearest hits.  Skipped, leaving tri_quads null
// (the exact-node path then serves the scene), when a coordinate is outside the range in which the
// f32 slab products are guaranteed finite, or when there is no inner node at all.
// What build_search_nodes leaves on the host until the uploader has copied it.
struct SearchStaging {
    pt::UnfilledVector<PtAabb> own_boxes;
    QuadHierarchy quads;
};

int build_search_nodes(const PtSceneDesc& d, const PairNodes& tri_pairs, bool reference_topology, PtScene* scene, SearchStaging* staging, Uploader& up) {
    DeviceScene& dev = scene->dev;
    dev.tri_quads = nullptr;
    dev.sphere_base = 0x7fffffff;
    dev.search_count = d.triangle_count;
    dev.search_root = dev.tri_tree.root;
    std::memcpy(dev.search_box, dev.tri_tree.root_box, sizeof(dev.search_box));
    // A hollow sphere (negative radius, src/path_tracing.cpp:190) has an inverted box (src/geometry.cpp:51-57),
    // and the boxes the reference builds above it (min of minima, max of maxima, :38-49) need not contain it:
    // which rays reach it then depends on the reference's own tree.  Such spheres stay out of this
    // library's hierarchy; the reference's sphere tree is walked as it is.
    bool proper_spheres = true;
    for (int i = 0; i < d.sphere_count; ++i) proper_spheres = proper_spheres && d.spheres[i].radius >= 0.0;
    const bool sphere_tree_nests = d.sphere_count == 0 || dev.sph_tree.nested != 0;
    const bool merge = d.sphere_count > 1 && proper_spheres && sphere_tree_nests && !reference_topology && d.sphere_count + d.triangle_count > 2;
    if (d.triangle_count > 0 && dev.tri_tree.nested == 0) return PT_OK;     // the caller's triangle tree is searched as it is

    StageClock clock;
    pt::UnfilledVector<PtAabb>& own_boxes = staging->own_boxes;     // every primitive's box exactly as the reference's BVH leaves store it
    const long long triangle_count = d.triangle_count, box_count = triangle_count + (merge ? d.sphere_count : 0);
    own_boxes.resize(box_count);
    if (own_boxes.empty()) return PT_OK;
    auto in_range = [](const PtAabb& b) {                       // false for NaN too
        return std::fabs(b.min.x) <= MAX_COORD32 && std::fabs(b.min.y) <= MAX_COORD32 && std::fabs(b.min.z) <= MAX_COORD32 &&
               std::fabs(b.max.x) <= MAX_COORD32 && std::fabs(b.max.y) <= MAX_COORD32 && std::fabs(b.max.z) <= MAX_COORD32;
    };
    const int team = box_count >= (1 << 16) ? pt::host_threads() : 1;
    struct Piece { PtAabb all; bool in_range; };
    std::vector<Piece> pieces(team);
    pt::in_parts(0, box_count, team, [&](long long from, long long to, int part) {
        Piece mine{PtAabb{}, true};
        for (long long i = from; i < to; ++i) {
            const PtAabb b = i < triangle_count ? pt::box_of(d.triangles[i]) : pt::box_of(d.spheres[i - triangle_count]);
            own_boxes[i] = b;
            if (i == from) mine.all = b;
            mine.all.min.x = pt::min_of(mine.all.min.x, b.min.x); mine.all.min.y = pt::min_of(mine.all.min.y, b.min.y); mine.all.min.z = pt::min_of(mine.all.min.z, b.min.z);
            mine.all.max.x = pt::max_of(mine.all.max.x, b.max.x); mine.all.max.y = pt::max_of(mine.all.max.y, b.max.y); mine.all.max.z = pt::max_of(mine.all.max.z, b.max.z);
            mine.in_range = mine.in_range && in_range(b);
        }
        pieces[part] = mine;
    });
    PtAabb all = pieces[0].all;
    for (const Piece& piece : pieces) {
        const PtAabb& b = piece.all;
        all.min.x = pt::min_of(all.min.x, b.min.x); all.min.y = pt::min_of(all.min.y, b.min.y); all.min.z = pt::min_of(all.min.z, b.min.z);
        all.max.x = pt::max_of(all.max.x, b.max.x); all.max.y = pt::max_of(all.max.y, b.max.y); all.max.z = pt::max_of(all.max.z, b.max.z);
        if (!piece.in_range) return PT_OK;
    }
    const double extent[6] = {all.min.x, all.min.y, all.min.z, all.max.x, all.max.y, all.max.z};
    if (d.sphere_count == 1 && !in_range(pt::box_of(d.spheres[0]))) return PT_OK;    // the lone sphere's box is tested by the same kernel
    clock.done("primitive boxes");

    // What the hierarchy is built over: one reference per primitive, and several for large triangles that fill little of
    // their boxes (presplit.cpp).  Every reference names its primitive, so a search still ends at primitives; only the
    // boxes it goes through are tighter.  The binary hierarchy `tri_pairs` is over whole triangles and stays what the
    // exact-node search uses (there a leaf's box in the tree is the reference's own leaf box).
    PairNodes search_pairs;
    const PairNodes* pairs = &tri_pairs;
    pt::UnfilledVector<PtAabb> reference_boxes;
    std::vector<int32_t> reference_primitive;
    // (at most four million added references where that is less: a scene of ten million triangles is bound by memory, and more
    // nodes cost it what fewer triangle tests gain)
    const double presplit = std::min(presplit_fraction(), std::max(0.25, 4.0e6 / static_cast<double>(std::max<long long>(1, triangle_count))));
    long long cut = 0;
    // (not in a scene of a few triangles: there every level the hierarchy gains is a node step for almost every ray, and the
    // Cornell box's 36 wall triangles cut up cost it 16 %)
    if (presplit > 0.0 && triangle_count >= presplit_least_triangles() && !reference_topology) {
        cut = pt::presplit_triangles(d.triangles, triangle_count, own_boxes.data(), presplit_floor(), presplit, &reference_boxes, &reference_primitive);
        clock.done("large triangles cut into several references");
    }
    if (merge || cut > 0) {
        const PtAabb* boxes = own_boxes.data();
        long long references = box_count;
        if (cut > 0) {
            for (long long i = triangle_count; i < box_count; ++i) {       // the spheres, uncut, behind the triangles' references
                reference_boxes.push_back(own_boxes[i]);
                reference_primitive.push_back(static_cast<int32_t>(i));
            }
            boxes = reference_boxes.data();
            references = static_cast<long long>(reference_boxes.size());
        }
        int root = 0, height = 0;
        if (pt::build_sah_pairs(boxes, static_cast<int>(references), search_height_cap(), &search_pairs, &root, &height) != PT_OK) {
            pt::clear_error();
            if (merge) return PT_OK;
            cut = 0;                                        // the triangles' own hierarchy serves
        } else {
            if (cut > 0) {
                // leaves name references: put the primitives in their place
                pt::in_parts(0, static_cast<long long>(search_pairs.size()), search_pairs.size() >= (1u << 16) ? pt::host_threads() : 1, [&](long long from, long long to, int) {
                    for (long long i = from; i < to; ++i) {
                        PairNodeHost& p = search_pairs[i];
                        if (p.left < 0) p.left = ~reference_primitive[~p.left];
                        if (p.right < 0) p.right = ~reference_primitive[~p.right];
                    }
                });
            }
            pairs = &search_pairs;
        }
        clock.done("search hierarchy over the references built (SAH)");
    }
    if (pairs->empty()) return PT_OK;
    static_assert(sizeof(QuadNode32Host) == sizeof(QuadNode32) && sizeof(QuadBoxes64Host) == sizeof(QuadBoxes64), "host and device four-wide nodes must match");
    QuadHierarchy& quads = staging->quads;
    pt::collapse_to_quads(pairs->data(), pairs->size(), &quads);
    clock.done("collapsed to four-wide nodes");
    if (quads.nodes.size() > (size_t(1) << 25) || triangle_count > (1ll << 25)) return PT_OK;   // the kernel addresses a node, and a triangle record, by a 32-bit byte offset
    scene->tri_stack_need = quads.deepest_pending + 4;      // a visit writes four slots above the pending entries
    if (std::getenv("PT_VERBOSE") != nullptr) {
        std::fprintf(stderr, "pathtracer_amd: %zu primitives%s, %lld references more for large triangles, %zu pairs -> %zu four-wide nodes, %d stack slots per lane\n",
                     own_boxes.size(), merge ? " (triangles and spheres in one hierarchy)" : "", cut, pairs->size(), quads.nodes.size(), scene->tri_stack_need);
    }
    static_assert(sizeof(LeafBox) == sizeof(PtAabb), "a leaf box on the device is the six doubles of PtAabb: min x, y, z, max x, y, z");
    int rc = up.send(scene->tri_quads, quads.nodes.data(), quads.nodes.size() * sizeof(QuadNode32));
    if (rc != PT_OK) return rc;
    rc = up.send(scene->tri_quad_boxes, quads.exact.data(), quads.exact.size() * sizeof(QuadBoxes64));
    if (rc != PT_OK) return rc;
    rc = up.send(scene->tri_leaf_boxes, own_boxes.data(), own_boxes.size() * sizeof(LeafBox));
    if (rc != PT_OK) return rc;
    dev.tri_quads = scene->tri_quads.as<QuadNode32>();
    dev.tri_quad_boxes = scene->tri_quad_boxes.as<QuadBoxes64>();
    dev.tri_boxes = scene->tri_leaf_boxes.as<LeafBox>();
    dev.search_root = 0;                                     // the hierarchy's own numbering: its root is node 0
    dev.search_count = static_cast<int32_t>(own_boxes.size());
    std::memcpy(dev.search_box, extent, sizeof(dev.search_box));
    if (merge) dev.sphere_base = d.triangle_count;
    return PT_OK;
}

int check_frame(const PtFrame* f) {
    if (f == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "null frame");
    if (f->width < 2 || f->height < 2) return pt::fail(PT_ERR_INVALID_ARGUMENT, "frame %dx%d: width and height must be at least 2", f->width, f->height);
    if ((long long)f->width * f->height > (1ll << 31) / 4) return pt::fail(PT_ERR_INVALID_ARGUMENT, "frame %dx%d is too large", f->width, f->height);
    if (f->row_begin < 0 || f->row_end > f->height || f->row_begin > f->row_end) return pt::fail(PT_ERR_INVALID_ARGUMENT, "row range [%d, %d) is outside the %d-row image", f->row_begin, f->row_end, f->height);
    if (f->sample_begin < 0 || f->sample_count < 0) return pt::fail(PT_ERR_INVALID_ARGUMENT, "sample range must be non-negative");
    if (f->max_bounces < 0 || f->max_bounces >= (1 << 24)) return pt::fail(PT_ERR_INVALID_ARGUMENT, "max_bounces must be non-negative and below 2^24");
    if (f->shard_count > 1 && (f->shard_index < 0 || f->shard_index >= f->shard_count)) return pt::fail(PT_ERR_INVALID_ARGUMENT, "shard %d of %d does not exist", f->shard_index, f->shard_count);
    return PT_OK;
}

FrameParams frame_params(const PtFrame& f) {
    FrameParams p;
    p.width = f.width; p.height = f.height;
    p.row_begin = f.row_begin; p.row_end = f.row_end;
    p.sample_begin = f.sample_begin; p.sample_count = f.sample_count;
    p.max_bounces = f.max_bounces; p.pad_ = 0;
    p.shard_count = f.shard_count > 1 ? f.shard_count : 1;
    p.shard_index = f.shard_count > 1 ? f.shard_index : 0;
    p.aperture = f.aperture;
    p.pos = d3(f.camera_position); p.cx = d3(f.camera_x); p.cy = d3(f.camera_y);
    p.bl = d3(f.bottom_left); p.sx = d3(f.step_x); p.sy = d3(f.step_y);
    return p;
}

// 8x8 tiles of the frame's rows that belong to the frame's shard.
long long local_tiles(const PtFrame& f) {
    const long long tiles_x = (f.width + 7) / 8;
    const long long blocks_y = (f.row_end - f.row_begin + 7) / 8;
    const long long shards = f.shard_count > 1 ? f.shard_count : 1;
    const long long index = f.shard_count > 1 ? f.shard_index : 0;
    const long long tiles_y = (blocks_y - index + shards - 1) / shards;   // row blocks index, index+shards, ...
    return tiles_x * tiles_y;
}

// One wave per tile, four waves per block.
unsigned grid_blocks(const PtFrame& f) {
    return static_cast<unsigned>((local_tiles(f) + (BLOCK_THREADS / 64) - 1) / (BLOCK_THREADS / 64));
}

// src/main.cpp:1024-1034, one lane per pixel: level = std::min<real>(sqrt(sum / samples), 1.0f), byte =
// (unsigned char)(255.0f * level), where a NaN level converts as on x86-64 (to INT_MIN, low byte 0).
__global__ __launch_bounds__(BLOCK_THREADS) void tonemap_kernel(const double* __restrict__ accum, long long pixel_count, double samples,
                                                               uchar4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * BLOCK_THREADS + threadIdx.x;
    if (i >= pixel_count) return;
    const double2 bg = reinterpret_cast<const double2*>(accum)[2 * i];
    const double r = accum[4 * i + 2];
    auto level = [samples](double sum) { return (unsigned char)to_int_x86(255.0f * std_min(sqrt(sum / samples), 1.0f)); };
    out[i] = uchar4{level(bg.x), level(bg.y), level(r), 255};
}

// Tunables read from the environment at every call (measurement and cross-checking aids, not
// part of the interface): tests switch them between renders to compare the code paths.
struct Tunables {
    bool force_pixels_kernel = false;   // PT_KERNEL=pixels: always the one-lane-per-pixel kernel
    int stats = 0;                      // PT_STATS=1: phase/lane counters + box-test cross-check; 2: counters + per-section clocks
    double workspace_gb = -1.0;         // PT_WORKSPACE_GB: overrides the scene's cap of the per-path colour buffer (pt_scene_set_workspace_limit)
    int blocks_per_cu = 0;              // PT_BLOCKS_PER_CU: persistent blocks per CU (0 = occupancy query)
    bool force_f64_nodes = false;       // PT_NODES=f64: double-precision inner nodes in the wavefront kernel
    bool reference_topology = false;    // PT_BVH=reference: search the reference's own tree shape instead of the SAH rebuild
    int shade_threshold = 40;           // PT_SHADE_THRESHOLD / PT_LEAF_THRESHOLD: see WorkParams (swept: profiles/r3/thresholds_final_flow.log)
    int leaf_threshold = 12;
    int stack_slots = 0;                // PT_STACK_SLOTS: per-lane stack slots of the 4-wide search (0 = what the scene needs, capped)
    int distrust_mask = -1;             // PT_DISTRUST_MASK: see WorkParams (3 sends every fourth winner to the fallback)
    bool parking = true;                // PT_PARKING=0: no backing store for the lanes' stacks (a full stack ends the ordered search)
    Tunables() {
        if (const char* v = std::getenv("PT_PARKING")) parking = std::atoi(v) != 0;
        if (const char* v = std::getenv("PT_NODES")) force_f64_nodes = std::strcmp(v, "f64") == 0;
        if (const char* v = std::getenv("PT_BVH")) reference_topology = std::strcmp(v, "reference") == 0;
        if (const char* v = std::getenv("PT_STACK_SLOTS")) stack_slots = std::atoi(v);
        if (const char* v = std::getenv("PT_DISTRUST_MASK")) distrust_mask = std::atoi(v);
        if (const char* v = std::getenv("PT_SHADE_THRESHOLD")) shade_threshold = std::max(1, std::atoi(v));
        if (const char* v = std::getenv("PT_LEAF_THRESHOLD")) leaf_threshold = std::max(1, std::atoi(v));
        if (const char* v = std::getenv("PT_KERNEL")) force_pixels_kernel = std::strcmp(v, "pixels") == 0;
        if (const char* v = std::getenv("PT_STATS")) stats = std::min(2, std::max(0, std::atoi(v)));
        if (const char* v = std::getenv("PT_WORKSPACE_GB")) workspace_gb = std::atof(v);
        if (const char* v = std::getenv("PT_BLOCKS_PER_CU")) blocks_per_cu = std::atoi(v);
    }
};
Tunables tunables() { return Tunables(); }

int launch_pixels_kernel(PtScene* scene, const PtFrame* frame, double* accum_device, hipStream_t stream) {
    hipLaunchKernelGGL(render_pixels_kernel, dim3(grid_blocks(*frame)), dim3(BLOCK_THREADS), 0, stream, scene->dev, frame_params(*frame), accum_device);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

// Device memory a wavefront launch writes besides the accumulation buffer.
struct TraceOutputs {
    double* rgb = nullptr;
    unsigned int* counter = nullptr;
    unsigned long long* stats = nullptr;
    int* hit_triangle = nullptr;     // both null unless the launch reports first-vertex indices
    int* hit_sphere = nullptr;
    int* park = nullptr;             // backing store of the lanes' stacks (WorkParams::parking)
};

template <int STATS, bool NODE32, bool MERGED>
int launch_trace_paths(unsigned blocks, size_t stack_bytes, hipStream_t stream, const DeviceScene& dev, const FrameParams& fp,
                       const WorkParams& wp, const TraceOutputs& out, int* blocks_per_cu) {
    auto kernel = trace_paths_kernel<STATS, NODE32, MERGED>;
    if (blocks_per_cu != nullptr) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, BLOCK_THREADS, stack_bytes));
        return PT_OK;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(BLOCK_THREADS), stack_bytes, stream,
                       TraceArgs{dev, fp, wp, out.rgb, out.counter, out.stats, out.hit_triangle, out.hit_sphere, out.park});
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

// Dispatch over the compile-time switches; with `blocks_per_cu` set it only queries occupancy.
template <int STATS>
int trace_paths_variant(bool node32, bool merged, unsigned blocks, size_t stack_bytes, hipStream_t stream, const DeviceScene& dev,
                        const FrameParams& fp, const WorkParams& wp, const TraceOutputs& out, int* blocks_per_cu) {
    if (!node32) return launch_trace_paths<STATS, false, false>(blocks, stack_bytes, stream, dev, fp, wp, out, blocks_per_cu);
    if (merged) return launch_trace_paths<STATS, true, true>(blocks, stack_bytes, stream, dev, fp, wp, out, blocks_per_cu);
    return launch_trace_paths<STATS, true, false>(blocks, stack_bytes, stream, dev, fp, wp, out, blocks_per_cu);
}

int trace_paths(int stats_build, bool node32, unsigned blocks, size_t stack_bytes, hipStream_t stream, const DeviceScene& dev, const FrameParams& fp,
                const WorkParams& wp, const TraceOutputs& out, int* blocks_per_cu) {
    const bool merged = node32 && dev.sphere_base != 0x7fffffff;
    if (stats_build == 1) return trace_paths_variant<1>(node32, merged, blocks, stack_bytes, stream, dev, fp, wp, out, blocks_per_cu);
    if (stats_build == 2) return trace_paths_variant<2>(node32, merged, blocks, stack_bytes, stream, dev, fp, wp, out, blocks_per_cu);
    return trace_paths_variant<0>(node32, merged, blocks, stack_bytes, stream, dev, fp, wp, out, blocks_per_cu);
}

// Where a launch keeps its per-path colours and its counters: the scene's own buffers, allocated and grown
// on demand (`fixed` false), or memory the caller of launch_wavefront provides and nothing may be allocated.
struct Workspace {
    double* rgb = nullptr;
    size_t rgb_bytes = 0;
    void* counters = nullptr;      // 1024 bytes: unit counter, then (at +512) the profiling counters
    bool fixed = false;
    int* hit_triangle = nullptr;   // set by pt_primary_hits: the launch reports the first-vertex indices of both traversals
    int* hit_sphere = nullptr;
};

// The scene's own workspace, grown to `want` bytes of colour records if the device has them, else to what it
// has (down to `least`): a smaller workspace only means more sample batches, never other bytes.
int grow_workspace(PtScene* scene, size_t want, size_t least) {
    if (scene->counters.p == nullptr) HIP_TRY(hipMalloc(&scene->counters.p, 1024));
    if (scene->rgb_bytes >= want) return PT_OK;      // (a buffer larger than a lowered limit was released by pt_scene_set_workspace_limit)
    if (scene->rgb.p != nullptr) { HIP_TRY(hipFree(scene->rgb.p)); scene->rgb.p = nullptr; scene->rgb_bytes = 0; }
    size_t free_bytes = 0, total_bytes = 0;
    HIP_TRY(hipMemGetInfo(&free_bytes, &total_bytes));
    size_t bytes = std::min(want, std::max(least, free_bytes - free_bytes / 8));
    bytes = least * std::max<size_t>(1, bytes / least);          // whole sample groups
    for (;;) {
        const hipError_t e = hipMalloc(&scene->rgb.p, bytes);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        scene->rgb.p = nullptr;
        if (bytes <= least) return pt::fail(PT_ERR_OUT_OF_MEMORY, "the colour workspace of one sample group (%zu bytes) does not fit the device: %s", least, hipGetErrorString(e));
        bytes = least * std::max<size_t>(1, bytes / least / 2);
    }
    scene->rgb_bytes = bytes;
    return PT_OK;
}

// Persistent wavefront path: sample batches sized to the workspace, each batch = trace + ordered sum.
int launch_wavefront(PtScene* scene, const PtFrame* frame, double* accum_device, hipStream_t stream, Workspace ws) {
    const Tunables t = tunables();
    const long long tiles_x = (frame->width + 7) / 8;
    const long long real_tiles = local_tiles(*frame);
    const size_t bytes_per_group = static_cast<size_t>(real_tiles) * 4 * CHUNK_PATHS * sizeof(double);
    const int total_groups = (frame->sample_count + SAMPLES_PER_CHUNK - 1) / SAMPLES_PER_CHUNK;
    if (ws.fixed) {
        if (bytes_per_group > ws.rgb_bytes) {
            return pt::fail(PT_ERR_UNSUPPORTED, "the frame's rows hold %lld tiles of 8x8 pixels; a launch that may not allocate takes at most %lld "
                            "(render the frame in several row ranges)", real_tiles, static_cast<long long>(ws.rgb_bytes / (4 * CHUNK_PATHS * sizeof(double))));
        }
    } else {
        const size_t cap = t.workspace_gb >= 0.0 ? static_cast<size_t>(t.workspace_gb * 1024.0 * 1024.0 * 1024.0) : scene->workspace_limit;
        const size_t wanted_groups = std::min<size_t>(std::max<size_t>(1, cap / bytes_per_group), total_groups);
        const int rc = grow_workspace(scene, bytes_per_group * wanted_groups, bytes_per_group);
        if (rc != PT_OK) return rc;
        ws.rgb = static_cast<double*>(scene->rgb.p);
        ws.rgb_bytes = std::min(scene->rgb_bytes, bytes_per_group * wanted_groups);    // PT_WORKSPACE_GB may ask for less than an earlier call left
        ws.counters = scene->counters.p;
    }
    const int groups_per_batch = static_cast<int>(std::min<size_t>(ws.rgb_bytes / bytes_per_group, total_groups));
    scene->last_counters = ws.counters;
    const bool node32 = scene->dev.tri_quads != nullptr && !t.force_f64_nodes;
    // per-lane stack slots: the triangle search's need, and the sphere tree's when it is walked in place
    // (a lane of the 4-wide search that would outgrow its slots falls back to the exhaustive traversal, so
    // any number >= 3 is correct there; the binary search and the sphere tree need their full height)
    // A launch that may allocate gives the four-wide search a backing store in device memory for the oldest entries of a
    // full stack; 24 slots per lane in LDS then serve any depth, and four blocks fit a CU.  One that may not allocate
    // gives it up to 36 slots (three blocks per CU when a scene needs more than 24) and no backing store.
    const bool may_park = node32 && !ws.fixed && t.parking;
    const int quad_slots = std::max(4, std::min(t.stack_slots > 0 ? t.stack_slots : static_cast<int>(may_park ? QUAD_STACK_SLOTS : MAX_WF_STACK_SLOTS),
                                                scene->tri_stack_need));
    const int stack_slots = std::min(std::max(std::max(node32 ? quad_slots : scene->tri_height, scene->sph_height), 4),
                                     static_cast<int>(MAX_WF_STACK_SLOTS));
    bool parking = may_park && scene->tri_stack_need > stack_slots && stack_slots >= PARK_UNIT + 4;
    const size_t stack_bytes = static_cast<size_t>(stack_slots) * BLOCK_THREADS * sizeof(int);
    const int variant = 4 * stack_slots + 2 * (t.stats & 1) + (node32 ? 1 : 0) + 1024 * (t.stats >> 1);
    if (scene->compute_units == 0 || scene->occupancy_variant != variant || t.blocks_per_cu > 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, scene->device));
        scene->compute_units = prop.multiProcessorCount;
        scene->occupancy_variant = variant;
        int per_cu = 0;
        const int rc = trace_paths(t.stats, node32, 0, stack_bytes, stream, scene->dev, FrameParams{}, WorkParams{}, TraceOutputs{}, &per_cu);
        if (rc != PT_OK) return rc;
        scene->blocks_per_cu = t.blocks_per_cu > 0 ? t.blocks_per_cu : std::max(1, per_cu);
    }
    if (parking) {
        // a search never has more entries pending than the hierarchy's deepest need, so never more parked units than this
        const size_t units = (static_cast<size_t>(scene->tri_stack_need) + PARK_UNIT - 1) / PARK_UNIT;
        const size_t bytes = units * PARK_UNIT * sizeof(int) * BLOCK_THREADS * static_cast<size_t>(scene->compute_units) * static_cast<size_t>(scene->blocks_per_cu);
        if (scene->park_bytes < bytes) {
            if (scene->park.p != nullptr) { HIP_TRY(hipFree(scene->park.p)); scene->park.p = nullptr; scene->park_bytes = 0; }
            if (hipMalloc(&scene->park.p, bytes) == hipSuccess) {
                scene->park_bytes = bytes;
            } else {
                (void)hipGetLastError();
                scene->park.p = nullptr;
                parking = false;            // full stacks end their ordered searches instead: the same bytes, slower
            }
        }
    }
    unsigned int* counter = static_cast<unsigned int*>(ws.counters);
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(static_cast<char*>(ws.counters) + 512);
    HIP_TRY(hipMemsetAsync(ws.counters, 0, 1024, stream));   // the unit counter + profiling counters

    FrameParams fp = frame_params(*frame);
    const int sample_end = frame->sample_begin + frame->sample_count;
    bool first = true;
    for (int begin = frame->sample_begin; begin < sample_end; begin += groups_per_batch * SAMPLES_PER_CHUNK) {
        fp.sample_begin = begin;
        fp.sample_count = std::min(groups_per_batch * SAMPLES_PER_CHUNK, sample_end - begin);
        WorkParams wp;
        wp.tiles_x = static_cast<int>(tiles_x);
        wp.groups_per_tile = (fp.sample_count + SAMPLES_PER_CHUNK - 1) / SAMPLES_PER_CHUNK;
        const long long tile_rows = real_tiles / tiles_x;
        const long long units = ((tiles_x + 3) / 4) * ((tile_rows + 3) / 4) * wp.groups_per_tile * SAMPLES_PER_CHUNK * SAMPLES_PER_CHUNK;   // 64 paths each
        wp.tile_count = static_cast<unsigned int>(real_tiles);
        if (units >= (1ll << 31)) return pt::fail(PT_ERR_UNSUPPORTED, "frame too large for one batch: %lld units of 64 paths", units);
        wp.unit_count = static_cast<unsigned int>(units);
        wp.shade_threshold = t.shade_threshold;
        wp.leaf_threshold = t.leaf_threshold;
        wp.stack_slots = stack_slots;
        wp.parking = parking ? 1 : 0;
        wp.distrust_mask = t.distrust_mask;
        wp.export_hits = ws.hit_triangle != nullptr ? 1 : 0;
        if (!first) HIP_TRY(hipMemsetAsync(counter, 0, 512, stream));
        first = false;
        const unsigned blocks = static_cast<unsigned>(std::min<long long>(static_cast<long long>(scene->compute_units) * scene->blocks_per_cu,
                                                                          (units + 3) / 4));
        const int rc = trace_paths(t.stats, node32, blocks, stack_bytes, stream, scene->dev, fp, wp,
                                   TraceOutputs{ws.rgb, counter, stats, ws.hit_triangle, ws.hit_sphere, parking ? static_cast<int*>(scene->park.p) : nullptr}, nullptr);
        if (rc != PT_OK) return rc;
        hipLaunchKernelGGL(sum_samples_kernel, dim3(grid_blocks(*frame)), dim3(BLOCK_THREADS), 0, stream,
                           fp, wp, static_cast<const double*>(ws.rgb), accum_device);
        HIP_TRY(hipGetLastError());
    }
    scene->has_stats = t.stats != 0;
    return PT_OK;
}

// Scenes the wavefront kernel renders (any other goes through the one-lane-per-pixel kernel): bounded albedos (see
// pt_wavefront.hpp), trees no deeper than its stacks, and a triangle tree whose boxes nest and are finite (its node steps
// skip what is entered beyond the best hit and use the short form of the slab test; a sphere tree of another kind is
// walked without skipping and with the exact form, see TreeRef).
bool wavefront_serves(const PtScene* scene) {
    const TreeRef& triangles = scene->dev.tri_tree;
    return scene->albedo_bounded && !tunables().force_pixels_kernel && scene->tri_height <= MAX_WF_STACK_SLOTS &&
           scene->sph_height <= MAX_WF_STACK_SLOTS && (triangles.count == 0 || (triangles.nested != 0 && triangles.bounded != 0));
}

int launch_render(PtScene* scene, const PtFrame* frame, double* accum_device, hipStream_t stream, Workspace ws) {
    if (frame->row_begin == frame->row_end || frame->sample_count == 0 || grid_blocks(*frame) == 0) return PT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    // A stream that is being captured into a graph takes the kernels and the memsets only: events recorded outside the
    // capture may not be queried or waited for there, and the graph itself orders what it holds.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream != nullptr) HIP_TRY(hipStreamIsCapturing(stream, &capture));
    const bool capturing = capture != hipStreamCaptureStatusNone;
    // A handle owns one work counter, one colour workspace and one pair of timing events.  Launches on one stream are
    // ordered by the stream; a launch on another stream is put behind the end of the handle's previous launch.  (Frames
    // that are meant to overlap take one handle each.)
    if (!capturing && scene->timed && stream != scene->last_stream) HIP_TRY(hipStreamWaitEvent(stream, scene->stop, 0));
    if (!capturing && ws.fixed) {
        std::lock_guard<std::mutex> guard(static_scratch_lock);
        if (static_scratch_used[scene->device]) HIP_TRY(hipStreamWaitEvent(stream, static_scratch_free[scene->device], 0));
    }
    scene->has_stats = false;
    scene->timed = false;        // until this launch is enqueued in full there is no pair of events to read a time from
    if (!capturing) HIP_TRY(hipEventRecord(scene->start, stream));
    // The wavefront kernel needs bounded albedos and at least one bounce (see pt_wavefront.hpp);
    // anything else goes through the one-lane-per-pixel kernel, which carries the colour explicitly.
    const bool wavefront = frame->max_bounces >= 1 && wavefront_serves(scene);
    const int rc = wavefront ? launch_wavefront(scene, frame, accum_device, stream, ws) : launch_pixels_kernel(scene, frame, accum_device, stream);
    if (rc != PT_OK) return rc;
    if (capturing) return PT_OK;
    HIP_TRY(hipEventRecord(scene->stop, stream));
    if (ws.fixed) {
        std::lock_guard<std::mutex> guard(static_scratch_lock);
        HIP_TRY(hipEventRecord(static_scratch_free[scene->device], stream));
        static_scratch_used[scene->device] = true;
    }
    scene->last_stream = stream;
    scene->timed = true;
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_scene_create(const PtSceneDesc* d, int32_t device, PtScene** out) {
    if (d == nullptr || out == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_create: null argument");
    *out = nullptr;
    if (d->material_count <= 0 || d->materials == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_create: a scene needs at least one material");
    if (d->sphere_count < 0 || d->triangle_count < 0) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_create: negative primitive count");
    for (int i = 0; i < d->material_count; ++i) {
        if (d->materials[i].type < PT_LAMBERTIAN || d->materials[i].type > PT_DIFFUSE_LIGHT) return pt::fail(PT_ERR_INVALID_ARGUMENT, "material %d has unknown type %d", i, d->materials[i].type);
    }
    if (d->sphere_count > 0 && (d->spheres == nullptr || d->sphere_bvh == nullptr || d->sphere_material_indices == nullptr)) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_create: sphere arrays missing");
    if (d->triangle_count > 0 && (d->triangles == nullptr || d->triangle_bvh == nullptr || d->triangle_material_indices == nullptr)) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_create: triangle arrays missing");
    for (int i = 0; i < d->sphere_count; ++i) {
        const int m = d->sphere_material_indices[i];
        if (m < 0 || m >= d->material_count) return pt::fail(PT_ERR_INVALID_ARGUMENT, "sphere %d names material %d of %d", i, m, d->material_count);
    }
    for (int i = 0; i < d->triangle_count; ++i) {
        const int m = d->triangle_material_indices[i];
        if (m < 0 || m >= d->material_count) return pt::fail(PT_ERR_INVALID_ARGUMENT, "triangle %d names material %d of %d", i, m, d->material_count);
    }

    int device_count = 0;
    HIP_TRY(hipGetDeviceCount(&device_count));
    if (device < 0 || device >= device_count) return pt::fail(PT_ERR_DEVICE, "device %d requested, %d visible", device, device_count);
    HIP_TRY(hipSetDevice(device));

    std::unique_ptr<PtScene> scene(new PtScene());
    scene->device = device;
    int rc;
    // Host arrays the uploader reads until up.finish(): declared before it, so that they outlive it on every way out.
    PairNodes pairs;
    pt::UnfilledVector<TriRecord> records;
    SearchStaging staging;
    Uploader up(device);
    if (d->triangle_count > 0) {
        rc = build_tree(d->triangle_bvh, d->triangle_bvh_node_count, d->triangle_count, "triangle", scene->tri_pairs, scene->tri_ref_nodes,
                        &scene->dev.tri_tree, &scene->tri_height, &pairs, !tunables().reference_topology, up);
        if (rc != PT_OK) return rc;
        // ray-independent part of the triangle test, same operations as src/geometry.cpp:111-114, :124, :84-90
        StageClock clock;
        records.resize(d->triangle_count);
        pt::in_parts(0, d->triangle_count, d->triangle_count >= (1 << 14) ? pt::host_threads() : 1, [&](long long from, long long to, int) {
            for (long long i = from; i < to; ++i) {
                const PtTriangle& t = d->triangles[i];
                using namespace pt;
                const PtVec3 ab = t.b - t.a, ac = t.c - t.a;
                const PtVec3 n3 = cross(ab, ac);
                const double nn = dot(n3, n3);
                const PtVec3 un = (1.0f / std::sqrt(nn)) * n3;
                TriRecord& r = records[i];
                r.a[0] = t.a.x; r.a[1] = t.a.y; r.a[2] = t.a.z;
                r.ab[0] = ab.x; r.ab[1] = ab.y; r.ab[2] = ab.z;
                r.ac[0] = ac.x; r.ac[1] = ac.y; r.ac[2] = ac.z;
                r.nn = nn;
                r.un[0] = un.x; r.un[1] = un.y; r.un[2] = un.z;
                r.material = d->triangle_material_indices[i];
                for (int32_t& unused : r.pad_) unused = 0;
            }
        });
        rc = up.send(scene->tri_records, records.data(), records.size() * sizeof(TriRecord));
        if (rc != PT_OK) return rc;
        scene->dev.tris = scene->tri_records.as<TriRecord>();
        clock.done("triangle records made");
    }
    if (d->sphere_count > 0) {
        rc = build_tree(d->sphere_bvh, d->sphere_bvh_node_count, d->sphere_count, "sphere", scene->sph_pairs, scene->sph_ref_nodes,
                        &scene->dev.sph_tree, &scene->sph_height, nullptr, false, up);
        if (rc != PT_OK) return rc;
        rc = upload(scene->spheres, d->spheres, sizeof(PtSphere) * d->sphere_count);
        if (rc != PT_OK) return rc;
        rc = upload(scene->sph_material, d->sphere_material_indices, sizeof(int32_t) * d->sphere_count);
        if (rc != PT_OK) return rc;
        scene->dev.spheres = scene->spheres.as<double>();
        scene->dev.sph_material = scene->sph_material.as<int32_t>();
    }
    rc = build_search_nodes(*d, pairs, tunables().reference_topology, scene.get(), &staging, up);
    if (rc != PT_OK) return rc;
    {
        StageClock clock;
        rc = up.finish();
        if (rc != PT_OK) return rc;
        clock.done("waited for the copies to the device");
    }
    scene->albedo_bounded = true;
    scene->dev.with_metals = 0;
    for (int i = 0; i < d->material_count; ++i) {
        const PtMaterial& m = d->materials[i];
        if (m.type == PT_METAL) scene->dev.with_metals = 1;
        if (m.type == PT_LAMBERTIAN || m.type == PT_METAL) {
            for (int c = 0; c < 3; ++c) scene->albedo_bounded = scene->albedo_bounded && std::fabs(m.v[c]) <= 1.0;   // false for NaN too
        }
    }
    rc = upload(scene->materials, d->materials, sizeof(PtMaterial) * d->material_count);
    if (rc != PT_OK) return rc;
    scene->dev.materials = scene->materials.as<PtMaterial>();
    scene->dev.bg0 = D3{d->background_gradient_start.r, d->background_gradient_start.g, d->background_gradient_start.b};
    scene->dev.bg1 = D3{d->background_gradient_end.r, d->background_gradient_end.g, d->background_gradient_end.b};
    HIP_TRY(hipEventCreate(&scene->start));
    HIP_TRY(hipEventCreate(&scene->stop));
    rc = static_scratch_of(device, &scene->static_workspace);     // this device is current: resolved once per device
    if (rc != PT_OK) return rc;
    *out = scene.release();
    return PT_OK;
}

void pt_scene_destroy(PtScene* scene) {
    if (scene == nullptr) return;
    (void)hipSetDevice(scene->device);
    delete scene;
}

int pt_render_device(PtScene* scene, const PtFrame* frame, double* accum_device, void* hip_stream) {
    if (scene == nullptr || accum_device == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_render_device: null argument");
    const int rc = check_frame(frame);
    if (rc != PT_OK) return rc;
    return launch_render(scene, frame, accum_device, static_cast<hipStream_t>(hip_stream), Workspace{});
}

int pt_viewer_frame(PtScene* scene, PtCamera* camera, const PtViewerInput* input, double aspect_ratio, int32_t width, int32_t height,
                    int32_t samples_per_frame, int32_t* sample, double* accum_device, void* hip_stream) {
    if (scene == nullptr || camera == nullptr || sample == nullptr || accum_device == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_viewer_frame: null argument");
    if (samples_per_frame < 1 || *sample < 0) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_viewer_frame: samples_per_frame must be positive and *sample non-negative");
    // Nothing of the caller's changes unless the frame is enqueued: the controls move a copy of the camera, and the copy,
    // the sample counter and (where the camera moved) the cleared buffer are handed back together at the end.
    PtCamera moved = *camera;
    int32_t modified = 0;
    if (input != nullptr) {
        const int rc = pt_camera_apply_input(&moved, input, &modified);
        if (rc != PT_OK) return rc;
    }
    PtFrame frame;
    int rc = pt_camera_frame(&moved, aspect_ratio, width, height, &frame);                     // src/main.cpp:987-994
    if (rc != PT_OK) return rc;
    rc = check_frame(&frame);
    if (rc != PT_OK) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipSetDevice(scene->device));
    const int32_t first_sample = modified != 0 ? 0 : *sample;                                  // :996-999
    frame.sample_begin = first_sample;
    frame.sample_count = samples_per_frame;
    if (modified != 0) {
        // the buffer is cleared on the frame's own stream, ahead of its kernels; should the launch then be refused, the
        // buffer is cleared and the camera and the counter are not moved: the caller's next frame starts the picture anew
        // from the camera he still holds, which is what a cleared buffer asks for
        HIP_TRY(hipMemsetAsync(accum_device, 0, 4ull * width * height * sizeof(double), stream));
    }
    rc = launch_render(scene, &frame, accum_device, stream, Workspace{});                      // :1001-1022
    if (rc != PT_OK) {
        if (modified != 0) *sample = 0;       // the buffer holds no sample any more
        return rc;
    }
    *camera = moved;
    *sample = first_sample + samples_per_frame;                                                // :1036
    return PT_OK;
}

int pt_scene_set_workspace_limit(PtScene* scene, uint64_t bytes) {
    if (scene == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_scene_set_workspace_limit: null scene");
    HIP_TRY(hipSetDevice(scene->device));
    scene->workspace_limit = static_cast<size_t>(bytes);
    if (scene->rgb.p != nullptr && scene->rgb_bytes > scene->workspace_limit) {
        // what the handle holds beyond the new limit goes back to the device now (after the launches that use it)
        if (scene->timed) HIP_TRY(hipEventSynchronize(scene->stop));
        HIP_TRY(hipFree(scene->rgb.p));
        scene->rgb.p = nullptr;
        scene->rgb_bytes = 0;
    }
    return PT_OK;
}

int pt_last_launch_shape(PtScene* scene, int32_t* blocks_per_cu, int32_t* compute_units, uint64_t* workspace_bytes) {
    if (scene == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_last_launch_shape: null scene");
    if (scene->compute_units == 0) return pt::fail(PT_ERR_UNSUPPORTED, "pt_last_launch_shape: this handle has not run the wavefront kernel yet");
    if (blocks_per_cu != nullptr) *blocks_per_cu = scene->blocks_per_cu;
    if (compute_units != nullptr) *compute_units = scene->compute_units;
    if (workspace_bytes != nullptr) *workspace_bytes = scene->rgb_bytes;
    return PT_OK;
}

int pt_render_device_noalloc(PtScene* scene, const PtFrame* frame, double* accum_device, void* hip_stream) {
    if (scene == nullptr || accum_device == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_render_device_noalloc: null argument");
    const int rc = check_frame(frame);
    if (rc != PT_OK) return rc;
    if (local_tiles(*frame) > PT_NOALLOC_MAX_TILES) {
        return pt::fail(PT_ERR_UNSUPPORTED, "pt_render_device_noalloc: the frame's rows hold %lld tiles of 8x8 pixels, the static workspace takes %d "
                        "(render the frame in several row ranges)", local_tiles(*frame), static_cast<int>(PT_NOALLOC_MAX_TILES));
    }
    // the address hipGetSymbolAddress gave for this scene's device when the scene was created
    unsigned char* const base = static_cast<unsigned char*>(scene->static_workspace);
    Workspace ws;
    ws.counters = base;
    ws.rgb = reinterpret_cast<double*>(base + STATIC_COUNTER_BYTES);
    ws.rgb_bytes = STATIC_COLOUR_BYTES;
    ws.fixed = true;
    return launch_render(scene, frame, accum_device, static_cast<hipStream_t>(hip_stream), ws);
}

int pt_tonemap_device(int32_t device, const double* accum_device, int64_t pixel_count, int32_t sample_count, uint8_t* bgra8_device,
                      void* hip_stream) {
    if (accum_device == nullptr || bgra8_device == nullptr || pixel_count < 0 || sample_count <= 0) {
        return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_tonemap_device: bad argument");
    }
    if (pixel_count == 0) return PT_OK;
    HIP_TRY(hipSetDevice(device));
    const long long blocks = (pixel_count + BLOCK_THREADS - 1) / BLOCK_THREADS;
    if (blocks > 0x7fffffffll) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_tonemap_device: %lld pixels are too many", (long long)pixel_count);
    hipLaunchKernelGGL(tonemap_kernel, dim3(static_cast<unsigned>(blocks)), dim3(BLOCK_THREADS), 0, static_cast<hipStream_t>(hip_stream),
                       accum_device, static_cast<long long>(pixel_count), static_cast<double>(sample_count),
                       reinterpret_cast<uchar4*>(bgra8_device));
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int pt_render(PtScene* scene, const PtFrame* frame, double* accum) {
    if (scene == nullptr || accum == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_render: null argument");
    int rc = check_frame(frame);
    if (rc != PT_OK) return rc;
    if (frame->row_begin == frame->row_end || frame->sample_count == 0) return PT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    // only the frame's rows travel: they are contiguous in the row-major buffer
    const size_t offset = 4ull * frame->row_begin * frame->width;
    const size_t bytes = 4ull * (frame->row_end - frame->row_begin) * frame->width * sizeof(double);
    DeviceBuffer rows;
    HIP_TRY(hipMalloc(&rows.p, bytes));
    HIP_TRY(hipMemcpy(rows.p, accum + offset, bytes, hipMemcpyHostToDevice));
    rc = launch_render(scene, frame, static_cast<double*>(rows.p) - offset, nullptr, Workspace{});
    if (rc != PT_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(accum + offset, rows.p, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_primary_hits(PtScene* scene, const PtFrame* frame, int32_t* triangle_index, int32_t* sphere_index) {
    if (scene == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_primary_hits: null scene");
    const int rc = check_frame(frame);
    if (rc != PT_OK) return rc;
    if (frame->row_begin == frame->row_end) return PT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    const size_t pixels = static_cast<size_t>(frame->width) * frame->height;
    DeviceBuffer tri, sph;
    HIP_TRY(hipMalloc(&tri.p, pixels * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&sph.p, pixels * sizeof(int32_t)));
    // rows of the range that belong to another shard keep what the caller's buffers hold
    const size_t offset = static_cast<size_t>(frame->row_begin) * frame->width;
    const size_t count = static_cast<size_t>(frame->row_end - frame->row_begin) * frame->width;
    if (triangle_index != nullptr) HIP_TRY(hipMemcpy(static_cast<int32_t*>(tri.p) + offset, triangle_index + offset, count * sizeof(int32_t), hipMemcpyHostToDevice));
    if (sphere_index != nullptr) HIP_TRY(hipMemcpy(static_cast<int32_t*>(sph.p) + offset, sphere_index + offset, count * sizeof(int32_t), hipMemcpyHostToDevice));
    // The indices come from the kernel that renders: the persistent wavefront kernel, run over the frame's
    // rows for one sample and one vertex per path with its two searches kept from shortening each other.
    // Scenes (and PT_KERNEL=pixels) that render with the one-lane-per-pixel kernel use that kernel's search.
    const bool wavefront = wavefront_serves(scene);
    if (wavefront) {
        PtFrame one = *frame;
        one.sample_count = 1;
        one.max_bounces = 1;
        DeviceBuffer unused_sums;      // the launch also adds the (one-vertex) colours somewhere
        HIP_TRY(hipMalloc(&unused_sums.p, pixels * 4 * sizeof(double)));
        Workspace ws;
        ws.hit_triangle = static_cast<int*>(tri.p);
        ws.hit_sphere = static_cast<int*>(sph.p);
        const int launched = launch_render(scene, &one, static_cast<double*>(unused_sums.p), nullptr, ws);
        if (launched != PT_OK) return launched;
        HIP_TRY(hipDeviceSynchronize());
    } else {
        hipLaunchKernelGGL(primary_hits_kernel, dim3(grid_blocks(*frame)), dim3(BLOCK_THREADS), 0, nullptr, scene->dev, frame_params(*frame),
                           static_cast<int*>(tri.p), static_cast<int*>(sph.p));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    if (triangle_index != nullptr) HIP_TRY(hipMemcpy(triangle_index + offset, static_cast<int32_t*>(tri.p) + offset, count * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (sphere_index != nullptr) HIP_TRY(hipMemcpy(sphere_index + offset, static_cast<int32_t*>(sph.p) + offset, count * sizeof(int32_t), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_last_render_ms(PtScene* scene, float* ms) {
    if (scene == nullptr || ms == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_last_render_ms: null argument");
    if (!scene->timed) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_last_render_ms: nothing has been rendered on this scene");
    HIP_TRY(hipSetDevice(scene->device));
    HIP_TRY(hipEventSynchronize(scene->stop));
    HIP_TRY(hipEventElapsedTime(ms, scene->start, scene->stop));
    return PT_OK;
}

int pt_last_render_stats(PtScene* scene, uint64_t* out) {
    if (scene == nullptr || out == nullptr) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_last_render_stats: null argument");
    if (!scene->has_stats) return pt::fail(PT_ERR_UNSUPPORTED, "pt_last_render_stats: the last render did not run the wavefront kernel with PT_STATS=1");
    HIP_TRY(hipSetDevice(scene->device));
    HIP_TRY(hipEventSynchronize(scene->stop));
    HIP_TRY(hipMemcpy(out, static_cast<char*>(scene->last_counters) + 512, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_device_eval(int32_t device, int32_t op, const void* in, int64_t n, void* out) {
    static const int IN_BYTES[] = {4, 12, 4, 24, 96, 80, 120, 72, 136, 16};
    static const int OUT_BYTES[] = {4, 4, 16, 24, 16, 24, 16, 24, 56, 16};
    if (op < 0 || op > PT_EVAL_SQRT_DIV) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_device_eval: unknown op %d", op);
    if (in == nullptr || out == nullptr || n < 0) return pt::fail(PT_ERR_INVALID_ARGUMENT, "pt_device_eval: bad argument");
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(device));
    DeviceBuffer din, dout;
    HIP_TRY(hipMalloc(&din.p, static_cast<size_t>(n) * IN_BYTES[op]));
    HIP_TRY(hipMalloc(&dout.p, static_cast<size_t>(n) * OUT_BYTES[op]));
    HIP_TRY(hipMemcpy(din.p, in, static_cast<size_t>(n) * IN_BYTES[op], hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, op, din.p, static_cast<long long>(n), dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, static_cast<size_t>(n) * OUT_BYTES[op], hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // extern "C"
