// Digests of what the host-side builders make for a built-in scene: the reference-shaped BVH (node for node), the
// binary search hierarchy and its four-wide form.  Linked against the builders' object files; the CPU tests run it
// with different thread counts (PT_BUILD_THREADS) and expect the same lines.
// usage: hierarchy_digest <scene kind> <models dir> [grid_n]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../pathtracer_amd/csrc/pt_internal.h"
#include "../../pathtracer_amd/csrc/pt_traverse_host.h"

static unsigned long long fnv(const void* data, size_t bytes) {
    unsigned long long h = 1469598103934665603ull;
    const unsigned long long* w = static_cast<const unsigned long long*>(data);
    for (size_t i = 0; i < bytes / 8; ++i) { h ^= w[i]; h *= 1099511628211ull; }
    return h;
}
static double seconds_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <scene kind> <models dir> [grid_n]\n", argv[0]); return 2; }
    const bool timing = std::getenv("PT_DIGEST_TIMES") != nullptr;
    PtHostScene* host = nullptr;
    auto t = std::chrono::steady_clock::now();
    if (pt_host_scene_build(std::atoi(argv[1]), argv[2], argc > 3 ? std::atoi(argv[3]) : 0, &host) != PT_OK) {
        std::fprintf(stderr, "%s\n", pt_last_error());
        return 1;
    }
    if (timing) std::fprintf(stderr, "host scene %.3f s\n", seconds_since(t));
    PtSceneDesc d;
    pt_host_scene_desc(host, &d);
    std::printf("triangles %d %016llx\n", d.triangle_count, fnv(d.triangles, sizeof(PtTriangle) * (size_t)d.triangle_count));
    std::printf("reference nodes %d %016llx\n", d.triangle_bvh_node_count, fnv(d.triangle_bvh, sizeof(PtBvhNode) * (size_t)d.triangle_bvh_node_count));
    std::vector<PtAabb> boxes(d.triangle_count);
    for (int i = 0; i < d.triangle_bvh_node_count; ++i) {
        if (d.triangle_bvh[i].left == 0 && d.triangle_bvh[i].right == 0) boxes[d.triangle_bvh[i].index] = d.triangle_bvh[i].aabb;
    }
    pt::UnfilledVector<ptd::PairNodeHost> pairs;
    int root = 0, height = 0;
    t = std::chrono::steady_clock::now();
    if (pt::build_sah_pairs(boxes.data(), (int)boxes.size(), 32, &pairs, &root, &height) != PT_OK) { std::fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (timing) std::fprintf(stderr, "search hierarchy %.3f s\n", seconds_since(t));
    for (ptd::PairNodeHost& p : pairs) std::memset(p.pad_, 0, sizeof(p.pad_));
    std::printf("pairs %zu root %d height %d %016llx\n", pairs.size(), root, height, fnv(pairs.data(), pairs.size() * sizeof(ptd::PairNodeHost)));
    ptd::QuadHierarchy quads;
    t = std::chrono::steady_clock::now();
    pt::collapse_to_quads(pairs.data(), pairs.size(), &quads);
    if (timing) std::fprintf(stderr, "four-wide %.3f s\n", seconds_since(t));
    std::printf("four-wide %zu pending %d %016llx %016llx\n", quads.nodes.size(), quads.deepest_pending,
                fnv(quads.nodes.data(), quads.nodes.size() * sizeof(ptd::QuadNode32Host)), fnv(quads.exact.data(), quads.exact.size() * sizeof(ptd::QuadBoxes64Host)));
    pt_host_scene_destroy(host);
    return 0;
}
