"""The HIP path against the reference, through the C ABI, on a real MI355X.

Bar: bit-exact.  Every comparison below is on the raw binary64 bits (any NaN equals any
NaN) or on integer indices; no tolerance is used anywhere.  Checkers: the committed vectors
produced by the reference's own code (`golden`), the CPU restatement (`orc`), and, where it
has been built, the reference build itself (`ref`).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import BENCHMARK_ROWS_1080P, MODELS, id_hash, same_bits, sha256_of
from pathtracer_amd import ctypes_abi as abi
from pathtracer_amd.ctypes_abi import PtFrame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scenes(api, orc):
    """(host arrays, device scene, checker scene) per scene name, built once."""
    cache = {}

    def get(name):
        if name not in cache:
            if name in ("rook", "pawn", "knight"):
                host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, name + ".triangles"))
            elif name in ("spheres", "cornell"):
                host = api.HostScene({"spheres": abi.PT_SCENE_RANDOM_SPHERES, "cornell": abi.PT_SCENE_CORNELL_BOX}[name])
            elif name.startswith("grid"):
                host = api.HostScene(abi.PT_SCENE_CHESS_GRID, MODELS, int(name[4:]))
            else:
                kind = {"chess6": abi.PT_SCENE_CHESS6_ROW, "chess32": abi.PT_SCENE_CHESS32, "mixed": abi.PT_SCENE_CHESS32_MIXED}[name]
                host = api.HostScene(kind, MODELS)
            cache[name] = (host, api.DeviceScene.from_host(host), orc.scene_create(host.arrays()))
        return cache[name]

    return get


# ------------------------------------------------------------------ device arithmetic, one block at a time
def test_device_hashes(api, golden):
    out = api.device_eval(abi.PT_EVAL_NOISE_1D, golden["noise1d_in"])
    assert out[:, 0].tolist() == golden["noise1d_out"].tolist()
    out = api.device_eval(abi.PT_EVAL_NOISE_3D, golden["noise3d_in"])
    assert out[:, 0].tolist() == golden["noise3d_out"].tolist()
    out = api.device_eval(abi.PT_EVAL_XORSHIFT_REAL, golden["xorshift_in"])
    assert out[:, 0].view(np.uint64).astype(np.uint32).tolist() == golden["xorshift_out"].tolist()
    assert same_bits(out[:, 1], golden["xorshift_real"])


def test_device_real_from_rng_matches_the_division(api):
    """The device's division-free real_from_rng on 4 M hashes (all bit patterns near the ends, powers
    of two and their neighbours, a stride through the rest) against r / UINT_MAX computed here."""
    seeds = np.concatenate([np.arange(0, 1 << 16), np.arange((1 << 32) - (1 << 16), 1 << 32),
                            np.arange(0, 1 << 32, 1109)]).astype(np.uint32)
    out = api.device_eval(abi.PT_EVAL_XORSHIFT_REAL, seeds)
    r = out[:, 0].view(np.uint64).astype(np.uint32)
    assert same_bits(out[:, 1], r.astype(np.float64) / 4294967295.0)


def test_device_sqrt_and_divide_are_correctly_rounded(api):
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(0, 4, 200000), 10.0 ** rng.uniform(-300, 300, 200000), [0.0, 1.0, 2.0, 4.0, 1e-310, np.inf]])
    b = np.concatenate([rng.uniform(-3, 3, 200000), 10.0 ** rng.uniform(-150, 150, 200000) * rng.choice([-1, 1], 200000), [3.0, 0.0, -0.0, 7.0, 3.0, 2.0]])
    out = api.device_eval(abi.PT_EVAL_SQRT_DIV, np.stack([a, b], 1))
    with np.errstate(all="ignore"):
        assert same_bits(out[:, 0], np.sqrt(a))
        assert same_bits(out[:, 1], a / b)


def test_device_random_unit_vector(api, golden):
    assert same_bits(api.device_eval(abi.PT_EVAL_RANDOM_UNIT_VECTOR, golden["ruv_in"]), golden["ruv_out"])


def test_device_ray_primitives(api, golden):
    out = api.device_eval(abi.PT_EVAL_RAY_AABB, np.concatenate([golden["aabb_rays"], golden["aabb_boxes"]], 1))
    assert same_bits(out, golden["aabb_out"])           # also proves fmin/fmax form == exact form on finite reciprocals
    out = api.device_eval(abi.PT_EVAL_RAY_SPHERE, np.concatenate([golden["aabb_rays"], golden["sphere_in"]], 1))
    assert same_bits(out, golden["sphere_out"])
    out = api.device_eval(abi.PT_EVAL_RAY_TRIANGLE, np.concatenate([golden["tri_rays"], golden["tri_in"]], 1))
    assert same_bits(out, golden["tri_out"])
    out = api.device_eval(abi.PT_EVAL_UNIT_NORMAL, golden["tri_in"][:-16])
    assert same_bits(out, golden["unit_normal_out"])


def test_device_triangle_test_on_edges_and_at_odd_scales(api, orc):
    """The wavefront kernel's lanes test triangles together and skip the two barycentric divisions wherever their outcome is
    certain (hit_triangle_together; the device compares it with the plain form on every row and answers NaN where they
    differ).  Rays aimed at points on the edges and vertices of lattice triangles, where alpha + beta is 1 or a numerator
    is 0 up to the last bit, and the same at scales where n.n leaves the range in which the shortcut is allowed.  Against
    the checker's intersect(ray, triangle) (src/geometry.cpp:106-133), bit for bit."""
    rng = np.random.default_rng(2024)
    count = 60000
    a = rng.integers(-8, 9, (count, 3)).astype(float) * 2
    b = a + rng.integers(-6, 7, (count, 3)) * 2
    c = a + rng.integers(-6, 7, (count, 3)) * 2
    # barycentric weights with many exact zeros, halves and sums of exactly one
    w = rng.choice([0.0, 0.0, 0.25, 0.5, 0.5, 0.75, 1.0, 0.3, 0.7, 1e-17, 1.0 - 2.0 ** -52], (count, 2))
    w[::3, 1] = 1.0 - w[::3, 0]                                     # on the edge opposite a
    target = a + w[:, :1] * (b - a) + w[:, 1:] * (c - a)
    origin = rng.integers(-20, 21, (count, 3)).astype(float)
    origin[::5] = target[::5] + np.eye(3)[rng.integers(0, 3, len(origin[::5]))] * rng.choice([-7.0, 9.0], (len(origin[::5]), 1))   # axis-parallel
    direction = target - origin
    length = np.sqrt((direction ** 2).sum(1, keepdims=True))
    direction = np.where(length > 0, direction / np.where(length > 0, length, 1.0), np.array([[0.0, 0.0, 1.0]]))
    rays = np.concatenate([origin, direction], 1)
    tris = np.concatenate([a, b, c], 1)
    blocks_r, blocks_t = [rays], [tris]
    for scale in (1e-30, 1e-38, 1e-45, 1e-80, 1e30, 1e38, 1e45, 1e80, 2.0 ** -20, 3.0):
        part = slice(0, 4000)
        blocks_r.append(np.concatenate([origin[part] * scale, direction[part]], 1))
        blocks_t.append(tris[part] * scale)
    rays, tris = np.concatenate(blocks_r), np.concatenate(blocks_t)
    with np.errstate(all="ignore"):
        out = api.device_eval(abi.PT_EVAL_RAY_TRIANGLE, np.concatenate([rays, tris], 1))
        expect = orc.ray_triangle(rays, tris)
    assert same_bits(out, expect)
    assert 0.2 < (expect[:count, 0] == 1.0).mean() < 0.95          # the rows do hit and miss


def test_device_scatter(api, golden):
    records = np.concatenate([golden["scatter_rays"], golden["scatter_materials"], golden["scatter_points"], golden["scatter_normals"]], 1)
    assert same_bits(api.device_eval(abi.PT_EVAL_SCATTER, records), golden["scatter_out"])


def test_device_tonemap(api, golden):
    """src/main.cpp:1024-1034 on the GPU against the host entry point and the reference-made image,
    including sums a render cannot produce (negative, NaN, infinite, far above the sample count)."""
    import torch
    rng = np.random.default_rng(5)
    odd = np.array([0.0, -0.0, 4.0, 3.999999999999999, 4.000000000000001, -1.0, np.nan, np.inf, 1e300, 1e-320, 2.25, 0.25])
    accum = np.concatenate([golden["rook_accum4"].ravel(), rng.uniform(0, 6, 4 * 5000), np.resize(odd, 4 * 300)])
    expect = api.tonemap_bgra8(accum, 4)
    assert np.array_equal(expect[:golden["rook_bgra8"].size], golden["rook_bgra8"].ravel())
    a = torch.from_numpy(accum).to("cuda:0")
    out = torch.zeros(accum.size, dtype=torch.uint8, device="cuda:0")
    api.tonemap_device(0, a.data_ptr(), accum.size // 4, 4, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), expect)
    with pytest.raises(api.PathTracerError):
        api.tonemap_device(0, a.data_ptr(), accum.size // 4, 0, out.data_ptr())


# ------------------------------------------------------------------ primary hits: bit-exact triangle ids
@pytest.mark.parametrize("name,size,key", [("rook", (64, 48), "rook"), ("pawn", (64, 36), "pawn")])
def test_primary_hits_small(scenes, golden, name, size, key):
    host, device, _ = scenes(name)
    frame = PtFrame.from_buffer_copy(golden[f"{key}_frame"].tobytes())
    tri, sph = device.primary_hits(frame)
    assert np.array_equal(tri, golden[f"{key}_tri_hits"])
    assert np.array_equal(sph, golden[f"{key}_sphere_hits"])


def test_primary_hit_digests(scenes, golden):
    host, device, _ = scenes("rook")
    tri, sph = device.primary_hits(host.frame(600, 600))
    assert [(tri >= 0).sum(), id_hash(tri), (sph >= 0).sum()] == golden["rook600_hits"].tolist()
    host, device, _ = scenes("pawn")
    # the fixture frame uses aspect 640/360 with the model scene's camera (whose own aspect is 1)
    from pathtracer_amd import api as papi
    frame = papi.camera_frame(host.camera, 640 / 360, 640, 360)
    tri, sph = device.primary_hits(frame)
    assert [(tri >= 0).sum(), id_hash(tri), (sph >= 0).sum()] == golden["pawn640_hits"].tolist()
    host, device, _ = scenes("chess6")
    tri, sph = device.primary_hits(host.frame(640, 360))
    assert [(tri >= 0).sum(), id_hash(tri), (sph >= 0).sum()] == golden["chess6_640_hits"].tolist()
    tri, sph = device.primary_hits(host.frame(64, 36))
    assert np.array_equal(tri, golden["chess6_tri_hits"]) and np.array_equal(sph, golden["chess6_sphere_hits"])


def test_primary_hits_chess32_vs_checker(scenes):
    host, device, checker = scenes("chess32")
    frame = host.frame(320, 180)
    tri, sph = device.primary_hits(frame)
    tri_c, sph_c = checker.primary_hits(frame)
    assert np.array_equal(tri, tri_c) and np.array_equal(sph, sph_c)
    assert (tri >= 0).mean() > 0.5


# ------------------------------------------------------------------ whole paths: accumulation buffers
@pytest.mark.parametrize("name,key", [("rook", "rook"), ("pawn", "pawn")])
def test_accumulation_buffer_matches_reference(scenes, golden, name, key):
    host, device, _ = scenes(name)
    frame = PtFrame.from_buffer_copy(golden[f"{key}_frame"].tobytes())
    frame.sample_count = 4
    accum = device.render(frame)
    assert same_bits(accum, golden[f"{key}_accum4"])


def test_plumbing_configuration_in_full(scenes, golden):
    """Single pawn model, 640x360, 16 spp, depth 50: digest of the buffer the reference's own code made."""
    import hashlib
    from pathtracer_amd import api as papi
    host, device, _ = scenes("pawn")
    frame = papi.camera_frame(host.camera, 640 / 360, 640, 360, sample_count=16)
    accum = device.render(frame)
    assert hashlib.sha256(np.ascontiguousarray(accum).tobytes()).digest() == golden["pawn640_accum16_sha"].tobytes()


def test_accumulation_buffer_chess6(scenes, golden):
    host, device, _ = scenes("chess6")
    accum = device.render(host.frame(64, 36, sample_count=4))
    assert same_bits(accum, golden["chess6_accum4"])


@pytest.mark.parametrize("name,size", [("spheres", (72, 48)), ("cornell", (48, 48))])
def test_reference_builtin_scenes(scenes, golden, name, size):
    """The reference's random-spheres and Cornell-box scenes (src/main.cpp:516-625, 627-798):
    488 spheres in a sphere BVH with no triangles at all; triangles, an area light and a glass sphere."""
    host, device, _ = scenes(name)
    frame = host.frame(*size)
    tri, sph = device.primary_hits(frame)
    assert np.array_equal(tri, golden[f"{name}_tri_hits"]) and np.array_equal(sph, golden[f"{name}_sphere_hits"])
    frame.sample_count = 4
    assert same_bits(device.render(frame), golden[f"{name}_accum4"])


@pytest.mark.parametrize("name,size,spp,bounces", [
    ("spheres", (150, 100), 3, 50),    # depth-of-field camera, hundreds of spheres, sky background
    ("cornell", (100, 100), 6, 50),
    ("cornell", (41, 41), 2, 5),
    ("grid2", (128, 72), 2, 50),       # four boards, 1.69 M triangles
    ("chess32", (160, 90), 4, 50),
    ("chess32", (96, 54), 3, 16),
    ("mixed", (160, 90), 4, 16),       # metal, glass, sky: every scatter branch and long paths
    ("mixed", (64, 36), 2, 50),
    ("knight", (97, 61), 2, 50),       # sizes that are not multiples of the 8x8 tile
    ("rook", (33, 17), 1, 1),
    ("rook", (33, 17), 2, 0),          # no bounce at all: zero colour, counts only
])
def test_accumulation_buffer_matches_checker(scenes, name, size, spp, bounces):
    host, device, checker = scenes(name)
    frame = host.frame(*size, sample_count=spp, max_bounces=bounces)
    assert same_bits(device.render(frame), checker.render(frame, threads=8))


def test_render_adds_into_the_buffer_and_resumes(scenes):
    """Progressive accumulation: samples [0,3) then [3,5) on the same buffer == [0,5) at once,
    and the count channel holds the number of samples (src/main.cpp:117-121)."""
    host, device, checker = scenes("chess6")
    whole = device.render(host.frame(80, 45, sample_count=5))
    parts = device.render(host.frame(80, 45, sample_count=3))
    device.render(host.frame(80, 45, sample_begin=3, sample_count=2), parts)
    assert same_bits(whole, parts)
    assert (whole.reshape(-1, 4)[:, 3] == 5.0).all()
    assert same_bits(whole, checker.render(host.frame(80, 45, sample_count=5), threads=8))


def test_row_ranges_tile_the_image(scenes):
    """Rows are independent (the reference's work queue hands them out one at a time): any
    split into row ranges gives the bytes of the full render, and rows outside are untouched."""
    host, device, _ = scenes("chess6")
    full = device.render(host.frame(72, 41, sample_count=2))
    pieces = np.zeros_like(full)
    for begin, end in [(0, 5), (5, 6), (6, 30), (30, 41)]:
        device.render(host.frame(72, 41, sample_count=2, row_begin=begin, row_end=end), pieces)
    assert same_bits(full, pieces)
    only = device.render(host.frame(72, 41, sample_count=2, row_begin=9, row_end=10))
    rows = only.reshape(41, 72, 4)
    assert (rows[:9] == 0).all() and (rows[10:] == 0).all() and same_bits(rows[9], full.reshape(41, 72, 4)[9])
    # empty ranges are legal no-ops
    assert (device.render(host.frame(72, 41, row_begin=7, row_end=7)) == 0).all()
    assert (device.render(host.frame(72, 41, sample_count=0)) == 0).all()


def test_full_size_properties(scenes, api):
    """1920x1080 chess set (the benchmark configuration) through properties that need no CPU
    render: sample-range additivity, the count channel, finiteness, and a CPU spot check of
    a few rows."""
    import torch
    host, device, checker = scenes("chess32")
    w, h = 1920, 1080
    buf = torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0")
    device.render_device(host.frame(w, h, sample_count=2), buf.data_ptr())
    device.render_device(host.frame(w, h, sample_begin=2, sample_count=1), buf.data_ptr())
    torch.cuda.synchronize()
    once = torch.zeros_like(buf)
    device.render_device(host.frame(w, h, sample_count=3), once.data_ptr())
    torch.cuda.synchronize()
    assert device.last_render_ms() > 0
    assert torch.equal(buf.view(torch.int64), once.view(torch.int64))
    image = once.cpu().numpy().reshape(h, w, 4)
    assert (image[..., 3] == 3.0).all() and np.isfinite(image).all() and (image[..., :3] >= 0).all()
    rows = (0, 397, 1079)
    for row in rows:
        expect = checker.render(host.frame(w, h, sample_count=3, row_begin=row, row_end=row + 1), threads=8)
        assert same_bits(image[row], expect.reshape(h, w, 4)[row])


def test_full_frame_matches_checker(scenes):
    """The whole 1920x1080 frame of the benchmark scene, 16 samples per pixel (33 M paths: one whole sample group, the kernel's
    steady state with every wave busy), against the CPU checker: every double; and, where the reference's own build has
    travelled with the tree (oracle/_ref), against the reference itself."""
    from oracle.bindings import RefLib, have_ref
    host, device, checker = scenes("chess32")
    frame = host.frame(1920, 1080, sample_count=16, max_bounces=50)
    got = device.render(frame)
    assert same_bits(got, checker.render(frame, threads=16))
    if have_ref():
        assert same_bits(got, RefLib().scene_create(host.arrays()).render(frame, threads=16))


@pytest.mark.parametrize("name, spp", [("chess32", 4), ("mixed", 4), ("grid2", 2)])
def test_benchmark_scenes_against_the_reference(scenes, benchmark_golden, name, spp):
    """The benchmark's scenes as the reference's own code rendered them (tests/golden/benchmark_scenes.npz): first-vertex
    indices and every double of a 160 x 90 frame at the reference's depth, three rows of the 1920 x 1080 frame at 16 samples
    per pixel by digest."""
    g = benchmark_golden
    host, device, _ = scenes(name)
    assert np.array_equal(sha256_of(host.arrays()["triangles"]), g[f"{name}_triangles_sha"])
    frame = host.frame(160, 90, sample_count=spp, max_bounces=50)
    tri, sphere = device.primary_hits(frame)
    assert np.array_equal(tri, g[f"{name}_tri_hits"]) and np.array_equal(sphere, g[f"{name}_sphere_hits"])
    assert same_bits(device.render(frame), g[f"{name}_accum"])
    if name != "grid2":
        for row in BENCHMARK_ROWS_1080P:
            big = host.frame(1920, 1080, sample_count=16, max_bounces=50, row_begin=row, row_end=row + 1)
            got = device.render(big).reshape(1080, 1920, 4)[row]
            assert np.array_equal(sha256_of(got), g[f"{name}_1080p_row{row}_sha"]), row


@pytest.mark.parametrize("name, size, spp, bounces", [
    ("spheres", (1200, 800), 4, 50),      # the reference's own scene at its own size: spheres and no triangle
    ("cornell", (512, 512), 4, 50),       # long paths, shared diagonals (ties)
    ("mixed", (960, 540), 4, 16),         # every material
])
def test_large_frames_of_the_other_scenes_match_checker(scenes, name, size, spp, bounces):
    host, device, checker = scenes(name)
    frame = host.frame(*size, sample_count=spp, max_bounces=bounces)
    assert same_bits(device.render(frame), checker.render(frame, threads=16))


def test_ten_million_triangles_match_checker(scenes):
    """25 boards, 10.56 M triangles: 1.35 GB of triangle records and 5 M four-wide nodes, so byte offsets
    pass 2^31 and the scene no longer fits any cache."""
    host, device, checker = scenes("grid5")
    assert host.desc.triangle_count == 10564950
    frame = host.frame(160, 90, sample_count=2, max_bounces=50)
    tri, sph = device.primary_hits(frame)
    tri_c, sph_c = checker.primary_hits(frame)
    assert np.array_equal(tri, tri_c) and np.array_equal(sph, sph_c) and tri.max() > 2 ** 23
    assert same_bits(device.render(frame), checker.render(frame, threads=16))


def test_live_reference_build(scenes, ref, api):
    """Where the reference's own build is present: its render of its own scene, bit for bit."""
    host, device, _ = scenes("rook")
    scene, cam, aspect = ref.model_scene(os.path.join(MODELS, "rook.triangles"))
    frame = host.frame(128, 128, sample_count=3)
    assert same_bits(device.render(frame), scene.render(frame, threads=8))
    tri, sph = device.primary_hits(frame)
    tri_r, sph_r = scene.primary_hits(frame)
    assert np.array_equal(tri, tri_r) and np.array_equal(sph, sph_r)


# ------------------------------------------------------------------ edge cases the traversal has special paths for
def test_tied_leaves_report_a_miss_like_the_reference(api, orc):
    """Two identical triangles tie at every hit; the reference's tournament then reports a
    miss at their common ancestor (src/path_tracing.cpp:161-167), and rays see what lies behind."""
    front = [-10.0, -10.0, 0.0, 10.0, -10.0, 0.0, 0.0, 10.0, 0.0]
    back = [-30.0, -30.0, -5.0, 30.0, -30.0, -5.0, 0.0, 30.0, -5.0]
    far = [-60.0, -60.0, -9.0, 60.0, -60.0, -9.0, 0.0, 60.0, -9.0]
    triangles = np.array([front, back, front, far])
    materials = np.zeros(2, abi.MATERIAL_DTYPE)
    materials[0]["v"] = [0.5, 0.6, 0.7, 0.0]
    materials[1]["v"] = [1.0, 1.0, 1.0, 4.0]
    materials[1]["type"] = abi.PT_DIFFUSE_LIGHT
    arrays = {"materials": materials, "triangles": triangles, "triangle_material_indices": np.array([0, 1, 0, 0], np.int32),
              "spheres": np.zeros((0, 4)), "sphere_material_indices": np.zeros(0, np.int32),
              "background": np.array([[0.2, 0.3, 0.4], [0.9, 0.8, 0.7]])}
    arrays["triangle_bvh"] = api.construct_triangle_bvh(triangles)
    from oracle.bindings import make_desc
    desc, keep = make_desc(arrays)
    device = api.DeviceScene(desc, keepalive=keep)
    checker = orc.scene_create(arrays)
    cam = api.camera_look_at((0.0, 0.0, 40.0), (0.0, 0.0, 0.0), 0.8, 0.0)
    frame = api.camera_frame(cam, 1.0, 48, 48, sample_count=2)
    tri, _ = device.primary_hits(frame)
    tri_c, _ = checker.primary_hits(frame)
    assert np.array_equal(tri, tri_c)
    assert not np.isin(tri, [0, 2]).any() and (tri == 1).any()      # the duplicated triangle is never reported
    assert same_bits(device.render(frame), checker.render(frame))


def test_axis_parallel_rays_take_the_exact_path(api, orc):
    """aperture 0 and a camera on an axis give primary rays with exactly zero direction
    components: infinite reciprocals, 0 * inf in the slab test, exact min/max order matters."""
    rng = np.random.default_rng(5)
    base = rng.integers(-8, 8, (400, 1, 3)).astype(float)
    triangles = (base + rng.integers(0, 3, (400, 3, 3))).reshape(-1, 9)
    materials = np.zeros(1, abi.MATERIAL_DTYPE)
    materials[0]["v"] = [0.7, 0.7, 0.7, 0.0]
    arrays = {"materials": materials, "triangles": triangles, "triangle_material_indices": np.zeros(len(triangles), np.int32),
              "spheres": np.array([[0.0, 0.0, 0.0, 2.0]]), "sphere_material_indices": np.zeros(1, np.int32),
              "background": np.array([[1.0, 1.0, 1.0], [0.5, 0.7, 1.0]])}
    arrays["triangle_bvh"] = api.construct_triangle_bvh(triangles)
    arrays["sphere_bvh"] = api.construct_sphere_bvh(arrays["spheres"])
    from oracle.bindings import make_desc
    desc, keep = make_desc(arrays)
    device = api.DeviceScene(desc, keepalive=keep)
    checker = orc.scene_create(arrays)
    frame = api.camera_frame(api.camera_look_at((0.0, 0.0, 30.0), (0.0, 0.0, 0.0), 0.9, 0.0), 1.0, 33, 33, sample_count=2)
    # a fan of rays in the plane x = 3 of the integer lattice the triangles live on: the x
    # component of every direction is exactly zero and (box.x - origin.x) is often exactly zero
    frame.camera_x.x, frame.camera_x.y, frame.camera_x.z = 1.0, 0.0, 0.0
    frame.camera_y.x, frame.camera_y.y, frame.camera_y.z = 0.0, 1.0, 0.0
    frame.step_x.x, frame.step_x.y, frame.step_x.z = 0.0, 0.0, 0.0
    frame.step_y.x, frame.step_y.y, frame.step_y.z = 0.0, 32.0, 0.0
    frame.bottom_left.x, frame.bottom_left.y, frame.bottom_left.z = 3.0, -14.0, 0.0
    frame.camera_position.x, frame.camera_position.y, frame.camera_position.z = 3.0, 2.0, 30.0
    tri, sph = device.primary_hits(frame)
    tri_c, sph_c = checker.primary_hits(frame)
    assert np.array_equal(tri, tri_c) and np.array_equal(sph, sph_c)
    assert same_bits(device.render(frame), checker.render(frame))


def test_scene_create_validates_trees(api):
    host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, "rook.triangles"))
    arrays = host.arrays()
    from oracle.bindings import make_desc
    nodes = arrays["triangle_bvh"].copy()
    nodes[5]["left"] = 10 ** 6
    desc, keep = make_desc({**arrays, "triangle_bvh": nodes})
    with pytest.raises(api.PathTracerError) as e:
        api.DeviceScene(desc, keepalive=keep)
    assert e.value.code == abi.PT_ERR_INVALID_ARGUMENT
    desc, keep = make_desc({**arrays, "triangle_bvh": arrays["triangle_bvh"][:-2]})
    with pytest.raises(api.PathTracerError):
        api.DeviceScene(desc, keepalive=keep)
    with pytest.raises(api.PathTracerError):
        api.DeviceScene.from_host(host, device=99)
