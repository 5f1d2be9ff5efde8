"""The CPU restatement (oracle/pt_oracle.cpp) against the reference.

`golden` holds outputs of the reference's own code (oracle/gen_golden.py); where the
reference build oracle/_ref/libptref.so is present the two are also compared live on
fresh inputs.  CPU only.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import BENCHMARK_ROWS_1080P, MODELS, id_hash, same_bits, sha256_of
from pathtracer_amd.ctypes_abi import PtCamera, PtFrame

PIECES = ["pawn", "rook", "knight", "bishop", "queen", "king"]


def frame_from(golden, key):
    return PtFrame.from_buffer_copy(golden[key].tobytes())


def camera_from(golden, key):
    return PtCamera.from_buffer_copy(golden[key].tobytes())


def test_hashes_and_xorshift(orc, golden):
    assert [orc.noise_1d(int(x)) for x in golden["noise1d_in"]] == golden["noise1d_out"].tolist()
    assert [orc.noise_3d(*map(int, r)) for r in golden["noise3d_in"]] == golden["noise3d_out"].tolist()
    nxt = [orc.random_number(int(s)) for s in golden["xorshift_in"]]
    assert nxt == golden["xorshift_out"].tolist()
    assert same_bits([orc.real_from_rng(s) for s in nxt], golden["xorshift_real"])
    # column >= 11 overflows the reference's int arithmetic (src/rng.cpp:25); it must wrap
    assert orc.noise_3d(0, 11, 0) == golden["noise3d_out"][list(map(tuple, golden["noise3d_in"])).index((0, 11, 0))]


def test_random_unit_vector(orc, golden):
    assert same_bits(orc.random_unit_vector(golden["ruv_in"]), golden["ruv_out"])


def test_ray_primitives(orc, golden):
    assert same_bits(orc.ray_aabb(golden["aabb_rays"], golden["aabb_boxes"]), golden["aabb_out"])
    assert np.isnan(golden["aabb_out"]).any(), "the fixture is meant to cover 0 * inf slab products"
    assert same_bits(orc.ray_sphere(golden["aabb_rays"], golden["sphere_in"]), golden["sphere_out"])
    out = orc.ray_triangle(golden["tri_rays"], golden["tri_in"])
    assert same_bits(out, golden["tri_out"])
    assert 0.3 < golden["tri_out"][:, 0].mean() < 0.9, "fixture should hold many hits and many misses"
    assert same_bits(orc.unit_normal(golden["tri_in"][:-16]), golden["unit_normal_out"])


def test_scatter_all_materials(orc, golden):
    out = orc.scatter(golden["scatter_rays"], golden["scatter_materials"], golden["scatter_points"], golden["scatter_normals"])
    assert same_bits(out, golden["scatter_out"])
    kinds = golden["scatter_materials"][:, 4].astype(int)
    assert set(kinds.tolist()) == {0, 1, 2, 3}
    assert (golden["scatter_out"][kinds == 3, 0] == 0).all()      # lights never scatter
    assert (golden["scatter_out"][kinds == 0, 0] == 1).all()      # lambertian always does
    valid_metal = golden["scatter_out"][kinds == 1, 0]
    assert 0 < valid_metal.mean() < 1                              # metal: both outcomes present


def test_bvh_build_matches_reference(orc, golden):
    for name in PIECES:
        triangles = np.fromfile(os.path.join(MODELS, name + ".triangles")).reshape(-1, 9)
        nodes = orc.construct_triangle_bvh(triangles)
        assert len(nodes) == int(golden[f"bvh_count_{name}"][0]) == 2 * len(triangles) - 1
        assert hashlib.sha256(nodes.tobytes()).digest() == golden[f"bvh_sha_{name}"].tobytes()
    nodes = orc.construct_sphere_bvh(golden["sphere_bvh_in"])
    assert nodes.tobytes() == golden["sphere_bvh_out"].tobytes()


@pytest.mark.parametrize("name,index,size", [("spheres", 0, (72, 48)), ("cornell", 1, (48, 48))])
def test_reference_builtin_scene_render(orc, golden, name, index, size):
    """The restatement's own construction of src/main.cpp:516-625 / :627-798 rendered by the
    restatement, against the reference-made accumulation buffer (metal, glass, sky, area light)."""
    scene, cam, aspect = orc.builtin_scene(index)
    assert bytes(cam) == golden[f"{name}_camera"].tobytes()
    frame = orc.camera_frame(cam, aspect, *size)
    tri, sph = scene.primary_hits(frame)
    assert np.array_equal(tri, golden[f"{name}_tri_hits"]) and np.array_equal(sph, golden[f"{name}_sphere_hits"])
    frame.sample_count = 4
    assert same_bits(scene.render(frame), golden[f"{name}_accum4"])


def test_division_free_real_from_rng_is_exact_for_every_input(orc):
    """The GPU maps a 32-bit hash to [0, 1] with a multiply and a fused multiply-add instead of the
    reference's division by UINT_MAX (src/rng.cpp:30-32); all 2^32 inputs must give the same double."""
    import ctypes as C
    check = orc.dll.orc_check_fused_unit_real
    check.restype = C.c_ulonglong
    check.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_int]
    assert check(0, 1 << 32, 8) == 0


def test_plumbing_configuration_in_full(orc, golden):
    """Single pawn model in the reference's scene, 640x360, 16 spp, depth 50: the CPU restatement
    against the digest of the buffer the reference's own code produced."""
    import hashlib
    scene, cam, _ = orc.model_scene(os.path.join(MODELS, "pawn.triangles"))
    frame = orc.camera_frame(cam, 640 / 360, 640, 360)
    frame.sample_count = 16
    accum = scene.render(frame, threads=8)
    assert hashlib.sha256(accum.tobytes()).digest() == golden["pawn640_accum16_sha"].tobytes()
    assert (accum.reshape(-1, 4)[:, 3] == 16.0).all()


@pytest.mark.parametrize("name,size", [("rook", (64, 48)), ("pawn", (64, 36))])
def test_model_scene_render(orc, golden, name, size):
    scene, cam, aspect = orc.model_scene(os.path.join(MODELS, name + ".triangles"))
    assert bytes(cam) == golden[f"{name}_camera"].tobytes()
    w, h = size
    frame = orc.camera_frame(cam, w / h if name == "pawn" else aspect, w, h)
    assert bytes(frame) == golden[f"{name}_frame"].tobytes()
    assert same_bits(orc.primary_rays(frame, 0), golden[f"{name}_rays"])
    tri, sph = scene.primary_hits(frame)
    assert np.array_equal(tri, golden[f"{name}_tri_hits"]) and np.array_equal(sph, golden[f"{name}_sphere_hits"])
    frame.sample_count = 4
    accum = scene.render(frame)
    assert same_bits(accum, golden[f"{name}_accum4"])
    assert np.array_equal(orc.tonemap_bgra8(accum, 4), golden[f"{name}_bgra8"])
    assert same_bits(scene.sample_colours(frame, 2), golden[f"{name}_sample2_rgb"])
    # threads change who renders a row, never the result
    assert same_bits(scene.render(frame, threads=3), accum)


def test_primary_hit_digests(orc, golden):
    scene, cam, aspect = orc.model_scene(os.path.join(MODELS, "rook.triangles"))
    tri, sph = scene.primary_hits(orc.camera_frame(cam, aspect, 600, 600))
    assert [(tri >= 0).sum(), id_hash(tri), (sph >= 0).sum()] == golden["rook600_hits"].tolist()
    assert golden["rook600_hits"].tolist()[:2] == [12746, 27538840616433]
    scene, cam, _ = orc.model_scene(os.path.join(MODELS, "pawn.triangles"))
    tri, sph = scene.primary_hits(orc.camera_frame(cam, 640 / 360, 640, 360))
    assert [(tri >= 0).sum(), id_hash(tri), (sph >= 0).sum()] == golden["pawn640_hits"].tolist()


def test_oracle_vs_live_reference(orc, ref):
    """Fresh inputs through both builds (skipped where the reference build is absent)."""
    rng = np.random.default_rng(7)
    d = rng.normal(size=(512, 3))
    d /= np.sqrt((d * d).sum(1))[:, None]
    assert same_bits(orc.random_unit_vector(d), ref.random_unit_vector(d))
    rays = np.concatenate([rng.uniform(-30, 30, (512, 3)), d], 1)
    lo = rng.uniform(-40, 30, (512, 3))
    boxes = np.concatenate([lo, lo + rng.uniform(0, 30, (512, 3))], 1)
    assert same_bits(orc.ray_aabb(rays, boxes), ref.ray_aabb(rays, boxes))
    scene_o, cam_o, _ = orc.model_scene(os.path.join(MODELS, "knight.triangles"))
    scene_r, cam_r, aspect = ref.model_scene(os.path.join(MODELS, "knight.triangles"))
    assert bytes(cam_o) == bytes(cam_r)
    eo, er = scene_o.export(), scene_r.export()
    for key in er:
        assert eo[key].tobytes() == er[key].tobytes(), key
    frame = ref.camera_frame(cam_r, aspect, 96, 96)
    frame.sample_begin, frame.sample_count = 3, 2
    assert same_bits(scene_o.render(frame), scene_r.render(frame))
    index_o, dist_o = scene_o.trace_triangles(rays)
    index_r, dist_r = scene_r.trace_triangles(rays)
    assert np.array_equal(index_o, index_r) and same_bits(dist_o, dist_r)


@pytest.mark.parametrize("name, param, spp", [("chess32", 0, 4), ("mixed", 0, 4), ("grid2", 2, 2)])
def test_benchmark_scenes_against_the_reference(orc, api, benchmark_golden, name, param, spp):
    """The scenes the benchmark renders - the 32-piece chess set, the same with every material, four boards of it - as the
    reference's own code rendered them (trees, traversal and bounce loop the reference's; fixtures written by
    oracle/gen_benchmark_golden.py): first-vertex indices and every double of a 160 x 90 frame, and three rows of the
    1920 x 1080 frame at 16 samples per pixel by digest."""
    from pathtracer_amd import ctypes_abi as abi
    g = benchmark_golden
    kind = {"chess32": abi.PT_SCENE_CHESS32, "mixed": abi.PT_SCENE_CHESS32_MIXED, "grid2": abi.PT_SCENE_CHESS_GRID}[name]
    host = api.HostScene(kind, MODELS, param)
    arrays = host.arrays()
    assert np.array_equal(sha256_of(arrays["triangles"]), g[f"{name}_triangles_sha"])           # the same inputs
    assert same_bits(np.frombuffer(bytes(host.camera), np.float64), g[f"{name}_camera"])
    scene = orc.scene_create(arrays)
    frame = host.frame(160, 90, sample_count=spp, max_bounces=50)
    tri, sphere = scene.primary_hits(frame)
    assert np.array_equal(tri, g[f"{name}_tri_hits"]) and np.array_equal(sphere, g[f"{name}_sphere_hits"])
    assert same_bits(scene.render(frame, threads=8), g[f"{name}_accum"])
    if name != "grid2":
        for row in BENCHMARK_ROWS_1080P:
            big = host.frame(1920, 1080, sample_count=16, max_bounces=50, row_begin=row, row_end=row + 1)
            got = scene.render(big, threads=8).reshape(1080, 1920, 4)[row]
            assert np.array_equal(sha256_of(got), g[f"{name}_1080p_row{row}_sha"]), row
