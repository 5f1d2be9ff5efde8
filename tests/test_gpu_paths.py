"""Every code path of the GPU library must give the same bits.

The library picks between two kernels (persistent wavefront / one lane per pixel), two inner-node
precisions (conservative f32 / exact f64) and two search hierarchies (its own SAH rebuild / the
reference's tree shape).  Each combination is rendered here and compared, bit for bit, with the
CPU checker; environment switches (read by the library at every call) select the path.
"""
import os

import numpy as np
import pytest

from conftest import MODELS, id_hash, same_bits
from pathtracer_amd import ctypes_abi as abi

pytestmark = pytest.mark.gpu

SWITCHES = ("PT_KERNEL", "PT_NODES", "PT_BVH", "PT_STATS", "PT_WORKSPACE_GB", "PT_SHADE_THRESHOLD", "PT_LEAF_THRESHOLD",
            "PT_STACK_SLOTS", "PT_DISTRUST_MASK", "PT_PARKING")


@pytest.fixture
def switches():
    saved = {k: os.environ.pop(k, None) for k in SWITCHES}

    def use(**values):
        for k in SWITCHES:
            os.environ.pop(k, None)
        for k, v in values.items():
            os.environ[k] = str(v)

    yield use
    for k in SWITCHES:
        os.environ.pop(k, None)
    for k, v in saved.items():
        if v is not None:
            os.environ[k] = v


@pytest.fixture(scope="module")
def chess(api, orc):
    host = api.HostScene(abi.PT_SCENE_CHESS32_MIXED, MODELS)
    frame = host.frame(128, 72, sample_count=5, max_bounces=16)
    expect = orc.scene_create(host.arrays()).render(frame, threads=8)
    return host, frame, expect


@pytest.mark.parametrize("bvh", ["sah", "reference"])
@pytest.mark.parametrize("kernel,nodes", [("wave", "f32"), ("wave", "f64"), ("pixels", "f64")])
def test_every_path_matches_the_checker(api, chess, switches, kernel, nodes, bvh):
    host, frame, expect = chess
    switches(PT_KERNEL=kernel, PT_NODES=nodes, PT_BVH=bvh)      # PT_BVH is read when the scene is created
    device = api.DeviceScene.from_host(host)
    assert same_bits(device.render(frame), expect)
    tri, sph = device.primary_hits(frame)
    switches()
    tri_default, sph_default = api.DeviceScene.from_host(host).primary_hits(frame)
    assert np.array_equal(tri, tri_default) and np.array_equal(sph, sph_default)


@pytest.mark.parametrize("bvh", ["sah", "reference"])
@pytest.mark.parametrize("kernel,nodes", [("wave", "f32"), ("wave", "f64"), ("pixels", "f64")])
def test_primary_hit_digests_on_every_search(api, golden, switches, kernel, nodes, bvh):
    """pt_primary_hits reports the first-vertex indices found by the kernel that renders (the persistent
    wavefront kernel, unless PT_KERNEL=pixels): the reference-made digests of the 600x600 rook frame and the
    640x360 six-piece frame, and the reference-made 64x36 maps, for each inner-node kind and hierarchy."""
    switches(PT_KERNEL=kernel, PT_NODES=nodes, PT_BVH=bvh)
    host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, "rook.triangles"))
    tri, sph = api.DeviceScene.from_host(host).primary_hits(host.frame(600, 600))
    assert [(tri >= 0).sum(), id_hash(tri), (sph >= 0).sum()] == golden["rook600_hits"].tolist()
    host = api.HostScene(abi.PT_SCENE_CHESS6_ROW, MODELS)
    device = api.DeviceScene.from_host(host)
    tri, sph = device.primary_hits(host.frame(640, 360))
    assert [(tri >= 0).sum(), id_hash(tri), (sph >= 0).sum()] == golden["chess6_640_hits"].tolist()
    tri, sph = device.primary_hits(host.frame(64, 36))
    assert np.array_equal(tri, golden["chess6_tri_hits"]) and np.array_equal(sph, golden["chess6_sphere_hits"])


def test_primary_hits_of_a_shard_leave_other_rows_alone(api, orc, switches):
    """Rows outside the frame's range, and rows of the range that another shard owns, keep what the caller's
    buffers held; a sample other than 0 is reported when sample_begin says so."""
    host = api.HostScene(abi.PT_SCENE_CHESS6_ROW, MODELS)
    device = api.DeviceScene.from_host(host)
    checker = orc.scene_create(host.arrays())
    frame = host.frame(90, 61, row_begin=3, row_end=50, sample_begin=7, shard_index=1, shard_count=2)
    tri, sph = device.primary_hits(frame)
    tri_c, sph_c = checker.primary_hits(host.frame(90, 61, sample_begin=7))
    rows = np.arange(61)
    mine = (rows >= 3) & (rows < 50) & (((rows - 3) // 8) % 2 == 1)
    assert np.array_equal(tri[mine], tri_c[mine]) and np.array_equal(sph[mine], sph_c[mine])
    assert (tri[~mine] == -2).all() and (sph[~mine] == -2).all()


def test_sample_batches_do_not_change_the_sum(api, chess, switches):
    """A workspace too small for the whole sample range splits it into batches; each pixel's
    samples are still added in ascending order."""
    host, frame, expect = chess
    switches(PT_WORKSPACE_GB="0.0001")       # one 16-sample group per batch
    device = api.DeviceScene.from_host(host)
    many = host.frame(64, 40, sample_count=37, max_bounces=16)
    batched = device.render(many)
    switches()
    whole = api.DeviceScene.from_host(host).render(many)
    assert same_bits(batched, whole)
    assert (whole.reshape(-1, 4)[:, 3] == 37.0).all()


@pytest.mark.parametrize("slots", [3, 6, 12])
def test_a_lane_that_outgrows_its_stack_falls_back_to_the_exhaustive_search(api, chess, switches, slots):
    """The 4-wide search keeps pending subtrees in a per-lane LDS stack that may be smaller than the
    deepest search needs; a lane about to overflow is resolved by the exhaustive traversal instead.
    With 3 slots every search overflows at once; the bytes must not change."""
    host, frame, expect = chess
    switches(PT_STACK_SLOTS=slots)
    small = host.frame(64, 36, sample_count=2, max_bounces=8)
    got = api.DeviceScene.from_host(host).render(small)
    switches()
    assert same_bits(got, api.DeviceScene.from_host(host).render(small))


@pytest.mark.parametrize("setting", [{"PT_STACK_SLOTS": 16}, {"PT_STACK_SLOTS": 17}, {"PT_STACK_SLOTS": 20}, {"PT_STACK_SLOTS": 24, "PT_STATS": 1},
                                     {"PT_STACK_SLOTS": 16, "PT_PARKING": 0}, {"PT_PARKING": 0}])
def test_a_full_stack_parks_its_oldest_entries_in_device_memory(api, chess, switches, setting):
    """From 16 slots on, a lane whose LDS stack is full does not give the ordered search up: the next shade phase moves
    the oldest 12 entries to a backing store in device memory and the lane searches on; when its stack runs empty the
    unit parked last comes back.  With 16 slots the chess set's searches do that all the time.  PT_PARKING=0 is the launch
    without a backing store (what pt_render_device_noalloc runs): up to 36 slots, then the exhaustive search."""
    host, frame, expect = chess
    switches(**setting)
    device = api.DeviceScene.from_host(host)
    got = device.render(frame)
    deep = host.frame(160, 90, sample_count=3, max_bounces=50)
    got_deep = device.render(deep)
    switches(PT_KERNEL="pixels")
    assert same_bits(got, expect)
    assert same_bits(got_deep, api.DeviceScene.from_host(host).render(deep))
    switches()


def test_a_winner_whose_box_test_fails_is_resolved_by_the_exhaustive_search(api, chess, switches):
    """The leaf phase tests triangles without their boxes; the winner's box is tested before shading
    and a failure sends the ray to the exhaustive traversal.  Real rays all but never fail it, so the
    switch makes the kernel distrust every fourth winner: the bytes must not change."""
    host, frame, expect = chess
    switches(PT_DISTRUST_MASK=3)
    small = host.frame(96, 54, sample_count=2, max_bounces=8)
    got = api.DeviceScene.from_host(host).render(small)
    switches()
    assert same_bits(got, api.DeviceScene.from_host(host).render(small))


def test_single_precision_nodes_never_skip_what_the_exact_test_accepts(api, switches):
    """Profiling build: beside every conservative f32 box test the kernel evaluates the exact f64
    test on the same node and counts boxes that f32 skipped but f64 accepts.  Must be zero."""
    host = api.HostScene(abi.PT_SCENE_CHESS32, MODELS)
    switches(PT_STATS=1)
    device = api.DeviceScene.from_host(host)
    device.render(host.frame(480, 270, sample_count=4))
    stats = device.last_render_stats()
    assert stats["paths"] == 480 * 272 * 4       # path slots are handed out per 8x8 tile: 270 rows round up to 272
    assert stats["node_lanes"] > 10 * stats["paths"]
    assert stats["unsafe_skips"] == 0


@pytest.fixture(scope="module")
def both_kinds(api, orc):
    """Triangles AND many spheres (a chess row with the reference's 488 random spheres around it): the
    wavefront kernel searches them in one hierarchy."""
    from oracle.bindings import make_desc
    row = api.HostScene(abi.PT_SCENE_CHESS6_ROW, MODELS).arrays()
    balls = api.HostScene(abi.PT_SCENE_RANDOM_SPHERES).arrays()
    spheres = balls["spheres"].copy()
    spheres[1:, :3] = spheres[1:, :3] * 12.0 + np.array([0.0, 20.0, 0.0])      # spread over the pieces
    spheres[1:, 3] *= 12.0
    spheres[0] = [0.0, -1000.0, 0.0, 1000.0]
    arrays = dict(row)
    arrays["materials"] = np.concatenate([row["materials"], balls["materials"]])
    arrays["spheres"] = spheres
    arrays["sphere_material_indices"] = (balls["sphere_material_indices"] + len(row["materials"])).astype(np.int32)
    arrays["sphere_bvh"] = api.construct_sphere_bvh(spheres)
    arrays["background"] = balls["background"]
    host = api.HostScene(abi.PT_SCENE_CHESS6_ROW, MODELS)
    frame = host.frame(96, 54, sample_count=3, max_bounces=12)
    expect = orc.scene_create(arrays).render(frame, threads=8)
    return arrays, frame, expect, make_desc


@pytest.mark.parametrize("setting", [{}, {"PT_NODES": "f64"}, {"PT_KERNEL": "pixels"}, {"PT_DISTRUST_MASK": 1}, {"PT_STACK_SLOTS": 5},
                                     {"PT_STATS": 1}, {"PT_BVH": "reference"}])
def test_triangles_and_spheres_in_one_hierarchy(api, both_kinds, switches, setting):
    arrays, frame, expect, make_desc = both_kinds
    switches(**setting)
    desc, keep = make_desc(arrays)
    device = api.DeviceScene(desc, keepalive=keep)
    got = device.render(frame)
    if setting.get("PT_STATS") == 1:
        assert device.last_render_stats()["unsafe_skips"] == 0
    switches()
    assert same_bits(got, expect)
    tri, sph = device.primary_hits(frame)
    assert (tri >= 0).any() and (sph > 0).any()


@pytest.mark.parametrize("setting", [{"PT_DISTRUST_MASK": 0}, {"PT_STACK_SLOTS": 3}, {"PT_STATS": 1}])
def test_sphere_only_scene_through_every_fallback(api, orc, switches, setting):
    """The reference's random-spheres scene has no triangle: every path ends at a sphere or the sky.
    PT_DISTRUST_MASK=0 sends every winner to the exhaustive traversal."""
    host = api.HostScene(abi.PT_SCENE_RANDOM_SPHERES)
    frame = host.frame(72, 48, sample_count=2, max_bounces=10)
    switches(**setting)
    device = api.DeviceScene.from_host(host)
    got = device.render(frame)
    if setting.get("PT_STATS") == 1:
        assert device.last_render_stats()["unsafe_skips"] == 0
    switches()
    assert same_bits(got, orc.scene_create(host.arrays()).render(frame, threads=8))


def test_thresholds_only_change_scheduling(api, chess, switches):
    host, frame, expect = chess
    device = api.DeviceScene.from_host(host)
    for shade, leaf in [(1, 1), (64, 64), (7, 33)]:
        switches(PT_SHADE_THRESHOLD=shade, PT_LEAF_THRESHOLD=leaf)
        assert same_bits(device.render(frame), expect)


def test_unbounded_albedo_falls_back_to_the_pixel_kernel(api, orc, switches):
    """|albedo| > 1 breaks the wavefront kernel's "colour is zero until the last vertex" premise;
    such scenes must take the kernel that carries the colour explicitly and still match."""
    from oracle.bindings import make_desc
    host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, "pawn.triangles"))
    arrays = host.arrays()
    arrays["materials"]["v"][0][:3] = [1.5, 0.9, 2.0]
    desc, keep = make_desc(arrays)
    device = api.DeviceScene(desc, keepalive=keep)
    frame = host.frame(64, 64, sample_count=3)
    switches(PT_STATS=1)
    accum = device.render(frame)
    with pytest.raises(api.PathTracerError):
        device.last_render_stats()           # no wavefront launch happened
    switches()
    assert same_bits(accum, orc.scene_create(arrays).render(frame, threads=4))
