"""Small synthetic scenes at the edges of the GPU path's code paths, each against the CPU checker,
bit for bit: no primitive at all, one and two triangles (a 4-wide node with empty slots), degenerate
triangles, coordinates outside the single-precision range, hollow spheres, spheres only.
"""
import os

import numpy as np
import pytest

from conftest import same_bits
from pathtracer_amd import ctypes_abi as abi

pytestmark = pytest.mark.gpu

from campaign import SKY, random_frame, random_scene  # noqa: F401
from campaign import materials as _materials  # noqa: F401
from campaign import scene_arrays as _scene


def _compare(api, orc, arrays, eye, target, size=(40, 30), spp=3, bounces=12, aperture=0.05, own_trees=True):
    from oracle.bindings import make_desc
    desc, keep = make_desc(arrays)
    device = api.DeviceScene(desc, keepalive=keep)
    checker = orc.scene_create(arrays, own_trees=own_trees)
    cam = api.camera_look_at(eye, target, 0.7, aperture)
    frame = api.camera_frame(cam, size[0] / size[1], size[0], size[1], sample_count=spp, max_bounces=bounces)
    tri, sph = device.primary_hits(frame)
    tri_c, sph_c = checker.primary_hits(frame)
    assert np.array_equal(tri, tri_c) and np.array_equal(sph, sph_c)
    assert same_bits(device.render(frame), checker.render(frame, threads=4))
    return tri, sph


def test_scene_without_primitives_is_all_background(api, orc):
    tri, sph = _compare(api, orc, _scene(api), (0.0, 1.0, 5.0), (0.0, 0.0, 0.0))
    assert (tri == -1).all() and (sph == -1).all()


@pytest.mark.parametrize("count", [1, 2, 3, 5])
def test_few_triangles(api, orc, count):
    """1: no inner node at all; 2, 3: one 4-wide node with two / one empty slots; 5: two levels."""
    rng = np.random.default_rng(count)
    triangles = rng.uniform(-3, 3, (count, 9))
    tri, _ = _compare(api, orc, _scene(api, triangles, tri_materials=[0, 3, 1, 2, 0][:count]), (0.0, 0.0, 12.0), (0.0, 0.0, 0.0))
    assert (tri >= 0).any()


def test_degenerate_triangles_are_never_hit(api, orc):
    """Zero-area triangles divide by n.n = 0 (src/geometry.cpp:124-126): NaN barycentrics, no hit; they
    still sit in the hierarchy with flat boxes."""
    rng = np.random.default_rng(3)
    good = rng.uniform(-4, 4, (40, 9))
    point = np.tile(rng.uniform(-4, 4, (10, 3)), 3)
    a, b = rng.uniform(-4, 4, (10, 3)), rng.uniform(-4, 4, (10, 3))
    line = np.concatenate([a, b, a + 0.5 * (b - a)], axis=1)
    triangles = np.concatenate([good, point, line])
    tri, _ = _compare(api, orc, _scene(api, triangles, tri_materials=np.zeros(60, np.int32)), (0.0, 0.0, 15.0), (0.0, 0.0, 0.0), spp=2)
    assert (tri >= 0).any() and not (tri >= 40).any()


def test_coordinates_outside_the_single_precision_range(api, orc):
    """Boxes beyond 2^20 make the library keep the exact double-precision nodes for the whole scene."""
    rng = np.random.default_rng(8)
    triangles = rng.uniform(-3, 3, (200, 9)) * 1.0e6 + np.tile([4.0e7, -2.0e7, 1.0e7], 3)
    tri, _ = _compare(api, orc, _scene(api, triangles), (4.0e7, -2.0e7, 4.0e7), (4.0e7, -2.0e7, 1.0e7), aperture=1.0e3)
    assert (tri >= 0).any()


def test_coordinates_near_the_edge_of_the_single_precision_range(api, orc):
    """Small triangles far from the origin, still inside the range of the single-precision nodes: a float
    is 0.06 wide out there, so the planes, the ray origin and their difference are all badly rounded.
    The conservative test must still never skip a box the exact test accepts (counted by the
    cross-checking build), and the bytes must match."""
    from oracle.bindings import make_desc
    rng = np.random.default_rng(12)
    centre = np.array([9.0e5, -9.0e5, 9.0e5])
    triangles = rng.uniform(-6, 6, (300, 1, 3)) + rng.uniform(-1.5, 1.5, (300, 3, 3)) + centre
    arrays = _scene(api, triangles.reshape(-1, 9))
    desc, keep = make_desc(arrays)
    os.environ["PT_STATS"] = "1"
    try:
        device = api.DeviceScene(desc, keepalive=keep)
        cam = api.camera_look_at(centre + [0.0, 3.0, 30.0], centre, 0.7, 0.05)
        frame = api.camera_frame(cam, 4 / 3, 48, 36, sample_count=4, max_bounces=12)
        got = device.render(frame)
        stats = device.last_render_stats()
    finally:
        del os.environ["PT_STATS"]
    assert stats["node_lanes"] > 0 and stats["unsafe_skips"] == 0
    checker = orc.scene_create(arrays)
    assert same_bits(got, checker.render(frame, threads=4))
    tri, _ = device.primary_hits(frame)
    assert (tri >= 0).any()


@pytest.mark.parametrize("hollow", [False, True])
def test_spheres_only(api, orc, hollow):
    """Several spheres and no triangle: the search hierarchy holds spheres alone.  Negative radii are
    the reference's hollow spheres (src/path_tracing.cpp:190-191); their boxes are inverted
    (src/geometry.cpp:51-57) and the reference's own inner boxes decide what reaches them, so such a
    scene walks the reference's sphere tree instead."""
    rng = np.random.default_rng(21)
    centres = rng.uniform(-4, 4, (30, 3))
    radii = rng.uniform(0.3, 1.2, 30) * (rng.choice([1.0, 1.0, -1.0], 30) if hollow else 1.0)
    spheres = np.concatenate([centres, radii[:, None]], axis=1)
    shell = -1.4 if hollow else 1.4
    spheres = np.concatenate([spheres, [[0.0, -104.0, 0.0, 100.0], [0.0, 1.0, 0.0, 1.5], [0.0, 1.0, 0.0, shell]]])   # ground, glass shell
    sph_materials = np.concatenate([np.arange(30) % 3, [0, 2, 2]])
    _, sph = _compare(api, orc, _scene(api, spheres=spheres, sph_materials=sph_materials), (0.0, 3.0, 14.0), (0.0, 0.0, 0.0), spp=4, bounces=20)
    assert (sph >= 0).any()


def test_two_spheres_and_two_triangles(api, orc):
    """The smallest scene whose spheres join the triangles in one hierarchy."""
    triangles = [[-5.0, -1.0, -5.0, 5.0, -1.0, -5.0, 0.0, -1.0, 5.0], [-2.0, 0.0, -3.0, 2.0, 0.0, -3.0, 0.0, 3.0, -3.0]]
    spheres = [[-1.5, 0.0, 0.0, 1.0], [1.5, 0.0, 0.0, 1.0]]
    tri, sph = _compare(api, orc, _scene(api, triangles, spheres, tri_materials=[0, 3], sph_materials=[1, 2]), (0.0, 2.0, 9.0), (0.0, 0.0, 0.0),
                        size=(48, 36), spp=4, bounces=16)
    assert (tri >= 0).any() and (sph == 0).any() and (sph == 1).any()


def test_a_callers_tree_whose_boxes_do_not_nest_is_searched_as_it_is(api, orc):
    """pt_scene_create takes the caller's BVH.  If an inner box does not contain what is below it, which leaves the reference
    reaches is decided by that box (src/path_tracing.cpp:141-169), not by the leaves' own boxes: the library then keeps the
    caller's tree, skips nothing by distance, and renders with the one-lane-per-pixel kernel.  Same for a sphere tree."""
    rng = np.random.default_rng(77)
    triangles = (rng.uniform(-6, 6, (120, 1, 3)) + rng.normal(0, 1.2, (120, 3, 3))).reshape(-1, 9)
    spheres = np.concatenate([rng.uniform(-6, 6, (12, 3)), rng.uniform(0.4, 1.5, (12, 1))], 1)
    arrays = _scene(api, triangles, spheres, tri_materials=rng.integers(0, 4, 120), sph_materials=rng.integers(0, 4, 12))
    whole_tri, whole_sph = _compare(api, orc, arrays, (3.0, 4.0, 22.0), (0.0, 0.0, 0.0), size=(56, 40), spp=2, bounces=8)
    for key in ("triangle_bvh", "sphere_bvh"):
        cut = dict(arrays)
        nodes = arrays[key].copy()
        inner = [i for i in range(1, len(nodes)) if nodes[i]["left"] != 0][:6]
        for i in inner[::2]:                         # halve some inner boxes towards their minimum corner
            nodes["aabb"][i, 3:6] = 0.5 * (nodes["aabb"][i, 0:3] + nodes["aabb"][i, 3:6])
        cut[key] = nodes
        tri, sph = _compare(api, orc, cut, (3.0, 4.0, 22.0), (0.0, 0.0, 0.0), size=(56, 40), spp=2, bounces=8, own_trees=False)
        changed = tri != whole_tri if key == "triangle_bvh" else sph != whole_sph
        assert changed.any(), key                     # the cut boxes hide primitives from some rays: the tree, not the leaves, decided
        # a NaN in a box: what the slab test returns then depends on the argument order of the reference's std::min / std::max
        # (src/geometry.cpp:33-34), which the exact form of the test keeps
        for axis in range(6):
            poisoned = dict(arrays)
            nodes = arrays[key].copy()
            nodes["aabb"][inner[axis % len(inner)], axis] = np.nan
            nodes["aabb"][0, (axis + 1) % 6] = np.nan
            poisoned[key] = nodes
            _compare(api, orc, poisoned, (3.0, 4.0, 22.0), (0.0, 0.0, 0.0), size=(56, 40), spp=2, bounces=8, own_trees=False)
    with pytest.raises(api.PathTracerError):          # the build sorts box minima: a NaN among them has no place in any order
        bad = triangles.copy()
        bad[5, :] = np.nan
        api.construct_triangle_bvh(bad)


def test_a_metal_among_the_materials_changes_nothing_else(api, orc):
    """A scene with a metal in its material table scatters Lambertian and metal lanes in one body, a scene without one
    keeps the Lambertian body alone: the same triangles and spheres rendered with a table that holds no metal, and with
    the same table plus a metal nothing uses, give the same bytes (and the checker's)."""
    from oracle.bindings import make_desc
    rng = np.random.default_rng(77)
    centre = rng.uniform(-5, 5, (200, 1, 3))
    triangles = (centre + rng.normal(0, 1.2, (200, 3, 3))).reshape(-1, 9)
    spheres = np.concatenate([rng.uniform(-5, 5, (6, 3)), rng.uniform(0.4, 1.5, (6, 1))], 1)
    without = _materials()[[0, 2, 3]]                      # Lambertian, glass, light
    arrays = _scene(api, triangles, spheres, tri_materials=rng.integers(0, 3, 200), sph_materials=rng.integers(0, 3, 6))
    arrays["materials"] = without
    unused = _materials()[[0, 2, 3, 1]]                    # the same three, then a metal no primitive refers to
    renders = []
    for table in (without, unused):
        arrays["materials"] = table
        desc, keep = make_desc(arrays)
        device = api.DeviceScene(desc, keepalive=keep)
        cam = api.camera_look_at((0.0, 2.0, 14.0), (0.0, 0.0, 0.0), 0.7, 0.05)
        frame = api.camera_frame(cam, 4 / 3, 64, 48, sample_count=4, max_bounces=20)
        renders.append(device.render(frame))
    assert same_bits(renders[0], renders[1])
    assert same_bits(renders[0], orc.scene_create(arrays).render(frame, threads=4))

# PT_RANDOM_SCENES=<n> widens the campaign (a one-off run of 500 is logged under profiles/r2/); seeds from 16 on also draw
# soups of 3,000 and 30,000 triangles.
@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("PT_RANDOM_SCENES", "16")))) | {27, 63}))     # 63: hollow spheres under boxes that do not hold them; 27: infinite coordinates
def test_random_scenes_against_the_checker(api, orc, seed):
    """Seeded random scenes of every shape the library takes: triangle soups of 1..400 triangles (some on an integer
    lattice, so that coplanar and duplicated triangles tie), 0..40 spheres (hollow ones in every fourth scene), all
    four materials, cameras inside and outside the geometry, odd image sizes.  Primary hit indices and buffers, bit for bit."""
    arrays, how, _ = random_scene(api, seed)
    _compare(api, orc, arrays, how["eye"], how["target"], size=how["size"], spp=how["spp"], bounces=how["bounces"], aperture=how["aperture"])


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_RANDOM_CUT_SCENES", "24"))))
def test_random_scenes_with_every_triangle_cut_up(api, orc, seed, monkeypatch, capfd):
    """The same random scenes with the search hierarchy built over *pieces* of the triangles (the library cuts large triangles
    that fill little of their boxes into several references; by default only in scenes of thousands of triangles): here
    every scene is cut, far finer than the default bar would, lattice ties, duplicates, degenerate, huge and infinite
    triangles included.  Primary hit indices and buffers, bit for bit.  PT_RANDOM_CUT_SCENES=<n> widens the campaign."""
    monkeypatch.setenv("PT_PRESPLIT_LEAST", "1")
    monkeypatch.setenv("PT_PRESPLIT_FLOOR", ["0.02", "0.2", "0.002"][seed % 3])
    monkeypatch.setenv("PT_PRESPLIT", ["8", "1.5", "30"][seed % 3])
    monkeypatch.setenv("PT_VERBOSE", "1")
    arrays, how, _ = random_scene(api, seed)
    _compare(api, orc, arrays, how["eye"], how["target"], size=how["size"], spp=how["spp"], bounces=how["bounces"], aperture=how["aperture"])
    import re
    added = [int(n) for n in re.findall(r"(\d+) references more", capfd.readouterr().err)]
    print("seed", seed, "triangles", len(arrays["triangles"]), "references added", added)
    if seed in (2, 8, 14, 17):
        assert added and max(added) > 0, "this scene's triangles should have been cut"


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_RANDOM_FRAMES", "12"))))
def test_random_frames_against_the_checker(api, orc, seed):
    """Seeded random *frames* over a small random scene: odd sizes, row ranges that start anywhere, shards, sample ranges that
    start anywhere and span several groups of 16, any depth (0 = the one-lane-per-pixel kernel), one group per launch or all
    at once, both kernels, into a buffer that already holds numbers.  Every double against the CPU checker.
    PT_RANDOM_FRAMES=<n> widens the campaign."""
    from oracle.bindings import make_desc
    arrays, frame, held, rng, _ = random_frame(api, seed)
    width, height = frame.width, frame.height
    desc, keep = make_desc(arrays)
    device = api.DeviceScene(desc, keepalive=keep)
    checker = orc.scene_create(arrays)
    saved = {k: os.environ.pop(k, None) for k in ("PT_WORKSPACE_GB", "PT_KERNEL")}
    try:
        if seed % 2 == 1:
            os.environ["PT_WORKSPACE_GB"] = "0"          # one sample group per launch
        if seed % 7 == 3:
            os.environ["PT_KERNEL"] = "pixels"
        if seed % 5 == 2:
            # the entry point that allocates nothing (frames of this size always fit its static workspace), from and to device memory
            import torch
            buffer = torch.from_numpy(held.copy()).to("cuda:0")
            device.render_device_noalloc(frame, buffer.data_ptr())
            torch.cuda.synchronize()
            got = buffer.cpu().numpy()
        else:
            got = device.render(frame, held.copy())
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert same_bits(got, checker.render(frame, held.copy(), threads=4))


def test_corrupted_trees_are_refused_or_followed(api, orc):
    """A caller's tree with one field overwritten: pt_scene_create either refuses it (an index the reference would follow out of
    its arrays, a node reached twice, a primitive named twice or never) or, if it is still a tree over every primitive, renders
    what the reference's traversal of those very nodes gives."""
    from oracle.bindings import make_desc
    rng = np.random.default_rng(404)
    triangles = (rng.uniform(-6, 6, (90, 1, 3)) + rng.normal(0, 1.0, (90, 3, 3))).reshape(-1, 9)
    arrays = _scene(api, triangles, tri_materials=rng.integers(0, 4, 90))
    refused = followed = 0
    for trial in range(60):
        nodes = arrays["triangle_bvh"].copy()
        i = int(rng.integers(0, len(nodes)))
        field = ("left", "right", "index")[trial % 3]
        nodes[field][i] = int(rng.choice([-1, 0, 1, i, len(nodes), len(nodes) + 7, 2 ** 31 - 1, -2 ** 31, int(rng.integers(0, len(nodes)))]))
        broken = dict(arrays)
        broken["triangle_bvh"] = nodes
        desc, keep = make_desc(broken)
        try:
            api.DeviceScene(desc, keepalive=keep)
        except api.PathTracerError as e:
            assert e.code == abi.PT_ERR_INVALID_ARGUMENT, e
            refused += 1
            continue
        _compare(api, orc, broken, (2.0, 3.0, 20.0), (0.0, 0.0, 0.0), size=(40, 28), spp=1, bounces=4, own_trees=False)
        followed += 1
    assert refused > 20 and followed > 0, (refused, followed)
