"""Host side of libpathtracer_amd.so: the entry points kept from the reference (model loading,
BVH build, camera, scenes, image output), error behaviour and the exported C ABI.  CPU only:
nothing here touches a GPU.
"""
import ctypes as C
import hashlib
import os
import subprocess
import re

import numpy as np
import pytest

from conftest import MODELS, ROOT, same_bits
from pathtracer_amd import ctypes_abi as abi

PIECES = ["pawn", "rook", "knight", "bishop", "queen", "king"]
COUNTS = {"pawn": 12286, "rook": 6716, "knight": 19972, "bishop": 15104, "queen": 14662, "king": 14764}


def test_library_exports_every_declared_symbol(api):
    header = open(os.path.join(ROOT, "include", "pathtracer_amd.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    for name in declared:
        assert getattr(api.lib(), name) is not None


def test_load_triangles_file(api, tmp_path):
    for name, count in COUNTS.items():
        path = os.path.join(MODELS, name + ".triangles")
        triangles = api.load_triangles_file(path)
        assert triangles.shape == (count, 9)
        assert triangles.tobytes() == open(path, "rb").read()
    # round trip through save_triangles_file
    out = tmp_path / "copy.triangles"
    api.save_triangles_file(str(out), triangles)
    assert out.read_bytes() == triangles.tobytes()
    # empty file: zero triangles, not an error (size % 72 == 0 holds)
    empty = tmp_path / "empty.triangles"
    empty.write_bytes(b"")
    assert api.load_triangles_file(str(empty)).shape == (0, 9)


def test_load_triangles_file_errors(api, tmp_path):
    with pytest.raises(api.PathTracerError) as e:
        api.load_triangles_file(str(tmp_path / "missing.triangles"))
    assert e.value.code == abi.PT_ERR_IO
    ragged = tmp_path / "ragged.triangles"
    ragged.write_bytes(b"\0" * 100)          # the reference asserts size % sizeof(Triangle) == 0
    with pytest.raises(api.PathTracerError) as e:
        api.load_triangles_file(str(ragged))
    assert e.value.code == abi.PT_ERR_IO and "72-byte" in str(e.value)


def test_bvh_build_matches_reference_digests(api, golden):
    for name in PIECES:
        triangles = api.load_triangles_file(os.path.join(MODELS, name + ".triangles"))
        nodes = api.construct_triangle_bvh(triangles)
        assert hashlib.sha256(nodes.tobytes()).digest() == golden[f"bvh_sha_{name}"].tobytes(), name
    assert api.construct_sphere_bvh(golden["sphere_bvh_in"]).tobytes() == golden["sphere_bvh_out"].tobytes()


def test_bvh_build_structure_and_edge_cases(api, orc):
    rng = np.random.default_rng(3)
    # one primitive: a single leaf node
    one = rng.uniform(-1, 1, (1, 9))
    nodes = api.construct_triangle_bvh(one)
    assert len(nodes) == 1 and nodes[0]["left"] == 0 and nodes[0]["right"] == 0 and nodes[0]["index"] == 0
    # many equal keys: topology depends on how std::sort orders ties; must equal the checker's
    grid = np.repeat(rng.integers(-3, 3, (300, 1, 3)).astype(float), 3, axis=1) + rng.integers(0, 2, (300, 3, 3))
    tri = grid.reshape(-1, 9)
    assert api.construct_triangle_bvh(tri).tobytes() == orc.construct_triangle_bvh(tri).tobytes()
    # the threaded build (>= 32768 primitives) lands every node in the same slot as the serial one
    big = rng.uniform(-100, 100, (40000, 9))
    assert api.construct_triangle_bvh(big).tobytes() == orc.construct_triangle_bvh(big).tobytes()
    # ranges of >= 65536 primitives are sorted by several threads together (partition steps of one std::sort dealt
    # out): with keys from a small lattice almost every comparison is a tie, and the permutation must still be std::sort's
    lattice = np.repeat(rng.integers(-12, 12, (150000, 1, 3)).astype(float), 3, axis=1) + rng.integers(0, 2, (150000, 3, 3))
    tri = lattice.reshape(-1, 9)
    assert api.construct_triangle_bvh(tri).tobytes() == orc.construct_triangle_bvh(tri).tobytes()
    with pytest.raises(api.PathTracerError):
        api.construct_triangle_bvh(np.zeros((0, 9)))


def test_host_builders_do_not_depend_on_the_thread_count():
    """The reference-shaped BVH, the search hierarchy and its four-wide form of the chess set (422,598 triangles: long enough
    for every several-threads-on-one-range path) built by 1, 3 and 8 threads: the same bytes."""
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "consumer", "hierarchy_digest")
    assert os.path.exists(tool), "run `make` first"
    lines = set()
    for threads in (1, 3, 8):
        done = subprocess.run([tool, str(abi.PT_SCENE_CHESS32), MODELS], env=dict(os.environ, PT_BUILD_THREADS=str(threads)),
                              capture_output=True, text=True, timeout=300)
        assert done.returncode == 0, done.stderr
        lines.add(done.stdout)
    assert len(lines) == 1, lines
    assert "reference nodes 845195" in lines.pop()


def test_camera_and_frame(api, orc, golden):
    cam = api.camera_look_at((0.0, 150.0, 150.0), (0.0, 0.0, 0.0), orc.degrees_to_radians(40.0), float(np.float32(0.1)))
    assert bytes(cam) == golden["rook_camera"].tobytes()
    assert bytes(api.camera_frame(cam, 1.0, 64, 48)) == golden["rook_frame"].tobytes()
    assert bytes(api.camera_frame(cam, 64 / 36, 64, 36)) == golden["pawn_frame"].tobytes()
    for angle, axis in [(0.3, (1, 2, 3)), (-1.25, (0, 0, 1)), (float(np.float32(np.pi)), (0, 1, 0))]:
        assert same_bits(api.rotation_matrix(angle, *axis), orc.rotation_matrix(angle, *axis))
    with pytest.raises(api.PathTracerError):
        api.camera_frame(cam, 1.0, 1, 48)       # u = column / (width - 1)


def test_model_scene_is_the_reference_scene(api, orc, golden):
    for name in ("rook", "pawn"):
        host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, name + ".triangles"))
        checker, cam, aspect = orc.model_scene(os.path.join(MODELS, name + ".triangles"))
        assert bytes(host.camera) == bytes(cam) == golden[f"{name}_camera"].tobytes()
        assert host.aspect_ratio == aspect == 1.0
        ours, theirs = host.arrays(), checker.export()
        for key in theirs:
            assert ours[key].tobytes() == theirs[key].tobytes(), (name, key)


@pytest.mark.parametrize("name,kind,index", [("spheres", abi.PT_SCENE_RANDOM_SPHERES, 0), ("cornell", abi.PT_SCENE_CORNELL_BOX, 1)])
def test_reference_builtin_scenes(api, orc, golden, name, kind, index):
    """src/main.cpp:516-625 (random spheres) and :627-798 (Cornell box): the library's scene arrays,
    BVHs and cameras against the CPU restatement and against the arrays the reference's own code made."""
    host = api.HostScene(kind)
    checker, cam, aspect = orc.builtin_scene(index)
    assert bytes(host.camera) == bytes(cam) == golden[f"{name}_camera"].tobytes()
    assert host.aspect_ratio == aspect == (1.5 if name == "spheres" else 1.0)
    ours, theirs = host.arrays(), checker.export()
    for key in theirs:
        assert ours[key].tobytes() == theirs[key].tobytes() == golden[f"{name}_scene_{key}"].tobytes(), (name, key)
    assert len(ours["spheres"]) == (488 if name == "spheres" else 1)
    assert len(ours["triangles"]) == (0 if name == "spheres" else 36)


def test_chess_scenes(api, golden):
    six = api.HostScene(abi.PT_SCENE_CHESS6_ROW, MODELS)
    arrays = six.arrays()
    assert len(arrays["triangles"]) == sum(COUNTS.values()) == 83504
    assert hashlib.sha256(arrays["triangles"].tobytes()).digest() == golden["chess6_triangles_sha"].tobytes()
    assert hashlib.sha256(arrays["triangle_bvh"].tobytes()).digest() == golden["chess6_bvh_sha"].tobytes()
    assert bytes(six.camera) == golden["chess6_camera"].tobytes()

    full = api.HostScene(abi.PT_SCENE_CHESS32, MODELS)
    arrays = full.arrays()
    expect = 16 * COUNTS["pawn"] + 4 * (COUNTS["rook"] + COUNTS["knight"] + COUNTS["bishop"]) + 2 * (COUNTS["queen"] + COUNTS["king"]) + 2
    assert len(arrays["triangles"]) == expect == 422598
    assert len(arrays["triangle_bvh"]) == 2 * expect - 1
    assert set(arrays["triangle_material_indices"].tolist()) == {0, 2, 3}
    assert (arrays["triangle_material_indices"] == 2).sum() == (arrays["triangle_material_indices"] == 0).sum()
    mixed = api.HostScene(abi.PT_SCENE_CHESS32_MIXED, MODELS).arrays()
    assert mixed["triangles"].tobytes() == arrays["triangles"].tobytes()
    used = set(mixed["materials"]["type"][np.unique(mixed["triangle_material_indices"])].tolist())
    assert used == {abi.PT_LAMBERTIAN, abi.PT_METAL, abi.PT_DIELECTRIC}

    with pytest.raises(api.PathTracerError) as e:
        api.HostScene(abi.PT_SCENE_CHESS32, str(ROOT))        # no model files there
    assert e.value.code == abi.PT_ERR_IO
    with pytest.raises(api.PathTracerError) as e:
        api.HostScene(99)
    assert e.value.code == abi.PT_ERR_UNSUPPORTED


def test_stl_parser_reproduces_the_reference_assets(api, tmp_path):
    """parse_stl_file (src/model_loading.cpp:63-196) is how the reference's .triangles assets were
    made from its .stl assets: parsing an .stl must give the bytes of the matching .triangles."""
    head = api.parse_stl_file(os.path.join(ROOT, "tests", "golden", "rook_head.stl"))
    rook = np.fromfile(os.path.join(MODELS, "rook.triangles")).reshape(-1, 9)
    assert head.shape == (60, 9) and head.tobytes() == rook[:60].tobytes()
    stl_dir = "/root/reference/models/stl"
    if os.path.isdir(stl_dir):                      # the full assets, where the reference is present
        for name, count in COUNTS.items():
            parsed = api.parse_stl_file(os.path.join(stl_dir, name + ".stl"))
            assert len(parsed) == count
            assert parsed.tobytes() == np.fromfile(os.path.join(MODELS, name + ".triangles")).tobytes()
    # round trip through the .triangles writer, and malformed input
    out = tmp_path / "head.triangles"
    api.save_triangles_file(str(out), head)
    assert api.load_triangles_file(str(out)).tobytes() == head.tobytes()
    bad = tmp_path / "bad.stl"
    bad.write_text("solid x\n facet normal 0 0 1\n outer loop\n vertex 0 0 0\n vertex 1 0 zero\n")
    with pytest.raises(api.PathTracerError) as e:
        api.parse_stl_file(str(bad))
    assert e.value.code == abi.PT_ERR_IO
    two = tmp_path / "two.stl"
    two.write_text("solid x\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nendloop\nendfacet\n")
    with pytest.raises(api.PathTracerError):
        api.parse_stl_file(str(two))


def test_tonemap_and_pixel_file(api, golden, tmp_path):
    for name in ("rook", "pawn"):
        assert np.array_equal(api.tonemap_bgra8(golden[f"{name}_accum4"], 4), golden[f"{name}_bgra8"])
    # clamp at 1, zero stays zero, alpha always 255
    accum = np.array([0.0, 4.0, 400.0, 4.0, 1.0, 2.25, 0.25, 4.0])
    assert api.tonemap_bgra8(accum, 4).tolist() == [0, 255, 255, 255, 127, 191, 63, 255]
    bgra = golden["rook_bgra8"]
    path = tmp_path / "pixels.data"
    api.write_pixel_data_file(str(path), bgra)
    data = np.frombuffer(path.read_bytes(), np.uint8).reshape(-1, 4)
    assert np.array_equal(data[:, [2, 1, 0, 3]], bgra.reshape(-1, 4))     # RGBA on disk, rows untouched (bottom-up)
    with pytest.raises(api.PathTracerError):
        api.tonemap_bgra8(accum, 0)
    # PPM convenience: header, RGB order, top row first
    ppm = tmp_path / "image.ppm"
    api.write_ppm_file(str(ppm), bgra, 64, 48)
    raw = ppm.read_bytes()
    assert raw.startswith(b"P6\n64 48\n255\n")
    body = np.frombuffer(raw[len(b"P6\n64 48\n255\n"):], np.uint8).reshape(48, 64, 3)
    assert np.array_equal(body[::-1], bgra.reshape(48, 64, 4)[:, :, [2, 1, 0]])


def test_scene_create_rejects_bad_input_without_a_gpu(api):
    from oracle.bindings import make_desc
    host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, "rook.triangles"))
    arrays = host.arrays()
    handle = C.c_void_p()

    def create(**changes):
        desc, keep = make_desc({**arrays, **changes})
        code = api.lib().pt_scene_create(C.byref(desc), 0, C.byref(handle))
        return code, api.lib().pt_last_error().decode()

    bad = arrays["triangle_material_indices"].copy()
    bad[17] = 5
    code, message = create(triangle_material_indices=bad)
    assert code == abi.PT_ERR_INVALID_ARGUMENT and "triangle 17" in message
    mats = arrays["materials"].copy()
    mats[0]["type"] = 9
    code, message = create(materials=mats)
    assert code == abi.PT_ERR_INVALID_ARGUMENT and "unknown type" in message
    assert api.lib().pt_scene_create(None, 0, C.byref(handle)) == abi.PT_ERR_INVALID_ARGUMENT


def test_viewer_camera_controls_follow_the_reference(api, orc, golden):
    """pt_camera_apply_input (src/main.cpp:890-985) over 300 frames of mouse, wheel and key input applied one after
    another: the camera after every frame, and whether it changed, as the reference's own code left them; the CPU
    restatement agrees too."""
    host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, "rook.triangles"))
    mine, theirs = abi.PtCamera.from_buffer_copy(bytes(host.camera)), abi.PtCamera.from_buffer_copy(bytes(host.camera))
    for i in range(len(golden["viewer_inputs"])):
        v = abi.PtViewerInput(*[int(x) for x in golden["viewer_inputs"][i]], float(golden["viewer_durations"][i]))
        assert api.camera_apply_input(mine, v) == bool(golden["viewer_modified"][i])
        assert orc.camera_apply_input(theirs, v) == bool(golden["viewer_modified"][i])
        assert bytes(mine) == golden["viewer_cameras"][i].tobytes() == bytes(theirs), i
    assert not golden["viewer_modified"][0] and golden["viewer_modified"].sum() > 200
    with pytest.raises(api.PathTracerError):
        api.check(api.lib().pt_camera_apply_input(None, None, None))
