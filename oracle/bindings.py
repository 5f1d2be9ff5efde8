"""TEST INFRASTRUCTURE — ctypes bindings of the two CPU checkers.

  RefLib     oracle/_ref/libptref.so   the reference's own sources (ref_harness.cpp)
  OracleLib  oracle/libpt_oracle.so    the CPU restatement (pt_oracle.cpp)

Both expose the same Python surface (`prefix` differs) so a test can run the same
check against either.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
import ctypes as C
import os

import numpy as np

from pathtracer_amd.ctypes_abi import (MATERIAL_DTYPE, NODE_DTYPE, PtCamera, PtColour, PtFrame, PtMat3,
                                       PtSceneDesc, PtVec3, ptr)

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PATH = os.path.join(HERE, "_ref", "libptref.so")
ORACLE_PATH = os.path.join(HERE, "libpt_oracle.so")


class CpuScene:
    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    def __del__(self):
        if self.handle:
            self.lib._call("scene_destroy")(self.handle)
            self.handle = None

    def counts(self):
        out = np.zeros(5, np.int32)
        self.lib._call("scene_counts")(self.handle, ptr(out))
        return dict(zip(["materials", "spheres", "sphere_nodes", "triangles", "triangle_nodes"], out.tolist()))

    def export(self):
        n = self.counts()
        out = {
            "materials": np.zeros(n["materials"], MATERIAL_DTYPE),
            "spheres": np.zeros((n["spheres"], 4)),
            "sphere_material_indices": np.zeros(n["spheres"], np.int32),
            "sphere_bvh": np.zeros(n["sphere_nodes"], NODE_DTYPE),
            "triangles": np.zeros((n["triangles"], 9)),
            "triangle_material_indices": np.zeros(n["triangles"], np.int32),
            "triangle_bvh": np.zeros(n["triangle_nodes"], NODE_DTYPE),
            "background": np.zeros((2, 3)),
        }
        self.lib._call("scene_export")(self.handle, ptr(out["materials"]), ptr(out["spheres"]),
                                       ptr(out["sphere_material_indices"]), ptr(out["sphere_bvh"]),
                                       ptr(out["triangles"]), ptr(out["triangle_material_indices"]),
                                       ptr(out["triangle_bvh"]), ptr(out["background"]))
        return out

    def render(self, frame, accum=None, threads=1):
        if accum is None:
            accum = np.zeros(4 * frame.width * frame.height)
        rc = self.lib._call("render")(self.handle, C.byref(frame), ptr(accum), int(threads))
        if rc != 0:
            raise RuntimeError(f"{self.lib.prefix}render failed ({rc})")
        return accum

    def sample_colours(self, frame, sample):
        rgb = np.zeros((frame.height, frame.width, 3))
        rc = self.lib._call("sample_colours")(self.handle, C.byref(frame), int(sample), ptr(rgb))
        if rc != 0:
            raise RuntimeError("sample_colours failed")
        return rgb

    def primary_hits(self, frame):
        tri = np.full((frame.height, frame.width), -2, np.int32)
        sph = np.full((frame.height, frame.width), -2, np.int32)
        self.lib._call("primary_hits")(self.handle, C.byref(frame), ptr(tri), ptr(sph))
        return tri, sph

    def trace_triangles(self, rays):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        index = np.zeros(len(rays), np.int32)
        distance = np.zeros(len(rays))
        self.lib._call("trace_triangles")(self.handle, ptr(rays), C.c_longlong(len(rays)), ptr(index), ptr(distance))
        return index, distance


class _CpuLib:
    prefix = ""
    path = ""

    def __init__(self):
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path)
        self.dll = C.CDLL(self.path)
        for name, restype in [("noise_1d", C.c_uint), ("noise_3d", C.c_uint), ("random_number", C.c_uint),
                              ("real_from_rng", C.c_double), ("degrees_to_radians", C.c_double),
                              ("scene_create", C.c_void_p), ("model_scene", C.c_void_p),
                              ("builtin_scene", C.c_void_p)]:
            self._call(name).restype = restype
        self._call("real_from_rng").argtypes = [C.c_uint]
        self._call("random_number").argtypes = [C.c_uint]
        self._call("degrees_to_radians").argtypes = [C.c_double]
        self._call("scene_destroy").argtypes = [C.c_void_p]
        self._call("scene_counts").argtypes = [C.c_void_p, C.c_void_p]
        self._call("scene_export").argtypes = [C.c_void_p] * 9
        self._call("render").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self._call("sample_colours").argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self._call("primary_hits").argtypes = [C.c_void_p] * 4
        self._call("primary_rays").argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._call("trace_triangles").argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        self._call("rotation_matrix").argtypes = [C.c_double] * 4 + [C.c_void_p]
        self._call("camera_look_at").argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        self._call("camera_frame").argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p]
        self._call("camera_apply_input").argtypes = [C.c_void_p, C.c_void_p]
        self._call("tonemap_bgra8").argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
        self._call("construct_triangle_bvh").argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._call("construct_sphere_bvh").argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._call("model_scene").argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        self._call("builtin_scene").argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        self._call("scene_create").argtypes = [C.c_void_p]

    def _call(self, name):
        return getattr(self.dll, self.prefix + name)

    # ---- known-answer hooks
    def noise_1d(self, x):
        return self._call("noise_1d")(C.c_int(int(x)))

    def noise_3d(self, x, y, z):
        return self._call("noise_3d")(C.c_int(int(x)), C.c_int(int(y)), C.c_int(int(z)))

    def random_number(self, seed):
        return self._call("random_number")(int(seed))

    def real_from_rng(self, rng):
        return self._call("real_from_rng")(int(rng))

    def _rowwise(self, name, out_width, *inputs):
        arrays = [np.ascontiguousarray(a, np.float64) for a in inputs]
        n = len(arrays[0])
        out = np.zeros((n, out_width))
        fn = self._call(name)
        fn.argtypes = [C.c_void_p] * (len(arrays) + 1)
        strides = [a.strides[0] for a in arrays]
        bases = [a.ctypes.data for a in arrays]
        for i in range(n):
            fn(*[b + i * s for b, s in zip(bases, strides)], out.ctypes.data + i * out.strides[0])
        return out

    def random_unit_vector(self, seeds):
        return self._rowwise("random_unit_vector", 3, np.reshape(seeds, (-1, 3)))

    def ray_aabb(self, rays, aabbs):
        return self._rowwise("ray_aabb", 2, np.reshape(rays, (-1, 6)), np.reshape(aabbs, (-1, 6)))

    def ray_sphere(self, rays, spheres):
        return self._rowwise("ray_sphere", 3, np.reshape(rays, (-1, 6)), np.reshape(spheres, (-1, 4)))

    def ray_triangle(self, rays, triangles):
        return self._rowwise("ray_triangle", 2, np.reshape(rays, (-1, 6)), np.reshape(triangles, (-1, 9)))

    def unit_normal(self, triangles):
        return self._rowwise("unit_normal", 3, np.reshape(triangles, (-1, 9)))

    def scatter(self, rays, materials5, points, normals):
        return self._rowwise("scatter", 7, np.reshape(rays, (-1, 6)), np.reshape(materials5, (-1, 5)),
                             np.reshape(points, (-1, 3)), np.reshape(normals, (-1, 3)))

    # ---- host entry points
    def construct_triangle_bvh(self, triangles):
        triangles = np.ascontiguousarray(triangles, np.float64).reshape(-1, 9)
        nodes = np.zeros(2 * len(triangles) - 1, NODE_DTYPE)
        n = self._call("construct_triangle_bvh")(ptr(triangles), len(triangles), ptr(nodes))
        assert n == len(nodes)
        return nodes

    def construct_sphere_bvh(self, spheres):
        spheres = np.ascontiguousarray(spheres, np.float64).reshape(-1, 4)
        nodes = np.zeros(2 * len(spheres) - 1, NODE_DTYPE)
        n = self._call("construct_sphere_bvh")(ptr(spheres), len(spheres), ptr(nodes))
        assert n == len(nodes)
        return nodes

    def rotation_matrix(self, angle, ax, ay, az):
        m = PtMat3()
        self._call("rotation_matrix")(angle, ax, ay, az, C.byref(m))
        return np.array([[m.rows[i][j] for j in range(3)] for i in range(3)])

    def degrees_to_radians(self, degrees):
        return self._call("degrees_to_radians")(float(degrees))

    def camera_look_at(self, position, target, fov_y, aperture, focus_distance=0.0):
        cam = PtCamera()
        p, t = PtVec3(*position), PtVec3(*target)
        self._call("camera_look_at")(C.byref(p), C.byref(t), fov_y, aperture, focus_distance, C.byref(cam))
        return cam

    def camera_frame(self, camera, aspect_ratio, width, height):
        frame = PtFrame()
        self._call("camera_frame")(C.byref(camera), float(aspect_ratio), int(width), int(height), C.byref(frame))
        return frame

    def camera_apply_input(self, camera, viewer_input):
        """The frame loop's camera controls (src/main.cpp:890-985) applied to `camera` in place; True if it changed."""
        return bool(self._call("camera_apply_input")(C.byref(camera), C.byref(viewer_input)))

    def primary_rays(self, frame, sample):
        rays = np.zeros((frame.height, frame.width, 6))
        self._call("primary_rays")(C.byref(frame), int(sample), ptr(rays))
        return rays

    def tonemap_bgra8(self, accum, sample_count):
        accum = np.ascontiguousarray(accum, np.float64)
        out = np.zeros(accum.size, np.uint8)
        self._call("tonemap_bgra8")(ptr(accum), accum.size // 4, int(sample_count), ptr(out))
        return out

    # ---- scenes
    def scene_create(self, arrays, own_trees=True):
        """arrays: dict with materials, spheres, sphere_material_indices, triangles,
        triangle_material_indices, background (2x3); BVHs are built by the checker itself, unless
        `own_trees` is False: then the trees in `arrays` are traversed as they are."""
        desc, keep = make_desc(arrays, with_bvh=not own_trees)
        handle = self._call("scene_create")(C.byref(desc))
        del keep
        if not handle:
            raise RuntimeError("scene_create failed")
        return CpuScene(self, handle)

    def model_scene(self, triangles_path):
        cam, aspect = PtCamera(), C.c_double(0.0)
        handle = self._call("model_scene")(os.fsencode(triangles_path), C.byref(cam), C.byref(aspect))
        if not handle:
            raise RuntimeError(f"model_scene({triangles_path}) failed")
        return CpuScene(self, handle), cam, aspect.value

    def builtin_scene(self, kind):
        """kind 0 = the reference's random-spheres scene, 1 = its Cornell box."""
        cam, aspect = PtCamera(), C.c_double(0.0)
        handle = self._call("builtin_scene")(int(kind), C.byref(cam), C.byref(aspect))
        if not handle:
            raise RuntimeError(f"builtin_scene({kind}) failed")
        return CpuScene(self, handle), cam, aspect.value


def make_desc(arrays, with_bvh=True):
    """PtSceneDesc over a dict of numpy arrays; returns (desc, keepalive)."""
    keep = {}

    def arr(key, dtype, shape=None):
        a = arrays.get(key)
        if a is None or len(a) == 0:
            return None
        a = np.ascontiguousarray(a, dtype)
        if shape is not None:
            a = a.reshape(shape)
        keep[key] = a
        return a

    d = PtSceneDesc()
    m = arr("materials", MATERIAL_DTYPE)
    d.materials, d.material_count = ptr(m), len(m)
    s = arr("spheres", np.float64, (-1, 4))
    if s is not None:
        d.spheres, d.sphere_count = ptr(s), len(s)
        d.sphere_material_indices = ptr(arr("sphere_material_indices", np.int32))
        if with_bvh:
            b = arr("sphere_bvh", NODE_DTYPE)
            d.sphere_bvh, d.sphere_bvh_node_count = ptr(b), len(b)
    t = arr("triangles", np.float64, (-1, 9))
    if t is not None:
        d.triangles, d.triangle_count = ptr(t), len(t)
        d.triangle_material_indices = ptr(arr("triangle_material_indices", np.int32))
        if with_bvh:
            b = arr("triangle_bvh", NODE_DTYPE)
            d.triangle_bvh, d.triangle_bvh_node_count = ptr(b), len(b)
    bg = np.asarray(arrays["background"], np.float64).reshape(2, 3)
    d.background_gradient_start = PtColour(*bg[0])
    d.background_gradient_end = PtColour(*bg[1])
    return d, keep


class RefLib(_CpuLib):
    prefix = "ref_"
    path = REF_PATH


class OracleLib(_CpuLib):
    prefix = "orc_"
    path = ORACLE_PATH


def have_ref():
    return os.path.exists(REF_PATH)


def have_oracle():
    return os.path.exists(ORACLE_PATH)
