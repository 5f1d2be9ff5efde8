"""Thin Python binding of libpathtracer_amd.so (the C ABI in include/pathtracer_amd.h).

Used by the tests and bench.py; it adds no behaviour of its own.  There is no CPU fallback:
if the shared library is missing, or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

from . import ctypes_abi as abi
from .ctypes_abi import (MATERIAL_DTYPE, NODE_DTYPE, PtCamera, PtFrame, PtMat3, PtSceneDesc, PtVec3, PtViewerInput, ptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, "libpathtracer_amd.so")
MODELS_DIR = os.path.join(os.path.dirname(_HERE), "assets", "models")

EXPORTS = [
    "pt_last_error", "pt_load_triangles_file", "pt_save_triangles_file", "pt_parse_stl_file", "pt_free", "pt_construct_triangle_bvh",
    "pt_construct_sphere_bvh", "pt_rotation_matrix", "pt_look_at_matrix", "pt_camera_look_at", "pt_camera_frame", "pt_camera_apply_input", "pt_viewer_frame",
    "pt_host_scene_build", "pt_host_scene_desc", "pt_host_scene_camera", "pt_host_scene_destroy", "pt_tonemap_bgra8", "pt_tonemap_device",
    "pt_write_pixel_data_file", "pt_write_ppm_file", "pt_scene_create", "pt_scene_destroy", "pt_render", "pt_render_device",
    "pt_render_device_noalloc", "pt_primary_hits", "pt_last_render_ms", "pt_last_render_stats", "pt_device_eval",
    "pt_scene_set_workspace_limit", "pt_last_launch_shape",
]


class PathTracerError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"pathtracer_amd error {code}: {message}")
        self.code = code


_lib = None


def lib():
    """The loaded shared library (loads it on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBRARY_PATH):
            raise FileNotFoundError(f"{LIBRARY_PATH} is missing: run `make` (or __graft_entry__.build()) first; "
                                    "there is no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm carries its own libamdhip64.so (SONAME
        # libamdhip64.so.7).  Loaded first, it satisfies this library's dependency by SONAME; loaded
        # second, the loader would bring in a second runtime that finds no usable GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        dll = C.CDLL(LIBRARY_PATH)
        dll.pt_last_error.restype = C.c_char_p
        dll.pt_free.restype = None
        dll.pt_free.argtypes = [C.c_void_p]
        dll.pt_host_scene_destroy.restype = None
        dll.pt_host_scene_destroy.argtypes = [C.c_void_p]
        dll.pt_scene_destroy.restype = None
        dll.pt_scene_destroy.argtypes = [C.c_void_p]
        dll.pt_load_triangles_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        dll.pt_parse_stl_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        dll.pt_save_triangles_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
        dll.pt_construct_triangle_bvh.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        dll.pt_construct_sphere_bvh.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        dll.pt_rotation_matrix.argtypes = [C.c_double] * 4 + [C.c_void_p]
        dll.pt_look_at_matrix.argtypes = [C.c_void_p] * 3
        dll.pt_camera_look_at.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
        dll.pt_camera_frame.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_int32, C.c_void_p]
        dll.pt_camera_apply_input.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        dll.pt_viewer_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                        C.c_void_p, C.c_void_p]
        dll.pt_host_scene_build.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
        dll.pt_host_scene_desc.argtypes = [C.c_void_p, C.c_void_p]
        dll.pt_host_scene_camera.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        dll.pt_tonemap_bgra8.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        dll.pt_tonemap_device.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        dll.pt_write_pixel_data_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
        dll.pt_write_ppm_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        dll.pt_scene_create.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        dll.pt_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        dll.pt_render_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        dll.pt_render_device_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        dll.pt_primary_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        dll.pt_last_render_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        dll.pt_last_render_stats.argtypes = [C.c_void_p, C.c_void_p]
        dll.pt_scene_set_workspace_limit.argtypes = [C.c_void_p, C.c_uint64]
        dll.pt_last_launch_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        dll.pt_device_eval.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        _lib = dll
    return _lib


def check(code):
    if code != abi.PT_OK:
        raise PathTracerError(code, lib().pt_last_error().decode())


# ------------------------------------------------------------------ host entry points
def load_triangles_file(path):
    """(n, 9) float64 array of the file's triangles."""
    data, count = C.c_void_p(), C.c_int64()
    check(lib().pt_load_triangles_file(os.fsencode(path), C.byref(data), C.byref(count)))
    try:
        out = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_double)), shape=(count.value, 9)).copy()
    finally:
        lib().pt_free(data)
    return out


def parse_stl_file(path):
    """(n, 9) float64 array of the triangles of an ASCII STL file."""
    data, count = C.c_void_p(), C.c_int64()
    check(lib().pt_parse_stl_file(os.fsencode(path), C.byref(data), C.byref(count)))
    if count.value == 0:
        return np.zeros((0, 9))
    try:
        out = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_double)), shape=(count.value, 9)).copy()
    finally:
        lib().pt_free(data)
    return out


def save_triangles_file(path, triangles):
    triangles = np.ascontiguousarray(triangles, np.float64).reshape(-1, 9)
    check(lib().pt_save_triangles_file(os.fsencode(path), ptr(triangles), len(triangles)))


def construct_triangle_bvh(triangles):
    triangles = np.ascontiguousarray(triangles, np.float64).reshape(-1, 9)
    nodes = np.zeros(max(2 * len(triangles) - 1, 0), NODE_DTYPE)
    check(lib().pt_construct_triangle_bvh(ptr(triangles), len(triangles), ptr(nodes)))
    return nodes


def construct_sphere_bvh(spheres):
    spheres = np.ascontiguousarray(spheres, np.float64).reshape(-1, 4)
    nodes = np.zeros(max(2 * len(spheres) - 1, 0), NODE_DTYPE)
    check(lib().pt_construct_sphere_bvh(ptr(spheres), len(spheres), ptr(nodes)))
    return nodes


def rotation_matrix(angle, ax, ay, az):
    m = PtMat3()
    check(lib().pt_rotation_matrix(angle, ax, ay, az, C.byref(m)))
    return np.array([[m.rows[i][j] for j in range(3)] for i in range(3)])


def camera_look_at(position, target, fov_y, aperture, focus_distance=0.0):
    cam, p, t = PtCamera(), PtVec3(*position), PtVec3(*target)
    check(lib().pt_camera_look_at(C.byref(p), C.byref(t), fov_y, aperture, focus_distance, C.byref(cam)))
    return cam


def camera_frame(camera, aspect_ratio, width, height, **changes):
    frame = PtFrame()
    check(lib().pt_camera_frame(C.byref(camera), float(aspect_ratio), int(width), int(height), C.byref(frame)))
    for key, value in changes.items():
        setattr(frame, key, value)
    return frame


def camera_apply_input(camera, viewer_input):
    """pt_camera_apply_input: updates `camera` in place; returns True if it changed."""
    modified = C.c_int32()
    check(lib().pt_camera_apply_input(C.byref(camera), C.byref(viewer_input), C.byref(modified)))
    return bool(modified.value)


def tonemap_bgra8(accum, sample_count):
    accum = np.ascontiguousarray(accum, np.float64)
    out = np.zeros(accum.size, np.uint8)
    check(lib().pt_tonemap_bgra8(ptr(accum), accum.size // 4, int(sample_count), ptr(out)))
    return out


def tonemap_device(device, accum_ptr, pixel_count, sample_count, out_ptr, stream=None):
    """GPU tone-map: device pointers (ints) of the f64 BGRA+count buffer and of the BGRA8 output."""
    check(lib().pt_tonemap_device(int(device), C.c_void_p(accum_ptr), int(pixel_count), int(sample_count), C.c_void_p(out_ptr),
                                  C.c_void_p(stream or 0)))


def write_pixel_data_file(path, bgra8):
    bgra8 = np.ascontiguousarray(bgra8, np.uint8)
    check(lib().pt_write_pixel_data_file(os.fsencode(path), ptr(bgra8), bgra8.size // 4))


def write_ppm_file(path, bgra8, width, height):
    bgra8 = np.ascontiguousarray(bgra8, np.uint8)
    assert bgra8.size == 4 * width * height
    check(lib().pt_write_ppm_file(os.fsencode(path), ptr(bgra8), int(width), int(height)))


class HostScene:
    """Host-side scene arrays built by pt_host_scene_build."""

    def __init__(self, kind, path=None, param=0):
        handle = C.c_void_p()
        check(lib().pt_host_scene_build(int(kind), None if path is None else os.fsencode(path), int(param), C.byref(handle)))
        self.handle = handle
        self.desc = PtSceneDesc()
        check(lib().pt_host_scene_desc(self.handle, C.byref(self.desc)))
        self.camera = PtCamera()
        aspect = C.c_double()
        check(lib().pt_host_scene_camera(self.handle, C.byref(self.camera), C.byref(aspect)))
        self.aspect_ratio = aspect.value

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:       # at interpreter shutdown the module globals may be gone
            lib().pt_host_scene_destroy(self.handle)
            self.handle = None

    def frame(self, width, height, **changes):
        return camera_frame(self.camera, self.aspect_ratio, width, height, **changes)

    def arrays(self):
        """Copies of the scene arrays as numpy (same keys as the checkers' export())."""
        d = self.desc

        def view(address, count, dtype):
            if not address or count == 0:
                return np.zeros(0, dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(address)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()

        return {
            "materials": view(d.materials, d.material_count, MATERIAL_DTYPE),
            "spheres": view(d.spheres, d.sphere_count * 4, np.float64).reshape(-1, 4),
            "sphere_material_indices": view(d.sphere_material_indices, d.sphere_count, np.int32),
            "sphere_bvh": view(d.sphere_bvh, d.sphere_bvh_node_count, NODE_DTYPE),
            "triangles": view(d.triangles, d.triangle_count * 9, np.float64).reshape(-1, 9),
            "triangle_material_indices": view(d.triangle_material_indices, d.triangle_count, np.int32),
            "triangle_bvh": view(d.triangle_bvh, d.triangle_bvh_node_count, NODE_DTYPE),
            "background": np.array([[d.background_gradient_start.r, d.background_gradient_start.g, d.background_gradient_start.b],
                                    [d.background_gradient_end.r, d.background_gradient_end.g, d.background_gradient_end.b]]),
        }


# ------------------------------------------------------------------ the GPU path
class DeviceScene:
    """A scene resident in the HBM of one GPU (pt_scene_create)."""

    def __init__(self, desc, device=0, keepalive=None):
        handle = C.c_void_p()
        check(lib().pt_scene_create(C.byref(desc), int(device), C.byref(handle)))
        self.handle = handle
        self.device = device
        del keepalive

    @classmethod
    def from_host(cls, host_scene, device=0):
        return cls(host_scene.desc, device)

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:       # at interpreter shutdown the module globals may be gone
            lib().pt_scene_destroy(self.handle)
            self.handle = None

    def render(self, frame, accum=None):
        """pt_render: adds into a host buffer (allocated zeroed if not given)."""
        if accum is None:
            accum = np.zeros(4 * frame.width * frame.height)
        assert accum.dtype == np.float64 and accum.size == 4 * frame.width * frame.height
        check(lib().pt_render(self.handle, C.byref(frame), ptr(accum)))
        return accum

    def render_device(self, frame, accum_device_ptr, stream=None):
        """pt_render_device: accum_device_ptr is a raw HBM address; returns after the launch."""
        check(lib().pt_render_device(self.handle, C.byref(frame), C.c_void_p(accum_device_ptr),
                                     C.c_void_p(stream) if stream else None))

    def viewer_frame(self, camera, viewer_input, aspect_ratio, width, height, sample, accum_device_ptr, samples_per_frame=1, stream=None):
        """pt_viewer_frame: one iteration of the reference's frame loop; returns the advanced sample counter."""
        counter = C.c_int32(int(sample))
        check(lib().pt_viewer_frame(self.handle, C.byref(camera), None if viewer_input is None else C.byref(viewer_input), float(aspect_ratio),
                                    int(width), int(height), int(samples_per_frame), C.byref(counter), C.c_void_p(accum_device_ptr),
                                    C.c_void_p(stream) if stream else None))
        return counter.value

    def render_device_noalloc(self, frame, accum_device_ptr, stream=None):
        """pt_render_device_noalloc: same as render_device, without any device allocation (bounded frame size)."""
        check(lib().pt_render_device_noalloc(self.handle, C.byref(frame), C.c_void_p(accum_device_ptr),
                                             C.c_void_p(stream) if stream else None))

    def last_render_ms(self):
        ms = C.c_float()
        check(lib().pt_last_render_ms(self.handle, C.byref(ms)))
        return ms.value

    def set_workspace_limit(self, nbytes):
        """Cap of the device memory the handle keeps for its per-path colour records."""
        check(lib().pt_scene_set_workspace_limit(self.handle, int(nbytes)))

    def last_launch_shape(self):
        """(persistent blocks per compute unit = waves per SIMD, compute units, bytes of colour workspace held)."""
        blocks, units, held = C.c_int32(), C.c_int32(), C.c_uint64()
        check(lib().pt_last_launch_shape(self.handle, C.byref(blocks), C.byref(units), C.byref(held)))
        return blocks.value, units.value, held.value

    def last_render_stats(self):
        out = np.zeros(16, np.uint64)
        check(lib().pt_last_render_stats(self.handle, ptr(out)))
        names = ["node_phases", "node_lanes", "leaf_phases", "leaf_lanes", "shade_phases", "shade_lanes", "loops", "paths",
                 "unsafe_skips", "t_node", "t_leaf", "t_shade", "t_refill", "t_setup", "t_total"]
        return dict(zip(names, out.tolist()))

    def primary_hits(self, frame):
        tri = np.full((frame.height, frame.width), -2, np.int32)
        sph = np.full((frame.height, frame.width), -2, np.int32)
        check(lib().pt_primary_hits(self.handle, C.byref(frame), ptr(tri), ptr(sph)))
        return tri, sph


_EVAL_SHAPES = {  # op: (input dtype, input width, output dtype, output width)
    abi.PT_EVAL_NOISE_1D: (np.int32, 1, np.uint32, 1),
    abi.PT_EVAL_NOISE_3D: (np.int32, 3, np.uint32, 1),
    abi.PT_EVAL_XORSHIFT_REAL: (np.uint32, 1, np.float64, 2),
    abi.PT_EVAL_RANDOM_UNIT_VECTOR: (np.float64, 3, np.float64, 3),
    abi.PT_EVAL_RAY_AABB: (np.float64, 12, np.float64, 2),
    abi.PT_EVAL_RAY_SPHERE: (np.float64, 10, np.float64, 3),
    abi.PT_EVAL_RAY_TRIANGLE: (np.float64, 15, np.float64, 2),
    abi.PT_EVAL_UNIT_NORMAL: (np.float64, 9, np.float64, 3),
    abi.PT_EVAL_SCATTER: (np.float64, 17, np.float64, 7),
    abi.PT_EVAL_SQRT_DIV: (np.float64, 2, np.float64, 2),
}


def device_eval(op, records, device=0):
    """Runs one device building block over packed records (see PT_EVAL_* in the header)."""
    in_dtype, in_width, out_dtype, out_width = _EVAL_SHAPES[op]
    records = np.ascontiguousarray(records, in_dtype).reshape(-1, in_width)
    out = np.zeros((len(records), out_width), out_dtype)
    check(lib().pt_device_eval(int(device), int(op), ptr(records), len(records), ptr(out)))
    return out
