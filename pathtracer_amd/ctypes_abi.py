"""ctypes mirror of include/pathtracer_amd.h (records and constants only, no library load).

Each record is layout-identical to the reference struct named in the header
(src/linear_algebra.h, src/geometry.h, src/bvh.h, src/material.h, src/path_tracing.h).
"""
import ctypes as C

import numpy as np

PT_OK = 0
PT_ERR_INVALID_ARGUMENT = 1
PT_ERR_IO = 2
PT_ERR_DEVICE = 3
PT_ERR_OUT_OF_MEMORY = 4
PT_ERR_UNSUPPORTED = 5

PT_ROW_BLOCK = 8
PT_NOALLOC_MAX_TILES = 8192

PT_LAMBERTIAN, PT_METAL, PT_DIELECTRIC, PT_DIFFUSE_LIGHT = 0, 1, 2, 3

PT_SCENE_MODEL = 0
PT_SCENE_CHESS6_ROW = 1
PT_SCENE_CHESS32 = 2
PT_SCENE_CHESS32_MIXED = 3
PT_SCENE_CHESS_GRID = 4
PT_SCENE_RANDOM_SPHERES = 5
PT_SCENE_CORNELL_BOX = 6

(PT_EVAL_NOISE_1D, PT_EVAL_NOISE_3D, PT_EVAL_XORSHIFT_REAL, PT_EVAL_RANDOM_UNIT_VECTOR, PT_EVAL_RAY_AABB,
 PT_EVAL_RAY_SPHERE, PT_EVAL_RAY_TRIANGLE, PT_EVAL_UNIT_NORMAL, PT_EVAL_SCATTER, PT_EVAL_SQRT_DIV) = range(10)


class PtVec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)

    def tuple(self):
        return (self.x, self.y, self.z)


class PtMat3(C.Structure):
    _fields_ = [("rows", (C.c_double * 3) * 3)]


class PtColour(C.Structure):
    _fields_ = [("r", C.c_double), ("g", C.c_double), ("b", C.c_double)]


class PtCamera(C.Structure):
    _fields_ = [("orientation", PtMat3), ("target", PtVec3), ("distance", C.c_double), ("fov_y", C.c_double),
                ("aperture", C.c_double), ("focus_distance", C.c_double)]


class PtViewerInput(C.Structure):
    _fields_ = [("left_button_held", C.c_int32), ("mouse_delta_x", C.c_int32), ("mouse_delta_y", C.c_int32), ("scroll", C.c_int32),
                ("key_a", C.c_int32), ("key_d", C.c_int32), ("key_w", C.c_int32), ("key_s", C.c_int32),
                ("key_q", C.c_int32), ("key_e", C.c_int32), ("key_up", C.c_int32), ("key_down", C.c_int32),
                ("key_left", C.c_int32), ("key_right", C.c_int32), ("frame_duration", C.c_double)]


class PtSceneDesc(C.Structure):
    _fields_ = [
        ("materials", C.c_void_p), ("material_count", C.c_int32),
        ("spheres", C.c_void_p), ("sphere_count", C.c_int32),
        ("sphere_bvh", C.c_void_p), ("sphere_bvh_node_count", C.c_int32),
        ("sphere_material_indices", C.c_void_p),
        ("triangles", C.c_void_p), ("triangle_count", C.c_int32),
        ("triangle_bvh", C.c_void_p), ("triangle_bvh_node_count", C.c_int32),
        ("triangle_material_indices", C.c_void_p),
        ("background_gradient_start", PtColour), ("background_gradient_end", PtColour),
    ]


class PtFrame(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("sample_begin", C.c_int32), ("sample_count", C.c_int32),
        ("max_bounces", C.c_int32), ("reserved_", C.c_int32),
        ("aperture", C.c_double),
        ("camera_position", PtVec3), ("camera_x", PtVec3), ("camera_y", PtVec3),
        ("bottom_left", PtVec3), ("step_x", PtVec3), ("step_y", PtVec3),
        ("shard_index", C.c_int32), ("shard_count", C.c_int32),
    ]

    def copy(self, **changes):
        out = PtFrame()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(PtFrame))
        for key, value in changes.items():
            setattr(out, key, value)
        return out


# numpy views of the array records
TRIANGLE_DTYPE = np.dtype((np.float64, (9,)))                     # PtTriangle, 72 B
SPHERE_DTYPE = np.dtype((np.float64, (4,)))                       # PtSphere, 32 B
NODE_DTYPE = np.dtype([("aabb", np.float64, (6,)), ("index", np.int32), ("left", np.int32),
                       ("right", np.int32), ("pad_", np.int32)])  # PtBvhNode, 64 B
MATERIAL_DTYPE = np.dtype([("v", np.float64, (4,)), ("type", np.int32), ("pad_", np.int32)])  # PtMaterial, 40 B

assert C.sizeof(PtVec3) == 24 and C.sizeof(PtMat3) == 72 and C.sizeof(PtCamera) == 128
assert NODE_DTYPE.itemsize == 64 and MATERIAL_DTYPE.itemsize == 40
assert C.sizeof(PtFrame) == 8 * 4 + 8 + 6 * 24 + 8
assert C.sizeof(PtSceneDesc) == 144 and C.sizeof(PtViewerInput) == 64


def ptr(array):
    """Raw address of a C-contiguous numpy array (None -> null)."""
    if array is None:
        return None
    assert array.flags["C_CONTIGUOUS"]
    return array.ctypes.data_as(C.c_void_p)
