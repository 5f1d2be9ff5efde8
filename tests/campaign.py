"""Seeded random inputs shared by the GPU-versus-checker campaigns (tests/test_gpu_edges.py) and by the campaign that runs
the same inputs through the reference's own build and the CPU restatement (tests/test_reference_campaign.py)."""
import numpy as np

from pathtracer_amd import ctypes_abi as abi

SKY = np.array([[1.0, 1.0, 1.0], [0.5, 0.7, 1.0]])


def materials():
    m = np.zeros(4, abi.MATERIAL_DTYPE)
    m[0]["v"] = [0.8, 0.6, 0.4, 0.0]                                   # lambertian
    m[1]["v"] = [0.9, 0.9, 0.8, 0.2]; m[1]["type"] = abi.PT_METAL
    m[2]["v"] = [1.5, 0.0, 0.0, 0.0]; m[2]["type"] = abi.PT_DIELECTRIC
    m[3]["v"] = [1.0, 0.9, 0.8, 5.0]; m[3]["type"] = abi.PT_DIFFUSE_LIGHT
    return m


def scene_arrays(api, triangles=None, spheres=None, tri_materials=None, sph_materials=None):
    """Scene arrays with the library's own host-side trees (pt_construct_*_bvh) over the given primitives."""
    triangles = np.zeros((0, 9)) if triangles is None else np.asarray(triangles, float).reshape(-1, 9)
    spheres = np.zeros((0, 4)) if spheres is None else np.asarray(spheres, float).reshape(-1, 4)
    arrays = {"materials": materials(), "triangles": triangles, "spheres": spheres, "background": SKY,
              "triangle_material_indices": np.asarray(tri_materials if tri_materials is not None else np.arange(len(triangles)) % 4, np.int32),
              "sphere_material_indices": np.asarray(sph_materials if sph_materials is not None else np.arange(len(spheres)) % 3, np.int32)}
    if len(triangles):
        arrays["triangle_bvh"] = api.construct_triangle_bvh(triangles)
    if len(spheres):
        arrays["sphere_bvh"] = api.construct_sphere_bvh(spheres)
    return arrays


def random_scene(api, seed):
    """Scene `seed` of the random-scene campaign: triangle soups of 1..400 triangles (30,000 from seed 16 on; some on an
    integer lattice, so that coplanar and duplicated triangles tie), 0..40 spheres (hollow ones in every fourth scene), all
    four materials, cameras inside and outside the geometry, odd image sizes.  Returns the scene arrays and the keyword
    arguments of one comparison: eye, target, size, spp, bounces, aperture.  `unbounded` tells whether infinite or huge
    coordinates or arbitrary material parameters were mixed in (inputs on which the reference's own asserts may stop it)."""
    rng = np.random.default_rng(1000 + seed)
    count = int(rng.choice([1, 2, 7, 33, 150, 400] if seed < 16 else [1, 2, 7, 33, 150, 400, 3000, 30000]))
    if seed % 3 == 0:
        base = rng.integers(-6, 7, (count, 1, 3)).astype(float)
        triangles = (base + rng.integers(0, 3, (count, 3, 3))).reshape(-1, 9)          # lattice: exact ties and degenerate triangles
        triangles = np.concatenate([triangles, triangles[: max(1, count // 5)]])            # and exact duplicates
    else:
        centre = rng.uniform(-8, 8, (count, 1, 3))
        triangles = (centre + rng.normal(0, rng.uniform(0.2, 3.0), (count, 3, 3))).reshape(-1, 9)
    spheres = None
    if seed % 2 == 1:
        n = int(rng.choice([1, 2, 5, 40]))
        spheres = np.concatenate([rng.uniform(-8, 8, (n, 3)), rng.uniform(0.3, 2.5, (n, 1))], 1)
        if seed % 4 == 3:
            spheres[::2, 3] *= -1.0                                                    # hollow: inverted boxes
        if seed % 6 == 3 and seed >= 16:
            spheres = np.concatenate([rng.integers(-6, 7, (n, 3)), rng.integers(1, 3, (n, 1))], 1).astype(float)   # on the triangles' lattice
            spheres = np.concatenate([spheres, spheres[: max(1, n // 3)]])                                         # and duplicated: tied spheres
    unbounded = False
    if seed % 11 == 5:                                                                 # unbounded and huge coordinates
        odd = np.array([np.inf, -np.inf, 1e300, -1e300, 1e160, 3e6])
        hit = rng.random(triangles.shape) < 0.01
        triangles[hit] = rng.choice(odd, hit.sum())
        if spheres is not None:
            spheres[rng.integers(0, len(spheres)), rng.integers(0, 4)] = rng.choice(odd)
        unbounded = True
    arrays = scene_arrays(api, triangles, spheres, tri_materials=rng.integers(0, 4, len(triangles)),
                          sph_materials=None if spheres is None else rng.integers(0, 4, len(spheres)))
    if seed % 7 == 6:                                                                  # any numbers as material parameters
        values = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1.5, 2.0, 1e-300, 1e300, np.inf, -np.inf, np.nan, 0.999, 1.0000001, 7.0])
        changed = arrays["materials"].copy()
        for m in changed:
            m["v"] = np.where(rng.random(4) < 0.5, rng.choice(values, 4), rng.uniform(-0.2, 1.6, 4))
        arrays["materials"] = changed
        unbounded = True
    eye = tuple(rng.uniform(-4, 4, 3)) if seed % 5 == 0 else tuple(rng.uniform(12, 25, 3) * rng.choice([-1, 1], 3))
    if seed % 12 == 9:
        eye = tuple(float(v) for v in rng.integers(8, 20, 3) * rng.choice([-1, 1], 3))   # a lattice point: with aperture 0 every ray starts there
    size = (int(rng.integers(9, 70)), int(rng.integers(9, 50)))
    how = dict(eye=eye, target=(0.0, 0.0, 0.0), size=size, spp=int(rng.integers(1, 4)), bounces=int(rng.choice([1, 4, 12, 50])),
               aperture=float(rng.choice([0.0, 0.05, 0.6])))
    return arrays, how, unbounded


def random_frame(api, seed):
    """Case `seed` of the random-frame campaign: a small random scene and a frame with an odd size, a row range that starts
    anywhere, shards, a sample range that starts anywhere and spans several groups of 16, any depth, and the numbers the
    buffer already holds.  Returns (arrays, frame, held, rng, unbounded); `rng` is left where the GPU test draws on."""
    rng = np.random.default_rng(5000 + seed)
    count = int(rng.choice([3, 40, 300]))
    triangles = (rng.uniform(-7, 7, (count, 1, 3)) + rng.normal(0, rng.uniform(0.3, 2.5), (count, 3, 3))).reshape(-1, 9)
    n = int(rng.choice([0, 1, 6]))
    spheres = np.concatenate([rng.uniform(-7, 7, (n, 3)), rng.uniform(0.3, 2.0, (n, 1))], 1) if n else None
    arrays = scene_arrays(api, triangles, spheres, tri_materials=rng.integers(0, 4, count), sph_materials=None if spheres is None else rng.integers(0, 4, n))
    width, height = int(rng.integers(2, 150)), int(rng.integers(2, 90))
    cam = api.camera_look_at(tuple(rng.uniform(10, 24, 3) * rng.choice([-1, 1], 3)), (0.0, 0.0, 0.0), 0.7, float(rng.choice([0.0, 0.1])))
    rows = sorted(int(v) for v in rng.integers(0, height + 1, 2))
    if rows[0] == rows[1]:
        rows = [0, height]
    shards = int(rng.choice([1, 1, 2, 3, 5]))
    frame = api.camera_frame(cam, width / height, width, height, row_begin=rows[0], row_end=rows[1],
                             sample_begin=int(rng.integers(0, 40)), sample_count=int(rng.choice([1, 5, 16, 17, 33, 50])),
                             max_bounces=int(rng.choice([0, 1, 3, 12, 50])), shard_count=shards, shard_index=int(rng.integers(0, shards)))
    unbounded = False
    if seed % 13 == 4:                                   # any numbers in the camera: the rays may be NaN, infinite, parallel to an axis
        values = np.array([0.0, -0.0, 1.0, -1.0, 1e-300, 1e160, 1e300, -1e300, np.inf, -np.inf, np.nan])
        for name in ("camera_position", "camera_x", "camera_y", "bottom_left", "step_x", "step_y"):
            vector = getattr(frame, name)
            for axis in "xyz":
                if rng.random() < 0.25:
                    setattr(vector, axis, float(rng.choice(values)))
        if rng.random() < 0.5:
            frame.aperture = float(rng.choice([-1.0, 0.0, 1e300, np.inf, np.nan, 3.0]))
        unbounded = True
    held = rng.uniform(0, 3, 4 * width * height)
    return arrays, frame, held, rng, unbounded
