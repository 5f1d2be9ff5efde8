#!/usr/bin/env python3
"""One-off campaign on the device's building blocks (box, sphere and triangle tests, unit normal, random unit vector,
scatter of all four materials): millions of random inputs over many magnitudes, with the awkward ones mixed in (axis-parallel
rays, origins on box planes, rays aimed at edges and vertices, degenerate triangles, zero and denormal lengths, infinities and
NaNs), every output double against the CPU checker's, bit for bit.
usage: python profiles/random_primitives.py [rows per op] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import api, ctypes_abi as abi
from oracle.bindings import OracleLib

orc = OracleLib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4


nan_only = {}


def same(a, b, what="", inputs=None):
    """Bit for bit, except that a NaN may differ from a NaN in sign and payload (counted in `nan_only`: the reference only
    ever compares such a value, and every comparison with any NaN is false)."""
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    differ = a.view(np.uint64) != b.view(np.uint64)
    both_nan = np.isnan(a) & np.isnan(b)
    nan_only[what] = nan_only.get(what, 0) + int((differ & both_nan).sum())
    real = differ & ~both_nan
    if real.any():
        rows = np.unique(np.nonzero(real)[0])[:5]
        for r in rows:
            print(what, "row", r, "device", a[r].tolist(), "checker", b[r].tolist(), "inputs", None if inputs is None else inputs[r].tolist(), flush=True)
        return False
    return True


def unit_rows(rng, rows):
    v = rng.normal(0, 1, (rows, 3))
    return v / np.sqrt((v * v).sum(1))[:, None]


def spoil(rng, a, share=0.002):
    """Scatters zeros, denormals, huge values, infinities and NaNs over a copy of `a`."""
    a = a.copy()
    odd = np.array([0.0, -0.0, 5e-324, 1e-310, 1e-160, 1e160, 1e300, np.inf, -np.inf, np.nan, 1.0, -1.0])
    hit = rng.random(a.shape) < share
    a[hit] = rng.choice(odd, hit.sum())
    return a


for round_ in range(rounds):
    rng = np.random.default_rng(777 + round_)
    scale = 10.0 ** rng.uniform(-3, 5)                      # scene size of this round
    o = rng.uniform(-50, 50, (n, 3)) * scale
    d = unit_rows(rng, n)
    axis = rng.random(n) < 0.05
    d[axis] = np.eye(3)[rng.integers(0, 3, axis.sum())] * rng.choice([-1.0, 1.0], (axis.sum(), 1))
    planar = rng.random(n) < 0.05                            # one zero component
    d[planar, rng.integers(0, 3, planar.sum())] = 0.0
    rays = spoil(rng, np.concatenate([o, d], 1))
    lo = rng.uniform(-60, 40, (n, 3)) * scale
    boxes = np.concatenate([lo, lo + rng.uniform(0.0, 40.0, (n, 3)) * scale * rng.choice([0.0, 1e-9, 1.0], (n, 1))], 1)
    on_plane = rng.random(n) < 0.05
    k = rng.integers(0, 6, n)
    boxes[on_plane, k[on_plane]] = rays[on_plane, k[on_plane] % 3]
    boxes = spoil(rng, boxes)
    assert same(api.device_eval(abi.PT_EVAL_RAY_AABB, np.concatenate([rays, boxes], 1)), orc.ray_aabb(rays, boxes), "box", np.concatenate([rays, boxes], 1)), "box"
    spheres = np.concatenate([rng.uniform(-60, 60, (n, 3)) * scale, rng.uniform(-30, 30, (n, 1)) * scale], 1)
    sure = rng.random(n) < 0.5
    spheres[sure, :3] = rays[sure, :3] + rays[sure, 3:] * rng.uniform(-20, 60, (sure.sum(), 1)) * scale
    graze = rng.random(n) < 0.1                              # tangent rays
    spheres[graze, :3] = rays[graze, :3] + rays[graze, 3:] * 30 * scale + np.cross(rays[graze, 3:], unit_rows(rng, graze.sum())) * np.abs(spheres[graze, 3:4])
    spheres = spoil(rng, spheres)
    assert same(api.device_eval(abi.PT_EVAL_RAY_SPHERE, np.concatenate([rays, spheres], 1)), orc.ray_sphere(rays, spheres), "sphere", np.concatenate([rays, spheres], 1)), "sphere"
    a = rng.uniform(-60, 60, (n, 3)) * scale
    spread = scale * 10.0 ** rng.uniform(-6, 1.5, (n, 1))
    tris = np.concatenate([a, a + rng.normal(0, 1, (n, 3)) * spread, a + rng.normal(0, 1, (n, 3)) * spread], 1)
    lattice = rng.random(n) < 0.2
    tris[lattice] = np.round(tris[lattice] / scale) * scale
    w = rng.dirichlet([1, 1, 1], n)
    edge = rng.random(n) < 0.1
    w[edge] = np.array([0.5, 0.5, 0.0])
    corner = rng.random(n) < 0.1
    w[corner] = np.array([1.0, 0.0, 0.0])
    target = (tris.reshape(n, 3, 3) * w[:, :, None]).sum(1)
    aimed = target - rays[:, :3]
    with np.errstate(all="ignore"):
        aimed /= np.sqrt((aimed * aimed).sum(1))[:, None]
    tri_rays = rays.copy()
    aim = rng.random(n) < 0.75
    tri_rays[aim, 3:] = aimed[aim]
    flat = rng.random(n) < 0.02
    tris[flat, 6:9] = tris[flat, 3:6]
    tris = spoil(rng, tris)
    assert same(api.device_eval(abi.PT_EVAL_RAY_TRIANGLE, np.concatenate([tri_rays, tris], 1)), orc.ray_triangle(tri_rays, tris), "triangle", np.concatenate([tri_rays, tris], 1)), "triangle"
    assert same(api.device_eval(abi.PT_EVAL_UNIT_NORMAL, tris), orc.unit_normal(tris), "unit normal", tris), "unit normal"
    seeds = spoil(rng, np.where(rng.random((n, 1)) < 0.5, unit_rows(rng, n), rng.uniform(-3, 3, (n, 3)) * 10.0 ** rng.uniform(-12, 2, (n, 1))))
    assert same(api.device_eval(abi.PT_EVAL_RANDOM_UNIT_VECTOR, seeds), orc.random_unit_vector(seeds), "random unit vector", seeds), "random unit vector"
    normals = unit_rows(rng, n)
    incoming = unit_rows(rng, n)
    grazing = rng.random(n) < 0.1
    incoming[grazing] = np.cross(normals[grazing], unit_rows(rng, grazing.sum())) + normals[grazing] * rng.normal(0, 1e-9, (grazing.sum(), 1))
    mats = np.zeros((n, 5))
    kinds = rng.integers(0, 4, n)
    mats[:, 4] = kinds
    mats[:, :3] = rng.uniform(0, 1, (n, 3))
    mats[:, 3] = np.where(kinds == 1, rng.uniform(0, 1, n), rng.uniform(1, 20, n))
    mats[kinds == 2, 0] = rng.choice([1.5, 1.33, 2.4, 1.0, 0.7, 1.0000001], (kinds == 2).sum())
    points = rng.uniform(-50, 50, (n, 3)) * scale
    srays = np.concatenate([rng.uniform(-50, 50, (n, 3)) * scale, spoil(rng, incoming)], 1)
    normals = spoil(rng, normals)
    records = np.concatenate([srays, mats, points, normals], 1)
    assert same(api.device_eval(abi.PT_EVAL_SCATTER, records), orc.scatter(srays, mats, points, normals), "scatter", records), "scatter"
    print(f"round {round_}: scale {scale:.3g}, {n} rows per building block: box, sphere, triangle, unit normal, random unit vector, scatter identical; NaNs of another sign or payload so far: {nan_only}", flush=True)
