#!/usr/bin/env python3
"""One-off parity campaign on large random triangle soups (the sizes at which several host threads share one range of the
hierarchy builders): primary hit indices and accumulation buffers against the CPU checker, bit for bit.
usage: python profiles/random_big_scenes.py [seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pathtracer_amd import api
from oracle.bindings import OracleLib
import test_gpu_edges as edges

orc = OracleLib()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rng = np.random.default_rng(9000 + seed)
    count = int(rng.choice([70000, 300000, 800000]))
    if seed % 2 == 0:
        base = rng.integers(-40, 41, (count, 1, 3)).astype(float)
        triangles = (base + rng.integers(0, 3, (count, 3, 3))).reshape(-1, 9)       # lattice: ties everywhere
    else:
        triangles = (rng.uniform(-60, 60, (count, 1, 3)) + rng.normal(0, rng.uniform(0.2, 2.0), (count, 3, 3))).reshape(-1, 9)
    spheres = np.concatenate([rng.uniform(-60, 60, (50, 3)), rng.uniform(0.5, 6.0, (50, 1))], 1) if seed % 3 == 0 else None
    t0 = time.time()
    arrays = edges._scene(api, triangles, spheres, tri_materials=rng.integers(0, 4, count),
                          sph_materials=None if spheres is None else rng.integers(0, 4, 50))
    tri, sph = edges._compare(api, orc, arrays, tuple(rng.uniform(70, 120, 3) * rng.choice([-1, 1], 3)), (0.0, 0.0, 0.0), size=(96, 64), spp=2, bounces=12)
    print(f"seed {seed}: {count} triangles{' + 50 spheres' if spheres is not None else ''}: {(tri >= 0).mean():.2f} of the pixels hit, identical "
          f"({time.time() - t0:.1f} s)", flush=True)
