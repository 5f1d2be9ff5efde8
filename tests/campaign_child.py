"""TEST INFRASTRUCTURE - one case of the random campaigns run through the reference's own build (oracle/_ref) and through the
CPU restatement (oracle/libpt_oracle.so); prints one JSON line saying whether every bit agrees.

Run as a child process by tests/test_reference_campaign.py: the reference asserts on some inputs (unit-length rays and
normals, src/geometry.cpp:9, src/path_tracing.cpp:8-37, a hit at +inf :184, :196), and an assert ends the process.

usage: python tests/campaign_child.py scene|frame <seed>
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from campaign import random_frame, random_scene  # noqa: E402
from conftest import same_bits  # noqa: E402
from oracle.bindings import OracleLib, RefLib  # noqa: E402
from pathtracer_amd import api  # noqa: E402


def main():
    kind, seed = sys.argv[1], int(sys.argv[2])
    ref, orc = RefLib(), OracleLib()
    if kind == "scene":
        arrays, how, _ = random_scene(api, seed)
        cam = api.camera_look_at(how["eye"], how["target"], 0.7, how["aperture"])
        w, h = how["size"]
        # the reference's depth is a compile-time 50 (src/path_tracing.cpp:172): both sides render at that depth
        frame = api.camera_frame(cam, w / h, w, h, sample_count=how["spp"], max_bounces=50)
        held = None
    else:
        arrays, frame, held, _, _ = random_frame(api, seed)
        frame.max_bounces = 50
    print(json.dumps({"stage": "built"}), flush=True)
    theirs, ours = ref.scene_create(arrays), orc.scene_create(arrays)
    tri_r, sph_r = theirs.primary_hits(frame)
    print(json.dumps({"stage": "reference hits"}), flush=True)
    tri_o, sph_o = ours.primary_hits(frame)
    hits = bool(np.array_equal(tri_r, tri_o) and np.array_equal(sph_r, sph_o))
    got_r = theirs.render(frame, None if held is None else held.copy(), threads=4)
    print(json.dumps({"stage": "reference render"}), flush=True)
    got_o = ours.render(frame, None if held is None else held.copy(), threads=4)
    print(json.dumps({"stage": "done", "hits_equal": hits, "buffers_equal": bool(same_bits(got_r, got_o)),
                      "pixels": int(frame.width * frame.height), "triangle_hits": int((tri_r >= 0).sum()), "sphere_hits": int((sph_r >= 0).sum())}), flush=True)


if __name__ == "__main__":
    main()
