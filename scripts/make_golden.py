#!/usr/bin/env python3
"""Writes tests/golden/*.json: SHA-256 digests of what the CPU oracle produces
for the reference's default scene (config 1).  The reference itself cannot be
run here (it is Rust and only ever executes as a SPIR-V shader), so these are
regression pins of the oracle, not outputs of the reference."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from ray_tracing_amd import default_scene  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    scene, cam = default_scene(640, 360)
    out = {"scene": {"triangles": digest(scene.triangles), "bvh": digest(scene.bvh),
                     "vertices": int(len(scene.vertices)), "n_triangles": int(scene.n_triangles),
                     "n_bvh": int(len(scene.bvh))}}
    r = O.OracleRenderer(scene, cam, trig=O.TRIG_DET).render(1)
    out["frame0"] = {"hit_triangle": digest(r.hit_triangle), "hit_instance": digest(r.hit_instance),
                     "hit_t_bits": digest(r.hit_t_bits), "box_tests": digest(r.box_tests),
                     "triangle_tests": digest(r.triangle_tests), "acc": digest(r.acc)}
    out["stats_frame0"] = r.stats_dict()
    r.render(3)
    out["acc_after_4_frames"] = digest(r.acc)
    path = os.path.join(ROOT, "tests", "golden", "c1_640x360.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
