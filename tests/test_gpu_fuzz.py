"""Randomised differential test: small random scenes (triangle soups and closed meshes, random
instance transforms, random per-instance materials, random cameras, random BVH build
parameters) rendered by both GPU kernels and by the CPU oracle.  Everything must agree word
for word: hit buffers, the reference's debug counters, accumulators, run totals."""
import math

import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import (Material, RT_MAT_EMISSIVE, RT_MAT_GENERIC, RT_MAT_NORMAL)
from ray_tracing_amd import scene as S
from helpers import assert_bits_equal, gpu_render

pytestmark = pytest.mark.gpu
F = np.float32


def random_material(rng):
    m = Material()
    m.kind = int(rng.choice([RT_MAT_GENERIC, RT_MAT_NORMAL, RT_MAT_EMISSIVE]))
    m.backface_culling = int(rng.integers(0, 2))
    m.color[:] = [float(x) for x in rng.uniform(0.1, 3.0, 3)]
    m.specular = float(rng.choice([0.0, 0.0, rng.uniform(0.0, 1.0)]))
    m.specular_roughness = float(rng.uniform(0.0, 0.5))
    m.roughness = float(rng.uniform(0.0, 1.0))
    m.ior = float(rng.choice([0.0, 1.33, 1.5, 2.4]))
    return m


def random_mesh(rng):
    kind = rng.integers(0, 3)
    if kind == 0:  # soup of random triangles in a box, some of them degenerate or duplicated
        n = int(rng.integers(1, 60))
        base = rng.uniform(-1.0, 1.0, (n, 1, 3))
        verts = (base + rng.uniform(-0.6, 0.6, (n, 3, 3))).reshape(-1, 3)
        idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        if n > 4:
            idx[1] = idx[0]                 # the same triangle twice: equal distance, later one wins
            verts[3 * 2 + 2] = verts[3 * 2]  # a zero-area triangle
        return verts.astype(F), idx
    if kind == 1:
        v, f = S.icosphere(int(rng.integers(0, 3)))
        return np.asarray(v, F), np.asarray(f, np.uint32)
    v, f = S.cube_mesh(inward=bool(rng.integers(0, 2)))
    return np.asarray(v, F), np.asarray(f, np.uint32)


def random_scene(rng, w, h):
    b = S.SceneBuilder()
    n_meshes = int(rng.integers(1, 4))
    n_instances = 0
    for _ in range(n_meshes):
        verts, idx = random_mesh(rng)
        transforms = []
        for _ in range(int(rng.integers(1, 3))):
            axis, angle = "xyz"[int(rng.integers(0, 3))], float(rng.uniform(-math.pi, math.pi))
            scale = [float(x) for x in rng.uniform(0.5, 2.5, 3)]
            if rng.integers(0, 4) == 0:
                scale[0] = -scale[0]  # a mirroring instance
            transforms.append(S.affine_from_srt(scale, S.quat_from_rotation(axis, angle),
                                                [float(x) for x in rng.uniform(-3.0, 3.0, 3)]))
        params = S.bvh_params(leaf_max=int(rng.integers(1, 9)), root_leaf_max=int(rng.choice([0, 5, 100])),
                              max_depth=int(rng.choice([3, 32])), threads=1)
        b.add_mesh(verts, idx, transforms, params)
        n_instances += len(transforms)
    scene = b.build()
    eye = rng.uniform(-1.0, 1.0, 3)
    eye = eye / np.linalg.norm(eye) * rng.uniform(4.0, 9.0)
    cam = S.make_cam(w, h, S.look_at(tuple(eye), tuple(rng.uniform(-1.0, 1.0, 3))), fov=float(rng.uniform(40.0, 110.0)),
                     depth=int(rng.integers(1, 7)), frame=int(rng.integers(0, 1000)))
    return scene, cam, [random_material(rng) for _ in range(n_instances)]


@pytest.mark.parametrize("seed", range(16))
def test_random_scene_matches_the_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(5, 70)), int(rng.integers(5, 50))
    scene, cam, materials = random_scene(rng, w, h)
    # one frame when a NaN can reach the accumulator (its payload would seed the next frame)
    ref1 = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, materials=materials).render(1)
    has_nan = bool(np.isnan(ref1.acc).any() or np.isnan(ref1.hit_t_bits.view(F)).any())
    frames = 1 if has_nan else 3
    ref = ref1 if frames == 1 else O.OracleRenderer(scene, cam, trig=O.TRIG_DET, materials=materials).render(frames)
    for pipeline in (1, 0):
        g = gpu_render(scene, cam, frames, materials=materials, pipeline=pipeline)
        tri, inst, tb = g["hits"]
        assert_bits_equal(tri, ref.hit_triangle, f"triangle_id (pipeline {pipeline})")
        assert_bits_equal(inst, ref.hit_instance, "instance_id")
        assert_bits_equal(tb, ref.hit_t_bits, "t bits", nan_equal=True)
        box, tt = g["counters"]
        assert_bits_equal(box, ref.box_tests, "box_tests")
        assert_bits_equal(tt, ref.triangle_tests, "triangle_tests")
        assert_bits_equal(g["acc"], ref.acc, "accumulator", nan_equal=True)
        assert g["stats"] == ref.stats_dict()
