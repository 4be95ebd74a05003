"""Randomised differential test: small random scenes (triangle soups and closed meshes, random
instance transforms, random per-instance materials, random cameras, random BVH build
parameters) rendered by both GPU kernels and by the CPU oracle.  Everything must agree word
for word: hit buffers, the reference's debug counters, accumulators, run totals."""
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import (Material, RT_MAT_DIFFUSE, RT_MAT_EMISSIVE, RT_MAT_GENERIC, RT_MAT_METAL, RT_MAT_NORMAL,
                             RT_MAT_REFRACTIVE, RT_MAT_UV)
from ray_tracing_amd import scene as S
from helpers import assert_bits_equal, gpu_render

pytestmark = pytest.mark.gpu
F = np.float32


REFERENCE_KINDS = [RT_MAT_GENERIC, RT_MAT_NORMAL, RT_MAT_EMISSIVE]  # the three the reference selects
ALL_KINDS = REFERENCE_KINDS + [RT_MAT_REFRACTIVE, RT_MAT_REFRACTIVE, RT_MAT_UV, RT_MAT_DIFFUSE, RT_MAT_METAL]


def random_material(rng, kinds=REFERENCE_KINDS):
    m = Material()
    m.kind = int(rng.choice(kinds))
    m.backface_culling = int(rng.integers(0, 2))
    m.color[:] = [float(x) for x in rng.uniform(0.1, 3.0, 3)]
    m.specular = float(rng.choice([0.0, 0.0, rng.uniform(0.0, 1.0)]))
    m.specular_roughness = float(rng.uniform(0.0, 0.5))
    m.roughness = float(rng.uniform(0.0, 1.0))
    m.ior = float(rng.choice([0.0, 1.33, 1.5, 2.4]))
    return m


def random_mesh(rng, big=False):
    kind = rng.integers(0, 4 if big else 3)
    if kind == 3:  # axis-aligned rectangles: flat leaf boxes, rays along box faces, many exact ties
        n = int(rng.integers(4, 120))
        verts, idx = [], []
        for q in range(n):
            axis = int(rng.integers(0, 3))
            c = np.round(rng.uniform(-2.0, 2.0, 3) * 4.0) / 4.0
            e = np.round(rng.uniform(0.25, 1.5, 2) * 4.0) / 4.0
            u, v = [a for a in range(3) if a != axis]
            corners = []
            for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = c.copy()
                p[u] += du * e[0]
                p[v] += dv * e[1]
                corners.append(p)
            b = 4 * q
            verts += corners
            idx += [(b, b + 1, b + 2), (b, b + 2, b + 3)]
        return np.asarray(verts, F), np.asarray(idx, np.uint32)
    if kind == 0:  # soup of random triangles in a box, some of them degenerate or duplicated
        n = int(rng.integers(1, 600 if big else 60))
        base = rng.uniform(-1.0, 1.0, (n, 1, 3))
        verts = (base + rng.uniform(-0.6, 0.6, (n, 3, 3))).reshape(-1, 3)
        idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        if n > 4:
            idx[1] = idx[0]                 # the same triangle twice: equal distance, later one wins
            if rng.integers(0, 2):
                verts[3 * 2 + 2] = verts[3 * 2]  # a zero-area triangle: its distance is NaN and is recorded
        return verts.astype(F), idx
    if kind == 1:
        v, f = S.icosphere(int(rng.integers(0, 3)))
        return np.asarray(v, F), np.asarray(f, np.uint32)
    v, f = S.cube_mesh(inward=bool(rng.integers(0, 2)))
    return np.asarray(v, F), np.asarray(f, np.uint32)


def random_scene(rng, w, h, big=False, unit=1.0, kinds=REFERENCE_KINDS, builder="host"):
    """`unit` rescales object space (vertices by unit, instance scale by 1/unit): the same world-space
    scene, with object-space coordinates inside or outside the ranges the kernels' shortcuts accept."""
    b = S.SceneBuilder()
    n_meshes = int(rng.integers(1, 6 if big else 4))
    n_instances = 0
    for _ in range(n_meshes):
        verts, idx = random_mesh(rng, big)
        verts = (verts * F(unit)).astype(F)
        transforms = []
        for _ in range(int(rng.integers(1, 3))):
            axis, angle = "xyz"[int(rng.integers(0, 3))], float(rng.uniform(-math.pi, math.pi))
            scale = [float(F(x) / F(unit)) for x in rng.uniform(0.5, 2.5, 3)]
            if rng.integers(0, 4) == 0:
                scale[0] = -scale[0]  # a mirroring instance
            transforms.append(S.affine_from_srt(scale, S.quat_from_rotation(axis, angle),
                                                [float(x) for x in rng.uniform(-3.0, 3.0, 3)]))
        params = S.bvh_params(leaf_max=int(rng.integers(1, 70 if big else 9)), root_leaf_max=int(rng.choice([0, 5, 100])),
                              max_depth=int(rng.choice([3, 32])), threads=1)
        b.add_mesh(verts, idx, transforms, params, builder=builder)
        n_instances += len(transforms)
    scene = b.build()
    eye = rng.uniform(-1.0, 1.0, 3)
    eye = eye / np.linalg.norm(eye) * rng.uniform(4.0, 9.0)
    cam = S.make_cam(w, h, S.look_at(tuple(eye), tuple(rng.uniform(-1.0, 1.0, 3))), fov=float(rng.uniform(40.0, 110.0)),
                     depth=int(rng.integers(1, 11 if big else 7)), frame=int(rng.integers(0, 1000)))
    return scene, cam, [random_material(rng, kinds) for _ in range(n_instances)]


def compare(scene, cam, materials):
    # Three chained frames, unless a NaN reaches the accumulator in any of them: the seed of the
    # next frame hashes the accumulator's bits (lib.rs:57-60), and the sign and payload of a NaN
    # differ between x86 and the GPU.  A NaN stays in an accumulator once it is there, so looking
    # at the last frame is enough.  Then only the first frame is compared.
    ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, materials=materials).render(3)
    frames = 3
    if bool(np.isnan(ref.acc).any() or np.isnan(ref.hit_t_bits.view(F)).any()):
        frames = 1
        ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, materials=materials).render(1)
    for pipeline in (1, 0, 2):  # persistent, plain, pooled persistent
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


@pytest.mark.parametrize("seed", range(64))
def test_random_scene_matches_the_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(5, 70)), int(rng.integers(5, 50))
    compare(*random_scene(rng, w, h))


@pytest.mark.parametrize("seed", range(24))
def test_larger_random_scene_matches_the_oracle(seed):
    """More triangles, leaves of up to 69 triangles, axis-aligned geometry (flat boxes, exact
    ties), depth up to 10, and object spaces rescaled by 2^-30 .. 2^30."""
    rng = np.random.default_rng(5000 + seed)
    unit = float(F(2.0) ** int(rng.choice([0, 0, 0, -30, -12, 12, 26, 30])))
    compare(*random_scene(rng, int(rng.integers(40, 130)), int(rng.integers(30, 90)), big=True, unit=unit))


@pytest.mark.parametrize("seed", range(32))
def test_random_scene_with_every_material_kind_matches_the_oracle(seed):
    """As above, with the materials the reference implements but never selects in the draw."""
    rng = np.random.default_rng(9000 + seed)
    big = seed % 4 == 3
    w, h = (int(rng.integers(40, 100)), int(rng.integers(30, 70))) if big else (int(rng.integers(5, 70)), int(rng.integers(5, 50)))
    compare(*random_scene(rng, w, h, big=big, kinds=ALL_KINDS))


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_FUZZ_EXTRA", "0"))))
def test_extra_random_scenes_on_request(seed):
    """A soak run: RT_FUZZ_EXTRA=n adds n more scenes (all material kinds, every fourth one large,
    the trees by the host builder and the two GPU builders in turn); RT_FUZZ_BASE=m moves on to
    scenes no earlier soak has seen."""
    rng = np.random.default_rng(20000 + int(os.environ.get("RT_FUZZ_BASE", "0")) + seed)
    big = seed % 4 == 3
    builder = ("host", "gpu_sah", "gpu_linear")[(seed // 4) % 3]
    w, h = (int(rng.integers(40, 100)), int(rng.integers(30, 70))) if big else (int(rng.integers(5, 70)), int(rng.integers(5, 50)))
    unit = float(F(2.0) ** int(rng.choice([0, 0, 0, -30, -12, 12, 26, 30]))) if big else 1.0
    compare(*random_scene(rng, w, h, big=big, unit=unit, kinds=ALL_KINDS, builder=builder))


@pytest.mark.parametrize("builder", ["gpu_sah", "gpu_linear"])
@pytest.mark.parametrize("seed", range(12))
def test_random_scene_on_a_gpu_built_tree_matches_the_oracle(seed, builder):
    """The same random scenes with every mesh's tree built on the GPU: duplicated and zero-area
    triangles, coincident centroids, flat boxes and rescaled object spaces go through the builders,
    and the kernels must agree with the oracle on whatever tree comes out."""
    rng = np.random.default_rng(7000 + seed)
    unit = float(F(2.0) ** int(rng.choice([0, 0, -30, 12, 30])))
    compare(*random_scene(rng, int(rng.integers(40, 110)), int(rng.integers(30, 80)), big=True, unit=unit, builder=builder))
