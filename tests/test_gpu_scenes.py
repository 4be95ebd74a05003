"""GPU parity beyond the default scene: synthetic scenes of every configuration
at sizes the oracle renders in seconds, hand-built BVHs with the edge cases the
traversal rules care about, the material table, the partition and seed options,
and size-independent properties at the full 1080p / 1M-triangle size.

Everything goes through the C ABI (ray_tracing_amd.App is a ctypes wrapper).
The bar is bit-exactness: hit buffers, counters and accumulators are compared
as 32-bit words.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import (App, Material, RT_MAT_EMISSIVE, RT_MAT_GENERIC, RT_MAT_NORMAL, RT_OPT_COUNTERS,
                             RT_OPT_FRAME_STRIDE, RT_OPT_PIPELINE, RT_OPT_PROBE_HITS, RT_OPT_SEED_MODE, RtError,
                             default_scene, sphere_field_scene, terrain_scene)
from ray_tracing_amd import _lib as L
from ray_tracing_amd import scene as S
from ray_tracing_amd.distributed import strip_rows
from helpers import assert_bits_equal, copy_cam, gpu_render, make_manual_scene

pytestmark = pytest.mark.gpu
F = np.float32


def check_against_oracle(scene, cam, frames=2, materials=None, pipeline=1, nan_equal=False, options=()):
    g = gpu_render(scene, cam, frames, materials=materials, pipeline=pipeline, options=options)
    ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, materials=materials).render(frames)
    tri, inst, tb = g["hits"]
    assert_bits_equal(tri, ref.hit_triangle, "triangle_id")
    assert_bits_equal(inst, ref.hit_instance, "instance_id")
    assert_bits_equal(tb, ref.hit_t_bits, "t bits", nan_equal)
    box, tt = g["counters"]
    assert_bits_equal(box, ref.box_tests, "box_tests")
    assert_bits_equal(tt, ref.triangle_tests, "triangle_tests")
    assert_bits_equal(g["acc"], ref.acc, "accumulator", nan_equal)
    assert g["stats"] == ref.stats_dict()
    return g, ref


@pytest.mark.parametrize("pipeline", [0, 1])
def test_sphere_field_small(pipeline):
    """Config 2 in miniature: tessellated spheres (the reference has no sphere
    intersector), diffuse instance 1 under the sky, primary + 4 bounces."""
    scene, cam = sphere_field_scene(320, 180, n_spheres=300, depth=5, threads=4, rmin=2.0, rmax=6.0)
    g, ref = check_against_oracle(scene, cam, frames=2, pipeline=pipeline)
    assert (ref.hit_triangle != 0xFFFFFFFF).mean() > 0.05


@pytest.mark.parametrize("leaf_max", [16, 4])
def test_terrain_small(leaf_max):
    """Config 3 in miniature (glass / diffuse terrain / emissive, depth 10), both leaf sizes."""
    scene, cam = terrain_scene(320, 180, n=80, depth=10, threads=4, leaf_max=leaf_max)
    g, ref = check_against_oracle(scene, cam, frames=3)
    assert set(np.unique(ref.hit_instance[ref.hit_triangle != 0xFFFFFFFF])) == {0, 1, 2}


def test_closed_box_diffuse_stress_small():
    """Config 5 in miniature: the closed box keeps paths alive for all 9 segments."""
    scene, cam = terrain_scene(256, 144, n=60, depth=9, closed_box=True, light=False, threads=4, leaf_max=4)
    g, ref = check_against_oracle(scene, cam, frames=2)
    assert ref.stats_dict()["rays"] / ref.stats_dict()["paths"] > 6.0


def test_odd_canvas_sizes():
    scene, _ = default_scene(64, 36)
    for w, h in ((101, 37), (8, 8), (1, 1), (130, 9)):
        check_against_oracle(scene, S.make_cam(w, h), frames=2)


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_degenerate_depths_canvases_and_frame_counts(pipeline):
    """Depth 0 (Ray::get_color's loop runs zero times: every sample is black and nothing is traced,
    trace.rs:193-205), depth 1 and 2; canvases of one pixel, one row, one column; a call for zero
    frames; extreme fields of view -- on every kernel, against the oracle."""
    scene, cam = default_scene(32, 18)
    for depth in (0, 1, 2):
        c = copy_cam(cam)
        c.depth = depth
        g, ref = check_against_oracle(scene, c, frames=3, pipeline=pipeline)
        if depth == 0:
            assert g["stats"]["rays"] == 0 and g["stats"]["paths"] == 3 * 32 * 18
            assert (g["acc"][..., :3] == 0.0).all() and (g["acc"][..., 3] == 3.0).all()
            assert (g["hits"][0] == 0xFFFFFFFF).all()
    for w, h in ((1, 1), (7, 1), (1, 9)):
        check_against_oracle(scene, S.make_cam(w, h), frames=2, pipeline=pipeline)
    g = gpu_render(scene, cam, 0, pipeline=pipeline)  # a call for no frames: a canvas, nothing on it
    assert g["stats"]["paths"] == 0 and g["stats"]["rays"] == 0 and g["acc"].shape == (18, 32, 4) and (g["acc"] == 0.0).all()
    for fov in (1.0, 179.0):
        c = copy_cam(cam)
        c.fov = fov
        check_against_oracle(scene, c, frames=1, pipeline=pipeline)


def test_extreme_camera_fields_and_a_singular_instance():
    """A depth well beyond the usual 10, a frame counter that wraps past 2^32 (the event loop's
    `frame += 1`, main.rs:147), a frames_without_move so large that adding 1 no longer changes it,
    and an instance whose transform has no inverse (its object-space rays are not finite and hit
    nothing) -- against the oracle; a canvas without pixels is refused."""
    scene, cam = default_scene(32, 18)
    for field, value, frames in (("depth", 64, 2), ("frame", 0xFFFFFFFE, 4), ("frames_without_move", 3.0e7, 3)):
        c = copy_cam(cam)
        setattr(c, field, value)
        check_against_oracle(scene, c, frames=frames)
    flat = S.Scene(scene.info, scene.vertices, scene.triangles, scene.objects, scene.instances.copy(), scene.bvh)
    flat.instances["transform"][1] = np.frombuffer(bytes(S.affine_from_srt((1.0, 0.0, 1.0), (0, 0, 0, 1), (0, 0, 0))), F)
    for pipeline in (0, 1, 2):
        check_against_oracle(flat, cam, frames=2, pipeline=pipeline, nan_equal=True)
    for w, h in ((0, 5), (5, 0)):
        c = copy_cam(cam)
        c.canvas_width, c.canvas_height = w, h
        with pytest.raises(RtError) as e:
            gpu_render(scene, c, 1)
        assert e.value.code == L.RT_ERR_INVALID
    # a canvas within the pixel limit whose 8x8 tiles no longer fit the queues' 32-bit slot numbers
    c = copy_cam(cam)
    c.canvas_width, c.canvas_height = 600_000_000, 1
    with pytest.raises(RtError) as e:
        gpu_render(scene, c, 1)
    assert e.value.code == L.RT_ERR_LIMIT


def test_a_scene_without_instances_is_refused():
    scene, cam = default_scene(16, 9)
    info = L.SceneInfo(0, scene.info.num_bvh_nodes, scene.info.num_triangles, scene.info.sun_orientation)
    empty = S.Scene(info, scene.vertices, scene.triangles, scene.objects, scene.instances[:0].copy(), scene.bvh)
    with pytest.raises(RtError) as e:
        App.create(copy_cam(cam), empty)
    assert e.value.code == L.RT_ERR_INVALID


def test_material_table():
    """Per-instance materials (the reference hard-wires them, trace.rs:15-23): every
    branch of GenericMaterial::bxdf -- specular, refract/reflect, rough diffuse."""
    scene, cam = default_scene(200, 112)

    def mat(kind, cull, color=(1, 1, 1), specular=0.0, spec_rough=0.0, rough=0.0, ior=0.0):
        m = Material()
        m.kind, m.backface_culling = kind, cull
        m.color[:] = color
        m.specular, m.specular_roughness, m.roughness, m.ior = specular, spec_rough, rough, ior
        return m

    tables = [
        [mat(RT_MAT_GENERIC, 0, (0.9, 0.8, 0.7), specular=0.3, spec_rough=0.2, rough=0.5, ior=1.33),
         mat(RT_MAT_GENERIC, 1, (0.7, 0.7, 0.9), rough=1.0), mat(RT_MAT_EMISSIVE, 1, (4.0, 3.0, 2.0))],
        [mat(RT_MAT_NORMAL, 1), mat(RT_MAT_EMISSIVE, 0, (1.0, 2.0, 3.0)),
         mat(RT_MAT_GENERIC, 1, (0.5, 0.5, 0.5), specular=1.0, spec_rough=0.05)],
    ]
    for table in tables:
        g, ref = check_against_oracle(scene, cam, frames=3, materials=table)
    # NULL restores the reference constants
    app = App.create(copy_cam(cam), scene)
    try:
        app.set_materials(tables[0])
        app.set_materials(None)
        app.render_frames(2)
        assert_bits_equal(app.read_acc(), O.OracleRenderer(scene, cam, trig=O.TRIG_DET).render(2).acc, "acc")
        with pytest.raises(RtError):
            app.set_materials(tables[0][:2])
    finally:
        app.destroy()


@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_overbright_samples_trip_the_accumulator_guard(pipeline):
    """shader/src/lib.rs:71-98: a sample with a channel above 1000 makes the accumulator
    (0, 1e6, 0, 1) on every frame but the first after a move, which stores the sample as it is.
    A light of (40000, 20, 20) behind the default scene's emissive instance trips it wherever a
    path ends on the light; the rest of the image accumulates as usual."""
    scene, cam = default_scene(160, 90)
    table = []
    for i in range(int(scene.info.num_instances)):
        m = Material()
        L.load().rt_default_material(i, C.byref(m))
        table.append(m)
    table[2].kind = RT_MAT_EMISSIVE
    table[2].color[:] = (40000.0, 20.0, 20.0)
    g, ref = check_against_oracle(scene, cam, frames=1, materials=table, pipeline=pipeline)
    first = g["acc"]
    assert (first[..., 0] > 1000.0).sum() > 50, "no path ends on the light: the test tests nothing"
    assert not (first == np.array([0.0, 1.0e6, 0.0, 1.0], F)).all(-1).any()  # the first frame is stored as it is
    g, ref = check_against_oracle(scene, cam, frames=4, materials=table, pipeline=pipeline)
    tripped = (g["acc"] == np.array([0.0, 1.0e6, 0.0, 1.0], F)).all(-1)
    assert tripped.sum() > 50 and not tripped.all()
    shown = g["present"][tripped]  # (0, 1e6 / 4, 0) through the gamma curve
    np.testing.assert_allclose(shown[:, 1], F(1.0e6 / 4.0) ** F(1.0 / 2.2), rtol=1e-5)
    assert (shown[:, 0] == 0.0).all() and (shown[:, 2] == 0.0).all()


# ------------------------------------------------------------ hand-built BVHs ------
def ortho_like_cam(w, h, z=6.0, fov=60.0, depth=3):
    """Looks down -z onto the xy-plane from (0, 0, z)."""
    # columns: x' = (-1,0,0), y' = (0,1,0), z' = (0,0,-1)
    t = S.affine_from_columns((-1, 0, 0), (0, 1, 0), (0, 0, -1), (0, 0, z))
    return S.make_cam(w, h, t, fov=fov, depth=depth)


def test_several_hundred_instances_of_small_meshes():
    """Every ray walks the whole instance list (trace.rs:76): 400 instances of two small meshes,
    most of whose root boxes a given ray misses, some nested inside each other, plus one
    instance whose root is a single leaf."""
    rng = np.random.default_rng(11)
    sv, sf = S.icosphere(1)
    cv, cf = S.cube_mesh(inward=False)
    def xf(scale_lo, scale_hi, spread):
        axis, angle = "xyz"[int(rng.integers(0, 3))], float(rng.uniform(-3.1, 3.1))
        return S.affine_from_srt([float(x) for x in rng.uniform(scale_lo, scale_hi, 3)], S.quat_from_rotation(axis, angle),
                                 [float(x) for x in rng.uniform(-spread, spread, 3)])
    b = S.SceneBuilder()
    b.add_mesh(np.asarray(sv, F), np.asarray(sf, np.uint32), [xf(0.5, 2.0, 14.0) for _ in range(250)],
               S.bvh_params(leaf_max=3, root_leaf_max=0, threads=1))
    b.add_mesh(np.asarray(cv, F), np.asarray(cf, np.uint32), [xf(0.5, 6.0, 10.0) for _ in range(149)],
               S.bvh_params(leaf_max=2, root_leaf_max=0, threads=1))
    rv, rf = S.cube_mesh(inward=True)  # a room around everything, culled from outside; its root is a leaf
    b.add_mesh(np.asarray(rv, F), np.asarray(rf, np.uint32), [xf(20.0, 25.0, 1.0)],
               S.bvh_params(leaf_max=16, root_leaf_max=100, threads=1))
    scene = b.build()
    assert len(scene.instances) == 400
    kinds = [RT_MAT_GENERIC, RT_MAT_NORMAL, RT_MAT_EMISSIVE]
    materials = []
    for i in range(400):
        m = Material()
        m.kind, m.backface_culling = kinds[i % 3], int(i % 2)
        m.color[:] = [0.9, 0.8, 0.7] if m.kind != RT_MAT_EMISSIVE else [3.0, 3.0, 2.0]
        m.ior, m.roughness = (1.5 if i % 6 == 0 else 0.0), 0.4
        materials.append(m)
    cam = S.make_cam(96, 64, S.look_at((0.0, 4.0, -24.0), (0.0, 0.0, 0.0)), fov=70.0, depth=6)
    for pipeline in (1, 0):
        g, ref = check_against_oracle(scene, cam, frames=2, materials=materials, pipeline=pipeline)
    assert len(np.unique(ref.hit_instance[ref.hit_triangle != 0xFFFFFFFF])) > 30
    # the cull above the instance list (on by default, and so in the renders above) changes no word:
    # the same frames with every instance taken the reference's way, on both persistent kernels
    for pipeline in (1, 2):
        off = gpu_render(scene, cam, 2, materials=materials, pipeline=pipeline, options=((L.RT_OPT_INSTANCE_CULL, 0),))
        assert_bits_equal(off["acc"], g["acc"], "accumulator with the instance cull off")
        assert_bits_equal(off["counters"][0], g["counters"][0], "box tests with the instance cull off")
        assert off["stats"] == g["stats"]


def test_hand_built_trees_with_edge_cases():
    # 1. two coplanar triangles sharing an edge, in two leaves (tie rules, hit.rs:94-100, 159)
    left = [(-2, -1, 0), (0, -1, 0), (0, 1, 0)]
    right = [(0, -1, 0), (2, -1, 0), (0, 1, 0)]
    verts = np.array(left + right, F)
    idx = np.arange(6, dtype=np.uint32).reshape(2, 3)
    nodes = [((-2, -1, -0.5), (2, 1, 0.5), 1, 2, 0), ((-2, -1, -0.5), (0, 1, 0.5), 0, 0, 1),
             ((0, -1, -0.5), (2, 1, 0.5), 1, 1, 1)]
    check_against_oracle(make_manual_scene(verts, idx, nodes), ortho_like_cam(128, 64), frames=2)
    # 2. a flat box (never hit, trace.rs:225), a zero-area triangle (NaN is recorded, hit.rs:115),
    #    and an empty leaf range (first = last + 1)
    verts = np.array([(-1, -1, 0), (1, -1, 0), (0, 1, 0),      # real
                      (0, 0, 1), (1, 0, 1), (2, 0, 1),          # zero area
                      (-3, -1, 0), (-2, -1, 0), (-2.5, 1, 0)],  # inside a flat box
                     F)
    idx = np.arange(9, dtype=np.uint32).reshape(3, 3)
    nodes = [((-3, -1, -0.5), (2, 1, 1.5), 1, 2, 0),
             ((-1, -1, -0.5), (2, 1, 1.5), 3, 4, 0),   # internal: real + degenerate
             ((-3, -1, 0), (-2, 1, 0), 2, 2, 1),        # flat leaf
             ((-1, -1, -0.5), (2, 1, 1.5), 0, 1, 1),    # leaf with triangles 0..1
             ((-1, -1, -0.5), (1, 1, 0.5), 1, 0, 1)]    # empty leaf: 1..=0
    # one frame only, NaNs compared as a class: a NaN accumulator would feed its payload bits
    # into the next frame's seed, and NaN payloads are not portable between x86 and the GPU
    g, ref = check_against_oracle(make_manual_scene(verts, idx, nodes), ortho_like_cam(96, 48), frames=1,
                                  nan_equal=True)
    assert np.isnan(ref.hit_t_bits.view(F)).any()  # the NaN hit really occurs
    # 3. a single-leaf object (root_leaf_max, bvh.rs:17)
    check_against_oracle(make_manual_scene(verts[:3], idx[:1], [((-1, -1, -0.5), (1, 1, 0.5), 0, 0, 1)]),
                         ortho_like_cam(64, 64), frames=1)


def test_deep_tree_uses_the_large_stack():
    """A 45-level chain: deeper than the reference's 32-entry stack (hit.rs:73)."""
    n = 45
    verts, nodes = [], []
    for k in range(n):
        x = -6.0 + k * 0.27
        verts += [(x, -1, 0), (x + 0.25, -1, 0), (x + 0.125, 1, 0)]
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    # node 2k: internal (children 2k+1 = leaf k, 2k+2 = rest); the last one is a leaf
    for k in range(n - 1):
        x0 = -6.0 + k * 0.27
        nodes.append(((x0, -1, -0.5), (-6.0 + (n - 1) * 0.27 + 0.25, 1, 0.5), 2 * k + 1, 2 * k + 2, 0))
        nodes.append(((x0, -1, -0.5), (x0 + 0.25, 1, 0.5), k, k, 1))
    xl = -6.0 + (n - 1) * 0.27
    nodes.append(((xl, -1, -0.5), (xl + 0.25, 1, 0.5), n - 1, n - 1, 1))
    scene = make_manual_scene(np.array(verts, F), idx, nodes)
    for pipeline in (0, 1):
        check_against_oracle(scene, ortho_like_cam(160, 32, z=7.0, fov=90.0), frames=2, pipeline=pipeline)


def test_leaves_with_more_triangles_than_a_wave_has_lanes():
    """The persistent kernel shares a leaf's triangles out over the 64 lanes of a wave, 64
    (lane, triangle) pairs per pass.  A single 320-triangle leaf needs five passes per visit; two
    leaves of 313 and 7 triangles leave lanes with unequal remainders between passes."""
    verts, faces = S.icosphere(2)
    verts = np.asarray(verts, F) * F(1.5)
    faces = np.asarray(faces, np.uint32)
    assert len(faces) == 320
    lo, hi = (-1.6, -1.6, -1.6), (1.6, 1.6, 1.6)
    one_leaf = make_manual_scene(verts, faces, [(lo, hi, 0, 319, 1)])
    two_leaves = make_manual_scene(verts, faces, [(lo, hi, 1, 2, 0), (lo, hi, 0, 312, 1), (lo, hi, 313, 319, 1)])
    for scene in (one_leaf, two_leaves):
        for pipeline in (0, 1):
            g, ref = check_against_oracle(scene, ortho_like_cam(96, 64, z=5.0), frames=2, pipeline=pipeline)
        assert (ref.hit_triangle != 0xFFFFFFFF).mean() > 0.1


def flat_pad(pos):
    """What the GPU builders widen boxes without thickness by: 2^-16 of the mesh's largest extent
    (bvh_pad.hpp), in float32 like the builders."""
    lo, hi = pos.min(0).astype(F), pos.max(0).astype(F)
    ext, mag = F((hi - lo).max()), F(max(np.abs(lo).max(), np.abs(hi).max()))
    scale = ext if ext > 0 else (mag if mag > 0 else F(1.0))
    return F(scale * F(2.0 ** -16))


def widened(lo, hi, pad):
    """Boxes (n, 3) with every axis thinner than `pad` widened by it on both sides."""
    lo, hi = lo.astype(F).copy(), hi.astype(F).copy()
    thin = (hi - lo) < pad
    lo[thin] = (lo[thin] - pad).astype(F)
    hi[thin] = (hi[thin] + pad).astype(F)
    return lo, hi


def check_tree(nodes, n_triangles, leaf_max, pad=0.0):
    """Every triangle in exactly one leaf, every node reachable once, boxes nested (a widened flat
    child may stick out of its parent by the widening)."""
    seen_tri = np.zeros(n_triangles, np.int32)
    seen_node = np.zeros(len(nodes), np.int32)
    stack = [(0, None)]
    while stack:
        i, parent_box = stack.pop()
        seen_node[i] += 1
        nd = nodes[i]
        if parent_box is not None:
            assert (nd["min"] >= parent_box[0] - pad).all() and (nd["max"] <= parent_box[1] + pad).all()
        if nd["mode"] == 1:
            assert nd["c1"] <= nd["c2"] < n_triangles and nd["c2"] - nd["c1"] < leaf_max
            seen_tri[nd["c1"]:nd["c2"] + 1] += 1
        else:
            stack += [(int(nd["c1"]), (nd["min"], nd["max"])), (int(nd["c2"]), (nd["min"], nd["max"]))]
    assert (seen_tri == 1).all() and (seen_node == 1).all()


@pytest.mark.parametrize("leaf_max", [1, 4, 7])
def test_gpu_built_linear_bvh_is_a_valid_tree_and_renders_like_the_oracle(leaf_max):
    """rt_create_bvh_device: a different tree from the reference builder's, in the same format.
    It must be a proper BVH over a permutation of the input triangles, and the kernels must
    agree with the oracle on it as on any other tree (the oracle walks the very same nodes)."""
    verts, faces = S.terrain(n=40)
    faces = np.asarray(faces, np.uint32)
    tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max)
    assert sorted(map(tuple, tris.tolist())) == sorted(map(tuple, faces.tolist()))
    assert 2 * ((len(faces) + leaf_max - 1) // leaf_max) - 1 <= len(nodes) <= 2 * len(faces) - 1
    pos = np.asarray(verts, F)
    check_tree(nodes, len(tris), leaf_max, flat_pad(pos))
    check_tree_vectorised(nodes, tris, pos, leaf_max)  # every box: the bounds of its triangles
    ident = S.affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    shifted = S.affine_from_srt((0.5, 0.5, 0.5), S.quat_from_rotation("y", 0.7), (10, 25, 0))
    scene = S.single_object_scene(pos, tris, nodes, [ident, shifted])
    cam = S.make_cam(160, 90, S.look_at((60.0, 45.0, -70.0), (0.0, 0.0, 0.0)), fov=70.0, depth=4)
    g, ref = check_against_oracle(scene, cam, frames=2)
    assert (ref.hit_triangle != 0xFFFFFFFF).mean() > 0.3
    # degenerate inputs: one triangle, and many triangles with one and the same centroid
    one_t, one_n = S.create_bvh_gpu(pos[:3], np.array([[0, 1, 2]], np.uint32), leaf_max=4)
    assert len(one_n) == 1 and one_n[0]["mode"] == 1 and (one_n[0]["c1"], one_n[0]["c2"]) == (0, 0)
    same = np.tile(np.array([[0, 1, 2]], np.uint32), (100, 1))
    same_t, same_n = S.create_bvh_gpu(pos[:3], same, leaf_max=1)
    check_tree(same_n, 100, 1)


def check_tree_vectorised(nodes, tris, pos, leaf_max):
    """check_tree for trees too large to walk node by node in Python, and stricter: every node
    reached exactly once from the root, leaves tile the triangle array, every box is exactly the
    bounds of the triangles below it (widened by the builders' rule where those are flat)."""
    n_nodes, n_tris = len(nodes), len(tris)
    internal = nodes["mode"] == 0
    c1, c2 = nodes["c1"].astype(np.int64), nodes["c2"].astype(np.int64)
    kids = np.concatenate([c1[internal], c2[internal]])
    assert kids.min() >= 1 and kids.max() < n_nodes
    refs = np.bincount(kids, minlength=n_nodes)
    assert refs[0] == 0 and (refs[1:] == 1).all(), "a node with no parent or with two"
    seen, frontier, depth, levels = np.zeros(n_nodes, bool), np.array([0]), 0, []
    while len(frontier):
        assert not seen[frontier].any()
        seen[frontier] = True
        levels.append(frontier)
        inner = frontier[internal[frontier]]
        frontier = np.concatenate([c1[inner], c2[inner]])
        depth += 1
        assert depth <= 64
    assert seen.all(), "nodes not reachable from the root"
    leaves = np.flatnonzero(~internal)
    order = np.argsort(c1[leaves], kind="stable")
    first, last = c1[leaves][order], c2[leaves][order]
    assert first[0] == 0 and last[-1] == n_tris - 1 and (first[1:] == last[:-1] + 1).all(), "leaves do not tile the triangles"
    assert ((last - first) < leaf_max).all() and (last >= first).all()
    # every box is exactly the bounds of the triangles below it, widened where those bounds have no
    # thickness along an axis
    corners = pos[tris.ravel()].reshape(n_tris, 3, 3)
    tmin, tmax = corners.min(1), corners.max(1)
    tight_lo, tight_hi = np.zeros((n_nodes, 3), F), np.zeros((n_nodes, 3), F)
    tight_lo[leaves[order]] = np.minimum.reduceat(tmin, first)
    tight_hi[leaves[order]] = np.maximum.reduceat(tmax, first)
    for level_nodes in reversed(levels):  # children before parents
        inner = level_nodes[internal[level_nodes]]
        tight_lo[inner] = np.minimum(tight_lo[c1[inner]], tight_lo[c2[inner]])
        tight_hi[inner] = np.maximum(tight_hi[c1[inner]], tight_hi[c2[inner]])
    want_lo, want_hi = widened(tight_lo, tight_hi, flat_pad(pos[tris.ravel()]))  # the pad follows the vertices in use
    assert (want_lo == nodes["min"]).all(), "a box is not the (widened) bounds of its triangles"
    assert (want_hi == nodes["max"]).all(), "a box is not the (widened) bounds of its triangles"
    return depth


def test_gpu_built_bvh_of_a_million_triangles_is_exact_everywhere():
    """The linear-BVH builder at the bench scene's size: boxes are merged bottom-up by whichever
    thread reaches a node second, across workgroups and XCDs; a stale sibling box would give a
    silently too-small parent.  Every box of the whole tree is compared with what it must hold,
    on several builds in a row."""
    verts, faces = S.terrain(n=700)
    pos = np.asarray(verts, F)
    faces = np.asarray(faces, np.uint32)
    assert len(faces) >= 970_000
    for leaf_max in (4, 1, 16):
        tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max)
        assert len(tris) == len(faces)
        depth = check_tree_vectorised(nodes, tris, pos, leaf_max)
        assert depth <= 64
    key = lambda t: np.sort(t[:, 0].astype(np.uint64) << np.uint64(42) | t[:, 1].astype(np.uint64) << np.uint64(21) | t[:, 2].astype(np.uint64))
    assert (key(tris) == key(faces)).all(), "the triangles are not a permutation of the input"


@pytest.mark.parametrize("leaf_max", [1, 4, 7])
def test_gpu_built_sah_bvh_is_a_valid_tree_and_renders_like_the_oracle(leaf_max):
    """rt_create_bvh_device_sah: binned surface-area splits on the GPU.  Like the linear BVH it
    must be a proper BVH over a permutation of the input triangles with exact boxes, the kernels
    must agree with the oracle on it, and the same input must give the same array twice."""
    verts, faces = S.terrain(n=40)
    faces = np.asarray(faces, np.uint32)
    pos = np.asarray(verts, F)
    tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max, method="sah")
    assert sorted(map(tuple, tris.tolist())) == sorted(map(tuple, faces.tolist()))
    check_tree(nodes, len(tris), leaf_max, flat_pad(pos))
    assert check_tree_vectorised(nodes, tris, pos, leaf_max) <= 33
    tris2, nodes2 = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max, method="sah")
    assert tris2.tobytes() == tris.tobytes() and nodes2.tobytes() == nodes.tobytes()
    ident = S.affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    shifted = S.affine_from_srt((0.5, 0.5, 0.5), S.quat_from_rotation("y", 0.7), (10, 25, 0))
    scene = S.single_object_scene(pos, tris, nodes, [ident, shifted])
    cam = S.make_cam(160, 90, S.look_at((60.0, 45.0, -70.0), (0.0, 0.0, 0.0)), fov=70.0, depth=4)
    g, ref = check_against_oracle(scene, cam, frames=2)
    assert (ref.hit_triangle != 0xFFFFFFFF).mean() > 0.3
    # degenerate inputs: one triangle; many triangles with one and the same centroid (no plane
    # separates them: cut by position); a level budget that forces cuts in the middle
    one_t, one_n = S.create_bvh_gpu(pos[:3], np.array([[0, 1, 2]], np.uint32), leaf_max=4, method="sah")
    assert len(one_n) == 1 and one_n[0]["mode"] == 1 and (one_n[0]["c1"], one_n[0]["c2"]) == (0, 0)
    same = np.tile(np.array([[0, 1, 2]], np.uint32), (100, 1))
    same_t, same_n = S.create_bvh_gpu(pos[:3], same, leaf_max=1, method="sah")
    check_tree(same_n, 100, 1)
    assert check_tree_vectorised(same_n, same_t, pos, 1) == 8  # 100 leaves of one: a balanced tree
    need = int(np.ceil(np.log2(np.ceil(len(faces) / leaf_max)))) + 1
    tight_t, tight_n = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max, method="sah", max_levels=need)
    assert check_tree_vectorised(tight_n, tight_t, pos, leaf_max) == need
    # one level too few: the last level's nodes stay leaves whatever they hold (bvh.rs:75, 85)
    short_t, short_n = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max, method="sah", max_levels=need - 1)
    assert check_tree_vectorised(short_n, short_t, pos, 2 * leaf_max) == need - 1
    leaves = short_n[short_n["mode"] == 1]
    assert (leaves["c2"] - leaves["c1"] + 1).max() > leaf_max


def test_gpu_built_sah_bvh_of_a_million_triangles_is_exact_everywhere():
    """The binned builder at the bench scene's size: every box of the whole tree against what it
    must hold, the depth within what the 32-entry traversal stack takes."""
    verts, faces = S.terrain(n=700)
    pos = np.asarray(verts, F)
    faces = np.asarray(faces, np.uint32)
    for leaf_max in (4, 1, 16):
        tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max, method="sah")
        assert len(tris) == len(faces)
        assert check_tree_vectorised(nodes, tris, pos, leaf_max) <= 33
    key = lambda t: np.sort(t[:, 0].astype(np.uint64) << np.uint64(42) | t[:, 1].astype(np.uint64) << np.uint64(21) | t[:, 2].astype(np.uint64))
    assert (key(tris) == key(faces)).all(), "the triangles are not a permutation of the input"


def assert_same_tree(nodes_a, nodes_b):
    """Two node arrays describe one and the same tree -- the same box, mode and triangle range at every
    node, the same children in the same order -- whatever numbers the nodes carry.  Level by level from
    the two roots."""
    ia, ib = np.array([0], np.int64), np.array([0], np.int64)
    seen = 0
    while len(ia):
        a, b = nodes_a[ia], nodes_b[ib]
        assert (a["mode"] == b["mode"]).all()
        assert a["min"].tobytes() == b["min"].tobytes() and a["max"].tobytes() == b["max"].tobytes()
        leaf = a["mode"] == 1
        assert (a["c1"][leaf] == b["c1"][leaf]).all() and (a["c2"][leaf] == b["c2"][leaf]).all()
        seen += len(ia)
        inner = ~leaf
        ia = np.concatenate([a["c1"][inner], a["c2"][inner]]).astype(np.int64)
        ib = np.concatenate([b["c1"][inner], b["c2"][inner]]).astype(np.int64)
    assert seen == len(nodes_a) == len(nodes_b)


@pytest.mark.parametrize("grid,leaf_max,max_levels", [(40, 4, 0), (40, 1, 0), (40, 7, 12), (200, 4, 0), (200, 16, 0),
                                                     (700, 4, 0)])
def test_subtrees_finished_by_one_workgroup_are_the_ones_the_passes_make(grid, leaf_max, max_levels):
    """rt_create_bvh_device_sah hands the subtree of a node of 2048 triangles or fewer to one workgroup;
    rt_create_bvh_device_sah_passes_only takes every level through the level-wide passes.  Same bins,
    same cost, same rules: the same tree and the same triangle order, numbered differently."""
    verts, faces = S.terrain(n=grid)
    faces = np.asarray(faces, np.uint32)
    tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max, method="sah", max_levels=max_levels)
    tris_p, nodes_p = S.create_bvh_gpu(verts, faces, leaf_max=leaf_max, method="sah_passes", max_levels=max_levels)
    assert tris.tobytes() == tris_p.tobytes()
    assert_same_tree(nodes, nodes_p)
    # coincident centroids (cuts by position) inside a subtree and above it
    pos = np.asarray(verts, F)
    same = np.tile(np.array([[0, 1, 2]], np.uint32), (5000, 1))
    t1, n1 = S.create_bvh_gpu(pos[:3], same, leaf_max=leaf_max, method="sah")
    t2, n2 = S.create_bvh_gpu(pos[:3], same, leaf_max=leaf_max, method="sah_passes")
    assert t1.tobytes() == t2.tobytes()
    assert_same_tree(n1, n2)


def parse_obj(name):
    data = S._resource(name).encode()
    lib = L.load()
    nv, nt = C.c_size_t(0), C.c_size_t(0)
    lib.rt_parse_obj(data, len(data), None, C.byref(nv), None, C.byref(nt))
    v = np.zeros(nv.value, S.VERTEX_DT)
    t = np.zeros((nt.value, 3), np.uint32)
    lib.rt_parse_obj(data, len(data), S._ptr(v), C.byref(nv), S._ptr(t), C.byref(nt))
    return v["pos"].copy(), t


def test_coordinates_outside_the_reciprocal_division_range():
    """The slab test divides by cached reciprocals only while box coordinates, ray origin and
    direction are 0 or within [2^-24, 2^24]; outside it uses plain divisions.  Results are
    bit-exact either way."""
    base, cam = default_scene(96, 54)
    for k in (2.0 ** 28, 2.0 ** -28):
        b = S.SceneBuilder()
        for idx, name in enumerate(("default_cube.obj", "cornel_box.obj", "teapot.obj")):
            # object space scaled by k, instance transform by 1/k: the same world-space scene
            t = L.Affine3A.from_buffer_copy(base.instances["transform"][idx].tobytes())
            for axis in (t.x_axis, t.y_axis, t.z_axis):
                for i in range(3):
                    axis[i] = float(F(axis[i]) / F(k))
            v, f = parse_obj(name)
            b.add_mesh(v * F(k), f, [t])
        check_against_oracle(b.build(), cam, frames=2)
    # a tiny canvas whose centre column looks straight down the axis
    check_against_oracle(base, S.make_cam(9, 9), frames=1)


def test_invalid_scenes_are_rejected_before_any_launch():
    verts = np.array([(-1, -1, 0), (1, -1, 0), (0, 1, 0)], F)
    idx = np.array([[0, 1, 2]], np.uint32)
    good = [((-1, -1, -0.5), (1, 1, 0.5), 0, 0, 1)]
    app = App.create(S.make_cam(8, 8), None)
    try:
        with pytest.raises(RtError) as e:
            app.render()  # no scene yet
        assert e.value.code == L.RT_ERR_NO_SCENE
        for nodes in ([((-1, -1, -1), (1, 1, 1), 0, 5, 1)],            # triangle range out of bounds
                      [((-1, -1, -1), (1, 1, 1), 1, 2, 0)],            # child index out of bounds
                      [((-1, -1, -1), (1, 1, 1), 0, 0, 0)],            # a cycle
                      [((-1, -1, -1), (1, 1, 1), 0, 0, 7)]):           # unknown mode
            with pytest.raises(RtError) as e:
                app.upload_scene(make_manual_scene(verts, idx, nodes))
            assert e.value.code == L.RT_ERR_INVALID
        bad_idx = make_manual_scene(verts, np.array([[0, 1, 9]], np.uint32), good)
        with pytest.raises(RtError):
            app.upload_scene(bad_idx)
        app.upload_scene(make_manual_scene(verts, idx, good))  # and a good one still works
        app.render()
        with pytest.raises(RtError):
            app.set_rows(12, 0, 2)  # strips must be a multiple of 8 rows
        with pytest.raises(RtError):
            app.set_option(RT_OPT_PIPELINE, 9)
    finally:
        app.destroy()


# ------------------------------------------------------------ options ------
def test_tile_partition_is_byte_identical_to_one_context():
    """Tile mode: each 'rank' renders its interleaved strips in its own context; the rows
    put together equal the single-context image (seeds depend on absolute x, y)."""
    scene, cam = default_scene(320, 180)
    full = gpu_render(scene, cam, 3)["acc"]
    for world, strip in ((2, 8), (3, 16), (8, 8)):
        combined = np.zeros_like(full)
        for rank in range(world):
            app = App.create(copy_cam(cam), scene)
            try:
                app.set_rows(strip, rank, world)
                app.render_frames(3)
                acc = app.read_acc()
                stats = app.stats()
            finally:
                app.destroy()
            rows = strip_rows(180, strip, rank, world)
            assert stats["paths"] == 3 * len(rows) * 320
            other = np.setdiff1d(np.arange(180), rows)
            assert not acc[other].any()  # rows of other ranks are left untouched (zero)
            combined[rows] = acc[rows]
        assert_bits_equal(combined, full, f"tile mode world={world}")


def test_sample_split_options_match_the_oracle():
    scene, cam = default_scene(160, 90)
    total = np.zeros((90, 160, 4), F)
    for rank in range(3):
        c = copy_cam(cam)
        c.frame = rank
        app = App.create(c, scene)
        try:
            app.set_option(RT_OPT_FRAME_STRIDE, 3)
            app.set_option(RT_OPT_SEED_MODE, 1)
            app.render_frames(2)
            assert app.cam_data.frame == rank + 6
            acc = app.read_acc()
        finally:
            app.destroy()
        c = copy_cam(cam)
        c.frame = rank
        ref = O.OracleRenderer(scene, c, trig=O.TRIG_DET, frame_stride=3, independent_seeds=True).render(2)
        assert_bits_equal(acc, ref.acc, f"sample mode rank {rank}")
        total += acc
    assert np.all(total[..., 3] == 6.0)


def test_accumulator_write_read_clear_and_frame_by_frame_equals_batched():
    scene, cam = default_scene(96, 54)
    app = App.create(copy_cam(cam), scene)
    try:
        rng = np.random.default_rng(0)
        img = rng.random((54, 96, 4)).astype(F)
        app.write_acc(img)
        assert_bits_equal(app.read_acc(), img, "write/read")
        app.clear_acc()
        assert not app.read_acc().any()
        for _ in range(5):
            app.render()  # one frame per call, counters advanced like main.rs:147-148
        one_by_one = app.read_acc()
        assert app.cam_data.frame == 5 and app.cam_data.frames_without_move == 5.0
    finally:
        app.destroy()
    assert_bits_equal(gpu_render(scene, cam, 5)["acc"], one_by_one, "5 calls of 1 frame vs 1 call of 5 frames")
    # a camera move resets the accumulation (frames_without_move = 0, lib.rs:85-87) but the
    # seed still hashes what is in the buffer
    app = App.create(copy_cam(cam), scene)
    try:
        app.render_frames(3)
        app.update_pos((0.5, 0.0, 0.0))
        app.cam_data.frames_without_move = 0.0
        app.render_frames(2)
        got = app.read_acc()
        cam2 = copy_cam(app.cam_data)
    finally:
        app.destroy()
    ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET).render(3)
    ref.cam.transform = cam2.transform
    ref.cam.frames_without_move = 0.0
    ref.render(2)
    assert_bits_equal(got, ref.acc, "after a camera move")


# ------------------------------------------------------------ full size ------
@pytest.fixture(scope="module")
def c3():
    return terrain_scene(1920, 1080, n=700, depth=10, threads=16, leaf_max=4)


def test_full_size_rows_match_the_oracle(c3):
    """1.0 M triangles at 1920x1080: the oracle renders a band of rows of the same
    canvas (seeds depend on absolute coordinates), the GPU the whole frame."""
    scene, cam = c3
    g = gpu_render(scene, cam, 2)
    rows = np.concatenate([np.arange(0, 4), np.arange(536, 544), np.arange(1076, 1080)])
    ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, threads=16).render(2, rows=rows)
    tri, inst, tb = g["hits"]
    assert_bits_equal(tri[rows], ref.hit_triangle[rows], "triangle_id")
    assert_bits_equal(tb[rows], ref.hit_t_bits[rows], "t bits")
    assert_bits_equal(g["counters"][0][rows], ref.box_tests[rows], "box_tests")
    assert_bits_equal(g["acc"][rows], ref.acc[rows], "accumulator")


def compare_everything(g, ref):
    tri, inst, tb = g["hits"]
    assert_bits_equal(tri, ref.hit_triangle, "triangle_id")
    assert_bits_equal(inst, ref.hit_instance, "instance_id")
    assert_bits_equal(tb, ref.hit_t_bits, "t bits")
    assert_bits_equal(g["counters"][0], ref.box_tests, "box_tests")
    assert_bits_equal(g["counters"][1], ref.triangle_tests, "triangle_tests")
    assert_bits_equal(g["acc"], ref.acc, "accumulator")
    assert g["stats"] == ref.stats_dict()


def test_whole_canvas_of_one_rank_of_config_4_matches_the_oracle():
    """Config 4 at its full size: 10.0 M triangles (tree from the binned GPU builder), 3840x2160,
    sample-split -- rank 5 of 8 renders frames 5 and 13 with the independent-seed variant; every
    pixel of its accumulator, hit records and counters against the oracle in the same mode."""
    scene, cam = terrain_scene(3840, 2160, n=2236, depth=10, leaf_max=4, builder="gpu_sah", device=0)
    assert scene.n_triangles > 9_900_000
    cam.frame = 5
    g = gpu_render(scene, cam, 2, options=((RT_OPT_FRAME_STRIDE, 8), (RT_OPT_SEED_MODE, 1)))
    ref = O.OracleRenderer(scene, copy_cam(cam), trig=O.TRIG_DET, threads=16, frame_stride=8, independent_seeds=True).render(2)
    compare_everything(g, ref)
    assert g["stats"]["paths"] == 2 * 3840 * 2160


@pytest.mark.parametrize("config", ["c3", "c2", "c5"])
def test_whole_canvas_of_the_bench_scenes_matches_the_oracle(config):
    """The scenes bench.py times, at their full size and on the trees of its default builder (the
    binned GPU builder), every pixel against the oracle -- hit records, counters, accumulator and
    totals, not a band of rows.  c3: one whole bench step (1.0 M triangles, 1920x1080, 16 chained
    frames, depth 10); c2: 10 000 tessellated spheres (3.2 M triangles), 1 frame, depth 5; c5: the
    terrain in a closed box, all diffuse, depth 9, 4 frames."""
    if config == "c3":
        scene, cam = terrain_scene(1920, 1080, n=700, depth=10, leaf_max=4, builder="gpu_sah", device=0)
        frames = 16
    elif config == "c2":
        scene, cam = sphere_field_scene(1920, 1080, n_spheres=10000, depth=5, leaf_max=4, builder="gpu_sah", device=0)
        frames = 1
    else:
        scene, cam = terrain_scene(1920, 1080, n=700, depth=9, closed_box=True, light=False, leaf_max=4,
                                   builder="gpu_sah", device=0)
        frames = 4
    g = gpu_render(scene, cam, frames)
    ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, threads=16).render(frames)
    compare_everything(g, ref)
    assert g["stats"]["paths"] == frames * 1920 * 1080
    if config == "c5":  # nothing leaves the box and nothing in it emits: every path runs to the depth limit
        assert g["stats"]["rays"] == 9 * g["stats"]["paths"]


def test_full_size_properties(c3):
    """Size-independent properties at the benchmark size: both kernels agree, a
    render is reproducible, the 8-way strip partition reassembles to the same image,
    and the counters add up."""
    scene, cam = c3
    a = gpu_render(scene, cam, 4, pipeline=1)
    b = gpu_render(scene, cam, 4, pipeline=1)
    assert_bits_equal(a["acc"], b["acc"], "reproducibility")
    c = gpu_render(scene, cam, 4, pipeline=0)
    assert_bits_equal(a["acc"], c["acc"], "plain vs persistent kernel")
    assert a["stats"] == c["stats"]
    assert a["stats"]["paths"] == 4 * 1920 * 1080
    assert a["stats"]["paths"] <= a["stats"]["rays"] <= 10 * a["stats"]["paths"]
    assert np.all(a["acc"][..., 3] == 4.0)
    combined = np.zeros_like(a["acc"])
    for rank in range(8):
        app = App.create(copy_cam(cam), scene)
        try:
            app.set_rows(8, rank, 8)
            app.render_frames(4)
            rows = strip_rows(1080, 8, rank, 8)
            combined[rows] = app.read_acc()[rows]
        finally:
            app.destroy()
    assert_bits_equal(combined, a["acc"], "8-way tile partition")


def test_hardware_reciprocal_is_within_one_ulp_over_the_divisor_range():
    """The instance walk first decides root boxes from v_rcp_f32 reciprocals with a margin
    derived from a 1-ulp bound; this checks that bound for every float in the range."""
    worst = C.c_uint32(99)
    assert L.load().rt_selftest_rcp_device(0, C.byref(worst)) == 0
    assert worst.value <= 1


def test_short_sqrt_reciprocal_and_rand_quotient_are_correctly_rounded_for_every_operand():
    """fastexact.hpp: compared on the device with the compiler's IEEE results over the whole
    range in which the kernels use them."""
    bad = (C.c_uint64 * 3)(9, 9, 9)
    assert L.load().rt_selftest_fastmath_device(0, bad) == 0
    assert list(bad) == [0, 0, 0]


def test_moving_instances_between_frames_equals_uploading_the_moved_scene():
    """rt_update_instances: the reference refills its instance buffer before every frame
    (vulkan.rs:238-353); here only the instance records are rewritten."""
    scene, cam = default_scene(96, 54)
    moved = scene.instances.copy()
    t = S.affine_from_srt((4.0, 6.0, 4.0), S.quat_from_rotation("z", 0.6), (3.0, -1.0, 2.0))
    moved["transform"][0] = np.frombuffer(bytes(t), F)
    moved["object_id"][[1, 2]] = moved["object_id"][[2, 1]]  # instances 1 and 2 swap their meshes
    moved_scene = S.Scene(scene.info, scene.vertices, scene.triangles, scene.objects, moved, scene.bvh)
    ref = O.OracleRenderer(moved_scene, cam, trig=O.TRIG_DET).render(2)
    with App.create(copy_cam(cam), scene) as app:
        app.set_option(RT_OPT_PROBE_HITS, 1)
        app.render_frames(3)          # some frames of the original scene first
        app.update_instances(moved)
        app.cam_data = copy_cam(cam)
        app.clear_acc()
        app.render_frames(2)
        app.synchronize()
        assert_bits_equal(app.read_acc().reshape(-1, 4), ref.acc.reshape(-1, 4), "accumulator after the move")
        tri, inst, tb = app.read_hits()
        assert_bits_equal(tri, ref.hit_triangle, "triangle_id")
        assert_bits_equal(inst, ref.hit_instance, "instance_id")
        # two instances showing one object (its triangles' world-space normals can no longer be formed once
        # per triangle), then every object shown once again
        shared = moved.copy()
        shared["object_id"][2] = shared["object_id"][1]
        shared_scene = S.Scene(scene.info, scene.vertices, scene.triangles, scene.objects, shared, scene.bvh)
        for instances, its_scene in ((shared, shared_scene), (moved, moved_scene)):
            want = O.OracleRenderer(its_scene, cam, trig=O.TRIG_DET).render(2)
            app.update_instances(instances)
            app.cam_data = copy_cam(cam)
            app.clear_acc()
            app.render_frames(2)
            assert_bits_equal(app.read_acc().reshape(-1, 4), want.acc.reshape(-1, 4), "accumulator after the second move")
        with pytest.raises(RtError):
            app.update_instances(moved[:2])
        bad = moved.copy()
        bad["object_id"][0] = 99
        with pytest.raises(RtError):
            app.update_instances(bad)


def test_record_number_item_names_for_arenas_beyond_4_gib():
    """Above 4 GiB of nodes and triangle records the persistent kernel names its items by record
    number instead of byte offset.  RT_OPT_FORCE_BIG_OFFSETS forces that variant on scenes the
    oracle can render."""
    big = [(L.RT_OPT_FORCE_BIG_OFFSETS, 1)]
    scene, cam = terrain_scene(200, 112, n=60, depth=8, threads=4, leaf_max=4)
    check_against_oracle(scene, cam, frames=2, options=big)
    scene, cam = default_scene(128, 72)
    check_against_oracle(scene, cam, frames=3, options=big)


def test_tuning_options_change_no_pixel():
    """The persistent kernel's tuning switches (tile order, vote weights, resident workgroups,
    plain division in the slab test, how sorted tiles are dealt) only move work around."""
    scene, cam = terrain_scene(200, 112, n=60, depth=8, threads=4, leaf_max=4)
    for opts in ([(L.RT_OPT_TILE_ORDER, 0)], [(L.RT_OPT_VOTE_WEIGHTS, 1 | 1 << 8 | 1 << 16)],
                 # one kind of step outvoting the others whenever a single lane waits for it: the kernel's two
                 # loops change hands at every turn, or hardly ever
                 [(L.RT_OPT_VOTE_WEIGHTS, 255 | 1 << 8 | 1 << 16)], [(L.RT_OPT_VOTE_WEIGHTS, 1 | 255 << 8 | 1 << 16)],
                 [(L.RT_OPT_VOTE_WEIGHTS, 1 | 1 << 8 | 255 << 16)],
                 [(L.RT_OPT_BLOCKS_PER_CU, 1)], [(L.RT_OPT_FAST_DIV, 0)], [(L.RT_OPT_TILE_DEAL, 0)],
                 [(L.RT_OPT_TILE_DEAL, 1)]):
        check_against_oracle(scene, cam, frames=2, options=opts)
    with App.create(copy_cam(cam), scene) as app:
        for opt, bad in ((L.RT_OPT_VOTE_WEIGHTS, 0), (L.RT_OPT_BLOCKS_PER_CU, 9), (L.RT_OPT_TILE_DEAL, 2)):
            with pytest.raises(RtError):
                app.set_option(opt, bad)


def test_eight_bit_present_is_the_float_image_quantised_like_the_png_writer():
    """rt_read_present_rgba8 converts binding 8 on the GPU: clamp to [0, 1], times 255, round to
    nearest (even), NaN -> 0 -- the bytes rt_save_png derives from rt_read_present on the host."""
    scene, cam = S.default_scene(333, 127)  # odd sizes: the last workgroup is partly empty
    with App.create(cam, scene) as app:
        with pytest.raises(L.RtError):
            app.read_present_rgba8()  # nothing rendered yet
        app.render_frames(3)
        f = app.read_present()
        b = app.read_present_rgba8()
        again = app.read_present_rgba8()
        # a render with the gamma pass switched off leaves binding 8 behind the accumulator: reading it
        # is refused, in every form, until a render with the pass on has brought it up to date
        app.set_option(L.RT_OPT_PRESENT, 0)
        app.render_frames(1)
        for read in (app.read_present, app.read_present_rgba8, app.present_begin):
            with pytest.raises(L.RtError):
                read()
        app.set_option(L.RT_OPT_PRESENT, 1)
        app.render_frames(1)
        assert app.read_present_rgba8().shape == (127, 333, 4)
    want = np.where(f > 0, np.minimum(np.rint(np.minimum(f, F(1.0)) * F(255.0)), 255), 0).astype(np.uint8)
    assert b.shape == (127, 333, 4) and np.array_equal(b, want) and np.array_equal(again, b)
    assert b[..., :3].max() == 255 and 0 < b[..., :3].mean() < 255 and (b[..., 3] == 255).all()


def test_present_begin_end_hands_out_the_frame_it_was_started_for():
    """rt_present_begin / rt_present_end: the copy of frame k runs beside the render of frame k + 1 and
    must still be frame k; a second begin waits for the first copy before it reuses the buffers."""
    scene, cam = S.default_scene(320, 180)
    with App.create(copy_cam(cam), scene) as ref:
        frames = []
        for _ in range(3):
            ref.render_frames(1)
            frames.append(ref.read_present_rgba8())
    assert not np.array_equal(frames[0], frames[1])
    with App.create(copy_cam(cam), scene) as app:
        with pytest.raises(L.RtError):
            app.present_end()  # nothing begun
        app.render_frames(1)
        app.present_begin()
        app.render_frames(1)  # enqueued behind the quantisation, beside the copy
        assert np.array_equal(app.present_end(), frames[0])
        app.present_begin()
        app.render_frames(1)
        app.present_begin()  # frame 3 before frame 2 was asked for: waits for that copy, then replaces it
        assert np.array_equal(app.present_end(), frames[2])
        assert np.array_equal(app.read_present_rgba8(), frames[2])


@pytest.mark.parametrize("builder", ["gpu_sah", "gpu_linear"])
def test_gpu_builders_keep_axis_aligned_walls_visible(builder):
    """The reference's slab test never accepts a box without thickness (trace.rs:208-229), and a
    builder that splits by surface area puts an axis-aligned wall into leaves of its own, whose
    tight boxes are exactly that.  The GPU builders widen such boxes a little; inside a closed
    box every primary ray must then hit something, at the distance it does on the tree of the
    reference's split rule, and the kernels must still agree with the oracle."""
    verts, faces = S.terrain(n=60, closed_box=True)
    pos, faces = np.asarray(verts, F), np.asarray(faces, np.uint32)
    ident = S.affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    cam = S.make_cam(160, 90, S.look_at((0.0, 55.0, -95.0), (0.0, 0.0, 0.0)), depth=4)
    hits = {}
    for b in ("host", builder):
        scene = S.SceneBuilder().add_mesh(pos, faces, [ident], S.bvh_params(leaf_max=4), builder=b).build()
        g, ref = check_against_oracle(scene, cam, frames=2)
        assert (ref.hit_triangle != 0xFFFFFFFF).all(), f"{b}: a primary ray left the closed box"
        hits[b] = ref.hit_t_bits.copy()
    # the same surfaces at the same distances, whatever the tree (ties aside: none between a wall and the terrain)
    assert (hits["host"] == hits[builder]).mean() > 0.999
    tris, nodes = S.create_bvh_gpu(pos, faces, leaf_max=4, method="sah" if builder == "gpu_sah" else "linear")
    check_tree_vectorised(nodes, tris, pos, 4)
    thin = ((nodes["max"] - nodes["min"]) < 2 * flat_pad(pos) * F(1.01)).any(1)
    assert thin.any() and ((nodes["max"] - nodes["min"])[thin].min() > 0)  # there were flat boxes, none is left flat
