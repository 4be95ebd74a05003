"""Scene assembly on the host: the Python face of the library's SceneBuilder
(program/src/modules/mod.rs:38-134 of the reference), the reference's default
scene (program/src/main.rs:22-66) and seeded synthetic scenes.

A `Scene` is five NumPy arrays laid out byte-for-byte like the reference's
`BufferSceneInfo` vectors, so the same object can be handed to the GPU library
and to any other consumer of that layout.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L

VERTEX_DT = np.dtype([("pos", "<f4", (3,)), ("pad", "u1", (4,))])
BVH_DT = np.dtype([("min", "<f4", (3,)), ("p1", "<u4"), ("max", "<f4", (3,)), ("p2", "<u4"),
                   ("c1", "<u4"), ("c2", "<u4"), ("mode", "<u4"), ("pad", "<u4")])
INSTANCE_DT = np.dtype([("transform", "<f4", (16,)), ("object_id", "<u4"), ("pad", "<u4", (3,))])
assert VERTEX_DT.itemsize == 16 and BVH_DT.itemsize == 48 and INSTANCE_DT.itemsize == 80

RESOURCES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources")


@dataclass
class Scene:
    info: L.SceneInfo
    vertices: np.ndarray   # VERTEX_DT
    triangles: np.ndarray  # (n, 3) uint32, GLOBAL vertex indices
    objects: np.ndarray    # uint32 bvh_root
    instances: np.ndarray  # INSTANCE_DT
    bvh: np.ndarray        # BVH_DT
    materials: np.ndarray = field(default=None)  # optional array of L.Material as bytes

    @property
    def n_triangles(self):
        return int(self.triangles.shape[0])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def bvh_params(leaf_max=16, root_leaf_max=5, max_depth=32, threads=1):
    p = L.BvhParams()
    L.load().rt_bvh_params_default(C.byref(p))
    p.leaf_max, p.root_leaf_max, p.max_depth, p.threads = leaf_max, root_leaf_max, max_depth, threads
    return p


def affine_from_srt(scale, quat_xyzw, translation):
    """glam Affine3A::from_scale_rotation_translation, as main.rs:23-57 uses it."""
    out = L.Affine3A()
    f3 = C.c_float * 3
    L.load().rt_affine_from_srt(f3(*scale), (C.c_float * 4)(*quat_xyzw), f3(*translation), C.byref(out))
    return out


def quat_from_rotation(axis, angle):
    q = (C.c_float * 4)()
    L.load().rt_quat_from_rotation("xyz".index(axis), C.c_float(angle), q)
    return tuple(q)


def affine_from_columns(x, y, z, t):
    out = L.Affine3A()
    out.x_axis[:] = [float(v) for v in x]
    out.y_axis[:] = [float(v) for v in y]
    out.z_axis[:] = [float(v) for v in z]
    out.translation[:] = [float(v) for v in t]
    return out


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """Camera-to-world transform whose +z axis points from eye to target
    (the reference's camera looks down +z, trace.rs:34).  Pixel row 0 is the top of
    the image and maps to the NEGATIVE camera y (trace.rs:30), so the camera's y
    axis points down in the world."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(np.asarray(up, np.float64), f)
    r /= np.linalg.norm(r)
    u = -np.cross(f, r)
    return affine_from_columns(r.astype(np.float32), u.astype(np.float32), f.astype(np.float32),
                               eye.astype(np.float32))


def create_bvh_gpu(positions, triangles, leaf_max=4, device=0, method="linear", max_levels=0):
    """A BVH built on the GPU: returns the triangles in the order the tree wants them and the
    nodes in the reference's 48-byte format (node 0 is the root).  method "linear": Morton-sorted
    linear BVH (rt_create_bvh_device); "sah": binned surface-area splits by the reference's cost
    function (rt_create_bvh_device_sah); "sah_passes": the same tree built by level-wide passes only
    (rt_create_bvh_device_sah_passes_only)."""
    lib = L.load()
    v = np.zeros(len(positions), VERTEX_DT)
    v["pos"] = np.ascontiguousarray(positions, np.float32)
    t = np.ascontiguousarray(triangles, np.uint32).copy()
    n = C.c_size_t(0)
    if method == "linear":
        def call(nodes_ptr):
            return lib.rt_create_bvh_device(device, _ptr(v), len(v), _ptr(t), len(t), leaf_max, nodes_ptr, C.byref(n))
    elif method in ("sah", "sah_passes"):
        build = lib.rt_create_bvh_device_sah if method == "sah" else lib.rt_create_bvh_device_sah_passes_only

        def call(nodes_ptr):
            return build(device, _ptr(v), len(v), _ptr(t), len(t), leaf_max, max_levels, nodes_ptr, C.byref(n))
    else:
        raise ValueError(method)
    rc = call(None)
    if rc != L.RT_OK:
        raise L.RtError(rc, "GPU BVH builder: bad mesh")
    nodes = np.zeros(n.value, BVH_DT)
    rc = call(_ptr(nodes))
    if rc != L.RT_OK:
        raise L.RtError(rc, "GPU BVH builder failed (no GPU?)")
    return t, nodes[:n.value].copy()


def single_object_scene(positions, triangles, nodes, transforms):
    """A scene of one mesh with a ready-made BVH (root = node 0) and one instance per transform."""
    v = np.zeros(len(positions), VERTEX_DT)
    v["pos"] = np.ascontiguousarray(positions, np.float32)
    inst = np.zeros(len(transforms), INSTANCE_DT)
    for i, t in enumerate(transforms):
        inst["transform"][i] = np.frombuffer(bytes(t), np.float32)
    info = L.SceneInfo(len(transforms), len(nodes), len(triangles), (C.c_float * 3)(1, -1, 1))
    return Scene(info, v, np.ascontiguousarray(triangles, np.uint32), np.zeros(1, np.uint32), inst,
                 np.ascontiguousarray(nodes))


class SceneBuilder:
    """program/src/modules/mod.rs:38-134, backed by the library's C++ builder."""

    def __init__(self, device=0):
        """`device`: the GPU the GPU BVH builders run on, for meshes added with builder="gpu_..."."""
        self._lib = L.load()
        self._b = self._lib.rt_scene_builder_new()
        self._device = device

    def __del__(self):
        if getattr(self, "_b", None):
            self._lib.rt_scene_builder_free(self._b)
            self._b = None

    @staticmethod
    def _xforms(transforms):
        arr = (L.Affine3A * len(transforms))(*transforms)
        return arr, len(transforms)

    _METHODS = {"host": L.RT_BVH_HOST, "gpu_linear": L.RT_BVH_DEVICE_LINEAR, "gpu_sah": L.RT_BVH_DEVICE_SAH}

    def _use(self, builder):
        if self._lib.rt_scene_builder_use_bvh_builder(self._b, self._METHODS[builder], self._device) != L.RT_OK:
            raise L.RtError(L.RT_ERR_INVALID, "rt_scene_builder_use_bvh_builder")

    def add_obj_file(self, text, transforms, params=None, builder="host"):
        data = text.encode() if isinstance(text, str) else bytes(text)
        arr, n = self._xforms(transforms)
        p = params or bvh_params()
        self._use(builder)
        rc = self._lib.rt_scene_builder_add_obj(self._b, data, len(data), arr, n, C.byref(p))
        if rc != L.RT_OK:
            raise L.RtError(rc, "rt_scene_builder_add_obj: unusable OBJ text")
        return self

    def add_mesh(self, positions, triangles, transforms, params=None, builder="host"):
        """builder "host": the reference's split rule on the host (rt_create_bvh); "gpu_sah" /
        "gpu_linear": one of the GPU builders, with the leaf size of `params`."""
        positions = np.ascontiguousarray(positions, np.float32)
        v = np.zeros(len(positions), VERTEX_DT)
        v["pos"] = positions
        t = np.ascontiguousarray(triangles, np.uint32)
        arr, n = self._xforms(transforms)
        p = params or bvh_params()
        self._use(builder)
        rc = self._lib.rt_scene_builder_add_mesh(self._b, _ptr(v), len(v), _ptr(t), len(t), arr, n, C.byref(p))
        if rc != L.RT_OK:
            raise L.RtError(rc, "rt_scene_builder_add_mesh: bad mesh")
        return self

    def sun_orientation(self, x, y, z):
        self._lib.rt_scene_builder_sun_orientation(self._b, x, y, z)
        return self

    def build(self):
        info = L.SceneInfo()
        ptrs = [C.c_void_p() for _ in range(5)]
        ns = [C.c_size_t() for _ in range(5)]
        args = []
        for p, n in zip(ptrs, ns):
            args += [C.byref(p), C.byref(n)]
        rc = self._lib.rt_scene_builder_build(self._b, C.byref(info), *args)
        if rc != L.RT_OK:
            raise L.RtError(rc, "rt_scene_builder_build")

        def grab(p, n, dt):
            if n.value == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(p.value)
            return np.frombuffer(buf, dtype=dt).copy()

        verts = grab(ptrs[0], ns[0], VERTEX_DT)
        tris = grab(ptrs[1], ns[1], np.dtype(("<u4", (3,)))).reshape(-1, 3)
        objs = grab(ptrs[2], ns[2], np.dtype("<u4"))
        inst = grab(ptrs[3], ns[3], INSTANCE_DT)
        bvh = grab(ptrs[4], ns[4], BVH_DT)
        return Scene(info, verts, tris, objs, inst, bvh)


def make_cam(width, height, transform=None, fov=90.0, depth=10, frame=0, frames_without_move=0.0,
             debug_number=128, debug_information=L.RT_DEBUG_NONE):
    cam = L.CamData()
    cam.depth = depth
    cam.transform = transform if transform is not None else affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, -20))
    cam.canvas_width, cam.canvas_height = width, height
    cam.fov = fov
    cam.frame = frame
    cam.debug_number = debug_number
    cam.debug_information = debug_information
    cam.frames_without_move = frames_without_move
    return cam


def _resource(name):
    with open(os.path.join(RESOURCES, name), "r") as f:
        return f.read()


def default_scene(width=1280, height=720):
    """The reference's start-up scene and camera, program/src/main.rs:22-66:
    glass cube (instance 0), Cornell box (1), emissive teapot (2)."""
    pi = np.float32(math.pi)
    t_teapot = affine_from_srt((np.float32(0.6) * 2, np.float32(1.01) * 2, np.float32(0.6) * 2),
                               quat_from_rotation("x", float(pi)), (0.0, -9.8, 0.0))
    t_cube = affine_from_srt((5.0, 5.0, 5.0), quat_from_rotation("y", float(pi / np.float32(4.0))), (0.0, 1.9, 0.0))
    t_box = affine_from_srt((10.0, 10.0, 10.0), (0.0, 0.0, 0.0, 1.0), (0.0, 0.0, 0.0))
    scene = (SceneBuilder()
             .add_obj_file(_resource("default_cube.obj"), [t_cube])
             .add_obj_file(_resource("cornel_box.obj"), [t_box])
             .add_obj_file(_resource("teapot.obj"), [t_teapot])
             .sun_orientation(1.0, -1.0, 1.0)
             .build())
    return scene, make_cam(width, height)


# --------------------------------------------------------------------------
# Seeded synthetic scenes.  Only integer hashing and IEEE + - * / sqrt are
# used, so the same seed gives the same bytes on every machine.
# --------------------------------------------------------------------------
def _hash_u32(x):
    """splitmix-style avalanche of a uint64 array, returns uint64 in [0, 2^32)."""
    x = np.asarray(x, np.uint64)
    m = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & m
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & m
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & m
        x = x ^ (x >> np.uint64(31))
    return x >> np.uint64(32)


def _uniform(seed, stream, n):
    """n doubles in [0, 1), a pure function of (seed, stream, index)."""
    idx = np.arange(n, dtype=np.uint64)
    key = (np.uint64(seed) << np.uint64(40)) ^ (np.uint64(stream) << np.uint64(32)) ^ idx
    return _hash_u32(key).astype(np.float64) / 4294967296.0


def icosphere(subdivisions):
    """Unit icosphere: 20 * 4^s outward-facing (counter-clockwise) triangles."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(p, np.float64) / math.sqrt(1.0 + t * t) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
             (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache = {}

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / math.sqrt(float(m @ m)))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts, np.float64), np.array(faces, np.uint32)


_CUBE_V = np.array([(1, 1, -1), (1, -1, -1), (1, 1, 1), (1, -1, 1), (-1, 1, -1), (-1, -1, -1), (-1, 1, 1),
                    (-1, -1, 1)], np.float64)
# outward-facing quads of the cube above, fan-triangulated
_CUBE_Q = [(0, 4, 6, 2), (3, 2, 6, 7), (7, 6, 4, 5), (5, 1, 3, 7), (1, 0, 2, 3), (5, 4, 0, 1)]


def cube_mesh(inward=False):
    tris = []
    for a, b, c, d in _CUBE_Q:
        tris += [(a, b, c), (a, c, d)]
    tris = np.array(tris, np.uint32)
    if inward:
        tris = tris[:, ::-1].copy()
    return _CUBE_V.copy(), tris


def sphere_field(n_spheres=10000, seed=1, subdivisions=2, extent=50.0, rmin=0.3, rmax=1.0):
    """n tessellated spheres, centres uniform in [-extent, extent]^3, radii uniform in [rmin, rmax]."""
    sv, sf = icosphere(subdivisions)
    cx = (_uniform(seed, 0, n_spheres) * 2.0 - 1.0) * extent
    cy = (_uniform(seed, 1, n_spheres) * 2.0 - 1.0) * extent
    cz = (_uniform(seed, 2, n_spheres) * 2.0 - 1.0) * extent
    r = rmin + _uniform(seed, 3, n_spheres) * (rmax - rmin)
    centres = np.stack([cx, cy, cz], 1)
    verts = (sv[None, :, :] * r[:, None, None] + centres[:, None, :]).reshape(-1, 3)
    tris = (sf[None, :, :].astype(np.int64) + (np.arange(n_spheres, dtype=np.int64) * len(sv))[:, None, None])
    return verts.astype(np.float32), tris.reshape(-1, 3).astype(np.uint32)


def _value_noise(x, y, seed, stream):
    """Smooth lattice noise in [0, 1): hashed corners, smoothstep blend."""
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx, fy = x - x0, y - y0
    ix, iy = x0.astype(np.int64) & 0xFFFF, y0.astype(np.int64) & 0xFFFF

    def corner(dx, dy):
        key = (np.uint64(seed) << np.uint64(40)) ^ (np.uint64(stream) << np.uint64(32)) ^ \
              ((((ix + dx) & 0xFFFF) << 16) | ((iy + dy) & 0xFFFF)).astype(np.uint64)
        return _hash_u32(key).astype(np.float64) / 4294967296.0

    sx = fx * fx * (3.0 - 2.0 * fx)
    sy = fy * fy * (3.0 - 2.0 * fy)
    a = corner(0, 0) * (1.0 - sx) + corner(1, 0) * sx
    b = corner(0, 1) * (1.0 - sx) + corner(1, 1) * sx
    return a * (1.0 - sy) + b * sy


def terrain(n=700, seed=2, extent=100.0, height=18.0, n_blobs=16, blob_subdivisions=3, closed_box=False):
    """Procedural height field of n*n*2 triangles over [-extent, extent]^2 (y up),
    plus `n_blobs` displaced icospheres resting on it; optionally a closed,
    inward-facing box around everything so that paths cannot escape."""
    g = np.arange(n + 1, dtype=np.float64)
    gx, gz = np.meshgrid(g, g, indexing="xy")
    u, w = gx / n, gz / n
    h = np.zeros_like(u)
    amp, freq = 1.0, 4.0
    for octave in range(5):
        h += amp * (_value_noise(u * freq, w * freq, seed, 10 + octave) - 0.5)
        amp *= 0.5
        freq *= 2.0
    px = (u * 2.0 - 1.0) * extent
    pz = (w * 2.0 - 1.0) * extent
    py = h * height
    verts = [np.stack([px.ravel(), py.ravel(), pz.ravel()], 1)]
    i = (np.arange(n, dtype=np.int64)[None, :] + np.arange(n, dtype=np.int64)[:, None] * (n + 1))
    a, b, c, d = i, i + 1, i + (n + 1), i + (n + 2)
    # upward-facing winding: (a, c, b) and (b, c, d)
    tris = [np.stack([a.ravel(), c.ravel(), b.ravel()], 1), np.stack([b.ravel(), c.ravel(), d.ravel()], 1)]
    offset = (n + 1) * (n + 1)
    if n_blobs:
        sv, sf = icosphere(blob_subdivisions)
        bx = (_uniform(seed, 20, n_blobs) * 1.6 - 0.8) * extent
        bz = (_uniform(seed, 21, n_blobs) * 1.6 - 0.8) * extent
        br = 3.0 + _uniform(seed, 22, n_blobs) * 6.0
        for k in range(n_blobs):
            bump = 1.0 + 0.25 * (_value_noise(sv[:, 0] * 3.0 + 7.0, (sv[:, 1] + sv[:, 2]) * 3.0 + 7.0, seed, 30 + k) - 0.5)
            p = sv * (br[k] * bump)[:, None]
            p[:, 0] += bx[k]
            p[:, 2] += bz[k]
            p[:, 1] += height * 0.2 + br[k]
            verts.append(p)
            tris.append(sf.astype(np.int64) + offset)
            offset += len(sv)
    if closed_box:
        cv, cf = cube_mesh(inward=True)
        cv = cv * np.array([extent * 1.05, extent * 0.6, extent * 1.05]) + np.array([0.0, extent * 0.45, 0.0])
        verts.append(cv)
        tris.append(cf.astype(np.int64) + offset)
        offset += len(cv)
    return np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.uint32)


def _small_glass_cube(scale, translation):
    cv, cf = cube_mesh()
    return cv.astype(np.float32), cf, affine_from_srt((scale,) * 3, (0, 0, 0, 1), translation)


def sphere_field_scene(width=1920, height=1080, n_spheres=10000, seed=1, subdivisions=2, depth=5, threads=8,
                       rmin=0.3, rmax=1.0, leaf_max=16, builder="host", device=0):
    """10k tessellated spheres under the sky, all in instance 1 (the diffuse
    NormalMaterial); instance 0 is a small glass cube at the centre."""
    cv, cf, ct = _small_glass_cube(2.0, (0.0, 0.0, 0.0))
    sv, sf = sphere_field(n_spheres, seed, subdivisions, rmin=rmin, rmax=rmax)
    ident = affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    scene = (SceneBuilder(device).add_mesh(cv, cf, [ct])
             .add_mesh(sv, sf, [ident], bvh_params(leaf_max=leaf_max, threads=threads), builder=builder).build())
    cam = make_cam(width, height, affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0.0, 0.0, -115.0)), depth=depth)
    return scene, cam


def terrain_scene(width=1920, height=1080, n=700, seed=2, depth=10, closed_box=False, threads=8, light=True,
                  leaf_max=16, builder="host", device=0):
    """~1M-triangle procedural terrain as instance 1 (diffuse), a glass cube as
    instance 0 and, with `light`, an emissive sphere as instance 2."""
    cv, cf, ct = _small_glass_cube(6.0, (10.0, 22.0, 5.0))
    tv, tf = terrain(n=n, seed=seed, closed_box=closed_box)
    ident = affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    b = SceneBuilder(device).add_mesh(cv, cf, [ct]).add_mesh(tv, tf, [ident], bvh_params(leaf_max=leaf_max, threads=threads),
                                                        builder=builder)
    if light:
        lv, lf = icosphere(2)
        b.add_mesh(lv.astype(np.float32), lf, [affine_from_srt((12.0,) * 3, (0, 0, 0, 1), (-30.0, 45.0, 20.0))])
    scene = b.build()
    cam = make_cam(width, height, look_at((0.0, 55.0, -95.0), (0.0, 0.0, 0.0)), depth=depth)
    return scene, cam
