"""A second, independent restatement of the small arithmetic pieces of the
reference's trace path, in NumPy float32 scalars and Python integers.  It was
written from the Rust sources, not from the C oracle, so that two restatements
agreeing bit for bit is evidence about the reading of the reference rather than
about one implementation.

  get_seed / xor_shift / rand_float   shader/src/modules/mod.rs:9-49
  acos_approx                         shared/src/lib.rs:47-57
  hits_bounding                       shader/src/modules/trace.rs:208-229
  triangle_ray_intersect              shader/src/modules/hit.rs:137-191
  is_inf                              shader/src/modules/mod.rs:59-62
"""
import numpy as np

F = np.float32
M32 = 0xFFFFFFFF
EPS = F(1.1920928955078125e-07)
INF = F(np.inf)
PI = F(3.14159274101257324)


def bits(x):
    return int(np.asarray(x, dtype=np.float32).view(np.uint32))


def get_seed(frame, x, y, r, g, b):
    h = 0
    h ^= (frame * 0x9E3779B9) & M32
    h ^= (x * 0x85EBCA77) & M32
    h ^= (y * 0xC2B2AE3D) & M32
    h ^= (bits(r) * 0x27D4EB2F) & M32
    h ^= (bits(g) * 0x165667B1) & M32
    h ^= (bits(b) * 0x7F4A7C15) & M32
    h ^= h >> 16
    h = (h * 0x7FEB352D) & M32
    h ^= h >> 15
    h = (h * 0x846CA68B) & M32
    h ^= h >> 16
    return h


def xor_shift(seed):
    x = seed & M32
    x ^= (x << 13) & M32
    x ^= x >> 17
    x ^= (x << 5) & M32
    return x


def rand_float(seed, lo, hi):
    """returns (new_seed, value)"""
    num = xor_shift(seed)
    with np.errstate(all="ignore"):
        v = F(num & 65535) / F(65535.0) * (F(hi) - F(lo)) + F(lo)
    return num, v


def acos_approx(v):
    v = F(v)
    x = np.abs(v)
    res = F(-0.155972) * x + F(1.56467)
    res = res * np.sqrt(F(1.0) - x)
    return res if v >= 0 else PI - res


def _fmax(a, b):
    """f32::max: a NaN operand is ignored"""
    if np.isnan(a):
        return b
    if np.isnan(b):
        return a
    return a if a > b else b


def _fmin(a, b):
    if np.isnan(a):
        return b
    if np.isnan(b):
        return a
    return a if a < b else b


def hits_bounding(pos, direction, bmin, bmax):
    pos, direction = np.asarray(pos, F), np.asarray(direction, F)
    bmin, bmax = np.asarray(bmin, F), np.asarray(bmax, F)
    with np.errstate(all="ignore"):
        t_min = (bmin - pos) / direction
        t_max = (bmax - pos) / direction
    t_min, t_max = list(t_min), list(t_max)
    for k in range(3):
        if t_min[k] > t_max[k]:
            t_min[k], t_max[k] = t_max[k], t_min[k]
    t_near = _fmax(t_min[0], _fmax(t_min[1], t_min[2]))
    t_far = _fmin(t_max[0], _fmin(t_max[1], t_max[2]))
    if t_near < INF and t_near < t_far and t_far > 0:
        return F(t_near)
    return INF


def dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], F)


def normalize(v):
    with np.errstate(all="ignore"):
        r = F(1.0) / np.sqrt(dot(v, v))
        return v * r


def triangle_ray_intersect(p0, p1, p2, pos, direction, t_lo, t_hi, backface_cull):
    p0, p1, p2 = (np.asarray(p, F) for p in (p0, p1, p2))
    pos, direction = np.asarray(pos, F), np.asarray(direction, F)
    t_lo, t_hi = F(t_lo), F(t_hi)
    with np.errstate(all="ignore"):
        a = p1 - p0
        b = p2 - p0
        normal = normalize(cross(a, b))
        d = -(dot(normal, p0))
        dot_prod = dot(normal, direction)
        if np.abs(dot_prod) < EPS:
            return INF
        if backface_cull and dot_prod > 0:
            return INF
        t = -(dot(normal, pos) + d) / dot(normal, direction)
        if t < t_lo or t > t_hi:
            t = INF
        hit = pos + direction * t
        for e, v in ((p1 - p0, hit - p0), (p2 - p1, hit - p1), (p0 - p2, hit - p2)):
            c = cross(e, v)
            if dot(normal, c) < 0:
                t = INF
    return F(t)


def is_inf(t):
    return bits(t) in (0x7F800000, 0xFF800000)


# ---- the materials the reference implements but never selects (material.rs:196-260, 389-530) ----
def det_sphere(lib, seed):
    """rand_vec_in_unit_sphere (material.rs:25-35) with the oracle's deterministic sin/cos"""
    seed, phi = rand_float(seed, 0.0, F(2.0) * PI)
    seed, costheta = rand_float(seed, -1.0, 1.0)
    theta = acos_approx(costheta)
    st = F(lib.orc_det_sin(theta))
    return seed, np.array([st * F(lib.orc_det_cos(phi)), st * F(lib.orc_det_sin(phi)), costheta], F)


def diffuse_ray_direction(lib, seed, normal):
    seed, v = det_sphere(lib, seed)
    res = v + normal
    if dot(res, res) < F(np.finfo(F).eps):
        return seed, normal
    return seed, normalize(res)


def reflect(d, n):
    return d - n * (F(2.0) * dot(d, n))


def lerp(a, b, s):
    """glam >= 0.25 Vec3::lerp"""
    return a * (F(1.0) - F(s)) + b * F(s)


def powi5(x):
    a = x * x
    a = a * a
    return x * a


def refractive_bxdf(ior, pos, d, n, t, seed):
    """RefractiveMaterial::bxdf, material.rs:444-486; returns (seed, new_pos, new_dir)"""
    ior = F(ior)
    with np.errstate(all="ignore"):
        front = dot(d, n) < 0
        ratio = F(1.0) / ior if front else ior
        cos_theta = dot(n, -d)
        sin_theta = np.sqrt(F(1.0) - cos_theta * cos_theta)
        r0 = (F(1.0) - ior) / (F(1.0) + ior)
        r0 = r0 * r0
        reflectance = r0 + (F(1.0) - r0) * powi5(F(1.0) - cos_theta)
        total = ratio * sin_theta > 1
        if not total:
            seed, u = rand_float(seed, 0.0, 1.0)
        if total or reflectance > u:
            nd = normalize(reflect(d, n))
        else:
            unit_in = normalize(d)
            c = _fmin(dot(-unit_in, n), F(1.0))
            perp = (unit_in + n * c) * ratio
            par = -n * np.sqrt(np.abs(F(1.0) - dot(perp, perp)))
            nd = perp + par
        return seed, pos + d * F(t), nd


def metal_bxdf(lib, color, roughness, pos, d, n, t, seed):
    """MetalMaterial, material.rs:239-259; returns (seed, new_pos, new_dir)"""
    new_ray = normalize(reflect(d, n))
    seed, rv = diffuse_ray_direction(lib, seed, n)
    return seed, pos + d * F(t), lerp(new_ray, normalize(rv), roughness)
