// Device functions of the trace path: RNG, camera ray, slab test, triangle
// test, BVH traversal, the three live materials, sky, accumulate.
//
// Arithmetic contract (what makes the hit buffers bit-exact): IEEE binary32,
// no FMA contraction (this file is compiled with -ffp-contract=off and the
// pragma below), correctly rounded '/' and sqrt (hipcc's default), and the
// operation order of the reference's Rust + glam 0.30.5:
//   shader/src/modules/mod.rs      get_seed, xor_shift, rand_float, is_inf
//   shader/src/modules/trace.rs    camera ray, trace_ray, get_color, hits_bounding
//   shader/src/modules/hit.rs      hit_bvh, triangle_ray_intersect
//   shader/src/modules/material.rs Generic / Normal / Emmissive / Background
//   shader/src/lib.rs              accumulate + present
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.hpp"
#include "fastdiv.hpp"
#include "fastexact.hpp"

#pragma clang fp contract(off)

namespace rt {
namespace dev {

#define RT_DEV __device__ __forceinline__

constexpr float kEps = 1.1920928955078125e-07f;  // f32::EPSILON
constexpr float kPi = 3.14159274101257324f;      // core::f32::consts::PI
#define RT_INF __builtin_huge_valf()

struct V3 {
    float x, y, z;
};

RT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// sqrt and 1/x, correctly rounded: the short forms of fastexact.hpp when every active lane's
// operand is in their checked range, the plain operators for the whole wave otherwise
RT_DEV float exact_sqrt(float x) {
    if (__ballot(!fx_ok(x)) == 0ull) return fx_sqrt(x);
    return __builtin_sqrtf(x);
}
RT_DEV float exact_rcp(float x) {
    if (__ballot(!fx_ok(fabsf(x))) == 0ull) return fx_rcp(x);
    return 1.0f / x;
}
RT_DEV float length(V3 a) { return exact_sqrt(dot(a, a)); }
// glam Vec3::normalize: self * length().recip()
RT_DEV V3 normalize(V3 a) {
    const float l2 = dot(a, a);
    float r;
    if (__ballot(!fx_ok(l2)) == 0ull) r = fx_rcp(fx_sqrt(l2));  // the root of an in-range number is in range
    else r = 1.0f / __builtin_sqrtf(l2);
    return a * r;
}
RT_DEV V3 lerp(V3 a, V3 b, float s) { return a * (1.0f - s) + b * s; }
RT_DEV V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }

RT_DEV bool is_inf_bits(float v) {
    uint32_t b = __float_as_uint(v);
    return (b & 0x7FFFFFFFu) == 0x7F800000u;
}

// ---- RNG: shader/src/modules/mod.rs:9-49 --------------------------------
RT_DEV uint32_t get_seed(uint32_t frame, uint32_t x, uint32_t y, float r, float g, float b) {
    uint32_t h = 0;
    h ^= frame * 0x9E3779B9u;
    h ^= x * 0x85EBCA77u;
    h ^= y * 0xC2B2AE3Du;
    h ^= __float_as_uint(r) * 0x27D4EB2Fu;
    h ^= __float_as_uint(g) * 0x165667B1u;
    h ^= __float_as_uint(b) * 0x7F4A7C15u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

RT_DEV float rand_float(uint32_t& seed, float lo, float hi) {
    uint32_t x = seed;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    seed = x;
    return fx_div_65535((float)(x & 65535u)) * (hi - lo) + lo;
}

// shared/src/lib.rs:47-57
RT_DEV float acos_approx(float v) {
    float x = fabsf(v);
    float res = -0.155972f * x + 1.56467f;
    res *= exact_sqrt(1.0f - x);
    return v >= 0.0f ? res : kPi - res;
}

// ---- sin / cos -------------------------------------------------------------
// Cody-Waite reduction by pi/2 and the single-precision Cephes minimax
// polynomials, written with + - * only in a fixed order so that the CPU
// checker can evaluate the very same expression tree.  Domain |x| < 2^15.
RT_DEV void det_reduce(float x, float& r, int& q) {
    float fk = x * 0.636619772f;
    int k = (int)(fk + (fk >= 0.0f ? 0.5f : -0.5f));
    float kf = (float)k;
    float t = x - kf * 1.5703125f;
    t = t - kf * 4.837512969970703125e-4f;
    t = t - kf * 7.54978995489188216e-8f;
    r = t;
    q = k & 3;
}
RT_DEV float det_sin_poly(float r) {
    float z = r * r;
    float p = -1.9515295891e-4f;
    p = p * z + 8.3321608736e-3f;
    p = p * z + -1.6666654611e-1f;
    return r + r * (z * p);
}
RT_DEV float det_cos_poly(float r) {
    float z = r * r;
    float p = 2.443315711809948e-5f;
    p = p * z + -1.388731625493765e-3f;
    p = p * z + 4.166664568298827e-2f;
    return (1.0f - 0.5f * z) + (z * z) * p;
}
RT_DEV float det_sin(float x) {
    float r;
    int q;
    det_reduce(x, r, q);
    float s = det_sin_poly(r), c = det_cos_poly(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
RT_DEV float det_cos(float x) {
    float r;
    int q;
    det_reduce(x, r, q);
    float s = det_sin_poly(r), c = det_cos_poly(r);
    float v = (q & 1) ? -s : c;
    return (q & 2) ? -v : v;
}

// ---- glam Affine3A on 12 packed floats (X, Y, Z, T) ---------------------------
RT_DEV V3 xf_vector(const float* m, V3 v) { return (ld3(m) * v.x + ld3(m + 3) * v.y) + ld3(m + 6) * v.z; }
RT_DEV V3 xf_point(const float* m, V3 v) { return xf_vector(m, v) + ld3(m + 9); }

struct Ray {
    V3 pos, dir;
};

// ---- camera: trace.rs:25-40 (tan hoisted to the host) ---------------------------
RT_DEV Ray camera_ray(const DevCam& cam, float px, float py) {
    // px = x + jitter is 0 or at least 1/65535 and at most W + 1: inside the numerator range of div_by_rcp
    // whenever the divisor is (cam.fast_px)
    float qx, qy;
    if (cam.fast_px != 0u) {
        qx = div_by_rcp(px, cam.fw, cam.rcp_fw);
        qy = div_by_rcp(py, cam.fh, cam.rcp_fh);
    } else {
        qx = px / cam.fw;
        qy = py / cam.fh;
    }
    float fx = qx * 2.0f - 1.0f;
    float fy = qy * 2.0f - 1.0f;
    float fy_adj = fy * cam.aspect;
    V3 o = v3(fx * cam.edge_dist, fy_adj * cam.edge_dist, 1.0f);
    Ray r;
    r.dir = (ld3(cam.X) * o.x + ld3(cam.Y) * o.y) + ld3(cam.Z) * o.z;
    r.pos = ld3(cam.pos);  // transform_point3(0), the same for every ray: formed once on the host
    return r;
}

// ---- slab test: trace.rs:208-229 --------------------------------------------------
RT_DEV float slab_finish(float nx, float ny, float nz, float fx, float fy, float fz) {
    bool sx = nx > fx, sy = ny > fy, sz = nz > fz;
    float ax = sx ? fx : nx, bx = sx ? nx : fx;
    float ay = sy ? fy : ny, by = sy ? ny : fy;
    float az = sz ? fz : nz, bz = sz ? nz : fz;
    float t_near = fmaxf(ax, fmaxf(ay, az));
    float t_far = fminf(bx, fminf(by, bz));
    return (t_near < RT_INF && t_near < t_far && t_far > 0.0f) ? t_near : RT_INF;
}

RT_DEV float hits_bounding(V3 pos, V3 dir, float4 lo, float4 hi) {
    float nx = (lo.x - pos.x) / dir.x, ny = (lo.y - pos.y) / dir.y, nz = (lo.z - pos.z) / dir.z;
    float fx = (hi.x - pos.x) / dir.x, fy = (hi.y - pos.y) / dir.y, fz = (hi.z - pos.z) / dir.z;
    return slab_finish(nx, ny, nz, fx, fy, fz);
}

// Reciprocals of one traversal's ray direction.  `ok` says that origin and
// direction satisfy the range conditions of div_by_rcp (fastdiv.hpp); the
// condition on the box coordinates is a property of the scene (DevScene::fast_div).
struct DirRcp {
    float x, y, z;
    bool ok;
};
RT_DEV bool dir_rcp_ok(V3 pos, V3 dir, bool scene_ok) {
    // bitwise on purpose: six cheap tests, no branches
    return (int(scene_ok) & int(div_den_ok(dir.x)) & int(div_den_ok(dir.y)) & int(div_den_ok(dir.z)) &
            int(div_term_ok(pos.x)) & int(div_term_ok(pos.y)) & int(div_term_ok(pos.z))) != 0;
}
RT_DEV DirRcp make_dir_rcp(V3 pos, V3 dir, bool scene_ok) {
    DirRcp r;
    r.ok = dir_rcp_ok(pos, dir, scene_ok);
    if (__ballot(!r.ok) == 0ull) {  // the divisor range of fastdiv.hpp lies inside fastexact.hpp's
        r.x = fx_rcp(dir.x);
        r.y = fx_rcp(dir.y);
        r.z = fx_rcp(dir.z);
    } else {
        r.x = 1.0f / dir.x;
        r.y = 1.0f / dir.y;
        r.z = 1.0f / dir.z;
    }
    return r;
}

// hits_bounding with the six quotients formed by div_by_rcp: same bits.
RT_DEV float hits_bounding_rcp(V3 pos, V3 dir, const DirRcp& rcp, float4 lo, float4 hi) {
    float nx = div_by_rcp(lo.x - pos.x, dir.x, rcp.x), ny = div_by_rcp(lo.y - pos.y, dir.y, rcp.y),
          nz = div_by_rcp(lo.z - pos.z, dir.z, rcp.z);
    float fx = div_by_rcp(hi.x - pos.x, dir.x, rcp.x), fy = div_by_rcp(hi.y - pos.y, dir.y, rcp.y),
          fz = div_by_rcp(hi.z - pos.z, dir.z, rcp.z);
    return slab_finish(nx, ny, nz, fx, fy, fz);
}

// ---- triangle: hit.rs:137-191, plane precomputed per triangle --------------------
RT_DEV float triangle_ray_intersect(float4 ta, float4 tb, float4 tc, float td, V3 pos, V3 dir, float t_lo,
                                    float t_hi, bool backface_cull) {
    V3 p0 = v3(ta.x, ta.y, ta.z), p1 = v3(tb.x, tb.y, tb.z), p2 = v3(tc.x, tc.y, tc.z);
    V3 normal = v3(ta.w, tb.w, tc.w);
    float dot_prod = dot(normal, dir);
    if (fabsf(dot_prod) < kEps) return RT_INF;
    if (backface_cull && dot_prod > 0.0f) return RT_INF;
    float t = -(dot(normal, pos) + td) / dot_prod;
    if (t < t_lo || t > t_hi) return RT_INF;  // every later step can only keep INF
    V3 hit = pos + dir * t;
    V3 c = cross(p1 - p0, hit - p0);
    if (dot(normal, c) < 0.0f) t = RT_INF;
    c = cross(p2 - p1, hit - p1);
    if (dot(normal, c) < 0.0f) t = RT_INF;
    c = cross(p0 - p2, hit - p2);
    if (dot(normal, c) < 0.0f) t = RT_INF;
    return t;
}

// ---- HitRecord: hit.rs:10-39 --------------------------------------------------------
struct HitRecord {
    uint32_t triangle_id;
    float t;
    uint32_t instance_id;
    uint32_t triangle_tests;
    uint32_t box_tests;
    uint32_t internal_visits;
};
RT_DEV void record_new(HitRecord& r) {
    r.t = RT_INF;
    r.triangle_id = 0xFFFFFFFFu;
    r.instance_id = 0;
    r.triangle_tests = 0;
    r.box_tests = 0;
    r.internal_visits = 0;
}

RT_DEV bool culled(float dist, float t_hi) { return dist == RT_INF || dist > t_hi; }

// ---- Mesh::hit_bvh: hit.rs:65-123 ----------------------------------------------------
// Same visit order, same tests, same counters as the reference's stack
// machine.  What differs is where things are kept: a popped entry names a
// child SLOT of an internal node (node*2 + slot), the node record carries both
// children's boxes, so "pop, fetch node, test its box" is one 64-byte fetch; a
// node that passes and is internal costs one 64-byte fetch that brings both
// child boxes for the near/far rule; the nearer child is tested at once (the
// reference would pop it next with nothing in between), the farther one waits
// on the stack and is tested when popped, against the clamp of that moment.
// `stack` points at this lane's column of an LDS array with `stride` entries
// between levels.
RT_DEV void load_child(const DevNode* n, uint32_t k, float4& lo, float4& hi) {
    lo = make_float4(n->lox[k], n->loy[k], n->loz[k], __uint_as_float(n->a[k]));
    hi = make_float4(n->hix[k], n->hiy[k], n->hiz[k], __uint_as_float(n->b[k]));
}

template <bool COUNT>
RT_DEV void hit_bvh(const DevScene& sc, const DevNode* root, V3 pos, V3 dir, float t_lo, HitRecord& rec,
                    uint32_t instance_id, bool backface_cull, uint32_t* stack, uint32_t stride) {
    float t_hi = rec.t;
    uint32_t sp = 0;
    float4 lo, hi;
    load_child(root, 0, lo, hi);
    if (COUNT) rec.box_tests += 1;
    bool alive = !culled(hits_bounding(pos, dir, lo, hi), t_hi);
    uint32_t a = __float_as_uint(lo.w), b = __float_as_uint(hi.w);

    for (;;) {
        if (alive && b != kInternalTag) {
            // leaf: `for i in first..=last` (hit.rs:111-121)
            if (a <= b) {
                uint32_t i = a;
                for (;;) {
                    const float4* tp = reinterpret_cast<const float4*>(&sc.tris[i]);
                    float4 ta = tp[0], tb = tp[1], tc = tp[2];
                    float td = tp[3].x;
                    float t = triangle_ray_intersect(ta, tb, tc, td, pos, dir, t_lo, t_hi, backface_cull);
                    if (!is_inf_bits(t)) {
                        t_hi = t;
                        rec.t = t;
                        rec.triangle_id = i;
                        rec.instance_id = instance_id;
                    }
                    if (i == b) break;
                    i += 1;
                }
            }
            if (COUNT) rec.triangle_tests += b - a + 1;
            alive = false;
        }
        if (alive) {
            // internal node that passed its test (hit.rs:88-107)
            const DevNode* np = &sc.nodes[a];
            float4 lo1, hi1, lo2, hi2;
            load_child(np, 0, lo1, hi1);
            load_child(np, 1, lo2, hi2);
            // BoundingBox::center (shared/src/lib.rs:155-157) and hit.rs:92-100
            V3 c1 = (v3(lo1.x, lo1.y, lo1.z) + v3(hi1.x, hi1.y, hi1.z)) * 0.5f;
            V3 c2 = (v3(lo2.x, lo2.y, lo2.z) + v3(hi2.x, hi2.y, hi2.z)) * 0.5f;
            V3 e1 = c1 - pos, e2 = c2 - pos;
            bool first_is_1 = dot(e1, e1) < dot(e2, e2);
            if (COUNT) rec.internal_visits += 1;
            stack[sp * stride] = (a << 1) | (first_is_1 ? 1u : 0u);  // the farther child's slot
            sp += 1;
            lo = first_is_1 ? lo1 : lo2;
            hi = first_is_1 ? hi1 : hi2;
        } else {
            if (sp == 0) break;
            sp -= 1;
            uint32_t ref = stack[sp * stride];
            load_child(&sc.nodes[ref >> 1], ref & 1u, lo, hi);
        }
        // the one place a box is tested: the node on top of the reference's stack
        if (COUNT) rec.box_tests += 1;
        alive = !culled(hits_bounding(pos, dir, lo, hi), t_hi);
        a = __float_as_uint(lo.w);
        b = __float_as_uint(hi.w);
    }
}

// ---- materials ---------------------------------------------------------------------
enum { STATE_ABSORB = 0, STATE_STOP = 1, STATE_RAY = 2 };

// material.rs:25-35
RT_DEV V3 rand_vec_in_unit_sphere(uint32_t& seed) {
    float phi = rand_float(seed, 0.0f, 2.0f * kPi);
    float costheta = rand_float(seed, -1.0f, 1.0f);
    float theta = acos_approx(costheta);
    float st = det_sin(theta);
    return v3(st * det_cos(phi), st * det_sin(phi), costheta);
}

// material.rs:37-47
RT_DEV V3 diffuse_ray_direction(uint32_t& seed, V3 normal) {
    V3 res = rand_vec_in_unit_sphere(seed) + normal;
    if (dot(res, res) < kEps) return normal;
    return normalize(res);
}

// GenericMaterial::reflect, material.rs:80-86
RT_DEV V3 generic_reflect(V3 in_dir, V3 normal, float roughness, uint32_t& seed) {
    float dot_product = dot(in_dir, normal);
    V3 new_ray = normalize(in_dir - normal * (2.0f * dot_product));
    V3 rand_vec = normalize(diffuse_ray_direction(seed, normal));
    return lerp(new_ray, rand_vec, roughness);
}

// glam Vec3::refract
RT_DEV V3 glam_refract(V3 i, V3 n, float eta) {
    float n_dot_i = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - n_dot_i * n_dot_i);
    if (k >= 0.0f) return i * eta - n * (eta * n_dot_i + exact_sqrt(k));
    return v3(0.0f, 0.0f, 0.0f);
}

RT_DEV float powi5(float x) {
    float a = x * x;
    a = a * a;
    return x * a;
}

// GenericMaterial::bxdf, material.rs:153-189
RT_DEV int generic_bxdf(const DevInstance& m, V3& color, Ray& ray, V3 normal, float t, uint32_t& seed) {
    Ray in_ray = ray;
    float cos_theta = dot(in_ray.dir, normal);
    float sin_theta = exact_sqrt(1.0f - cos_theta * cos_theta);
    V3 hit_pos = in_ray.pos + in_ray.dir * t;

    if (m.specular > 0.0f) {
        float rand_specular = rand_float(seed, 0.0f, 1.0f);
        if (m.specular > rand_specular) {
            ray.dir = generic_reflect(in_ray.dir, normal, m.specular_roughness, seed);
            ray.pos = hit_pos;
            return STATE_RAY;
        }
    }
    if (m.ior > 0.0f) {
        bool front_face = dot(in_ray.dir, normal) < 0.0f;
        float ior1 = front_face ? 1.0f : m.ior;
        float ior2 = front_face ? m.ior : 1.0f;
        float r0 = (ior1 - ior2) / (ior1 + ior2);
        float r0_squared = r0 * r0;
        float reflectance = r0_squared + (1.0f - r0_squared) * powi5(1.0f - fabsf(cos_theta));
        reflectance = fminf(reflectance, 1.0f);
        float rand_refract = rand_float(seed, 0.0f, 1.0f);
        float sintheta2 = ior1 / ior2 * sin_theta;
        if (rand_refract > reflectance && sintheta2 < 1.0f) {
            V3 normal_incoming = front_face ? normal : -normal;
            float refraction_ratio = front_face ? 1.0f / m.ior : m.ior;
            V3 perfect = normalize(glam_refract(in_ray.dir, normal_incoming, refraction_ratio));
            V3 diffuse = normalize(diffuse_ray_direction(seed, normal));
            V3 next = lerp(perfect, diffuse, m.roughness);
            ray.pos = hit_pos;
            ray.dir = normalize(next);
            return STATE_RAY;
        }
        ray.dir = generic_reflect(in_ray.dir, normal, m.specular_roughness, seed);
        ray.pos = hit_pos;
        return STATE_RAY;
    }
    ray.dir = generic_reflect(in_ray.dir, normal, m.roughness, seed);
    ray.pos = hit_pos;
    color = color * ld3(m.color);
    return STATE_RAY;
}

// NormalMaterial::bxdf, material.rs:262-320
RT_DEV int normal_bxdf(V3& color, Ray& ray, V3 normal, float t, uint32_t& seed) {
    Ray in_ray = ray;
    V3 direction = diffuse_ray_direction(seed, normal);
    V3 albedo;
    if (fabsf(dot(normal, v3(0.0f, 1.0f, 0.0f))) > 0.9f) albedo = v3(1.0f, 1.0f, 1.0f);
    else if (dot(normal, v3(-1.0f, 0.0f, 0.0f)) > 0.9f) albedo = v3(0.1f, 1.0f, 0.1f);
    else if (dot(normal, v3(1.0f, 0.0f, 0.0f)) > 0.9f) albedo = v3(0.1f, 0.1f, 1.0f);
    else if (dot(normal, v3(0.0f, 0.0f, -1.0f)) > 0.9f) albedo = v3(1.0f, 0.1f, 0.1f);
    else albedo = v3(0.0f, 0.0f, 0.0f);
    ray.pos = in_ray.pos + in_ray.dir * t;
    ray.dir = direction;
    color = color * albedo;
    return STATE_RAY;
}

// EmmissiveMaterial::bxdf, material.rs:340-387
RT_DEV int emissive_bxdf(const DevInstance& m, V3& color, Ray& ray, V3 normal, float t) {
    Ray in_ray = ray;
    V3 ray_reversed = -normalize(in_ray.dir);
    float dot_product = fabsf(dot(ray_reversed, normal));
    V3 stop = ld3(m.color) * exact_sqrt(dot_product);
    ray.pos = in_ray.pos + in_ray.dir * t;
    ray.dir = v3(0.0f, 0.0f, 0.0f);
    color = color * stop;
    return STATE_STOP;
}

// RefractiveMaterial::bxdf, material.rs:389-487
RT_DEV int refractive_bxdf(const DevInstance& m, Ray& ray, V3 normal, float t, uint32_t& seed) {
    Ray in_ray = ray;
    bool front_face = dot(in_ray.dir, normal) < 0.0f;
    float refraction_ratio = front_face ? 1.0f / m.ior : m.ior;
    float cos_theta = dot(normal, -in_ray.dir);
    float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
    float r0 = (1.0f - m.ior) / (1.0f + m.ior);
    r0 *= r0;
    float reflectance = r0 + (1.0f - r0) * powi5(1.0f - cos_theta);
    V3 direction;
    // the random number is drawn only when the first test fails (material.rs:451)
    if (refraction_ratio * sin_theta > 1.0f || reflectance > rand_float(seed, 0.0f, 1.0f)) {
        direction = normalize(in_ray.dir - normal * (2.0f * dot(in_ray.dir, normal)));
    } else {
        V3 unit_in = normalize(in_ray.dir);
        float c = fminf(dot(-unit_in, normal), 1.0f);
        V3 r_out_perp = (unit_in + normal * c) * refraction_ratio;
        V3 r_out_parallel = -normal * __builtin_sqrtf(fabsf(1.0f - dot(r_out_perp, r_out_perp)));
        direction = r_out_perp + r_out_parallel;
    }
    ray.pos = in_ray.pos + in_ray.dir * t;
    ray.dir = direction;
    return STATE_RAY;
}

// UVMaterial::bxdf, material.rs:489-530; uv is (0, 0) at the only call site (trace.rs:159)
RT_DEV int uv_bxdf(V3& color, Ray& ray, float t) {
    Ray in_ray = ray;
    ray.pos = in_ray.pos + in_ray.dir * t;
    ray.dir = v3(0.0f, 0.0f, 0.0f);
    color = color * v3(0.0f, 0.0f, 0.0f);
    return STATE_STOP;
}

// DiffuseMaterial, material.rs:196-225, assembled the way NormalMaterial::bxdf is
RT_DEV int diffuse_bxdf(const DevInstance& m, V3& color, Ray& ray, V3 normal, float t, uint32_t& seed) {
    Ray in_ray = ray;
    V3 direction = diffuse_ray_direction(seed, normal);
    ray.pos = in_ray.pos + in_ray.dir * t;
    ray.dir = direction;
    color = color * ld3(m.color);
    return STATE_RAY;
}

// MetalMaterial, material.rs:227-260: the same arithmetic as GenericMaterial::reflect
RT_DEV int metal_bxdf(const DevInstance& m, V3& color, Ray& ray, V3 normal, float t, uint32_t& seed) {
    Ray in_ray = ray;
    ray.dir = generic_reflect(in_ray.dir, normal, m.roughness, seed);
    ray.pos = in_ray.pos + in_ray.dir * t;
    color = color * ld3(m.color);
    return STATE_RAY;
}

// BackgroundMaterial::get_stop_color, material.rs:332-337
RT_DEV V3 sky_color(V3 ray_dir) {
    V3 temp = normalize(ray_dir);
    float factor = temp.y + 0.5f;
    if (factor < 0.0f) factor = 0.0f;
    if (factor > 1.0f) factor = 1.0f;
    return v3(1.0f, 1.0f, 1.0f) * (1.0f - factor) + v3(0.5f, 0.7f, 1.0f) * factor;
}

// The shading normal of a hit, trace.rs:140-157: the triangle's vertices moved to world space by the
// instance's transform, then normalize(cross(w0 - w1, w0 - w2)); never flipped toward the viewer.
RT_DEV V3 world_normal(const float* fwd, float4 ta, float4 tb, float4 tc) {
    V3 w0 = xf_point(fwd, v3(ta.x, ta.y, ta.z));
    V3 w1 = xf_point(fwd, v3(tb.x, tb.y, tb.z));
    V3 w2 = xf_point(fwd, v3(tc.x, tc.y, tc.z));
    return normalize(cross(w0 - w1, w0 - w2));
}

// ---- the part of trace_ray after the instance loop: trace.rs:97-172 ------------------
RT_DEV int shade(const DevScene& sc, const DevCam& cam, const HitRecord& record, Ray& self, uint32_t& seed,
                 V3& color) {
    if (cam.debug_information == 1u) {
        if (record.triangle_tests > cam.debug_number) {
            color = v3(1.0f, 0.0f, 0.0f);
        } else {
            float v = (float)record.triangle_tests / (float)cam.debug_number;
            color = v3(v, v, v);
        }
        return STATE_STOP;
    }
    if (cam.debug_information == 2u) {
        if (record.box_tests > cam.debug_number) {
            color = v3(1.0f, 0.0f, 0.0f);
        } else {
            float v = (float)record.box_tests / (float)cam.debug_number;
            color = v3(v, v, v);
        }
        return STATE_STOP;
    }
    if (record.t == RT_INF) {
        color = sky_color(self.dir);
        return STATE_STOP;
    }
    const DevInstance& inst = sc.instances[record.instance_id];
    const float4* tp = reinterpret_cast<const float4*>(&sc.tris[record.triangle_id]);
    V3 normal;
    if (inst.normals_ready != 0u) {
        // formed once per triangle by world_normal() below, for the one instance that shows this object
        const float4 td = tp[3];
        normal = v3(td.y, td.z, td.w);
    } else {
        normal = world_normal(inst.fwd, tp[0], tp[1], tp[2]);
    }
    if (inst.mat_kind == 0u) return generic_bxdf(inst, color, self, normal, record.t, seed);
    if (inst.mat_kind == 1u) return normal_bxdf(color, self, normal, record.t, seed);
    if (inst.mat_kind == 3u) return refractive_bxdf(inst, self, normal, record.t, seed);
    if (inst.mat_kind == 4u) return uv_bxdf(color, self, record.t);
    if (inst.mat_kind == 5u) return diffuse_bxdf(inst, color, self, normal, record.t, seed);
    if (inst.mat_kind == 6u) return metal_bxdf(inst, color, self, normal, record.t, seed);
    return emissive_bxdf(inst, color, self, normal, record.t);
}

// ---- the instance loop of trace_ray: trace.rs:73-95 -----------------------------------
template <bool COUNT>
RT_DEV void closest_hit(const DevScene& sc, Ray& self, HitRecord& record, uint32_t* stack, uint32_t stride) {
    self.dir = normalize(self.dir);
    record_new(record);
    for (uint32_t i = 0; i < sc.n_instances; ++i) {
        const DevInstance& inst = sc.instances[i];
        V3 lpos = xf_point(inst.inv, self.pos);
        V3 ldir = xf_vector(inst.inv, self.dir);
        hit_bvh<COUNT>(sc, &sc.objects[inst.object_id].root, lpos, ldir, kEps, record, i, inst.cull != 0u, stack,
                       stride);
    }
}

// ---- accumulate + present: shader/src/lib.rs:71-108 ------------------------------------
RT_DEV void accumulate(float frames_without_move, float4 prev, V3 c, float4& acc_out, float4& present_out) {
    bool nan = c.x > 1000.0f || c.y > 1000.0f || c.z > 1000.0f;
    float4 rendered = make_float4(c.x, c.y, c.z, 1.0f);
    float4 new_color;
    if (frames_without_move < 0.5f) {
        acc_out = rendered;
        new_color = rendered;
    } else {
        float4 acc_color;
        if (nan) {
            acc_color = make_float4(0.0f, 1000000.0f, 0.0f, 1.0f);
        } else {
            acc_color = make_float4(prev.x + rendered.x, prev.y + rendered.y, prev.z + rendered.z, prev.w + rendered.w);
        }
        acc_out = acc_color;
        float div = frames_without_move + 1.0f;
        new_color = make_float4(acc_color.x / div, acc_color.y / div, acc_color.z / div, acc_color.w / div);
    }
    // gamma 1/2.2 is only evaluated by the caller when the frame is presented
    present_out = new_color;
}

// The accumulator half of `accumulate`: what lib.rs:85-98 stores to binding 7.
RT_DEV float4 accumulate_only(float frames_without_move, float4 prev, V3 c) {
    bool nan = c.x > 1000.0f || c.y > 1000.0f || c.z > 1000.0f;
    float4 rendered = make_float4(c.x, c.y, c.z, 1.0f);
    if (frames_without_move < 0.5f) return rendered;
    if (nan) return make_float4(0.0f, 1000000.0f, 0.0f, 1.0f);
    return make_float4(prev.x + rendered.x, prev.y + rendered.y, prev.z + rendered.z, prev.w + rendered.w);
}

// The presented half: the colour lib.rs:85-98 derives from the value it has just
// stored.  It is a function of that stored value and the frame's
// frames_without_move alone, so it can be evaluated after the fact.
RT_DEV float4 present_from_acc(float frames_without_move, float4 acc) {
    if (frames_without_move < 0.5f) return acc;
    float div = frames_without_move + 1.0f;
    return make_float4(acc.x / div, acc.y / div, acc.z / div, acc.w / div);
}

RT_DEV float4 gamma_correct(float4 c) {
    const float g = 1.0f / 2.2f;
    return make_float4(powf(c.x, g), powf(c.y, g), powf(c.z, g), 1.0f);
}

}  // namespace dev
}  // namespace rt
