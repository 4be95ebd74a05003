/*
 * rt_oracle.c -- CPU restatement of the reference's per-pixel trace path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED BY THE REFERENCE (see rt_oracle.h).
 *
 * Follows, function by function:
 *   shader/src/modules/mod.rs      get_seed, xor_shift, rand_float, is_inf
 *   shader/src/modules/trace.rs    camera ray, trace_ray, get_color, hits_bounding
 *   shader/src/modules/hit.rs      HitRecord, hit_bvh, triangle_ray_intersect
 *   shader/src/modules/material.rs Generic / Normal / Emmissive / Background
 *   shader/src/lib.rs              accumulate + present
 *   shared/src/lib.rs              acos_approx, BoundingBox::center
 * glam 0.30.5 (not vendored in the reference) supplies the vector algebra; its
 * published formulas are restated in the v3 / affine helpers below.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS 1.1920928955078125e-07f /* f32::EPSILON */
#define ORC_PI 3.14159274101257324f     /* core::f32::consts::PI */
#define ORC_STACK 128                   /* reference: 32 (hit.rs:73) */

typedef struct { float x, y, z; } v3;

static inline v3 v3_new(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_ld(const float* p) { return v3_new(p[0], p[1], p[2]); }
static inline void v3_st(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static inline v3 v3_add(v3 a, v3 b) { return v3_new(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_new(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_new(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_div(v3 a, v3 b) { return v3_new(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_new(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return v3_new(-a.x, -a.y, -a.z); }
/* glam Vec3::dot: (x*x' + y*y') + z*z' */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* glam Vec3::cross */
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_new(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
/* glam Vec3::normalize: self * length().recip() */
static inline v3 v3_normalize(v3 a) {
    float r = 1.0f / v3_length(a);
    return v3_scale(a, r);
}
/* glam Vec3::lerp (>= 0.25): self * (1 - s) + rhs * s */
static inline v3 v3_lerp(v3 a, v3 b, float s) {
    return v3_add(v3_scale(a, 1.0f - s), v3_scale(b, s));
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* shader/src/modules/mod.rs:59-62 */
static inline int is_inf_bits(float v) {
    uint32_t b = f2u(v);
    return b == 0x7F800000u || b == 0xFF800000u;
}

/* ---- RNG: shader/src/modules/mod.rs:9-49 -------------------------------- */
uint32_t orc_get_seed(uint32_t frame, uint32_t x, uint32_t y, float r, float g, float b) {
    uint32_t h = 0;
    h ^= frame * 0x9E3779B9u;
    h ^= x * 0x85EBCA77u;
    h ^= y * 0xC2B2AE3Du;
    h ^= f2u(r) * 0x27D4EB2Fu;
    h ^= f2u(g) * 0x165667B1u;
    h ^= f2u(b) * 0x7F4A7C15u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

uint32_t orc_xor_shift(uint32_t seed) {
    uint32_t x = seed;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}

float orc_rand_float(uint32_t* seed, float lo, float hi) {
    uint32_t num = orc_xor_shift(*seed);
    *seed = num;
    return (float)(num & 65535u) / 65535.0f * (hi - lo) + lo;
}

/* shared/src/lib.rs:47-57 */
float orc_acos_approx(float v) {
    float x = fabsf(v);
    float res = -0.155972f * x + 1.56467f;
    res *= sqrtf(1.0f - x);
    return v >= 0.0f ? res : ORC_PI - res;
}

/* ---- deterministic sin/cos ------------------------------------------------
 * Cody-Waite reduction by pi/2 followed by degree-7 / degree-8 minimax
 * polynomials (the single-precision Cephes coefficients).  Only + - * and
 * float<->int conversion are used, in a fixed order, so any IEEE binary32
 * machine without contraction produces the same bits.  Domain |x| < 2^15. */
static inline void det_reduce(float x, float* r, int* q) {
    float fk = x * 0.636619772f; /* 2/pi */
    int k = (int)(fk + (fk >= 0.0f ? 0.5f : -0.5f));
    float kf = (float)k;
    float t = x - kf * 1.5703125f;
    t = t - kf * 4.837512969970703125e-4f;
    t = t - kf * 7.54978995489188216e-8f;
    *r = t;
    *q = k & 3;
}
static inline float det_sin_poly(float r) {
    float z = r * r;
    float p = -1.9515295891e-4f;
    p = p * z + 8.3321608736e-3f;
    p = p * z + -1.6666654611e-1f;
    return r + r * (z * p);
}
static inline float det_cos_poly(float r) {
    float z = r * r;
    float p = 2.443315711809948e-5f;
    p = p * z + -1.388731625493765e-3f;
    p = p * z + 4.166664568298827e-2f;
    return (1.0f - 0.5f * z) + (z * z) * p;
}
float orc_det_sin(float x) {
    float r; int q;
    det_reduce(x, &r, &q);
    switch (q) {
        case 0: return det_sin_poly(r);
        case 1: return det_cos_poly(r);
        case 2: return -det_sin_poly(r);
        default: return -det_cos_poly(r);
    }
}
float orc_det_cos(float x) {
    float r; int q;
    det_reduce(x, &r, &q);
    switch (q) {
        case 0: return det_cos_poly(r);
        case 1: return -det_sin_poly(r);
        case 2: return -det_cos_poly(r);
        default: return det_sin_poly(r);
    }
}

/* ---- glam Affine3A -------------------------------------------------------- */
static inline v3 aff_vec(const rt_affine3a* m, v3 v) {
    v3 X = v3_ld(m->x_axis), Y = v3_ld(m->y_axis), Z = v3_ld(m->z_axis);
    return v3_add(v3_add(v3_scale(X, v.x), v3_scale(Y, v.y)), v3_scale(Z, v.z));
}
static inline v3 aff_point(const rt_affine3a* m, v3 v) {
    return v3_add(aff_vec(m, v), v3_ld(m->translation));
}
void orc_transform_point3(const rt_affine3a* m, const float v[3], float out[3]) {
    v3_st(out, aff_point(m, v3_ld(v)));
}
void orc_transform_vector3(const rt_affine3a* m, const float v[3], float out[3]) {
    v3_st(out, aff_vec(m, v3_ld(v)));
}
/* glam Mat3A::inverse + Affine3A::inverse (called at trace.rs:87) */
void orc_affine_inverse(const rt_affine3a* m, rt_affine3a* out) {
    v3 X = v3_ld(m->x_axis), Y = v3_ld(m->y_axis), Z = v3_ld(m->z_axis);
    v3 T = v3_ld(m->translation);
    v3 t0 = v3_cross(Y, Z), t1 = v3_cross(Z, X), t2 = v3_cross(X, Y);
    float det = v3_dot(Z, t2);
    float inv_det = 1.0f / det;
    v3 c0 = v3_scale(t0, inv_det), c1 = v3_scale(t1, inv_det), c2 = v3_scale(t2, inv_det);
    /* transpose of the matrix whose columns are c0, c1, c2 */
    v3 ix = v3_new(c0.x, c1.x, c2.x);
    v3 iy = v3_new(c0.y, c1.y, c2.y);
    v3 iz = v3_new(c0.z, c1.z, c2.z);
    v3 mt = v3_add(v3_add(v3_scale(ix, T.x), v3_scale(iy, T.y)), v3_scale(iz, T.z));
    memset(out, 0, sizeof *out);
    v3_st(out->x_axis, ix);
    v3_st(out->y_axis, iy);
    v3_st(out->z_axis, iz);
    v3_st(out->translation, v3_neg(mt));
}

/* ---- camera: trace.rs:25-40 ---------------------------------------------- */
void orc_camera_ray(const rt_cam_data* cam, float px, float py, float pos[3], float dir[3]) {
    float fov = (cam->fov * ORC_PI / 180.0f) / 2.0f;
    float edge_dist = tanf(fov);
    float fx = (px / (float)cam->canvas_width) * 2.0f - 1.0f;
    float fy = (py / (float)cam->canvas_height) * 2.0f - 1.0f;
    float fy_adj = fy * ((float)cam->canvas_height / (float)cam->canvas_width);
    v3 o = v3_new(fx * edge_dist, fy_adj * edge_dist, 1.0f);
    v3_st(dir, aff_vec(&cam->transform, o));
    v3_st(pos, aff_point(&cam->transform, v3_new(0.0f, 0.0f, 0.0f)));
}

/* ---- slab test: trace.rs:208-229 ------------------------------------------ */
static inline float hits_bounding(v3 pos, v3 dir, v3 bmin, v3 bmax) {
    v3 t_min = v3_div(v3_sub(bmin, pos), dir);
    v3 t_max = v3_div(v3_sub(bmax, pos), dir);
    float s;
    if (t_min.x > t_max.x) { s = t_min.x; t_min.x = t_max.x; t_max.x = s; }
    if (t_min.y > t_max.y) { s = t_min.y; t_min.y = t_max.y; t_max.y = s; }
    if (t_min.z > t_max.z) { s = t_min.z; t_min.z = t_max.z; t_max.z = s; }
    float t_near = fmaxf(t_min.x, fmaxf(t_min.y, t_min.z));
    float t_far = fminf(t_max.x, fminf(t_max.y, t_max.z));
    if (t_near < INFINITY && t_near < t_far && t_far > 0.0f) return t_near;
    return INFINITY;
}
float orc_hits_bounding(const float pos[3], const float dir[3], const float bmin[3], const float bmax[3]) {
    return hits_bounding(v3_ld(pos), v3_ld(dir), v3_ld(bmin), v3_ld(bmax));
}

/* ---- triangle: hit.rs:137-191 ---------------------------------------------- */
static inline float triangle_ray_intersect(v3 p0, v3 p1, v3 p2, v3 pos, v3 dir,
                                           float t_lo, float t_hi, int backface_cull) {
    v3 a = v3_sub(p1, p0);
    v3 b = v3_sub(p2, p0);
    v3 normal = v3_normalize(v3_cross(a, b));
    float d = -(v3_dot(normal, p0));
    float dot_prod = v3_dot(normal, dir);

    if (fabsf(dot_prod) < ORC_EPS) return INFINITY;
    if (backface_cull && dot_prod > 0.0f) return INFINITY;

    float t = -(v3_dot(normal, pos) + d) / v3_dot(normal, dir);
    if (t < t_lo || t > t_hi) t = INFINITY;
    v3 hit = v3_add(pos, v3_scale(dir, t));
    v3 c;

    v3 edge0 = v3_sub(p1, p0);
    v3 vp0 = v3_sub(hit, p0);
    c = v3_cross(edge0, vp0);
    if (v3_dot(normal, c) < 0.0f) t = INFINITY;

    v3 edge1 = v3_sub(p2, p1);
    v3 vp1 = v3_sub(hit, p1);
    c = v3_cross(edge1, vp1);
    if (v3_dot(normal, c) < 0.0f) t = INFINITY;

    v3 edge2 = v3_sub(p0, p2);
    v3 vp2 = v3_sub(hit, p2);
    c = v3_cross(edge2, vp2);
    if (v3_dot(normal, c) < 0.0f) t = INFINITY;

    return t;
}
float orc_triangle_ray_intersect(const float p0[3], const float p1[3], const float p2[3],
                                 const float pos[3], const float dir[3],
                                 float t_lo, float t_hi, int backface_cull) {
    return triangle_ray_intersect(v3_ld(p0), v3_ld(p1), v3_ld(p2), v3_ld(pos), v3_ld(dir),
                                  t_lo, t_hi, backface_cull);
}

/* ---- HitRecord + BVH traversal: hit.rs:10-134 ------------------------------- */
typedef struct {
    uint32_t triangle_id;
    float t;
    uint32_t instance_id;
    uint32_t triangle_tests;
    uint32_t box_tests;
    uint32_t internal_visits; /* not in the reference; feeds the byte model */
} hit_record;

static inline void record_new(hit_record* r) {
    r->t = INFINITY;
    r->triangle_id = 0xFFFFFFFFu;
    r->instance_id = 0;
    r->triangle_tests = 0;
    r->box_tests = 0;
    r->internal_visits = 0;
}

static void hit_bvh(const orc_scene* s, uint32_t bvh_root, v3 pos, v3 dir, float t_lo, float t_hi,
                    hit_record* record, uint32_t instance_id, int backface_cull) {
    uint32_t stack[ORC_STACK];
    int stack_size = 1;
    stack[0] = bvh_root;
    const rt_bvh* nodes = s->bvh;
    while (stack_size > 0) {
        const rt_bvh* node = &nodes[stack[stack_size - 1]];
        record->box_tests += 1;

        float dist = hits_bounding(pos, dir, v3_ld(node->bounding_box.min), v3_ld(node->bounding_box.max));
        if (dist == INFINITY || dist > t_hi) {
            stack_size -= 1;
            continue;
        }

        if (node->mode == RT_MODE_CHILDREN) {
            const rt_bvh* first = &nodes[node->child_1_or_first_tri];
            const rt_bvh* second = &nodes[node->child_2_or_last_tri];
            /* BoundingBox::center, shared/src/lib.rs:155-157 */
            v3 c1 = v3_scale(v3_add(v3_ld(first->bounding_box.min), v3_ld(first->bounding_box.max)), 0.5f);
            v3 c2 = v3_scale(v3_add(v3_ld(second->bounding_box.min), v3_ld(second->bounding_box.max)), 0.5f);
            v3 e1 = v3_sub(c1, pos), e2 = v3_sub(c2, pos);
            float d1 = v3_dot(e1, e1), d2 = v3_dot(e2, e2);
            uint32_t first_child, last_child;
            if (d1 < d2) {
                first_child = node->child_1_or_first_tri;
                last_child = node->child_2_or_last_tri;
            } else {
                first_child = node->child_2_or_last_tri;
                last_child = node->child_1_or_first_tri;
            }
            record->internal_visits += 1;
            stack[stack_size - 1] = last_child;
            if (stack_size >= ORC_STACK) abort(); /* the reference would be out of bounds at 32 */
            stack[stack_size] = first_child;
            stack_size += 1;
            continue;
        }

        stack_size -= 1;

        uint32_t first_triangle = node->child_1_or_first_tri;
        uint32_t last_triangle = node->child_2_or_last_tri;
        /* `for i in first..=last`: empty when first > last */
        if (first_triangle <= last_triangle) {
            uint32_t i = first_triangle;
            for (;;) {
                rt_tri tri = s->triangles[i];
                float t = triangle_ray_intersect(v3_ld(s->vertices[tri.i0].pos), v3_ld(s->vertices[tri.i1].pos),
                                                 v3_ld(s->vertices[tri.i2].pos), pos, dir, t_lo, t_hi,
                                                 backface_cull);
                if (!is_inf_bits(t)) {
                    t_hi = t;
                    record->t = t;
                    record->triangle_id = i;
                    record->instance_id = instance_id;
                }
                if (i == last_triangle) break;
                i++;
            }
        }
        record->triangle_tests += last_triangle - first_triangle + 1;
    }
}

void orc_mesh_hit(const orc_scene* s, uint32_t bvh_root, const float pos[3], const float dir[3],
                  float t_lo, float t_hi, uint32_t instance_id, int backface_cull, uint32_t rec[5]) {
    hit_record r;
    r.triangle_id = rec[0];
    r.t = u2f(rec[1]);
    r.instance_id = rec[2];
    r.triangle_tests = rec[3];
    r.box_tests = rec[4];
    r.internal_visits = 0;
    hit_bvh(s, bvh_root, v3_ld(pos), v3_ld(dir), t_lo, t_hi, &r, instance_id, backface_cull);
    rec[0] = r.triangle_id;
    rec[1] = f2u(r.t);
    rec[2] = r.instance_id;
    rec[3] = r.triangle_tests;
    rec[4] = r.box_tests;
}

/* ---- materials: trace.rs:15-23, 232-244; material.rs ------------------------ */
void orc_default_material(uint32_t instance_id, rt_material* m) {
    memset(m, 0, sizeof *m);
    if (instance_id == 0) { /* MATERIAL_0, trace.rs:15-21 */
        m->kind = RT_MAT_GENERIC;
        m->backface_culling = 0; /* material.rs:191-193 */
        m->color[0] = m->color[1] = m->color[2] = 1.0f;
        m->specular = 0.0f;
        m->specular_roughness = 0.0f;
        m->roughness = 0.0f;
        m->ior = 1.5f;
    } else if (instance_id == 1) { /* MATERIAL_1 */
        m->kind = RT_MAT_NORMAL;
        m->backface_culling = 1;
    } else { /* MATERIAL_2, also every material_id >= 2 (trace.rs:165-167) */
        m->kind = RT_MAT_EMISSIVE;
        m->backface_culling = 1;
        m->color[0] = m->color[1] = m->color[2] = 15.0f;
    }
}

typedef struct { int trig_mode; } shade_cfg;

enum { STATE_ABSORB = 0, STATE_STOP = 1, STATE_RAY = 2 };

typedef struct { v3 pos, dir; } ray_t;

/* material.rs:25-35 */
static inline v3 rand_vec_in_unit_sphere(uint32_t* seed, int trig_mode) {
    float phi = orc_rand_float(seed, 0.0f, 2.0f * ORC_PI);
    float costheta = orc_rand_float(seed, -1.0f, 1.0f);
    float theta = orc_acos_approx(costheta);
    float st, cp, sp;
    if (trig_mode == ORC_TRIG_DET) {
        st = orc_det_sin(theta); cp = orc_det_cos(phi); sp = orc_det_sin(phi);
    } else {
        st = sinf(theta); cp = cosf(phi); sp = sinf(phi);
    }
    return v3_new(st * cp, st * sp, costheta);
}

/* material.rs:37-47 */
static inline v3 diffuse_ray_direction(uint32_t* seed, v3 normal, int trig_mode) {
    v3 rand_vec = rand_vec_in_unit_sphere(seed, trig_mode);
    v3 res = v3_add(rand_vec, normal);
    if (v3_dot(res, res) < ORC_EPS) return normal;
    return v3_normalize(res);
}

/* GenericMaterial::reflect, material.rs:80-86 */
static inline v3 generic_reflect(v3 in_dir, v3 normal, float roughness, uint32_t* seed, int trig_mode) {
    float dot_product = v3_dot(in_dir, normal);
    v3 new_ray = v3_normalize(v3_sub(in_dir, v3_scale(normal, 2.0f * dot_product)));
    v3 rand_vec = v3_normalize(diffuse_ray_direction(seed, normal, trig_mode));
    return v3_lerp(new_ray, rand_vec, roughness);
}

/* glam Vec3::refract */
static inline v3 glam_refract(v3 i, v3 n, float eta) {
    float n_dot_i = v3_dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - n_dot_i * n_dot_i);
    if (k >= 0.0f) {
        return v3_sub(v3_scale(i, eta), v3_scale(n, eta * n_dot_i + sqrtf(k)));
    }
    return v3_new(0.0f, 0.0f, 0.0f);
}

/* compiler-rt __powisf2(x, 5): r = x; a = (x*x)*(x*x); r * a */
static inline float powi5(float x) {
    float r = x;
    float a = x * x;
    a = a * a;
    return r * a;
}

/* Rust f32::clamp */
static inline float clampf(float v, float lo, float hi) {
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

/* GenericMaterial::bxdf, material.rs:153-189 */
static int generic_bxdf(const rt_material* m, v3* color, ray_t* ray, v3 normal, float t, uint32_t* seed,
                        int trig_mode) {
    ray_t in_ray = *ray;
    v3 mcolor = v3_ld(m->color);
    float cos_theta = v3_dot(in_ray.dir, normal);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    v3 hit_pos = v3_add(in_ray.pos, v3_scale(in_ray.dir, t));

    if (m->specular > 0.0f) {
        float rand_specular = orc_rand_float(seed, 0.0f, 1.0f);
        if (m->specular > rand_specular) {
            /* reflect_specular, material.rs:99-111 */
            ray->dir = generic_reflect(in_ray.dir, normal, m->specular_roughness, seed, trig_mode);
            ray->pos = hit_pos;
            return STATE_RAY;
        }
    }

    if (m->ior > 0.0f) {
        float ior1, ior2;
        if (v3_dot(in_ray.dir, normal) < 0.0f) { ior1 = 1.0f; ior2 = m->ior; }
        else { ior1 = m->ior; ior2 = 1.0f; }
        /* schlick_reflectance, material.rs:93-97 */
        float r0 = (ior1 - ior2) / (ior1 + ior2);
        float r0_squared = r0 * r0;
        float reflectance = r0_squared + (1.0f - r0_squared) * powi5(1.0f - fabsf(cos_theta));
        reflectance = fminf(reflectance, 1.0f);
        float rand_refract = orc_rand_float(seed, 0.0f, 1.0f);
        /* can_refract, material.rs:88-91 */
        float sintheta2 = ior1 / ior2 * sin_theta;
        if (rand_refract > reflectance && sintheta2 < 1.0f) {
            /* refract, material.rs:126-150 */
            int front_face = v3_dot(in_ray.dir, normal) < 0.0f;
            v3 normal_incoming = front_face ? normal : v3_neg(normal);
            float refraction_ratio = front_face ? 1.0f / m->ior : m->ior;
            v3 perfect = v3_normalize(glam_refract(in_ray.dir, normal_incoming, refraction_ratio));
            v3 diffuse = v3_normalize(diffuse_ray_direction(seed, normal, trig_mode));
            v3 next = v3_lerp(perfect, diffuse, m->roughness);
            ray->pos = hit_pos;
            ray->dir = v3_normalize(next);
            return STATE_RAY;
        }
        ray->dir = generic_reflect(in_ray.dir, normal, m->specular_roughness, seed, trig_mode);
        ray->pos = hit_pos;
        return STATE_RAY;
    }

    /* reflect_regular, material.rs:113-124 */
    ray->dir = generic_reflect(in_ray.dir, normal, m->roughness, seed, trig_mode);
    ray->pos = hit_pos;
    *color = v3_mul(*color, mcolor);
    return STATE_RAY;
}

/* NormalMaterial::bxdf, material.rs:262-320 */
static int normal_bxdf(v3* color, ray_t* ray, v3 normal, float t, uint32_t* seed, int trig_mode) {
    ray_t in_ray = *ray;
    v3 direction = diffuse_ray_direction(seed, normal, trig_mode);
    v3 albedo;
    if (fabsf(v3_dot(normal, v3_new(0.0f, 1.0f, 0.0f))) > 0.9f) albedo = v3_new(1.0f, 1.0f, 1.0f);
    else if (v3_dot(normal, v3_new(-1.0f, 0.0f, 0.0f)) > 0.9f) albedo = v3_new(0.1f, 1.0f, 0.1f);
    else if (v3_dot(normal, v3_new(1.0f, 0.0f, 0.0f)) > 0.9f) albedo = v3_new(0.1f, 0.1f, 1.0f);
    else if (v3_dot(normal, v3_new(0.0f, 0.0f, -1.0f)) > 0.9f) albedo = v3_new(1.0f, 0.1f, 0.1f);
    else albedo = v3_new(0.0f, 0.0f, 0.0f);
    ray->pos = v3_add(in_ray.pos, v3_scale(in_ray.dir, t));
    ray->dir = direction;
    *color = v3_mul(*color, albedo);
    return STATE_RAY;
}

/* EmmissiveMaterial::bxdf, material.rs:340-387 */
static int emissive_bxdf(const rt_material* m, v3* color, ray_t* ray, v3 normal, float t) {
    ray_t in_ray = *ray;
    v3 ray_reversed = v3_neg(v3_normalize(in_ray.dir));
    float dot_product = fabsf(v3_dot(ray_reversed, normal));
    v3 stop = v3_scale(v3_ld(m->color), sqrtf(dot_product));
    ray->pos = v3_add(in_ray.pos, v3_scale(in_ray.dir, t));
    ray->dir = v3_new(0.0f, 0.0f, 0.0f);
    *color = v3_mul(*color, stop);
    return STATE_STOP;
}

/* RefractiveMaterial::bxdf, material.rs:389-487 */
static int refractive_bxdf(const rt_material* m, ray_t* ray, v3 normal, float t, uint32_t* seed) {
    ray_t in_ray = *ray;
    /* get_next_ray_dir, material.rs:444-462 */
    int front_face = v3_dot(in_ray.dir, normal) < 0.0f;
    float refraction_ratio = front_face ? 1.0f / m->ior : m->ior;
    float cos_theta = v3_dot(normal, v3_neg(in_ray.dir));
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    /* reflectance, material.rs:399-405 */
    float r0 = (1.0f - m->ior) / (1.0f + m->ior);
    r0 *= r0;
    float reflectance = r0 + (1.0f - r0) * powi5(1.0f - cos_theta);
    v3 direction;
    /* `||` short-circuits: no random draw under total internal reflection */
    if (refraction_ratio * sin_theta > 1.0f || reflectance > orc_rand_float(seed, 0.0f, 1.0f)) {
        /* reflect, material.rs:407-410; glam Vec3::reflect = self - 2 * self.dot(n) * n */
        direction = v3_normalize(v3_sub(in_ray.dir, v3_scale(normal, 2.0f * v3_dot(in_ray.dir, normal))));
    } else {
        /* refract, material.rs:412-432 */
        v3 unit_in = v3_normalize(in_ray.dir);
        float c = fminf(v3_dot(v3_neg(unit_in), normal), 1.0f);
        v3 r_out_perp = v3_scale(v3_add(unit_in, v3_scale(normal, c)), refraction_ratio);
        v3 r_out_parallel = v3_scale(v3_neg(normal), sqrtf(fabsf(1.0f - v3_dot(r_out_perp, r_out_perp))));
        direction = v3_add(r_out_perp, r_out_parallel);
    }
    ray->pos = v3_add(in_ray.pos, v3_scale(in_ray.dir, t));
    ray->dir = direction;
    return STATE_RAY; /* get_color returns the colour unchanged, material.rs:434-442 */
}

/* UVMaterial::bxdf, material.rs:489-530, with uv = (0, 0) (trace.rs:159) */
static int uv_bxdf(v3* color, ray_t* ray, float t) {
    ray_t in_ray = *ray;
    ray->pos = v3_add(in_ray.pos, v3_scale(in_ray.dir, t));
    ray->dir = v3_new(0.0f, 0.0f, 0.0f);
    *color = v3_mul(*color, v3_new(0.0f, 0.0f, 0.0f));
    return STATE_STOP;
}

/* DiffuseMaterial, material.rs:196-225 */
static int diffuse_bxdf(const rt_material* m, v3* color, ray_t* ray, v3 normal, float t, uint32_t* seed,
                        int trig_mode) {
    ray_t in_ray = *ray;
    v3 direction = diffuse_ray_direction(seed, normal, trig_mode);
    ray->pos = v3_add(in_ray.pos, v3_scale(in_ray.dir, t));
    ray->dir = direction;
    *color = v3_mul(*color, v3_ld(m->color));
    return STATE_RAY;
}

/* MetalMaterial, material.rs:227-260 */
static int metal_bxdf(const rt_material* m, v3* color, ray_t* ray, v3 normal, float t, uint32_t* seed,
                      int trig_mode) {
    ray_t in_ray = *ray;
    v3 new_ray = v3_normalize(v3_sub(in_ray.dir, v3_scale(normal, 2.0f * v3_dot(in_ray.dir, normal))));
    v3 rand_vec = v3_normalize(diffuse_ray_direction(seed, normal, trig_mode));
    ray->pos = v3_add(in_ray.pos, v3_scale(in_ray.dir, t));
    ray->dir = v3_lerp(new_ray, rand_vec, m->roughness);
    *color = v3_mul(*color, v3_ld(m->color));
    return STATE_RAY;
}

/* BackgroundMaterial::get_stop_color, material.rs:332-337 */
static inline v3 sky_color(v3 ray_dir) {
    v3 temp = v3_normalize(ray_dir);
    float factor = clampf(temp.y + 0.5f, 0.0f, 1.0f);
    return v3_add(v3_scale(v3_new(1.0f, 1.0f, 1.0f), 1.0f - factor), v3_scale(v3_new(0.5f, 0.7f, 1.0f), factor));
}
void orc_sky(const float dir[3], float out[3]) { v3_st(out, sky_color(v3_ld(dir))); }

/* One material response on its own, for known-answer tests: colour, ray and seed are updated in
 * place exactly as trace_ray's dispatch (trace.rs:161-172) would; returns the RayReturnState. */
static int dispatch_bxdf(const rt_material* m, v3* color, ray_t* ray, v3 normal, float t, uint32_t* seed,
                         int trig_mode) {
    switch (m->kind) {
        case RT_MAT_GENERIC: return generic_bxdf(m, color, ray, normal, t, seed, trig_mode);
        case RT_MAT_NORMAL: return normal_bxdf(color, ray, normal, t, seed, trig_mode);
        case RT_MAT_REFRACTIVE: return refractive_bxdf(m, ray, normal, t, seed);
        case RT_MAT_UV: return uv_bxdf(color, ray, t);
        case RT_MAT_DIFFUSE: return diffuse_bxdf(m, color, ray, normal, t, seed, trig_mode);
        case RT_MAT_METAL: return metal_bxdf(m, color, ray, normal, t, seed, trig_mode);
        default: return emissive_bxdf(m, color, ray, normal, t);
    }
}

int orc_bxdf(const rt_material* m, float color[3], float pos[3], float dir[3], const float normal[3], float t,
             uint32_t* seed, int trig_mode) {
    v3 c = v3_ld(color);
    ray_t r;
    r.pos = v3_ld(pos);
    r.dir = v3_ld(dir);
    int state = dispatch_bxdf(m, &c, &r, v3_ld(normal), t, seed, trig_mode);
    v3_st(color, c);
    v3_st(pos, r.pos);
    v3_st(dir, r.dir);
    return state;
}

static inline void material_for(const orc_scene* s, uint32_t instance_id, rt_material* m) {
    if (s->materials && instance_id < s->n_materials) *m = s->materials[instance_id];
    else orc_default_material(instance_id, m);
}

/* ---- Ray::trace_ray, trace.rs:65-173 ---------------------------------------- */
static int trace_ray(const orc_scene* s, const rt_cam_data* cam, ray_t* self, uint32_t* seed, v3* color,
                     int trig_mode, hit_record* out_record) {
    self->dir = v3_normalize(self->dir);
    hit_record record;
    record_new(&record);

    for (uint32_t i = 0; i < s->info.num_instances; i++) {
        const rt_instance* instance = &s->instances[i];
        const rt_object* object = &s->objects[instance->object_id];
        rt_affine3a inv;
        orc_affine_inverse(&instance->transform, &inv);
        v3 lpos = aff_point(&inv, self->pos);
        v3 ldir = aff_vec(&inv, self->dir);
        rt_material m;
        material_for(s, i, &m);
        hit_bvh(s, object->bvh_root, lpos, ldir, ORC_EPS, record.t, &record, i, (int)m.backface_culling);
    }
    *out_record = record;

    if (cam->debug_information == RT_DEBUG_TRIANGLE_INTERSECTION) {
        if (record.triangle_tests > cam->debug_number) {
            *color = v3_new(1.0f, 0.0f, 0.0f);
        } else {
            float v = (float)record.triangle_tests / (float)cam->debug_number;
            *color = v3_new(v, v, v);
        }
        return STATE_STOP;
    }
    if (cam->debug_information == RT_DEBUG_BVH_INTERSECTION) {
        if (record.box_tests > cam->debug_number) {
            *color = v3_new(1.0f, 0.0f, 0.0f);
        } else {
            float v = (float)record.box_tests / (float)cam->debug_number;
            *color = v3_new(v, v, v);
        }
        return STATE_STOP;
    }

    if (record.t == INFINITY) {
        *color = sky_color(self->dir);
        return STATE_STOP;
    }

    const rt_instance* instance = &s->instances[record.instance_id];
    rt_tri tri = s->triangles[record.triangle_id];
    v3 w0 = aff_point(&instance->transform, v3_ld(s->vertices[tri.i0].pos));
    v3 w1 = aff_point(&instance->transform, v3_ld(s->vertices[tri.i1].pos));
    v3 w2 = aff_point(&instance->transform, v3_ld(s->vertices[tri.i2].pos));
    v3 normal = v3_normalize(v3_cross(v3_sub(w0, w1), v3_sub(w0, w2)));

    rt_material m;
    material_for(s, record.instance_id, &m);
    return dispatch_bxdf(&m, color, self, normal, record.t, seed, trig_mode);
}

typedef struct {
    hit_record first;   /* record of segment 0 */
    orc_stats stats;
} path_probe;

/* ---- Ray::get_color, trace.rs:175-206 --------------------------------------- */
static v3 get_color(const orc_scene* s, const rt_cam_data* cam, uint32_t x, uint32_t y, uint32_t rng_seed,
                    int trig_mode, path_probe* pp) {
    v3 color = v3_new(1.0f, 1.0f, 1.0f);
    float jx = orc_rand_float(&rng_seed, 0.0f, 1.0f);
    float jy = orc_rand_float(&rng_seed, 0.0f, 1.0f);
    float pos[3], dir[3];
    orc_camera_ray(cam, (float)x + jx, (float)y + jy, pos, dir);
    ray_t vec = {v3_ld(pos), v3_ld(dir)};
    vec.dir = v3_normalize(vec.dir);
    if (pp) pp->stats.paths += 1;

    for (uint32_t seg = 0; seg < cam->depth; seg++) {
        hit_record rec;
        int state = trace_ray(s, cam, &vec, &rng_seed, &color, trig_mode, &rec);
        if (pp) {
            if (seg == 0) pp->first = rec;
            pp->stats.rays += 1;
            pp->stats.box_tests += rec.box_tests;
            pp->stats.internal_visits += rec.internal_visits;
            pp->stats.triangle_tests += rec.triangle_tests;
        }
        if (state != STATE_RAY) return color;
    }
    return v3_new(0.0f, 0.0f, 0.0f);
}

void orc_get_color(const orc_scene* s, const rt_cam_data* cam, uint32_t x, uint32_t y, uint32_t seed,
                   int trig_mode, float out_rgb[3]) {
    v3_st(out_rgb, get_color(s, cam, x, y, seed, trig_mode, NULL));
}

/* ---- shader main: shader/src/lib.rs:52-108 ---------------------------------- */
static void shade_pixel(const orc_scene* s, const rt_cam_data* cam, uint32_t x, uint32_t y, float* acc,
                        float* present, const orc_probe* probe, orc_stats* stats, int trig_mode) {
    uint32_t idx = x + y * cam->canvas_width;
    float* a = &acc[(size_t)idx * 4];
    float prev[4] = {a[0], a[1], a[2], a[3]};
    uint32_t seed = (trig_mode & ORC_SEED_INDEPENDENT) ? orc_get_seed(cam->frame, x, y, 0.0f, 0.0f, 0.0f)
                                                       : orc_get_seed(cam->frame, x, y, prev[0], prev[1], prev[2]);
    trig_mode &= 0xFF;

    path_probe pp;
    memset(&pp, 0, sizeof pp);
    record_new(&pp.first);
    v3 c = get_color(s, cam, x, y, seed, trig_mode, &pp);

    int nan = c.x > 1000.0f || c.y > 1000.0f || c.z > 1000.0f;
    float rendered[4] = {c.x, c.y, c.z, 1.0f};
    float new_color[4];
    if (cam->frames_without_move < 0.5f) {
        for (int k = 0; k < 4; k++) { a[k] = rendered[k]; new_color[k] = rendered[k]; }
    } else {
        float acc_color[4];
        if (nan) {
            acc_color[0] = 0.0f; acc_color[1] = 1000000.0f; acc_color[2] = 0.0f; acc_color[3] = 1.0f;
        } else {
            for (int k = 0; k < 4; k++) acc_color[k] = prev[k] + rendered[k];
        }
        float div = cam->frames_without_move + 1.0f;
        for (int k = 0; k < 4; k++) { a[k] = acc_color[k]; new_color[k] = acc_color[k] / div; }
    }
    if (present) {
        float* p = &present[(size_t)idx * 4];
        p[0] = powf(new_color[0], 1.0f / 2.2f);
        p[1] = powf(new_color[1], 1.0f / 2.2f);
        p[2] = powf(new_color[2], 1.0f / 2.2f);
        p[3] = 1.0f;
    }
    if (probe) {
        if (probe->hit_triangle) probe->hit_triangle[idx] = pp.first.triangle_id;
        if (probe->hit_instance) probe->hit_instance[idx] = pp.first.instance_id;
        if (probe->hit_t_bits) probe->hit_t_bits[idx] = f2u(pp.first.t);
        if (probe->box_tests) probe->box_tests[idx] = pp.first.box_tests;
        if (probe->triangle_tests) probe->triangle_tests[idx] = pp.first.triangle_tests;
    }
    stats->rays += pp.stats.rays;
    stats->box_tests += pp.stats.box_tests;
    stats->internal_visits += pp.stats.internal_visits;
    stats->triangle_tests += pp.stats.triangle_tests;
    stats->paths += pp.stats.paths;
}

typedef struct {
    const orc_scene* s;
    const rt_cam_data* cam;
    float* acc;
    float* present;
    const orc_probe* probe;
    orc_stats stats;
    uint32_t y0, y1;
    int tid, n_threads, trig_mode;
} worker_arg;

#define ORC_ROW_BLOCK 4

static void* worker(void* p) {
    worker_arg* w = (worker_arg*)p;
    uint32_t W = w->cam->canvas_width;
    /* interleaved blocks of rows so that every thread sees a similar mix */
    for (uint32_t yb = w->y0 + (uint32_t)w->tid * ORC_ROW_BLOCK; yb < w->y1;
         yb += (uint32_t)w->n_threads * ORC_ROW_BLOCK) {
        uint32_t ye = yb + ORC_ROW_BLOCK < w->y1 ? yb + ORC_ROW_BLOCK : w->y1;
        for (uint32_t y = yb; y < ye; y++)
            for (uint32_t x = 0; x < W; x++)
                shade_pixel(w->s, w->cam, x, y, w->acc, w->present, w->probe, &w->stats, w->trig_mode);
    }
    return NULL;
}

int orc_render_frame(const orc_scene* s, const rt_cam_data* cam, float* acc, float* present,
                     const orc_probe* probe, orc_stats* stats, uint32_t y0, uint32_t y1, int n_threads,
                     int trig_mode) {
    if (n_threads <= 0) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (y1 > cam->canvas_height) y1 = cam->canvas_height;
    worker_arg args[256];
    pthread_t th[256];
    for (int i = 0; i < n_threads; i++) {
        memset(&args[i], 0, sizeof args[i]);
        args[i].s = s; args[i].cam = cam; args[i].acc = acc; args[i].present = present;
        args[i].probe = probe; args[i].y0 = y0; args[i].y1 = y1; args[i].tid = i;
        args[i].n_threads = n_threads; args[i].trig_mode = trig_mode;
    }
    if (n_threads == 1) {
        worker(&args[0]);
    } else {
        for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &args[i]);
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    }
    if (stats) {
        for (int i = 0; i < n_threads; i++) {
            stats->rays += args[i].stats.rays;
            stats->box_tests += args[i].stats.box_tests;
            stats->internal_visits += args[i].stats.internal_visits;
            stats->triangle_tests += args[i].stats.triangle_tests;
            stats->paths += args[i].stats.paths;
        }
    }
    return 0;
}
