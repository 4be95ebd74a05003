// The host half of the C ABI (include/rt_hip.h): OBJ parsing, BVH build, scene
// assembly and the camera helpers.  No GPU calls in this file (a scene builder told to use a GPU
// BVH builder calls the library's own rt_create_bvh_device / rt_create_bvh_device_sah).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/rt_hip.h"
#include "fastdiv.hpp"
#include "host_math.hpp"
#include "instance_cull.hpp"
#include "host_scene.hpp"
#include "trifilter.hpp"

using namespace rt;

struct rt_scene_builder {
    SceneBuilder sb;
    int bvh_device = -1;  // >= 0: meshes get their tree from a GPU builder on that device
    int bvh_method = RT_BVH_HOST;
};

namespace {

// One mesh into the scene, its tree by whichever builder the scene builder is set to.
int add_mesh_by_method(rt_scene_builder* b, std::vector<rt_vertex> v, std::vector<rt_tri> f,
                       const rt_affine3a* transforms, size_t n_transforms, const BvhBuildParams& params) {
    if (b->bvh_method == RT_BVH_HOST) {
        b->sb.add_mesh(std::move(v), std::move(f), transforms, n_transforms, params);
        return RT_OK;
    }
    // the reference's rule for the root: a mesh of up to root_leaf_max triangles stays one leaf (bvh.rs:17)
    const uint32_t leaf_max = f.size() <= params.root_leaf_max ? (uint32_t)f.size() : params.leaf_max;
    size_t n_nodes = 2 * f.size();
    std::vector<rt_bvh> nodes(n_nodes);
    const int rc = b->bvh_method == RT_BVH_DEVICE_LINEAR
                       ? rt_create_bvh_device(b->bvh_device, v.data(), v.size(), f.data(), f.size(), leaf_max, nodes.data(), &n_nodes)
                       : rt_create_bvh_device_sah(b->bvh_device, v.data(), v.size(), f.data(), f.size(), leaf_max,
                                                  params.max_depth + 1u, nodes.data(), &n_nodes);
    if (rc != RT_OK) return rc;
    nodes.resize(n_nodes);
    b->sb.add_built_mesh(v, f, nodes, transforms, n_transforms);
    return RT_OK;
}

BvhBuildParams to_params(const rt_bvh_params* p) {
    BvhBuildParams out;
    if (p) {
        out.root_leaf_max = p->root_leaf_max;
        out.leaf_max = p->leaf_max < 1 ? 1 : p->leaf_max;
        out.max_depth = p->max_depth < 1 ? 1 : p->max_depth;
        out.threads = p->threads;
    }
    return out;
}

}  // namespace

extern "C" {

void rt_bvh_params_default(rt_bvh_params* p) {
    if (!p) return;
    BvhBuildParams d;
    p->root_leaf_max = d.root_leaf_max;
    p->leaf_max = d.leaf_max;
    p->max_depth = d.max_depth;
    p->threads = d.threads;
}

int rt_parse_obj(const char* text, size_t len, rt_vertex* vertices, size_t* n_vertices, rt_tri* triangles,
                 size_t* n_triangles) {
    if (!text || !n_vertices || !n_triangles) return RT_ERR_INVALID;
    std::vector<rt_vertex> v;
    std::vector<rt_tri> f;
    parse_obj(text, len, v, f);
    if (vertices) {
        if (*n_vertices < v.size()) return RT_ERR_INVALID;
        if (!v.empty()) std::memcpy(vertices, v.data(), v.size() * sizeof(rt_vertex));
    }
    if (triangles) {
        if (*n_triangles < f.size()) return RT_ERR_INVALID;
        if (!f.empty()) std::memcpy(triangles, f.data(), f.size() * sizeof(rt_tri));
    }
    *n_vertices = v.size();
    *n_triangles = f.size();
    return RT_OK;
}

int rt_create_bvh(const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles, size_t n_triangles,
                  const rt_bvh_params* params, rt_bvh* nodes, size_t* n_nodes) {
    if (!vertices || !triangles || !n_nodes || n_triangles == 0) return RT_ERR_INVALID;
    if (!mesh_is_buildable(vertices, n_vertices, triangles, n_triangles)) return RT_ERR_INVALID;
    // The build sorts a copy: the caller's triangles change only together with a tree handed out, so
    // the counting call and a refused call leave them as they were (the order of equal centroids
    // depends on the order coming in, and with it the tree and its size).
    std::vector<rt_tri> sorted(triangles, triangles + n_triangles);
    std::vector<rt_bvh> out = create_bvh(vertices, n_vertices, sorted.data(), n_triangles, to_params(params));
    if (nodes) {
        if (*n_nodes < out.size()) return RT_ERR_INVALID;
        std::memcpy(nodes, out.data(), out.size() * sizeof(rt_bvh));
        std::memcpy(triangles, sorted.data(), n_triangles * sizeof(rt_tri));
    }
    *n_nodes = out.size();
    return RT_OK;
}

rt_scene_builder* rt_scene_builder_new(void) { return new rt_scene_builder; }

void rt_scene_builder_free(rt_scene_builder* b) { delete b; }

int rt_scene_builder_add_obj(rt_scene_builder* b, const char* text, size_t len, const rt_affine3a* transforms,
                             size_t n_transforms, const rt_bvh_params* params) {
    if (!b || !text || (n_transforms && !transforms)) return RT_ERR_INVALID;
    std::vector<rt_vertex> v;
    std::vector<rt_tri> f;
    parse_obj(text, len, v, f);
    if (f.empty() || !mesh_is_buildable(v.data(), v.size(), f.data(), f.size())) return RT_ERR_INVALID;
    return add_mesh_by_method(b, std::move(v), std::move(f), transforms, n_transforms, to_params(params));
}

int rt_scene_builder_use_bvh_builder(rt_scene_builder* b, int method, int device) {
    if (!b || method < RT_BVH_HOST || method > RT_BVH_DEVICE_SAH) return RT_ERR_INVALID;
    if (method != RT_BVH_HOST && device < 0) return RT_ERR_INVALID;
    b->bvh_method = method;
    b->bvh_device = method == RT_BVH_HOST ? -1 : device;
    return RT_OK;
}

int rt_scene_builder_add_mesh(rt_scene_builder* b, const rt_vertex* vertices, size_t n_vertices,
                              const rt_tri* triangles, size_t n_triangles, const rt_affine3a* transforms,
                              size_t n_transforms, const rt_bvh_params* params) {
    if (!b || !vertices || !triangles || n_triangles == 0 || (n_transforms && !transforms)) return RT_ERR_INVALID;
    if (!mesh_is_buildable(vertices, n_vertices, triangles, n_triangles)) return RT_ERR_INVALID;
    std::vector<rt_vertex> v(vertices, vertices + n_vertices);
    std::vector<rt_tri> f(triangles, triangles + n_triangles);
    return add_mesh_by_method(b, std::move(v), std::move(f), transforms, n_transforms, to_params(params));
}

int rt_scene_builder_add_built_mesh(rt_scene_builder* b, const rt_vertex* vertices, size_t n_vertices,
                                    const rt_tri* triangles, size_t n_triangles, const rt_bvh* nodes, size_t n_nodes,
                                    const rt_affine3a* transforms, size_t n_transforms) {
    if (!b || !vertices || !triangles || !nodes || n_triangles == 0 || n_nodes == 0 || (n_transforms && !transforms))
        return RT_ERR_INVALID;
    if (!mesh_is_buildable(vertices, n_vertices, triangles, n_triangles)) return RT_ERR_INVALID;
    // links must stay inside the mesh (rt_upload_scene checks the shape of the tree later)
    for (size_t i = 0; i < n_nodes; ++i) {
        const rt_bvh& nd = nodes[i];
        const size_t limit = nd.mode == RT_MODE_CHILDREN ? n_nodes : n_triangles;
        if (nd.mode != RT_MODE_CHILDREN && nd.mode != RT_MODE_TRIANGLES) return RT_ERR_INVALID;
        if (nd.child_1_or_first_tri >= limit || nd.child_2_or_last_tri >= limit) return RT_ERR_INVALID;
    }
    std::vector<rt_vertex> v(vertices, vertices + n_vertices);
    std::vector<rt_tri> f(triangles, triangles + n_triangles);
    std::vector<rt_bvh> t(nodes, nodes + n_nodes);
    b->sb.add_built_mesh(v, f, t, transforms, n_transforms);
    return RT_OK;
}

void rt_scene_builder_sun_orientation(rt_scene_builder* b, float x, float y, float z) {
    if (b) b->sb.sun_orientation(x, y, z);
}

int rt_scene_builder_build(rt_scene_builder* b, rt_scene_info* info, const rt_vertex** vertices, size_t* n_vertices,
                           const rt_tri** triangles, size_t* n_triangles, const rt_object** objects,
                           size_t* n_objects, const rt_instance** instances, size_t* n_instances,
                           const rt_bvh** bvh, size_t* n_bvh) {
    if (!b || !info) return RT_ERR_INVALID;
    *info = b->sb.info();
    if (vertices) *vertices = b->sb.vertices.data();
    if (n_vertices) *n_vertices = b->sb.vertices.size();
    if (triangles) *triangles = b->sb.tris.data();
    if (n_triangles) *n_triangles = b->sb.tris.size();
    if (objects) *objects = b->sb.objects.data();
    if (n_objects) *n_objects = b->sb.objects.size();
    if (instances) *instances = b->sb.instances.data();
    if (n_instances) *n_instances = b->sb.instances.size();
    if (bvh) *bvh = b->sb.bvh.data();
    if (n_bvh) *n_bvh = b->sb.bvh.size();
    return RT_OK;
}

int rt_upload_scene_from_builder(rt_ctx* ctx, rt_scene_builder* b) {
    if (!ctx || !b) return RT_ERR_INVALID;
    rt_scene_info info = b->sb.info();
    return rt_upload_scene(ctx, &info, b->sb.vertices.data(), b->sb.vertices.size(), b->sb.tris.data(),
                           b->sb.tris.size(), b->sb.objects.data(), b->sb.objects.size(), b->sb.instances.data(),
                           b->sb.instances.size(), b->sb.bvh.data(), b->sb.bvh.size());
}

void rt_affine_from_srt(const float scale[3], const float q[4], const float translation[3], rt_affine3a* out) {
    *out = affine_from_srt(ld(scale), q, ld(translation));
}

// glam Quat::from_rotation_{x,y,z}: (sin, cos) of half the angle
void rt_quat_from_rotation(int axis, float angle, float q[4]) {
    float h = angle * 0.5f;
    float s = std::sin(h), c = std::cos(h);
    q[0] = q[1] = q[2] = 0.0f;
    q[3] = c;
    if (axis >= 0 && axis < 3) q[axis] = s;
}

void rt_affine_inverse(const rt_affine3a* m, rt_affine3a* out) { *out = affine_inverse(*m); }

// App::update_pos, vulkan.rs:116-121: the scale and rotation survive a
// decompose / recompose round trip, so only the translation moves.
void rt_cam_update_pos(rt_cam_data* cam, const float delta[3]) {
    if (!cam) return;
    cam->transform.translation[0] += delta[0];
    cam->transform.translation[1] += delta[1];
    cam->transform.translation[2] += delta[2];
}

// App::update_mouse, vulkan.rs:95-114: yaw/pitch integrate the mouse motion,
// pitch is clamped to +-pi/2, the rotation is rebuilt as Euler YXZ and the
// old scale and translation are kept.
void rt_cam_update_mouse(rt_cam_data* cam, float orientation[2], float mouse_x, float mouse_y) {
    if (!cam || !orientation) return;
    const float pi = 3.14159274101257324f;
    orientation[0] += mouse_x * 0.006f;
    orientation[1] += -mouse_y * 0.006f;
    if (orientation[1] < -pi / 2.0f) orientation[1] = -pi / 2.0f;
    if (orientation[1] > pi / 2.0f) orientation[1] = pi / 2.0f;
    // scale = axis lengths, the first one signed by the determinant
    V3 X = ld(cam->transform.x_axis), Y = ld(cam->transform.y_axis), Z = ld(cam->transform.z_axis);
    float det = dot(Z, cross(X, Y));
    V3 scale = v3(length(X) * (det < 0.0f ? -1.0f : 1.0f), length(Y), length(Z));
    // q = qy(yaw) * qx(pitch)   (roll is 0)
    float hy = orientation[0] * 0.5f, hx = orientation[1] * 0.5f;
    float sy = std::sin(hy), cy = std::cos(hy), sx = std::sin(hx), cx = std::cos(hx);
    float q[4] = {cy * sx, sy * cx, -sy * sx, cy * cx};
    cam->transform = affine_from_srt(scale, q, ld(cam->transform.translation));
}

uint64_t rt_selftest_div_by_rcp(uint64_t n, uint64_t seed) {
    // splitmix64 stream -> operands: random sign and mantissa, exponents spread over the
    // whole admitted range, numerators formed as differences like the slab test forms them
    auto next = [&seed]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto make = [&next](int lo_exp, int hi_exp) {
        uint64_t r = next();
        uint32_t mant = uint32_t(r) & 0x7FFFFFu;
        uint32_t sign = uint32_t(r >> 63) << 31;
        int e = lo_exp + int((r >> 32) % uint64_t(hi_exp - lo_exp));  // value in [2^e, 2^(e+1))
        uint32_t bits = sign | (uint32_t(e + 127) << 23) | mant;
        float f;
        std::memcpy(&f, &bits, 4);
        return f;
    };
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        float d = make(-24, 24);
        float b = (i % 7 == 0) ? 0.0f : make(-24, 24);
        float p = (i % 11 == 0) ? 0.0f : ((i % 3 == 0) ? b * (1.0f + float(int(next() % 64) - 32) * 1.1920929e-7f) : make(-24, 24));
        if (!div_den_ok(d) || !div_term_ok(b) || !div_term_ok(p)) continue;
        float a = b - p;
        float r = 1.0f / d;
        float want = a / d;
        float got = div_by_rcp(a, d, r);
        if (a == 0.0f) {
            if (got != 0.0f) ++bad;  // a zero quotient; its sign is not part of the contract
        } else if (std::memcmp(&want, &got, 4) != 0) {
            ++bad;
        }
    }
    return bad;
}

uint64_t rt_selftest_instance_cull(uint64_t n, uint64_t seed, uint64_t* n_passed_by) {
    auto next = [&seed]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto unit = [&next]() { return float(next() >> 40) * (1.0f / 16777216.0f); };  // [0, 1)
    auto sym = [&]() { return unit() * 2.0f - 1.0f; };
    uint64_t bad = 0, passed = 0;
    rt_affine3a xf;
    float inv[12] = {0}, lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, wlo[3], whi[3];
    rt_affine3a fwd_inv;
    bool have = false;
    for (uint64_t i = 0; i < n; ++i) {
        if (i % 64 == 0) {
            // an instance: rotation about a random axis, scales up to 1 : 8 apart, a translation that may dwarf
            // the box, and an object box anywhere near the object's origin
            float q[4] = {sym(), sym(), sym(), sym()};
            const float ql = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            if (!(ql > 1e-3f)) continue;
            for (float& c : q) c /= ql;
            const float base = std::ldexp(1.0f + unit(), int(next() % 16) - 8);
            const V3 scale = v3(base * (1.0f + 7.0f * unit() * float(next() % 2)), base * (1.0f + 3.0f * unit() * float(next() % 2)),
                                (next() % 8 == 0 ? -base : base));
            const float far = std::ldexp(1.0f, int(next() % 14) - 2);
            xf = affine_from_srt(scale, q, v3(sym() * far, sym() * far, sym() * far));
            fwd_inv = affine_inverse(xf);  // what the kernels multiply with (trace.rs:87)
            std::memcpy(inv + 0, fwd_inv.x_axis, 12);
            std::memcpy(inv + 3, fwd_inv.y_axis, 12);
            std::memcpy(inv + 6, fwd_inv.z_axis, 12);
            std::memcpy(inv + 9, fwd_inv.translation, 12);
            const float ext = std::ldexp(1.0f + unit(), int(next() % 10) - 5);
            for (int k = 0; k < 3; ++k) {
                const float c = sym() * ext * float(next() % 3);
                lo[k] = c - ext * unit();
                hi[k] = c + ext * unit() * (next() % 16 == 0 ? 0.0f : 1.0f);  // now and then flat
            }
            have = instance_world_bound(inv, lo, hi, wlo, whi);
        }
        if (!have) continue;
        // a world ray: origin anywhere up to far away, aimed at a point on or just off the rim of the object
        // box (mapped to world), or anywhere; now and then parallel to a world axis
        const float reach = std::ldexp(1.0f, int(next() % 12));
        V3 p = v3(sym() * reach, sym() * reach, sym() * reach);
        V3 target_o = v3(next() % 2 ? lo[0] : hi[0], next() % 2 ? lo[1] : hi[1], lo[2] + (hi[2] - lo[2]) * unit());
        const float off = std::ldexp(sym(), -int(next() % 24));  // from far off the rim down to rounding distance
        target_o = target_o + v3(off * (hi[0] - lo[0] + 1e-3f), off * (hi[1] - lo[1] + 1e-3f), 0.0f);
        V3 target_w = transform_point3(xf, target_o);
        V3 d = (next() % 8 == 0) ? v3(sym(), sym(), sym()) : target_w - p;
        if (next() % 16 == 0) d.x = 0.0f;
        if (next() % 16 == 0) d.y = 0.0f;
        if (!(dot(d, d) > 0.0f)) continue;
        d = normalize(d);
        if (next() % 8 == 0) d = -d;  // the box behind the origin
        const float t_hi = (next() % 4 == 0) ? length(target_w - p) * (0.5f + unit()) : INFINITY;
        // the cull
        const float m_p = kCullEta * std::fmax(std::fmax(std::fabs(p.x), std::fabs(p.y)), std::fabs(p.z));
        if (!instance_clearly_missed(wlo, whi, p.x, p.y, p.z, 1.0f / d.x, 1.0f / d.y, 1.0f / d.z, m_p, t_hi)) continue;
        ++passed;
        // the reference: ray into object space (trace.rs:88-91), slab test of the root box (trace.rs:208-229), hit.rs:83
        const V3 np = transform_point3(fwd_inv, p), nd = transform_vector3(fwd_inv, d);
        const float pp[3] = {np.x, np.y, np.z}, dd[3] = {nd.x, nd.y, nd.z};
        float t_near = -INFINITY, t_far = INFINITY;
        bool first = true;
        for (int k = 0; k < 3; ++k) {
            float a = (lo[k] - pp[k]) / dd[k], b = (hi[k] - pp[k]) / dd[k];
            if (a > b) std::swap(a, b);
            t_near = first ? a : std::fmax(t_near, a);  // f32::max: a NaN operand is ignored
            t_far = first ? b : std::fmin(t_far, b);
            first = false;
        }
        const float dist = (t_near < INFINITY && t_near < t_far && t_far > 0.0f) ? t_near : INFINITY;
        const bool reference_leaves = dist == INFINITY || dist > t_hi;
        if (!reference_leaves) ++bad;
    }
    if (n_passed_by) *n_passed_by = passed;
    return bad;
}

uint64_t rt_selftest_slab_filter(uint64_t n, uint64_t seed, uint64_t* n_clear) {
    auto next = [&seed]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto unit = [&next]() { return float(next() >> 40) * (1.0f / 16777216.0f); };  // [0, 1)
    auto signed_pow2 = [&](int lo_exp, int hi_exp) {  // +-[2^lo, 2^hi), log-uniform
        float m = 1.0f + unit();
        float v = std::ldexp(m, lo_exp + int(next() % uint64_t(hi_exp - lo_exp)));
        return (next() & 1) ? -v : v;
    };
    auto nudge = [&](float v, int ulps) {
        for (int i = 0; i < (ulps < 0 ? -ulps : ulps); ++i) v = std::nextafter(v, ulps < 0 ? -INFINITY : INFINITY);
        return v;
    };
    uint64_t bad = 0, clear = 0;
    for (uint64_t i = 0; i < n; ++i) {
        float p[3], d[3], r[3], lo[3], hi[3];
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            p[k] = (next() % 8 == 0) ? 0.0f : signed_pow2(-6, 10);
            d[k] = (next() % 4 == 0) ? signed_pow2(-20, -6) : signed_pow2(-6, 1);
            r[k] = 1.0f / d[k];
            ok = ok && div_den_ok(d[k]) && div_term_ok(p[k]);
        }
        // a box around a point of the ray; some corners ON the ray (near-ties between axes),
        // some boxes flat (exact ties between the two planes of an axis)
        float s = (next() % 8 == 0) ? -std::ldexp(1.0f + unit(), int(next() % 8) - 2) : std::ldexp(1.0f + unit(), int(next() % 12) - 4);
        int kind = int(next() % 6);
        for (int k = 0; k < 3; ++k) {
            float c = p[k] + d[k] * s;
            float e0 = std::ldexp(unit(), int(next() % 10) - 6), e1 = std::ldexp(unit(), int(next() % 10) - 6);
            if (kind == 0) e0 = 0.0f;                           // the corner lo lies on the ray
            if (kind == 1) e1 = 0.0f;                           // the corner hi lies on the ray
            if (kind == 2 && k == int(i % 3)) e0 = e1 = 0.0f;   // flat along one axis, crossed by the ray
            lo[k] = c - e0;
            hi[k] = c + e1;
            if (kind == 3 && k == int(i % 3)) hi[k] = lo[k];    // flat, usually missed
            if (kind == 4) {                                    // a corner a few ulps off the ray
                lo[k] = nudge(c, int(next() % 9) - 4);
                hi[k] = lo[k] + e1;
            }
            ok = ok && div_term_ok(lo[k]) && div_term_ok(hi[k]);
        }
        if (!ok) continue;
        // the exact test, trace.rs:208-229
        float q[6];
        for (int k = 0; k < 3; ++k) {
            q[k] = (lo[k] - p[k]) / d[k];
            q[3 + k] = (hi[k] - p[k]) / d[k];
        }
        float t_near = -INFINITY, t_far = INFINITY;
        for (int k = 0; k < 3; ++k) {
            float a = q[k], b = q[3 + k];
            if (a > b) std::swap(a, b);
            t_near = std::fmax(t_near, a);
            t_far = std::fmin(t_far, b);
        }
        bool hit = t_near < INFINITY && t_near < t_far && t_far > 0.0f;
        float t_hi;
        switch (next() % 4) {
            case 0: t_hi = INFINITY; break;
            case 1: t_hi = nudge(t_near, int(next() % 17) - 8); break;
            case 2: t_hi = t_near; break;
            default: t_hi = std::ldexp(1.0f + unit(), int(next() % 14) - 5); break;
        }
        SlabApprox a = slab_approx(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], p[0], p[1], p[2], r[0], r[1], r[2]);
        if (slab_ambiguous(a, t_hi)) continue;
        ++clear;
        float dist = slab_dist(a);
        bool hit_a = dist != INFINITY;
        if (hit_a != hit) {
            ++bad;
        } else if (hit && ((dist > t_hi) != (t_near > t_hi))) {
            ++bad;
        }
    }
    if (n_clear) *n_clear = clear;
    return bad;
}

uint64_t rt_selftest_tri_filter(uint64_t n, uint64_t seed, uint64_t* n_clear) {
    auto next = [&seed]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto unit = [&next]() { return float(next() >> 40) * (1.0f / 16777216.0f); };  // [0, 1)
    auto sym = [&unit]() { return unit() * 2.0f - 1.0f; };
    auto nudge = [](float v, int ulps) {
        for (int i = 0; i < (ulps < 0 ? -ulps : ulps); ++i) v = std::nextafter(v, ulps < 0 ? -INFINITY : INFINITY);
        return v;
    };
    uint64_t bad = 0, clear = 0;
    for (uint64_t i = 0; i < n; ++i) {
        // a triangle of size `size` somewhere within `offset` of the origin
        float size = std::ldexp(1.0f, int(next() % 24) - 12), offset = std::ldexp(1.0f, int(next() % 16) - 4);
        V3 c = v3(sym() * offset, sym() * offset, sym() * offset);
        V3 p[3];
        for (int k = 0; k < 3; ++k) p[k] = c + v3(sym() * size, sym() * size, sym() * size);
        if (next() % 16 == 0) p[2] = p[1] + (p[1] - p[0]) * unit();  // a sliver
        // the plane as the scene upload computes it (hit.rs:145-148)
        V3 normal = normalize(cross(p[1] - p[0], p[2] - p[0]));
        float d = -(dot(normal, p[0]));
        float pf[3][3], nf[3] = {normal.x, normal.y, normal.z};
        for (int k = 0; k < 3; ++k) st(pf[k], p[k]);
        DevTriTest rec;
        tri_filter_build(pf[0], pf[1], pf[2], nf, d, rec);
        // a point: on an edge, at a vertex, a few ulps off either, inside, outside, far away
        int kind = int(next() % 8), k = int(next() % 3);
        V3 a = p[k], b = p[(k + 1) % 3], h;
        float s = unit();
        switch (kind) {
            case 0: h = a + (b - a) * s; break;
            case 1: h = a; break;
            case 2:
            case 3: {
                h = kind == 2 ? a + (b - a) * s : a;
                h.x = nudge(h.x, int(next() % 9) - 4);
                h.y = nudge(h.y, int(next() % 9) - 4);
                h.z = nudge(h.z, int(next() % 9) - 4);
                break;
            }
            case 4: {
                float u = unit(), v = unit() * (1.0f - u);
                h = p[0] + (p[1] - p[0]) * u + (p[2] - p[0]) * v;
                break;
            }
            case 5: h = a + (b - a) * s + normal * (sym() * size); break;
            case 6: h = c + v3(sym() * size, sym() * size, sym() * size) * 3.0f; break;
            default: h = c + v3(sym(), sym(), sym()) * (offset * 8.0f); break;
        }
        // the exact signs, hit.rs:160-188
        bool inside = true;
        for (int e = 0; e < 3; ++e) {
            V3 cr = cross(p[(e + 1) % 3] - p[e], h - p[e]);
            if (dot(normal, cr) < 0.0f) inside = false;
        }
        V3 g = h - p[0];
        float s0 = (rec.m01x[0] * g.x + rec.m01y[0] * g.y) + rec.m01z[0] * g.z;
        float s1 = (rec.m01x[1] * g.x + rec.m01y[1] * g.y) + rec.m01z[1] * g.z;
        TriSigns sg = tri_filter_signs(s0, s1, rec.c1, rec.margin_scale, rec.e_len, g.x, g.y, g.z);
        if (tri_filter_unclear(sg)) continue;
        ++clear;
        if ((sg.s_min >= 0.0f) != inside) ++bad;
    }
    if (n_clear) *n_clear = clear;
    return bad;
}

}  // extern "C"
