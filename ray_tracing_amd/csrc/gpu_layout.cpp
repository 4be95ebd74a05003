#include "gpu_layout.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

#include "fastdiv.hpp"
#include "host_math.hpp"
#include "host_scene.hpp"
#include "instance_cull.hpp"
#include "trifilter.hpp"

namespace rt {

// trace.rs:15-23 (the three constants) and trace.rs:232-244 (culling by index)
void default_material(uint32_t instance_id, rt_material& m) {
    std::memset(&m, 0, sizeof m);
    if (instance_id == 0) {
        m.kind = RT_MAT_GENERIC;
        m.backface_culling = 0;  // material.rs:191-193
        m.color[0] = m.color[1] = m.color[2] = 1.0f;
        m.ior = 1.5f;
    } else if (instance_id == 1) {
        m.kind = RT_MAT_NORMAL;
        m.backface_culling = 1;
    } else {
        m.kind = RT_MAT_EMISSIVE;  // material_id >= 2, trace.rs:165-167
        m.backface_culling = 1;
        m.color[0] = m.color[1] = m.color[2] = 15.0f;
    }
}

void apply_material(DevInstance& inst, const rt_material& m) {
    inst.mat_kind = m.kind;
    inst.cull = m.backface_culling ? 1u : 0u;
    inst.color[0] = m.color[0];
    inst.color[1] = m.color[1];
    inst.color[2] = m.color[2];
    inst.specular = m.specular;
    inst.specular_roughness = m.specular_roughness;
    inst.roughness = m.roughness;
    inst.ior = m.ior;
}

namespace {

void pack_affine(const rt_affine3a& m, float out[12]) {
    std::memcpy(out + 0, m.x_axis, 12);
    std::memcpy(out + 3, m.y_axis, 12);
    std::memcpy(out + 6, m.z_axis, 12);
    std::memcpy(out + 9, m.translation, 12);
}

// Fills child slot k of `out` from reference node `idx`.
bool make_child(const rt_bvh* bvh, size_t n_bvh, size_t n_triangles, const std::vector<uint32_t>& dense, uint32_t idx,
                DevNode& out, int k, std::string& err) {
    if (idx >= n_bvh) {
        err = "BVH child index " + std::to_string(idx) + " out of range";
        return false;
    }
    const rt_bvh& n = bvh[idx];
    out.lox[k] = n.bounding_box.min[0];
    out.loy[k] = n.bounding_box.min[1];
    out.loz[k] = n.bounding_box.min[2];
    out.hix[k] = n.bounding_box.max[0];
    out.hiy[k] = n.bounding_box.max[1];
    out.hiz[k] = n.bounding_box.max[2];
    if (n.mode == RT_MODE_CHILDREN) {
        out.a[k] = dense[idx];
        out.b[k] = kInternalTag;
    } else if (n.mode == RT_MODE_TRIANGLES) {
        uint32_t first = n.child_1_or_first_tri, last = n.child_2_or_last_tri;
        if (first == last + 1u) {  // `first..=last` with nothing in it
            out.a[k] = 1;
            out.b[k] = 0;
        } else if (first > last || last >= n_triangles) {
            err = "BVH node " + std::to_string(idx) + ": triangle range [" + std::to_string(first) + ", " +
                  std::to_string(last) + "] invalid for " + std::to_string(n_triangles) + " triangles";
            return false;
        } else {
            out.a[k] = first;
            out.b[k] = last;
        }
    } else {
        err = "BVH node " + std::to_string(idx) + ": unknown mode " + std::to_string(n.mode);
        return false;
    }
    return true;
}

}  // namespace

void set_instance_transform(DevInstance& di, const rt_instance& inst, const std::vector<DevObject>& objects) {
    rt_affine3a inv = affine_inverse(inst.transform);  // trace.rs:87, hoisted
    pack_affine(inv, di.inv);
    pack_affine(inst.transform, di.fwd);
    di.object_id = inst.object_id;
    di.root = objects[inst.object_id].root;
    const float lo[3] = {di.root.lox[0], di.root.loy[0], di.root.loz[0]}, hi[3] = {di.root.hix[0], di.root.hiy[0], di.root.hiz[0]};
    di.bound_ok = instance_world_bound(di.inv, lo, hi, di.wlo, di.whi) ? 1u : 0u;
}

// Which instances the cull above the instance list is tried on.  The world-space test costs about a third
// of what it saves when it succeeds, and it cannot succeed often for an instance whose box takes up most of
// the scene (the ground under everything, a room around everything): a ray anywhere in the scene is in or
// near that box.  Such instances keep their box but are marked "not tried" (bound_ok = 0).  An instance
// counts as filling the scene when its box spans at least 0.6 of the extent of all the boxes together along
// every axis.  Purely a matter of speed: tried or not, the pixels and counters are the same.
void choose_normal_owners(std::vector<DevInstance>& instances, const std::vector<uint8_t>& object_shares,
                          std::vector<uint32_t>& object_instance) {
    std::vector<uint32_t> shown(object_shares.size(), 0);
    for (const DevInstance& di : instances) shown[di.object_id] += 1;
    object_instance.assign(object_shares.size(), kNoOwner);
    for (size_t i = 0; i < instances.size(); ++i) {
        DevInstance& di = instances[i];
        di.normals_ready = (shown[di.object_id] == 1 && !object_shares[di.object_id]) ? 1u : 0u;
        if (di.normals_ready) object_instance[di.object_id] = (uint32_t)i;
    }
}

void choose_cull_candidates(std::vector<DevInstance>& instances) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const DevInstance& di : instances) {
        if (!di.bound_ok) continue;
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], di.wlo[k]);
            hi[k] = std::max(hi[k], di.whi[k]);
        }
    }
    for (DevInstance& di : instances) {
        if (!di.bound_ok) continue;
        bool fills = true;
        for (int k = 0; k < 3; ++k) fills = fills && (di.whi[k] - di.wlo[k]) >= 0.6f * (hi[k] - lo[k]);
        if (fills) di.bound_ok = 0;
    }
}

bool build_gpu_scene(const rt_scene_info& info, const rt_vertex* vertices, size_t n_vertices, const rt_tri* triangles,
                     size_t n_triangles, const rt_object* objects, size_t n_objects, const rt_instance* instances,
                     size_t n_instances, const rt_bvh* bvh, size_t n_bvh, GpuScene& out, std::string& err) {
    if (info.num_instances != n_instances || info.num_bvh_nodes != n_bvh || info.num_triangles != n_triangles) {
        err = "SceneInfo counts do not match the buffer lengths";
        return false;
    }
    if (n_instances == 0 || n_objects == 0 || n_bvh == 0) {
        err = "scene has no instances, objects or BVH nodes";
        return false;
    }
    if (n_triangles >= 0x7FFFFFFFull || n_bvh >= 0x7FFFFFFFull || n_vertices > 0xFFFFFFFFull) {
        err = "scene too large for 32-bit indices";
        return false;
    }
    if ((n_triangles && !triangles) || (n_vertices && !vertices) || !objects || !instances || !bvh) {
        err = "null scene buffer";
        return false;
    }

    // triangles: gather the vertices, precompute the plane (hit.rs:145-148)
    // (every triangle on its own: large scenes share the range out over host threads)
    out.tris.resize(n_triangles);
    out.tri_tests.resize(n_triangles);
    std::atomic<size_t> first_bad{n_triangles};
    auto triangle_range = [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; ++i) {
            const rt_tri& t = triangles[i];
            if (t.i0 >= n_vertices || t.i1 >= n_vertices || t.i2 >= n_vertices) {
                size_t seen = first_bad.load();
                while (i < seen && !first_bad.compare_exchange_weak(seen, i)) {
                }
                return;
            }
            V3 p0 = ld(vertices[t.i0].pos), p1 = ld(vertices[t.i1].pos), p2 = ld(vertices[t.i2].pos);
            V3 a = p1 - p0, b = p2 - p0;
            V3 normal = normalize(cross(a, b));
            float d = -(dot(normal, p0));
            DevTri& o = out.tris[i];
            st(o.p0, p0);
            st(o.p1, p1);
            st(o.p2, p2);
            o.nx = normal.x;
            o.ny = normal.y;
            o.nz = normal.z;
            o.d = d;
            o.wn[0] = o.wn[1] = o.wn[2] = 0.0f;  // world_normals_kernel fills it in where it applies
            const float nf[3] = {normal.x, normal.y, normal.z};
            tri_filter_build(o.p0, o.p1, o.p2, nf, d, out.tri_tests[i]);
        }
    };
    const size_t per_thread = 65536;
    const size_t n_workers = std::min<size_t>({(n_triangles + per_thread - 1) / per_thread,
                                               std::max(1u, std::thread::hardware_concurrency()), 16});
    if (n_workers <= 1) {
        triangle_range(0, n_triangles);
    } else {
        std::vector<std::thread> workers;
        const size_t share = (n_triangles + n_workers - 1) / n_workers;
        for (size_t w = 0; w < n_workers; ++w)
            workers.emplace_back(triangle_range, std::min(w * share, n_triangles), std::min((w + 1) * share, n_triangles));
        for (auto& th : workers) th.join();
    }
    if (first_bad.load() < n_triangles) {
        err = "triangle " + std::to_string(first_bad.load()) + " references a vertex out of range";
        return false;
    }

    // internal nodes get dense indices in input order
    out.fast_div = true;
    for (size_t i = 0; i < n_bvh; ++i)
        for (int k = 0; k < 3; ++k)
            // an inverted box also rules the shortcut out: it picks each slab's entry plane by the sign of
            // the ray direction, which is the nearer of the two only when min <= max
            if (!div_term_ok(bvh[i].bounding_box.min[k]) || !div_term_ok(bvh[i].bounding_box.max[k]) ||
                !(bvh[i].bounding_box.min[k] <= bvh[i].bounding_box.max[k]))
                out.fast_div = false;

    std::vector<uint32_t> dense(n_bvh, 0);
    uint32_t n_internal = 0;
    for (size_t i = 0; i < n_bvh; ++i)
        if (bvh[i].mode == RT_MODE_CHILDREN) dense[i] = n_internal++;
    out.nodes.resize(n_internal);
    for (size_t i = 0; i < n_bvh; ++i) {
        if (bvh[i].mode != RT_MODE_CHILDREN) continue;
        DevNode& dn = out.nodes[dense[i]];
        if (!make_child(bvh, n_bvh, n_triangles, dense, bvh[i].child_1_or_first_tri, dn, 0, err)) return false;
        if (!make_child(bvh, n_bvh, n_triangles, dense, bvh[i].child_2_or_last_tri, dn, 1, err)) return false;
    }

    out.objects.resize(n_objects);
    std::memset(out.objects.data(), 0, n_objects * sizeof(DevObject));
    out.max_depth = 0;
    for (size_t i = 0; i < n_objects; ++i) {
        if (!make_child(bvh, n_bvh, n_triangles, dense, objects[i].bvh_root, out.objects[i].root, 0, err)) return false;
        if (!make_child(bvh, n_bvh, n_triangles, dense, objects[i].bvh_root, out.objects[i].root, 1, err)) return false;
        uint32_t depth = bvh_depth(bvh, n_bvh, objects[i].bvh_root, kMaxBvhDepth);
        if (depth == 0) {
            err = "object " + std::to_string(i) + ": BVH is not a tree of at most " + std::to_string(kMaxBvhDepth) +
                  " levels";
            return false;
        }
        if (depth > out.max_depth) out.max_depth = depth;
    }

    // which object shows which triangle (every link has been checked above)
    out.tri_owner.assign(n_triangles, kNoOwner);
    out.object_shares.assign(n_objects, 0);
    {
        std::vector<uint32_t> stack;
        for (size_t obj = 0; obj < n_objects; ++obj) {
            stack.assign(1, objects[obj].bvh_root);
            while (!stack.empty()) {
                const rt_bvh& nd = bvh[stack.back()];
                stack.pop_back();
                if (nd.mode == RT_MODE_CHILDREN) {
                    stack.push_back(nd.child_1_or_first_tri);
                    stack.push_back(nd.child_2_or_last_tri);
                    continue;
                }
                if (nd.child_1_or_first_tri == nd.child_2_or_last_tri + 1u) continue;  // an empty leaf
                for (uint32_t t = nd.child_1_or_first_tri; t <= nd.child_2_or_last_tri; ++t) {
                    uint32_t& owner = out.tri_owner[t];
                    if (owner == kNoOwner) {
                        owner = (uint32_t)obj;
                    } else if (owner != obj) {
                        out.object_shares[owner] = 1;
                        out.object_shares[obj] = 1;
                    }
                }
            }
        }
    }

    out.instances.resize(n_instances);
    for (size_t i = 0; i < n_instances; ++i) {
        if (instances[i].object_id >= n_objects) {
            err = "instance " + std::to_string(i) + " references an object out of range";
            return false;
        }
        DevInstance& di = out.instances[i];
        std::memset(&di, 0, sizeof di);
        set_instance_transform(di, instances[i], out.objects);
        rt_material m;
        default_material(uint32_t(i), m);
        apply_material(di, m);
    }
    choose_cull_candidates(out.instances);
    return true;
}

}  // namespace rt
