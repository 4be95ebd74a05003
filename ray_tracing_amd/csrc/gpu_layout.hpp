// Host code that turns the reference-format scene buffers (rt_abi.h) into the
// device layout of device_types.hpp, validating every index on the way so
// that no kernel can be launched on a scene that would make it read out of
// bounds or loop forever.
#pragma once

#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "device_types.hpp"

namespace rt {

struct GpuScene {
    std::vector<DevNode> nodes;  // one per INTERNAL reference node
    std::vector<DevTri> tris;    // one per triangle, same numbering as the input
    std::vector<DevTriTest> tri_tests;  // one per triangle
    std::vector<DevObject> objects;
    std::vector<DevInstance> instances;
    // which object's tree holds each triangle (kNoOwner: none), and the objects that share a triangle with
    // another object: what decides whether a triangle's world-space normal can be formed once (DevTri::wn)
    std::vector<uint32_t> tri_owner;
    std::vector<uint8_t> object_shares;
    uint32_t max_depth = 0;  // longest root-to-leaf path in nodes, over all objects
    bool fast_div = true;    // all box coordinates are in the range fastdiv.hpp asks for
};

constexpr uint32_t kNoOwner = 0xFFFFFFFFu;
constexpr uint32_t kMaxBvhDepth = 65;  // deepest tree any kernel variant has stack for

// Returns false and fills `err` when the buffers are inconsistent.
bool build_gpu_scene(const rt_scene_info& info, const rt_vertex* vertices, size_t n_vertices, const rt_tri* triangles,
                     size_t n_triangles, const rt_object* objects, size_t n_objects, const rt_instance* instances,
                     size_t n_instances, const rt_bvh* bvh, size_t n_bvh, GpuScene& out, std::string& err);

void apply_material(DevInstance& inst, const rt_material& m);
// The transform part of a DevInstance (inverse, forward, object id, copy of the object's root
// record) from an rt_instance; the material fields are left alone.
void set_instance_transform(DevInstance& di, const rt_instance& inst, const std::vector<DevObject>& objects);
// After every instance's transform part is in place: which instances the cull above the instance list is
// tried on (bound_ok is cleared for instances whose box fills the scene; pixels and counters do not change).
void choose_cull_candidates(std::vector<DevInstance>& instances);
void default_material(uint32_t instance_id, rt_material& m);
// After the instances' object ids are in place: normals_ready of every instance (the one instance of an
// object that shares no triangle with another object), and per object the index of that instance
// (kNoOwner: none) for world_normals_kernel.
void choose_normal_owners(std::vector<DevInstance>& instances, const std::vector<uint8_t>& object_shares,
                          std::vector<uint32_t>& object_instance);

}  // namespace rt
