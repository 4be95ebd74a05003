// Host-side scene assembly: OBJ parsing, BVH construction and the merge step
// that produces the five buffers the trace kernels consume.  Mirrors
// program/src/modules/mod.rs (parse_obj_file, SceneBuilder, BufferSceneInfo)
// and program/src/modules/bvh.rs (create_bvh) of the reference.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rt {

struct BvhBuildParams {
    uint32_t root_leaf_max = 5;  // bvh.rs:17  a mesh this small stays one leaf
    uint32_t leaf_max = 16;      // bvh.rs:75, 85  a side is split only while larger
    uint32_t max_depth = 32;     // bvh.rs:5   MAX_DEPTH
    int threads = 1;
};

// program/src/modules/mod.rs:6-36
void parse_obj(const char* text, size_t len, std::vector<rt_vertex>& vertices, std::vector<rt_tri>& faces);

// What every builder entry point checks before it builds: all indices in range and every vertex
// position finite.  The reference panics on a NaN centroid (partial_cmp().unwrap(), bvh.rs:139);
// here such a mesh is refused, since a NaN key would leave the sort's comparator no strict order.
bool mesh_is_buildable(const rt_vertex* vertices, size_t n_vertices, const rt_tri* triangles, size_t n_triangles);

// program/src/modules/bvh.rs:7-28.  Re-orders `triangles` in place and returns
// the node array with LOCAL indices (node 0 is the root).
std::vector<rt_bvh> create_bvh(const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles,
                               size_t n_triangles, const BvhBuildParams& params);

// Longest root-to-leaf path, counted in nodes (a lone root leaf has depth 1).
// Returns 0 when the array is not a tree reachable from `root` (cycle, index
// out of range, more than `limit` levels).
uint32_t bvh_depth(const rt_bvh* nodes, size_t n_nodes, uint32_t root, uint32_t limit);

// program/src/modules/mod.rs:38-142
class SceneBuilder {
   public:
    SceneBuilder();
    void add_mesh(std::vector<rt_vertex> vertices, std::vector<rt_tri> tris, const rt_affine3a* transforms,
                  size_t n_transforms, const BvhBuildParams& params);
    // a mesh whose tree is already built (LOCAL indices, node 0 the root, triangles in the tree's order)
    void add_built_mesh(const std::vector<rt_vertex>& vertices, const std::vector<rt_tri>& tris,
                        const std::vector<rt_bvh>& mesh_bvh, const rt_affine3a* transforms, size_t n_transforms);
    void add_obj_file(const char* text, size_t len, const rt_affine3a* transforms, size_t n_transforms,
                      const BvhBuildParams& params);
    void sun_orientation(float x, float y, float z);
    rt_scene_info info() const;

    std::vector<rt_vertex> vertices;
    std::vector<rt_tri> tris;
    std::vector<rt_bvh> bvh;
    std::vector<rt_instance> instances;
    std::vector<rt_object> objects;
    float sun[3];
};

}  // namespace rt
