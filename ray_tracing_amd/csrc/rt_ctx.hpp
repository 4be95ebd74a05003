// The context behind the C ABI (include/rt_hip.h), shared by rt_ctx.cpp (one GPU) and rt_group.cpp
// (several contexts driven together).
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "gpu_layout.hpp"
#include "kernels.hpp"

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene, resident in HBM
    bool have_scene = false;
    char* d_arena = nullptr;  // nodes | triangle test records | instances
    size_t node_off = 0, tri_test_off = 0, inst_off = 0, arena_bytes = 0;
    rt::DevTri* d_tris = nullptr;
    rt::DevObject* d_objects = nullptr;
    std::vector<rt::DevInstance> h_instances;
    std::vector<rt::DevObject> h_objects;
    // world-space shading normals (DevTri::wn): each triangle's object, the objects that share triangles,
    // and per object its one instance
    uint32_t* d_tri_owner = nullptr;
    uint32_t* d_object_instance = nullptr;
    std::vector<uint8_t> h_object_shares;
    size_t n_tris = 0;
    uint32_t n_instances = 0;
    uint32_t max_depth = 0;
    bool fast_div = false;

    // frame buffers, sized for the current canvas
    uint32_t W = 0, H = 0;
    float* d_acc = nullptr;
    float* d_present = nullptr;
    bool present_valid = false;  // d_present is in step with the accumulator: zeroed with it, or written by the last render
    uint32_t* d_present8 = nullptr;  // the 8-bit image on the device, allocated on first use
    // rt_present_begin / rt_present_end: the copy of the 8-bit image to the host runs on a stream
    // of its own, so that it overlaps the next render
    uint8_t* h_present8 = nullptr;  // pinned
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_quantised = nullptr, ev_copied = nullptr;
    bool copy_pending = false;
    uint32_t* d_hit_tri = nullptr;
    uint32_t* d_hit_inst = nullptr;
    uint32_t* d_hit_t = nullptr;
    uint32_t* d_box = nullptr;
    uint32_t* d_tritests = nullptr;
    rt::DevStats* d_stats = nullptr;
    uint32_t* d_queue = nullptr;  // work-queue heads of the persistent kernel
    void* d_pool = nullptr;       // ray slots of the pooled kernel
    size_t pool_bytes = 0;
    // cost-aware tile order: costs of the last launch, the order derived from them (two buffers:
    // one read by a launch, the other written after it), and the canvas partition they belong to
    uint32_t* d_tile_cost = nullptr;
    uint32_t* d_tile_order[2] = {nullptr, nullptr};
    uint32_t tile_capacity = 0;
    int order_current = -1;  // which of d_tile_order holds a usable order; -1: none
    uint32_t order_sig[5] = {0, 0, 0, 0, 0};
    bool tile_order_enabled = true;
    int n_cus = 0;
    int vote_w[3] = {4, 5, 15};  // vote weights: control, box, triangle steps
    uint32_t pool_slots = 0;     // RT_OPT_POOL_SLOTS
    int tile_deal = -1;          // -1: by launch length; 0 / 1: RT_OPT_TILE_DEAL
    int blocks_per_cu = 0;       // 0: as many as fit; RT_OPT_BLOCKS_PER_CU
    bool force_big_offsets = false;
    bool fast_div_enabled = true;
    bool instance_cull = true;   // RT_OPT_INSTANCE_CULL
    int per_cu_cache[2][2][2][2] = {};  // occupancy answers, 0 = not asked yet

    // options
    int pipeline = 1;
    bool probe_hits = false;
    bool counters = false;
    bool present = true;
    uint32_t frame_stride = 1;
    uint32_t seed_mode = 0;
    uint32_t strip_h = 0, rank = 0, world = 1;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};


// ---- what rt_group.cpp needs of a context beyond the public entry points ----
namespace rt {
// rt_render_frames without the bookkeeping on *cam and, with `present` false, without the gamma pass
// whatever RT_OPT_PRESENT says (the group runs one pass over the assembled image instead).
int ctx_render(rt_ctx* c, const rt_cam_data& cam, uint32_t n_frames, bool present);
// Binding 8 of the whole canvas from an accumulator on this context's GPU (`acc`: the context's own
// when NULL), as after `n_frames` frames that began at `cam`: enqueued on the context's stream.
int ctx_present_whole_canvas(rt_ctx* c, const rt_cam_data& cam, uint32_t n_frames, const float* acc);
int ctx_fail(rt_ctx* c, int code, const std::string& msg);
}  // namespace rt
