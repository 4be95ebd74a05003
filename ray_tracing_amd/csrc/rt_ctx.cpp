// The GPU half of the C ABI (include/rt_hip.h): context, scene upload, render,
// read-back.  Stands in for program/src/modules/vulkan.rs of the reference.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "fastdiv.hpp"
#include "gpu_layout.hpp"
#include "host_scene.hpp"
#include "kernels.hpp"
#include "rt_ctx.hpp"

using namespace rt;

namespace {

thread_local std::string g_create_err;

int fail(rt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_create_err = msg;
    return code;
}

#define RT_HIP(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(ctx, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

template <typename T>
void free_dev(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_scene(rt_ctx* c) {
    free_dev(c->d_arena);
    free_dev(c->d_tris);
    free_dev(c->d_objects);
    free_dev(c->d_tri_owner);
    free_dev(c->d_object_instance);
    c->have_scene = false;
}

// The instance records (c->h_instances) to the device, and with them what follows from which instance shows
// which object: normals_ready, and the world-space normals of the triangles it holds for (DevTri::wn).  The
// caller has made sure that no frame in flight still reads the old records.
int upload_instances(rt_ctx* c) {
    std::vector<uint32_t> object_instance;
    choose_normal_owners(c->h_instances, c->h_object_shares, object_instance);
    RT_HIP(c, hipMemcpy(c->d_arena + c->inst_off, c->h_instances.data(), c->h_instances.size() * sizeof(DevInstance),
                        hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->d_object_instance, object_instance.data(), object_instance.size() * 4, hipMemcpyHostToDevice));
    RT_HIP(c, launch_world_normals(c->d_tris, (uint32_t)c->n_tris, c->d_tri_owner, c->d_object_instance,
                                   reinterpret_cast<const DevInstance*>(c->d_arena + c->inst_off), c->stream));
    return RT_OK;
}

void free_frame(rt_ctx* c) {
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);  // a copy may still read d_present8
    c->copy_pending = false;
    if (c->h_present8) {
        (void)hipHostFree(c->h_present8);
        c->h_present8 = nullptr;
    }
    free_dev(c->d_acc);
    free_dev(c->d_present);
    c->present_valid = false;
    free_dev(c->d_present8);
    free_dev(c->d_hit_tri);
    free_dev(c->d_hit_inst);
    free_dev(c->d_hit_t);
    free_dev(c->d_box);
    free_dev(c->d_tritests);
    c->W = c->H = 0;
}

int ensure_frame(rt_ctx* c, uint32_t W, uint32_t H) {
    if (W == 0 || H == 0 || (uint64_t)W * H > 0x7FFFFFFFull) return fail(c, RT_ERR_INVALID, "bad canvas size");
    RT_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)W * H;
    if (c->W != W || c->H != H) {
        free_frame(c);
        RT_HIP(c, hipMalloc(&c->d_acc, n * 16));
        RT_HIP(c, hipMalloc(&c->d_present, n * 16));
        // binding 7 starts zeroed: the first frame's seeds hash zeros (lib.rs:57-60)
        RT_HIP(c, hipMemsetAsync(c->d_acc, 0, n * 16, c->stream));
        RT_HIP(c, hipMemsetAsync(c->d_present, 0, n * 16, c->stream));
        c->W = W;
        c->H = H;
        c->present_valid = true;  // a fresh canvas: nothing accumulated, nothing shown
    }
    if (c->probe_hits && !c->d_hit_tri) {
        RT_HIP(c, hipMalloc(&c->d_hit_tri, n * 4));
        RT_HIP(c, hipMalloc(&c->d_hit_inst, n * 4));
        RT_HIP(c, hipMalloc(&c->d_hit_t, n * 4));
        RT_HIP(c, hipMemsetAsync(c->d_hit_tri, 0xFF, n * 4, c->stream));
        RT_HIP(c, hipMemsetAsync(c->d_hit_inst, 0, n * 4, c->stream));
        RT_HIP(c, hipMemsetAsync(c->d_hit_t, 0, n * 4, c->stream));
    }
    if (c->counters && !c->d_box) {
        RT_HIP(c, hipMalloc(&c->d_box, n * 4));
        RT_HIP(c, hipMalloc(&c->d_tritests, n * 4));
        RT_HIP(c, hipMemsetAsync(c->d_box, 0, n * 4, c->stream));
        RT_HIP(c, hipMemsetAsync(c->d_tritests, 0, n * 4, c->stream));
    }
    return RT_OK;
}

// Rows of the canvas this context owns under the strip partition.
uint32_t local_rows(uint32_t H, uint32_t strip_h, uint32_t rank, uint32_t world) {
    if (world <= 1) return H;
    uint32_t rows = 0;
    for (uint32_t s = rank; (uint64_t)s * strip_h < H; s += world) {
        uint32_t y0 = s * strip_h;
        uint32_t y1 = y0 + strip_h < H ? y0 + strip_h : H;
        rows += y1 - y0;
    }
    return rows;
}

DevCam make_cam(const rt_cam_data& cam, uint32_t frame_stride, uint32_t seed_mode) {
    DevCam d;
    std::memcpy(d.X, cam.transform.x_axis, 12);
    std::memcpy(d.Y, cam.transform.y_axis, 12);
    std::memcpy(d.Z, cam.transform.z_axis, 12);
    std::memcpy(d.T, cam.transform.translation, 12);
    // trace.rs:27-28; tan is evaluated once here instead of once per ray
    const float pi = 3.14159274101257324f;
    float fov = (cam.fov * pi / 180.0f) / 2.0f;
    d.edge_dist = std::tan(fov);
    d.fw = (float)cam.canvas_width;
    d.fh = (float)cam.canvas_height;
    d.aspect = (float)cam.canvas_height / (float)cam.canvas_width;  // trace.rs:31
    // transform_point3(0), trace.rs:37: ((X*0 + Y*0) + Z*0) + T in binary32, one rounding per operation (this
    // file is compiled without contraction and without value-changing optimisations, like the kernels)
    for (int k = 0; k < 3; ++k) {
        volatile float zx = d.X[k] * 0.0f, zy = d.Y[k] * 0.0f, zz = d.Z[k] * 0.0f;
        volatile float s1 = zx + zy;
        volatile float s2 = s1 + zz;
        d.pos[k] = s2 + d.T[k];
    }
    d.rcp_fw = 1.0f / d.fw;
    d.rcp_fh = 1.0f / d.fh;
    d.fast_px = (div_den_ok(d.fw) && div_den_ok(d.fh)) ? 1u : 0u;
    d.W = cam.canvas_width;
    d.H = cam.canvas_height;
    d.depth = cam.depth;
    d.frame = cam.frame;
    d.debug_number = cam.debug_number;
    d.debug_information = cam.debug_information;
    d.frames_without_move = cam.frames_without_move;
    d.frame_stride = frame_stride;
    d.seed_mode = seed_mode;
    return d;
}

// A launch of at least this many frames on a canvas without a learnt tile order sends its first frame ahead
constexpr uint32_t kScoutMinFrames = 4;

// Where the persistent kernel's queue heads, tile costs and tile orders live, and the host-side
// state of the cost-aware order that goes with those buffers.
struct QueueBufs {
    uint32_t* head = nullptr;
    uint32_t* cost = nullptr;
    uint32_t* order[2] = {nullptr, nullptr};
    int* order_current = nullptr;
    uint32_t* order_sig = nullptr;
};

int blocks_per_cu(rt_ctx* c, uint32_t stack_needed, bool count, bool big, bool pool) {
    int& cached = c->per_cu_cache[stack_needed > 32 ? 1 : 0][count ? 1 : 0][big ? 1 : 0][pool ? 1 : 0];
    if (cached == 0) cached = persistent_blocks_per_cu(stack_needed, count, big, pool);
    int per_cu = cached;
    if (c->blocks_per_cu >= 1 && c->blocks_per_cu < per_cu) per_cu = c->blocks_per_cu;
    return per_cu;
}

// The context's own buffers, grown to `n_tiles` when needed.
int ctx_queue_bufs(rt_ctx* c, uint32_t n_tiles, QueueBufs& qb) {
    qb.head = c->d_queue;
    if (!c->tile_order_enabled) return RT_OK;
    if (n_tiles > c->tile_capacity) {
        free_dev(c->d_tile_cost);
        free_dev(c->d_tile_order[0]);
        free_dev(c->d_tile_order[1]);
        c->tile_capacity = 0;
        c->order_current = -1;
        RT_HIP(c, hipMalloc(&c->d_tile_cost, (size_t)n_tiles * 4));
        RT_HIP(c, hipMalloc(&c->d_tile_order[0], (size_t)n_tiles * 4));
        RT_HIP(c, hipMalloc(&c->d_tile_order[1], (size_t)n_tiles * 4));
        c->tile_capacity = n_tiles;
    }
    qb.cost = c->d_tile_cost;
    qb.order[0] = c->d_tile_order[0];
    qb.order[1] = c->d_tile_order[1];
    qb.order_current = &c->order_current;
    qb.order_sig = c->order_sig;
    return RT_OK;
}

// One launch of `n_frames` frames into the buffers of `fr`.  `own_buffers`: the frame and queue
// buffers are the context's and may be (re)allocated here; otherwise `fr` and `qb` come from the
// caller and nothing is allocated.
int render(rt_ctx* c, const rt_cam_data& cam, uint32_t n_frames, const DevFrame* caller_frame,
           const QueueBufs* caller_bufs) {
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if (cam.debug_information > RT_DEBUG_BVH_INTERSECTION) return fail(c, RT_ERR_INVALID, "bad debug_information");
    const bool own_buffers = caller_frame == nullptr;
    // the work queues name a pixel slot by 32 bits: 64 slots per 8x8 tile (a canvas of one row still
    // takes a tile per 8 pixels)
    if (((uint64_t)cam.canvas_width + 7u) / 8u * (((uint64_t)cam.canvas_height + 7u) / 8u) >= (1ull << 26))
        return fail(c, RT_ERR_LIMIT, "canvas of 2^26 or more 8x8 tiles");
    if (own_buffers) {
        int rc = ensure_frame(c, cam.canvas_width, cam.canvas_height);
        if (rc != RT_OK) return rc;
    }
    if (n_frames == 0) return RT_OK;

    DevScene sc;
    sc.nodes = reinterpret_cast<const DevNode*>(c->d_arena + c->node_off);
    sc.tris = c->d_tris;
    sc.objects = c->d_objects;
    sc.instances = reinterpret_cast<const DevInstance*>(c->d_arena + c->inst_off);
    sc.arena = c->d_arena;
    sc.node_off = (uint32_t)c->node_off;
    sc.tri_test_off = (uint32_t)c->tri_test_off;
    sc.inst_off = (uint32_t)c->inst_off;
    sc.node_unit = (uint32_t)(c->node_off / 64);
    sc.tri_test_unit = (uint32_t)(c->tri_test_off / 64);
    sc.n_instances = c->n_instances;
    sc.fast_div = (c->fast_div && c->fast_div_enabled) ? 1u : 0u;
    sc.instance_cull = c->instance_cull ? 1u : 0u;
    // the persistent kernel names items by 32-bit byte offsets into the arena, or by 32-bit record
    // numbers (64-byte units) when the arena is 4 GiB or more (RT_OPT_FORCE_BIG_OFFSETS: always)
    const bool big = c->arena_bytes > 0xFFFFFFFFull || c->force_big_offsets;
    const bool offsets_fit = c->arena_bytes / 64 <= 0xFFFFFFFEull;
    DevCam dc = make_cam(cam, c->frame_stride, c->seed_mode);
    DevRows rows;
    rows.strip_h = c->world > 1 ? c->strip_h : cam.canvas_height;
    rows.rank = c->rank;
    rows.world = c->world;
    rows.local_rows = local_rows(cam.canvas_height, rows.strip_h, c->rank, c->world);
    DevFrame fr;
    if (own_buffers) {
        fr.acc = c->d_acc;
        fr.present = c->present ? c->d_present : nullptr;
        fr.hit_triangle = c->probe_hits ? c->d_hit_tri : nullptr;
        fr.hit_instance = c->probe_hits ? c->d_hit_inst : nullptr;
        fr.hit_t_bits = c->probe_hits ? c->d_hit_t : nullptr;
        fr.box_tests = c->counters ? c->d_box : nullptr;
        fr.triangle_tests = c->counters ? c->d_tritests : nullptr;
    } else {
        fr = *caller_frame;
    }
    fr.stats = c->d_stats;

    // the debug heat-maps read the counters (trace.rs:97-125)
    const bool count = (own_buffers && c->counters) || cam.debug_information != RT_DEBUG_NONE;
    const uint32_t stack_needed = c->max_depth > 1 ? c->max_depth - 1 : 1;

    if (!own_buffers && (c->pipeline == 0 || !offsets_fit))
        return fail(c, RT_ERR_INVALID, "the no-allocation entry point runs the persistent kernel only");
    // depth 0: Ray::get_color's loop runs zero times and every sample is black (trace.rs:193-205); the
    // persistent kernels are built around "a path has a first segment", the plain one follows the loop
    const bool plain = c->pipeline == 0 || !offsets_fit || cam.depth == 0;
    RT_HIP(c, hipEventRecord(c->ev0, c->stream));
    if (plain) {
        RT_HIP(c, launch_path_kernel(sc, dc, rows, fr, n_frames, stack_needed, count, c->stream));
    } else {
        const uint32_t n_tiles = persistent_tile_count(dc, rows);
        QueueBufs qb;
        if (own_buffers) {
            int rc = ctx_queue_bufs(c, n_tiles, qb);
            if (rc != RT_OK) return rc;
        } else {
            qb = *caller_bufs;
            if (!c->tile_order_enabled) qb.cost = nullptr;
        }
        DevQueue q;
        std::memset(&q, 0, sizeof q);
        q.head = qb.head;
        q.w_ctl = c->vote_w[0];
        q.w_box = c->vote_w[1];
        q.w_leaf = c->vote_w[2];
        q.pool_slots = c->pool_slots;
        // the order this launch leaves behind is for the next launch, most likely one of the same length
        q.deal = c->tile_deal >= 0 ? (uint32_t)c->tile_deal : (n_frames <= 4u ? 1u : 0u);
        uint32_t* order_out = nullptr;
        int next = -1;
        if (qb.cost) {
            const uint32_t sig[5] = {dc.W, dc.H, rows.strip_h, rows.rank, rows.world};
            if (std::memcmp(sig, qb.order_sig, sizeof sig) != 0) {
                *qb.order_current = -1;  // another canvas or another share of it: start from the plain order
                std::memcpy(qb.order_sig, sig, sizeof sig);
            }
            q.cost = qb.cost;
            q.order = *qb.order_current >= 0 ? qb.order[*qb.order_current] : nullptr;
            next = *qb.order_current == 0 ? 1 : 0;
            order_out = qb.order[next];
            // the buffer written below only counts as an order once every launch has been enqueued
            *qb.order_current = -1;
        }
        // the pooled form packs a pixel's coordinates into one word and is the context's own (it needs memory)
        const bool pool = c->pipeline == 2 && own_buffers && dc.W < 65536u && dc.H < 65536u;
        const uint32_t n_blocks = (uint32_t)(c->n_cus * blocks_per_cu(c, stack_needed, count, big, pool));
        if (pool && pool_record_bytes(n_blocks) > c->pool_bytes) {
            RT_HIP(c, hipStreamSynchronize(c->stream));
            free_dev(c->d_pool);
            c->pool_bytes = 0;
            RT_HIP(c, hipMalloc(&c->d_pool, pool_record_bytes(n_blocks)));
            c->pool_bytes = pool_record_bytes(n_blocks);
        }
        uint32_t frames_left = n_frames;
        if (qb.cost && q.order == nullptr && n_frames >= kScoutMinFrames && !pool) {
            // No tile order has been learnt for this canvas yet, and a launch of many frames in the plain
            // order ends on the few lanes that happened to get the dearest pixels last (a fifth longer on
            // the benchmark's scene).  So the first frame goes out as a launch of its own, which costs half
            // a millisecond more than its share of a fused launch, and the others follow in the order it
            // has learnt: frame k + 1 of a pixel needs only the accumulator frame k stored, so two
            // launches give the pixels one gives.
            DevFrame first = fr;
            first.present = nullptr;
            DevQueue q1 = q;
            q1.deal = 0;  // the order is for the long launch that follows
            RT_HIP(c, launch_persistent_kernel(sc, dc, rows, first, q1, order_out, 1u, stack_needed, count, big, n_blocks,
                                               nullptr, c->stream));
            q.order = order_out;
            order_out = qb.order[next == 0 ? 1 : 0];
            next = next == 0 ? 1 : 0;
            dc.frame += dc.frame_stride;  // main.rs:147-148
            dc.frames_without_move += 1.0f;
            frames_left -= 1u;
        }
        RT_HIP(c, launch_persistent_kernel(sc, dc, rows, fr, q, order_out, frames_left, stack_needed, count, big, n_blocks,
                                           pool ? c->d_pool : nullptr, c->stream));
        if (qb.cost && n_tiles != 0) *qb.order_current = next;
    }
    RT_HIP(c, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    if (own_buffers) c->present_valid = fr.present != nullptr;
    return RT_OK;
}

// The pooled kernel's waves record it when they give up with live ray slots; the image is then incomplete.
int check_pool(rt_ctx* c, const DevStats& s) {
    if (s.pool_stalled != 0)
        return fail(c, RT_ERR_HIP, "pooled kernel: " + std::to_string(s.pool_stalled) +
                                       " wave(s) stopped with unfinished pixels (scheduler fault); the image is incomplete");
    return RT_OK;
}

}  // namespace

namespace rt {

int ctx_fail(rt_ctx* c, int code, const std::string& msg) { return fail(c, code, msg); }

int ctx_render(rt_ctx* c, const rt_cam_data& cam, uint32_t n_frames, bool present) {
    RT_HIP(c, hipSetDevice(c->device));
    const bool wanted = c->present;
    c->present = wanted && present;
    const int rc = render(c, cam, n_frames, nullptr, nullptr);
    c->present = wanted;
    return rc;
}

int ctx_present_whole_canvas(rt_ctx* c, const rt_cam_data& cam, uint32_t n_frames, const float* acc) {
    RT_HIP(c, hipSetDevice(c->device));
    int rc = ensure_frame(c, cam.canvas_width, cam.canvas_height);
    if (rc != RT_OK) return rc;
    const DevCam dc = make_cam(cam, c->frame_stride, c->seed_mode);
    DevRows rows;
    rows.strip_h = cam.canvas_height;
    rows.rank = 0;
    rows.world = 1;
    rows.local_rows = cam.canvas_height;
    DevFrame fr;
    std::memset(&fr, 0, sizeof fr);
    fr.acc = const_cast<float*>(acc ? acc : c->d_acc);  // read only by the pass
    fr.present = c->d_present;
    RT_HIP(c, launch_present(dc, rows, fr, n_frames, c->stream));
    c->present_valid = n_frames != 0;
    return RT_OK;
}

}  // namespace rt

extern "C" {

const char* rt_version(void) { return "ray_tracing_amd 0.1 (gfx950)"; }

int rt_device_count(int* n) {
    if (!n) return RT_ERR_INVALID;
    *n = 0;
    const hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_create(int device, rt_ctx** out) {
    if (!out) return fail(nullptr, RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, RT_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, RT_ERR_INVALID, "device index out of range");
    rt_ctx* c = new rt_ctx;
    c->device = device;
    auto bail = [&](const char* what, hipError_t err) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(err);
        rt_destroy(c);
        return fail(nullptr, RT_ERR_HIP, msg);
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc(&c->d_stats, sizeof(DevStats))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(c->d_stats, 0, sizeof(DevStats))) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMalloc(&c->d_queue, 8 * 32 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->n_cus = prop.multiProcessorCount;
    // the address of this GPU's copy of the no-allocation entry point's scratch array; the lookup
    // also loads the code object, so no later launch has to
    if ((e = noalloc_scratch_init(device)) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    free_frame(c);
    free_dev(c->d_stats);
    free_dev(c->d_queue);
    free_dev(c->d_pool);
    free_dev(c->d_tile_cost);
    free_dev(c->d_tile_order[0]);
    free_dev(c->d_tile_order[1]);
    if (c->ev_quantised) (void)hipEventDestroy(c->ev_quantised);
    if (c->ev_copied) (void)hipEventDestroy(c->ev_copied);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int rt_set_stream(rt_ctx* c, void* hip_stream) {
    if (!c) return RT_ERR_INVALID;
    // launches on the old stream still use the queue, cost and order buffers the next launch reuses;
    // a caller that has already destroyed the old stream has waited for them itself
    RT_HIP(c, hipSetDevice(c->device));
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipErrorInvalidHandle || e == hipErrorContextIsDestroyed || e == hipErrorInvalidResourceHandle)
        (void)hipGetLastError();
    else if (e != hipSuccess)
        return fail(c, RT_ERR_HIP, std::string("hipStreamSynchronize(old stream): ") + hipGetErrorString(e));
    c->stream = static_cast<hipStream_t>(hip_stream);
    return RT_OK;
}

int rt_set_option(rt_ctx* c, int option, int64_t value) {
    if (!c) return RT_ERR_INVALID;
    switch (option) {
        case RT_OPT_PIPELINE:
            if (value < 0 || value > 2) return fail(c, RT_ERR_INVALID, "unknown pipeline");
            c->pipeline = (int)value;
            return RT_OK;
        case RT_OPT_PROBE_HITS: c->probe_hits = value != 0; return RT_OK;
        case RT_OPT_COUNTERS: c->counters = value != 0; return RT_OK;
        case RT_OPT_PRESENT: c->present = value != 0; return RT_OK;  // a render without it leaves binding 8 stale
        case RT_OPT_FRAME_STRIDE:
            if (value < 1 || value > 0x7FFFFFFF) return fail(c, RT_ERR_INVALID, "frame stride must be >= 1");
            c->frame_stride = (uint32_t)value;
            return RT_OK;
        case RT_OPT_SEED_MODE:
            if (value < 0 || value > 1) return fail(c, RT_ERR_INVALID, "unknown seed mode");
            c->seed_mode = (uint32_t)value;
            return RT_OK;
        case RT_OPT_TILE_ORDER:
            c->tile_order_enabled = value != 0;
            c->order_current = -1;
            return RT_OK;
        case RT_OPT_VOTE_WEIGHTS: {
            const int a = (int)(value & 0xFF), b = (int)((value >> 8) & 0xFF), d = (int)((value >> 16) & 0xFF);
            if (value < 0 || value > 0xFFFFFF || a == 0 || b == 0 || d == 0)
                return fail(c, RT_ERR_INVALID, "vote weights: three bytes, each 1..255");
            c->vote_w[0] = a;
            c->vote_w[1] = b;
            c->vote_w[2] = d;
            return RT_OK;
        }
        case RT_OPT_BLOCKS_PER_CU:
            if (value < 0 || value > 8) return fail(c, RT_ERR_INVALID, "blocks per CU: 0 (as many as fit) .. 8");
            c->blocks_per_cu = (int)value;
            return RT_OK;
        case RT_OPT_FORCE_BIG_OFFSETS: c->force_big_offsets = value != 0; return RT_OK;
        case RT_OPT_FAST_DIV: c->fast_div_enabled = value != 0; return RT_OK;
        case RT_OPT_INSTANCE_CULL: c->instance_cull = value != 0; return RT_OK;
        case RT_OPT_POOL_SLOTS:
            if (value != 0 && (value < 64 || value > 192)) return fail(c, RT_ERR_INVALID, "pool slots: 0 (all) or 64..192");
            c->pool_slots = (uint32_t)value;
            return RT_OK;
        case RT_OPT_TILE_DEAL:
            if (value < -1 || value > 1) return fail(c, RT_ERR_INVALID, "tile deal: -1 (by launch length), 0 or 1");
            c->tile_deal = (int)value;
            return RT_OK;
        default: return fail(c, RT_ERR_INVALID, "unknown option");
    }
}

int rt_upload_scene(rt_ctx* c, const rt_scene_info* info, const rt_vertex* vertices, size_t n_vertices,
                    const rt_tri* triangles, size_t n_triangles, const rt_object* objects, size_t n_objects,
                    const rt_instance* instances, size_t n_instances, const rt_bvh* bvh, size_t n_bvh) {
    if (!c || !info) return RT_ERR_INVALID;
    GpuScene g;
    std::string err;
    if (!build_gpu_scene(*info, vertices, n_vertices, triangles, n_triangles, objects, n_objects, instances,
                         n_instances, bvh, n_bvh, g, err))
        return fail(c, RT_ERR_INVALID, err);
    if (g.max_depth > kMaxBvhDepth) return fail(c, RT_ERR_LIMIT, "BVH deeper than this build supports");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    // a scene whose only nodes are leaves still needs a valid pointer
    const size_t n_nodes = g.nodes.empty() ? 1 : g.nodes.size();
    const size_t n_tris = g.tris.empty() ? 1 : g.tris.size();
    c->node_off = 0;
    c->tri_test_off = c->node_off + n_nodes * sizeof(DevNode);
    c->inst_off = c->tri_test_off + n_tris * sizeof(DevTriTest);
    c->arena_bytes = c->inst_off + g.instances.size() * sizeof(DevInstance);
    RT_HIP(c, hipMalloc(&c->d_arena, c->arena_bytes));
    RT_HIP(c, hipMalloc(&c->d_tris, n_tris * sizeof(DevTri)));
    RT_HIP(c, hipMalloc(&c->d_objects, g.objects.size() * sizeof(DevObject)));
    if (!g.nodes.empty())
        RT_HIP(c, hipMemcpy(c->d_arena + c->node_off, g.nodes.data(), g.nodes.size() * sizeof(DevNode),
                            hipMemcpyHostToDevice));
    if (!g.tris.empty())
        RT_HIP(c, hipMemcpy(c->d_tris, g.tris.data(), g.tris.size() * sizeof(DevTri), hipMemcpyHostToDevice));
    if (!g.tri_tests.empty())
        RT_HIP(c, hipMemcpy(c->d_arena + c->tri_test_off, g.tri_tests.data(), g.tri_tests.size() * sizeof(DevTriTest),
                            hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->d_objects, g.objects.data(), g.objects.size() * sizeof(DevObject), hipMemcpyHostToDevice));
    RT_HIP(c, hipMalloc(&c->d_tri_owner, n_tris * 4));
    RT_HIP(c, hipMalloc(&c->d_object_instance, g.objects.size() * 4));
    if (!g.tri_owner.empty())
        RT_HIP(c, hipMemcpy(c->d_tri_owner, g.tri_owner.data(), g.tri_owner.size() * 4, hipMemcpyHostToDevice));
    c->n_tris = g.tris.size();
    c->h_instances = g.instances;
    c->h_objects = g.objects;
    c->h_object_shares = g.object_shares;
    {
        const int rc = upload_instances(c);
        if (rc != RT_OK) return rc;
    }
    c->n_instances = (uint32_t)g.instances.size();
    c->max_depth = g.max_depth;
    c->fast_div = g.fast_div;
    c->have_scene = true;
    // ask for the occupancy of the kernel variants this scene can launch now, so that a later
    // render call has nothing left to look up
    const uint32_t stack_needed = c->max_depth > 1 ? c->max_depth - 1 : 1;
    for (int big = 0; big < 2; ++big) (void)blocks_per_cu(c, stack_needed, false, big != 0, false);
    return RT_OK;
}

int rt_update_instances(rt_ctx* c, const rt_instance* instances, size_t n) {
    if (!c) return RT_ERR_INVALID;
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if (!instances || n != c->n_instances) return fail(c, RT_ERR_INVALID, "the number of instances is fixed by rt_upload_scene");
    for (size_t i = 0; i < n; ++i)
        if (instances[i].object_id >= c->h_objects.size())
            return fail(c, RT_ERR_INVALID, "instance " + std::to_string(i) + " references an object out of range");
    for (size_t i = 0; i < n; ++i) set_instance_transform(c->h_instances[i], instances[i], c->h_objects);
    choose_cull_candidates(c->h_instances);
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));  // frames in flight still read the old records
    return upload_instances(c);
}

void rt_default_material(uint32_t instance_id, rt_material* out) {
    if (out) default_material(instance_id, *out);
}

int rt_set_materials(rt_ctx* c, const rt_material* materials, size_t n) {
    if (!c) return RT_ERR_INVALID;
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if (materials && n != c->n_instances) return fail(c, RT_ERR_INVALID, "one material per instance is required");
    for (uint32_t i = 0; i < c->n_instances; ++i) {
        rt_material m;
        if (materials) m = materials[i];
        else default_material(i, m);
        if (m.kind >= RT_MAT_KIND_COUNT) return fail(c, RT_ERR_INVALID, "unknown material kind");
        apply_material(c->h_instances[i], m);
    }
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(c->d_arena + c->inst_off, c->h_instances.data(), c->h_instances.size() * sizeof(DevInstance),
                        hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_set_rows(rt_ctx* c, uint32_t strip_height, uint32_t rank, uint32_t world) {
    if (!c) return RT_ERR_INVALID;
    if (world <= 1) {
        c->strip_h = 0;
        c->rank = 0;
        c->world = 1;
        return RT_OK;
    }
    if (strip_height == 0 || strip_height % 8 != 0) return fail(c, RT_ERR_INVALID, "strip height must be a multiple of 8");
    if (rank >= world) return fail(c, RT_ERR_INVALID, "rank >= world");
    c->strip_h = strip_height;
    c->rank = rank;
    c->world = world;
    return RT_OK;
}

int rt_render_frame(rt_ctx* c, const rt_cam_data* cam) {
    if (!c || !cam) return RT_ERR_INVALID;
    RT_HIP(c, hipSetDevice(c->device));
    return render(c, *cam, 1, nullptr, nullptr);
}

int rt_render_frames(rt_ctx* c, rt_cam_data* cam, uint32_t n) {
    if (!c || !cam) return RT_ERR_INVALID;
    RT_HIP(c, hipSetDevice(c->device));
    int rc = render(c, *cam, n, nullptr, nullptr);
    if (rc != RT_OK) return rc;
    for (uint32_t i = 0; i < n; ++i) {  // main.rs:147-148
        cam->frame += c->frame_stride;
        cam->frames_without_move += 1.0f;
    }
    return RT_OK;
}

int rt_render_frames_noalloc(rt_ctx* c, rt_cam_data* cam, uint32_t n, float* acc_device, float* present_device) {
    if (!c || !cam) return RT_ERR_INVALID;
    if (!acc_device) return fail(c, RT_ERR_INVALID, "acc_device is NULL");
    if (cam->canvas_width == 0 || cam->canvas_height == 0 ||
        (uint64_t)cam->canvas_width * cam->canvas_height > 0x7FFFFFFFull)
        return fail(c, RT_ERR_INVALID, "bad canvas size");
    if (c->probe_hits || c->counters)
        return fail(c, RT_ERR_INVALID, "RT_OPT_PROBE_HITS / RT_OPT_COUNTERS need buffers this entry point does not allocate");
    const uint64_t tiles = (uint64_t)((cam->canvas_width + 7u) / 8u) * ((cam->canvas_height + 7u) / 8u);
    if (tiles > kNoAllocMaxTiles)
        return fail(c, RT_ERR_LIMIT, "canvas of " + std::to_string(tiles) + " 8x8 tiles exceeds the " +
                                         std::to_string(kNoAllocMaxTiles) + " the static scratch holds");
    RT_HIP(c, hipSetDevice(c->device));
    NoAllocView v;
    if (!noalloc_scratch_view(c->device, &v)) return fail(c, RT_ERR_HIP, "the scratch array of this device was never looked up");
    QueueBufs qb;
    qb.head = v.head;
    qb.cost = v.cost;
    qb.order[0] = v.order[0];
    qb.order[1] = v.order[1];
    qb.order_current = v.order_current;
    qb.order_sig = v.order_sig;
    DevFrame fr;
    std::memset(&fr, 0, sizeof fr);
    fr.acc = acc_device;
    fr.present = present_device;
    int rc = render(c, *cam, n, &fr, &qb);
    if (rc != RT_OK) return rc;
    for (uint32_t i = 0; i < n; ++i) {  // main.rs:147-148
        cam->frame += c->frame_stride;
        cam->frames_without_move += 1.0f;
    }
    return RT_OK;
}

int rt_synchronize(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

static int read_back(rt_ctx* c, void* dst, const void* src, size_t bytes, size_t n_pixels, const char* what) {
    if (!c || !dst) return RT_ERR_INVALID;
    if (!src) return fail(c, RT_ERR_INVALID, std::string(what) + " was never produced (option off or nothing rendered)");
    if (n_pixels != (size_t)c->W * c->H) return fail(c, RT_ERR_INVALID, "n_pixels does not match the canvas");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_acc(rt_ctx* c, float* rgba, size_t n) {
    int rc = read_back(c, rgba, c ? c->d_acc : nullptr, n * 16, n, "acc");
    if (rc != RT_OK || c->pipeline != 2) return rc;
    DevStats s;
    RT_HIP(c, hipMemcpy(&s, c->d_stats, sizeof s, hipMemcpyDeviceToHost));
    return check_pool(c, s);
}

int rt_read_present(rt_ctx* c, float* rgba, size_t n) {
    return read_back(c, rgba, (c && c->present_valid) ? c->d_present : nullptr, n * 16, n, "present");
}

// Binding 8 quantised into d_present8, enqueued on the render stream (after any copy that still
// reads the buffer).
static int quantise_present(rt_ctx* c) {
    if (!c->d_present || !c->present_valid)
        return fail(c, RT_ERR_INVALID, "present was never produced (option off or nothing rendered)");
    RT_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->W * c->H;
    if (!c->d_present8) RT_HIP(c, hipMalloc(&c->d_present8, n * 4));
    if (c->copy_pending) RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copied, 0));
    RT_HIP(c, launch_present_bytes(c->d_present, (uint32_t)n, c->d_present8, c->stream));
    return RT_OK;
}

int rt_read_present_rgba8(rt_ctx* c, uint8_t* rgba8, size_t n) {
    if (!c || !rgba8) return RT_ERR_INVALID;
    if (n != (size_t)c->W * c->H) return fail(c, RT_ERR_INVALID, "n_pixels does not match the canvas");
    const int rc = quantise_present(c);
    if (rc != RT_OK) return rc;
    RT_HIP(c, hipMemcpyAsync(rgba8, c->d_present8, n * 4, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_present_begin(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    int rc = quantise_present(c);
    if (rc != RT_OK) return rc;
    const size_t n = (size_t)c->W * c->H;
    if (!c->copy_stream) RT_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->ev_quantised) RT_HIP(c, hipEventCreateWithFlags(&c->ev_quantised, hipEventDisableTiming));
    if (!c->ev_copied) RT_HIP(c, hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming));
    if (!c->h_present8) RT_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_present8), n * 4, hipHostMallocDefault));
    RT_HIP(c, hipEventRecord(c->ev_quantised, c->stream));
    RT_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_quantised, 0));
    RT_HIP(c, hipMemcpyAsync(c->h_present8, c->d_present8, n * 4, hipMemcpyDeviceToHost, c->copy_stream));
    RT_HIP(c, hipEventRecord(c->ev_copied, c->copy_stream));
    c->copy_pending = true;
    return RT_OK;
}

int rt_present_end(rt_ctx* c, const uint8_t** rgba8, size_t* n_pixels) {
    if (!c || !rgba8) return RT_ERR_INVALID;
    if (!c->copy_pending) return fail(c, RT_ERR_INVALID, "rt_present_begin has not been called for this canvas");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipEventSynchronize(c->ev_copied));
    *rgba8 = c->h_present8;
    if (n_pixels) *n_pixels = (size_t)c->W * c->H;
    return RT_OK;
}

int rt_write_acc(rt_ctx* c, const float* rgba, size_t n, uint32_t width, uint32_t height) {
    if (!c || !rgba) return RT_ERR_INVALID;
    if (n != (size_t)width * height) return fail(c, RT_ERR_INVALID, "n_pixels does not match width*height");
    int rc = ensure_frame(c, width, height);
    if (rc != RT_OK) return rc;
    RT_HIP(c, hipMemcpyAsync(c->d_acc, rgba, n * 16, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_clear_acc(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    if (!c->d_acc) return RT_OK;  // nothing rendered yet: the buffer will start zeroed
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemsetAsync(c->d_acc, 0, (size_t)c->W * c->H * 16, c->stream));
    return RT_OK;
}

int rt_read_hits(rt_ctx* c, uint32_t* tri, uint32_t* inst, uint32_t* t_bits, size_t n) {
    int rc = read_back(c, tri, c ? c->d_hit_tri : nullptr, n * 4, n, "hit buffer");
    if (rc != RT_OK) return rc;
    rc = read_back(c, inst, c->d_hit_inst, n * 4, n, "hit buffer");
    if (rc != RT_OK) return rc;
    return read_back(c, t_bits, c->d_hit_t, n * 4, n, "hit buffer");
}

int rt_read_counters(rt_ctx* c, uint32_t* box_tests, uint32_t* triangle_tests, size_t n) {
    int rc = read_back(c, box_tests, c ? c->d_box : nullptr, n * 4, n, "counter buffer");
    if (rc != RT_OK) return rc;
    return read_back(c, triangle_tests, c->d_tritests, n * 4, n, "counter buffer");
}

int rt_get_stats(rt_ctx* c, rt_stats* out) {
    if (!c || !out) return RT_ERR_INVALID;
    DevStats s;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemcpyAsync(&s, c->d_stats, sizeof s, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    out->rays = s.rays;
    out->box_tests = s.box_tests;
    out->internal_visits = s.internal_visits;
    out->triangle_tests = s.triangle_tests;
    out->paths = s.paths;
    return check_pool(c, s);
}

int rt_get_sched_stats(rt_ctx* c, rt_sched_stats* out) {
    if (!c || !out) return RT_ERR_INVALID;
    DevStats s;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemcpyAsync(&s, c->d_stats, sizeof s, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) {
        out->iterations[k] = s.iters[k];
        out->lanes[k] = s.lanes[k];
        out->cycles[k] = s.cycles[k];
        out->wait_cycles[k] = s.wait_cycles[k];
    }
    for (int k = 0; k < 8; ++k) out->site_wait_cycles[k] = s.site_wait[k];
    out->shade_iterations = s.shade_iters;
    out->shade_slots = s.shade_lanes;
    out->shade_cycles = s.shade_cycles;
    return RT_OK;
}

int rt_get_kernel_clock(rt_ctx* c, uint64_t* shader_cycles, uint64_t* ticks_100mhz) {
    if (!c || !shader_cycles || !ticks_100mhz) return RT_ERR_INVALID;
    DevStats s;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemcpyAsync(&s, c->d_stats, sizeof s, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    *shader_cycles = s.clk_cycles;
    *ticks_100mhz = s.clk_ticks;
    return RT_OK;
}

int rt_reset_stats(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemsetAsync(c->d_stats, 0, sizeof(DevStats), c->stream));
    return RT_OK;
}

int rt_acc_device_ptr(rt_ctx* c, uint32_t width, uint32_t height, void** out) {
    if (!c || !out) return RT_ERR_INVALID;
    int rc = ensure_frame(c, width, height);
    if (rc != RT_OK) return rc;
    *out = c->d_acc;
    return RT_OK;
}

int rt_last_render_ms(rt_ctx* c, float* ms) {
    if (!c || !ms) return RT_ERR_INVALID;
    if (!c->timed) return fail(c, RT_ERR_INVALID, "nothing has been rendered");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipEventSynchronize(c->ev1));
    RT_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return RT_OK;
}

int rt_upload_scene_from_builder(rt_ctx* c, rt_scene_builder* b);  // defined in rt_host_api.cpp

}  // extern "C"
