// Several GPUs of one node driven as one renderer through the C ABI (include/rt_hip.h, rt_group_*).
//
// The reference drives one App on one GPU from one thread (program/src/main.rs:129-149,
// program/src/modules/vulkan.rs:124-236).  Every pixel sample depends only on the scene and on its own
// accumulator element (shader/src/lib.rs:57-69), so the work shards without any exchange while it
// is being done: the scene is replicated, one context per GPU, one host thread per context, and the
// only data that moves between GPUs is the finished image -- strips of rows copied device to device
// into the first GPU's accumulator (tile mode), or whole accumulators copied to the first GPU and
// added there in rank order (sample mode).
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_ctx.hpp"

using namespace rt;

namespace {

constexpr uint32_t kStripRows = 8;  // tile mode: interleaved strips of this many rows (a multiple of the 8x8 pixel tiles)

// One host thread per context: takes one job at a time, runs it, keeps its return code.
class Worker {
   public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<int()> job) {
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = std::move(job);
            busy_ = true;
        }
        cv_.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return !busy_; });
        return rc_;
    }

   private:
    void loop() {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || (busy_ && job_); });
                if (stop_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            const int rc = job();
            {
                std::lock_guard<std::mutex> l(m_);
                rc_ = rc;
                busy_ = false;
            }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool busy_ = false, stop_ = false;
    int rc_ = 0;
    std::thread th_;  // last: started when everything above exists
};

thread_local std::string g_group_create_err;

}  // namespace

struct rt_group {
    std::vector<rt_ctx*> ctx;
    std::vector<std::unique_ptr<Worker>> worker;
    std::vector<hipStream_t> stream;    // one per context, created here
    std::vector<hipEvent_t> ev_ready;   // context g's share has arrived on the first GPU
    hipEvent_t ev_assembled = nullptr;  // the first GPU is through with the assembled image of the last call
    bool assembled_once = false;
    std::string err;

    // sample mode: the accumulators of contexts 1.. staged on the first GPU, and their sum
    std::vector<float*> d_staged;
    float* d_total = nullptr;
    size_t total_pixels = 0;
    uint64_t samples_done = 0;             // samples in the sum since the last reset
    std::vector<uint64_t> rank_samples;   // of which context g rendered this many
    int last_mode = RT_GROUP_TILES;
    uint32_t W = 0, H = 0;
};

namespace {

int gfail(rt_group* g, int code, const std::string& msg) {
    if (g) g->err = msg;
    else g_group_create_err = msg;
    return code;
}

#define RT_GHIP(g, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return gfail(g, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Run `job(i)` on every context's thread at once; the first failure (lowest context) is the group's.
int on_all(rt_group* g, const std::function<int(size_t)>& job) {
    for (size_t i = 0; i < g->ctx.size(); ++i) g->worker[i]->post([&job, i] { return job(i); });
    int first = RT_OK;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const int rc = g->worker[i]->wait();
        if (rc != RT_OK && first == RT_OK) {
            first = rc;
            g->err = "context " + std::to_string(i) + " (GPU " + std::to_string(g->ctx[i]->device) + "): " + rt_last_error(g->ctx[i]);
        }
    }
    return first;
}

// `bytes` from `src` on context `from`'s GPU to `dst` on the first context's GPU, enqueued on `from`'s stream.
hipError_t copy_to_first(rt_group* g, size_t from, void* dst, const void* src, size_t bytes) {
    const int d0 = g->ctx[0]->device, d1 = g->ctx[from]->device;
    if (d0 == d1) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g->stream[from]);
    return hipMemcpyPeerAsync(dst, d0, src, d1, bytes, g->stream[from]);
}

// Context `from`'s strips of the accumulator into the first context's accumulator.
int gather_strips(rt_group* g, size_t from, uint32_t W, uint32_t H) {
    rt_ctx* c = g->ctx[from];
    const size_t world = g->ctx.size(), row_bytes = (size_t)W * 16, strip_bytes = row_bytes * kStripRows;
    char* dst = reinterpret_cast<char*>(g->ctx[0]->d_acc);
    const char* src = reinterpret_cast<const char*>(c->d_acc);
    const uint32_t n_strips = (H + kStripRows - 1) / kStripRows;
    uint32_t full = 0;  // this context's strips of full height
    for (uint32_t s = (uint32_t)from; s < n_strips; s += (uint32_t)world)
        if ((s + 1) * kStripRows <= H) ++full;
    if (g->ctx[0]->device == c->device) {
        // one strided copy: `full` pieces of strip_bytes, world * strip_bytes apart
        if (full)
            if (hipMemcpy2DAsync(dst + from * strip_bytes, world * strip_bytes, src + from * strip_bytes, world * strip_bytes,
                                 strip_bytes, full, hipMemcpyDeviceToDevice, g->stream[from]) != hipSuccess)
                return ctx_fail(c, RT_ERR_HIP, "hipMemcpy2DAsync of the strips failed");
    } else {
        for (uint32_t k = 0; k < full; ++k) {
            const size_t off = ((size_t)from + (size_t)k * world) * strip_bytes;
            if (copy_to_first(g, from, dst + off, src + off, strip_bytes) != hipSuccess)
                return ctx_fail(c, RT_ERR_HIP, "hipMemcpyPeerAsync of a strip failed");
        }
    }
    // the canvas's last strip may be shorter
    const uint32_t last = n_strips - 1;
    if (H % kStripRows != 0 && last % world == from) {
        const size_t off = (size_t)last * strip_bytes;
        if (copy_to_first(g, from, dst + off, src + off, (size_t)(H - last * kStripRows) * row_bytes) != hipSuccess)
            return ctx_fail(c, RT_ERR_HIP, "copy of the last strip failed");
    }
    return RT_OK;
}

void free_sample_buffers(rt_group* g) {
    if (g->ctx.empty()) return;
    (void)hipSetDevice(g->ctx[0]->device);
    for (float*& p : g->d_staged) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    if (g->d_total) (void)hipFree(g->d_total);
    g->d_total = nullptr;
    g->total_pixels = 0;
}

}  // namespace

extern "C" {

int rt_group_create(const int* devices, size_t n, rt_group** out) {
    if (!out) return gfail(nullptr, RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n == 0) return gfail(nullptr, RT_ERR_INVALID, "no devices");
    if (n > kMaxGroup) return gfail(nullptr, RT_ERR_LIMIT, "a group holds at most " + std::to_string(kMaxGroup) + " contexts");
    std::unique_ptr<rt_group> g(new rt_group);
    auto bail = [&](int code, const std::string& msg) {
        rt_group_destroy(g.release());
        return gfail(nullptr, code, msg);
    };
    for (size_t i = 0; i < n; ++i) {
        rt_ctx* c = nullptr;
        const int rc = rt_create(devices[i], &c);
        if (rc != RT_OK) return bail(rc, "GPU " + std::to_string(devices[i]) + ": " + rt_last_error(nullptr));
        g->ctx.push_back(c);
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        // rt_create left devices[i] current
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return bail(RT_ERR_HIP, "hipStreamCreate failed");
        g->stream.push_back(s);
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(RT_ERR_HIP, "hipEventCreate failed");
        g->ev_ready.push_back(e);
        if (rt_set_stream(c, s) != RT_OK) return bail(RT_ERR_HIP, rt_last_error(c));
        g->worker.emplace_back(new Worker);
        // a GPU that can reach the first one directly does so (xGMI); the copies work either way
        if (i > 0 && devices[i] != devices[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(devices[0], 0);
                if (pe != hipSuccess) (void)hipGetLastError();  // already enabled, or not allowed: staged copies still work
            }
        }
    }
    g->d_staged.assign(n, nullptr);
    g->rank_samples.assign(n, 0);
    (void)hipSetDevice(devices[0]);
    if (hipEventCreateWithFlags(&g->ev_assembled, hipEventDisableTiming) != hipSuccess) return bail(RT_ERR_HIP, "hipEventCreate failed");
    *out = g.release();
    return RT_OK;
}

void rt_group_destroy(rt_group* g) {
    if (!g) return;
    g->worker.clear();  // joins the threads
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        if (i < g->stream.size() && g->stream[i]) (void)hipStreamSynchronize(g->stream[i]);
    }
    free_sample_buffers(g);
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        rt_ctx* c = g->ctx[i];
        const int dev = c->device;
        rt_destroy(c);  // waits for the stream it was given
        (void)hipSetDevice(dev);
        if (i < g->ev_ready.size() && g->ev_ready[i]) (void)hipEventDestroy(g->ev_ready[i]);
        if (i < g->stream.size() && g->stream[i]) (void)hipStreamDestroy(g->stream[i]);
    }
    if (g->ev_assembled) (void)hipEventDestroy(g->ev_assembled);
    delete g;
}

const char* rt_group_last_error(const rt_group* g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }
size_t rt_group_size(const rt_group* g) { return g ? g->ctx.size() : 0; }
rt_ctx* rt_group_context(rt_group* g, size_t i) { return (g && i < g->ctx.size()) ? g->ctx[i] : nullptr; }

int rt_group_upload_scene(rt_group* g, const rt_scene_info* info, const rt_vertex* vertices, size_t n_vertices,
                          const rt_tri* triangles, size_t n_triangles, const rt_object* objects, size_t n_objects,
                          const rt_instance* instances, size_t n_instances, const rt_bvh* bvh, size_t n_bvh) {
    if (!g) return RT_ERR_INVALID;
    return on_all(g, [&](size_t i) {
        return rt_upload_scene(g->ctx[i], info, vertices, n_vertices, triangles, n_triangles, objects, n_objects, instances,
                               n_instances, bvh, n_bvh);
    });
}

int rt_group_upload_scene_from_builder(rt_group* g, rt_scene_builder* b) {
    if (!g || !b) return RT_ERR_INVALID;
    rt_scene_info info;
    const rt_vertex* v;
    const rt_tri* t;
    const rt_object* o;
    const rt_instance* in;
    const rt_bvh* bv;
    size_t nv, nt, no, ni, nb;
    const int rc = rt_scene_builder_build(b, &info, &v, &nv, &t, &nt, &o, &no, &in, &ni, &bv, &nb);  // once, not once per thread
    if (rc != RT_OK) return gfail(g, rc, "rt_scene_builder_build failed");
    return rt_group_upload_scene(g, &info, v, nv, t, nt, o, no, in, ni, bv, nb);
}

int rt_group_set_materials(rt_group* g, const rt_material* materials, size_t n) {
    if (!g) return RT_ERR_INVALID;
    return on_all(g, [&](size_t i) { return rt_set_materials(g->ctx[i], materials, n); });
}

int rt_group_update_instances(rt_group* g, const rt_instance* instances, size_t n) {
    if (!g) return RT_ERR_INVALID;
    return on_all(g, [&](size_t i) { return rt_update_instances(g->ctx[i], instances, n); });
}

int rt_group_set_option(rt_group* g, int option, int64_t value) {
    if (!g) return RT_ERR_INVALID;
    if (option == RT_OPT_FRAME_STRIDE || option == RT_OPT_SEED_MODE)
        return gfail(g, RT_ERR_INVALID, "the group sets the frame stride and the seed mode itself, by the mode of each render call");
    if (option == RT_OPT_PROBE_HITS || option == RT_OPT_COUNTERS)
        return gfail(g, RT_ERR_INVALID, "the parity probes belong to one context: set them on rt_group_context(g, i)");
    for (rt_ctx* c : g->ctx) {
        const int rc = rt_set_option(c, option, value);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(c));
    }
    return RT_OK;
}

int rt_group_render_frames(rt_group* g, rt_cam_data* cam, uint32_t n, int mode) {
    if (!g || !cam) return RT_ERR_INVALID;
    if (mode != RT_GROUP_TILES && mode != RT_GROUP_SAMPLES) return gfail(g, RT_ERR_INVALID, "unknown group mode");
    const size_t world = g->ctx.size();
    const uint32_t W = cam->canvas_width, H = cam->canvas_height;
    rt_ctx* c0 = g->ctx[0];
    // every context's canvas exists before anything is copied into one
    for (rt_ctx* c : g->ctx) {
        void* p = nullptr;
        const int rc = rt_acc_device_ptr(c, W, H, &p);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(c));
    }
    const size_t n_pixels = (size_t)W * H;
    if (W != g->W || H != g->H) {
        g->W = W;
        g->H = H;
        g->samples_done = 0;
        std::fill(g->rank_samples.begin(), g->rank_samples.end(), 0);
    }
    const rt_cam_data cam0 = *cam;
    const bool want_present = c0->present;

    if (mode == RT_GROUP_SAMPLES) {
        RT_GHIP(g, hipSetDevice(c0->device));
        if (g->total_pixels != n_pixels) {
            RT_GHIP(g, hipStreamSynchronize(g->stream[0]));
            free_sample_buffers(g);
            RT_GHIP(g, hipMalloc(&g->d_total, n_pixels * 16));
            for (size_t i = 1; i < world; ++i) RT_GHIP(g, hipMalloc(&g->d_staged[i], n_pixels * 16));
            g->total_pixels = n_pixels;
        }
        if (cam0.frames_without_move < 0.5f) {  // the accumulate / reset rule of shader/src/lib.rs:85-87, for the whole group
            g->samples_done = 0;
            std::fill(g->rank_samples.begin(), g->rank_samples.end(), 0);
        }
    }

    // which samples of this call context i renders (sample mode): those whose running index is congruent to i
    std::vector<uint32_t> first_k(world, 0), count(world, 0);
    if (mode == RT_GROUP_SAMPLES) {
        for (size_t i = 0; i < world; ++i) {
            const uint64_t k0 = (i + world - g->samples_done % world) % world;
            first_k[i] = (uint32_t)k0;
            count[i] = k0 < n ? (uint32_t)((n - k0 + world - 1) / world) : 0u;
        }
    }

    const int rc = on_all(g, [&](size_t i) -> int {
        rt_ctx* c = g->ctx[i];
        if (hipSetDevice(c->device) != hipSuccess) return ctx_fail(c, RT_ERR_HIP, "hipSetDevice failed");
        // the first GPU may still be reading what this call is about to overwrite there
        if (i > 0 && g->assembled_once && hipStreamWaitEvent(g->stream[i], g->ev_assembled, 0) != hipSuccess)
            return ctx_fail(c, RT_ERR_HIP, "hipStreamWaitEvent failed");
        int r;
        if (mode == RT_GROUP_TILES) {
            c->frame_stride = 1;
            c->seed_mode = 0;
            if ((r = rt_set_rows(c, kStripRows, (uint32_t)i, (uint32_t)world)) != RT_OK) return r;
            if ((r = ctx_render(c, cam0, n, false)) != RT_OK) return r;
            if (i > 0 && n > 0 && (r = gather_strips(g, i, W, H)) != RT_OK) return r;
        } else {
            c->frame_stride = (uint32_t)world;
            c->seed_mode = 1;  // the reference's seed chains a pixel's samples through its accumulator; independent samples cannot
            if ((r = rt_set_rows(c, 0, 0, 1)) != RT_OK) return r;
            rt_cam_data mine = cam0;
            mine.frame = cam0.frame + first_k[i];
            mine.frames_without_move = (float)g->rank_samples[i];
            if (g->rank_samples[i] == 0 && count[i] == 0) {
                if ((r = rt_clear_acc(c)) != RT_OK) return r;  // a share without samples adds zeros
            } else if ((r = ctx_render(c, mine, count[i], false)) != RT_OK) {
                return r;
            }
            if (i > 0 && copy_to_first(g, i, g->d_staged[i], c->d_acc, n_pixels * 16) != hipSuccess)
                return ctx_fail(c, RT_ERR_HIP, "copy of the accumulator to the first GPU failed");
        }
        if (hipEventRecord(g->ev_ready[i], g->stream[i]) != hipSuccess) return ctx_fail(c, RT_ERR_HIP, "hipEventRecord failed");
        return RT_OK;
    });
    if (rc != RT_OK) return rc;

    // the first GPU assembles: waits for every share, adds them (sample mode), derives binding 8
    RT_GHIP(g, hipSetDevice(c0->device));
    for (size_t i = 1; i < world; ++i) RT_GHIP(g, hipStreamWaitEvent(g->stream[0], g->ev_ready[i], 0));
    rt_cam_data shown = cam0;
    uint32_t shown_frames = n;
    if (mode == RT_GROUP_SAMPLES) {
        AccumulatorList list;
        std::memset(&list, 0, sizeof list);
        list.n = (uint32_t)world;
        list.src[0] = c0->d_acc;
        for (size_t i = 1; i < world; ++i) list.src[i] = g->d_staged[i];
        RT_GHIP(g, launch_sum_accumulators(g->d_total, list, (uint32_t)n_pixels, g->stream[0]));
        for (size_t i = 0; i < world; ++i) g->rank_samples[i] += count[i];
        shown.frames_without_move = (float)g->samples_done;  // the sum holds samples_done + n samples
        g->samples_done += n;
        if (g->samples_done == 0) shown_frames = 0;
    }
    if (want_present && shown_frames > 0) {
        const int prc = ctx_present_whole_canvas(c0, shown, shown_frames, mode == RT_GROUP_SAMPLES ? g->d_total : nullptr);
        if (prc != RT_OK) return gfail(g, prc, rt_last_error(c0));
    }
    RT_GHIP(g, hipEventRecord(g->ev_assembled, g->stream[0]));
    g->assembled_once = true;
    g->last_mode = mode;
    for (uint32_t k = 0; k < n; ++k) {  // main.rs:147-148
        cam->frame += 1;
        cam->frames_without_move += 1.0f;
    }
    return RT_OK;
}

int rt_group_synchronize(rt_group* g) {
    if (!g) return RT_ERR_INVALID;
    for (size_t i = g->ctx.size(); i-- > 0;) {  // the first context last: its stream waits for the others
        RT_GHIP(g, hipSetDevice(g->ctx[i]->device));
        RT_GHIP(g, hipStreamSynchronize(g->stream[i]));
    }
    return RT_OK;
}

int rt_group_acc_device_ptr(rt_group* g, void** out) {
    if (!g || !out) return RT_ERR_INVALID;
    *out = g->last_mode == RT_GROUP_SAMPLES ? (void*)g->d_total : (void*)g->ctx[0]->d_acc;
    if (!*out) return gfail(g, RT_ERR_INVALID, "nothing has been rendered");
    return RT_OK;
}

int rt_group_read_acc(rt_group* g, float* rgba, size_t n_pixels) {
    if (!g || !rgba) return RT_ERR_INVALID;
    rt_ctx* c0 = g->ctx[0];
    if (g->last_mode != RT_GROUP_SAMPLES) {
        const int rc = rt_read_acc(c0, rgba, n_pixels);
        return rc == RT_OK ? rc : gfail(g, rc, rt_last_error(c0));
    }
    if (!g->d_total || n_pixels != g->total_pixels) return gfail(g, RT_ERR_INVALID, "n_pixels does not match the canvas");
    RT_GHIP(g, hipSetDevice(c0->device));
    RT_GHIP(g, hipMemcpyAsync(rgba, g->d_total, n_pixels * 16, hipMemcpyDeviceToHost, g->stream[0]));
    RT_GHIP(g, hipStreamSynchronize(g->stream[0]));
    return RT_OK;
}

int rt_group_read_present(rt_group* g, float* rgba, size_t n_pixels) {
    if (!g || !rgba) return RT_ERR_INVALID;
    const int rc = rt_read_present(g->ctx[0], rgba, n_pixels);
    return rc == RT_OK ? rc : gfail(g, rc, rt_last_error(g->ctx[0]));
}

}  // extern "C"
