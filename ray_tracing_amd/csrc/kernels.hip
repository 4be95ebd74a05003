// HIP kernels for gfx950 (MI355X).  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "trace_device.hpp"
#include "trifilter.hpp"
#include "wave_scan.hpp"

namespace rt {

using namespace dev;

namespace {

constexpr int kBlock = 256;  // 4 waves
constexpr uint32_t kPT = 8;         // side of the persistent kernel's pixel tiles
constexpr uint32_t kPTShift = 6;   // log2(kPT * kPT)
constexpr int kTileW = 32;   // plain kernel: block tile 32 x 8 pixels, one 8x8 tile per wave
constexpr int kTileH = 8;

// Sum over the 64 lanes of a wave; the total is valid in lane 0.
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Consecutive workgroup ids are dealt round-robin over the 8 XCDs, each with
// its own L2.  Give every XCD a contiguous run of tiles (so neighbouring
// tiles, which walk the same BVH nodes, share an L2) with a mapping that stays
// a bijection when the tile count is not a multiple of 8.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t n) {
    uint32_t xcd = bid & 7u, k = bid >> 3;
    uint32_t q = n >> 3, r = n & 7u;
    uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + k;
}

// Row `ly` of this context's share of the canvas -> canvas row.
__device__ __forceinline__ uint32_t global_row(const DevRows& rows, uint32_t ly) {
    if (rows.world <= 1u) return ly;
    uint32_t s = ly / rows.strip_h, o = ly - s * rows.strip_h;
    return (s * rows.world + rows.rank) * rows.strip_h + o;
}

__device__ __forceinline__ void flush_stats(const DevFrame& fr, bool count, uint32_t lane, unsigned long long n_rays,
                                            unsigned long long n_paths, unsigned long long n_box,
                                            unsigned long long n_internal, unsigned long long n_tri) {
    n_rays = wave_sum(n_rays);
    n_paths = wave_sum(n_paths);
    if (count) {
        n_box = wave_sum(n_box);
        n_internal = wave_sum(n_internal);
        n_tri = wave_sum(n_tri);
    }
    if (lane == 0 && fr.stats) {
        atomicAdd(&fr.stats->rays, n_rays);
        atomicAdd(&fr.stats->paths, n_paths);
        if (count) {
            atomicAdd(&fr.stats->box_tests, n_box);
            atomicAdd(&fr.stats->internal_visits, n_internal);
            atomicAdd(&fr.stats->triangle_tests, n_tri);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// Plain whole-path kernel: the shape of the reference's shader::main
// (shader/src/lib.rs:23-109).  One lane = one pixel = `n_frames` consecutive
// one-sample frames of that pixel; the accumulator value stays in registers
// between them (a pixel's frame k+1 depends only on its own frame k through
// the seed, lib.rs:57-60) and is written once.  Kept as the simplest correct
// form; the persistent kernel below computes the same thing faster.
// ---------------------------------------------------------------------------
template <int STACK_N, bool COUNT>
__global__ __launch_bounds__(kBlock) void path_kernel(DevScene sc, DevCam cam, DevRows rows, DevFrame fr,
                                                      uint32_t n_frames, uint32_t tiles_x, uint32_t n_tiles) {
    __shared__ uint32_t s_stack[STACK_N * kBlock];

    const uint32_t tile = xcd_remap(blockIdx.x, n_tiles);
    const uint32_t tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = tile_x * kTileW + wave * 8u + (lane & 7u);
    const uint32_t ly = tile_y * kTileH + (lane >> 3);
    const bool inside = x < cam.W && ly < rows.local_rows;
    const uint32_t y = global_row(rows, ly);

    unsigned long long n_rays = 0, n_box = 0, n_internal = 0, n_tri = 0, n_paths = 0;

    if (inside && y < cam.H) {
        uint32_t* stack = &s_stack[threadIdx.x];
        const uint32_t idx = x + y * cam.W;
        float4 acc = reinterpret_cast<const float4*>(fr.acc)[idx];
        float4 shown = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        HitRecord first;
        record_new(first);

        for (uint32_t f = 0; f < n_frames; ++f) {
            // shader/src/lib.rs:57-69
            uint32_t seed = cam.seed_mode ? get_seed(cam.frame, x, y, 0.0f, 0.0f, 0.0f)
                                          : get_seed(cam.frame, x, y, acc.x, acc.y, acc.z);
            // Ray::get_color, trace.rs:175-206
            V3 color = v3(1.0f, 1.0f, 1.0f);
            float jx = rand_float(seed, 0.0f, 1.0f);
            float jy = rand_float(seed, 0.0f, 1.0f);
            Ray ray = camera_ray(cam, (float)x + jx, (float)y + jy);
            ray.dir = normalize(ray.dir);
            bool finished = false;
            for (uint32_t seg = 0; seg < cam.depth; ++seg) {
                HitRecord rec;
                closest_hit<COUNT>(sc, ray, rec, stack, kBlock);
                n_rays += 1;
                if (COUNT) {
                    n_box += rec.box_tests;
                    n_internal += rec.internal_visits;
                    n_tri += rec.triangle_tests;
                }
                if (seg == 0) first = rec;
                int state = shade(sc, cam, rec, ray, seed, color);
                if (state != STATE_RAY) {
                    finished = true;
                    break;
                }
            }
            if (!finished) color = v3(0.0f, 0.0f, 0.0f);  // trace.rs:204-205
            n_paths += 1;
            float4 next_acc;
            accumulate(cam.frames_without_move, acc, color, next_acc, shown);
            acc = next_acc;
            // the event loop's bookkeeping between frames, main.rs:147-148
            cam.frame += cam.frame_stride;
            cam.frames_without_move += 1.0f;
        }

        reinterpret_cast<float4*>(fr.acc)[idx] = acc;
        if (fr.present) reinterpret_cast<float4*>(fr.present)[idx] = gamma_correct(shown);
        if (fr.hit_triangle) {
            fr.hit_triangle[idx] = first.triangle_id;
            fr.hit_instance[idx] = first.instance_id;
            fr.hit_t_bits[idx] = __float_as_uint(first.t);
        }
        if (COUNT && fr.box_tests) {
            fr.box_tests[idx] = first.box_tests;
            fr.triangle_tests[idx] = first.triangle_tests;
        }
    }
    flush_stats(fr, COUNT, lane, n_rays, n_paths, n_box, n_internal, n_tri);
}

// ---------------------------------------------------------------------------
// Persistent whole-path kernel.
//
// Same per-lane work, in the same per-lane order, as path_kernel -- so the
// results are identical -- but scheduled for 64-wide waves:
//
//  * Each lane is a small state machine that owns ONE pending 64-byte item: an
//    internal node (both children).  The load of the item is issued at the end
//    of one step and consumed by the next, so a traversing lane never has two
//    dependent fetches in a step.  Items are named by 32-bit byte offsets into
//    one arena.
//  * A step is one of three kinds: CONTROL (generate a path, start a segment,
//    walk the instance list -- ray into the instance's space, root box -- until
//    an instance is entered, shade + accumulate after the last instance), BOX
//    (test the child box on top of the reference's stack, descend or pop) and
//    LEAF (test the triangles of a leaf).  Per iteration the wave runs only the
//    kind most of its lanes are waiting for; the others keep their item and
//    wait.  No lane ever idles until its neighbours finish a subtree, an
//    instance, a bounce or a frame.
//  * The instance walk of a CONTROL step runs over the instance indices in
//    order, wave-uniformly, every lane joining in at its own index: instance
//    records are scalar loads, and an instance whose root box is missed costs a
//    lane no step of its own.
//  * The BOX and LEAF steps are written without divergent branches: every
//    lane computes, lanes that do not take part discard the results through
//    selects.  The arithmetic of a BOX step covers both children of the node at
//    once with packed 2 x f32 instructions.
//  * A LEAF step is a job of the whole wave: the triangles of all lanes that
//    stand at a leaf are dealt out over the 64 lanes, tested in parallel, and
//    each owner then picks from its results what the sequential loop of
//    hit.rs:111-121 would have kept.
//  * Waves are persistent: a lane whose pixel has had all its frames pulls the
//    next pixel from a work queue, so lanes never retire early.  There is one
//    queue per XCD group (workgroups with equal blockIdx % 8 share an L2), each
//    covering a contiguous range of 8x8 pixel tiles; a wave that drains its own
//    queue moves on to the next one.
// ---------------------------------------------------------------------------
enum : uint32_t { ST_CONTROL = 0, ST_DESC = 1, ST_POP = 2, ST_LEAF = 3, ST_DEAD = 4 };
enum : uint32_t { CT_NEW_PIXEL = 0, CT_NEW_PATH = 1, CT_START_SEGMENT = 2, CT_NEXT_INSTANCE = 3, CT_SHADE = 4 };
// instance records read with a wave-uniform index: the constant address space lets them come
// through the scalar cache into scalar registers
typedef const __attribute__((address_space(4))) DevInstance* ConstInstance;
constexpr uint32_t kNoItem = 0xFFFFFFFFu;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 vmin(f2 a, f2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ f2 vmax(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }

// The fast variant is held to 128 registers (4 waves per SIMD); the counting variant, which
// carries the counters and the scheduling profile as well, is left alone.
// BIG: items are named by record number instead of byte offset (arenas of 4 GiB and more); the
// address of an item then takes 64-bit arithmetic, two or three more instructions per fetch.
template <int STACK_N, bool COUNT, bool BIG>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((COUNT || STACK_N > 32) ? 1 : 4))) void path_kernel_persistent(DevScene sc, DevCam cam, DevRows rows, DevFrame fr,
                                                                 DevQueue q, uint32_t n_frames) {
    __shared__ uint32_t s_stack[STACK_N * kBlock];
    __shared__ uint32_t s_tmp[kBlock];  // 64 words per wave: who works for whom in a LEAF step
    __shared__ unsigned long long s_best[kBlock];  // per lane: what an owner keeps of a LEAF step's results

    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* const stack = &s_stack[threadIdx.x];
    const unsigned long long lanes_below = (1ull << lane) - 1ull;

    unsigned long long n_rays = 0, n_box = 0, n_internal = 0, n_tri = 0, n_paths = 0;
    unsigned long long sched_iters[3] = {0, 0, 0}, sched_lanes[3] = {0, 0, 0}, sched_cycles[3] = {0, 0, 0},
                       sched_wait[3] = {0, 0, 0};  // wave-uniform
    unsigned long long site_wait[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-uniform, profile only
    // profile only: wait here for every outstanding memory operation and book the time to `site`
    auto wait_site = [&](int site) {
        if (COUNT) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            site_wait[site] += __builtin_readcyclecounter() - t0;
        }
    };

    // what names node record `a` / where a named item and a triangle's test record live
    auto node_item = [&](uint32_t a) -> uint32_t { return BIG ? sc.node_unit + a : sc.node_off + a * 64u; };
    auto item_ptr = [&](uint32_t item) -> const f4* {
        return reinterpret_cast<const f4*>(BIG ? sc.arena + ((unsigned long long)item << 6) : sc.arena + item);
    };
    auto tri_test_ptr = [&](uint32_t tri) -> const f4* {
        return reinterpret_cast<const f4*>(BIG ? sc.arena + ((unsigned long long)(sc.tri_test_unit + tri) << 6)
                                               : sc.arena + (sc.tri_test_off + tri * 64u));
    };
    const ConstInstance inst_base = (ConstInstance)(uintptr_t)sc.instances;
    uint32_t phase = n_frames > 0 ? ST_CONTROL : ST_DEAD;
    uint32_t ctl = CT_NEW_PIXEL;
    uint32_t queues_tried = 0;  // wave-uniform: how many queues this wave has found empty
    // pixel state
    uint32_t x = 0, y = 0, tile = 0;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t frame = cam.frame;
    float fwm = cam.frames_without_move;
    uint32_t frames_left = 0;
    // path state
    uint32_t seed = 0, seg = 0, inst = 0;
    V3 color = v3(1.0f, 1.0f, 1.0f);
    Ray ray;
    ray.pos = v3(0.0f, 0.0f, 0.0f);
    ray.dir = v3(0.0f, 0.0f, 1.0f);
    HitRecord rec;  // rec.t doubles as the traversal's clamp: trace.rs:89 hands it to hit_bvh as t_max
    record_new(rec);
    // traversal state of the current (segment, instance): the ray in object space as
    // (origin, direction) pairs per axis, the reciprocals of the direction
    f2 dpx = {0.0f, 0.0f}, dpy = {0.0f, 0.0f}, dpz = {0.0f, 1.0f};
    float rcx = 1.0f, rcy = 1.0f, rcz = 1.0f;
    bool rcp_ok = false;
    bool cull = false;
    f4 it0 = {0.0f, 0.0f, 0.0f, 0.0f}, it1 = it0, it2 = it0, it3 = it0;  // the pending item
    uint32_t cur = 0;  // ST_DESC / ST_POP: (node index << 1) | child slot;  ST_LEAF: triangle index
    uint32_t tri_last = 0, sp = 0;

    // Wave-uniform condition: lanes that have run out of pixels stay in the loop, inert, until
    // the whole wave is done, so that the cross-lane operations of the LEAF step always see all
    // 64 lanes.
    while (__ballot(phase != ST_DEAD) != 0ull) {
        uint32_t next_off = kNoItem;
        const unsigned long long t_begin = COUNT ? __builtin_readcyclecounter() : 0ull;
        // wave-uniform choice of the one kind of step this iteration runs
        const int w_ctl = __popcll(__ballot(phase == ST_CONTROL)) * q.w_ctl;
        const int w_box = __popcll(__ballot(phase == ST_DESC || phase == ST_POP)) * q.w_box;
        const int w_leaf = __popcll(__ballot(phase == ST_LEAF)) * q.w_leaf;
        int pick = 1, w_best = w_box;
        if (w_leaf > w_best) pick = 2, w_best = w_leaf;
        if (w_ctl > w_best) pick = 0;
        if (COUNT) {
            // profile only: how long the prefetched item still takes to arrive
            const unsigned long long t0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sched_wait[pick] += __builtin_readcyclecounter() - t0;
            sched_iters[pick] += 1;
            sched_lanes[pick] += (unsigned)(pick == 0 ? w_ctl / q.w_ctl : pick == 1 ? w_box / q.w_box : w_leaf / q.w_leaf);
        }

        if (pick == 0) {
            // one queue per XCD group; every lane that needs a pixel takes part in one fetch
            const bool want = phase == ST_CONTROL && ctl == CT_NEW_PIXEL;
            const unsigned long long wm = __ballot(want);
            if (wm != 0ull) {
                const uint32_t n_want = (uint32_t)__popcll(wm);
                const uint32_t qi = ((blockIdx.x & 7u) + queues_tried) & 7u;
                const uint32_t q_first = q.first_item[qi], q_size = q.first_item[qi + 1] - q_first;
                uint32_t base = 0;
                if (lane == (uint32_t)__ffsll((long long)wm) - 1u) base = atomicAdd(&q.head[qi * 32u], n_want);
                base = __shfl(base, __ffsll((long long)wm) - 1, 64);
                if (base + n_want >= q_size) queues_tried += 1u;  // this queue is (about to be) empty
                if (want) {
                    const uint32_t k = base + (uint32_t)__popcll(wm & lanes_below);
                    if (k < q_size) {
                        const uint32_t item = q_first + k;
                        const uint32_t l = item & (kPT * kPT - 1u);
                        tile = q.order ? q.order[item >> kPTShift] : item >> kPTShift;
                        const uint32_t tile_y = tile / q.tiles_x, tile_x = tile - tile_y * q.tiles_x;
                        x = tile_x * kPT + (l & (kPT - 1u));
                        const uint32_t ly = tile_y * kPT + l / kPT;
                        y = global_row(rows, ly);
                        if (x < cam.W && ly < rows.local_rows && y < cam.H) {
                            acc = reinterpret_cast<const float4*>(fr.acc)[x + y * cam.W];
                            frame = cam.frame;
                            fwm = cam.frames_without_move;
                            frames_left = n_frames;
                            ctl = CT_NEW_PATH;
                        }
                    } else if (queues_tried >= 8u) {
                        phase = ST_DEAD;
                    }
                }
                wait_site(0);
            }
            // Two rounds of: (1) the instance loop of trace_ray (trace.rs:76-94) for every lane that
            // stands between two instances, (2) the rest of trace_ray and whatever follows it for
            // the lanes whose instances are exhausted.  A lane that starts a new segment in (2)
            // of the first round finds its first instance in (1) of the second.
            for (int round = 0; round < 2; ++round) {
                // (1) One pass over the instance indices, in order; index u is wave-uniform, so the
                // instance record comes through the scalar cache.  A lane takes part at its own
                // index: the ray moves into the instance's object space, the root box is tested
                // against the clamp (EPS, record.t) -- rec.t itself -- and the lane either enters
                // the instance's tree or moves on to index u + 1 within the same pass.
                bool looking = phase == ST_CONTROL && ctl == CT_NEXT_INSTANCE;
                for (uint32_t u = 0; u < sc.n_instances; ++u) {
                    const bool mine = looking && inst == u;
                    if (__ballot(mine) == 0ull) continue;
                    const ConstInstance I = inst_base + u;
                    float m[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) m[k] = I->inv[k];
                    const V3 np = xf_point(m, ray.pos), nd = xf_vector(m, ray.dir);
                    const bool in_range = dir_rcp_ok(np, nd, sc.fast_div != 0u);
                    const float blox = I->root.lox[0], bloy = I->root.loy[0], bloz = I->root.loz[0];
                    const float bhix = I->root.hix[0], bhiy = I->root.hiy[0], bhiz = I->root.hiz[0];
                    const uint32_t a = I->root.a[0], b = I->root.b[0];
                    wait_site(1);
                    const bool is_internal = b == kInternalTag, has_tris = a <= b;
                    // First from the hardware's approximate reciprocals (three instructions): most
                    // lanes miss most root boxes, clearly.  Only when some lane hits one, or cannot
                    // tell, are the correctly rounded reciprocals formed -- the traversal needs them
                    // then anyway.
                    const SlabApprox sa = slab_approx(blox, bloy, bloz, bhix, bhiy, bhiz, np.x, np.y, np.z,
                                                      __builtin_amdgcn_rcpf(nd.x), __builtin_amdgcn_rcpf(nd.y),
                                                      __builtin_amdgcn_rcpf(nd.z));
                    const float dist_apx = slab_dist(sa);
                    const bool vague = slab_ambiguous(sa, rec.t, kSlabMarginApproxRcp);
                    const bool amb = mine && (!in_range || vague);
                    const bool may_enter = amb || (mine && dist_apx != RT_INF && !(dist_apx > rec.t));
                    bool enter = false;
                    if (__ballot(may_enter) != 0ull) {
                        const DirRcp nr = make_dir_rcp(np, nd, sc.fast_div != 0u);
                        float dist = dist_apx;
                        if (__ballot(amb) != 0ull) {
                            const float4 lo4 = make_float4(blox, bloy, bloz, 0.0f), hi4 = make_float4(bhix, bhiy, bhiz, 0.0f);
                            const bool out_of_range = mine && !in_range;
                            if (__ballot(out_of_range) == 0ull) dist = hits_bounding_rcp(np, nd, nr, lo4, hi4);
                            else dist = hits_bounding(np, nd, lo4, hi4);
                        }
                        const bool hit_box = dist != RT_INF, within = !(dist > rec.t);
                        enter = mine && hit_box && within && (is_internal || has_tris);
                        const bool enter_node = enter && is_internal, enter_leaf = enter && !is_internal;
                        dpx = enter ? f2{np.x, nd.x} : dpx;
                        dpy = enter ? f2{np.y, nd.y} : dpy;
                        dpz = enter ? f2{np.z, nd.z} : dpz;
                        rcx = enter ? nr.x : rcx;
                        rcy = enter ? nr.y : rcy;
                        rcz = enter ? nr.z : rcz;
                        rcp_ok = enter ? nr.ok : rcp_ok;
                        cull = enter ? I->cull != 0u : cull;
                        sp = enter ? 0u : sp;
                        phase = enter_node ? (uint32_t)ST_DESC : enter_leaf ? (uint32_t)ST_LEAF : phase;
                        cur = enter_node ? a << 1 : enter_leaf ? a : cur;
                        tri_last = enter_leaf ? b : tri_last;
                        next_off = enter_node ? node_item(a) : next_off;
                    }
                    if (COUNT) rec.box_tests += mine ? 1u : 0u;
                    inst = (mine && !enter) ? u + 1u : inst;
                    looking = looking && !enter;
                }
                if (looking) ctl = CT_SHADE;  // every instance has been visited
                if (round == 1) break;  // whoever is through with its instances now is shaded in the next CONTROL step
                // (2)
                if (phase == ST_CONTROL) {
                    if (ctl == CT_SHADE) {
                        // the rest of trace_ray, trace.rs:97-172
                        const uint32_t idx = x + y * cam.W;
                        if (seg == 0 && fr.hit_triangle) {
                            fr.hit_triangle[idx] = rec.triangle_id;
                            fr.hit_instance[idx] = rec.instance_id;
                            fr.hit_t_bits[idx] = __float_as_uint(rec.t);
                        }
                        if (COUNT) {
                            if (seg == 0 && fr.box_tests) {
                                fr.box_tests[idx] = rec.box_tests;
                                fr.triangle_tests[idx] = rec.triangle_tests;
                            }
                            n_box += rec.box_tests;
                            n_internal += rec.internal_visits;
                            n_tri += rec.triangle_tests;
                        }
                        int state = shade(sc, cam, rec, ray, seed, color);
                        wait_site(2);
                        seg += 1;
                        if (state == STATE_RAY && seg < cam.depth) {
                            ctl = CT_START_SEGMENT;
                        } else {
                            if (state == STATE_RAY) color = v3(0.0f, 0.0f, 0.0f);  // never finished bouncing, trace.rs:204-205
                            // shader/src/lib.rs:71-98 (the presented colour is derived from the accumulator by
                            // present_kernel afterwards), then the event loop's bookkeeping, main.rs:147-148
                            acc = accumulate_only(fwm, acc, color);
                            n_paths += 1;
                            frame += cam.frame_stride;
                            fwm += 1.0f;
                            frames_left -= 1u;
                            ctl = CT_NEW_PATH;
                            if (frames_left == 0u) {
                                reinterpret_cast<float4*>(fr.acc)[idx] = acc;
                                if (q.cost) atomicAdd(&q.cost[tile], seg);  // length of the pixel's last path
                                ctl = CT_NEW_PIXEL;
                            }
                        }
                    }
                    if (ctl == CT_NEW_PATH) {
                        // shader/src/lib.rs:57-69 and Ray::get_color, trace.rs:175-191
                        seed = cam.seed_mode ? get_seed(frame, x, y, 0.0f, 0.0f, 0.0f)
                                             : get_seed(frame, x, y, acc.x, acc.y, acc.z);
                        color = v3(1.0f, 1.0f, 1.0f);
                        float jx = rand_float(seed, 0.0f, 1.0f);
                        float jy = rand_float(seed, 0.0f, 1.0f);
                        ray = camera_ray(cam, (float)x + jx, (float)y + jy);
                        ray.dir = normalize(ray.dir);
                        seg = 0;
                        ctl = CT_START_SEGMENT;
                    }
                    if (ctl == CT_START_SEGMENT) {
                        // head of Ray::trace_ray, trace.rs:73-74
                        ray.dir = normalize(ray.dir);
                        record_new(rec);
                        inst = 0;
                        n_rays += 1;
                        ctl = CT_NEXT_INSTANCE;
                    }
                }
                if (__ballot(phase == ST_CONTROL && ctl == CT_NEXT_INSTANCE) == 0ull) break;
            }
        } else {
            bool finished;  // this lane has to pop (or its instance is done)
            const V3 lpos = v3(dpx.x, dpy.x, dpz.x), ldir = v3(dpx.y, dpy.y, dpz.y);
            if (pick == 2) {
                // `for i in first..=last`, hit.rs:111-121, for every lane that stands at a leaf --
                // shared out over the whole wave.  The (lane, triangle) pairs of all those lanes
                // ("owners") are numbered consecutively; worker lane w takes pair w: it borrows the
                // owner's ray, fetches the triangle's DevTriTest and tests it (plane and lower range
                // check as in triangle_ray_intersect, then the three edge signs from the approximate
                // edge functions; unclear signs -> the exact test on the full triangle).  The upper
                // range check, t > t_max, is the one part of the test that depends on the triangles
                // before it, so it is left to the owner: each owner walks through its results in
                // triangle order and keeps a candidate when it is not beyond the best so far --
                // which lets a later triangle at the same distance win, as in hit.rs:115, and lets
                // anything follow a NaN distance (a zero-area triangle), as the reference's
                // comparisons do.  One pass covers 64 pairs; the rest stays for the next LEAF step.
                const bool owner = phase == ST_LEAF;
                const uint32_t c = owner ? tri_last - cur + 1u : 0u;
                const uint32_t incl = wave_incl_sum(c), excl = incl - c;
                const uint32_t room = excl < 64u ? 64u - excl : 0u;
                const uint32_t n_proc = c < room ? c : room;
                // worker w -> owner: owners mark the first of their pairs, a running maximum spreads the mark
                // (a wave's LDS operations complete in program order; the empty asm statements keep the
                // compiler from forwarding this lane's own store to its load, which must see other
                // lanes' stores)
                uint32_t* const tmp = &s_tmp[threadIdx.x & ~63u];
                tmp[lane] = 0u;
                asm volatile("" ::: "memory");
                if (n_proc != 0u) tmp[excl] = lane + 1u;
                asm volatile("" ::: "memory");
                uint32_t tmp_mine = tmp[lane];
                wait_site(4);
                const uint32_t own1 = wave_incl_max(tmp_mine);
                const uint32_t o = own1 != 0u ? own1 - 1u : lane;
                // the owner's first pair, pair count and culling rule travel in one word
                const uint32_t o_info = (uint32_t)__shfl((int)(excl | (n_proc << 8) | (cull ? 1u << 16 : 0u)), (int)o, 64);
                const uint32_t kw = lane - (o_info & 0xFFu);
                const bool has_owner = own1 != 0u, in_range = kw < ((o_info >> 8) & 0xFFu);
                const bool work = has_owner && in_range;
                const uint32_t tri = (uint32_t)__shfl((int)cur, (int)o, 64) + kw;
                f4 b0 = {0.0f, 0.0f, 0.0f, 0.0f}, b1 = b0, b2 = b0, b3 = b0;
                if (work) {
                    const f4* p = tri_test_ptr(tri);
                    b0 = p[0];
                    b1 = p[1];
                    b2 = p[2];
                    b3 = p[3];
                }
                // the owner's ray, clamp and culling rule
                const f2 wx = {__shfl(dpx.x, (int)o, 64), __shfl(dpx.y, (int)o, 64)};
                const f2 wy = {__shfl(dpy.x, (int)o, 64), __shfl(dpy.y, (int)o, 64)};
                const f2 wz = {__shfl(dpz.x, (int)o, 64), __shfl(dpz.y, (int)o, 64)};
                const bool w_cull = (o_info & (1u << 16)) != 0u;
                wait_site(5);
                const V3 wpos = v3(wx.x, wy.x, wz.x), wdir = v3(wx.y, wy.y, wz.y);
                // (n . pos, n . dir) in one go: dot's order, (x + y) + z, in both halves
                const f2 np_nd = (splat(b0.x) * wx + splat(b0.y) * wy) + splat(b0.z) * wz;
                const float dot_prod = np_nd.y;
                const bool grazing = fabsf(dot_prod) < kEps, back = dot_prod > 0.0f;
                float t = -(np_nd.x + b0.w) / dot_prod;
                const bool before = t < kEps;
                const bool back_culled = w_cull && back;
                const bool out = grazing || back_culled || before;
                const V3 g = (wpos + wdir * t) - v3(b1.x, b1.y, b1.z);
                const f2 s01 = (b2.xy * splat(g.x) + b2.zw * splat(g.y)) + b3.xy * splat(g.z);
                const TriSigns sg = tri_filter_signs(s01.x, s01.y, b1.w, b3.z, b3.w, g.x, g.y, g.z);
                const bool vague = tri_filter_unclear(sg), outside = sg.s_min < 0.0f;
                const bool unclear = work && !out && vague;
                const bool miss = !work || out || outside;
                t = miss ? RT_INF : t;
                if (__ballot(unclear) != 0ull) {
                    if (unclear) {
                        const float4* fp = reinterpret_cast<const float4*>(&sc.tris[tri]);
                        t = triangle_ray_intersect(fp[0], fp[1], fp[2], fp[3].x, wpos, wdir, kEps, RT_INF, w_cull);
                    }
                    wait_site(6);
                }
                // What an owner keeps of its results is what the sequential loop would keep: a
                // candidate (a finite distance) is accepted when it is not beyond the best so far,
                // so the survivor is the smallest candidate that is not beyond the clamp, the last
                // one among equals.  With no NaN among the results that is a minimum over
                // (distance, position from the end): the workers fold their candidates into the
                // owner's LDS slot with one 64-bit atomic minimum each (distances are >= EPS > 0,
                // so their bit patterns order like the numbers).  A NaN clamp accepts everything,
                // as `t > NaN` is false.  A NaN result (a zero-area triangle) makes what follows
                // it depend on the order, so such a wave walks through the results one by one.
                float best = rec.t;
                uint32_t best_k = 0xFFFFFFFFu;
                const bool t_nan = t != t;
                if (__ballot(t_nan) == 0ull) {
                    const float o_clamp = __shfl(rec.t, (int)o, 64);
                    unsigned long long* const slot = &s_best[threadIdx.x & ~63u];
                    slot[lane] = ~0ull;
                    asm volatile("" ::: "memory");
                    const bool cand = work && !is_inf_bits(t) && !(t > o_clamp);
                    if (cand) atomicMin(&slot[o], ((unsigned long long)__float_as_uint(t) << 32) | (0xFFFFFFFFu - kw));
                    asm volatile("" ::: "memory");
                    const unsigned long long kept = slot[lane];
                    wait_site(7);
                    const bool any = owner && kept != ~0ull;
                    best = any ? __uint_as_float((uint32_t)(kept >> 32)) : best;
                    best_k = any ? 0xFFFFFFFFu - (uint32_t)kept : best_k;
                } else {
                    for (uint32_t k0 = 0; __ballot(k0 < n_proc) != 0ull; k0 += 4u) {
                        float tk[4];  // four results on their way at a time
#pragma unroll
                        for (uint32_t j = 0; j < 4u; ++j) tk[j] = __shfl(t, (int)((excl + k0 + j) & 63u), 64);
#pragma unroll
                        for (uint32_t j = 0; j < 4u; ++j) {
                            const bool mine = k0 + j < n_proc, finite = !is_inf_bits(tk[j]), farther = tk[j] > best;
                            const bool keep = mine && finite && !farther;
                            best = keep ? tk[j] : best;
                            best_k = keep ? k0 + j : best_k;
                        }
                    }
                }
                const bool hit = best_k != 0xFFFFFFFFu;
                rec.t = hit ? best : rec.t;
                rec.triangle_id = hit ? cur + best_k : rec.triangle_id;
                rec.instance_id = hit ? inst : rec.instance_id;
                if (COUNT) rec.triangle_tests += n_proc;
                cur += n_proc;
                finished = owner && cur == tri_last + 1u;
            } else {
                // The item is an internal node; both child boxes are tested at once.
                //   ST_DESC: the node passed its own test (hit.rs:88-107).  The children are
                //     ordered by centre distance; the nearer one is the node the reference pops
                //     next.  The farther one it would test when popping it later: a ray that
                //     misses that box altogether misses it whenever it is popped, so that outcome
                //     is taken now and the entry is never pushed; a box that IS hit still waits on
                //     the stack and is tested again at pop time against the clamp of that moment.
                //   ST_POP: only child `cur & 1` is looked at: the node on top of the reference's
                //     stack (hit.rs:77-86).
                const bool desc = phase == ST_DESC;
                const bool popping = phase == ST_POP;
                const bool act = desc || popping;
                const float lox[2] = {it0.x, it0.y}, hix[2] = {it0.z, it0.w};
                const float loy[2] = {it1.x, it1.y}, hiy[2] = {it1.z, it1.w};
                const float loz[2] = {it2.x, it2.y}, hiz[2] = {it2.z, it2.w};
                const uint32_t a0 = __float_as_uint(it3.x), a1 = __float_as_uint(it3.y);
                const uint32_t b0 = __float_as_uint(it3.z), b1 = __float_as_uint(it3.w);
                const float px = dpx.x, py = dpy.x, pz = dpz.x;
                // BoundingBox::center (shared/src/lib.rs:155-157) and hit.rs:92-100
                float d2c[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float ex = (lox[k] + hix[k]) * 0.5f - px, ey = (loy[k] + hiy[k]) * 0.5f - py,
                                ez = (loz[k] + hiz[k]) * 0.5f - pz;
                    d2c[k] = (ex * ex + ey * ey) + ez * ez;
                }
                const bool first_is_1 = d2c[0] < d2c[1];
                // the child tested "now" is child 0.  Every condition of this step is computed into a
                // named flag first and the flags are then combined: the step is meant to be one
                // straight run of instructions, not a tree of branches.
                const bool slot0 = (cur & 1u) == 0u;
                const bool by_order = desc && first_is_1, by_slot = !desc && slot0;
                const bool near0 = by_order || by_slot;
                if (COUNT) {
                    // a popped child was counted when its parent was visited
                    rec.internal_visits += desc ? 1u : 0u;
                    rec.box_tests += desc ? 2u : 0u;
                }
                // Three ways to the same decisions, chosen per wave.  The plain division is right
                // for every lane; quotients through the cached reciprocal need the lane's operands
                // in range; approximate quotients also need every comparison to be clear-cut.
                // One lane outside a premise moves the whole wave to the next, dearer form.
                // The approximate form picks each slab's entry and exit plane by the sign of the
                // reciprocal (a bitwise select) instead of ordering the two quotients: the same values
                // whenever min <= max and the reciprocal is finite, which the premises include.
                const uint32_t mx = (uint32_t)((int32_t)__float_as_uint(rcx) >> 31),
                               my = (uint32_t)((int32_t)__float_as_uint(rcy) >> 31),
                               mz = (uint32_t)((int32_t)__float_as_uint(rcz) >> 31);
                auto pick = [](uint32_t m, float when_set, float when_clear) {
                    return __uint_as_float((__float_as_uint(when_set) & m) | (__float_as_uint(when_clear) & ~m));
                };
                float tn_[2], tf_[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float nx = (pick(mx, hix[k], lox[k]) - px) * rcx, fx = (pick(mx, lox[k], hix[k]) - px) * rcx;
                    const float ny = (pick(my, hiy[k], loy[k]) - py) * rcy, fy = (pick(my, loy[k], hiy[k]) - py) * rcy;
                    const float nz = (pick(mz, hiz[k], loz[k]) - pz) * rcz, fz = (pick(mz, loz[k], hiz[k]) - pz) * rcz;
                    tn_[k] = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
                    tf_[k] = __builtin_fminf(__builtin_fminf(fx, fy), fz);
                }
                const f2 tn = {tn_[0], tn_[1]}, tf = {tf_[0], tf_[1]};
                const SlabApprox s0 = {tn.x, tf.x}, s1 = {tn.y, tf.y};
                const bool amb0 = slab_ambiguous(s0, rec.t), amb1 = slab_ambiguous(s1, rec.t);
                // a descent needs both children clear-cut, a pop only the child it looks at
                const bool look0 = desc || near0, look1 = desc || !near0;
                const bool unclear0 = amb0 && look0, unclear1 = amb1 && look1;
                const bool amb = act && (!rcp_ok || unclear0 || unclear1);
                float dist0 = slab_dist(s0), dist1 = slab_dist(s1);
                if (__ballot(amb) != 0ull) {
                    const bool out_of_range = act && !rcp_ok;
                    const bool all_fast = __ballot(out_of_range) == 0ull;
                    const float4 lo0 = make_float4(lox[0], loy[0], loz[0], 0.0f), hi0 = make_float4(hix[0], hiy[0], hiz[0], 0.0f);
                    const float4 lo1 = make_float4(lox[1], loy[1], loz[1], 0.0f), hi1 = make_float4(hix[1], hiy[1], hiz[1], 0.0f);
                    if (all_fast) {
                        DirRcp r;
                        r.x = rcx;
                        r.y = rcy;
                        r.z = rcz;
                        r.ok = true;
                        dist0 = hits_bounding_rcp(lpos, ldir, r, lo0, hi0);
                        dist1 = hits_bounding_rcp(lpos, ldir, r, lo1, hi1);
                    } else {
                        dist0 = hits_bounding(lpos, ldir, lo0, hi0);
                        dist1 = hits_bounding(lpos, ldir, lo1, hi1);
                    }
                }
                const float near_dist = near0 ? dist0 : dist1, far_dist = near0 ? dist1 : dist0;
                const bool far_hit = far_dist != RT_INF, far_within = !(far_dist > rec.t);
                const bool near_hit = near_dist != RT_INF, near_within = !(near_dist > rec.t);
                const bool far_alive = desc && far_hit;
                const bool near_alive = near_hit && near_within;
                const bool push = near_alive && far_alive;
                // the nearer child is out: the reference pops the farther one next, clamp unchanged
                const bool take_far = !near_alive && far_alive;
                const bool far_taken = take_far && far_within;
                const bool alive = near_alive || far_taken;
                if (push) stack[sp * kBlock] = (cur & ~1u) | (near0 ? 1u : 0u);  // the farther child's slot
                sp += push ? 1u : 0u;
                const bool chosen0 = near0 != take_far;
                const uint32_t a = chosen0 ? a0 : a1, b = chosen0 ? b0 : b1;
                const bool is_internal = b == kInternalTag, has_tris = a <= b;  // a > b: `first..=last` with nothing in it
                const bool go = act && alive;
                const bool internal = go && is_internal;
                const bool leaf = go && !is_internal && has_tris;
                finished = act && !internal && !leaf;
                phase = internal ? (uint32_t)ST_DESC : leaf ? (uint32_t)ST_LEAF : phase;
                cur = internal ? a << 1 : leaf ? a : cur;
                tri_last = leaf ? b : tri_last;
                next_off = internal ? node_item(a) : next_off;  // a leaf's triangles are fetched by the LEAF step
            }
            // a lane that is through with its node or leaf: pop, or this instance is done (trace.rs:76)
            const bool empty = sp == 0u;
            const bool pop = finished && !empty;
            const bool inst_end = finished && empty;
            uint32_t ref = 0;
            if (pop) ref = stack[(sp - 1u) * kBlock];
            wait_site(3);
            sp -= pop ? 1u : 0u;
            inst += inst_end ? 1u : 0u;
            phase = pop ? (uint32_t)ST_POP : inst_end ? (uint32_t)ST_CONTROL : phase;
            cur = pop ? ref : cur;
            ctl = inst_end ? (uint32_t)CT_NEXT_INSTANCE : ctl;  // on to the next instance, if there is one
            next_off = pop ? node_item(ref >> 1) : next_off;
        }
        if (next_off != kNoItem) {
            const f4* p = item_ptr(next_off);
            it0 = p[0];
            it1 = p[1];
            it2 = p[2];
            it3 = p[3];
        }
        if (COUNT) sched_cycles[pick] += __builtin_readcyclecounter() - t_begin;
    }
    flush_stats(fr, COUNT, lane, n_rays, n_paths, n_box, n_internal, n_tri);
    if (COUNT && lane == 0 && fr.stats) {
        for (int k = 0; k < 3; ++k) {
            atomicAdd(&fr.stats->iters[k], sched_iters[k]);
            atomicAdd(&fr.stats->lanes[k], sched_lanes[k]);
            atomicAdd(&fr.stats->cycles[k], sched_cycles[k]);
            atomicAdd(&fr.stats->wait_cycles[k], sched_wait[k]);
        }
        for (int k = 0; k < 8; ++k) atomicAdd(&fr.stats->site_wait[k], site_wait[k]);
    }
}

// The image the reference shows after a frame (lib.rs:85-108): the accumulator divided by
// the sample count, gamma 1/2.2.  `fwm_last` is frames_without_move of the last frame
// rendered into the accumulator.  Launched after the persistent kernel, which leaves the
// presented image to this pass: the colour is a function of the stored accumulator alone.
__global__ __launch_bounds__(kBlock) void present_kernel(DevCam cam, DevRows rows, DevFrame fr, float fwm_last) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= cam.W * rows.local_rows) return;
    const uint32_t ly = i / cam.W, x = i - ly * cam.W;
    const uint32_t y = global_row(rows, ly);
    if (y >= cam.H) return;
    const uint32_t idx = x + y * cam.W;
    const float4 acc = reinterpret_cast<const float4*>(fr.acc)[idx];
    reinterpret_cast<float4*>(fr.present)[idx] = gamma_correct(present_from_acc(fwm_last, acc));
}

// Next launch's tile order from this launch's costs: a counting sort into 32 cost classes,
// dearest class first.  Within a class the order is whatever the atomics make it: the order of
// pixels changes no pixel's value.  Two ways of sharing the tiles out over the 8 queues:
//   banded (8 workgroups, one per queue): every queue keeps its contiguous band of the canvas --
//     an XCD's L2 keeps seeing neighbouring rays -- and sorts the tiles inside it;
//   dealt (1 workgroup): all tiles are sorted together and dealt to the queues in turn, so every
//     queue runs from the dearest tiles of the canvas to the cheapest and the launch ends on
//     short pixels everywhere.  That shortens the tail, which is what a launch of one or two
//     frames mostly consists of; over many frames the bands' locality is worth more.
__global__ __launch_bounds__(kBlock) void order_tiles_kernel(DevQueue q, uint32_t* order_out) {
    __shared__ uint32_t s_count[32], s_start[32];
    const bool dealt = q.deal != 0u;
    const uint32_t t0 = dealt ? 0u : q.first_item[blockIdx.x] >> kPTShift;
    const uint32_t t1 = (dealt ? q.first_item[8] : q.first_item[blockIdx.x + 1]) >> kPTShift;
    if (threadIdx.x < 32) s_count[threadIdx.x] = 0;
    __syncthreads();
    // a full tile's cost is 64 pixels x 1..10+ segments; 24 per class keeps the classes fine enough
    auto cls = [](uint32_t c) { return c * 64u / (kPT * kPT * 24u) < 31u ? c * 64u / (kPT * kPT * 24u) : 31u; };
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += kBlock) atomicAdd(&s_count[cls(q.cost[t])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        for (int c = 31; c >= 0; --c) {
            s_start[c] = at;
            at += s_count[c];
        }
    }
    __syncthreads();
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += kBlock) {
        const uint32_t r = atomicAdd(&s_start[cls(q.cost[t])], 1u);  // rank in the sorted order
        order_out[dealt ? (q.first_item[r & 7u] >> kPTShift) + (r >> 3) : t0 + r] = t;
    }
}

hipError_t launch_path_kernel(const DevScene& sc, const DevCam& cam, const DevRows& rows, const DevFrame& fr,
                              uint32_t n_frames, uint32_t stack_needed, bool count, hipStream_t stream) {
    const uint32_t tiles_x = (cam.W + kTileW - 1) / kTileW;
    const uint32_t tiles_y = (rows.local_rows + kTileH - 1) / kTileH;
    const uint32_t n_tiles = tiles_x * tiles_y;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles), block(kBlock);
#define RT_LAUNCH(N, C) \
    hipLaunchKernelGGL((path_kernel<N, C>), grid, block, 0, stream, sc, cam, rows, fr, n_frames, tiles_x, n_tiles)
    if (stack_needed <= 32) {
        if (count) RT_LAUNCH(32, true); else RT_LAUNCH(32, false);
    } else if (stack_needed <= 64) {
        if (count) RT_LAUNCH(64, true); else RT_LAUNCH(64, false);
    } else {
        return hipErrorInvalidValue;
    }
#undef RT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_persistent_kernel(const DevScene& sc, const DevCam& cam, const DevRows& rows, const DevFrame& fr,
                                    DevQueue q, uint32_t* order_out, uint32_t n_frames, uint32_t stack_needed,
                                    bool count, bool big, uint32_t n_blocks, hipStream_t stream) {
    // 8x8-pixel tiles of this context's rows, split into 8 contiguous ranges
    q.tiles_x = (cam.W + kPT - 1u) / kPT;
    const uint32_t tiles_y = (rows.local_rows + kPT - 1u) / kPT;
    const uint32_t n_tiles = q.tiles_x * tiles_y;
    if (n_tiles == 0 || n_blocks == 0) return hipSuccess;
    // queue i holds the tiles of rank i, i + 8, i + 16, ... of whatever order is in force
    q.first_item[0] = 0;
    for (uint32_t i = 0; i < 8; ++i) q.first_item[i + 1] = q.first_item[i] + ((n_tiles - i + 7u) / 8u) * (kPT * kPT);
    hipError_t e = hipMemsetAsync(q.head, 0, 8 * 32 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (q.cost && (e = hipMemsetAsync(q.cost, 0, (size_t)n_tiles * sizeof(uint32_t), stream)) != hipSuccess) return e;
    dim3 grid(n_blocks), block(kBlock);
#define RT_LAUNCH(N, C, B) \
    hipLaunchKernelGGL((path_kernel_persistent<N, C, B>), grid, block, 0, stream, sc, cam, rows, fr, q, n_frames)
#define RT_LAUNCH_NC(N, C) \
    do {                   \
        if (big) RT_LAUNCH(N, C, true); else RT_LAUNCH(N, C, false); \
    } while (0)
    if (stack_needed <= 32) {
        if (count) RT_LAUNCH_NC(32, true); else RT_LAUNCH_NC(32, false);
    } else if (stack_needed <= 64) {
        if (count) RT_LAUNCH_NC(64, true); else RT_LAUNCH_NC(64, false);
    } else {
        return hipErrorInvalidValue;
    }
#undef RT_LAUNCH_NC
#undef RT_LAUNCH
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return e2;
    if (q.cost && order_out) {
        hipLaunchKernelGGL(order_tiles_kernel, dim3(q.deal ? 1 : 8), block, 0, stream, q, order_out);
        if ((e2 = hipGetLastError()) != hipSuccess) return e2;
    }
    if (!fr.present) return hipSuccess;
    const uint32_t n_pixels = cam.W * rows.local_rows;
    // frames_without_move of the last frame, advanced the way the kernel advances it
    float fwm_last = cam.frames_without_move;
    for (uint32_t i = 1; i < n_frames; ++i) {
        const float next = fwm_last + 1.0f;
        if (next == fwm_last) break;  // the increments have stopped registering
        fwm_last = next;
    }
    hipLaunchKernelGGL(present_kernel, dim3((n_pixels + kBlock - 1) / kBlock), block, 0, stream, cam, rows, fr, fwm_last);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Scratch of the render entry point that allocates nothing (rt_render_frames_noalloc): the
// persistent kernel's queue heads, per-tile costs and the two tile-order buffers, in a
// file-scope device array instead of hipMalloc'ed memory.  Every GPU the library runs on holds
// its own copy at its own address until the process ends.  In host code the array's name is
// only a handle for hipGetSymbolAddress: the address kernels and memsets are given is the one
// that call returned for the current device.
// ---------------------------------------------------------------------------
namespace {

struct NoAllocScratch {
    uint32_t head[8 * 32];                     // queue heads, one 128-byte line each
    uint32_t cost[kNoAllocMaxTiles];
    uint32_t order[2][kNoAllocMaxTiles];
};

static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

constexpr int kMaxDevices = 64;
struct ScratchHost {
    NoAllocScratch* ptr = nullptr;  // device address of `scratch` on this GPU
    int order_current = -1;         // tile-order state of the launches that share this GPU's scratch
    uint32_t order_sig[5] = {0, 0, 0, 0, 0};
};
ScratchHost g_scratch[kMaxDevices];

}  // namespace

hipError_t noalloc_scratch_init(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (g_scratch[device].ptr) return hipSuccess;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // `device` is current
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch[device].ptr = static_cast<NoAllocScratch*>(p);
    return hipSuccess;
}

bool noalloc_scratch_view(int device, NoAllocView* out) {
    if (device < 0 || device >= kMaxDevices || !g_scratch[device].ptr) return false;
    ScratchHost& h = g_scratch[device];
    out->head = h.ptr->head;
    out->cost = h.ptr->cost;
    out->order[0] = h.ptr->order[0];
    out->order[1] = h.ptr->order[1];
    out->order_current = &h.order_current;
    out->order_sig = h.order_sig;
    return true;
}

uint32_t persistent_tile_count(const DevCam& cam, const DevRows& rows) {
    return ((cam.W + kPT - 1u) / kPT) * ((rows.local_rows + kPT - 1u) / kPT);
}

int persistent_blocks_per_cu(uint32_t stack_needed, bool count, bool big) {
    int n = 0;
    hipError_t e;
#define RT_OCC(N, C, B) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, path_kernel_persistent<N, C, B>, kBlock, 0)
    if (stack_needed <= 32) {
        e = count ? (big ? RT_OCC(32, true, true) : RT_OCC(32, true, false))
                  : (big ? RT_OCC(32, false, true) : RT_OCC(32, false, false));
    } else {
        e = count ? (big ? RT_OCC(64, true, true) : RT_OCC(64, true, false))
                  : (big ? RT_OCC(64, false, true) : RT_OCC(64, false, false));
    }
#undef RT_OCC
    return e == hipSuccess && n > 0 ? n : 1;
}

}  // namespace rt
