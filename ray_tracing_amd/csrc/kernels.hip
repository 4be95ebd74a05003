// HIP kernels for gfx950 (MI355X).  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <mutex>

#include "instance_cull.hpp"
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

// The shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime), read
// together.  A persistent wave lives as long as the launch, so one pair of stamps per workgroup
// gives the clock the launch ran at without touching the steps.
struct ClockStamp {
    unsigned long long cycles, ticks;
};
__device__ __forceinline__ ClockStamp clock_stamp() {
    return ClockStamp{__builtin_amdgcn_s_memtime(), __builtin_amdgcn_s_memrealtime()};
}
__device__ __forceinline__ void flush_clock(const DevFrame& fr, ClockStamp begin) {
    const ClockStamp end = clock_stamp();
    if (threadIdx.x == 0 && fr.stats) {
        atomicAdd(&fr.stats->clk_cycles, end.cycles - begin.cycles);
        atomicAdd(&fr.stats->clk_ticks, end.ticks - begin.ticks);
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
//    instance, a bounce or a frame.  Which kind a lane waits for is kept as four
//    lane masks in scalar registers, not as a number per lane: the vote counts
//    bits, and a step hands its own yes/no masks on as the new phases.
//  * The instance walk of a CONTROL step runs over the instance indices in
//    order, wave-uniformly, every lane joining in at its own index: instance
//    records are scalar loads, and an instance whose root box is missed costs a
//    lane no step of its own.
//  * The BOX and LEAF steps are straight-line code: every lane computes, lanes
//    that do not take part discard the results through selects (the one branch
//    is the exact form of a test whose approximate form could not tell, taken by
//    the lanes concerned).  A BOX step tests both children of the node at once,
//    and every yes/no of it is a lane mask in scalar registers.
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
// (the phase numbers of the pooled kernel's lanes; the persistent kernel keeps one lane mask per phase)
enum : uint32_t { ST_CONTROL = 0, ST_DESC = 1, ST_POP = 2, ST_LEAF = 3, ST_DEAD = 4 };
enum : uint32_t { CT_NEW_PIXEL = 0, CT_NEW_PATH = 1, CT_START_SEGMENT = 2, CT_NEXT_INSTANCE = 3, CT_SHADE = 4 };
// instance records read with a wave-uniform index: the constant address space lets them come
// through the scalar cache into scalar registers
typedef const __attribute__((address_space(4))) DevInstance* ConstInstance;

// A yes/no per lane held as a 64-bit mask in scalar registers: lanes_where() is what a vector
// comparison writes anyway, lane_of() hands a mask back as a per-lane condition (for a select, a
// store, a branch) without an instruction, and everything in between is scalar arithmetic.
typedef unsigned long long lmask;
__device__ __forceinline__ lmask lanes_where(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_of(lmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// x + 1 / x - 1 in the lanes of a mask, x elsewhere: the mask goes in as the carry of one add / subtract
// (a select and an add otherwise)
__device__ __forceinline__ uint32_t add_lanes(uint32_t x, lmask m) {
    uint32_t r;
    lmask carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t sub_lanes(uint32_t x, lmask m) {
    uint32_t r;
    lmask carry_out;
    asm("v_subb_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(m));
    return r;
}

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
#ifndef RT_EXP_WAVES
#define RT_EXP_WAVES 4
#endif
#ifndef RT_EXP_STACK
#define RT_EXP_STACK 32
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((COUNT || STACK_N > 32) ? 1 : RT_EXP_WAVES))) void path_kernel_persistent(DevScene sc, DevCam cam, DevRows rows, DevFrame fr,
                                                                 DevQueue q, uint32_t n_frames) {
    __shared__ uint32_t s_stack[(STACK_N == 32 ? RT_EXP_STACK : STACK_N) * kBlock];
    __shared__ uint32_t s_tmp[kBlock];  // 64 words per wave: who works for whom in a LEAF step
    __shared__ unsigned long long s_best[kBlock];  // per lane: what an owner keeps of a LEAF step's results

    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* const stack = &s_stack[threadIdx.x];
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const ClockStamp clk_begin = clock_stamp();

    // per lane, summed over the wave at the end.  Rays and paths of ONE lane in ONE launch are counted in
    // 32 bits (a lane traces a few hundred thousand rays a second: 2^32 of them would be a launch of
    // more than an hour); the sums are 64-bit.
    uint32_t n_rays = 0, n_paths = 0;
    unsigned long long n_box = 0, n_internal = 0, n_tri = 0;
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
    // `cur` of a lane in ST_DESC / ST_POP names a child slot of a node record: the record's byte offset in the
    // arena with the slot (0 / 1) in bit 0 -- records are 64 bytes apart, so the record is `cur & ~1` -- or, BIG,
    // (node index << 1) | slot.  slot_of_node: slot 0 of node a; slot_item: the item of the slot's record.
    auto slot_of_node = [&](uint32_t a) -> uint32_t { return BIG ? a << 1 : sc.node_off + a * 64u; };
    auto slot_item = [&](uint32_t c) -> uint32_t { return BIG ? sc.node_unit + (c >> 1) : c & ~1u; };
    auto item_ptr = [&](uint32_t item) -> const f4* {
        return reinterpret_cast<const f4*>(BIG ? sc.arena + ((unsigned long long)item << 6) : sc.arena + item);
    };
    auto tri_test_ptr = [&](uint32_t tri) -> const f4* {
        return reinterpret_cast<const f4*>(BIG ? sc.arena + ((unsigned long long)(sc.tri_test_unit + tri) << 6)
                                               : sc.arena + (sc.tri_test_off + tri * 64u));
    };
    const ConstInstance inst_base = (ConstInstance)(uintptr_t)sc.instances;
    // What kind of step each lane waits for, as four wave-uniform lane masks (a lane in none of them is out of
    // pixels): the vote counts their bits, a step hands its own yes/no masks on as the new phases, and no lane
    // carries a phase number that would have to be compared four times per iteration and selected four times
    // per BOX step.
    lmask pm_ctl = n_frames > 0 ? ~0ull : 0ull, pm_desc = 0ull, pm_pop = 0ull, pm_leaf = 0ull;
#define RT_PHASE_IN_CONTROL lane_of(pm_ctl)
#define RT_PHASE_AT_LEAF lane_of(pm_leaf)
#define RT_PHASE_LEAF_LANES pm_leaf
#define RT_PHASE_AFTER_BOX(m_internal_, m_leaf_) (pm_desc = (m_internal_), pm_pop = 0ull, pm_leaf |= (m_leaf_))
#define RT_PHASE_AFTER_LEAF(m_finished_) (pm_leaf &= ~(m_finished_))
#define RT_PHASE_AFTER_POP(m_pop_, m_inst_end_) (pm_pop |= (m_pop_), pm_ctl |= (m_inst_end_))
#define RT_PHASE_ENTER(enter_node_, enter_leaf_)                                                          \
    (pm_desc |= lanes_where(enter_node_), pm_leaf |= lanes_where(enter_leaf_), pm_ctl &= ~lanes_where((enter_node_) || (enter_leaf_)))
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
    lmask m_rcp_ok = 0ull;  // wave-uniform: the lanes whose ray is in the operand range of the cached-reciprocal division
    bool cull = false;
    f4 it0 = {0.0f, 0.0f, 0.0f, 0.0f}, it1 = it0, it2 = it0, it3 = it0;  // the pending item
    uint32_t cur = 0;       // ST_DESC / ST_POP: a child slot of a node record (see slot_of_node)
    uint32_t leaf_cur = 0;  // ST_LEAF: the next triangle of the leaf (a register of its own, so that a BOX step need
                            // not spare the lanes that stand at a leaf when it writes `cur`)
    uint32_t tri_last = 0, sp = 0;

    // Wave-uniform condition: lanes that have run out of pixels stay in the loop, inert, until
    // the whole wave is done, so that the cross-lane operations of the LEAF step always see all
    // 64 lanes.  (A launch for no frames has nothing to do and would never see a lane run out.)
    // Two loops, one in the other.  The inner one runs BOX and LEAF steps for as long as the vote asks for one
    // of them; when it asks for a CONTROL step the inner loop ends, the outer loop's body runs that step and
    // goes back in.  (One loop with the three kinds as branches of its body gives the same steps in the same
    // order; but then the way out -- the wave is done, found in the CONTROL step -- leaves the loop from
    // inside a branch, and the compiler pays for that with flags tested and registers copied on every
    // iteration of every kind: 2 % of the launch.)
    if (n_frames != 0u) for (;;) {
        lmask m_load = 0ull;  // wave-uniform: the lanes that have moved to a slot of another node record in this iteration
        unsigned long long t_begin = 0ull;
        for (;;) {
            m_load = 0ull;
            t_begin = COUNT ? __builtin_readcyclecounter() : 0ull;
            // wave-uniform choice of the one kind of step this iteration runs (a wave in which no lane waits for
            // any kind is done: every lane is out of pixels, which only a CONTROL step can bring about, so that
            // is where it is looked for)
            const lmask m_ctl = pm_ctl, m_leaf_wait = pm_leaf;
            const lmask m_desc_top = pm_desc, m_pop_top = pm_pop;  // (what the BOX step calls the lanes it works for)
            const lmask m_box_wait = m_desc_top | m_pop_top;
            const int w_ctl = __popcll(m_ctl) * q.w_ctl;
            const int w_box = __popcll(m_box_wait) * q.w_box;
            const int w_leaf = __popcll(m_leaf_wait) * q.w_leaf;
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

            if (pick == 0) break;
            // (each kind of step with its own copy of the pop that follows it; the BOX step, four iterations
            // in five, also with its own copy of the fetch and its own way back to the top, so that none of
            // its results crosses a point where the two kinds meet: 0.7 % of the launch)
            if (pick == 1) {
#define RT_STEP_ONLY_BOX
#include "traverse_steps.inc"
#undef RT_STEP_ONLY_BOX
                // the records of the lanes that moved on are fetched for their next step
                if (lane_of(m_load)) {
                    const f4* p = item_ptr(slot_item(cur));
                    it0 = p[0];
                    it1 = p[1];
                    it2 = p[2];
                    it3 = p[3];
                }
                if (COUNT) sched_cycles[pick] += __builtin_readcyclecounter() - t_begin;
                continue;
            }
            {
#define RT_STEP_ONLY_LEAF
#include "traverse_steps.inc"
#undef RT_STEP_ONLY_LEAF
            }
            if (lane_of(m_load)) {
                const f4* p = item_ptr(slot_item(cur));
                it0 = p[0];
                it1 = p[1];
                it2 = p[2];
                it3 = p[3];
            }
            if (COUNT) sched_cycles[pick] += __builtin_readcyclecounter() - t_begin;
        }
        const int pick = 0;  // the CONTROL step
        {
            // one queue per XCD group; every lane that needs a pixel takes part in one fetch
            const bool want = RT_PHASE_IN_CONTROL && ctl == CT_NEW_PIXEL;
            const unsigned long long wm = __ballot(want);
            bool out_of_pixels = false;
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
                        out_of_pixels = true;
                    }
                }
                wait_site(0);
            }
            pm_ctl &= ~lanes_where(out_of_pixels);
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
#include "instance_walk.inc"
                if (looking) ctl = CT_SHADE;  // every instance has been visited
                if (round == 1) break;  // whoever is through with its instances now is shaded in the next CONTROL step
                // (2)
                if (RT_PHASE_IN_CONTROL) {
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
                if (__ballot(RT_PHASE_IN_CONTROL && ctl == CT_NEXT_INSTANCE) == 0ull) break;
            }
        }
        if ((pm_ctl | pm_desc | pm_pop | pm_leaf) == 0ull) break;  // the wave is done
        // the records of the lanes that moved on are fetched for their next step
        if (lane_of(m_load)) {
            const f4* p = item_ptr(slot_item(cur));
            it0 = p[0];
            it1 = p[1];
            it2 = p[2];
            it3 = p[3];
        }
        if (COUNT) sched_cycles[pick] += __builtin_readcyclecounter() - t_begin;
    }
    flush_stats(fr, COUNT, lane, n_rays, n_paths, n_box, n_internal, n_tri);
    flush_clock(fr, clk_begin);
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
#undef RT_PHASE_IN_CONTROL
#undef RT_PHASE_AT_LEAF
#undef RT_PHASE_LEAF_LANES
#undef RT_PHASE_AFTER_BOX
#undef RT_PHASE_AFTER_LEAF
#undef RT_PHASE_AFTER_POP
#undef RT_PHASE_ENTER

// ---------------------------------------------------------------------------
// Pooled persistent kernel.
//
// The same steps on the same per-ray arithmetic as path_kernel_persistent, with one change of
// ownership: a lane no longer keeps one pixel from its first frame to its last.  Every wave owns
// a pool of 192 ray slots (three per lane), each a 128-byte record in global memory that holds a
// pixel's bookkeeping (position, accumulator, frames left) and its current path (seed, segment
// number, throughput, ray, and the hit record of the segment just traced).  Lanes only
// TRAVERSE: a lane takes a slot whose next segment is ready, loads the ray, walks the instances
// and their trees (CONTROL, BOX and LEAF steps as before), writes the hit record back and takes
// the next ready slot.  Everything else a path needs -- shading, accumulation, the next frame's
// seed and camera ray, the next pixel from the work queues -- happens in SHADE steps, which the
// wave runs when 64 slots are waiting for one: every lane then shades a slot, whichever lane
// traced it.  Two things follow.  Shading, the longest piece of straight code in the kernel,
// runs with all 64 lanes instead of the 24 a CONTROL step used to gather; and a lane that
// finishes a segment does not wait for its neighbours to finish theirs, it goes on with
// another ray, so the BOX and LEAF steps find more lanes that want them.
//
// The pool is private to the wave (no other wave reads or writes its slots or its slot states),
// so all bookkeeping is wave-synchronous: ballots, prefix counts and a few LDS words, no atomics
// and no waiting on other waves.  Which lane traces which pixel changes no pixel's value.
// ---------------------------------------------------------------------------
constexpr uint32_t kSlotsPerLane = 3u;
constexpr uint32_t kWaveSlots = 64u * kSlotsPerLane;
enum : uint32_t { SL_NEW = 0, SL_READY = 1, SL_TRAVERSING = 2, SL_PENDING = 3, SL_RETIRED = 4 };
enum : uint32_t { CT_IDLE = 5, CT_DONE = 6, CT_READY = 7, CT_RETIRE = 8 };

// line 0: ray origin, seed | 1: ray direction, segment number | 2: throughput, pixel (x | y << 16)
// | 3: accumulator | 4: hit record of the traced segment (t, triangle, instance, box tests)
// | 5: frames_without_move, frame, tile, frames left | 6: internal visits, triangle tests | 7: --
struct alignas(128) RayRecord {
    f4 line[8];
};
static_assert(sizeof(RayRecord) == 128, "one record is 128 bytes");

template <int STACK_N, bool COUNT, bool BIG>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu((COUNT || STACK_N > 32) ? 1 : 4))) void path_kernel_pool(
    DevScene sc, DevCam cam, DevRows rows, DevFrame fr, DevQueue q, uint32_t n_frames, RayRecord* records) {
    __shared__ uint32_t s_stack[STACK_N * kBlock];
    __shared__ uint32_t s_tmp[kBlock];  // 64 words per wave: who works for whom in a LEAF step; slots dealt to lanes
    __shared__ unsigned long long s_best[kBlock];  // per lane: what an owner keeps of a LEAF step's results
    __shared__ uint32_t s_state[kBlock * kSlotsPerLane];  // per wave: the states of its 192 slots

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* const stack = &s_stack[threadIdx.x];
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const ClockStamp clk_begin = clock_stamp();
    uint32_t* const state = &s_state[wave * kWaveSlots];
    uint32_t* const deal = &s_tmp[threadIdx.x & ~63u];
    RayRecord* const pool = records + (size_t)(blockIdx.x * (kBlock / 64u) + wave) * kWaveSlots;

    unsigned long long n_rays = 0, n_box = 0, n_internal = 0, n_tri = 0, n_paths = 0;
    unsigned long long sched_iters[4] = {0, 0, 0, 0}, sched_lanes[4] = {0, 0, 0, 0}, sched_cycles[4] = {0, 0, 0, 0},
                       sched_wait[4] = {0, 0, 0, 0};  // wave-uniform
    unsigned long long site_wait[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-uniform, profile only
    auto wait_site = [&](int site) {
        if (COUNT) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            site_wait[site] += __builtin_readcyclecounter() - t0;
        }
    };
    // `cur` of a lane in ST_DESC / ST_POP names a child slot of a node record: the record's byte offset in the
    // arena with the slot (0 / 1) in bit 0 -- records are 64 bytes apart, so the record is `cur & ~1` -- or, BIG,
    // (node index << 1) | slot.  slot_of_node: slot 0 of node a; slot_item: the item of the slot's record.
    auto slot_of_node = [&](uint32_t a) -> uint32_t { return BIG ? a << 1 : sc.node_off + a * 64u; };
    auto slot_item = [&](uint32_t c) -> uint32_t { return BIG ? sc.node_unit + (c >> 1) : c & ~1u; };
    auto item_ptr = [&](uint32_t item) -> const f4* {
        return reinterpret_cast<const f4*>(BIG ? sc.arena + ((unsigned long long)item << 6) : sc.arena + item);
    };
    auto tri_test_ptr = [&](uint32_t tri) -> const f4* {
        return reinterpret_cast<const f4*>(BIG ? sc.arena + ((unsigned long long)(sc.tri_test_unit + tri) << 6)
                                               : sc.arena + (sc.tri_test_off + tri * 64u));
    };
    const ConstInstance inst_base = (ConstInstance)(uintptr_t)sc.instances;

    // the wave's slots: all waiting for a pixel
    const uint32_t slots_in_use = q.pool_slots >= 64u && q.pool_slots <= kWaveSlots ? q.pool_slots : kWaveSlots;
    state[lane] = SL_NEW;
    state[lane + 64u] = lane + 64u < slots_in_use ? (uint32_t)SL_NEW : (uint32_t)SL_RETIRED;
    state[lane + 128u] = lane + 128u < slots_in_use ? (uint32_t)SL_NEW : (uint32_t)SL_RETIRED;
    asm volatile("" ::: "memory");
    // wave-uniform census of the slot states (kept in step with every change below)
    uint32_t n_new = n_frames > 0 ? slots_in_use : 0u, n_ready = 0, n_pending = 0, n_live = n_new;
    uint32_t queues_tried = 0;  // wave-uniform: how many queues this wave has found empty
    uint32_t stalls = 0;        // wave-uniform: iterations that found nothing to do (cannot happen; bounds the loop)

    // the lane as a traverser
    uint32_t phase = ST_CONTROL, ctl = CT_IDLE;
    // (this kernel keeps a phase number per lane; the shared step texts reach it through these)
#define RT_PHASE_IN_CONTROL (phase == ST_CONTROL)
#define RT_PHASE_AT_LEAF (phase == ST_LEAF)
#define RT_PHASE_LEAF_LANES lanes_where(phase == ST_LEAF)
#define RT_PHASE_AFTER_BOX(m_internal_, m_leaf_) \
    (phase = lane_of(m_internal_) ? (uint32_t)ST_DESC : lane_of(m_leaf_) ? (uint32_t)ST_LEAF : phase)
#define RT_PHASE_AFTER_LEAF(m_finished_) ((void)0)
#define RT_PHASE_AFTER_POP(m_pop_, m_inst_end_) \
    (phase = lane_of(m_pop_) ? (uint32_t)ST_POP : lane_of(m_inst_end_) ? (uint32_t)ST_CONTROL : phase)
#define RT_PHASE_ENTER(enter_node_, enter_leaf_) \
    (phase = (enter_node_) ? (uint32_t)ST_DESC : (enter_leaf_) ? (uint32_t)ST_LEAF : phase)
    uint32_t slot = 0, inst = 0;
    Ray ray;
    ray.pos = v3(0.0f, 0.0f, 0.0f);
    ray.dir = v3(0.0f, 0.0f, 1.0f);
    HitRecord rec;  // rec.t doubles as the traversal's clamp: trace.rs:89 hands it to hit_bvh as t_max
    record_new(rec);
    f2 dpx = {0.0f, 0.0f}, dpy = {0.0f, 0.0f}, dpz = {0.0f, 1.0f};
    float rcx = 1.0f, rcy = 1.0f, rcz = 1.0f;
    lmask m_rcp_ok = 0ull;  // wave-uniform: the lanes whose ray is in the operand range of the cached-reciprocal division
    bool cull = false;
    f4 it0 = {0.0f, 0.0f, 0.0f, 0.0f}, it1 = it0, it2 = it0, it3 = it0;  // the pending item
    uint32_t cur = 0;       // ST_DESC / ST_POP: a child slot of a node record (see slot_of_node)
    uint32_t leaf_cur = 0;  // ST_LEAF: the next triangle of the leaf (a register of its own, so that a BOX step need
                            // not spare the lanes that stand at a leaf when it writes `cur`)
    uint32_t tri_last = 0, sp = 0;

    while (n_live != 0u) {
        lmask m_load = 0ull;  // wave-uniform: the lanes that have moved to a slot of another node record in this iteration
        const unsigned long long t_begin = COUNT ? __builtin_readcyclecounter() : 0ull;
        // wave-uniform choice of the one kind of step this iteration runs
        const uint32_t n_ctl = (uint32_t)__popcll(__ballot(phase == ST_CONTROL && ctl != CT_IDLE));
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(phase == ST_CONTROL && ctl == CT_IDLE));
        const uint32_t n_claim = n_idle < n_ready ? n_idle : n_ready;
        const int w_ctl = (int)(n_ctl + n_claim) * q.w_ctl;
        const lmask m_desc_top = lanes_where(phase == ST_DESC), m_pop_top = lanes_where(phase == ST_POP);
        const int w_box = __popcll(m_desc_top | m_pop_top) * q.w_box;
        const int w_leaf = __popcll(__ballot(phase == ST_LEAF)) * q.w_leaf;
        const uint32_t n_shadeable = n_pending + n_new;
        int pick = 1, w_best = w_box;
        if (w_leaf > w_best) pick = 2, w_best = w_leaf;
        if (w_ctl > w_best) pick = 0, w_best = w_ctl;
        // a full batch is shaded at once; a partial one when lanes are out of rays to trace
        if (n_shadeable >= 64u || (n_shadeable != 0u && (w_best == 0 || (n_ready == 0u && n_idle >= 16u)))) pick = 3;
        else if (w_best == 0) {
            if (++stalls > 1000u) {
                // live slots and nothing to schedule: must not happen.  The wave gives up rather than
                // spin for ever, and says so: the host turns a non-zero count into an error.
                if (lane == 0 && fr.stats) atomicAdd(&fr.stats->pool_stalled, 1ull);
                break;
            }
            continue;
        }
        if (COUNT) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sched_wait[pick] += __builtin_readcyclecounter() - t0;
            sched_iters[pick] += 1;
            sched_lanes[pick] += (unsigned)(pick == 0 ? w_ctl / q.w_ctl : pick == 1 ? w_box / q.w_box
                                            : pick == 2 ? w_leaf / q.w_leaf : (n_shadeable < 64u ? n_shadeable : 64u));
        }

        if (pick == 3) {
            // ---- SHADE: up to 64 slots that wait for a pixel or for the rest of trace_ray ----
            asm volatile("" ::: "memory");
            const uint32_t st0 = state[lane], st1 = state[lane + 64u], st2 = state[lane + 128u];
            const bool f0 = st0 == SL_PENDING || st0 == SL_NEW, f1 = st1 == SL_PENDING || st1 == SL_NEW,
                       f2_ = st2 == SL_PENDING || st2 == SL_NEW;
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2_);
            const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1);
            const uint32_t r0 = (uint32_t)__popcll(m0 & lanes_below), r1 = c0 + (uint32_t)__popcll(m1 & lanes_below),
                           r2 = c0 + c1 + (uint32_t)__popcll(m2 & lanes_below);
            const uint32_t n_have = c0 + c1 + (uint32_t)__popcll(m2), n_take = n_have < 64u ? n_have : 64u;
            if (f0 && r0 < 64u) deal[r0] = lane | (st0 == SL_NEW ? 0x80000000u : 0u);
            if (f1 && r1 < 64u) deal[r1] = (lane + 64u) | (st1 == SL_NEW ? 0x80000000u : 0u);
            if (f2_ && r2 < 64u) deal[r2] = (lane + 128u) | (st2 == SL_NEW ? 0x80000000u : 0u);
            asm volatile("" ::: "memory");
            const bool active = lane < n_take;
            const uint32_t dealt = deal[lane];
            const uint32_t my = dealt & 0xFFFFu;
            const bool fresh = (dealt >> 31) != 0u;
            f4* const r = pool[active ? my : 0u].line;
            uint32_t sctl = active ? (fresh ? (uint32_t)CT_NEW_PIXEL : (uint32_t)CT_SHADE) : (uint32_t)CT_IDLE;
            n_pending -= (uint32_t)__popcll(__ballot(active && !fresh));
            n_new -= (uint32_t)__popcll(__ballot(active && fresh));
            // the slot's pixel and path
            uint32_t x = 0, y = 0, tile = 0, frame = 0, frames_left = 0, seed = 0, seg = 0;
            float fwm = 0.0f;
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            V3 color = v3(1.0f, 1.0f, 1.0f);
            Ray pray;
            pray.pos = v3(0.0f, 0.0f, 0.0f);
            pray.dir = v3(0.0f, 0.0f, 1.0f);
            HitRecord prec;
            record_new(prec);
            if (sctl == CT_SHADE) {
                const f4 l0 = r[0], l1 = r[1], l2 = r[2], l3 = r[3], l4 = r[4], l5 = r[5];
                pray.pos = v3(l0.x, l0.y, l0.z);
                seed = __float_as_uint(l0.w);
                pray.dir = v3(l1.x, l1.y, l1.z);
                seg = __float_as_uint(l1.w);
                color = v3(l2.x, l2.y, l2.z);
                x = __float_as_uint(l2.w) & 0xFFFFu;
                y = __float_as_uint(l2.w) >> 16;
                acc = make_float4(l3.x, l3.y, l3.z, l3.w);
                prec.t = l4.x;
                prec.triangle_id = __float_as_uint(l4.y);
                prec.instance_id = __float_as_uint(l4.z);
                prec.box_tests = __float_as_uint(l4.w);
                fwm = l5.x;
                frame = __float_as_uint(l5.y);
                tile = __float_as_uint(l5.z);
                frames_left = __float_as_uint(l5.w);
                if (COUNT) {
                    const f4 l6 = r[6];
                    prec.internal_visits = __float_as_uint(l6.x);
                    prec.triangle_tests = __float_as_uint(l6.y);
                }
                // the rest of trace_ray, trace.rs:97-172
                const uint32_t idx = x + y * cam.W;
                if (seg == 0 && fr.hit_triangle) {
                    fr.hit_triangle[idx] = prec.triangle_id;
                    fr.hit_instance[idx] = prec.instance_id;
                    fr.hit_t_bits[idx] = __float_as_uint(prec.t);
                }
                if (COUNT) {
                    if (seg == 0 && fr.box_tests) {
                        fr.box_tests[idx] = prec.box_tests;
                        fr.triangle_tests[idx] = prec.triangle_tests;
                    }
                    n_box += prec.box_tests;
                    n_internal += prec.internal_visits;
                    n_tri += prec.triangle_tests;
                }
                int st = shade(sc, cam, prec, pray, seed, color);
                seg += 1;
                if (st == STATE_RAY && seg < cam.depth) {
                    sctl = CT_START_SEGMENT;
                } else {
                    if (st == STATE_RAY) color = v3(0.0f, 0.0f, 0.0f);  // never finished bouncing, trace.rs:204-205
                    // shader/src/lib.rs:71-98, then the event loop's bookkeeping, main.rs:147-148
                    acc = accumulate_only(fwm, acc, color);
                    n_paths += 1;
                    frame += cam.frame_stride;
                    fwm += 1.0f;
                    frames_left -= 1u;
                    sctl = CT_NEW_PATH;
                    if (frames_left == 0u) {
                        reinterpret_cast<float4*>(fr.acc)[idx] = acc;
                        if (q.cost) atomicAdd(&q.cost[tile], seg);  // length of the pixel's last path
                        sctl = CT_NEW_PIXEL;
                    }
                }
            }
            {
                // one queue per XCD group; every lane that needs a pixel takes part in one fetch
                const bool want = sctl == CT_NEW_PIXEL;
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
                                sctl = CT_NEW_PATH;
                            }
                        } else if (queues_tried >= 8u) {
                            sctl = CT_RETIRE;
                        }
                    }
                }
            }
            if (sctl == CT_NEW_PATH) {
                // shader/src/lib.rs:57-69 and Ray::get_color, trace.rs:175-191
                seed = cam.seed_mode ? get_seed(frame, x, y, 0.0f, 0.0f, 0.0f) : get_seed(frame, x, y, acc.x, acc.y, acc.z);
                color = v3(1.0f, 1.0f, 1.0f);
                float jx = rand_float(seed, 0.0f, 1.0f);
                float jy = rand_float(seed, 0.0f, 1.0f);
                pray = camera_ray(cam, (float)x + jx, (float)y + jy);
                pray.dir = normalize(pray.dir);
                seg = 0;
                sctl = CT_START_SEGMENT;
            }
            if (sctl == CT_START_SEGMENT) {
                // head of Ray::trace_ray, trace.rs:73-74; the traversing lane does the rest
                pray.dir = normalize(pray.dir);
                n_rays += 1;
                r[0] = f4{pray.pos.x, pray.pos.y, pray.pos.z, __uint_as_float(seed)};
                r[1] = f4{pray.dir.x, pray.dir.y, pray.dir.z, __uint_as_float(seg)};
                r[2] = f4{color.x, color.y, color.z, __uint_as_float(x | (y << 16))};
                r[3] = f4{acc.x, acc.y, acc.z, acc.w};
                r[5] = f4{fwm, __uint_as_float(frame), __uint_as_float(tile), __uint_as_float(frames_left)};
                sctl = CT_READY;
            }
            if (active) state[my] = sctl == CT_READY ? (uint32_t)SL_READY : sctl == CT_RETIRE ? (uint32_t)SL_RETIRED : (uint32_t)SL_NEW;
            asm volatile("" ::: "memory");
            n_ready += (uint32_t)__popcll(__ballot(sctl == CT_READY));
            n_new += (uint32_t)__popcll(__ballot(sctl == CT_NEW_PIXEL));
            n_live -= (uint32_t)__popcll(__ballot(sctl == CT_RETIRE));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the records are read again by this wave only, after this
        } else if (pick == 0) {
            // Two rounds of: (1) the instance loop of trace_ray (trace.rs:76-94) for every lane that
            // stands between two instances, (2) hand the finished segments over, take ready ones.
            for (int round = 0; round < 2; ++round) {
#include "instance_walk.inc"
                if (looking) ctl = CT_DONE;  // every instance has been visited
                if (round == 1) break;
                // (2a) segments whose instances are exhausted: the hit record goes to the slot
                const bool done = phase == ST_CONTROL && ctl == CT_DONE;
                const unsigned long long dm = __ballot(done);
                if (dm != 0ull) {
                    if (done) {
                        f4* const r = pool[slot].line;
                        r[4] = f4{rec.t, __uint_as_float(rec.triangle_id), __uint_as_float(rec.instance_id),
                                  __uint_as_float(rec.box_tests)};
                        if (COUNT) r[6] = f4{__uint_as_float(rec.internal_visits), __uint_as_float(rec.triangle_tests), 0.0f, 0.0f};
                        state[slot] = SL_PENDING;
                        ctl = CT_IDLE;
                    }
                    n_pending += (uint32_t)__popcll(dm);
                    asm volatile("" ::: "memory");
                }
                // (2b) lanes without a ray take ready slots, in slot order
                const bool idle = phase == ST_CONTROL && ctl == CT_IDLE;
                const unsigned long long im = __ballot(idle);
                const uint32_t n_free = (uint32_t)__popcll(im), n_take = n_free < n_ready ? n_free : n_ready;
                if (n_take != 0u) {
                    asm volatile("" ::: "memory");
                    const uint32_t st0 = state[lane], st1 = state[lane + 64u], st2 = state[lane + 128u];
                    const bool f0 = st0 == SL_READY, f1 = st1 == SL_READY, f2_ = st2 == SL_READY;
                    const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2_);
                    const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1);
                    const uint32_t r0 = (uint32_t)__popcll(m0 & lanes_below), r1 = c0 + (uint32_t)__popcll(m1 & lanes_below),
                                   r2 = c0 + c1 + (uint32_t)__popcll(m2 & lanes_below);
                    if (f0 && r0 < n_take) deal[r0] = lane;
                    if (f1 && r1 < n_take) deal[r1] = lane + 64u;
                    if (f2_ && r2 < n_take) deal[r2] = lane + 128u;
                    asm volatile("" ::: "memory");
                    const uint32_t k = (uint32_t)__popcll(im & lanes_below);
                    if (idle && k < n_take) {
                        slot = deal[k];
                        state[slot] = SL_TRAVERSING;
                        const f4* const r = pool[slot].line;
                        const f4 l0 = r[0], l1 = r[1];
                        ray.pos = v3(l0.x, l0.y, l0.z);
                        ray.dir = v3(l1.x, l1.y, l1.z);
                        record_new(rec);
                        inst = 0;
                        ctl = CT_NEXT_INSTANCE;
                    }
                    asm volatile("" ::: "memory");
                    n_ready -= n_take;
                }
                if (__ballot(phase == ST_CONTROL && ctl == CT_NEXT_INSTANCE) == 0ull) break;
            }
        } else {
#include "traverse_steps.inc"
        }
        if (lane_of(m_load)) {
            const f4* p = item_ptr(slot_item(cur));
            it0 = p[0];
            it1 = p[1];
            it2 = p[2];
            it3 = p[3];
        }
        if (COUNT) sched_cycles[pick] += __builtin_readcyclecounter() - t_begin;
    }
    flush_stats(fr, COUNT, lane, n_rays, n_paths, n_box, n_internal, n_tri);
    flush_clock(fr, clk_begin);
    if (COUNT && lane == 0 && fr.stats) {
        for (int k = 0; k < 3; ++k) {
            atomicAdd(&fr.stats->iters[k], sched_iters[k]);
            atomicAdd(&fr.stats->lanes[k], sched_lanes[k]);
            atomicAdd(&fr.stats->cycles[k], sched_cycles[k]);
            atomicAdd(&fr.stats->wait_cycles[k], sched_wait[k]);
        }
        atomicAdd(&fr.stats->shade_iters, sched_iters[3]);
        atomicAdd(&fr.stats->shade_lanes, sched_lanes[3]);
        atomicAdd(&fr.stats->shade_cycles, sched_cycles[3]);
        for (int k = 0; k < 8; ++k) atomicAdd(&fr.stats->site_wait[k], site_wait[k]);
    }
}
#undef RT_PHASE_IN_CONTROL
#undef RT_PHASE_AT_LEAF
#undef RT_PHASE_LEAF_LANES
#undef RT_PHASE_AFTER_BOX
#undef RT_PHASE_AFTER_LEAF
#undef RT_PHASE_AFTER_POP
#undef RT_PHASE_ENTER

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

// The presented image as the 8-bit RGBA a display takes (what the reference's blit of binding 8
// into the swapchain image produces, vulkan.rs:1045-1101): every channel clamped to [0, 1], times
// 255, rounded to nearest; NaN -> 0.  The same rule as the PNG writer's, so the bytes agree.
__global__ __launch_bounds__(kBlock) void present_bytes_kernel(const float4* present, uint32_t n, uint32_t* rgba8) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = present[i];
    auto byte_of = [](float v) -> uint32_t {
        if (!(v > 0.0f)) return 0u;
        if (v >= 1.0f) return 255u;
        return (uint32_t)__float2int_rn(v * 255.0f);
    };
    rgba8[i] = byte_of(p.x) | (byte_of(p.y) << 8) | (byte_of(p.z) << 16) | (byte_of(p.w) << 24);
}

// DevTri::wn of every triangle whose object is shown by exactly one instance: the world-space shading
// normal, by the very function shade() would call at every hit of that triangle.  Run after a scene upload
// and after the instances changed.  owner[i]: the object whose tree holds triangle i; object_instance[o]:
// the one instance of object o, or 0xFFFFFFFF.
__global__ __launch_bounds__(kBlock) void world_normals_kernel(DevTri* tris, uint32_t n, const uint32_t* owner,
                                                               const uint32_t* object_instance,
                                                               const DevInstance* instances) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = owner[i];
    if (o == 0xFFFFFFFFu) return;
    const uint32_t k = object_instance[o];
    if (k == 0xFFFFFFFFu) return;
    const float4* tp = reinterpret_cast<const float4*>(&tris[i]);
    const V3 nrm = world_normal(instances[k].fwd, tp[0], tp[1], tp[2]);
    tris[i].wn[0] = nrm.x;
    tris[i].wn[1] = nrm.y;
    tris[i].wn[2] = nrm.z;
}

hipError_t launch_world_normals(DevTri* tris, uint32_t n_tris, const uint32_t* owner, const uint32_t* object_instance,
                                const DevInstance* instances, hipStream_t stream) {
    if (n_tris == 0) return hipSuccess;
    hipLaunchKernelGGL(world_normals_kernel, dim3((n_tris + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, tris, n_tris,
                       owner, object_instance, instances);
    return hipGetLastError();
}

hipError_t launch_present_bytes(const float* present, uint32_t n_pixels, uint32_t* rgba8, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(present_bytes_kernel, dim3((n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       reinterpret_cast<const float4*>(present), n_pixels, rgba8);
    return hipGetLastError();
}

// Next launch's tile order from this launch's costs: a counting sort into 32 cost classes,
// dearest class first, one workgroup per queue.  Within a class the order is whatever the atomics
// make it: the order of pixels changes no pixel's value.  Two ways of sharing the tiles out over
// the 8 queues:
//   banded: every queue keeps its contiguous band of the canvas -- an XCD's L2 keeps seeing
//     neighbouring rays -- and sorts the tiles inside it;
//   dealt: queue i takes every eighth tile of the whole canvas (i, i + 8, ...), so every queue is
//     a sample of all of it, runs from its dearest tiles to its cheapest and the launch ends on
//     short pixels everywhere.  That shortens the tail, which is what a launch of one or two
//     frames mostly consists of; over many frames the bands' locality is worth more.
constexpr int kOrderBlock = 1024;  // the sort is a chain of dependent passes over the costs: many threads, short chains
__global__ __launch_bounds__(kOrderBlock) void order_tiles_kernel(DevQueue q, uint32_t* order_out) {
    __shared__ uint32_t s_count[32], s_start[32];
    // this workgroup's queue takes a contiguous band of the canvas, or every eighth tile of all of it
    const bool dealt = q.deal != 0u;
    const uint32_t base = q.first_item[blockIdx.x] >> kPTShift;  // where the queue's part of the order begins
    const uint32_t t0 = dealt ? blockIdx.x : base;
    const uint32_t t1 = (dealt ? q.first_item[8] : q.first_item[blockIdx.x + 1]) >> kPTShift;
    const uint32_t step = dealt ? 8u : 1u;
    if (threadIdx.x < 32) s_count[threadIdx.x] = 0;
    __syncthreads();
    // a full tile's cost is 64 pixels x 1..10+ segments; 24 per class keeps the classes fine enough
    auto cls = [](uint32_t c) { return c * 64u / (kPT * kPT * 24u) < 31u ? c * 64u / (kPT * kPT * 24u) : 31u; };
    for (uint32_t t = t0 + threadIdx.x * step; t < t1; t += kOrderBlock * step) atomicAdd(&s_count[cls(q.cost[t])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        for (int c = 31; c >= 0; --c) {
            s_start[c] = at;
            at += s_count[c];
        }
    }
    __syncthreads();
    for (uint32_t t = t0 + threadIdx.x * step; t < t1; t += kOrderBlock * step) {
        const uint32_t r = atomicAdd(&s_start[cls(q.cost[t])], 1u);  // rank in the sorted order
        order_out[base + r] = t;
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
                                    bool count, bool big, uint32_t n_blocks, void* pool_records, hipStream_t stream) {
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
#define RT_LAUNCH(N, C, B)                                                                                             \
    do {                                                                                                               \
        if (pool_records)                                                                                              \
            hipLaunchKernelGGL((path_kernel_pool<N, C, B>), grid, block, 0, stream, sc, cam, rows, fr, q, n_frames,    \
                               static_cast<RayRecord*>(pool_records));                                                 \
        else                                                                                                           \
            hipLaunchKernelGGL((path_kernel_persistent<N, C, B>), grid, block, 0, stream, sc, cam, rows, fr, q,        \
                               n_frames);                                                                              \
    } while (0)
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
        hipLaunchKernelGGL(order_tiles_kernel, dim3(8), dim3(kOrderBlock), 0, stream, q, order_out);
        if ((e2 = hipGetLastError()) != hipSuccess) return e2;
    }
    if (!fr.present) return hipSuccess;
    return launch_present(cam, rows, fr, n_frames, stream);
}

hipError_t launch_present(const DevCam& cam, const DevRows& rows, const DevFrame& fr, uint32_t n_frames, hipStream_t stream) {
    const uint32_t n_pixels = cam.W * rows.local_rows;
    if (n_pixels == 0 || n_frames == 0) return hipSuccess;
    // frames_without_move of the last frame, advanced the way the kernel advances it
    float fwm_last = cam.frames_without_move;
    for (uint32_t i = 1; i < n_frames; ++i) {
        const float next = fwm_last + 1.0f;
        if (next == fwm_last) break;  // the increments have stopped registering
        fwm_last = next;
    }
    hipLaunchKernelGGL(present_kernel, dim3((n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, cam, rows, fr, fwm_last);
    return hipGetLastError();
}

// out[i] = ((src[0][i] + src[1][i]) + src[2][i]) + ...: the per-GPU accumulators of a render split by
// sample index, added in rank order (so that the sum is one definite value, the one a single process
// adding the shares in that order gets).  `out` may be src[0].
__global__ __launch_bounds__(kBlock) void sum_accumulators_kernel(float4* out, AccumulatorList list, uint32_t n) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 a = reinterpret_cast<const float4*>(list.src[0])[i];
    for (uint32_t k = 1; k < list.n; ++k) {
        const float4 b = reinterpret_cast<const float4*>(list.src[k])[i];
        a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    out[i] = a;
}

hipError_t launch_sum_accumulators(float* out, const AccumulatorList& list, uint32_t n_pixels, hipStream_t stream) {
    if (n_pixels == 0 || list.n == 0) return hipSuccess;
    hipLaunchKernelGGL(sum_accumulators_kernel, dim3((n_pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       reinterpret_cast<float4*>(out), list, n_pixels);
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
std::mutex g_scratch_mutex;  // contexts may be created on one GPU from several threads

}  // namespace

hipError_t noalloc_scratch_init(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch[device].ptr) return hipSuccess;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // `device` is current
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch[device].ptr = static_cast<NoAllocScratch*>(p);
    return hipSuccess;
}

bool noalloc_scratch_view(int device, NoAllocView* out) {
    if (device < 0 || device >= kMaxDevices) return false;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch[device].ptr) return false;
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

size_t pool_record_bytes(uint32_t n_blocks) { return (size_t)n_blocks * (kBlock / 64u) * kWaveSlots * sizeof(RayRecord); }

int persistent_blocks_per_cu(uint32_t stack_needed, bool count, bool big, bool pool) {
    int n = 0;
    hipError_t e;
#define RT_OCC(N, C, B)                                                                                    \
    (pool ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, path_kernel_pool<N, C, B>, kBlock, 0)         \
          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, path_kernel_persistent<N, C, B>, kBlock, 0))
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
