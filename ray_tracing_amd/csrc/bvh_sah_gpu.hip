// A second GPU builder: top-down binned surface-area splits, level by level.
//
// The reference's builder (program/src/modules/bvh.rs:30-120) sorts a node's triangles by centroid
// along each axis, prices n/4 split positions per axis with  area(L) * |L| + area(R) * |R|  and
// recurses until a side holds no more than 16 triangles.  host_scene.cpp reproduces that tree
// exactly; the linear BVH of bvh_gpu.hip is built a hundred times sooner but follows a space-filling
// curve, not surface areas, and traces slower.  This builder keeps the reference's cost function and
// its "split while the node is too big" rule and evaluates the cost where a GPU can do it for every
// node of a level at once: at 15 planes per axis between 16 equal bins of the node's centroid
// bounds (Wald, "On fast Construction of SAH-based Bounding Volume Hierarchies", 2007).  Same node
// format (shared::Bvh, 48 bytes, inclusive global triangle ranges), a different tree again.
//
// One level = five passes over the triangles, all nodes of the level in parallel:
//   1. bin:     every triangle of an active node adds its box, its centroid and a count to one bin
//               per axis (integer atomics on the float bit patterns: exact, order-independent)
//   2. split:   one thread per active node sweeps the 3 x 15 planes and picks the cheapest; the
//               children's boxes and centroid bounds are the unions of the bins on either side
//   3. flag + prefix sum (hipCUB) + scatter: a stable partition of every node's range at once --
//               the global prefix sum of "goes left" gives each triangle its rank inside its node
//   4. children get their node numbers (two per active node, in the order of the level's list) and
//               the ones still too big their place in the next level's list (a second prefix sum),
//               so the same input always gives the same array of nodes.
// Once a node is down to kSmallMax triangles it leaves the level-by-level passes: its whole subtree is
// finished by ONE workgroup (finish_small_kernel) that holds the triangle order in LDS, gives every
// node of the subtree to a wave -- the same bins, the same cost, the same rules, decided from the same
// numbers, so the subtree is the one the passes would have made -- and writes its nodes into a scratch
// array; a prefix sum over the subtrees' node counts and a copy then number them behind the nodes of the
// passes, subtree after subtree, each in level order.  The passes spend their time on atomics into
// global bins (39 per triangle and level); in LDS the same atomics cost a tenth, and a million
// triangles need ten levels of passes instead of twenty-two.
// A node whose centroids all fall into one bin on every axis (coincident centroids), or whose
// cheapest split would leave a side too big for the levels that remain, is cut in the middle of
// its range instead; its children's boxes are then gathered by the scatter pass.  Boxes without
// thickness along an axis, which the reference's slab test never accepts, are widened a little at
// the end (bvh_pad.hpp).
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstring>
#include <utility>

#include "../../include/rt_hip.h"
#include "bvh_pad.hpp"
#include "host_scene.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kBins = 16;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kAxisByPosition = 3u;

struct Bounds {
    float lo[3], hi[3];
};

struct TriInfo {
    Bounds box;
    float c[3];  // centroid
};

struct Bin {
    uint32_t count;
    Bounds box;   // of the triangles
    Bounds cent;  // of their centroids
};

// A node of the level being split.
struct Active {
    uint32_t node;          // its place in the output array
    uint32_t start, count;  // its triangles: positions start .. start + count - 1 of the current order
    Bounds cent;            // bounds of their centroids
    // the decision of the split pass
    uint32_t axis;       // 0..2, or kAxisByPosition
    uint32_t split_bin;  // bins 0..split_bin go left
    uint32_t n_left;
    uint32_t child_slot[2];  // the children's places in the next level's list, kNone for a leaf
};

// What the split pass knows of a child before it has a place in the next level's list.
struct Child {
    uint32_t start, count;
    uint32_t active;  // 0: a leaf; 1: too big to be a leaf, split by the next level's passes; 2: too big to be a
                      // leaf and small enough for finish_small_kernel
    Bounds box, cent;
};

// Triangles of a subtree one workgroup finishes in LDS (0 as the argument of the builder: never, every
// level goes through the passes).
constexpr uint32_t kSmallMax = 2048;

// A node handed to finish_small_kernel.
struct SmallRoot {
    uint32_t node;          // its record in the output array, written by the passes as a leaf
    uint32_t start, count;  // its triangles
    uint32_t level;         // its level, the root of the tree being 0
};

__device__ __host__ inline void bounds_empty(Bounds& b) {
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = __builtin_huge_valf();
        b.hi[k] = -__builtin_huge_valf();
    }
}
__device__ inline void bounds_merge(Bounds& a, const Bounds& b) {
    for (int k = 0; k < 3; ++k) {
        a.lo[k] = b.lo[k] < a.lo[k] ? b.lo[k] : a.lo[k];
        a.hi[k] = b.hi[k] > a.hi[k] ? b.hi[k] : a.hi[k];
    }
}
// bvh.rs:122-127
__device__ inline float bounds_area(const Bounds& b) {
    const float x = fabsf(b.hi[0] - b.lo[0]), y = fabsf(b.hi[1] - b.lo[1]), z = fabsf(b.hi[2] - b.lo[2]);
    return 2.0f * (x * y + x * z + y * z);
}

// order-preserving for all finite floats: positives compare as ints, negatives reversed as uints
// (-0 is folded into +0 first: as an int it would sort below every negative number)
__device__ inline void atomic_min_f(float* a, float v) {
    v += 0.0f;
    if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(a), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}
__device__ inline void atomic_max_f(float* a, float v) {
    v += 0.0f;
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(a), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}
__device__ inline void atomic_merge(Bounds* a, const Bounds& b) {
    for (int k = 0; k < 3; ++k) {
        atomic_min_f(&a->lo[k], b.lo[k]);
        atomic_max_f(&a->hi[k], b.hi[k]);
    }
}
__device__ inline void atomic_merge_point(Bounds* a, const float* p) {
    for (int k = 0; k < 3; ++k) {
        atomic_min_f(&a->lo[k], p[k]);
        atomic_max_f(&a->hi[k], p[k]);
    }
}

// The bin of a centroid coordinate between the node's centroid bounds; the bin pass and the flag
// pass must agree, so both come here.
__device__ inline uint32_t bin_of(float c, float lo, float hi) {
    const float extent = hi - lo;
    if (!(extent > 0.0f)) return 0u;
    const float k = (c - lo) * ((float)kBins / extent);
    const int b = (int)k;
    return (uint32_t)(b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b));
}

// Levels a node of `count` triangles still needs below it when every split from here on cuts in
// the middle: 0 for a leaf.
__device__ inline uint32_t levels_needed(uint32_t count, uint32_t leaf_max) {
    const uint32_t m = (count + leaf_max - 1u) / leaf_max;
    return m <= 1u ? 0u : 32u - (uint32_t)__clz(m - 1u);
}

__global__ void tri_info_kernel(const rt_vertex* v, const rt_tri* t, uint32_t n, TriInfo* info, uint32_t* ids,
                                uint32_t* slot_of, Bounds* root_box, Bounds* root_cent) {
    // the workgroup's triangles are folded in LDS first: a million atomics on the root's twelve words
    // would stand in one queue
    __shared__ Bounds s_box, s_cent;
    if (threadIdx.x == 0) {
        bounds_empty(s_box);
        bounds_empty(s_cent);
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) {
        const uint32_t idx[3] = {t[i].i0, t[i].i1, t[i].i2};
        TriInfo o;
        bounds_empty(o.box);
        for (int k = 0; k < 3; ++k) {
            const float a = v[idx[0]].pos[k], b = v[idx[1]].pos[k], c = v[idx[2]].pos[k];
            o.box.lo[k] = fminf(fminf(a, b), c);
            o.box.hi[k] = fmaxf(fmaxf(a, b), c);
            // the centroid the reference sorts by (bvh.rs:131-134), kept inside the triangle's own bounds
            const float m = (a + b + c) / 3.0f;
            o.c[k] = m < o.box.lo[k] ? o.box.lo[k] : (m > o.box.hi[k] ? o.box.hi[k] : m);
        }
        info[i] = o;
        ids[i] = i;
        slot_of[i] = 0u;
        atomic_merge(&s_box, o.box);
        atomic_merge_point(&s_cent, o.c);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomic_merge(root_box, s_box);
        atomic_merge(root_cent, s_cent);
    }
}

__global__ void init_bins_kernel(Bin* bins, size_t n_bins) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_bins) return;
    Bin b;
    b.count = 0u;
    bounds_empty(b.box);
    bounds_empty(b.cent);
    bins[i] = b;
}

// Every workgroup bins a run of kChunk consecutive positions.  On the upper levels a whole run
// lies inside one node (a node's triangles are consecutive and the level's list is in position
// order, so equal slots at both ends mean equal slots throughout): its 3 x 16 bins are then
// filled in LDS and each non-empty word goes to memory once per workgroup instead of once per
// triangle -- the top levels would otherwise queue a million atomics on a few hundred addresses.
constexpr uint32_t kChunk = 2048;

__device__ inline void bin_one(Bin* b, const TriInfo& ti) {
    atomicAdd(&b->count, 1u);
    atomic_merge(&b->box, ti.box);
    atomic_merge_point(&b->cent, ti.c);
}

__global__ void bin_kernel(uint32_t n, const uint32_t* ids, const uint32_t* slot_of, const Active* active,
                           const TriInfo* info, Bin* bins) {
    __shared__ Bin s_bins[3 * kBins];
    const uint32_t first = blockIdx.x * kChunk;
    const uint32_t end = first + kChunk < n ? first + kChunk : n;
    const uint32_t s_first = slot_of[first], s_last = slot_of[end - 1u];
    const bool one_node = s_first != kNone && s_first == s_last;  // uniform for the workgroup
    if (one_node) {
        for (uint32_t k = threadIdx.x; k < 3u * kBins; k += kThreads) {
            Bin e;
            e.count = 0u;
            bounds_empty(e.box);
            bounds_empty(e.cent);
            s_bins[k] = e;
        }
        __syncthreads();
        const Bounds cb = active[s_first].cent;
        for (uint32_t i = first + threadIdx.x; i < end; i += kThreads) {
            const TriInfo ti = info[ids[i]];
            for (int a = 0; a < 3; ++a) bin_one(&s_bins[(uint32_t)a * kBins + bin_of(ti.c[a], cb.lo[a], cb.hi[a])], ti);
        }
        __syncthreads();
        Bin* out = &bins[(size_t)s_first * 3u * kBins];
        for (uint32_t k = threadIdx.x; k < 3u * kBins; k += kThreads) {
            const Bin e = s_bins[k];
            if (e.count == 0u) continue;
            atomicAdd(&out[k].count, e.count);
            atomic_merge(&out[k].box, e.box);
            atomic_merge(&out[k].cent, e.cent);
        }
        return;
    }
    for (uint32_t i = first + threadIdx.x; i < end; i += kThreads) {
        const uint32_t s = slot_of[i];
        if (s == kNone) continue;
        const TriInfo ti = info[ids[i]];
        const Bounds cb = active[s].cent;
        for (int a = 0; a < 3; ++a)
            bin_one(&bins[((size_t)s * 3u + (uint32_t)a) * kBins + bin_of(ti.c[a], cb.lo[a], cb.hi[a])], ti);
    }
}

// One thread per active node: the cheapest of the 3 x 15 planes (bvh.rs:109), or a cut in the
// middle of the range.  Writes the decision into the node's Active record, the two children into
// children[2 s], children[2 s + 1], and the node's own record (its children are numbered
// node_base + 2 s and + 1).
__global__ void split_kernel(uint32_t n_active, Active* active, const Bin* bins, Child* children, rt_bvh* nodes,
                             uint32_t node_base, uint32_t leaf_max, uint32_t levels_below_children, uint32_t small_max) {
    const uint32_t s = blockIdx.x * kThreads + threadIdx.x;
    if (s >= n_active) return;
    Active a = active[s];
    float best_cost = __builtin_huge_valf();
    uint32_t best_axis = kAxisByPosition, best_bin = 0, best_left = 0;
    for (uint32_t axis = 0; axis < 3u; ++axis) {
        const Bin* b = &bins[((size_t)s * 3u + axis) * kBins];
        // right[j]: bins j+1 .. kBins-1
        float right_area[kBins];
        uint32_t right_count[kBins];
        Bounds acc;
        bounds_empty(acc);
        uint32_t cnt = 0;
        for (int j = kBins - 1; j >= 1; --j) {
            if (b[j].count != 0u) bounds_merge(acc, b[j].box);
            cnt += b[j].count;
            right_area[j - 1] = cnt ? bounds_area(acc) : 0.0f;
            right_count[j - 1] = cnt;
        }
        bounds_empty(acc);
        cnt = 0;
        for (int j = 0; j < kBins - 1; ++j) {
            if (b[j].count != 0u) bounds_merge(acc, b[j].box);
            cnt += b[j].count;
            if (cnt == 0u || right_count[j] == 0u) continue;
            const float cost = bounds_area(acc) * (float)cnt + right_area[j] * (float)right_count[j];
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = axis;
                best_bin = (uint32_t)j;
                best_left = cnt;
            }
        }
    }
    // a side too big for the levels left below the children: cut in the middle instead
    if (best_axis != kAxisByPosition) {
        const uint32_t big = best_left > a.count - best_left ? best_left : a.count - best_left;
        if (levels_needed(big, leaf_max) > levels_below_children) best_axis = kAxisByPosition;
    }
    Child c[2];
    if (best_axis == kAxisByPosition) {
        best_left = a.count / 2u;
        for (int side = 0; side < 2; ++side) {
            bounds_empty(c[side].box);  // gathered by the scatter pass
            bounds_empty(c[side].cent);
        }
    } else {
        const Bin* b = &bins[((size_t)s * 3u + best_axis) * kBins];
        for (int side = 0; side < 2; ++side) {
            bounds_empty(c[side].box);
            bounds_empty(c[side].cent);
        }
        for (uint32_t j = 0; j < (uint32_t)kBins; ++j) {
            if (b[j].count == 0u) continue;
            const int side = j <= best_bin ? 0 : 1;
            bounds_merge(c[side].box, b[j].box);
            bounds_merge(c[side].cent, b[j].cent);
        }
    }
    c[0].start = a.start;
    c[0].count = best_left;
    c[1].start = a.start + best_left;
    c[1].count = a.count - best_left;
    for (int side = 0; side < 2; ++side) {
        // levels_below_children == 0: the children are on the last level and stay leaves whatever they hold
        c[side].active = (c[side].count > leaf_max && levels_below_children > 0u) ? (c[side].count <= small_max ? 2u : 1u) : 0u;
        children[2u * s + (uint32_t)side] = c[side];
    }
    a.axis = best_axis;
    a.split_bin = best_bin;
    a.n_left = best_left;
    active[s] = a;
    rt_bvh* me = &nodes[a.node];
    me->child_1_or_first_tri = node_base + 2u * s;
    me->child_2_or_last_tri = node_base + 2u * s + 1u;
    me->mode = RT_MODE_CHILDREN;
}

__global__ void child_flags_kernel(uint32_t n_children, const Child* children, uint32_t* flags, uint32_t* small_flags) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_children) {
        flags[i] = children[i].active == 1u ? 1u : 0u;
        small_flags[i] = children[i].active == 2u ? 1u : 0u;
    }
}

// After the prefix sum over the children's `active` flags: node records of the children, the next
// level's list, and the parents' note of where their children went.
__global__ void emit_children_kernel(uint32_t n_children, const Child* children, const uint32_t* slot,
                                     const uint32_t* small_slot, Active* active, Active* next, rt_bvh* nodes,
                                     uint32_t node_base, SmallRoot* small_roots, uint32_t child_level) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_children) return;
    const Child c = children[i];
    rt_bvh o;
    memset(&o, 0, sizeof o);
    for (int k = 0; k < 3; ++k) {
        o.bounding_box.min[k] = c.box.lo[k];
        o.bounding_box.max[k] = c.box.hi[k];
    }
    // a child that is split on the next level has its links written then
    o.child_1_or_first_tri = c.start;
    o.child_2_or_last_tri = c.start + c.count - 1u;
    o.mode = RT_MODE_TRIANGLES;
    nodes[node_base + i] = o;
    uint32_t place = kNone;
    if (c.active == 2u) {
        // finished later by one workgroup; the passes see a leaf (its triangles pass through them unmoved)
        SmallRoot r;
        r.node = node_base + i;
        r.start = c.start;
        r.count = c.count;
        r.level = child_level;
        small_roots[small_slot[i]] = r;
    }
    if (c.active == 1u) {
        place = slot[i];
        Active a;
        memset(&a, 0, sizeof a);
        a.node = node_base + i;
        a.start = c.start;
        a.count = c.count;
        a.cent = c.cent;
        next[place] = a;
    }
    active[i >> 1].child_slot[i & 1u] = place;
}

__global__ void flag_kernel(uint32_t n, const uint32_t* ids, const uint32_t* slot_of, const Active* active,
                            const TriInfo* info, uint32_t* flags) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i];
    uint32_t f = 0u;
    if (s != kNone) {
        const Active& a = active[s];
        if (a.axis == kAxisByPosition) f = (i - a.start) < a.n_left ? 1u : 0u;
        else f = bin_of(info[ids[i]].c[a.axis], a.cent.lo[a.axis], a.cent.hi[a.axis]) <= a.split_bin ? 1u : 0u;
    }
    flags[i] = f;
}

// Stable partition of every active node's range: `pre` is the exclusive prefix sum of `flags`
// over all positions, so pre[i] - pre[start] is the number of left-going triangles of i's node
// before i.  Children cut by position have their boxes gathered here.
__global__ void scatter_kernel(uint32_t n, const uint32_t* ids, const uint32_t* slot_of, const Active* active,
                               const uint32_t* flags, const uint32_t* pre, const TriInfo* info, uint32_t* ids_out,
                               uint32_t* slot_out, Active* next, rt_bvh* nodes, uint32_t node_base) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i];
    if (s == kNone) {
        ids_out[i] = ids[i];
        slot_out[i] = kNone;
        return;
    }
    const Active& a = active[s];
    const uint32_t rank_left = pre[i] - pre[a.start];
    const uint32_t side = flags[i] ? 0u : 1u;
    const uint32_t to = side == 0u ? a.start + rank_left : a.start + a.n_left + ((i - a.start) - rank_left);
    const uint32_t id = ids[i];
    ids_out[to] = id;
    slot_out[to] = a.child_slot[side];
    if (a.axis == kAxisByPosition) {
        const TriInfo ti = info[id];
        rt_bvh* c = &nodes[node_base + 2u * s + side];
        for (int k = 0; k < 3; ++k) {
            atomic_min_f(&c->bounding_box.min[k], ti.box.lo[k]);
            atomic_max_f(&c->bounding_box.max[k], ti.box.hi[k]);
        }
        if (a.child_slot[side] != kNone) atomic_merge_point(&next[a.child_slot[side]].cent, ti.c);
    }
}

// Last pass over the finished nodes: boxes with (next to) no thickness along an axis are widened (bvh_pad.hpp).
__global__ void widen_flat_boxes_kernel(rt_bvh* nodes, uint32_t n_nodes, float pad) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_nodes) rt::widen_flat_box(nodes[i].bounding_box.min, nodes[i].bounding_box.max, pad);
}

__global__ void gather_triangles_kernel(const rt_tri* in, const uint32_t* ids, uint32_t n, rt_tri* out) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) out[i] = in[ids[i]];
}

// ---------------------------------------------------------------------------------------------
// One workgroup finishes the subtree of one small node.
//
// The subtree's triangle order lives in LDS (two buffers, one per level parity); a level's nodes are
// dealt to the four waves, and a wave does for its node what the passes do for a level: centroid bounds
// (what the passes carry along in Active::cent: the exact bounds of the member centroids), bins (LDS
// atomics into the wave's own 3 x 16 bins), the 45 planes priced by 45 lanes, the cheapest by a minimum
// over (cost, plane number) -- the serial sweep keeps the first of equal costs in that order --, the
// children's boxes from the bins or, after a cut by position, from the triangles, and a stable
// partition of the node's part of the order by ballots.  New nodes are numbered the way the passes number
// them, two per node in the order of the level's list, counted from the start of the subtree ("local"
// numbers; the root of the subtree keeps its record in the output array), and go to
// scratch[2 * start + local]: a subtree of c triangles has at most 2 c - 2 new nodes, and the subtrees'
// triangle ranges are disjoint.  number_small_kernel moves them to their final places.
// A wave's LDS operations are carried out in program order, so within a wave "all lanes have written"
// needs no barrier, only the compiler kept from moving accesses across (wave_fence).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFinishThreads = 256;
constexpr uint32_t kFinishWaves = kFinishThreads / 64u;
constexpr uint32_t kRootLocal = 0xFFFFFFFFu;

struct LNode {
    uint32_t local;        // its record: scratch[2 * start of the subtree + local]; kRootLocal: the subtree's root
    uint32_t start_count;  // start within the subtree | count << 16
};

__device__ inline void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(kFinishThreads) void finish_small_kernel(const SmallRoot* roots, uint32_t* ids,
                                                                      const TriInfo* info, rt_bvh* scratch,
                                                                      uint32_t* n_new, uint32_t leaf_max,
                                                                      uint32_t max_levels) {
    typedef hipcub::BlockScan<uint32_t, kFinishThreads> Scan;
    constexpr uint32_t kPerThread = kSmallMax / kFinishThreads;  // children of a level per thread, at most
    __shared__ uint32_t s_ids[2][kSmallMax];
    __shared__ LNode s_cur[kSmallMax / 2u];  // a node that is split holds two triangles or more
    __shared__ LNode s_child[kSmallMax];
    __shared__ uint32_t s_flag[kSmallMax];
    __shared__ Bin s_bins[kFinishWaves][3 * kBins];
    __shared__ Bounds s_tmp[kFinishWaves][2];
    __shared__ typename Scan::TempStorage s_scan;

    const SmallRoot r = roots[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    rt_bvh* const out = scratch + 2u * (size_t)r.start;
    for (uint32_t i = threadIdx.x; i < r.count; i += kFinishThreads) s_ids[0][i] = ids[r.start + i];
    if (threadIdx.x == 0) {
        s_cur[0].local = kRootLocal;
        s_cur[0].start_count = r.count << 16;
    }
    __syncthreads();
    uint32_t n_cur = 1, local_base = 0, level = r.level, buf = 0;
    while (n_cur != 0u) {
        const uint32_t levels_below_children = max_levels - 2u - level;
        Bin* const bins = s_bins[wave];
        Bounds* const tmp = s_tmp[wave];
        for (uint32_t s = wave; s < n_cur; s += kFinishWaves) {  // wave-uniform
            const LNode node = s_cur[s];
            const uint32_t start = node.start_count & 0xFFFFu, count = node.start_count >> 16;
            const uint32_t* const my_ids = &s_ids[buf][start];
            // (a) bounds of the centroids; the bins emptied
            if (lane == 0) bounds_empty(tmp[0]);
            for (uint32_t k = lane; k < 3u * kBins; k += 64u) {
                Bin e;
                e.count = 0u;
                bounds_empty(e.box);
                bounds_empty(e.cent);
                bins[k] = e;
            }
            wave_fence();
            for (uint32_t i = lane; i < count; i += 64u) atomic_merge_point(&tmp[0], info[my_ids[i]].c);
            wave_fence();
            const Bounds cent = tmp[0];
            // (b) the bins
            for (uint32_t i = lane; i < count; i += 64u) {
                const TriInfo ti = info[my_ids[i]];
                for (int a = 0; a < 3; ++a) bin_one(&bins[(uint32_t)a * kBins + bin_of(ti.c[a], cent.lo[a], cent.hi[a])], ti);
            }
            wave_fence();
            // (c) lane 15 a + j prices the plane behind bin j of axis a
            unsigned long long key = ~0ull;
            uint32_t my_left = 0;
            if (lane < 3u * (kBins - 1u)) {
                const uint32_t axis = lane / (kBins - 1u), j = lane % (kBins - 1u);
                const Bin* b = &bins[axis * kBins];
                Bounds acc_l, acc_r;
                bounds_empty(acc_l);
                bounds_empty(acc_r);
                uint32_t cnt_l = 0, cnt_r = 0;
                for (uint32_t k = 0; k < (uint32_t)kBins; ++k) {
                    const Bin e = b[k];
                    if (e.count == 0u) continue;
                    if (k <= j) {
                        bounds_merge(acc_l, e.box);
                        cnt_l += e.count;
                    } else {
                        bounds_merge(acc_r, e.box);
                        cnt_r += e.count;
                    }
                }
                if (cnt_l != 0u && cnt_r != 0u) {
                    const float right_area = bounds_area(acc_r);
                    const float cost = bounds_area(acc_l) * (float)cnt_l + right_area * (float)cnt_r;
                    if (cost < __builtin_huge_valf()) key = ((unsigned long long)__float_as_uint(cost) << 32) | lane;
                }
                my_left = cnt_l;
            }
            unsigned long long best = key;
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long other = __shfl_xor(best, off, 64);
                best = other < best ? other : best;
            }
            uint32_t best_axis = kAxisByPosition, best_bin = 0, n_left = 0;
            if (best != ~0ull) {
                const uint32_t plane = (uint32_t)best;  // the lane that priced it
                best_axis = plane / (kBins - 1u);
                best_bin = plane % (kBins - 1u);
                n_left = (uint32_t)__shfl((int)my_left, (int)plane, 64);
                // a side too big for the levels left below the children: cut in the middle instead
                const uint32_t big = n_left > count - n_left ? n_left : count - n_left;
                if (levels_needed(big, leaf_max) > levels_below_children) best_axis = kAxisByPosition;
            }
            if (best_axis == kAxisByPosition) n_left = count / 2u;
            // (d) the children's boxes
            if (lane < 2u) bounds_empty(tmp[lane]);
            wave_fence();
            if (best_axis == kAxisByPosition) {
                for (uint32_t i = lane; i < count; i += 64u) atomic_merge(&tmp[i < n_left ? 0 : 1], info[my_ids[i]].box);
            } else if (lane < 2u) {
                const Bin* b = &bins[best_axis * kBins];
                Bounds acc;
                bounds_empty(acc);
                for (uint32_t k = 0; k < (uint32_t)kBins; ++k) {
                    if (b[k].count == 0u) continue;
                    if ((k <= best_bin) == (lane == 0u)) bounds_merge(acc, b[k].box);
                }
                tmp[lane] = acc;
            }
            wave_fence();
            // (e) the children's records, the node's links, the children's entries for the next level
            const uint32_t c_count[2] = {n_left, count - n_left};
            const uint32_t c_start[2] = {start, start + n_left};
            uint32_t c_active[2];
            for (int side = 0; side < 2; ++side)
                c_active[side] = (c_count[side] > leaf_max && levels_below_children > 0u) ? 1u : 0u;
            if (lane < 2u) {
                const uint32_t side = lane;
                const Bounds cb = tmp[side];
                rt_bvh o;
                memset(&o, 0, sizeof o);
                for (int k = 0; k < 3; ++k) {
                    o.bounding_box.min[k] = cb.lo[k];
                    o.bounding_box.max[k] = cb.hi[k];
                }
                o.child_1_or_first_tri = r.start + c_start[side];
                o.child_2_or_last_tri = r.start + c_start[side] + c_count[side] - 1u;
                o.mode = RT_MODE_TRIANGLES;  // a child that is split on the next level has its links written then
                out[local_base + 2u * s + side] = o;
                LNode c;
                c.local = local_base + 2u * s + side;
                c.start_count = c_start[side] | (c_count[side] << 16);
                s_child[2u * s + side] = c;
                s_flag[2u * s + side] = c_active[side];
            }
            if (lane == 0u && node.local != kRootLocal) {
                rt_bvh* me = &out[node.local];
                me->child_1_or_first_tri = local_base + 2u * s;  // local numbers: number_small_kernel adds the subtree's base
                me->child_2_or_last_tri = local_base + 2u * s + 1u;
                me->mode = RT_MODE_CHILDREN;
            }
            // (f) stable partition of the node's part of the order; a leaf's part is final and goes out
            uint32_t left_done = 0, right_done = 0;
            for (uint32_t i0 = 0; i0 < count; i0 += 64u) {
                const uint32_t i = i0 + lane;
                const bool valid = i < count;
                const uint32_t id = valid ? my_ids[i] : 0u;
                bool go_left = false;
                if (valid) {
                    if (best_axis == kAxisByPosition) go_left = i < n_left;
                    else go_left = bin_of(info[id].c[best_axis], cent.lo[best_axis], cent.hi[best_axis]) <= best_bin;
                }
                const unsigned long long ml = __ballot(valid && go_left), mr = __ballot(valid && !go_left);
                if (valid) {
                    const uint32_t pos = go_left ? left_done + (uint32_t)__popcll(ml & below)
                                                 : n_left + right_done + (uint32_t)__popcll(mr & below);
                    s_ids[buf ^ 1u][start + pos] = id;
                    if (c_active[go_left ? 0 : 1] == 0u) ids[r.start + start + pos] = id;
                }
                left_done += (uint32_t)__popcll(ml);
                right_done += (uint32_t)__popcll(mr);
            }
        }
        __syncthreads();
        // the children that are split in their turn, in order, are the next level's list
        const uint32_t n_children = 2u * n_cur;
        uint32_t flags[kPerThread], places[kPerThread], n_next = 0;
        for (uint32_t k = 0; k < kPerThread; ++k) {
            const uint32_t idx = threadIdx.x * kPerThread + k;
            flags[k] = idx < n_children ? s_flag[idx] : 0u;
        }
        Scan(s_scan).ExclusiveSum(flags, places, n_next);
        for (uint32_t k = 0; k < kPerThread; ++k) {
            const uint32_t idx = threadIdx.x * kPerThread + k;
            if (flags[k]) s_cur[places[k]] = s_child[idx];
        }
        local_base += n_children;
        n_cur = n_next;
        level += 1u;
        buf ^= 1u;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_new[blockIdx.x] = local_base;
}

// The subtrees' nodes to their final places, behind the nodes of the passes: subtree k's node number
// `local` becomes n_big + (nodes of the subtrees before k) + local, and the links follow.
__global__ __launch_bounds__(kFinishThreads) void number_small_kernel(const SmallRoot* roots, const rt_bvh* scratch,
                                                                      const uint32_t* n_new, const uint32_t* before,
                                                                      uint32_t n_big, rt_bvh* nodes) {
    const SmallRoot r = roots[blockIdx.x];
    const uint32_t base = n_big + before[blockIdx.x], m = n_new[blockIdx.x];
    const rt_bvh* const in = scratch + 2u * (size_t)r.start;
    for (uint32_t j = threadIdx.x; j < m; j += kFinishThreads) {
        rt_bvh o = in[j];
        if (o.mode == RT_MODE_CHILDREN) {
            o.child_1_or_first_tri += base;
            o.child_2_or_last_tri += base;
        }
        nodes[base + j] = o;
    }
    if (threadIdx.x == 0 && m != 0u) {
        rt_bvh* me = &nodes[r.node];
        me->child_1_or_first_tri = base;
        me->child_2_or_last_tri = base + 1u;
        me->mode = RT_MODE_CHILDREN;
    }
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t alloc(size_t n) {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = n ? n : 1;
        return hipMalloc(&p, cap * sizeof(T));
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

#define RT_TRY(call)                                 \
    do {                                             \
        if ((call) != hipSuccess) return RT_ERR_HIP; \
    } while (0)

inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + kThreads - 1) / kThreads)); }

}  // namespace

// small_max: triangles of a subtree that one workgroup finishes in LDS, at most kSmallMax; 0: every level
// goes through the passes (the same tree, numbered level by level throughout).
static int build_sah(int device, const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles, size_t n_triangles,
                     uint32_t leaf_max, uint32_t max_levels, uint32_t small_max, rt_bvh* nodes, size_t* n_nodes) {
    if (!vertices || !triangles || !n_nodes || n_triangles == 0 || leaf_max == 0) return RT_ERR_INVALID;
    if (max_levels == 0) max_levels = 33;  // what a 32-entry traversal stack takes
    if (max_levels > 65 || small_max > kSmallMax) return RT_ERR_INVALID;
    if (n_triangles >= 0x7FFFFFFFull || n_vertices > 0xFFFFFFFFull) return RT_ERR_LIMIT;
    if (!rt::mesh_is_buildable(vertices, n_vertices, triangles, n_triangles)) return RT_ERR_INVALID;
    const uint32_t n = (uint32_t)n_triangles;
    const size_t total = 2 * (size_t)n - 1;  // upper bound
    if (!nodes) {
        *n_nodes = total;
        return RT_OK;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return RT_ERR_HIP;
    RT_TRY(hipSetDevice(device));
    const size_t max_active = (size_t)n / ((size_t)leaf_max + 1u) + 2u;
    // a small root holds more than leaf_max triangles, and the small roots' ranges are disjoint
    const size_t max_small = small_max ? max_active : 1;
    DevBuf<rt_vertex> d_v;
    DevBuf<rt_tri> d_t, d_sorted;
    DevBuf<TriInfo> d_info;
    DevBuf<uint32_t> d_ids[2], d_slot[2], d_flags, d_pre, d_cflags, d_cslot, d_sflags, d_sslot, d_new, d_before;
    DevBuf<Active> d_active[2];
    DevBuf<Child> d_children;
    DevBuf<Bin> d_bins;
    DevBuf<rt_bvh> d_nodes, d_scratch;
    DevBuf<SmallRoot> d_small;
    DevBuf<char> d_tmp;
    RT_TRY(d_v.alloc(n_vertices));
    RT_TRY(d_t.alloc(n));
    RT_TRY(d_sorted.alloc(n));
    RT_TRY(d_info.alloc(n));
    for (int k = 0; k < 2; ++k) {
        RT_TRY(d_ids[k].alloc(n));
        RT_TRY(d_slot[k].alloc(n));
        RT_TRY(d_active[k].alloc(max_active));
    }
    RT_TRY(d_flags.alloc(n));
    RT_TRY(d_pre.alloc(n));
    RT_TRY(d_cflags.alloc(2 * max_active));
    RT_TRY(d_cslot.alloc(2 * max_active));
    RT_TRY(d_sflags.alloc(2 * max_active));
    RT_TRY(d_sslot.alloc(2 * max_active));
    RT_TRY(d_children.alloc(2 * max_active));
    RT_TRY(d_nodes.alloc(total));
    RT_TRY(d_small.alloc(max_small));
    RT_TRY(d_new.alloc(max_small));
    RT_TRY(d_before.alloc(max_small));
    if (small_max) RT_TRY(d_scratch.alloc(2 * (size_t)n));
    RT_TRY(hipMemcpy(d_v.p, vertices, n_vertices * sizeof(rt_vertex), hipMemcpyHostToDevice));
    RT_TRY(hipMemcpy(d_t.p, triangles, n * sizeof(rt_tri), hipMemcpyHostToDevice));
    size_t tmp_bytes = 0, tmp_bytes2 = 0, tmp_bytes3 = 0;
    RT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flags.p, d_pre.p, (int)n));
    RT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes2, d_cflags.p, d_cslot.p, (int)(2 * max_active)));
    RT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes3, d_new.p, d_before.p, (int)max_small));
    tmp_bytes = tmp_bytes > tmp_bytes2 ? tmp_bytes : tmp_bytes2;
    RT_TRY(d_tmp.alloc(tmp_bytes > tmp_bytes3 ? tmp_bytes : tmp_bytes3));
    tmp_bytes = d_tmp.cap;

    // the root: node 0, every triangle
    Active root;
    memset(&root, 0, sizeof root);
    root.node = 0;
    root.start = 0;
    root.count = n;
    bounds_empty(root.cent);
    RT_TRY(hipMemcpy(d_active[0].p, &root, sizeof root, hipMemcpyHostToDevice));
    rt_bvh root_node;
    memset(&root_node, 0, sizeof root_node);
    for (int k = 0; k < 3; ++k) {
        root_node.bounding_box.min[k] = __builtin_huge_valf();
        root_node.bounding_box.max[k] = -__builtin_huge_valf();
    }
    root_node.child_1_or_first_tri = 0;
    root_node.child_2_or_last_tri = n - 1;
    root_node.mode = RT_MODE_TRIANGLES;
    RT_TRY(hipMemcpy(d_nodes.p, &root_node, sizeof root_node, hipMemcpyHostToDevice));
    // rt_bounding_box is (min[3], pad, max[3], pad): not a Bounds; gather the root's box into a Bounds first
    DevBuf<Bounds> d_root_box;
    RT_TRY(d_root_box.alloc(1));
    Bounds empty;
    bounds_empty(empty);
    RT_TRY(hipMemcpy(d_root_box.p, &empty, sizeof empty, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tri_info_kernel, grid_for(n), dim3(kThreads), 0, 0, d_v.p, d_t.p, n, d_info.p, d_ids[0].p,
                       d_slot[0].p, d_root_box.p, &d_active[0].p->cent);
    RT_TRY(hipGetLastError());
    Bounds rb;
    RT_TRY(hipMemcpy(&rb, d_root_box.p, sizeof rb, hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) {
        root_node.bounding_box.min[k] = rb.lo[k];
        root_node.bounding_box.max[k] = rb.hi[k];
    }
    RT_TRY(hipMemcpy(d_nodes.p, &root_node, sizeof root_node, hipMemcpyHostToDevice));

    const bool root_splits = n > leaf_max && max_levels > 1;
    uint32_t n_active = (root_splits && n > small_max) ? 1u : 0u;
    uint32_t n_small = 0;
    if (root_splits && n <= small_max) {  // the whole tree is one workgroup's
        SmallRoot r;
        r.node = 0;
        r.start = 0;
        r.count = n;
        r.level = 0;
        RT_TRY(hipMemcpy(d_small.p, &r, sizeof r, hipMemcpyHostToDevice));
        n_small = 1;
    }
    uint32_t n_nodes_out = 1;
    int cur = 0;
    for (uint32_t level = 0; n_active != 0u; ++level) {
        // children of this level's nodes are on level + 1 (0-based); below them max_levels - 2 - level more
        const uint32_t levels_below_children = max_levels - 2u - level;
        const size_t n_bins = (size_t)n_active * 3u * kBins;
        if (d_bins.cap < n_bins) RT_TRY(d_bins.alloc(n_bins + n_bins / 2));
        hipLaunchKernelGGL(init_bins_kernel, grid_for(n_bins), dim3(kThreads), 0, 0, d_bins.p, n_bins);
        hipLaunchKernelGGL(bin_kernel, dim3((n + kChunk - 1) / kChunk), dim3(kThreads), 0, 0, n, d_ids[cur].p, d_slot[cur].p, d_active[cur].p,
                           d_info.p, d_bins.p);
        hipLaunchKernelGGL(split_kernel, grid_for(n_active), dim3(kThreads), 0, 0, n_active, d_active[cur].p, d_bins.p,
                           d_children.p, d_nodes.p, n_nodes_out, leaf_max, levels_below_children, small_max);
        const uint32_t n_children = 2u * n_active;
        hipLaunchKernelGGL(child_flags_kernel, grid_for(n_children), dim3(kThreads), 0, 0, n_children, d_children.p,
                           d_cflags.p, d_sflags.p);
        RT_TRY(hipGetLastError());
        RT_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_cflags.p, d_cslot.p, (int)n_children));
        RT_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_sflags.p, d_sslot.p, (int)n_children));
        hipLaunchKernelGGL(emit_children_kernel, grid_for(n_children), dim3(kThreads), 0, 0, n_children, d_children.p,
                           d_cslot.p, d_sslot.p, d_active[cur].p, d_active[cur ^ 1].p, d_nodes.p, n_nodes_out,
                           d_small.p + n_small, level + 1u);
        hipLaunchKernelGGL(flag_kernel, grid_for(n), dim3(kThreads), 0, 0, n, d_ids[cur].p, d_slot[cur].p,
                           d_active[cur].p, d_info.p, d_flags.p);
        RT_TRY(hipGetLastError());
        RT_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_flags.p, d_pre.p, (int)n));
        hipLaunchKernelGGL(scatter_kernel, grid_for(n), dim3(kThreads), 0, 0, n, d_ids[cur].p, d_slot[cur].p,
                           d_active[cur].p, d_flags.p, d_pre.p, d_info.p, d_ids[cur ^ 1].p, d_slot[cur ^ 1].p,
                           d_active[cur ^ 1].p, d_nodes.p, n_nodes_out);
        RT_TRY(hipGetLastError());
        // how many of the children go on, and how many are left to the workgroups: the last child's place + its flag
        uint32_t last[4] = {0, 0, 0, 0};
        RT_TRY(hipMemcpy(&last[0], d_cflags.p + (n_children - 1), 4, hipMemcpyDeviceToHost));
        RT_TRY(hipMemcpy(&last[1], d_cslot.p + (n_children - 1), 4, hipMemcpyDeviceToHost));
        RT_TRY(hipMemcpy(&last[2], d_sflags.p + (n_children - 1), 4, hipMemcpyDeviceToHost));
        RT_TRY(hipMemcpy(&last[3], d_sslot.p + (n_children - 1), 4, hipMemcpyDeviceToHost));
        n_nodes_out += n_children;
        n_active = last[0] + last[1];
        n_small += last[2] + last[3];
        if (n_active > max_active || n_small > max_small) return RT_ERR_HIP;  // cannot happen: each holds more than leaf_max triangles
        cur ^= 1;
    }
    if (n_small != 0u) {
        // the small subtrees, each by one workgroup; their nodes numbered behind those of the passes
        hipLaunchKernelGGL(finish_small_kernel, dim3(n_small), dim3(kFinishThreads), 0, 0, d_small.p, d_ids[cur].p,
                           d_info.p, d_scratch.p, d_new.p, leaf_max, max_levels);
        RT_TRY(hipGetLastError());
        RT_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, d_new.p, d_before.p, (int)n_small));
        uint32_t last_new = 0, last_before = 0;
        RT_TRY(hipMemcpy(&last_new, d_new.p + (n_small - 1), 4, hipMemcpyDeviceToHost));
        RT_TRY(hipMemcpy(&last_before, d_before.p + (n_small - 1), 4, hipMemcpyDeviceToHost));
        const size_t n_all = (size_t)n_nodes_out + last_new + last_before;
        if (n_all > total) return RT_ERR_HIP;  // cannot happen: a tree of n leaves at most
        hipLaunchKernelGGL(number_small_kernel, dim3(n_small), dim3(kFinishThreads), 0, 0, d_small.p, d_scratch.p,
                           d_new.p, d_before.p, n_nodes_out, d_nodes.p);
        RT_TRY(hipGetLastError());
        n_nodes_out = (uint32_t)n_all;
    }
    if (*n_nodes < n_nodes_out) return RT_ERR_INVALID;
    hipLaunchKernelGGL(widen_flat_boxes_kernel, grid_for(n_nodes_out), dim3(kThreads), 0, 0, d_nodes.p, n_nodes_out,
                       rt::flat_box_pad(rb.lo, rb.hi));
    hipLaunchKernelGGL(gather_triangles_kernel, grid_for(n), dim3(kThreads), 0, 0, d_t.p, d_ids[cur].p, n, d_sorted.p);
    RT_TRY(hipGetLastError());
    RT_TRY(hipDeviceSynchronize());
    RT_TRY(hipMemcpy(nodes, d_nodes.p, (size_t)n_nodes_out * sizeof(rt_bvh), hipMemcpyDeviceToHost));
    RT_TRY(hipMemcpy(triangles, d_sorted.p, n * sizeof(rt_tri), hipMemcpyDeviceToHost));
    *n_nodes = n_nodes_out;
    return RT_OK;
}

extern "C" int rt_create_bvh_device_sah(int device, const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles,
                                        size_t n_triangles, uint32_t leaf_max, uint32_t max_levels, rt_bvh* nodes,
                                        size_t* n_nodes) {
    return build_sah(device, vertices, n_vertices, triangles, n_triangles, leaf_max, max_levels, kSmallMax, nodes, n_nodes);
}

extern "C" int rt_create_bvh_device_sah_passes_only(int device, const rt_vertex* vertices, size_t n_vertices,
                                                    rt_tri* triangles, size_t n_triangles, uint32_t leaf_max,
                                                    uint32_t max_levels, rt_bvh* nodes, size_t* n_nodes) {
    return build_sah(device, vertices, n_vertices, triangles, n_triangles, leaf_max, max_levels, 0u, nodes, n_nodes);
}
