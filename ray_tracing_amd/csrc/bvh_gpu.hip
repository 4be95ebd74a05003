// A BVH builder that runs on the GPU: linear BVH over Morton-sorted triangles
// (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and
// k-d Trees", 2012).  It is NOT the reference's builder (bvh.rs evaluates
// surface-area costs at candidate splits, host_scene.cpp reproduces that): it
// produces a different, somewhat less tight tree, two orders of magnitude
// sooner, in the reference's own node format (shared::Bvh, 48 bytes, inclusive
// global triangle ranges), so the result feeds rt_upload_scene -- or the
// reference's shader -- like any other tree.
//
//   1. centroid of every triangle -> 63-bit Morton code inside the scene box
//   2. radix sort of (code, triangle index) pairs            (hipCUB)
//   3. a binary radix tree over the sorted triangles: one thread per internal
//      node finds its range and split from the common prefixes of the codes
//      (ties broken by position)
//   4. boxes bottom-up: the second child to arrive at a node merges
//   5. every subtree of at most `leaf_max` triangles whose parent holds more
//      becomes one leaf (its triangles are consecutive in the sorted order);
//      the surviving nodes are numbered by a prefix sum and written out
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rt_hip.h"
#include "bvh_pad.hpp"
#include "host_scene.hpp"

namespace {

constexpr int kThreads = 256;

struct Box {
    float lo[3], hi[3];
};

__device__ __forceinline__ void atomic_min_f(float* a, float v) {
    // order-preserving for all finite floats: positives compare as ints, negatives reversed as uints
    if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(a), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* a, float v) {
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(a), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}

__global__ void scene_bounds(const rt_vertex* v, const rt_tri* t, uint32_t n, float* bounds /* lo[3], hi[3] */) {
    uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t idx[3] = {t[i].i0, t[i].i1, t[i].i2};
    for (int k = 0; k < 3; ++k) {
        float c = (v[idx[0]].pos[k] + v[idx[1]].pos[k] + v[idx[2]].pos[k]) * (1.0f / 3.0f);
        if (!(fabsf(c) < __builtin_huge_valf())) continue;  // NaN or infinite: takes no part in the scene box
        atomic_min_f(&bounds[k], c);
        atomic_max_f(&bounds[3 + k], c);
    }
}

__global__ void morton_codes(const rt_vertex* v, const rt_tri* t, uint32_t n, const float* bounds, uint64_t* keys,
                             uint32_t* order) {
    uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t idx[3] = {t[i].i0, t[i].i1, t[i].i2};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float c = (v[idx[0]].pos[k] + v[idx[1]].pos[k] + v[idx[2]].pos[k]) * (1.0f / 3.0f);
        float ext = bounds[3 + k] - bounds[k];
        float u = ext > 0.0f ? (c - bounds[k]) / ext : 0.0f;
        u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);  // also maps NaN to 0
        q[k] = (uint32_t)(u * 2097151.0f);
    }
    // The 63 key bits, most significant first, always come from the axis along which the current
    // cell is longest (each bit halves the cell along its axis), not from x, y, z in turn: a flat
    // or elongated scene is then not split along its short axes until the cells are about cubic.
    float e[3] = {bounds[3] - bounds[0], bounds[4] - bounds[1], bounds[5] - bounds[2]};
    int used[3] = {0, 0, 0};
    uint64_t key = 0;
    for (int b = 0; b < 63; ++b) {
        int a = -1;
        for (int k = 0; k < 3; ++k)
            if (used[k] < 21 && (a < 0 || e[k] > e[a])) a = k;
        key = key << 1 | ((q[a] >> (20 - used[a])) & 1u);
        used[a] += 1;
        e[a] *= 0.5f;
    }
    keys[i] = key;
    order[i] = i;
}

__global__ void gather_triangles(const rt_tri* in, const uint32_t* order, uint32_t n, rt_tri* out) {
    uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) out[i] = in[order[i]];
}

// Length of the common prefix of the keys of leaves a and b; equal keys are told apart by
// the leaf numbers themselves.  -1 outside the array.
__device__ __forceinline__ int delta(const uint64_t* keys, int n_leaves, int a, int b) {
    if (b < 0 || b >= n_leaves) return -1;
    uint64_t ka = keys[a], kb = keys[b];
    if (ka == kb) return 64 + __clz((unsigned)a ^ (unsigned)b);
    return __clzll((long long)(ka ^ kb));
}

// Internal node i (0 .. n_leaves-2) of the full tree, whose leaves are single triangles.
// Node numbers of the full tree: internal k -> k, leaf k -> n_leaves - 1 + k.
__global__ void build_internal(const uint64_t* keys, int n_leaves, uint32_t* left, uint32_t* right, uint32_t* parent,
                               uint32_t* range_lo, uint32_t* range_hi) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_leaves - 1) return;
    auto D = [&](int a, int b) { return delta(keys, n_leaves, a, b); };
    const int d = D(i, i + 1) - D(i, i - 1) >= 0 ? 1 : -1;
    const int d_min = D(i, i - d);
    int l_max = 2;
    while (D(i, i + l_max * d) > d_min) l_max *= 2;
    int l = 0;
    for (int t = l_max / 2; t >= 1; t /= 2)
        if (D(i, i + (l + t) * d) > d_min) l += t;
    const int j = i + l * d;
    const int d_node = D(i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (D(i, i + (s + t) * d) > d_node) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t l_node = lo == gamma ? (uint32_t)(n_leaves - 1 + gamma) : (uint32_t)gamma;
    const uint32_t r_node = hi == gamma + 1 ? (uint32_t)(n_leaves - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    left[i] = l_node;
    right[i] = r_node;
    parent[l_node] = (uint32_t)i;
    parent[r_node] = (uint32_t)i;
    range_lo[i] = (uint32_t)lo;
    range_hi[i] = (uint32_t)hi;
}

__device__ __forceinline__ void store_box(rt_bvh& n, const Box& b) {
    for (int k = 0; k < 3; ++k) {
        n.bounding_box.min[k] = b.lo[k];
        n.bounding_box.max[k] = b.hi[k];
    }
}

// One thread per leaf: its box, then up the tree; the second child to reach a node merges.
__global__ void fit_boxes(const rt_vertex* v, const rt_tri* sorted, int n_leaves, const uint32_t* left,
                          const uint32_t* right, const uint32_t* parent, uint32_t* arrived, rt_bvh* nodes) {
    int leaf = blockIdx.x * kThreads + threadIdx.x;
    if (leaf >= n_leaves) return;
    const uint32_t first = (uint32_t)leaf, last = first;
    Box b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = __builtin_huge_valf();
        b.hi[k] = -__builtin_huge_valf();
    }
    for (uint32_t t = first; t <= last; ++t) {
        const uint32_t idx[3] = {sorted[t].i0, sorted[t].i1, sorted[t].i2};
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) {
                float p = v[idx[c]].pos[k];
                b.lo[k] = p < b.lo[k] ? p : b.lo[k];
                b.hi[k] = p > b.hi[k] ? p : b.hi[k];
            }
    }
    uint32_t node = (uint32_t)(n_leaves - 1 + leaf);
    rt_bvh out;
    memset(&out, 0, sizeof out);
    store_box(out, b);
    out.child_1_or_first_tri = first;
    out.child_2_or_last_tri = last;
    out.mode = RT_MODE_TRIANGLES;
    nodes[node] = out;
    while (node != 0u) {
        const uint32_t p = parent[node];
        // Release: the box written above must have left this XCD's L2 before the arrival counts
        // (the explicit wait keeps the counter from overtaking the write-back).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_fetch_add(&arrived[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
        // Acquire: the sibling's box was written by another workgroup, maybe on another XCD; drop
        // whatever stale copy of that line this CU's L1 holds before reading it.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const rt_bvh a = nodes[left[p]], c = nodes[right[p]];
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = a.bounding_box.min[k] < c.bounding_box.min[k] ? a.bounding_box.min[k] : c.bounding_box.min[k];
            b.hi[k] = a.bounding_box.max[k] > c.bounding_box.max[k] ? a.bounding_box.max[k] : c.bounding_box.max[k];
        }
        memset(&out, 0, sizeof out);
        store_box(out, b);
        out.child_1_or_first_tri = left[p];
        out.child_2_or_last_tri = right[p];
        out.mode = RT_MODE_CHILDREN;
        nodes[p] = out;
        node = p;
    }
}

// Full-tree node -> how many triangles it holds and where they start.
__device__ __forceinline__ void node_range(uint32_t node, int n_leaves, const uint32_t* range_lo, const uint32_t* range_hi,
                                           uint32_t& lo, uint32_t& hi) {
    if (node >= (uint32_t)(n_leaves - 1)) {
        lo = hi = node - (uint32_t)(n_leaves - 1);
    } else {
        lo = range_lo[node];
        hi = range_hi[node];
    }
}

// keep[node] = 1 when the node survives the collapse: the root, and every node whose parent
// holds more than leaf_max triangles.
__global__ void mark_kept(int n_leaves, uint32_t leaf_max, const uint32_t* parent, const uint32_t* range_lo,
                          const uint32_t* range_hi, uint32_t* keep) {
    uint32_t node = blockIdx.x * kThreads + threadIdx.x;
    if (node >= (uint32_t)(2 * n_leaves - 1)) return;
    uint32_t k = 1;
    if (node != 0u) {
        uint32_t lo, hi;
        node_range(parent[node], n_leaves, range_lo, range_hi, lo, hi);
        k = hi - lo + 1u > leaf_max ? 1u : 0u;
    }
    keep[node] = k;
}

__global__ void emit_nodes(int n_leaves, uint32_t leaf_max, const uint32_t* left, const uint32_t* right,
                           const uint32_t* range_lo, const uint32_t* range_hi, const uint32_t* keep,
                           const uint32_t* slot, const rt_bvh* full, rt_bvh* out) {
    uint32_t node = blockIdx.x * kThreads + threadIdx.x;
    if (node >= (uint32_t)(2 * n_leaves - 1) || !keep[node]) return;
    uint32_t lo, hi;
    node_range(node, n_leaves, range_lo, range_hi, lo, hi);
    rt_bvh o = full[node];  // the box; widened where it has no thickness along an axis (bvh_pad.hpp)
    const rt_bvh root = full[0];
    rt::widen_flat_box(o.bounding_box.min, o.bounding_box.max, rt::flat_box_pad(root.bounding_box.min, root.bounding_box.max));
    if (hi - lo + 1u <= leaf_max) {
        o.child_1_or_first_tri = lo;
        o.child_2_or_last_tri = hi;
        o.mode = RT_MODE_TRIANGLES;
    } else {
        o.child_1_or_first_tri = slot[left[node]];
        o.child_2_or_last_tri = slot[right[node]];
        o.mode = RT_MODE_CHILDREN;
    }
    out[slot[node]] = o;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc(&p, (n ? n : 1) * sizeof(T)); }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

#define RT_TRY(call)                        \
    do {                                    \
        if ((call) != hipSuccess) return RT_ERR_HIP; \
    } while (0)

}  // namespace

extern "C" int rt_create_bvh_device(int device, const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles,
                                    size_t n_triangles, uint32_t leaf_max, rt_bvh* nodes, size_t* n_nodes) {
    if (!vertices || !triangles || !n_nodes || n_triangles == 0 || leaf_max == 0) return RT_ERR_INVALID;
    if (n_triangles >= 0x7FFFFFFFull || n_vertices > 0xFFFFFFFFull) return RT_ERR_LIMIT;
    if (!rt::mesh_is_buildable(vertices, n_vertices, triangles, n_triangles)) return RT_ERR_INVALID;
    const uint32_t n = (uint32_t)n_triangles;
    const int n_leaves = (int)n;  // the full tree has one triangle per leaf
    const size_t total = 2 * (size_t)n_leaves - 1;  // upper bound on what survives the collapse
    if (n_leaves > (1 << 29)) return RT_ERR_LIMIT;  // the range search doubles a leaf count in an int
    if (!nodes) {
        *n_nodes = total;
        return RT_OK;
    }

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return RT_ERR_HIP;
    RT_TRY(hipSetDevice(device));
    DevBuf<rt_vertex> d_v;
    DevBuf<rt_tri> d_t, d_sorted;
    DevBuf<float> d_bounds;
    DevBuf<uint64_t> d_keys, d_keys_out;
    DevBuf<uint32_t> d_order, d_order_out, d_left, d_right, d_parent, d_arrived, d_lo, d_hi, d_keep, d_slot;
    DevBuf<rt_bvh> d_full, d_out;
    DevBuf<char> d_tmp, d_tmp2;
    RT_TRY(d_v.alloc(n_vertices));
    RT_TRY(d_t.alloc(n));
    RT_TRY(d_sorted.alloc(n));
    RT_TRY(d_bounds.alloc(6));
    RT_TRY(d_keys.alloc(n));
    RT_TRY(d_keys_out.alloc(n));
    RT_TRY(d_order.alloc(n));
    RT_TRY(d_order_out.alloc(n));
    RT_TRY(d_left.alloc(n_leaves));
    RT_TRY(d_right.alloc(n_leaves));
    RT_TRY(d_lo.alloc(n_leaves));
    RT_TRY(d_hi.alloc(n_leaves));
    RT_TRY(d_parent.alloc(total));
    RT_TRY(d_arrived.alloc(n_leaves));
    RT_TRY(d_keep.alloc(total));
    RT_TRY(d_slot.alloc(total));
    RT_TRY(d_full.alloc(total));
    RT_TRY(hipMemcpy(d_v.p, vertices, n_vertices * sizeof(rt_vertex), hipMemcpyHostToDevice));
    RT_TRY(hipMemcpy(d_t.p, triangles, n * sizeof(rt_tri), hipMemcpyHostToDevice));
    const float inf = __builtin_huge_valf();
    const float init[6] = {inf, inf, inf, -inf, -inf, -inf};
    RT_TRY(hipMemcpy(d_bounds.p, init, sizeof init, hipMemcpyHostToDevice));
    RT_TRY(hipMemset(d_arrived.p, 0, (size_t)n_leaves * sizeof(uint32_t)));

    const dim3 block(kThreads), grid_t((n + kThreads - 1) / kThreads),
        grid_n((unsigned)((total + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(scene_bounds, grid_t, block, 0, 0, d_v.p, d_t.p, n, d_bounds.p);
    hipLaunchKernelGGL(morton_codes, grid_t, block, 0, 0, d_v.p, d_t.p, n, d_bounds.p, d_keys.p, d_order.p);
    RT_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    RT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys.p, d_keys_out.p, d_order.p, d_order_out.p, (int)n, 0,
                                              63));
    RT_TRY(d_tmp.alloc(tmp_bytes));
    RT_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_keys.p, d_keys_out.p, d_order.p, d_order_out.p, (int)n, 0,
                                              63));
    hipLaunchKernelGGL(gather_triangles, grid_t, block, 0, 0, d_t.p, d_order_out.p, n, d_sorted.p);
    if (n_leaves > 1)
        hipLaunchKernelGGL(build_internal, grid_t, block, 0, 0, d_keys_out.p, n_leaves, d_left.p, d_right.p, d_parent.p,
                           d_lo.p, d_hi.p);
    hipLaunchKernelGGL(fit_boxes, grid_t, block, 0, 0, d_v.p, d_sorted.p, n_leaves, d_left.p, d_right.p, d_parent.p,
                       d_arrived.p, d_full.p);
    hipLaunchKernelGGL(mark_kept, grid_n, block, 0, 0, n_leaves, leaf_max, d_parent.p, d_lo.p, d_hi.p, d_keep.p);
    RT_TRY(hipGetLastError());
    size_t tmp2_bytes = 0;
    RT_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2_bytes, d_keep.p, d_slot.p, (int)total));
    RT_TRY(d_tmp2.alloc(tmp2_bytes));
    RT_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp2.p, tmp2_bytes, d_keep.p, d_slot.p, (int)total));
    uint32_t last_keep = 0, last_slot = 0;
    RT_TRY(hipMemcpy(&last_keep, d_keep.p + (total - 1), 4, hipMemcpyDeviceToHost));
    RT_TRY(hipMemcpy(&last_slot, d_slot.p + (total - 1), 4, hipMemcpyDeviceToHost));
    const size_t kept = (size_t)last_slot + last_keep;
    if (*n_nodes < kept) return RT_ERR_INVALID;
    RT_TRY(d_out.alloc(kept));
    hipLaunchKernelGGL(emit_nodes, grid_n, block, 0, 0, n_leaves, leaf_max, d_left.p, d_right.p, d_lo.p, d_hi.p, d_keep.p,
                       d_slot.p, d_full.p, d_out.p);
    RT_TRY(hipGetLastError());
    RT_TRY(hipDeviceSynchronize());
    RT_TRY(hipMemcpy(nodes, d_out.p, kept * sizeof(rt_bvh), hipMemcpyDeviceToHost));
    RT_TRY(hipMemcpy(triangles, d_sorted.p, n * sizeof(rt_tri), hipMemcpyDeviceToHost));
    *n_nodes = kept;
    return RT_OK;
}
