/*
 * rt_hip.h -- C ABI of librt_hip.so, the MI355X (gfx950) replacement for the
 * reference's Vulkan compute backend.
 *
 * The reference has no FFI of its own.  Its seam is the Rust-level
 * `vulkan::App` (program/src/modules/vulkan.rs:78-168) and the kernel's
 * descriptor set 0, bindings 0..8 (shader/src/lib.rs:27-42).  Each entry point
 * below names the piece of that seam it stands in for.  All structs are the
 * byte-exact mirrors in rt_abi.h, so a Rust `extern "C"` block can pass
 * `&CamData`, `&SceneInfo` and the `BufferSceneInfo` vectors straight through.
 *
 * Conventions: every function returns 0 (RT_OK) or a negative RT_ERR_* code;
 * rt_last_error() gives the message (the counterpart of the reference's
 * anyhow::Result).  One context per GPU.  Calls on one context are not
 * thread-safe (the reference drives its App from the single winit thread);
 * different contexts may be driven from different host threads or processes.
 * Nothing here falls back to the CPU: without a usable HIP device rt_create
 * fails.  Every call that works on a GPU (a context's, or the `device` argument
 * of the builders and probes) makes that GPU the calling thread's current HIP
 * device and leaves it current.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)  /* bad argument / inconsistent scene buffers    */
#define RT_ERR_HIP (-2)      /* a HIP runtime call failed                     */
#define RT_ERR_NO_SCENE (-3) /* render/read before rt_upload_scene            */
#define RT_ERR_LIMIT (-4)    /* scene exceeds a capacity of this build        */
#define RT_ERR_IO (-5)       /* a file could not be written                   */

typedef struct rt_ctx rt_ctx;

/* Sums over every ray segment (trace_ray invocation, trace.rs:195) since the
 * last rt_reset_stats.  The three traversal sums are only maintained while
 * RT_OPT_COUNTERS is on. */
typedef struct rt_stats {
    uint64_t rays;
    uint64_t box_tests;       /* hit.rs:80  */
    uint64_t internal_visits; /* internal nodes that passed their box test */
    uint64_t triangle_tests;  /* hit.rs:121 */
    uint64_t paths;           /* pixel samples */
} rt_stats;

enum {
    /* Which kernel renders.  All compute the same bits.
     * 0 = plain: one lane per pixel runs nested frame / segment / instance /
     *     traversal loops, the shape of the reference's shader::main;
     * 1 = persistent (default): per-lane state machine (generate, extend,
     *     shade, accumulate as steps), wave-wide vote on the kind of step to
     *     run, lanes refilled from per-XCD work queues;
     * 2 = pooled: the persistent kernel with a pool of ray slots per wave, lanes
     *     only traverse and paths are shaded 64 at a time (slower on every
     *     workload measured; kept as the measured alternative). */
    RT_OPT_PIPELINE = 1,
    /* 1 = also record, for the FIRST segment of every path, the HitRecord
     *     (triangle_id, instance_id, t bits) -> rt_read_hits. */
    RT_OPT_PROBE_HITS = 2,
    /* 1 = maintain the reference's debug counters (hit.rs:80, 121): first-segment
     *     per-pixel values -> rt_read_counters, sums -> rt_get_stats. */
    RT_OPT_COUNTERS = 3,
    /* 0 = skip the gamma pass (binding 8 is then never written). default 1 */
    RT_OPT_PRESENT = 4,
    /* How far CamData.frame advances after each frame of rt_render_frames
     * (default 1, main.rs:147).  With G GPUs splitting the samples of a pixel,
     * GPU g starts at frame g and uses stride G. */
    RT_OPT_FRAME_STRIDE = 5,
    /* 0 (default) = a frame's seed hashes the pixel's previous accumulator value
     *     (lib.rs:57-60), which chains the samples of a pixel;
     * 1 = it hashes zeros instead: get_seed(frame, x, y, 0, 0, 0).  Samples become
     *     independent of each other, which sample-splitting across GPUs needs.
     *     Not the reference's sequence: comparable to it only statistically. */
    RT_OPT_SEED_MODE = 6,
    /* Tuning and test switches of the persistent kernel.  None of them changes a pixel.
     * 1 (default) = hand the 8x8 tiles out in the cost order learnt from the previous launch of
     *     the same canvas; 0 = always in plain order. */
    RT_OPT_TILE_ORDER = 7,
    /* Weights of the wave's vote between control, box and triangle steps: three bytes,
     * control | box << 8 | triangle << 16, each 1..255.  Default 4, 5, 15. */
    RT_OPT_VOTE_WEIGHTS = 8,
    /* Resident workgroups per CU: 0 (default) = as many as fit, else at most this many. */
    RT_OPT_BLOCKS_PER_CU = 9,
    /* 1 = name BVH items by 64-byte record number even when byte offsets would fit (the
     *     addressing scenes of 4 GiB and more get by themselves).  default 0 */
    RT_OPT_FORCE_BIG_OFFSETS = 10,
    /* 0 = form every slab-test quotient by IEEE division instead of multiplying by a cached
     *     correctly rounded reciprocal.  default 1 */
    RT_OPT_FAST_DIV = 11,
    /* How the cost-sorted tiles are shared out over the 8 queues: 1 = dealt in turn, 0 = a
     * contiguous band of the canvas per queue, -1 (default) = dealt for launches of up to four
     * frames, banded for longer ones. */
    RT_OPT_TILE_DEAL = 12,
    /* Pooled kernel (RT_OPT_PIPELINE 2): ray slots each wave keeps in flight, 64..192; 0 (default) = 192. */
    RT_OPT_POOL_SLOTS = 13,
    /* 1 (default) = the persistent kernels pass an instance by when the world-space ray clearly misses a
     *     box stored with the instance, instead of moving the ray into the instance's space to find the
     *     same thing out from the object's root box (trace.rs:76-95, hit.rs:76-86); 0 = always the
     *     reference's way.  Same pixels and counters either way. */
    RT_OPT_INSTANCE_CULL = 14
};

/* ---- lifecycle: vulkan::App::create / App::destroy (vulkan.rs:124-168, 374) ---- */
/* How many GPUs rt_create can name (what the reference learns from enumerate_physical_devices,
 * vulkan.rs:124-168). */
int rt_device_count(int* n);
int rt_create(int device, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);
/* Message of the last failure on this context (ctx == NULL: last failure of
 * rt_create on this thread).  Never NULL. */
const char* rt_last_error(const rt_ctx* ctx);
/* Run this context's work on the given hipStream_t (NULL = the null stream). */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);
int rt_set_option(rt_ctx* ctx, int option, int64_t value);

/* ---- scene: bindings 1-6; replaces create_storage_buffers + the per-frame
 *      refill in update_buffers (vulkan.rs:238-353).  Copies; the caller keeps
 *      ownership.  Uploaded once, resident in HBM until the next upload. ---- */
int rt_upload_scene(rt_ctx* ctx, const rt_scene_info* info,
                    const rt_vertex* vertices, size_t n_vertices,
                    const rt_tri* triangles, size_t n_triangles,
                    const rt_object* objects, size_t n_objects,
                    const rt_instance* instances, size_t n_instances,
                    const rt_bvh* bvh, size_t n_bvh);
/* New transforms (and object ids) for the instances of the uploaded scene: what the reference
 * gets from refilling its instance buffer before every frame (vulkan.rs:238-353), at the cost of
 * one small copy instead of a scene upload.  n must equal the uploaded instance count; materials
 * stay as they are.  Waits for frames in flight on the context's stream. */
int rt_update_instances(rt_ctx* ctx, const rt_instance* instances, size_t n);
/* Per-instance materials (extension, see rt_abi.h).  n must equal the number
 * of instances.  NULL / 0 restores the reference's constants. */
int rt_set_materials(rt_ctx* ctx, const rt_material* materials, size_t n);
void rt_default_material(uint32_t instance_id, rt_material* out);

/* ---- multi-GPU image partition.  This context renders only the strips of
 *      `strip_height` rows whose strip number is congruent to `rank` modulo
 *      `world`.  Default (0, 0, 1): the whole canvas. ---- */
int rt_set_rows(rt_ctx* ctx, uint32_t strip_height, uint32_t rank, uint32_t world);

/* ---- render: App::render minus the swapchain (vulkan.rs:171-236); one
 *      sample per pixel with the accumulate/reset rule of shader/src/lib.rs:85-98.
 *      Asynchronous on the context's stream.  Canvases of at least one and fewer
 *      than 2^31 pixels (RT_ERR_INVALID otherwise) in fewer than 2^26 tiles of 8x8
 *      pixels (RT_ERR_LIMIT).  depth 0 renders black without tracing a ray
 *      (trace.rs:193-205). ---- */
int rt_render_frame(rt_ctx* ctx, const rt_cam_data* cam);
/* n consecutive frames; after each one frame += 1 and frames_without_move += 1.0
 * exactly as the event loop does (main.rs:147-148).  *cam is updated. */
int rt_render_frames(rt_ctx* ctx, rt_cam_data* cam, uint32_t n);
/* No counterpart in the reference.  rt_render_frames for callers that cannot allocate device
 * memory while they render (tight serving loops, harnesses that forbid hipMalloc): this call
 * allocates and frees nothing, on the device or through any library.  It is not a call to put
 * under HIP graph capture as it stands: the camera (frame number, frames_without_move) is a
 * by-value kernel argument, so a replayed graph would render the same frame again, and the call
 * records the timing events behind rt_last_render_ms.  The caller passes the frame buffers, both device memory of canvas_width *
 * canvas_height RGBA floats on the context's GPU: `acc_device` is binding 7 (read and written:
 * zero it before the first frame, lib.rs:57-60) and `present_device` binding 8 (written; NULL
 * skips the gamma pass).  The kernel's own work memory (queue heads, per-tile costs, tile
 * orders) is a fixed 3 MiB device array inside the library, one copy per GPU, shared by every
 * context on that GPU: calls that overlap in time on one GPU race, so keep them on one stream or
 * otherwise ordered.  Limits: the persistent kernel only (RT_OPT_PIPELINE 1, scenes below
 * 256 GiB), no RT_OPT_PROBE_HITS / RT_OPT_COUNTERS (RT_ERR_INVALID), and a canvas of at most
 * 262144 8x8-pixel tiles, about 16.7 M pixels (RT_ERR_LIMIT beyond).  Pixels are bit for bit
 * those of rt_render_frames.  Needs rt_upload_scene first, like every render call. */
int rt_render_frames_noalloc(rt_ctx* ctx, rt_cam_data* cam, uint32_t n, float* acc_device, float* present_device);
int rt_synchronize(rt_ctx* ctx);

/* ---- read-back.  n_pixels must be canvas_width*canvas_height of the last
 *      rendered frame.  Each call synchronises the stream. ---- */
int rt_read_acc(rt_ctx* ctx, float* rgba, size_t n_pixels);     /* binding 7 */
int rt_read_present(rt_ctx* ctx, float* rgba, size_t n_pixels);
/* Binding 8 as the 8-bit image a display takes: R, G, B, A bytes per pixel, every channel clamped
 * to [0, 1], times 255, rounded to nearest, NaN -> 0 -- what the reference's blit of the storage
 * image into its B8G8R8A8 swapchain image does (vulkan.rs:1045-1101), converted on the GPU, so a
 * frame costs 4 bytes per pixel over PCIe instead of 16.  Same bytes as rt_save_png writes. */
int rt_read_present_rgba8(rt_ctx* ctx, uint8_t* rgba8, size_t n_pixels);
/* The same image without stopping the renderer (no counterpart in the reference, whose render waits
 * on its fence every frame, vulkan.rs:171-236): rt_present_begin, called after a render, enqueues the
 * quantisation and a copy into pinned memory of the library's on a stream of its own and returns;
 * the caller may enqueue the next render at once; rt_present_end waits for the copy and hands out
 * the bytes (W * H * 4, valid until the next rt_present_begin, a change of canvas size or
 * rt_destroy).  The copy of frame k then overlaps the render of frame k + 1. */
int rt_present_begin(rt_ctx* ctx);
int rt_present_end(rt_ctx* ctx, const uint8_t** rgba8, size_t* n_pixels); /* binding 8 */
int rt_write_acc(rt_ctx* ctx, const float* rgba, size_t n_pixels, uint32_t width, uint32_t height);
/* Zero binding 7 on the device, as it is right after buffer creation. */
int rt_clear_acc(rt_ctx* ctx);
/* parity probes, no counterpart in the reference's API (its HitRecord and
 * debug counters never leave the shader): first-segment values per pixel */
int rt_read_hits(rt_ctx* ctx, uint32_t* triangle_id, uint32_t* instance_id, uint32_t* t_bits, size_t n_pixels);
int rt_read_counters(rt_ctx* ctx, uint32_t* box_tests, uint32_t* triangle_tests, size_t n_pixels);
int rt_get_stats(rt_ctx* ctx, rt_stats* out);
/* Scheduling profile of the persistent kernel since the last rt_reset_stats
 * (maintained while RT_OPT_COUNTERS is on): per kind of step (0 control,
 * 1 box, 2 triangle) the wave iterations that ran it and the lanes that took
 * part, and the shader-clock cycles the waves spent in them (summed over waves;
 * a wave's cycles include the time its SIMD gave to other waves).
 * lanes / (64 * iterations) is the lane utilisation of that kind. */
typedef struct rt_sched_stats {
    uint64_t iterations[3];
    uint64_t lanes[3];
    uint64_t cycles[3];
    uint64_t wait_cycles[3]; /* the part of `cycles` spent waiting for the step's prefetched item */
    /* the part of `cycles` spent waiting for outstanding memory operations at marked places:
     * 0 control: the fetched pixel's accumulator; 1 control: an instance record (scalar loads);
     * 2 control: what shading loaded; 3 box: the popped stack entry; 4 triangle: who works for
     * whom (LDS); 5 triangle: the owner's ray and the triangle record; 6 triangle: the exact
     * fallback's triangle; 7 triangle: the folded results (LDS atomics) */
    uint64_t site_wait_cycles[8];
    /* the pooled kernel's SHADE steps (RT_OPT_PIPELINE 2): iterations, slots shaded, cycles */
    uint64_t shade_iterations, shade_slots, shade_cycles;
} rt_sched_stats;
int rt_get_sched_stats(rt_ctx* ctx, rt_sched_stats* out);
int rt_reset_stats(rt_ctx* ctx);
/* The shader clock the persistent kernels ran at since the last rt_reset_stats (no counterpart in the
 * reference; what an issue-bound roofline is priced in).  Every workgroup's first wave reads the
 * shader-cycle counter and the constant 100 MHz counter when it starts and when it ends;
 * *shader_cycles and *ticks_100mhz receive the sums of the differences over all workgroups of all
 * launches, so shader_cycles / ticks_100mhz x 100 MHz is the clock, averaged over workgroups.  Both
 * are 0 when only the plain kernel ran.  Synchronises the stream. */
int rt_get_kernel_clock(rt_ctx* ctx, uint64_t* shader_cycles, uint64_t* ticks_100mhz);
/* Device address of the accumulation buffer (canvas W*H*4 f32) of the current
 * canvas size, for callers that reduce it across GPUs themselves. */
int rt_acc_device_ptr(rt_ctx* ctx, uint32_t width, uint32_t height, void** out);
/* Kernel-only time of the last rt_render_frame(s) call in milliseconds,
 * measured with HIP events on the context's stream (synchronises). */
int rt_last_render_ms(rt_ctx* ctx, float* ms);

/* ============ several GPUs of one node as one renderer (no counterpart in the reference) ============
 * The reference drives ONE App on one GPU (program/src/main.rs:129-149, vulkan.rs:124-236).  A pixel
 * sample depends on the scene and on its own accumulator element only (shader/src/lib.rs:57-69), so
 * the work shards with nothing exchanged while it is done: the scene is replicated, a group holds
 * one context per entry of `devices` (the same GPU may be named more than once: a rehearsal on a
 * box with fewer GPUs), each with a stream and a host thread of its own, and a render call returns
 * when every context's work is enqueued.  The assembled image lives on the FIRST context's GPU.
 *   RT_GROUP_TILES    the canvas is dealt to the contexts in interleaved strips of 8 rows
 *                     (rt_set_rows); each renders all n frames of its strips, then its strips are
 *                     copied device to device into the first context's binding 7.  Seeds depend on
 *                     absolute pixel coordinates, so the image is, bit for bit, the one context's.
 *   RT_GROUP_SAMPLES  every context renders the whole canvas for the samples whose running index is
 *                     congruent to its rank (frame numbers keep their meaning: sample k of the
 *                     call uses frame cam->frame + k) with RT_OPT_SEED_MODE 1 -- the reference's
 *                     seed chains a pixel's samples through its accumulator, independent samples
 *                     cannot -- then the accumulators are copied to the first GPU and added there
 *                     IN RANK ORDER, ((a0 + a1) + a2) + ...: one definite value, comparable with
 *                     the reference's sequence only statistically.  cam->frames_without_move < 0.5
 *                     starts a new sum (shader/src/lib.rs:85-87), anything else continues it.
 * Binding 8 is derived once, from the assembled accumulator, on the first GPU (RT_OPT_PRESENT of
 * the first context).  At most 16 contexts (RT_ERR_LIMIT).  Calls on one group are not thread-safe.
 * RT_OPT_FRAME_STRIDE / RT_OPT_SEED_MODE are the group's to set; the parity probes are per context. */
typedef struct rt_scene_builder rt_scene_builder;
typedef struct rt_group rt_group;
#define RT_GROUP_TILES 0
#define RT_GROUP_SAMPLES 1
int rt_group_create(const int* devices, size_t n, rt_group** out);
void rt_group_destroy(rt_group* group);
/* Message of the last failure on this group (group == NULL: of rt_group_create on this thread). */
const char* rt_group_last_error(const rt_group* group);
size_t rt_group_size(const rt_group* group);
/* Context i of the group (NULL when out of range): for per-context options, statistics and
 * rt_present_begin / rt_present_end on context 0, which holds binding 8 of the assembled image. */
rt_ctx* rt_group_context(rt_group* group, size_t i);
/* rt_upload_scene / rt_upload_scene_from_builder / rt_set_materials / rt_update_instances /
 * rt_set_option on every context, the uploads concurrently from the contexts' threads. */
int rt_group_upload_scene(rt_group* group, const rt_scene_info* info,
                          const rt_vertex* vertices, size_t n_vertices,
                          const rt_tri* triangles, size_t n_triangles,
                          const rt_object* objects, size_t n_objects,
                          const rt_instance* instances, size_t n_instances,
                          const rt_bvh* bvh, size_t n_bvh);
int rt_group_upload_scene_from_builder(rt_group* group, rt_scene_builder* b);
int rt_group_set_materials(rt_group* group, const rt_material* materials, size_t n);
int rt_group_update_instances(rt_group* group, const rt_instance* instances, size_t n);
int rt_group_set_option(rt_group* group, int option, int64_t value);
/* n frames in the given mode; *cam advances as rt_render_frames advances it (main.rs:147-148). */
int rt_group_render_frames(rt_group* group, rt_cam_data* cam, uint32_t n, int mode);
int rt_group_synchronize(rt_group* group);
/* The assembled binding 7 / binding 8 (n_pixels = canvas_width * canvas_height of the last call);
 * each synchronises.  rt_group_acc_device_ptr: its address on the first context's GPU. */
int rt_group_read_acc(rt_group* group, float* rgba, size_t n_pixels);
int rt_group_read_present(rt_group* group, float* rgba, size_t n_pixels);
int rt_group_acc_device_ptr(rt_group* group, void** out);

/* ======================= host-side scene assembly ===========================
 * No GPU needed.  Mirrors program/src/modules/mod.rs and bvh.rs. */

/* rt_scene_builder: declared above, with the group entry points */

typedef struct rt_bvh_params {
    uint32_t root_leaf_max; /* bvh.rs:17      default 5  */
    uint32_t leaf_max;      /* bvh.rs:75, 85  default 16 */
    uint32_t max_depth;     /* bvh.rs:5       default 32 */
    int32_t threads;        /* host threads for the build, default 1 */
} rt_bvh_params;
void rt_bvh_params_default(rt_bvh_params* p);

/* parse_obj_file (mod.rs:6-36).  Two-call pattern: pass NULL outputs to get
 * the counts, then buffers of that size. */
int rt_parse_obj(const char* text, size_t len, rt_vertex* vertices, size_t* n_vertices,
                 rt_tri* triangles, size_t* n_triangles);
/* bvh::create_bvh (bvh.rs:7-28): reorders `triangles` in place, writes up to
 * *n_nodes nodes (LOCAL indices) and returns the count in *n_nodes.  Pass
 * nodes == NULL to run the build and get only the count; that call, and a call
 * refused because *n_nodes is too small, leave `triangles` as they were. */
int rt_create_bvh(const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles, size_t n_triangles,
                  const rt_bvh_params* params, rt_bvh* nodes, size_t* n_nodes);

/* A second builder with no counterpart in the reference: a linear BVH (Morton sort,
 * Karras 2012) built on GPU `device`, for meshes where bvh.rs's O(n log^2 n) split
 * search on the host takes too long.  Same contract as rt_create_bvh (reorders
 * `triangles` in place, LOCAL indices, node 0 is the root, NULL `nodes` returns only
 * the count) and the same node format, so the tree can be uploaded, rendered by the
 * reference's shader, or checked by the oracle like any other; it is NOT the tree
 * bvh.rs would build.  Leaves are the subtrees of at most `leaf_max` triangles; with
 * NULL `nodes` the count returned is an upper bound (2 n - 1), the second call returns
 * the actual one.  A tree deeper than rt_upload_scene accepts is only found out there.
 * Boxes are the exact bounds of their triangles with one exception, shared with
 * rt_create_bvh_device_sah: a box thinner than 2^-16 of the mesh's largest extent along an
 * axis is widened by that much on both sides of the axis, because the reference's slab
 * test (trace.rs:225, t_near < t_far) never accepts a box without thickness and an
 * axis-aligned wall in a node of its own would be invisible. */
int rt_create_bvh_device(int device, const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles,
                         size_t n_triangles, uint32_t leaf_max, rt_bvh* nodes, size_t* n_nodes);

/* A third builder, also on the GPU and also without a counterpart in the reference: top-down
 * splits by the reference's own cost function (area(L) * |L| + area(R) * |R|, bvh.rs:109) and
 * its rule "split while a node holds more than leaf_max triangles" (bvh.rs:75-94), the cost
 * evaluated at 15 planes per axis between 16 bins of the node's centroid bounds, all nodes of a
 * level at once while nodes are large; the subtree of a node of 2048 triangles or fewer is finished
 * by one workgroup in LDS (same bins, same cost, same rules: the same subtree).  Same contract and
 * node format as rt_create_bvh_device; the same input always gives the same array.  `max_levels` bounds the number of node levels (0: 33, what the
 * 32-entry traversal stack of the fast kernels takes; at most 65): a split that would leave a
 * side too big for the levels that remain is replaced by a cut in the middle of the range.
 * When leaf_max * 2^(max_levels - 1) < n_triangles no tree of that many levels has leaves that
 * small: the nodes of the last level then stay leaves whatever they hold, as the reference's
 * nodes do at MAX_DEPTH (bvh.rs:75, 85). */
int rt_create_bvh_device_sah(int device, const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles,
                             size_t n_triangles, uint32_t leaf_max, uint32_t max_levels, rt_bvh* nodes,
                             size_t* n_nodes);
/* The same builder with every level taken through the level-wide passes (no subtree is handed to a
 * workgroup): the same tree -- the same boxes, triangle ranges and triangle order at every node --
 * with its nodes numbered level by level throughout.  Slower; kept as the check of the other. */
int rt_create_bvh_device_sah_passes_only(int device, const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles,
                                         size_t n_triangles, uint32_t leaf_max, uint32_t max_levels, rt_bvh* nodes,
                                         size_t* n_nodes);

/* SceneBuilder (mod.rs:38-134) */
rt_scene_builder* rt_scene_builder_new(void);
void rt_scene_builder_free(rt_scene_builder* b);
int rt_scene_builder_add_obj(rt_scene_builder* b, const char* text, size_t len,
                             const rt_affine3a* transforms, size_t n_transforms, const rt_bvh_params* params);
int rt_scene_builder_add_mesh(rt_scene_builder* b, const rt_vertex* vertices, size_t n_vertices,
                              const rt_tri* triangles, size_t n_triangles,
                              const rt_affine3a* transforms, size_t n_transforms, const rt_bvh_params* params);
/* Which builder the meshes added from now on get their tree from: the reference's rule on the host
 * (the default), or one of the GPU builders on `device` (ignored for RT_BVH_HOST), with the leaf
 * size and root rule of the rt_bvh_params passed with the mesh and max_depth + 1 levels.  The
 * add calls then return whatever the GPU builder returns (RT_ERR_HIP without a GPU). */
#define RT_BVH_HOST 0
#define RT_BVH_DEVICE_LINEAR 1
#define RT_BVH_DEVICE_SAH 2
int rt_scene_builder_use_bvh_builder(rt_scene_builder* b, int method, int device);
/* The same for a mesh whose tree is already built -- by rt_create_bvh, by one of the GPU builders
 * or by the reference's own bvh::create_bvh: `triangles` in the tree's order, `nodes` with LOCAL
 * indices and node 0 the root (what SceneBuilder::add_mesh, mod.rs:61-104, does after its call of
 * create_bvh). */
int rt_scene_builder_add_built_mesh(rt_scene_builder* b, const rt_vertex* vertices, size_t n_vertices,
                                    const rt_tri* triangles, size_t n_triangles, const rt_bvh* nodes, size_t n_nodes,
                                    const rt_affine3a* transforms, size_t n_transforms);
void rt_scene_builder_sun_orientation(rt_scene_builder* b, float x, float y, float z);
/* SceneBuilder::build: fills *info and hands out pointers that stay valid
 * until the builder is freed or extended. */
int rt_scene_builder_build(rt_scene_builder* b, rt_scene_info* info,
                           const rt_vertex** vertices, size_t* n_vertices,
                           const rt_tri** triangles, size_t* n_triangles,
                           const rt_object** objects, size_t* n_objects,
                           const rt_instance** instances, size_t* n_instances,
                           const rt_bvh** bvh, size_t* n_bvh);
/* rt_upload_scene with the builder's buffers */
int rt_upload_scene_from_builder(rt_ctx* ctx, rt_scene_builder* b);

/* glam Affine3A::from_scale_rotation_translation (main.rs:23-57); q = x,y,z,w */
void rt_affine_from_srt(const float scale[3], const float quat_xyzw[4], const float translation[3],
                        rt_affine3a* out);
/* glam Quat::from_rotation_x / _y / _z: axis 0/1/2 */
void rt_quat_from_rotation(int axis, float angle, float quat_xyzw[4]);
/* Affine3A::inverse in the operation order the trace path uses (trace.rs:87) */
void rt_affine_inverse(const rt_affine3a* m, rt_affine3a* out);

/* App::update_pos / App::update_mouse (vulkan.rs:95-121) on a CamData.
 * `orientation` is the App's (yaw, pitch) pair, updated in place. */
void rt_cam_update_pos(rt_cam_data* cam, const float delta[3]);
void rt_cam_update_mouse(rt_cam_data* cam, float orientation[2], float mouse_x, float mouse_y);

/* Self-test of the library's "divide by a cached reciprocal" routine, which the
 * slab test uses in place of six IEEE divisions per box: draws n (numerator,
 * divisor) pairs in the routine's operand range from `seed` and returns how many
 * quotients differ in any bit from the IEEE quotient a / d.  Must return 0. */
uint64_t rt_selftest_div_by_rcp(uint64_t n, uint64_t seed);
/* Self-test of the slab test's shortcut: box tests are first decided from
 * approximate quotients (numerator times cached reciprocal) and only fall back to
 * exact quotients when a comparison is not clear-cut.  Draws n (ray, box, clamp)
 * cases from `seed`, many of them with near-ties, and returns how many cases the
 * shortcut called clear-cut yet decided differently from the exact test (hit or
 * miss, and beyond the clamp or not).  Must return 0.  *n_clear receives the number
 * of cases that were called clear-cut. */
uint64_t rt_selftest_slab_filter(uint64_t n, uint64_t seed, uint64_t* n_clear);
/* Self-test of the cull above the instance list (RT_OPT_INSTANCE_CULL): draws n (instance transform,
 * object root box, world ray, clamp) cases from `seed` -- rotated, unevenly scaled, far-translated
 * instances; rays aimed at the rim of the box from any distance, axis-parallel rays, boxes behind the
 * origin -- and returns how many cases the world-space test passed by although the reference's own
 * arithmetic (ray into object space, slab test of the root box, clamp; trace.rs:88-94, 208-229,
 * hit.rs:83) enters the instance.  Must return 0.  *n_passed_by receives the number passed by. */
uint64_t rt_selftest_instance_cull(uint64_t n, uint64_t seed, uint64_t* n_passed_by);
/* Self-test of the triangle test's shortcut: whether the plane hit point lies
 * inside the triangle is first read off three approximate edge functions and the
 * exact cross products are only evaluated when a sign is not clear-cut.  Draws n
 * (triangle, point) cases from `seed`, most of them with the point on or within
 * a few units in the last place of an edge or a vertex, and returns how many
 * cases the shortcut called clear-cut yet decided differently from the exact
 * evaluation of hit.rs:160-188.  Must return 0.  *n_clear receives the number of
 * cases that were called clear-cut. */
uint64_t rt_selftest_tri_filter(uint64_t n, uint64_t seed, uint64_t* n_clear);

/* Device self-test of the one hardware property the instance walk's root-box shortcut
 * assumes: the approximate reciprocal instruction is within 1 unit in the last place of the
 * correctly rounded reciprocal.  Checks every float d with 2^-24 <= |d| <= 2^24 on `device`
 * and stores the largest error found, in units in the last place.  Must be <= 1. */
int rt_selftest_rcp_device(int device, uint32_t* max_ulp_error);

/* Device self-test of the short square root, reciprocal and n/65535 sequences the kernels use
 * in place of the compiler's general expansions: every operand of their ranges (all floats in
 * [2^-100, 2^100], both signs for the reciprocal; n = 0..65535) is evaluated on `device` both
 * ways.  mismatches[0..2] receive how many square roots, reciprocals and quotients differ in
 * any bit.  All three must be 0. */
int rt_selftest_fastmath_device(int device, uint64_t mismatches[3]);

/* Device micro-probe that prices the bound of the trace kernel, which is vector-instruction
 * issue: how many cycles of one SIMD a wave64 vector instruction of a given class takes with
 * `waves_per_simd` (1..8) waves resident on every SIMD of the chip.  Every wave runs an unrolled
 * body of 64 independent instructions of the class for `duration_us` microseconds and counts
 * the bodies.  simd_cycles_per_inst = window x clock x SIMDs / instructions issued by all waves;
 * simds x clock / simd_cycles_per_inst is the chip's issue rate for that class. */
enum {
    RT_VALU_MIX_FMA = 0,        /* v_fma_f32 */
    RT_VALU_MIX_MUL_ADD = 1,    /* v_mul_f32, v_add_f32 */
    RT_VALU_MIX_PK_MUL_ADD = 2, /* v_pk_mul_f32, v_pk_add_f32 */
    RT_VALU_MIX_PK_FMA = 3,     /* v_pk_fma_f32 */
    RT_VALU_MIX_MIN_MAX = 4,    /* v_min_f32, v_max_f32 */
    RT_VALU_MIX_CNDMASK = 5,    /* v_cndmask_b32 on a scalar mask */
    RT_VALU_MIX_CMP = 6,        /* v_cmp_lt_f32 into a scalar mask */
    RT_VALU_MIX_INT = 7,        /* v_add_u32, v_xor_b32, v_and_b32 */
    RT_VALU_MIX_TRANS = 8,      /* v_rcp_f32, v_sqrt_f32 */
    RT_VALU_MIX_BOX_STEP = 9,   /* the proportions of the persistent kernel's BOX step */
    RT_VALU_MIX_MIN3_MAX3 = 10,  /* v_min3_f32, v_max3_f32 */
    RT_VALU_MIX_CNDMASK_VCC = 11, /* v_cndmask_b32 on vcc (the short encoding) */
    RT_VALU_MIX_CMP_VCC = 12,    /* v_cmp_lt_f32 into vcc (the short encoding) */
    RT_VALU_MIX_MOV = 13,        /* v_mov_b32 */
    RT_VALU_MIX_SHIFT_CVT = 14,  /* v_lshlrev_b32, v_cvt_f32_u32 */
    RT_VALU_MIX_MUL_LO = 15,     /* v_mul_lo_u32 */
    RT_VALU_MIX_DPP = 16,        /* v_add_u32 with a DPP row shift (the wave scans) */
    RT_VALU_MIX_BFI = 17,        /* v_bfi_b32: a bitwise select, three vector operands */
    RT_VALU_MIX_AND_OR = 18,     /* v_and_or_b32 */
    /* not vector instructions: what a scalar instruction costs, alone and between vector ones
     * (valu_insts then counts all instructions of the body, scalar ones included) */
    RT_VALU_MIX_SALU = 19,       /* s_add_u32 */
    RT_VALU_MIX_VALU_SALU = 20,  /* v_mul_f32 and s_add_u32 in turn */
    RT_VALU_MIX_SLOW_SALU = 21,  /* v_cndmask_b32 and s_add_u32 in turn */
    RT_VALU_MIX_COUNT = 22
};
typedef struct rt_valu_probe {
    uint64_t valu_insts;        /* vector instructions all waves issued inside their timed windows */
    uint64_t waves;             /* waves launched, all resident at once */
    uint64_t simds;             /* 4 x compute units */
    double window_us;           /* length of a wave's timed window (median) */
    double start_spread_us;     /* between the first and the last wave's start: must be << window_us */
    double clock_mhz;           /* shader clock during the window (median over waves) */
    double kernel_ms;           /* HIP-event time of the launch */
    double wave_share_min;      /* least and most instructions one wave got, relative to an equal share */
    double wave_share_max;
    double simd_cycles_per_inst;
} rt_valu_probe;
int rt_probe_valu_issue(int device, int mix, uint32_t waves_per_simd, uint32_t duration_us, rt_valu_probe* out);

/* Image files, in place of the reference's copy of binding 8 into the swapchain
 * (vulkan.rs:1045-1101).  Both take a host buffer of width*height RGBA floats, as
 * rt_read_present / rt_read_acc fill it, row 0 at the top.
 * rt_save_png: 8-bit RGB of the displayed image (values clamped to [0,1], NaN -> 0),
 *   with a gAMA chunk for the 1/2.2 exponent of shader/src/lib.rs:101-105.
 * rt_save_exr: 32-bit float R,G,B scan lines, uncompressed; every value is multiplied
 *   by `scale` (1 / (frames_without_move + 1) turns the accumulator into mean radiance,
 *   shader/src/lib.rs:99-100). */
int rt_save_png(const char* path, const float* rgba, uint32_t width, uint32_t height);
int rt_save_exr(const char* path, const float* rgba, uint32_t width, uint32_t height, float scale);

const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
