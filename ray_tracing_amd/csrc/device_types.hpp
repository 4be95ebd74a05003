// Device-side data layout, shared by the host code that fills it and the HIP
// kernels that read it.
//
// The reference keeps the scene as 48-byte nodes, 16-byte vertices and 12-byte
// index triples and reaches a triangle through two dependent loads
// (shader/src/modules/hit.rs:57-60, 77, 90-91).  Here everything a traversal
// step needs sits in ONE 64-byte, 64-byte-aligned record:
//
//   DevNode     one INTERNAL node = the boxes of both children plus what each
//               child is, so a visit is four 16-byte loads from one line and the
//               near/far decision (hit.rs:92-100) needs no further fetch.
//               Leaves have no record of their own: their triangle range lives
//               in the parent's child slot.  The two children are interleaved
//               coordinate by coordinate, so that every step of the slab test
//               handles both of them in one packed (2 x f32) instruction.
//   DevTri      one triangle = its three vertices (already gathered) plus the
//               normalised plane normal and plane offset that hit.rs:145-148
//               recomputes on every test.  The host computes them once with the
//               same operation order, so the bits are identical.
//   DevTriTest  what the persistent kernel's triangle step reads instead.
#pragma once

#include <cstdint>

namespace rt {

constexpr uint32_t kInternalTag = 0xFFFFFFFFu;

// Child k of an internal node (k = 0: the reference's child_1):
//   box  = [lox[k], loy[k], loz[k]] .. [hix[k], hiy[k], hiz[k]]
//   internal child: a[k] = dense index of its DevNode, b[k] = kInternalTag
//   leaf child:     a[k] = first triangle, b[k] = last triangle (inclusive, global);
//                   an empty leaf is the canonical (1, 0)
// A stack entry names a child slot: (node index << 1) | k.
struct alignas(64) DevNode {
    float lox[2], hix[2];
    float loy[2], hiy[2];
    float loz[2], hiz[2];
    uint32_t a[2], b[2];
};
static_assert(sizeof(DevNode) == 64, "one 64-byte line");

struct alignas(64) DevTri {
    float p0[3]; float nx;
    float p1[3]; float ny;
    float p2[3]; float nz;
    float d;
    // The shading normal trace_ray forms at every hit (trace.rs:140-157: the triangle's vertices moved to
    // world space by the instance's transform, normalize(cross(w0 - w1, w0 - w2))), formed once per
    // triangle by the same operations (world_normals_kernel) when one and only one instance shows the
    // triangle's object (DevInstance::normals_ready); unspecified otherwise.
    float wn[3];
};

// What the persistent kernel's triangle step reads: the same plane, the first
// vertex, and the edge functions of trifilter.hpp.  The full DevTri is only
// fetched when that shortcut cannot decide.
struct alignas(64) DevTriTest {
    float n[3];  float d;             // plane, same bits as DevTri's
    float p0[3]; float c1;            // first vertex; constant of the second edge function
    // m0 = n x (p1 - p0) and m1 = n x (p2 - p1), interleaved component by component
    // so that both edge functions are evaluated by packed (2 x f32) instructions
    float m01x[2], m01y[2];
    float m01z[2]; float margin_scale;  // 2^-18 * e_len
    float e_len;                        // bound on the 1-norm of the edges
};
static_assert(sizeof(DevTriTest) == 64, "one 64-byte line");

// The root of an object's BVH in the shape of a node: child 0 is the root,
// child 1 a copy of it that is never visited.
struct alignas(64) DevObject {
    DevNode root;
};

// Four 64-byte lines.  The first is all that "move the ray into this
// instance's space" needs and the second all that "start traversing it" needs,
// so both steps are ordinary 64-byte item fetches.
struct alignas(64) DevInstance {
    float inv[12];   // inverse transform: X, Y, Z, T  (object <- world)
    uint32_t cull;
    uint32_t pad0[3];
    DevNode root;    // copy of the object's root record
    float fwd[12];   // transform:         X, Y, Z, T  (world <- object)
    uint32_t object_id;
    uint32_t mat_kind;
    float ior;
    uint32_t normals_ready;  // 1: DevTri::wn of this instance's triangles holds their world-space shading normal
    float color[3];
    float specular;
    float specular_roughness;
    float roughness;
    // A box in WORLD space that holds the object's root box with room to spare (instance_cull.hpp): a
    // ray that clearly misses it is known to fail the reference's object-space test of the root box,
    // so the instance walk can pass the instance by without moving the ray into its space.
    // bound_ok = 0: no such box could be made (transform without a usable inverse, non-finite box).
    float wlo[3], whi[3];
    uint32_t bound_ok;
    uint32_t pad2[3];
};
static_assert(sizeof(DevInstance) == 256, "four 64-byte lines");

struct DevCam {
    float X[3], Y[3], Z[3], T[3];  // camera-to-world
    float edge_dist;               // tan(fov/2), evaluated once on the host
    float fw, fh, aspect;          // W, H, H/W as f32
    // what every camera ray would compute anew with the same result (trace.rs:35-37): the ray origin
    // transform_point3(0) = ((X*0 + Y*0) + Z*0) + T, formed on the host by those very operations, and the
    // correctly rounded reciprocals of fw and fh with which px / fw and py / fh are formed as exact IEEE
    // quotients in five instructions (div_by_rcp of fastdiv.hpp; fast_px: W and H are in its divisor range)
    float pos[3];
    float rcp_fw, rcp_fh;
    uint32_t fast_px;
    uint32_t W, H;
    uint32_t depth;
    uint32_t frame;
    uint32_t debug_number;
    uint32_t debug_information;
    float frames_without_move;
    uint32_t frame_stride;  // how far `frame` moves between consecutive frames of one call
    uint32_t seed_mode;     // 0: the seed hashes the previous accumulator (lib.rs:57-60); 1: it hashes zeros
};

// Which rows of the canvas this context renders: strips of `strip_h` rows are
// dealt round-robin over `world` contexts and this one takes strip numbers
// congruent to `rank`.
struct DevRows {
    uint32_t strip_h, rank, world, local_rows;
};

// Nodes, triangle test records and instances live in ONE allocation (the
// arena), so that the persistent kernel can name any item by a 32-bit offset
// from one base address: a byte offset while the arena is below 4 GiB, a
// record number (64-byte units) above that.  tris and objects are separate.
struct DevScene {
    const DevNode* nodes;          // = arena + node_off
    const DevTri* tris;
    const DevObject* objects;
    const DevInstance* instances;  // = arena + inst_off
    const char* arena;
    uint32_t node_off, tri_test_off, inst_off;  // byte offsets into the arena (valid while the arena is below 4 GiB)
    uint32_t node_unit, tri_test_unit;          // the same offsets in 64-byte records (any arena below 256 GiB)
    uint32_t n_instances;
    uint32_t fast_div;  // every box coordinate is 0 or has 2^-24 <= |b| <= 2^24 (see fastdiv.hpp)
    uint32_t instance_cull;  // 1: the instance walk first tries DevInstance's world-space box
};

struct DevStats {
    unsigned long long rays, box_tests, internal_visits, triangle_tests, paths;
    // scheduling profile of the persistent kernel (only with counters on):
    // wave iterations that ran a control / box / triangle step,
    // the lanes that took part in them and the cycles they took
    unsigned long long iters[3], lanes[3], cycles[3], wait_cycles[3];
    // cycles waves spent waiting for memory (global, LDS, scalar) at marked places of the steps
    unsigned long long site_wait[8];
    // the pooled kernel's SHADE steps: iterations, slots shaded, cycles
    unsigned long long shade_iters, shade_lanes, shade_cycles;
    // the persistent kernels' own clock: every workgroup's first wave stamps the shader-cycle counter
    // and the constant 100 MHz counter when it starts and when it ends and adds both differences
    // here, so clk_cycles / clk_ticks x 100 MHz is the shader clock the launches ran at
    unsigned long long clk_cycles, clk_ticks;
    // pooled kernel: waves that left their loop with live ray slots because nothing was left to
    // schedule (a scheduler fault: the launch's image is then incomplete)
    unsigned long long pool_stalled;
};

// Work queues of the persistent kernel: 8 ranges of 64-pixel items (8x8 tiles),
// one per XCD group, each with a head counter on a cache line of its own.
struct DevQueue {
    uint32_t* head;          // 8 counters, 32 words apart; zeroed before every launch
    uint32_t first_item[9];  // queue i covers items [first_item[i], first_item[i + 1])
    uint32_t tiles_x;        // 8x8 tiles per row of tiles
    int w_ctl, w_box, w_leaf;  // vote weights of the three kinds of step
    // Cost-aware order (both optional).  `order` permutes the tiles inside each queue's range:
    // position p of the queue stands for tile order[p].  `cost` receives, per tile, the path
    // lengths of this launch; order_tiles_kernel turns them into the next launch's order, dearest
    // tiles first, so that the longest chains of dependent steps start early and the launch does
    // not end on a few lanes finishing them alone.
    const uint32_t* order;
    uint32_t* cost;
    uint32_t deal;  // how order_tiles_kernel shares the sorted tiles out: 1 dealt in turn, 0 a band per queue
    uint32_t pool_slots;  // pooled kernel: ray slots a wave uses, 64..192 (0: all)
};

struct DevFrame {
    float* acc;      // W*H*4
    float* present;  // W*H*4
    uint32_t* hit_triangle;
    uint32_t* hit_instance;
    uint32_t* hit_t_bits;
    uint32_t* box_tests;
    uint32_t* triangle_tests;
    DevStats* stats;
};

}  // namespace rt
