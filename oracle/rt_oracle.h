/*
 * rt_oracle.h -- CPU restatement of the reference's per-pixel trace path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ray_tracing_amd/ may include, link
 * or call this; it is loaded by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py, always as the checker and never as the thing
 * that is measured or shipped as the product.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference repository holds no tests,
 * golden vectors or fixtures for this path, and its Rust cannot be compiled
 * in this environment (no Rust toolchain).  Fidelity is argued by
 * construction: each function cites the reference lines it restates, and
 * hand-derived known-answer vectors plus an independent NumPy restatement
 * live in tests/.
 *
 * Arithmetic contract: IEEE-754 binary32, round-to-nearest-even, no FMA
 * contraction, `/` and sqrt correctly rounded, glam 0.30.5 operation order
 * (Cargo.lock:710-716).  Build with -O2 -ffp-contract=off -fno-fast-math.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How sin/cos in rand_vec_in_unit_sphere (material.rs:30-31) are evaluated. */
enum {
    ORC_TRIG_LIBM = 0, /* host libm sinf/cosf: what host Rust would call        */
    ORC_TRIG_DET  = 1, /* fixed polynomial, op order shared with the HIP kernel */
    /* OR-ed into the mode word of orc_render_frame: the per-frame seed hashes zeros
     * instead of the previous accumulator (a documented variant used when the samples
     * of a pixel are split over several GPUs; not the reference's sequence) */
    ORC_SEED_INDEPENDENT = 0x100
};

typedef struct orc_scene {
    rt_scene_info info;
    const rt_vertex* vertices;   size_t n_vertices;
    const rt_tri* triangles;     size_t n_triangles;
    const rt_object* objects;    size_t n_objects;
    const rt_instance* instances; size_t n_instances;
    const rt_bvh* bvh;           size_t n_bvh;
    const rt_material* materials; size_t n_materials; /* NULL/0: reference constants */
} orc_scene;

/* Sums over every trace_ray invocation of a render call. */
typedef struct orc_stats {
    uint64_t rays;            /* trace_ray invocations (trace.rs:195)         */
    uint64_t box_tests;       /* hit.rs:80                                     */
    uint64_t internal_visits; /* internal nodes that passed the test (hit.rs:88) */
    uint64_t triangle_tests;  /* hit.rs:121                                    */
    uint64_t paths;           /* pixel samples                                 */
} orc_stats;

/* Optional per-pixel outputs of the FIRST segment of each path.  Any pointer
 * may be NULL.  Arrays are canvas_width*canvas_height long. */
typedef struct orc_probe {
    uint32_t* hit_triangle;   /* HitRecord.triangle_id, u32::MAX on a miss */
    uint32_t* hit_instance;   /* HitRecord.instance_id                     */
    uint32_t* hit_t_bits;     /* HitRecord.t.to_bits()                     */
    uint32_t* box_tests;      /* HitRecord.box_tests                       */
    uint32_t* triangle_tests; /* HitRecord.triangle_tests                  */
} orc_probe;

/* --- unit functions, exported so tests can pin each one ------------------ */
uint32_t orc_get_seed(uint32_t frame, uint32_t x, uint32_t y, float r, float g, float b);
uint32_t orc_xor_shift(uint32_t seed);
float orc_rand_float(uint32_t* seed, float lo, float hi);
float orc_acos_approx(float v);
float orc_det_sin(float x);
float orc_det_cos(float x);
/* ray = {pos[3], dir[3]}; returns t_near or +inf (trace.rs:208-229) */
float orc_hits_bounding(const float pos[3], const float dir[3], const float bmin[3], const float bmax[3]);
float orc_triangle_ray_intersect(const float p0[3], const float p1[3], const float p2[3],
                                 const float pos[3], const float dir[3],
                                 float t_lo, float t_hi, int backface_cull);
void orc_affine_inverse(const rt_affine3a* m, rt_affine3a* out);
void orc_transform_point3(const rt_affine3a* m, const float v[3], float out[3]);
void orc_transform_vector3(const rt_affine3a* m, const float v[3], float out[3]);
/* camera ray for a (jittered) pixel position, NOT yet normalised (trace.rs:25-40) */
void orc_camera_ray(const rt_cam_data* cam, float px, float py, float pos[3], float dir[3]);
void orc_sky(const float dir[3], float out[3]);
void orc_default_material(uint32_t instance_id, rt_material* out);
/* One Material::bxdf call (material.rs) for a world-space ray that hit at distance t on a
 * surface with the given unit normal; colour, ray and seed are updated in place.  Returns the
 * RayReturnState: 0 Absorb, 1 Stop, 2 Ray. */
int orc_bxdf(const rt_material* m, float color[3], float pos[3], float dir[3], const float normal[3], float t,
             uint32_t* seed, int trig_mode);
/* Mesh::hit for one object-space ray; updates the record in place (hit.rs:65-134).
 * rec = {triangle_id, t_bits, instance_id, triangle_tests, box_tests}. */
void orc_mesh_hit(const orc_scene* s, uint32_t bvh_root, const float pos[3], const float dir[3],
                  float t_lo, float t_hi, uint32_t instance_id, int backface_cull, uint32_t rec[5]);
/* One full path for one pixel: Ray::get_color (trace.rs:175-206). */
void orc_get_color(const orc_scene* s, const rt_cam_data* cam, uint32_t x, uint32_t y,
                   uint32_t seed, int trig_mode, float out_rgb[3]);

/* --- whole-frame entry: shader main (shader/src/lib.rs:23-109) ------------
 * Renders rows [y0, y1) of one frame.  `acc` (W*H*4 floats) is read and
 * updated exactly like binding 7; `present` (W*H*4 floats, may be NULL)
 * receives binding 8.  n_threads <= 0 means one thread. Returns 0. */
int orc_render_frame(const orc_scene* s, const rt_cam_data* cam, float* acc, float* present,
                     const orc_probe* probe, orc_stats* stats, uint32_t y0, uint32_t y1,
                     int n_threads, int trig_mode);

#ifdef __cplusplus
}
#endif
#endif
