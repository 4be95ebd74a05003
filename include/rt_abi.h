/*
 * rt_abi.h -- byte-exact C mirror of the reference's `shared` crate types.
 *
 * Every struct here is laid out exactly like the Rust type it replaces on an
 * x86-64 host, so that the buffers the reference's `SceneBuilder` produces
 * (program/src/modules/mod.rs:116-133) and the `CamData` its event loop keeps
 * (program/src/main.rs:22-36) can be handed to this library unchanged.
 * The static asserts at the bottom pin every size and offset.
 *
 * Reference declarations:
 *   Vertex        shared/src/lib.rs:112-118   repr(C, align(16))
 *   triangle      shader/src/lib.rs:30        (u32, u32, u32)
 *   BoundingBox   shared/src/lib.rs:143-152   repr(C, align(16))
 *   Bvh           shared/src/lib.rs:169-176   repr(C, align(16))
 *   ChildTriangleMode shared/src/lib.rs:178-183 repr(u32)
 *   Object        shared/src/lib.rs:160-162
 *   Instance      shared/src/lib.rs:164-167
 *   SceneInfo     shared/src/lib.rs:87-93     repr(C)
 *   CamData       shared/src/lib.rs:66-78     repr(C)
 *   DebugInformation shared/src/lib.rs:80-85  (1-byte enum on the host)
 *   glam::Affine3A (glam 0.30.5)              4 x Vec3A, 64 bytes, align 16
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__cplusplus)
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* glam::Affine3A: matrix3 = three column Vec3A (x_axis, y_axis, z_axis), then
 * translation Vec3A.  Each Vec3A is 16 bytes; the 4th lane is padding. */
typedef struct __attribute__((aligned(16))) rt_affine3a {
    float x_axis[3];      float _pad0;
    float y_axis[3];      float _pad1;
    float z_axis[3];      float _pad2;
    float translation[3]; float _pad3;
} rt_affine3a;

/* shared::Vertex */
typedef struct __attribute__((aligned(16))) rt_vertex {
    float pos[3];
    uint8_t _padding[4];
} rt_vertex;

/* one element of the triangle buffer: three GLOBAL vertex indices */
typedef struct rt_tri {
    uint32_t i0, i1, i2;
} rt_tri;

/* shared::BoundingBox */
typedef struct __attribute__((aligned(16))) rt_bounding_box {
    float min[3];
    uint8_t padding_1[4];
    float max[3];
    uint8_t padding_2[4];
} rt_bounding_box;

/* shared::ChildTriangleMode */
enum {
    RT_MODE_CHILDREN  = 0, /* child_1 / child_2 are GLOBAL node indices          */
    RT_MODE_TRIANGLES = 1  /* first / last are GLOBAL triangle indices, INCLUSIVE */
};

/* shared::Bvh */
typedef struct __attribute__((aligned(16))) rt_bvh {
    rt_bounding_box bounding_box;
    uint32_t child_1_or_first_tri;
    uint32_t child_2_or_last_tri;
    uint32_t mode;
    uint32_t _pad;
} rt_bvh;

/* shared::Object */
typedef struct rt_object {
    uint32_t bvh_root; /* GLOBAL node index */
} rt_object;

/* shared::Instance -- instance i's material is selected by i itself
 * (shader/src/modules/trace.rs:84, 150). */
typedef struct __attribute__((aligned(16))) rt_instance {
    rt_affine3a transform;
    uint32_t object_id;
    uint32_t _pad[3];
} rt_instance;

/* shared::SceneInfo */
typedef struct rt_scene_info {
    uint32_t num_instances;
    uint32_t num_bvh_nodes;
    uint32_t num_triangles;
    float sun_orientation[3]; /* carried, never read by the trace path */
} rt_scene_info;

/* shared::DebugInformation */
enum {
    RT_DEBUG_NONE = 0,
    RT_DEBUG_TRIANGLE_INTERSECTION = 1, /* heat-map of triangle tests, trace.rs:97-110 */
    RT_DEBUG_BVH_INTERSECTION = 2       /* heat-map of box tests,      trace.rs:112-125 */
};

/* shared::CamData */
typedef struct __attribute__((aligned(16))) rt_cam_data {
    uint32_t depth;               /* max ray segments per path           */
    uint32_t _pad0[3];
    rt_affine3a transform;        /* camera-to-world                      */
    uint32_t canvas_width;
    uint32_t canvas_height;
    float fov;                    /* degrees, horizontal                  */
    uint32_t frame;
    uint32_t debug_number;
    uint8_t debug_information;    /* RT_DEBUG_*                           */
    uint8_t _pad1[3];
    float frames_without_move;    /* an f32 counter, main.rs:147-148      */
    uint32_t _pad2;
} rt_cam_data;

/* ------------------------------------------------------------------------
 * Material table (extension).  The reference hard-wires three materials and
 * picks one by instance index (trace.rs:15-23, 161-167, 232-244).  This table
 * lifts that choice into data; rt_default_material(i) reproduces the
 * reference's constants, so a scene that never sets a table renders exactly
 * like the reference.
 * ---------------------------------------------------------------------- */
enum {
    RT_MAT_GENERIC  = 0, /* GenericMaterial   material.rs:71-194  */
    RT_MAT_NORMAL   = 1, /* NormalMaterial    material.rs:262-320 */
    RT_MAT_EMISSIVE = 2, /* EmmissiveMaterial material.rs:340-387 */
    /* The reference implements these but never selects them (trace.rs:161-167). */
    RT_MAT_REFRACTIVE = 3, /* RefractiveMaterial material.rs:389-487: uses ior            */
    RT_MAT_UV         = 4, /* UVMaterial         material.rs:489-530: uv is always (0,0),  */
                           /* trace.rs:159, so the path stops with colour * 0              */
    /* These two have get_color / get_next_ray_dir but no `impl Material` in the reference
     * (material.rs:196-260); their bxdf is assembled the way NormalMaterial::bxdf
     * (material.rs:299-319) assembles its own: next ray from get_next_ray_dir at the hit
     * point, colour from get_color. */
    RT_MAT_DIFFUSE    = 5, /* DiffuseMaterial    material.rs:196-225: uses color           */
    RT_MAT_METAL      = 6, /* MetalMaterial      material.rs:227-260: uses color, roughness */
    RT_MAT_KIND_COUNT = 7
};

typedef struct rt_material {
    uint32_t kind;             /* RT_MAT_*                                  */
    uint32_t backface_culling; /* 0/1, Material::backface_culling            */
    float color[3];            /* Generic/Diffuse/Metal: albedo; Emissive: light */
    float specular;
    float specular_roughness;
    float roughness;
    float ior;
    uint32_t _pad;
} rt_material;

RT_STATIC_ASSERT(sizeof(rt_affine3a) == 64, "Affine3A is 64 bytes");
RT_STATIC_ASSERT(offsetof(rt_affine3a, y_axis) == 16, "Affine3A.y_axis");
RT_STATIC_ASSERT(offsetof(rt_affine3a, z_axis) == 32, "Affine3A.z_axis");
RT_STATIC_ASSERT(offsetof(rt_affine3a, translation) == 48, "Affine3A.translation");
RT_STATIC_ASSERT(sizeof(rt_vertex) == 16, "Vertex is 16 bytes");
RT_STATIC_ASSERT(sizeof(rt_tri) == 12, "triangle is 12 bytes");
RT_STATIC_ASSERT(sizeof(rt_bounding_box) == 32, "BoundingBox is 32 bytes");
RT_STATIC_ASSERT(offsetof(rt_bounding_box, max) == 16, "BoundingBox.max");
RT_STATIC_ASSERT(sizeof(rt_bvh) == 48, "Bvh is 48 bytes");
RT_STATIC_ASSERT(offsetof(rt_bvh, child_1_or_first_tri) == 32, "Bvh.child_1");
RT_STATIC_ASSERT(offsetof(rt_bvh, child_2_or_last_tri) == 36, "Bvh.child_2");
RT_STATIC_ASSERT(offsetof(rt_bvh, mode) == 40, "Bvh.mode");
RT_STATIC_ASSERT(sizeof(rt_object) == 4, "Object is 4 bytes");
RT_STATIC_ASSERT(sizeof(rt_instance) == 80, "Instance is 80 bytes");
RT_STATIC_ASSERT(offsetof(rt_instance, object_id) == 64, "Instance.object_id");
RT_STATIC_ASSERT(sizeof(rt_scene_info) == 24, "SceneInfo is 24 bytes");
RT_STATIC_ASSERT(offsetof(rt_scene_info, sun_orientation) == 12, "SceneInfo.sun");
RT_STATIC_ASSERT(sizeof(rt_cam_data) == 112, "CamData is 112 bytes");
RT_STATIC_ASSERT(offsetof(rt_cam_data, transform) == 16, "CamData.transform");
RT_STATIC_ASSERT(offsetof(rt_cam_data, canvas_width) == 80, "CamData.canvas_width");
RT_STATIC_ASSERT(offsetof(rt_cam_data, canvas_height) == 84, "CamData.canvas_height");
RT_STATIC_ASSERT(offsetof(rt_cam_data, fov) == 88, "CamData.fov");
RT_STATIC_ASSERT(offsetof(rt_cam_data, frame) == 92, "CamData.frame");
RT_STATIC_ASSERT(offsetof(rt_cam_data, debug_number) == 96, "CamData.debug_number");
RT_STATIC_ASSERT(offsetof(rt_cam_data, debug_information) == 100, "CamData.debug_information");
RT_STATIC_ASSERT(offsetof(rt_cam_data, frames_without_move) == 104, "CamData.frames_without_move");
RT_STATIC_ASSERT(sizeof(rt_material) == 40, "rt_material is 40 bytes");

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
