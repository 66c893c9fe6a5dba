/*
 * rtgo_oracle.h -- CPU ORACLE for the RayTracinGO hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's device pipeline (engine/kernel.cu) and of the host
 * feeders that define its inputs (engine/scene.cpp, shapefactory.cpp, primitive.cpp, light.cpp,
 * sutil/Matrix.h, sutil/Camera.cpp, cuda/random.h).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product path (raytracingo_amd/) never does.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - RNG (tea<16>/lcg/rnd), Matrix4x4 ops, vec helpers, Camera::UVWFrame, glm light-normal:
 *     pinned bit-exactly against the reference's own headers compiled here (oracle/_ref, golden
 *     vectors in tests/golden/ref_*.json).
 *   - Scene tables and the render itself: every engine .cpp file and kernel.cu include <optix.h>, which this
 *     image lacks, so they are UNBUILDABLE here and the reference ships no test/golden vector for
 *     them: PARITY UNPINNED for those, except for the known answers SURVEY.md section 6 recorded
 *     (primitive counts, ray counts by depth, C1 image means), which tests/ check.
 *
 * All arithmetic is IEEE float32, compiled with -ffp-contract=off; every function cites the
 * reference lines it restates.
 */
#ifndef RTGO_ORACLE_H
#define RTGO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_PRIMS 512
#define ORACLE_MAX_LIGHTS 10 /* params.h:115 */

/* PRIMITIVE_TYPE order of engine/primitive.h:16-22 */
enum { ORACLE_CYLINDER = 0, ORACLE_DISK = 1, ORACLE_RECTANGLE = 2, ORACLE_SPHERE = 3 };

/* type + HitGroupData (params.h:103-110): model matrix (row-major) + BasicMaterial (params.h:61-71) */
typedef struct {
    int32_t type;
    float M[16];
    float kd[3];
    float kr[3];
    float specularity;
    float Le[3];
} oracle_prim;

/* SurfaceLight, params.h:73-87 */
typedef struct {
    float corner[3], v1[3], v2[3], normal[3], color[3];
    float falloff;
} oracle_light;

typedef struct {
    int32_t n_prims;
    int32_t n_lights;
    oracle_prim prims[ORACLE_MAX_PRIMS];
    float aabb[ORACLE_MAX_PRIMS][6]; /* OptixAabb order: minX,minY,minZ,maxX,maxY,maxZ (primitive.cpp:100-115) */
    oracle_light lights[ORACLE_MAX_LIGHTS];
    float eye[3], U[3], V[3], W[3]; /* CameraData, params.h:89-95 */
    float bg[3];                    /* MissData, params.h:97-101 */
} oracle_scene;

/* One launch (= Params of params.h:112-140 + the tile window the MI355X path adds). */
typedef struct {
    uint32_t width, height;  /* FULL image size: seeds and ray directions use these (kernel.cu:187-217) */
    int32_t sqrt_spp;        /* Params::sqrtSamplePerPixel */
    int32_t max_depth;       /* Params::maxTraceDepth (5, renderer.cpp:616) */
    uint32_t frame_count;    /* Params::frameCount */
    int32_t path_tracing;    /* Params::enablePathTracing */
    int32_t use_ambient;     /* Params::useAmbientLight */
    /* window of the full image rendered by this call, global pixel coordinates */
    uint32_t x0, y0, w, h;
    /* row-band interleave inside the window: this call renders window rows r with (r / band_h) % n_ranks == rank.
       Output buffers are compact: local row index counts only owned rows. n_ranks = 1 -> all rows. */
    uint32_t band_h, n_ranks, rank;
    int32_t mode;    /* 0 = literal (brute force in SBT order, inverse() per test, recursion); 1 = canonical LBVH, hoisted inverse */
    int32_t threads; /* OpenMP threads (<=0: all) */
    /* 0 (canonical): sin/cos in GetRayOnHemisphere are the float overloads, as in CUDA device code (kernel.cu:118).
       1: they are evaluated in double and rounded, which is what a HOST C++ build of kernel.cu does with libstdc++
          (::sin(double) is the only global overload) -- the build SURVEY.md section 6 took its ray counts from.
          Used only by the known-answer tests that reproduce those counts exactly. */
    int32_t host_double_trig;
} oracle_frame;

typedef struct {
    uint64_t rays_radiance[8]; /* by depth 0..5 */
    uint64_t rays_occlusion;
    uint64_t rays_total;
    uint64_t node_visits; /* V: LBVH node records fetched+tested (mode 1) */
    uint64_t prim_tests;  /* T: intersector calls */
    uint64_t hits;        /* rays whose closest-hit program ran */
} oracle_counters;

/* canonical LBVH node (32 B): AABB 24 B + two links. internal nodes [0,n-2], leaves [n-1,2n-2].
   leaf: left = primitive index, right = -1. */
typedef struct {
    float bmin[3];
    float bmax[3];
    int32_t left, right;
} oracle_node;

/* --- building blocks (exported so tests can pin them one by one) --- */
uint32_t oracle_tea16(uint32_t v0, uint32_t v1);      /* cuda/random.h:30-45 */
uint32_t oracle_lcg(uint32_t* prev);                  /* cuda/random.h:48-54 */
float oracle_rnd(uint32_t* prev);                     /* cuda/random.h:63-66 */
void oracle_mat_identity(float* m);
void oracle_mat_mul(const float* a, const float* b, float* out);      /* Matrix.h:344-360 */
void oracle_mat_vec4(const float* m, const float* v, float* out);     /* Matrix.h:472-493 */
void oracle_mat_transpose(const float* m, float* out);                /* Matrix.h:570-577 */
float oracle_mat_det(const float* m);                                 /* Matrix.h:591-608 */
void oracle_mat_inverse(const float* m, float* out);                  /* Matrix.h:612-635 */
void oracle_mat_rotate(float radians, float ax, float ay, float az, float* out); /* Matrix.h:640-672 */
void oracle_mat_translate(float x, float y, float z, float* out);     /* Matrix.h:677-688 */
void oracle_mat_scale(float x, float y, float z, float* out);         /* Matrix.h:691-702 */
void oracle_normalize3(const float* v, float* out);                   /* vec_math.h:541-545 */
void oracle_camera_uvw(const float* eye, const float* lookat, const float* up, float fovy, float aspect,
                       float* U, float* V, float* W);                 /* Camera.cpp:34-45 */
void oracle_prim_aabb(const float* M, float* aabb6);                  /* primitive.cpp:35-79,100-115 */
void oracle_light_from_matrix(const float* M, const float* color, float falloff, oracle_light* out); /* light.cpp:9-28 */
/* intersection programs, kernel.cu:250-416. returns 1 and fills t, n[3] (world-space, un-normalised) when the
   program would call optixReportIntersection (before the [tmin,tmax] acceptance test). */
int oracle_intersect(const oracle_prim* p, const float* o, const float* d, float* t, float* n);
/* GetRayOnHemisphere, kernel.cu:101-122 */
void oracle_hemisphere(const float* normal, const float* direction, float coef, uint32_t* seed, float* out);

/* --- scenes: scene.cpp:29-671 + renderer.cpp:321-336,386-453,655-677 flattening. returns 0 or -1 (unknown name).
   names are the CLI names of main.cpp:44-52. */
int oracle_scene_create(const char* name, uint32_t width, uint32_t height, oracle_scene* out);
int oracle_material(const char* name, float* out10);           /* engine/materials.h:13-283 by name: kd, kr, Le, specularity */

/* --- canonical LBVH: nodes must hold 2n-1 entries; returns root index (0, or 0 for n==1 leaf) */
int oracle_lbvh_build(const float (*aabb)[6], int n, oracle_node* nodes, uint32_t* morton_sorted, int32_t* prim_sorted);

/* --- the render: __raygen__rg over the window/bands of fr. accum = float4 per local pixel, image = uchar4.
   accum is read when frame_count > 0 (kernel.cu:239-244). counters may be NULL. returns 0. */
int oracle_render(const oracle_scene* sc, const oracle_frame* fr, float* accum, uint8_t* image, oracle_counters* ctr);

/* number of local rows a (window h, band_h, n_ranks, rank) owns */
uint32_t oracle_local_rows(uint32_t h, uint32_t band_h, uint32_t n_ranks, uint32_t rank);

/* ---- the "whitted" triangle path (rtgo_oracle_whitted.c): cuda/whitted.cu + cuda/LocalGeometry.h ---- */
typedef struct { const uint8_t* px; uint32_t w, h; } oracle_tex;                /* RGBA8, row 0 first; px NULL: none (sutil/Scene.cpp:505-538) */
typedef struct { oracle_tex base_color, metallic_roughness, normal; } oracle_mat_tex;   /* MaterialData::Pbr's three handles */
typedef struct { float base_color[4]; float metallic, roughness; } oracle_pbr;                               /* MaterialData::Pbr */
typedef struct { float color[3]; float intensity; float position[3]; int32_t falloff; } oracle_point_light;   /* Light::Point */
typedef struct {
    const float* positions;          /* 3 per vertex */
    const float* normals;            /* 3 per vertex or NULL */
    const uint32_t* indices;         /* 3 per triangle */
    const uint32_t* tri_material;    /* per triangle or NULL */
    const oracle_pbr* materials;
    const oracle_point_light* lights;
    uint32_t n_vertices, n_triangles, n_materials, n_lights;
    float eye[3], U[3], V[3], W[3], miss[3];
    const float* texcoords;          /* 2 per vertex or NULL (UV = barycentrics, LocalGeometry.h:97-102) */
    const oracle_mat_tex* mat_tex;   /* per material or NULL */
} oracle_whitted_scene;
void oracle_tex2d(const oracle_tex* t, float u, float v, float* rgba);                                        /* tex2D<float4>: wrap, bilinear, 1.8 fixed-point weights */
uint32_t oracle_tea4(uint32_t v0, uint32_t v1);                                                               /* random.h:30-45, N = 4 */
int oracle_tri_intersect(const float* p0, const float* p1, const float* p2, const float* o, const float* d, float tmin, float tmax,
                         float* t, float* u, float* v);
int oracle_whitted_render(const oracle_whitted_scene* s, uint32_t width, uint32_t height, uint32_t subframe, float* accum, uint8_t* image,
                          uint64_t* rays, int threads);

#ifdef __cplusplus
}
#endif
#endif
