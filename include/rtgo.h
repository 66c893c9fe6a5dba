/*
 * rtgo.h -- C ABI of librtgo_hip.so: the MI355X (gfx950) replacement for the OptiX-7 host API as RayTracinGO's
 * Renderer uses it.  Plain pointers and sizes only; no torch, HIP or C++ types cross this boundary.
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference tree).  The way a
 * reference maintainer binds them from engine/renderer.cpp is shown in INTEGRATION.md.
 *
 * Conventions:
 *   - every function returns 0 on success, otherwise a non-zero code (RTGO_E_*; HIP errors are passed through as
 *     RTGO_E_HIP_BASE + hipError_t).  rtgo_last_error() gives the text.  The C++ Renderer re-throws as
 *     std::runtime_error, which is what OPTIX_CHECK / CUDA_CHECK do in the reference (sutil/Exception.h:93-157).
 *   - the context owns all device memory it allocates; host pointers passed in are copied before the call returns.
 *   - a context is not thread-safe; one context per GPU (the reference is single-threaded, renderer.cpp:213-215).
 *   - rtgo_launch is asynchronous on the context's stream; rtgo_sync blocks (optixLaunch + cudaStreamSynchronize,
 *     renderer.cpp:761-773, CUDAOutputBuffer.h:247-250).
 *   - there is NO CPU fallback: without a gfx950 device rtgo_create fails loudly.
 */
#ifndef RTGO_H
#define RTGO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGO_ABI_VERSION 6
#define RTGO_MAX_PRIMS 512  /* scene staged whole in LDS (largest reference scene: checkered, 390) */
#define RTGO_MAX_LIGHTS 10  /* Params::MAX_LIGHTS, engine/params.h:115 */

enum {
    RTGO_OK = 0,
    RTGO_E_INVALID = 1,      /* bad argument */
    RTGO_E_NO_DEVICE = 2,    /* no HIP device / not gfx950 */
    RTGO_E_STATE = 3,        /* call order: scene / camera / output missing */
    RTGO_E_UNSUPPORTED = 4,  /* scene outside the LDS-resident design limits */
    RTGO_E_HIP_BASE = 1000   /* + hipError_t */
};

/* engine/primitive.h:16-22 PRIMITIVE_TYPE (selects the intersection program, primitive.cpp:12-15, 81-98) */
enum { RTGO_CYLINDER = 0, RTGO_DISK = 1, RTGO_RECTANGLE = 2, RTGO_SPHERE = 3 };

/* One hit-group record: PRIMITIVE_TYPE + device::HitGroupData (engine/params.h:103-110) =
   row-major model matrix + device::BasicMaterial (params.h:61-71).  Replaces HitGroupSbtRecord (params.h:143-155). */
typedef struct rtgo_prim {
    uint32_t type;
    float model[16];
    float kd[3];
    float kr[3];
    float specularity;
    float Le[3];
} rtgo_prim;

/* OptixAabb as filled by Primitive::GetAabb (engine/primitive.cpp:100-115) */
typedef struct rtgo_aabb {
    float minX, minY, minZ, maxX, maxY, maxZ;
} rtgo_aabb;

/* device::SurfaceLight (engine/params.h:73-87) as written by Renderer::WriteLights (renderer.cpp:655-677) */
typedef struct rtgo_light {
    float corner[3];
    float v1[3];
    float v2[3];
    float normal[3];
    float color[3];
    float falloff;
} rtgo_light;

/* Per-launch constants: device::Params (engine/params.h:112-140) minus the pointers, plus the framebuffer window
   this GPU renders (multi-GPU tiling).  Pixel seeds and ray directions always use the FULL
   image_width/image_height (kernel.cu:187-217), so any window/band split is bitwise identical to a whole-image launch. */
typedef struct rtgo_frame {
    uint32_t image_width;      /* Params::image_width  (full image) */
    uint32_t image_height;     /* Params::image_height (full image) */
    int32_t sqrt_spp;          /* Params::sqrtSamplePerPixel (--sample=N, N*N spp) */
    int32_t max_trace_depth;   /* Params::maxTraceDepth (5: renderer.cpp:616) */
    uint32_t frame_count;      /* Params::frameCount */
    uint32_t path_tracing;     /* Params::enablePathTracing */
    uint32_t use_ambient;      /* Params::useAmbientLight */
    uint32_t x0, y0, w, h;     /* window in global pixel coordinates; w = h = 0 means the full image */
    uint32_t band_h;           /* row-band height of the interleave inside the window (0 = 4) */
    uint32_t n_ranks, rank;    /* this context renders window rows r with (r / band_h) % n_ranks == rank; 0/1 = all */
    uint32_t collect_stats;    /* 1: also count LBVH node visits / primitive tests / hits over every ray (slower instrumented kernel
                                  on the canonical LBVH, every pixel traced); 2: the same counters over the TRAVERSED rays only
                                  (pixels outside the scene's screen rectangle are answered without a walk, as in timed launches) */
    uint32_t reserve_cus;      /* leave this many CUs' worth of workgroup slots to other streams (the multi-GPU driver's RCCL
                                  gather overlaps the next frame's kernel; persistent workgroups would otherwise hold every CU) */
} rtgo_frame;

typedef struct rtgo_stats {
    uint64_t rays_total;      /* optixTrace equivalents: primary + bounce + shadow (kernel.cu:63-75) */
    uint64_t rays_occlusion;  /* RAY_TYPE_OCCLUSION rays among them */
    uint64_t node_visits;     /* collect_stats only: 32-byte LBVH node records fetched and box-tested */
    uint64_t prim_tests;      /* collect_stats only: intersection-program invocations */
    uint64_t hits;            /* collect_stats only: rays whose closest-hit program ran */
    float last_launch_ms;     /* HIP-event time of the last megakernel launch (valid after rtgo_sync) */
    float total_launch_ms;    /* sum over launches since rtgo_reset_stats */
    uint32_t launches;
    uint32_t lbvh_depth;      /* depth of the on-device LBVH */
    uint64_t dbg_fast_boxes;  /* diagnostic builds (-DRTGO_FAST_COUNTERS) only: boxes tested by the fast walk, else 0 */
    uint64_t dbg_fast_tests;  /* diagnostic builds only: leaf tests of the fast walk incl. the up-front list, else 0 */
    uint64_t rays_culled;     /* primary rays among rays_total that were answered (as misses) by the screen rectangle of the
                                 scene's bounds instead of a traversal; always 0 for collect_stats launches */
    uint32_t launches_canonical; /* launches since rtgo_reset_stats that walked the canonical LBVH: collect_stats launches, and
                                    launches beyond the far-field guard (guard_reach / guard_quadric below; several times slower;
                                    DESIGN.md 3.2) */
    uint32_t cuboid_groups;   /* scene property: groups of three rectangle pairs the build certified as the faces of one box or room
                                 (tested by the fast walk's cuboid test); the up-front list's counts as one */
    float guard_reach;        /* last launch: max(|scene bounds|, |eye|), world units -- the far-field guard's first quantity */
    float guard_quadric;      /* last launch: max over spheres / cylinders of D^2 smax / smin^2 (D: farthest ray origin -- eye or scene
                                 bounds -- to the primitive; s: its axis scales) -- the guard's second quantity; 0 without quadrics */
    uint32_t last_variant;    /* last launch: bit 0 streaming loop, bit 1 the second fast-walk structure, bit 2 canonical walk, bit 3 the
                                 launch was one of the launch-time trial's (the first launches of a job time the candidate
                                 (loop, structure) pairs -- same pixels either way -- and the fastest keeps the job) */
    uint32_t launches_trial;  /* launches since rtgo_reset_stats that were trial launches */
} rtgo_stats;

typedef struct rtgo_ctx rtgo_ctx;

/* optixInit + cudaSetDevice + optixDeviceContextCreate + module/pipeline creation
   (renderer.cpp:194-273, 613-634).  Fails with RTGO_E_NO_DEVICE when there is no gfx950 GPU. */
int rtgo_create(int device, rtgo_ctx** out);

/* Renderer::CleanUp (renderer.cpp:870-885) */
int rtgo_destroy(rtgo_ctx* ctx);

/* text of the last error on this context (ctx == NULL: last error of a failed rtgo_create) */
const char* rtgo_last_error(const rtgo_ctx* ctx);

/* Use an existing HIP stream (hipStream_t passed as void*) instead of the context's own
   (the reference creates its own: renderer.cpp:215).  NULL restores the context's stream. */
int rtgo_set_stream(rtgo_ctx* ctx, void* hip_stream);

/* Renderer::CreateShapes (renderer.cpp:400-453) = optixAccelBuild over n custom-primitive AABBs (:514-611) + the
   hit-group SBT upload (:636-653).  prims[i] is SBT index i (scene shape order, then primitive order).
   aabbs may be NULL: the boxes are then derived on the device with the CubeBox rule of primitive.cpp:35-79.
   Boxes must contain their primitives (the reference's always do: CubeBox bounds the unit cube's image, padded); the timed
   kernel relies on that and culls with its own tighter per-shape boxes.
   Builds M^-1 per primitive and the canonical LBVH on the device.  Synchronous. */
int rtgo_set_scene(rtgo_ctx* ctx, const rtgo_prim* prims, const rtgo_aabb* aabbs, uint32_t n);

/* raygen SBT record: Renderer::CreateRayGen / SyncCameraToSbt (renderer.cpp:321-336, 719-731); device::CameraData */
int rtgo_set_camera(rtgo_ctx* ctx, const float eye[3], const float U[3], const float V[3], const float W[3]);

/* miss SBT record: Renderer::CreateMiss (renderer.cpp:386-398); device::MissData */
int rtgo_set_background(rtgo_ctx* ctx, const float rgb[3]);

/* Renderer::WriteLights (renderer.cpp:655-677); n <= RTGO_MAX_LIGHTS */
int rtgo_set_lights(rtgo_ctx* ctx, const rtgo_light* lights, int n);

/* Allocate context-owned output for `pixels` local pixels: float4 accumulation buffer + uchar4 image
   (cudaMalloc of accum_buffer, renderer.cpp:805-808, 742-746; CUDAOutputBuffer<uchar4>, renderer.cpp:820). */
int rtgo_resize(rtgo_ctx* ctx, size_t pixels);

/* Alternative to rtgo_resize: render into caller-owned DEVICE memory (e.g. buffers the multi-GPU driver hands to
   RCCL): d_accum = float4[pixels], d_image = uchar4[pixels].  The caller keeps ownership. */
int rtgo_bind_output(rtgo_ctx* ctx, void* d_accum, void* d_image, size_t pixels);

/* optixLaunch(pipeline, stream, params, ..., width, height, 1) (renderer.cpp:749-774).  Asynchronous. */
int rtgo_launch(rtgo_ctx* ctx, const rtgo_frame* frame);

/* cudaStreamSynchronize + CUDA_SYNC_CHECK (CUDAOutputBuffer.h:247-250, renderer.cpp:773) */
int rtgo_sync(rtgo_ctx* ctx);

/* D2H of the output (CUDAOutputBuffer::getHostPointer, CUDAOutputBuffer.h:270-292).  bytes = pixels*4 / pixels*16. */
int rtgo_read_image(rtgo_ctx* ctx, void* host_uchar4, size_t bytes);
int rtgo_read_accum(rtgo_ctx* ctx, void* host_float4, size_t bytes);
/* H2D of an accumulation buffer (resume a progressive render; the reference never saves it: SURVEY section 5) */
int rtgo_write_accum(rtgo_ctx* ctx, const void* host_float4, size_t bytes);

/* ray counters and launch timing (no reference counterpart: the reference only shows an FPS overlay) */
int rtgo_get_stats(rtgo_ctx* ctx, rtgo_stats* out);
int rtgo_reset_stats(rtgo_ctx* ctx);

/* Debug/test access to what rtgo_set_scene built: nodes = 2n-1 records of 32 bytes
   {float bmin[3]; int32 left; float bmax[3]; int32 right} ... see DESIGN.md; inverses = n x 12 floats (rows 0..2). */
int rtgo_read_bvh(rtgo_ctx* ctx, void* host_nodes, size_t node_bytes, void* host_inverses, size_t inv_bytes,
                  void* host_aabbs, size_t aabb_bytes);

/* Multi-GPU presentation step (no reference counterpart: the reference is single-GPU).  d_gathered holds
   n_ranks compact band buffers back to back, rows_pad rows of w elements each (rank g's k-th owned row is row g*rows_pad + k:
   what the ranks' rtgo_bind_output buffers look like after a gather to one root); d_full receives window rows 0..h-1.
   elem_bytes = 4 (uchar4 image) or 16 (float4 accumulation).  Asynchronous on hip_stream (hipStream_t as void*; NULL = the
   context's stream); both pointers are device memory of ctx's device. */
int rtgo_assemble_bands(rtgo_ctx* ctx, void* hip_stream, const void* d_gathered, void* d_full, uint32_t w, uint32_t h,
                        uint32_t band_h, uint32_t n_ranks, uint32_t rows_pad, uint32_t elem_bytes);

/* ---- the "whitted" triangle path: cuda/whitted.cu + the mesh side of sutil/Scene.cpp (no glTF loading: the caller parses its files) ----
   One launch = one subframe of whitted.cu's pipeline: __raygen__pinhole (tea<4> seed, sub-pixel jitter from subframe 1 on, running
   average, gamma-2.2 image), __closesthit__radiance (GGX / Smith / Schlick direct lighting of point lights, one occlusion ray per
   light), __closesthit__occlusion, __miss__constant_radiance.  Camera and output buffers are the context's (rtgo_set_camera,
   rtgo_resize / rtgo_bind_output, rtgo_read_image / rtgo_read_accum). */
#define RTGO_MAX_TRIANGLES 8192

/* MaterialData::Pbr (cuda/MaterialData.h:43-52) without its three texture handles */
typedef struct rtgo_pbr {
    float base_color[4];
    float metallic;
    float roughness;
} rtgo_pbr;

/* Light::Point (cuda/Light.h:47-53) */
typedef struct rtgo_point_light {
    float color[3];
    float intensity;
    float position[3];
    int32_t falloff;   /* Light::Falloff; whitted.cu never reads it */
} rtgo_point_light;

/* sutil::Scene::addMesh + buildMeshAccels (sutil/Scene.cpp): one triangle mesh in world space = GeometryData::TriangleMesh
   (cuda/GeometryData.h:46-52; positions and optional vertex normals, 3 floats per vertex; 32-bit indices, 3 per triangle) with
   one material per triangle (material_of_triangle may be NULL: material 0).  Builds the triangle LBVH on the device.
   n_triangles <= RTGO_MAX_TRIANGLES.  Synchronous. */
int rtgo_whitted_set_mesh(rtgo_ctx* ctx, const float* positions, const float* normals, uint32_t n_vertices, const uint32_t* indices,
                          const uint32_t* material_of_triangle, uint32_t n_triangles, const rtgo_pbr* materials, uint32_t n_materials);

/* GeometryData::TriangleMesh::texcoords (cuda/GeometryData.h:46-52): one (u, v) per vertex of the mesh set by rtgo_whitted_set_mesh, or
   NULL for none -- getLocalGeometry then takes the barycentrics as UV (cuda/LocalGeometry.h:88-102).  Call after rtgo_whitted_set_mesh. */
int rtgo_whitted_set_texcoords(rtgo_ctx* ctx, const float* uv, uint32_t n_vertices);

/* One image of sutil::Scene::addImage + addSampler (sutil/Scene.cpp:478-538): 8-bit RGBA texels in HOST memory, row 0 first (glTF's
   v = 0 is the first row).  Sampled like the reference's cudaTextureObject_t: normalised coordinates, normalised float reads, no sRGB
   decode, and -- addSampler compares its CUDA enum arguments with GL constants, so whatever the glTF sampler says -- wrap addressing and
   bilinear filtering. */
typedef struct rtgo_texture {
    const void* rgba8;
    uint32_t width, height;
} rtgo_texture;

/* MaterialData::Pbr::base_color_tex / metallic_roughness_tex / normal_tex (cuda/MaterialData.h:43-52) of material `material` of the table
   given to rtgo_whitted_set_mesh; NULL = the material has no such texture (whitted.cu:264, 272, 288 test the handle).  The texels are
   copied.  Call after rtgo_whitted_set_mesh (which clears every texture). */
int rtgo_whitted_set_material_textures(rtgo_ctx* ctx, uint32_t material, const rtgo_texture* base_color,
                                       const rtgo_texture* metallic_roughness, const rtgo_texture* normal);

/* whitted::LaunchParams::lights (cuda/whitted.h:71); n <= RTGO_MAX_LIGHTS */
int rtgo_whitted_set_lights(rtgo_ctx* ctx, const rtgo_point_light* lights, uint32_t n);

/* whitted::LaunchParams::miss_color (cuda/whitted.h:72) */
int rtgo_whitted_set_miss_color(rtgo_ctx* ctx, const float rgb[3]);

/* optixLaunch of the whitted pipeline over width x height pixels for subframe `subframe_index` (whitted::LaunchParams,
   cuda/whitted.h:59-74).  Asynchronous on the context's stream; rays are added to rtgo_stats (rays_total, rays_occlusion). */
int rtgo_whitted_launch(rtgo_ctx* ctx, uint32_t width, uint32_t height, uint32_t subframe_index);

/* number of window rows a rank owns under the band interleave (pure host arithmetic) */
uint32_t rtgo_local_rows(uint32_t h, uint32_t band_h, uint32_t n_ranks, uint32_t rank);

uint32_t rtgo_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
