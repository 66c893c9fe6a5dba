/*
 * rtgo_host.h -- small C view of the C++ host surface (librtgo_host.so: Scene / Shape / ShapeFactory / Primitive /
 * SurfaceLight / Renderer, namespace engine::host, mirroring engine/scene.h, shapefactory.h, primitive.h, light.h,
 * renderer.h of the reference).  It exists so that Python drivers (bench.py, tests) can ask the PRODUCT host code for
 * the flattened scene tables and run the headless Renderer; C++ users include the headers under raytracingo_amd/host/ directly.
 */
#ifndef RTGO_HOST_H
#define RTGO_HOST_H

#include "rtgo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Flattened scene exactly as Renderer::CreateShapes / CreateRayGen / CreateMiss / WriteLights hand it to the ABI
   (engine/renderer.cpp:321-336, 386-453, 655-677). */
typedef struct rtgo_host_scene {
    uint32_t n_prims;
    uint32_t n_lights;
    rtgo_prim prims[RTGO_MAX_PRIMS];
    rtgo_aabb aabbs[RTGO_MAX_PRIMS];
    rtgo_light lights[RTGO_MAX_LIGHTS];
    float eye[3], U[3], V[3], W[3];
    float background[3];
} rtgo_host_scene;

/* scene_name: the --scene values of engine/main.cpp:44-52 (plateau, slide, cornell, mirror_spheres, soft_mirrors,
   window, balls, checkered).  returns 0, or RTGO_E_INVALID for an unknown name. */
int rtgo_host_scene_build(const char* scene_name, uint32_t width, uint32_t height, rtgo_host_scene* out);

/* engine::host::materials::<name> (engine/materials.h:13-283) as kd[3], kr[3], Le[3], specularity; RTGO_E_INVALID for an unknown name */
int rtgo_host_material(const char* name, float* out10);

/* Headless engine::host::Renderer run: `frames` progressive frames of scene_name at width x height.
   mode: "path" or "distributed" (engine/main.cpp:83-103); sample = --sample; ambient = --useAmbient.
   host_image (uchar4, may be NULL) / host_accum (float4, may be NULL) receive the last frame; stats may be NULL.
   returns 0 or an RTGO_E_* code; message via rtgo_host_last_error(). */
int rtgo_host_render(const char* scene_name, const char* mode, uint32_t width, uint32_t height, int sample, int ambient,
                     int frames, int device, void* host_image, void* host_accum, rtgo_stats* stats);

/* A scripted interactive session with engine::host::Renderer: what the GLFW callbacks + frame loop of the reference do
   (renderer.cpp:36-145, 679-747, 841-862), without a window.  All return 0 or an RTGO_E_* code. */
/* Headless engine::host::MultiGpuRenderer run (raytracingo_amd/host/multigpu.h): the frame tiled in 4-row bands over
   n_devices GPUs x launches_per_device shares, accumulation bands resident per share, the 8-bit bands gathered to devices[0]
   over RCCL (grouped ncclSend/ncclRecv) and assembled there every present_every-th frame and after the last one.
   rccl_for_local_shares != 0 moves the root GPU's own bands through RCCL as well (one-GPU boxes can then exercise that path).
   host_image / host_accum / stats / ms_per_frame may be NULL.  The assembled frame is bitwise what rtgo_host_render gives. */
int rtgo_host_render_multi(const char* scene_name, const char* mode, uint32_t width, uint32_t height, int sample, int ambient, int frames,
                           const int* devices, int n_devices, int launches_per_device, int present_every, int rccl_for_local_shares,
                           void* host_image, void* host_accum, rtgo_stats* stats, double* ms_per_frame);

typedef struct rtgo_host_session rtgo_host_session;
int rtgo_host_session_open(const char* scene_name, const char* mode, uint32_t width, uint32_t height, int sample, int ambient,
                           int device, rtgo_host_session** out);
int rtgo_host_session_frame(rtgo_host_session* s);                                   /* Update + LaunchFrame */
int rtgo_host_session_move_camera(rtgo_host_session* s, const float eye[3], const float lookat[3], const float up[3]);
int rtgo_host_session_resize(rtgo_host_session* s, uint32_t width, uint32_t height);
int rtgo_host_session_read(rtgo_host_session* s, void* host_image, void* host_accum, uint32_t* frame_count);
int rtgo_host_session_close(rtgo_host_session* s);

const char* rtgo_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
