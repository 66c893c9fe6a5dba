// capi_host.cpp -- the C view of the host surface declared in include/rtgo_host.h.
#include <rtgo_host.h>

#include "materials.h"
#include "multigpu.h"
#include "renderer.h"

#include <cstring>
#include <string>

using namespace engine::host;

namespace {
std::string g_error;

bool scene_from_name(const std::string& s, SceneModel& out)
{
    // the --scene table of engine/main.cpp:104-143
    if (s == "plateau") out = SceneModel::PLATE;
    else if (s == "cornell") out = SceneModel::CORNELL;
    else if (s == "slide") out = SceneModel::SLIDE;
    else if (s == "window") out = SceneModel::WINDOW;
    else if (s == "balls") out = SceneModel::BALLS;
    else if (s == "checkered") out = SceneModel::CHECKERED;
    else if (s == "mirror_spheres") out = SceneModel::MIRROR_SPHERES;
    else if (s == "soft_mirrors") out = SceneModel::SOFT_MIRRORS;
    else return false;
    return true;
}
}  // namespace

extern "C" {

const char* rtgo_host_last_error(void) { return g_error.c_str(); }

int rtgo_host_material(const char* name, float* out10)
{
    namespace M = materials;
    static const struct { const char* name; const BasicMaterial* m; } table[] = {
        {"blackMirror", &M::blackMirror}, {"blue", &M::blue}, {"grey", &M::grey}, {"cream", &M::cream}, {"white", &M::white},
        {"mirrorSpheresBlackMirror", &M::mirrorSpheresBlackMirror}, {"mirrorSpheresGroundMat", &M::mirrorSpheresGroundMat},
        {"mirrorSpheresMetallicOrange", &M::mirrorSpheresMetallicOrange}, {"mirrorSpheresSilver", &M::mirrorSpheresSilver},
        {"plateMetallicGold", &M::plateMetallicGold}, {"platePurple", &M::platePurple}, {"plateCyan", &M::plateCyan},
        {"platePrettyGreen", &M::platePrettyGreen}, {"plateDarkRed", &M::plateDarkRed}, {"plateYellow", &M::plateYellow},
        {"plateLight", &M::plateLight}, {"cornellMirror", &M::cornellMirror}, {"cornellWhite", &M::cornellWhite},
        {"cornellBlue", &M::cornellBlue}, {"cornellRed", &M::cornellRed}, {"cornellLight", &M::cornellLight},
        {"softMirrorsMirror0", &M::softMirrorsMirror0}, {"softMirrorsMirror1", &M::softMirrorsMirror1},
        {"softMirrorsMirror2", &M::softMirrorsMirror2}, {"softMirrorsMirror3", &M::softMirrorsMirror3},
        {"softMirrorsMirror4", &M::softMirrorsMirror4}, {"softMirrorsMirror5", &M::softMirrorsMirror5},
        {"softMirrorsMirror6", &M::softMirrorsMirror6}, {"softMirrorsMirror7", &M::softMirrorsMirror7},
        {"CheckeredLight", &M::CheckeredLight}, {"BallsLight", &M::BallsLight}, {"WindowLight", &M::WindowLight},
        {"windowWhite", &M::windowWhite},
    };
    if (!name || !out10) return RTGO_E_INVALID;
    for (const auto& e : table)
        if (std::strcmp(name, e.name) == 0) {
            const glm::vec3 kd = e.m->GetKd(), kr = e.m->GetKr(), le = e.m->GetLe();
            const float v[10] = {kd.x, kd.y, kd.z, kr.x, kr.y, kr.z, le.x, le.y, le.z, e.m->GetSpecularity()};
            std::memcpy(out10, v, sizeof v);
            return RTGO_OK;
        }
    g_error = std::string("rtgo_host_material: unknown material ") + name;
    return RTGO_E_INVALID;
}

int rtgo_host_scene_build(const char* scene_name, uint32_t width, uint32_t height, rtgo_host_scene* out)
{
    SceneModel model;
    if (!scene_name || !out || width == 0 || height == 0 || !scene_from_name(scene_name, model)) {
        g_error = "rtgo_host_scene_build: bad argument or unknown scene";
        return RTGO_E_INVALID;
    }
    std::memset(out, 0, sizeof *out);
    Scene scene(model, width, height);
    uint32_t n = 0;
    for (const std::shared_ptr<Shape>& shape : scene.GetShapes())
        for (const Primitive& p : shape->GetPrimitives()) {
            if (n >= RTGO_MAX_PRIMS) {
                g_error = "rtgo_host_scene_build: too many primitives";
                return RTGO_E_UNSUPPORTED;
            }
            p.CopyToDevice(out->prims[n]);
            out->aabbs[n] = p.GetAabb();
            ++n;
        }
    out->n_prims = n;
    uint32_t nl = 0;
    for (const SurfaceLight& l : scene.GetSurfaceLights()) {
        if (nl >= RTGO_MAX_LIGHTS) break;
        rtgo_light& r = out->lights[nl++];
        const glm::vec3 c = l.GetCorner(), v1 = l.GetV1(), v2 = l.GetV2(), nn = l.GetNormal(), col = l.GetColor();
        const float vals[15] = {c.x, c.y, c.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z, nn.x, nn.y, nn.z, col.x, col.y, col.z};
        std::memcpy(r.corner, vals, sizeof vals);
        r.falloff = l.GetFalloff();
    }
    out->n_lights = nl;
    float3 u, v, w;
    scene.GetCamera()->UVWFrame(u, v, w);
    const float3 e = scene.GetCamera()->eye();
    const float cam[12] = {e.x, e.y, e.z, u.x, u.y, u.z, v.x, v.y, v.z, w.x, w.y, w.z};
    std::memcpy(out->eye, cam, sizeof cam);
    const glm::vec3 bg = scene.GetBackgroundColor();
    out->background[0] = bg.r;
    out->background[1] = bg.g;
    out->background[2] = bg.b;
    return RTGO_OK;
}

int rtgo_host_render(const char* scene_name, const char* mode, uint32_t width, uint32_t height, int sample, int ambient, int frames,
                     int device, void* host_image, void* host_accum, rtgo_stats* stats)
{
    SceneModel model;
    if (!scene_name || !mode || !scene_from_name(scene_name, model) || width == 0 || height == 0 || sample < 1 || frames < 1) {
        g_error = "rtgo_host_render: bad argument";
        return RTGO_E_INVALID;
    }
    RenderMode rm;
    if (std::string(mode) == "path") rm = RenderMode::PATH_TRACING;
    else if (std::string(mode) == "distributed") rm = RenderMode::DISTRIBUTED_RAY_TRACING;
    else {
        g_error = "rtgo_host_render: mode must be path or distributed";
        return RTGO_E_INVALID;
    }
    try {
        auto scene = std::make_shared<Scene>(model, width, height);
        Renderer renderer(scene, rm, sample, ambient != 0);
        renderer.SetDevice(device);
        renderer.SetFrames(frames);
        renderer.Display();
        if (host_image) {
            const std::vector<unsigned char> img = renderer.ReadImage();
            std::memcpy(host_image, img.data(), img.size());
        }
        if (host_accum) {
            const std::vector<float> acc = renderer.ReadAccum();
            std::memcpy(host_accum, acc.data(), acc.size() * sizeof(float));
        }
        if (stats) *stats = renderer.Stats();
    } catch (const std::exception& e) {
        g_error = e.what();
        return RTGO_E_STATE;
    }
    return RTGO_OK;
}

int rtgo_host_render_multi(const char* scene_name, const char* mode, uint32_t width, uint32_t height, int sample, int ambient, int frames,
                           const int* devices, int n_devices, int launches_per_device, int present_every, int rccl_for_local_shares,
                           void* host_image, void* host_accum, rtgo_stats* stats, double* ms_per_frame)
{
    SceneModel model;
    if (!scene_name || !mode || !scene_from_name(scene_name, model) || width == 0 || height == 0 || sample < 1 || frames < 1 || !devices || n_devices < 1) {
        g_error = "rtgo_host_render_multi: bad argument";
        return RTGO_E_INVALID;
    }
    const std::string m(mode);
    if (m != "path" && m != "distributed") {
        g_error = "rtgo_host_render_multi: mode must be path or distributed";
        return RTGO_E_INVALID;
    }
    try {
        auto scene = std::make_shared<Scene>(model, width, height);
        MultiGpuRenderer::Options opt;
        opt.devices.assign(devices, devices + n_devices);
        opt.launchesPerDevice = launches_per_device > 0 ? launches_per_device : 1;
        opt.presentEvery = present_every > 0 ? present_every : 1;
        opt.rcclForLocalShares = rccl_for_local_shares != 0;
        MultiGpuRenderer renderer(scene, m == "path" ? RenderMode::PATH_TRACING : RenderMode::DISTRIBUTED_RAY_TRACING, sample, ambient != 0, opt);
        renderer.SetFrames(frames);
        renderer.Display();
        if (host_image) {
            const std::vector<unsigned char> img = renderer.ReadImage();
            std::memcpy(host_image, img.data(), img.size());
        }
        if (host_accum) {
            const std::vector<float> acc = renderer.ReadAccum();
            std::memcpy(host_accum, acc.data(), acc.size() * sizeof(float));
        }
        if (stats) *stats = renderer.Stats();
        if (ms_per_frame) *ms_per_frame = renderer.LastDisplayMsPerFrame();
    } catch (const std::exception& e) {
        g_error = e.what();
        return RTGO_E_STATE;
    }
    return RTGO_OK;
}

struct rtgo_host_session {
    std::shared_ptr<Scene> scene;
    std::unique_ptr<Renderer> renderer;
};

#define RTGO_SESSION_TRY(body)              \
    try {                                   \
        body;                               \
    } catch (const std::exception& e) {     \
        g_error = e.what();                 \
        return RTGO_E_STATE;                \
    }                                       \
    return RTGO_OK;

int rtgo_host_session_open(const char* scene_name, const char* mode, uint32_t width, uint32_t height, int sample, int ambient, int device,
                           rtgo_host_session** out)
{
    SceneModel model;
    if (!out || !scene_name || !mode || !scene_from_name(scene_name, model) || width == 0 || height == 0 || sample < 1) {
        g_error = "rtgo_host_session_open: bad argument";
        return RTGO_E_INVALID;
    }
    const std::string m(mode);
    if (m != "path" && m != "distributed") {
        g_error = "rtgo_host_session_open: mode must be path or distributed";
        return RTGO_E_INVALID;
    }
    *out = nullptr;
    RTGO_SESSION_TRY({
        auto* s = new rtgo_host_session();
        s->scene = std::make_shared<Scene>(model, width, height);
        s->renderer.reset(new Renderer(s->scene, m == "path" ? RenderMode::PATH_TRACING : RenderMode::DISTRIBUTED_RAY_TRACING, sample, ambient != 0));
        s->renderer->SetDevice(device);
        *out = s;
    })
}

int rtgo_host_session_frame(rtgo_host_session* s)
{
    if (!s) return RTGO_E_INVALID;
    RTGO_SESSION_TRY(s->renderer->RenderFrame())
}

int rtgo_host_session_move_camera(rtgo_host_session* s, const float eye[3], const float lookat[3], const float up[3])
{
    if (!s || !eye || !lookat || !up) return RTGO_E_INVALID;
    RTGO_SESSION_TRY(s->renderer->MoveCamera(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                                             make_float3(up[0], up[1], up[2])))
}

int rtgo_host_session_resize(rtgo_host_session* s, uint32_t width, uint32_t height)
{
    if (!s) return RTGO_E_INVALID;
    RTGO_SESSION_TRY(s->renderer->Resize(width, height))
}

int rtgo_host_session_read(rtgo_host_session* s, void* host_image, void* host_accum, uint32_t* frame_count)
{
    if (!s) return RTGO_E_INVALID;
    RTGO_SESSION_TRY({
        if (host_image) {
            const std::vector<unsigned char> img = s->renderer->ReadImage();
            std::memcpy(host_image, img.data(), img.size());
        }
        if (host_accum) {
            const std::vector<float> acc = s->renderer->ReadAccum();
            std::memcpy(host_accum, acc.data(), acc.size() * sizeof(float));
        }
        if (frame_count) *frame_count = s->renderer->FrameCount();
    })
}

int rtgo_host_session_close(rtgo_host_session* s)
{
    delete s;
    return RTGO_OK;
}

}  // extern "C"
