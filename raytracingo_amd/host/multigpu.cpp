// multigpu.cpp -- MultiGpuRenderer: band-interleaved shares over several GPUs, RCCL gather of the presented frame to the root.
// Host-side HIP runtime calls (streams, events, buffers) and RCCL only, through gpu_transport.h; every pixel is computed by
// librtgo_hip.so's megakernel.
#include "multigpu.h"

#include "gpu_transport.h"

#include <chrono>
#include <iostream>
#include <set>
#include <stdexcept>

namespace engine {
namespace host {


struct MultiGpuRenderer::Gpu {
    int device = 0;
    int commRank = -1;            // rank in the RCCL communicator (one rank per GPU)
    void* commStream = nullptr;   // hipStream_t
};

struct MultiGpuRenderer::Share {
    int gpu = 0;                  // index into m_gpus
    rtgo_ctx* ctx = nullptr;
    void* stream = nullptr;       // hipStream_t
    void* accum = nullptr;        // float4[rowsPad * W], resident for the whole progressive render
    void* image[2] = {nullptr, nullptr};   // uchar4 bands, double-buffered: the gather of frame f overlaps the kernel of frame f+1
    void* rendered[2] = {nullptr, nullptr};   // hipEvent_t: the frame's kernel on this share has completed
    void* gathered[2] = {nullptr, nullptr};   // hipEvent_t: the gather that read image[b] has completed
    bool gatherPending[2] = {false, false};
};

MultiGpuRenderer::MultiGpuRenderer(std::shared_ptr<Scene> scene, RenderMode renderMode, int sqrtSamplePerPixel, bool useAmbientCoeff,
                                   const Options& options)
    : m_scene(scene), m_renderMode(renderMode), m_useAmbient(useAmbientCoeff), m_sqrtSpp(sqrtSamplePerPixel), m_opt(options),
      m_width(scene->GetCameraWidth()), m_height(scene->GetCameraHeight()), m_rowsPad(0), m_frameCount(0), m_first(true), m_frames(1),
      m_lastBuffer(0), m_msPerFrame(0.0), m_comms(nullptr), m_fullImage(nullptr), m_gatherImage(nullptr), m_fullAccum(nullptr), m_gatherAccum(nullptr)
{
    if (m_opt.devices.empty()) m_opt.devices.push_back(0);
    if (m_opt.launchesPerDevice < 1 || m_opt.launchesPerDevice > 2) throw std::invalid_argument("MultiGpuRenderer: launchesPerDevice must be 1 or 2");
    if (m_opt.bandHeight == 0) m_opt.bandHeight = 4;
    if (m_opt.presentEvery < 1) m_opt.presentEvery = 1;
    if (std::set<int>(m_opt.devices.begin(), m_opt.devices.end()).size() != m_opt.devices.size())
        throw std::invalid_argument("MultiGpuRenderer: a device is listed twice (use launchesPerDevice for two shares on one GPU)");
    try {
        const int nGpus = static_cast<int>(m_opt.devices.size());
        const int V = nGpus * m_opt.launchesPerDevice;
        for (int v = 0; v < V; ++v) {
            const unsigned int rows = rtgo_local_rows(m_height, m_opt.bandHeight, static_cast<uint32_t>(V), static_cast<uint32_t>(v));
            m_rowsPad = rows > m_rowsPad ? rows : m_rowsPad;
        }
        if (m_rowsPad == 0) m_rowsPad = 1;
        // scene tables once, as Renderer::CreateShapes / CreateRayGen / CreateMiss / WriteLights flatten them (renderer.cpp:321-453, 655-677)
        std::vector<rtgo_prim> records;
        std::vector<rtgo_aabb> boxes;
        for (const std::shared_ptr<Shape>& shape : m_scene->GetShapes())
            for (const Primitive& primitive : shape->GetPrimitives()) {
                rtgo_prim rec;
                primitive.CopyToDevice(rec);
                records.push_back(rec);
                boxes.push_back(primitive.GetAabb());
            }
        std::vector<rtgo_light> lights;
        for (const SurfaceLight& l : m_scene->GetSurfaceLights()) {
            if (lights.size() >= RTGO_MAX_LIGHTS) break;
            rtgo_light r;
            const glm::vec3 c = l.GetCorner(), v1 = l.GetV1(), v2 = l.GetV2(), n = l.GetNormal(), col = l.GetColor();
            const float vals[15] = {c.x, c.y, c.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z, n.x, n.y, n.z, col.x, col.y, col.z};
            for (int k = 0; k < 15; ++k) (&r.corner[0])[k] = vals[k];
            r.falloff = l.GetFalloff();
            lights.push_back(r);
        }
        float3 u, v, w;
        m_scene->GetCamera()->UVWFrame(u, v, w);
        const float3 e = m_scene->GetCamera()->eye();
        const float eye[3] = {e.x, e.y, e.z}, U[3] = {u.x, u.y, u.z}, Vv[3] = {v.x, v.y, v.z}, W[3] = {w.x, w.y, w.z};
        const glm::vec3 bg = m_scene->GetBackgroundColor();
        const float rgb[3] = {bg.r, bg.g, bg.b};

        const size_t bandPixels = static_cast<size_t>(m_rowsPad) * m_width;
        for (int g = 0; g < nGpus; ++g) {
            std::unique_ptr<Gpu> gpu(new Gpu());
            gpu->device = m_opt.devices[static_cast<size_t>(g)];
            gpu::SetDevice(gpu->device);
            gpu->commStream = gpu::StreamCreate(true);
            m_gpus.push_back(std::move(gpu));
            for (int i = 0; i < m_opt.launchesPerDevice; ++i) {
                // (in m_shares BEFORE anything that can throw: CleanUp then destroys whatever of it exists -- context, stream, buffers, events)
                m_shares.push_back(std::unique_ptr<Share>(new Share()));
                Share* s = m_shares.back().get();
                s->gpu = g;
                const int rc = rtgo_create(m_opt.devices[static_cast<size_t>(g)], &s->ctx);
                if (rc != RTGO_OK) throw std::runtime_error(std::string("rtgo_create failed (") + std::to_string(rc) + "): " + rtgo_last_error(nullptr));
                auto check = [&](int r, const char* what) {
                    if (r != RTGO_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(r) + "): " + rtgo_last_error(s->ctx));
                };
                check(rtgo_set_scene(s->ctx, records.data(), boxes.data(), static_cast<uint32_t>(records.size())), "rtgo_set_scene");
                check(rtgo_set_camera(s->ctx, eye, U, Vv, W), "rtgo_set_camera");
                check(rtgo_set_background(s->ctx, rgb), "rtgo_set_background");
                check(rtgo_set_lights(s->ctx, lights.data(), static_cast<int>(lights.size())), "rtgo_set_lights");
                s->stream = gpu::StreamCreate(false);
                check(rtgo_set_stream(s->ctx, s->stream), "rtgo_set_stream");
                s->accum = gpu::Malloc(bandPixels * 16);
                for (int b = 0; b < 2; ++b) {
                    s->image[b] = gpu::Malloc(bandPixels * 4);
                    s->rendered[b] = gpu::EventCreate();
                    s->gathered[b] = gpu::EventCreate();
                }
            }
        }
        gpu::SetDevice(m_gpus[0]->device);
        m_fullImage = gpu::Malloc(static_cast<size_t>(m_width) * m_height * 4);
        m_gatherImage = gpu::Malloc(bandPixels * 4 * m_shares.size());
        for (auto& g : m_gpus) {
            gpu::SetDevice(g->device);
            gpu::DeviceSync();
        }
        if (nGpus > 1 || m_opt.rcclForLocalShares) {
            // one process, one rank per GPU (ncclCommInitAll); xGMI is point to point, and a gather to one root uses the root's links side by side
            m_comms = gpu::CommsCreate(m_opt.devices);
            for (int g = 0; g < nGpus; ++g) m_gpus[static_cast<size_t>(g)]->commRank = g;
        }
    } catch (...) {
        CleanUp();
        throw;
    }
}

MultiGpuRenderer::~MultiGpuRenderer() { CleanUp(); }

void MultiGpuRenderer::CleanUp()
{
    // (destructor path: nothing here throws)
    try {
        for (auto& g : m_gpus)
            if (g) {
                gpu::SetDevice(g->device);
                gpu::DeviceSync();
            }
    } catch (...) {
    }
    for (auto& s : m_shares) {
        if (!s) continue;
        try { gpu::SetDevice(m_gpus[static_cast<size_t>(s->gpu)]->device); } catch (...) {}
        if (s->ctx) rtgo_destroy(s->ctx);
        gpu::Free(s->accum);
        for (int b = 0; b < 2; ++b) {
            gpu::Free(s->image[b]);
            if (s->rendered[b]) gpu::EventDestroy(s->rendered[b]);
            if (s->gathered[b]) gpu::EventDestroy(s->gathered[b]);
        }
        if (s->stream) gpu::StreamDestroy(s->stream);
    }
    m_shares.clear();
    if (!m_gpus.empty()) {
        try { gpu::SetDevice(m_gpus[0]->device); } catch (...) {}
        gpu::Free(m_fullImage);
        gpu::Free(m_gatherImage);
        gpu::Free(m_fullAccum);
        gpu::Free(m_gatherAccum);
    }
    m_fullImage = m_gatherImage = m_fullAccum = m_gatherAccum = nullptr;
    gpu::CommsDestroy(m_comms);
    m_comms = nullptr;
    for (auto& g : m_gpus) {
        if (!g) continue;
        try { gpu::SetDevice(g->device); } catch (...) {}
        if (g->commStream) gpu::StreamDestroy(g->commStream);
    }
    m_gpus.clear();
}

void MultiGpuRenderer::RenderFrame()
{
    // frame counter rule of Renderer::Update (renderer.cpp:682) without camera / resize events
    m_frameCount = m_first ? 0 : m_frameCount + 1;
    m_first = false;
    const int b = static_cast<int>(m_frameCount & 1u);
    const uint32_t V = static_cast<uint32_t>(m_shares.size());
    for (uint32_t v = 0; v < V; ++v) {
        Share& s = *m_shares[v];
        gpu::SetDevice(m_gpus[static_cast<size_t>(s.gpu)]->device);
        if (s.gatherPending[b]) {   // do not overwrite a band buffer a gather is still reading
            gpu::StreamWaitEvent(s.stream, s.gathered[b]);
            s.gatherPending[b] = false;
        }
        auto check = [&](int r, const char* what) {
            if (r != RTGO_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(r) + "): " + rtgo_last_error(s.ctx));
        };
        check(rtgo_bind_output(s.ctx, s.accum, s.image[b], static_cast<size_t>(m_rowsPad) * m_width), "rtgo_bind_output");
        rtgo_frame f = rtgo_frame();
        f.image_width = m_width;
        f.image_height = m_height;
        f.sqrt_spp = m_sqrtSpp;
        f.max_trace_depth = 5;   // renderer.cpp:616
        f.frame_count = m_frameCount;
        f.path_tracing = m_renderMode == RenderMode::PATH_TRACING ? 1u : 0u;
        f.use_ambient = m_useAmbient ? 1u : 0u;
        f.band_h = m_opt.bandHeight;
        f.n_ranks = V;
        f.rank = v;
        f.reserve_cus = (m_gpus.size() > 1 || m_opt.rcclForLocalShares) ? m_opt.reserveCus : 0u;
        check(rtgo_launch(s.ctx, &f), "rtgo_launch");
        gpu::EventRecord(s.rendered[b], s.stream);
    }
    m_lastBuffer = b;
}

void MultiGpuRenderer::Gather(bool accum, int buffer)
{
    Gpu& root = *m_gpus[0];
    const size_t elem = accum ? 16 : 4;
    const size_t bandBytes = static_cast<size_t>(m_rowsPad) * m_width * elem;
    char* workspace = static_cast<char*>(accum ? m_gatherAccum : m_gatherImage);
    void* full = accum ? m_fullAccum : m_fullImage;
    // every GPU's communication stream waits for that GPU's kernels of this frame; the root's also waits for its own shares
    for (auto& sp : m_shares) {
        Share& s = *sp;
        Gpu& g = *m_gpus[static_cast<size_t>(s.gpu)];
        gpu::SetDevice(g.device);
        gpu::StreamWaitEvent(g.commStream, s.rendered[buffer]);
    }
    // the root GPU's own bands: a device-to-device copy, unless asked to go through RCCL as well
    bool anyRccl = false;
    for (size_t v = 0; v < m_shares.size(); ++v) {
        Share& s = *m_shares[v];
        const void* src = accum ? s.accum : s.image[buffer];
        if (s.gpu == 0 && !m_opt.rcclForLocalShares) {
            gpu::SetDevice(root.device);
            gpu::CopyDeviceToDeviceAsync(workspace + v * bandBytes, src, bandBytes, root.commStream);
        } else {
            anyRccl = true;
        }
    }
    if (anyRccl) {
        // grouped point-to-point gather: every band is one ncclSend on its GPU matched by one ncclRecv on the root
        gpu::GroupStart();
        for (size_t v = 0; v < m_shares.size(); ++v) {
            Share& s = *m_shares[v];
            if (s.gpu == 0 && !m_opt.rcclForLocalShares) continue;
            Gpu& g = *m_gpus[static_cast<size_t>(s.gpu)];
            const void* src = accum ? s.accum : s.image[buffer];
            gpu::Send(m_comms, g.commRank, root.commRank, src, bandBytes, g.commStream);
            gpu::Recv(m_comms, root.commRank, g.commRank, workspace + v * bandBytes, bandBytes, root.commStream);
        }
        gpu::GroupEnd();
    }
    // root: scatter the compact bands to the rows they belong to
    gpu::SetDevice(root.device);
    const int rc = rtgo_assemble_bands(m_shares[0]->ctx, root.commStream, workspace, full, m_width, m_height, m_opt.bandHeight,
                                       static_cast<uint32_t>(m_shares.size()), m_rowsPad, static_cast<uint32_t>(elem));
    if (rc != RTGO_OK) throw std::runtime_error(std::string("rtgo_assemble_bands failed: ") + rtgo_last_error(m_shares[0]->ctx));
    if (!accum)
        for (auto& sp : m_shares) {
            Share& s = *sp;
            Gpu& g = *m_gpus[static_cast<size_t>(s.gpu)];
            gpu::SetDevice(g.device);
            gpu::EventRecord(s.gathered[buffer], g.commStream);
            s.gatherPending[buffer] = true;
        }
}

void MultiGpuRenderer::Present()
{
    if (m_first) throw std::logic_error("MultiGpuRenderer::Present before the first frame");
    Gather(false, m_lastBuffer);
}

void MultiGpuRenderer::Sync()
{
    for (auto& sp : m_shares) {
        const int rc = rtgo_sync(sp->ctx);
        if (rc != RTGO_OK) throw std::runtime_error(std::string("rtgo_sync failed: ") + rtgo_last_error(sp->ctx));
    }
    for (auto& g : m_gpus) {
        gpu::SetDevice(g->device);
        gpu::StreamSync(g->commStream);
    }
}

void MultiGpuRenderer::Display()
{
    Sync();
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < m_frames; ++f) {
        RenderFrame();
        if ((f + 1) % m_opt.presentEvery == 0 || f == m_frames - 1) Present();
    }
    Sync();
    m_msPerFrame = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (m_frames > 0 ? m_frames : 1);
    if (!m_outputFile.empty()) {
        std::cout << "Saving to file " << m_outputFile << std::endl;
        const std::vector<unsigned char> img = ReadImage();
        Renderer::SavePPM(m_outputFile, img.data(), m_width, m_height);
        std::cout << "Save complete" << std::endl;
    }
    if (!m_accumFile.empty()) {
        const std::vector<float> acc = ReadAccum();
        Renderer::SavePFM(m_accumFile, acc.data(), m_width, m_height);
    }
}

std::vector<unsigned char> MultiGpuRenderer::ReadImage()
{
    Sync();
    std::vector<unsigned char> out(static_cast<size_t>(m_width) * m_height * 4);
    gpu::SetDevice(m_gpus[0]->device);
    gpu::CopyDeviceToHost(out.data(), m_fullImage, out.size());
    return out;
}

std::vector<float> MultiGpuRenderer::ReadAccum()
{
    if (m_first) throw std::logic_error("MultiGpuRenderer::ReadAccum before the first frame");
    gpu::SetDevice(m_gpus[0]->device);
    const size_t bandBytes = static_cast<size_t>(m_rowsPad) * m_width * 16;
    if (!m_fullAccum) m_fullAccum = gpu::Malloc(static_cast<size_t>(m_width) * m_height * 16);
    if (!m_gatherAccum) m_gatherAccum = gpu::Malloc(bandBytes * m_shares.size());
    Gather(true, m_lastBuffer);
    Sync();
    std::vector<float> out(static_cast<size_t>(m_width) * m_height * 4);
    gpu::CopyDeviceToHost(out.data(), m_fullAccum, out.size() * sizeof(float));
    return out;
}

rtgo_stats MultiGpuRenderer::Stats()
{
    rtgo_stats tot = rtgo_stats();
    for (auto& sp : m_shares) {
        rtgo_stats s;
        const int rc = rtgo_get_stats(sp->ctx, &s);
        if (rc != RTGO_OK) throw std::runtime_error(std::string("rtgo_get_stats failed: ") + rtgo_last_error(sp->ctx));
        tot.rays_total += s.rays_total;
        tot.rays_occlusion += s.rays_occlusion;
        tot.node_visits += s.node_visits;
        tot.prim_tests += s.prim_tests;
        tot.hits += s.hits;
        tot.rays_culled += s.rays_culled;
        tot.last_launch_ms = s.last_launch_ms > tot.last_launch_ms ? s.last_launch_ms : tot.last_launch_ms;
        tot.total_launch_ms = s.total_launch_ms > tot.total_launch_ms ? s.total_launch_ms : tot.total_launch_ms;
        tot.launches = s.launches;
        tot.launches_canonical += s.launches_canonical;
        tot.lbvh_depth = s.lbvh_depth;
        tot.cuboid_groups = s.cuboid_groups;
        tot.guard_reach = s.guard_reach;       // (the shares see one scene from one eye)
        tot.guard_quadric = s.guard_quadric;
        tot.last_variant |= s.last_variant;
        tot.launches_trial += s.launches_trial;
    }
    return tot;
}

}  // namespace host
}  // namespace engine
