// renderer.cpp -- Renderer over the rtgo C ABI.  Each private step names the reference step it stands in for.
#include "renderer.h"

#include <fstream>
#include <iostream>
#include <stdexcept>

namespace engine {
namespace host {

Renderer::Renderer(std::shared_ptr<Scene> scene, RenderMode renderMode, int sqrtSamplePerPixel, bool useAmbientCoeff)
    : m_scene(scene), m_renderMode(renderMode), m_useAmbientCoefficient(useAmbientCoeff), m_sqrtSamplePerPixel(sqrtSamplePerPixel),
      m_context(nullptr), m_params(), m_firstLaunch(true), m_cameraChangedFlag(false), m_windowResizeFlag(false), m_frames(1),
      m_device(0)
{
}

Renderer::~Renderer() { CleanUp(); }

void Renderer::Check(int rc, const char* what) const
{
    // OPTIX_CHECK / CUDA_CHECK throw sutil::Exception (a std::runtime_error) carrying the failing call (Exception.h:93-157)
    if (rc != RTGO_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + rtgo_last_error(m_context));
}

void Renderer::Initialize()
{
    // renderer.cpp:181-192
    CreateContext();
    CreateRayGen();
    CreateMiss();
    CreateShapes();
    WriteLights();
    const unsigned int width = m_scene->GetCameraWidth(), height = m_scene->GetCameraHeight();
    // Params of Renderer::Display (renderer.cpp:795-804)
    m_params = rtgo_frame();
    m_params.image_width = width;
    m_params.image_height = height;
    m_params.sqrt_spp = m_sqrtSamplePerPixel;
    m_params.use_ambient = m_useAmbientCoefficient ? 1u : 0u;
    m_params.max_trace_depth = 5;  // OptixPipelineLinkOptions::maxTraceDepth (renderer.cpp:616)
    m_params.path_tracing = m_renderMode == RenderMode::PATH_TRACING ? 1u : 0u;
    m_params.frame_count = 0;
    Check(rtgo_resize(m_context, static_cast<size_t>(width) * height), "rtgo_resize");
}

void Renderer::CreateContext()
{
    // InitOptix + CreateContext + CreateModule + CreatePipeline (renderer.cpp:194-273, 613-634)
    const int rc = rtgo_create(m_device, &m_context);
    if (rc != RTGO_OK) throw std::runtime_error(std::string("rtgo_create failed (") + std::to_string(rc) + "): " + rtgo_last_error(nullptr));
}

void Renderer::CreateRayGen()
{
    // raygen record = eye + UVW frame (renderer.cpp:321-336)
    const std::shared_ptr<sutil::Camera> camera = m_scene->GetCamera();
    float3 u, v, w;
    camera->UVWFrame(u, v, w);
    const float3 e = camera->eye();
    const float eye[3] = {e.x, e.y, e.z}, U[3] = {u.x, u.y, u.z}, V[3] = {v.x, v.y, v.z}, W[3] = {w.x, w.y, w.z};
    Check(rtgo_set_camera(m_context, eye, U, V, W), "rtgo_set_camera");
}

void Renderer::CreateMiss()
{
    // miss record = background colour (renderer.cpp:386-398)
    const glm::vec3 bg = m_scene->GetBackgroundColor();
    const float rgb[3] = {bg.r, bg.g, bg.b};
    Check(rtgo_set_background(m_context, rgb), "rtgo_set_background");
}

void Renderer::CreateShapes()
{
    // flatten shapes -> primitives in scene order; primitive i is SBT index i (renderer.cpp:400-453)
    std::vector<rtgo_prim> records;
    std::vector<rtgo_aabb> boxes;
    records.reserve(static_cast<size_t>(m_scene->GetNbObjects()));
    for (const std::shared_ptr<Shape>& shape : m_scene->GetShapes())
        for (const Primitive& primitive : shape->GetPrimitives()) {
            rtgo_prim rec;
            primitive.CopyToDevice(rec);
            records.push_back(rec);
            boxes.push_back(primitive.GetAabb());
        }
    // BuildAccelerationStructure + BuildHitGroupRecords (renderer.cpp:514-611, 636-653)
    Check(rtgo_set_scene(m_context, records.data(), boxes.data(), static_cast<uint32_t>(records.size())), "rtgo_set_scene");
}

void Renderer::WriteLights()
{
    // renderer.cpp:655-677
    std::vector<rtgo_light> out;
    for (const SurfaceLight& l : m_scene->GetSurfaceLights()) {
        if (out.size() >= RTGO_MAX_LIGHTS) break;
        rtgo_light r;
        const glm::vec3 c = l.GetCorner(), v1 = l.GetV1(), v2 = l.GetV2(), n = l.GetNormal(), col = l.GetColor();
        r.corner[0] = c.x; r.corner[1] = c.y; r.corner[2] = c.z;
        r.v1[0] = v1.x; r.v1[1] = v1.y; r.v1[2] = v1.z;
        r.v2[0] = v2.x; r.v2[1] = v2.y; r.v2[2] = v2.z;
        r.normal[0] = n.x; r.normal[1] = n.y; r.normal[2] = n.z;
        r.color[0] = col.x; r.color[1] = col.y; r.color[2] = col.z;
        r.falloff = l.GetFalloff();
        out.push_back(r);
    }
    Check(rtgo_set_lights(m_context, out.data(), static_cast<int>(out.size())), "rtgo_set_lights");
}

void Renderer::Update()
{
    // frame counter rule of renderer.cpp:682: any camera change or resize restarts the running average
    m_params.frame_count = (m_cameraChangedFlag || m_windowResizeFlag || m_firstLaunch) ? 0 : m_params.frame_count + 1;
    m_firstLaunch = false;
    UpdateCamera();
    ResizeBuffers();
}

void Renderer::UpdateCamera()
{
    // renderer.cpp:703-717: aspect ratio from the current image size, then the raygen record is re-uploaded
    if (!m_cameraChangedFlag) return;
    m_cameraChangedFlag = false;
    m_scene->GetCamera()->setAspectRatio(static_cast<float>(m_params.image_width) / static_cast<float>(m_params.image_height));
    CreateRayGen();
}

void Renderer::ResizeBuffers()
{
    // renderer.cpp:733-747: output buffer and accumulation buffer are re-allocated for the new size
    if (!m_windowResizeFlag) return;
    m_windowResizeFlag = false;
    Check(rtgo_resize(m_context, static_cast<size_t>(m_params.image_width) * m_params.image_height), "rtgo_resize");
}

void Renderer::MoveCamera(const float3& eye, const float3& lookat, const float3& up)
{
    if (!m_context) Initialize();
    const std::shared_ptr<sutil::Camera> camera = m_scene->GetCamera();
    camera->setEye(eye);
    camera->setLookat(lookat);
    camera->setUp(up);
    m_cameraChangedFlag = true;
}

void Renderer::Resize(unsigned int width, unsigned int height)
{
    if (!m_context) Initialize();
    if (width < 1 || height < 1) throw std::invalid_argument("Renderer::Resize: empty image");
    // windowSizeCallback (renderer.cpp:61-78) flags both: the aspect ratio feeds the camera frame
    m_params.image_width = width;
    m_params.image_height = height;
    m_cameraChangedFlag = true;
    m_windowResizeFlag = true;
}

void Renderer::LaunchFrame()
{
    // optixLaunch + stream sync + CUDA_SYNC_CHECK (renderer.cpp:749-774)
    Check(rtgo_launch(m_context, &m_params), "rtgo_launch");
    Check(rtgo_sync(m_context), "rtgo_sync");
}

void Renderer::RenderFrame()
{
    if (!m_context) Initialize();
    Update();
    LaunchFrame();
}

void Renderer::Display()
{
    if (!m_context) Initialize();
    for (int f = 0; f < m_frames; ++f) RenderFrame();
    if (!m_outputFile.empty()) {
        std::cout << "Saving to file " << m_outputFile << std::endl;
        const std::vector<unsigned char> img = ReadImage();
        SavePPM(m_outputFile, img.data(), m_scene->GetCameraWidth(), m_scene->GetCameraHeight());
        std::cout << "Save complete" << std::endl;
    }
    if (!m_accumFile.empty()) {
        const std::vector<float> acc = ReadAccum();
        SavePFM(m_accumFile, acc.data(), m_scene->GetCameraWidth(), m_scene->GetCameraHeight());
    }
}

std::vector<unsigned char> Renderer::ReadImage()
{
    std::vector<unsigned char> out(static_cast<size_t>(m_params.image_width) * m_params.image_height * 4);
    Check(rtgo_read_image(m_context, out.data(), out.size()), "rtgo_read_image");
    return out;
}

std::vector<float> Renderer::ReadAccum()
{
    std::vector<float> out(static_cast<size_t>(m_params.image_width) * m_params.image_height * 4);
    Check(rtgo_read_accum(m_context, out.data(), out.size() * sizeof(float)), "rtgo_read_accum");
    return out;
}

rtgo_stats Renderer::Stats()
{
    rtgo_stats s;
    Check(rtgo_get_stats(m_context, &s), "rtgo_get_stats");
    return s;
}

void Renderer::SavePPM(const std::string& path, const unsigned char* rgba, unsigned int width, unsigned int height)
{
    if (!rgba || width < 1 || height < 1) throw std::invalid_argument("Image is ill-formed. Not saving");
    std::ofstream out(path, std::ios::out | std::ios::binary);
    if (!out.is_open()) throw std::runtime_error("Could not open file for SavePPM");
    out << "P6\n" << width << " " << height << "\n255\n";
    std::vector<unsigned char> row(static_cast<size_t>(width) * 3);
    for (unsigned int y = height; y-- > 0;) {  // buffer row 0 is the bottom of the picture
        const unsigned char* src = rgba + static_cast<size_t>(y) * width * 4;
        for (unsigned int x = 0; x < width; ++x) {
            row[3 * x + 0] = src[4 * x + 0];
            row[3 * x + 1] = src[4 * x + 1];
            row[3 * x + 2] = src[4 * x + 2];
        }
        out.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size()));
    }
}

void Renderer::SavePFM(const std::string& path, const float* rgba, unsigned int width, unsigned int height)
{
    if (!rgba || width < 1 || height < 1) throw std::invalid_argument("Image is ill-formed. Not saving");
    std::ofstream out(path, std::ios::out | std::ios::binary);
    if (!out.is_open()) throw std::runtime_error("Could not open file for SavePFM");
    out << "PF\n" << width << " " << height << "\n-1.0\n";  // negative scale = little endian; rows go bottom to top
    std::vector<float> row(static_cast<size_t>(width) * 3);
    for (unsigned int y = 0; y < height; ++y) {
        const float* src = rgba + static_cast<size_t>(y) * width * 4;
        for (unsigned int x = 0; x < width; ++x) {
            row[3 * x + 0] = src[4 * x + 0];
            row[3 * x + 1] = src[4 * x + 1];
            row[3 * x + 2] = src[4 * x + 2];
        }
        out.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(float)));
    }
}

void Renderer::CleanUp()
{
    // renderer.cpp:870-885
    if (m_context) {
        rtgo_destroy(m_context);
        m_context = nullptr;
    }
}
}  // namespace host
}  // namespace engine
