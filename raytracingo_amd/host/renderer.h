// renderer.h -- engine::host::Renderer (engine/renderer.h:55-231) over the rtgo C ABI instead of the OptiX host API.
// Same constructor and Display() entry point; since the MI355X box has no display stack, Display() runs a fixed number
// of progressive frames headlessly (the reference's Update/LaunchFrame loop, renderer.cpp:841-862, without GLFW) and
// can write the Ctrl+S image (renderer.cpp:684-695).
#pragma once
#include "scene.h"
#include <rtgo.h>
#include <memory>
#include <string>
#include <vector>

namespace engine {
namespace host {
enum class RenderMode { DISTRIBUTED_RAY_TRACING, PATH_TRACING };

class Renderer {
public:
    /// \param sqrtSamplePerPixel N: N x N stratified samples per pixel per frame (--sample)
    Renderer(std::shared_ptr<Scene> scene, RenderMode renderMode, int sqrtSamplePerPixel, bool useAmbientCoeff);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    /// Run the frame loop: frames 0..m_frames-1 accumulate into the running average; then save if an output is set.
    void Display();

    // ---- headless controls (no reference counterpart: the reference is interactive only) ----
    void SetFrames(int frames) { m_frames = frames; }
    void SetOutputFile(const std::string& path) { m_outputFile = path; }
    /// also dump the float accumulation buffer (PFM, RGB float32, bottom row first as PFM specifies = buffer order)
    void SetAccumFile(const std::string& path) { m_accumFile = path; }
    void SetDevice(int device) { m_device = device; }
    /// render exactly one more frame (frameCount advances like Renderer::Update does)
    void RenderFrame();
    /// what the trackball callbacks do to the camera (renderer.cpp:36-145 set cameraChangedFlag): the next frame restarts the
    /// accumulation at frameCount 0 and re-uploads the raygen record (Renderer::UpdateCamera, renderer.cpp:703-717)
    void MoveCamera(const float3& eye, const float3& lookat, const float3& up);
    /// what windowSizeCallback does (renderer.cpp:61-78 set windowResizeFlag): new image size, aspect ratio, buffers; restarts at 0
    void Resize(unsigned int width, unsigned int height);
    unsigned int Width() const { return m_params.image_width; }
    unsigned int Height() const { return m_params.image_height; }
    unsigned int FrameCount() const { return m_params.frame_count; }
    /// the 8-bit image (uchar4, row 0 = bottom row) and the float accumulation buffer of the last frame
    std::vector<unsigned char> ReadImage();
    std::vector<float> ReadAccum();
    rtgo_stats Stats();
    /// P6 file, rows flipped so that the top row comes first, alpha dropped (sutil.cpp:81-101, 377-388)
    static void SavePPM(const std::string& path, const unsigned char* rgba, unsigned int width, unsigned int height);
    /// "PF" file, little-endian float RGB; the reference never saves its accumulation buffer, this does
    static void SavePFM(const std::string& path, const float* rgba, unsigned int width, unsigned int height);

private:
    std::shared_ptr<Scene> m_scene;
    RenderMode m_renderMode;
    bool m_useAmbientCoefficient;
    int m_sqrtSamplePerPixel;
    rtgo_ctx* m_context;
    rtgo_frame m_params;   // the launch constants the reference keeps in device::Params
    bool m_firstLaunch;
    bool m_cameraChangedFlag;   // RendererState::cameraChangedFlag
    bool m_windowResizeFlag;    // RendererState::windowResizeFlag
    int m_frames;
    int m_device;
    std::string m_outputFile;
    std::string m_accumFile;

    void Initialize();
    void CreateContext();
    void CreateRayGen();
    void CreateMiss();
    void CreateShapes();
    void WriteLights();
    void Update();
    void UpdateCamera();
    void ResizeBuffers();
    void LaunchFrame();
    void CleanUp();
    void Check(int rc, const char* what) const;
};
}  // namespace host
}  // namespace engine
