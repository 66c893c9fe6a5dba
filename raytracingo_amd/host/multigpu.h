// multigpu.h -- engine::host::MultiGpuRenderer: the Renderer's frame loop over several MI355X of one node.
//
// No reference counterpart: the reference is single-GPU (cudaSetDevice(0), renderer.cpp:213-214); the only multi-GPU hint in
// its tree is the SDK's unused StaticWorkDistribution (sutil/WorkDistribution.h:60-81: small tiles dealt round-robin to GPUs).
// Here the framebuffer is dealt in 4-row bands: share v of V renders the window rows r with (r / band_h) % V == v into its own
// compact buffers (rtgo_frame.band_h / n_ranks / rank), with the FULL image size in every launch, so seeds and ray directions
// are those of a single launch and the assembled frame is bitwise the single-GPU frame.  The accumulation
// bands stay resident on the GPU that renders them -- progressive frames need no exchange -- and the only collective is, per
// PRESENTED frame, an RCCL gather of the 8-bit bands to the root GPU (grouped ncclSend / ncclRecv: the root ingests over its
// xGMI links side by side, no ring), followed by a de-interleave kernel on the root (rtgo_assemble_bands).
//
// One host thread, one process: a context, a HIP stream and band buffers per share, a communication stream per GPU.
// A GPU may carry more than one share (launches_per_device = 2: the tail of one launch overlaps the body of the other, which
// pays once a share is only a few hundred microseconds long).
#pragma once
#include "renderer.h"

#include <memory>
#include <string>
#include <vector>

namespace engine {
namespace host {
namespace gpu {
struct Comms;
}

class MultiGpuRenderer {
public:
    struct Options {
        std::vector<int> devices;        ///< HIP device indices, the first one is the root (presents the frame); must be distinct
        int launchesPerDevice = 1;       ///< shares per GPU (1 or 2)
        unsigned int bandHeight = 4;     ///< rows per band of the interleave
        int presentEvery = 1;            ///< gather + assemble every k-th frame (and always the last one of Display())
        bool rcclForLocalShares = false; ///< move the root GPU's own bands through ncclSend/ncclRecv-to-self too instead of a
                                         ///< device-to-device copy (lets a one-GPU box exercise the RCCL path end to end)
        unsigned int reserveCus = 8;     ///< workgroup slots left free on every GPU for RCCL's kernels beside the persistent megakernel
    };

    MultiGpuRenderer(std::shared_ptr<Scene> scene, RenderMode renderMode, int sqrtSamplePerPixel, bool useAmbientCoeff, const Options& options);
    ~MultiGpuRenderer();
    MultiGpuRenderer(const MultiGpuRenderer&) = delete;
    MultiGpuRenderer& operator=(const MultiGpuRenderer&) = delete;

    /// frames 0..frames-1 accumulate into the running average on every share; presents per Options::presentEvery
    void Display();
    void SetFrames(int frames) { m_frames = frames; }
    void SetOutputFile(const std::string& path) { m_outputFile = path; }
    void SetAccumFile(const std::string& path) { m_accumFile = path; }

    /// launch the next frame on every share (asynchronous); frameCount advances like Renderer::Update (renderer.cpp:682)
    void RenderFrame();
    /// gather the 8-bit bands of the last rendered frame to the root and assemble the full image there (asynchronous)
    void Present();
    /// block until every launch and every gather issued so far has completed
    void Sync();
    unsigned int Width() const { return m_width; }
    unsigned int Height() const { return m_height; }
    unsigned int FrameCount() const { return m_frameCount; }
    int Shares() const { return static_cast<int>(m_shares.size()); }
    /// the assembled 8-bit image of the last PRESENTED frame (uchar4, row 0 = bottom row), from the root GPU
    std::vector<unsigned char> ReadImage();
    /// the float accumulation buffer: gathered from the shares on demand (float4 bands through the same path), not per frame
    std::vector<float> ReadAccum();
    /// ray counters summed over the shares; launch times are the slowest share's
    rtgo_stats Stats();
    /// wall time of the host loop per frame in Display(), milliseconds (launch + gather + assemble, all GPUs)
    double LastDisplayMsPerFrame() const { return m_msPerFrame; }

private:
    struct Share;
    struct Gpu;
    std::shared_ptr<Scene> m_scene;
    RenderMode m_renderMode;
    bool m_useAmbient;
    int m_sqrtSpp;
    Options m_opt;
    unsigned int m_width, m_height, m_rowsPad;
    unsigned int m_frameCount;
    bool m_first;
    int m_frames;
    int m_lastBuffer;      // image band buffer the last frame was rendered into
    double m_msPerFrame;
    std::string m_outputFile, m_accumFile;
    std::vector<std::unique_ptr<Gpu>> m_gpus;
    std::vector<std::unique_ptr<Share>> m_shares;
    gpu::Comms* m_comms;   // RCCL communicators (one rank per GPU); null when a single GPU copies its own bands
    void* m_fullImage;     // root: uchar4[W*H]
    void* m_gatherImage;   // root: V * rowsPad * W uchar4
    void* m_fullAccum;     // root, on demand: float4[W*H]
    void* m_gatherAccum;   // root, on demand
    void Gather(bool accum, int buffer);
    void CleanUp();
};

}  // namespace host
}  // namespace engine
