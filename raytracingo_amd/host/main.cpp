// main.cpp -- headless `engine` command: the flags of engine/main.cpp:38-167 (--dim, --mode, --scene, --sample,
// --useAmbient, --help; --mode and --scene mandatory) plus what a box without a display needs: --frames, --out, --device.
#include "multigpu.h"
#include "renderer.h"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>

using engine::host::RenderMode;
using engine::host::SceneModel;

static void printUsageAndExit(const char* argv0)
{
    std::cerr << "Usage  : " << argv0 << " [options]\n"
              << "         --help | -h                 Print this usage message\n"
              << "         --dim=<width>x<height>      Set image dimensions; defaults to 600x600\n"
              << "         --mode=distributed OR path  Distributed ray tracing or path tracing (mandatory)\n"
              << "         --scene=<scene>             plateau | slide | cornell | mirror_spheres | soft_mirrors | window | balls | checkered (mandatory)\n"
              << "         --sample=<N>                N*N stratified samples per pixel per frame; default 1\n"
              << "         --useAmbient                Use an ambient coefficient instead of diffuse inter-reflection\n"
              << "         --frames=<K>                Progressive frames to accumulate; default 1   [headless addition]\n"
              << "         --out=<file.ppm>            Write the final 8-bit image (P6)              [headless addition]\n"
              << "         --out-accum=<file.pfm>      Write the float accumulation buffer (PFM)       [headless addition]\n"
              << "         --device=<i>                GPU index; default 0                          [headless addition]\n"
              << "         --gpus=<n>                  Tile the frame in 4-row bands over GPUs 0..n-1, RCCL gather to GPU 0 [multi-GPU addition]\n"
              << "         --launches-per-gpu=<1|2>    Shares per GPU; default 1                       [multi-GPU addition]\n"
              << "         --present-every=<k>         Gather + assemble every k-th frame; default 1   [multi-GPU addition]\n";
    std::exit(1);
}

static void parseDimensions(const char* arg, int& width, int& height)
{
    // <width>x<height> (sutil.cpp:523-548)
    const char* x = std::strchr(arg, 'x');
    if (x && x[1] != '\0' && x != arg) {
        width = std::atoi(std::string(arg, x).c_str());
        height = std::atoi(x + 1);
        return;
    }
    throw std::invalid_argument("Failed to parse width, height from string '" + std::string(arg) + "'");
}

int main(int argc, char* argv[])
{
    int width = 600, height = 600, sample = 1, frames = 1, device = 0, gpus = 0, launchesPerGpu = 0, presentEvery = 1;
    bool modeFound = false, sceneFound = false, useAmbient = false;
    RenderMode mode = RenderMode::PATH_TRACING;
    SceneModel scene = SceneModel::CORNELL;
    std::string out, outAccum;
    static const struct { const char* name; SceneModel model; } kScenes[] = {
        {"plateau", SceneModel::PLATE}, {"cornell", SceneModel::CORNELL}, {"slide", SceneModel::SLIDE}, {"window", SceneModel::WINDOW},
        {"balls", SceneModel::BALLS}, {"checkered", SceneModel::CHECKERED}, {"mirror_spheres", SceneModel::MIRROR_SPHERES},
        {"soft_mirrors", SceneModel::SOFT_MIRRORS}};
    try {
        for (int i = 1; i < argc; ++i) {
            const std::string arg(argv[i]);
            auto value = [&](const char* key) { return arg.substr(std::strlen(key)); };
            auto is = [&](const char* key) { return arg.compare(0, std::strlen(key), key) == 0; };
            if (arg == "--help" || arg == "-h") printUsageAndExit(argv[0]);
            else if (is("--dim=")) parseDimensions(value("--dim=").c_str(), width, height);
            else if (is("--mode=")) {
                modeFound = true;
                if (value("--mode=") == "distributed") mode = RenderMode::DISTRIBUTED_RAY_TRACING;
                else if (value("--mode=") == "path") mode = RenderMode::PATH_TRACING;
                else { std::cerr << "Unknown option '" << arg << "'\n"; printUsageAndExit(argv[0]); }
            } else if (is("--scene=")) {
                sceneFound = true;
                bool ok = false;
                for (const auto& s : kScenes)
                    if (value("--scene=") == s.name) { scene = s.model; ok = true; }
                if (!ok) { std::cerr << "Unknown option '" << arg << "'\n"; printUsageAndExit(argv[0]); }
            } else if (is("--sample=")) sample = std::atoi(value("--sample=").c_str());
            else if (is("--useAmbient")) useAmbient = true;
            else if (is("--frames=")) frames = std::atoi(value("--frames=").c_str());
            else if (is("--out-accum=")) outAccum = value("--out-accum=");
            else if (is("--out=")) out = value("--out=");
            else if (is("--device=")) device = std::atoi(value("--device=").c_str());
            else if (is("--gpus=")) gpus = std::atoi(value("--gpus=").c_str());
            else if (is("--launches-per-gpu=")) launchesPerGpu = std::atoi(value("--launches-per-gpu=").c_str());
            else if (is("--present-every=")) presentEvery = std::atoi(value("--present-every=").c_str());
            else { std::cerr << "Unknown option '" << arg << "'\n"; printUsageAndExit(argv[0]); }
        }
        if (!modeFound) { std::cerr << "Argument manquant: --mode=" << std::endl; printUsageAndExit(argv[0]); }
        if (!sceneFound) { std::cerr << "Argument manquant: --scene=" << std::endl; printUsageAndExit(argv[0]); }

        auto sc = std::make_shared<engine::host::Scene>(scene, width, height);
        if (gpus >= 1) {
            engine::host::MultiGpuRenderer::Options opt;
            for (int g = 0; g < gpus; ++g) opt.devices.push_back(g);
            opt.launchesPerDevice = launchesPerGpu > 0 ? launchesPerGpu : 1;   // (two half-share launches stopped paying with round 3's kernel: DESIGN.md section 6)
            opt.presentEvery = presentEvery;
            engine::host::MultiGpuRenderer renderer(sc, mode, sample, useAmbient, opt);
            renderer.SetFrames(frames);
            renderer.SetOutputFile(out);
            renderer.SetAccumFile(outAccum);
            renderer.Display();
            const rtgo_stats st = renderer.Stats();
            std::cout << "gpus " << gpus << ", shares " << renderer.Shares() << ", frames " << st.launches << ", rays " << st.rays_total
                      << ", ms/frame (host wall, launch + gather + assemble) " << renderer.LastDisplayMsPerFrame() << ", Mray/s "
                      << (renderer.LastDisplayMsPerFrame() > 0 ? st.rays_total / (renderer.LastDisplayMsPerFrame() * frames) / 1e3 : 0.0) << std::endl;
            return 0;
        }
        engine::host::Renderer renderer(sc, mode, sample, useAmbient);
        renderer.SetDevice(device);
        renderer.SetFrames(frames);
        renderer.SetOutputFile(out);
        renderer.SetAccumFile(outAccum);
        renderer.Display();
        const rtgo_stats st = renderer.Stats();
        std::cout << "frames " << st.launches << ", rays " << st.rays_total << ", kernel ms " << st.total_launch_ms << ", Mray/s "
                  << (st.total_launch_ms > 0 ? st.rays_total / st.total_launch_ms / 1e3 : 0.0) << std::endl;
    } catch (std::exception& e) {
        std::cerr << "Caught exception: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
