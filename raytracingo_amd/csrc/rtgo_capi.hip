// rtgo_capi.hip -- the C ABI of include/rtgo.h over the gfx950 kernels of rtgo_device.h.
// Host side of the drop-in boundary: where the reference's Renderer calls the OptiX host API, a port calls these.
// There is no CPU fallback anywhere in this file: every path ends in a HIP launch or an error code.
#include "../../include/rtgo.h"
#include "rtgo_device.h"
#include "rtgo_whitted.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace rtgo;

static_assert(sizeof(rtgo_prim) == sizeof(PrimIn), "rtgo_prim layout");
static_assert(sizeof(rtgo_prim) == 108, "rtgo_prim is type + HitGroupData (104 B, params.h:103-110)");
static_assert(sizeof(rtgo_light) == sizeof(LightRec) && sizeof(rtgo_light) == 64, "SurfaceLight is 64 B");
static_assert(sizeof(rtgo_aabb) == 24, "OptixAabb is 24 B");
static_assert(RTGO_MAX_PRIMS == kMaxPrims && RTGO_MAX_LIGHTS == kMaxLights, "limits");

struct rtgo_ctx {
    int device = 0;
    int num_cus = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // ring of HIP-event pairs bracketing each megakernel launch on the launch stream (launches are asynchronous, so the
    // elapsed times are harvested later: at rtgo_sync, or when the ring wraps)
    static constexpr int kEvRing = 64;
    hipEvent_t ev_start[kEvRing] = {}, ev_stop[kEvRing] = {};
    int ev_head = 0, ev_pending = 0;
    unsigned char ev_tag[kEvRing] = {};    // 1 / 2: a trial launch of the streaming / the lock-step loop (see `trial` below)
    // Frames of several passes per pixel (> 16 spp) have two kernels with bitwise the same output: lanes streaming through their
    // samples (open scenes, where path lengths differ: plateau 3840x2160 spp 256 18.8 ms against 20.5) or the wave running pass by
    // pass in lock-step (closed scenes, where nearly every path runs to the depth limit and regeneration only costs: cornell spp 64
    // 4.42 ms against 4.9).  Which one is faster is a property of scene and frame that the host cannot see, but the launch times
    // it takes anyway tell: the first four launches of a (scene, frame geometry, spp, mode) alternate between the two, the faster
    // minimum keeps the job.
    // Round 3: the same trial also decides WHICH fast-walk structure a launch walks.  How big a primitive has to be to be tested up
    // front by every ray instead of sitting in the tree (build_kernel's big_frac) is worth 20 % on plateau (nearly everything up front:
    // a dozen tests at full lanes beat a walk at a third of them) and costs 20 % on cornell (its two boxes lose their cuboid leaves), and
    // no rule read off the scene predicts it (profiles/r03n/big_sweep.log); so rtgo_set_scene builds the structure twice -- 36 % and 15 % --
    // and the candidates of a trial are (loop, structure) pairs: every candidate gets two timed launches, the best minimum keeps the job.
    // All candidates return the same pixels bit for bit (any tree over the same primitives returns the same closest hit).
    struct Trial {
        std::vector<uint32_t> key;
        int issued = 0, done = 0;
        int n_cand = 0;
        float best[8] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
        int choice = -1;                   // index of the winning candidate, -1 = undecided
    } trial;
    // The third structure of the trial: a uniform grid over structure 0's small primitives (rtgo::fast_grid), built by the host from
    // the boxes build_kernel reports.  Scenes of many small primitives spread evenly (balls: 256 spheres in a room) walk it in a third of
    // the tree's instructions; where it is slower the trial drops it after two launches.
    struct Grid {
        void* d = nullptr;                 // n_cells words (first item | count << 16), then 16-bit items: positions into d_fprims
        int n_nodes = 0;                   // its size in 32-byte units (what LaunchParams::n_fnodes counts)
        int entries = 0;                   // list entries (rtgo_debug_grid)
        rtgo::GridParams gp = {};
        float reach_max = 0.0f;            // the pad of the binning covers the walk's rounding for rays that start within this reach
        bool have = false;
    } grid;
    struct FastTree {                      // what build_kernel makes for one big_frac (see the fields of the same names below)
        float4* d_fnodes = nullptr;
        float4* d_fprims = nullptr;
        int fast_depth = 0, n_small = 0, n_fnodes = 0, cuboid_groups = 0, tree_spheres = 0, list_cub = 0, n_big_pairs = 0;
        float cub_a = 0.0f, cub_b = 0.0f;
    } alt;                                 // the second structure (15 %); the first one lives in the fields below
    bool have_alt = false;                 // false: the two builds came out the same, or RTGO_BIG_PERCENT pins one
    int* d_meta_alt = nullptr;
    // scene
    uint32_t n_prims = 0;
    PrimIn* d_prims_in = nullptr;
    float* d_aabb = nullptr;
    float4* d_nodes = nullptr;
    float4* d_prims = nullptr;
    float4* d_fnodes = nullptr;   // collapsed LBVH of the fast walk
    float4* d_fprims = nullptr;   // Morton-ordered traversal records of the fast walk
    float4* d_frames = nullptr;   // shading frames of the flat primitives (2 float4 per primitive, SBT order)
    int* d_meta = nullptr;
    int lbvh_depth = 0;
    int fast_depth = 0;
    int n_small = 0;
    int n_fnodes = 0;             // nodes of the fast walk's tree
    int cuboid_groups = 0;        // certified groups in the scene (leaves + the list's)
    int tree_spheres = 0;         // every primitive of the fast walk's tree is a sphere
    int list_cub = 0;             // the up-front list starts with a certified box (1) / room (2): cuboid_range
    float cub_a = 0.0f, cub_b = 0.0f;   // its margin = kCuboidTol + K (cub_a R + cub_b), R = reach of the launch's rays
    int n_big_pairs = 0;
    float bounds[6] = {0, 0, 0, 0, 0, 0};  // tight world bounds of the scene (min xyz, max xyz)
    // far-field guard (rtgo_launch): per sphere / cylinder its centre and smax / smin^2 of its model matrix' axis scales -- the
    // reported hit of a quadric seen from distance D lies up to ~2^-25 D^2 smax / smin^2 off its surface (b^2 - 4ac cancels)
    struct Quadric {
        float c[3], w;
    };
    std::vector<Quadric> quadrics;
    float guard_reach = 0.0f, guard_quadric = 0.0f;   // of the last launch (rtgo_stats)
    float* d_tight = nullptr;              // the fast walk's box of every primitive (device), and its host copy
    std::vector<float> tight;
    // per-strip mask of the scene's screen rectangle (LaunchParams::hot_mask), kept until the launch geometry changes
    unsigned int* d_mask = nullptr;
    size_t mask_capacity = 0;              // words
    // pinned staging for its upload, two slots used in turn with an event each: a camera change (every frame of an interactive
    // drag) rebuilds the mask, and the upload must not make the host wait for the stream
    unsigned int* h_mask[2] = {nullptr, nullptr};
    size_t h_mask_capacity[2] = {0, 0};
    hipEvent_t mask_copied[2] = {nullptr, nullptr};
    int mask_slot = 0;
    std::vector<uint32_t> mask_key;        // what the cached mask was built for
    bool mask_all_hot = true;
    unsigned long long mask_cold_pixels = 0;
    int leaf_budget = kDefaultLeafBudget;
    LightRec* d_lights = nullptr;
    int n_lights = 0;
    bool have_camera = false;
    v3 eye{0, 0, 0}, U{0, 0, 0}, V{0, 0, 0}, W{0, 0, 0}, bg{0, 0, 0};
    // output
    float4* d_accum = nullptr;
    uchar4* d_image = nullptr;
    size_t pixels = 0;
    bool own_output = false;
    // queue + counters
    unsigned int* d_queue = nullptr;          // two sets of work-queue heads: a launch counts on one and zeroes the other for the next
    int queue_set = 0;
    unsigned long long rays_culled = 0;       // since rtgo_reset_stats (host arithmetic: the cold pixels of each launch x N*N)
    uint32_t launches_canonical = 0;          // since rtgo_reset_stats
    uint32_t launches_trial = 0, last_variant = 0;   // (rtgo_stats)
    unsigned long long* d_counters = nullptr;  // 8 x u64
#ifdef RTGO_CMPWALK
    float* d_cmp = nullptr;                    // diagnostic build: disagreements between the two walks
#endif
#ifdef RTGO_TIMELINE
    unsigned long long* d_timeline = nullptr;  // diagnostic build: 8 x u64 per wave
    unsigned int timeline_waves = 0;
#endif
    float total_ms = 0.0f, last_ms = 0.0f;
    uint32_t launches = 0;
    // the whitted triangle path (rtgo_whitted.h)
    float* w_positions = nullptr;
    float* w_normals = nullptr;
    unsigned int* w_indices = nullptr;
    unsigned int* w_tri_material = nullptr;
    whitted::Pbr* w_materials = nullptr;
    whitted::PointLight* w_lights = nullptr;
    float* w_texcoords = nullptr;              // 2 floats per vertex, or null
    whitted::MatTex* w_mat_tex = nullptr;      // device copy of w_mat_tex_host, or null while no material has a texture
    std::vector<whitted::MatTex> w_mat_tex_host;
    std::vector<void*> w_texels;               // device texel arrays the table points into (freed with the mesh)
    float4* w_nodes = nullptr;
    float4* w_recs = nullptr;              // the walk's records (4 float4 each) and the triangles in Morton order (3 float4 each)
    float4* w_tris = nullptr;
    uint4* w_qrecs = nullptr;              // the compact form: quantised records, (vertex indices | triangle index) per triangle
    uint2* w_tidx = nullptr;
    int w_n_vertices = 0;
    v3 w_grid_lo{0, 0, 0}, w_grid_step{0, 0, 0};
    int* w_scratch = nullptr;
    unsigned int* w_tile_counters = nullptr;   // two sets of tile-queue heads: a launch counts on one and zeroes the other
    int w_n_recs = 0, w_walk_depth = 0, w_launch_parity = 0;
    int w_triangles = 0, w_n_lights = 0, w_n_materials = 0;
    v3 w_miss{0, 0, 0};
    std::string err;
};

static_assert(sizeof(rtgo_pbr) == sizeof(whitted::Pbr) && sizeof(rtgo_point_light) == sizeof(whitted::PointLight) && sizeof(rtgo_point_light) == 32,
              "whitted records");
static_assert(RTGO_MAX_TRIANGLES == whitted::kMaxTriangles, "limits");

static std::string g_create_error;

// dynamic LDS of build_kernel: the fast walk's tree while it is built and rotated (2 * kMaxPrims nodes x (box 24 B + two links + parent))
static constexpr size_t kBuildDynLds = (size_t)2 * kMaxPrims * (6 * sizeof(float) + 3 * sizeof(int));

// every instantiation of the megakernel, in one place: rtgo_create raises the dynamic-LDS limit of each, rtgo_launch picks one
using RenderKernel = void (*)(const LaunchParams, const float4*);
struct RenderKernelEntry {
    bool path, canon, stream;
    int wpe;
    bool count, frames;
    RenderKernel fn;
    bool grid = false;
};
#define RTGO_K(P, S, W, T) {P, S, T, W, S, false, render_kernel<P, S, W, T>}
#define RTGO_KF(W, T) {true, false, T, W, false, true, render_kernel<true, false, W, T, false, true>}
#define RTGO_KG(P, W, T) {P, false, T, W, false, false, render_kernel<P, false, W, T, false, false, true>, true}
static const RenderKernelEntry kRenderKernels[] = {
    RTGO_K(true, false, 4, false),  RTGO_K(true, false, 5, false),    // path mode, fast walk
    RTGO_K(false, false, 4, false), RTGO_K(false, false, 5, false),   // distributed mode, fast walk
    RTGO_K(true, false, 4, true),   RTGO_K(true, false, 5, true),     // ... more than 16 spp: lanes stream through their samples
    RTGO_K(false, false, 4, true),  RTGO_K(false, false, 5, true),
    RTGO_K(true, true, 4, false),   RTGO_K(false, true, 4, false),    // canonical walk + V/T/h counters (collect_stats launches)
    {true, true, false, 4, false, false, render_kernel<true, true, 4, false, false>},     // canonical walk alone: launches beyond the far-field guard
    {false, true, false, 4, false, false, render_kernel<false, true, 4, false, false>},
    RTGO_KF(4, false), RTGO_KF(5, false), RTGO_KF(4, true), RTGO_KF(5, true),               // path mode, fast walk, scenes of flat primitives only: shading frames from LDS
    RTGO_KG(true, 4, false), RTGO_KG(true, 5, false), RTGO_KG(true, 4, true), RTGO_KG(true, 5, true),       // fast walk over the uniform grid instead of the tree (fast_grid)
    RTGO_KG(false, 4, false), RTGO_KG(false, 5, false), RTGO_KG(false, 4, true), RTGO_KG(false, 5, true),
};
#undef RTGO_K
#undef RTGO_KF
#undef RTGO_KG
static RenderKernel find_kernel(bool path, bool canon, int wpe, bool stream, bool count, bool frames, bool grid)
{
    for (const RenderKernelEntry& e : kRenderKernels)
        if (e.path == path && e.canon == canon && e.wpe == (canon ? 4 : wpe) && e.stream == (canon ? false : stream) && e.count == (canon && count) &&
            e.frames == (frames && path && !canon) && e.grid == (grid && !canon))
            return e.fn;
    return nullptr;
}

static int fail(rtgo_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define RTGO_HIP(ctx, call)                                                                                       \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fail(ctx, RTGO_E_HIP_BASE + (int)e_,                                                           \
                        std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

// read back the oldest `count` pending event pairs (blocks until their stop events have completed)
static int harvest_events(rtgo_ctx* c, int count)
{
    while (count-- > 0 && c->ev_pending > 0) {
        const int slot = (c->ev_head - c->ev_pending + 2 * rtgo_ctx::kEvRing) % rtgo_ctx::kEvRing;
        RTGO_HIP(c, hipEventSynchronize(c->ev_stop[slot]));
        float ms = 0.0f;
        RTGO_HIP(c, hipEventElapsedTime(&ms, c->ev_start[slot], c->ev_stop[slot]));
        c->last_ms = ms;
        c->total_ms += ms;
        c->ev_pending--;
        if (c->ev_tag[slot] != 0) {
            float& best = c->trial.best[(c->ev_tag[slot] - 1) & 7];
            best = ms < best ? ms : best;
            c->trial.done++;
            c->ev_tag[slot] = 0;
        }
    }
    return RTGO_OK;
}

// window rows below y that this rank owns under the band interleave
static uint32_t owned_rows_below(uint32_t y, uint32_t band_h, uint32_t n_ranks, uint32_t rank)
{
    if (n_ranks <= 1) return y;
    const uint32_t full = y / band_h, part = y % band_h;
    const uint32_t owned_full = full > rank ? (full - rank - 1) / n_ranks + 1 : 0;
    return owned_full * band_h + ((full % n_ranks == rank) ? part : 0);
}

static inline uint32_t passes_of(uint32_t nn) { return (nn + (uint32_t)kSamplesPerPass - 1u) / (uint32_t)kSamplesPerPass; }

// Far-field guard of rtgo_launch: the fast walk serves launches whose rays stay where every traversal returns the same closest hit;
// beyond, the canonical walk ("far field").  Set from tools/fuzz_farfield.py (-DRTGO_CMPWALK build: both walks on
// every ray; profiles/r03a/farfield_*.log: 3.6e9 rays over translated scenes and eye distances of 10 .. 3000 units):
//   kGuardQuadric  max over spheres / cylinders of Q = D^2 smax / smin^2 (D: farthest ray origin -- eye or scene bounds -- to the
//                  primitive; s: its axis scales).  A quadric's reported hit leaves its surface by ~2^-25 Q (b^2 - 4ac cancels), and
//                  the reference's own box (AABB_EPSILON = 1e-3, primitive.cpp:16) no longer holds it from Q ~ 2^25 * 1e-3 = 33554 on.
//                  Measured: no disagreement in 2.1e9 rays with Q < 32000, the first at Q = 33130; EVERY disagreement found, at any
//                  distance, involves a sphere or a cylinder.  8000 = that onset with a safety factor of 4 on the error.
//   kGuardReach    max(|scene bounds|, |eye|), world units, for what is linear in the coordinates (rectangles, disks, the cuboid
//                  margin): 500.  Flat primitives never disagreed up to the largest reach fuzzed (5000): a factor of 10.
static constexpr float kGuardReach = 500.0f;
static constexpr float kGuardQuadric = 8000.0f;

// tuning knobs for experiments (results never depend on them)
static float env_float(const char* name, float dflt)
{
    const char* v = std::getenv(name);
    return v ? (float)std::atof(v) : dflt;
}
static unsigned int env_uint(const char* name, unsigned int dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    const long k = std::strtol(v, nullptr, 10);
    return k > 0 ? (unsigned int)k : dflt;
}

// Window rectangle [wx0, wx1) x [wy0, wy1) that can contain geometry: the 8 corners of the scene's tight bounds through the
// pinhole camera of the raygen program (d = dx*U + dy*V + W, kernel.cu:214-220; a sample of pixel (x, y) has dx, dy inside
// that pixel's square).  Conservative: padded by two pixels, and the whole window whenever a corner is not in front of the
// eye.  The kernel traces nothing for pixels outside it (their primary rays cannot reach the bounds: they are misses).
static void box_screen_rect(const float* bounds, const LaunchParams& p, uint32_t& wx0, uint32_t& wx1, uint32_t& wy0, uint32_t& wy1)
{
    wx0 = 0;
    wy0 = 0;
    wx1 = p.w;
    wy1 = p.h;
    const double uu = (double)p.U.x * p.U.x + (double)p.U.y * p.U.y + (double)p.U.z * p.U.z;
    const double vv = (double)p.V.x * p.V.x + (double)p.V.y * p.V.y + (double)p.V.z * p.V.z;
    const double ww = (double)p.Wv.x * p.Wv.x + (double)p.Wv.y * p.Wv.y + (double)p.Wv.z * p.Wv.z;
    if (!(uu > 0 && vv > 0 && ww > 0)) return;
    // the three axes must be orthogonal for the projection below (they are for every Camera: Camera.cpp:55-69); else: no culling
    const double uv = (double)p.U.x * p.V.x + (double)p.U.y * p.V.y + (double)p.U.z * p.V.z;
    const double uw = (double)p.U.x * p.Wv.x + (double)p.U.y * p.Wv.y + (double)p.U.z * p.Wv.z;
    const double vw = (double)p.V.x * p.Wv.x + (double)p.V.y * p.Wv.y + (double)p.V.z * p.Wv.z;
    const double tol = 1e-5;
    if (uv * uv > tol * tol * uu * vv || uw * uw > tol * tol * uu * ww || vw * vw > tol * tol * vv * ww) return;
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int k = 0; k < 8; ++k) {
        const double px = bounds[(k & 1) ? 3 : 0] - p.eye.x, py = bounds[(k & 2) ? 4 : 1] - p.eye.y, pz = bounds[(k & 4) ? 5 : 2] - p.eye.z;
        const double a = (px * p.U.x + py * p.U.y + pz * p.U.z) / uu, b = (px * p.V.x + py * p.V.y + pz * p.V.z) / vv;
        const double w = (px * p.Wv.x + py * p.Wv.y + pz * p.Wv.z) / ww;
        if (!(w > 1e-3)) return;  // a corner beside or behind the eye: no useful rectangle
        const double sx = (a / w + 1.0) * 0.5 * p.W, sy = (b / w + 1.0) * 0.5 * p.H;
        x0 = sx < x0 ? sx : x0;
        x1 = sx > x1 ? sx : x1;
        y0 = sy < y0 ? sy : y0;
        y1 = sy > y1 ? sy : y1;
    }
    if (!(x0 <= x1 && y0 <= y1)) return;   // NaN bounds
    // to window pixels, padded, clamped
    auto clampd = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
    wx0 = (uint32_t)clampd(x0 - 2.0 - p.x0, 0.0, p.w);
    wx1 = (uint32_t)clampd(x1 + 3.0 - p.x0, 0.0, p.w);
    wy0 = (uint32_t)clampd(y0 - 2.0 - p.y0, 0.0, p.h);
    wy1 = (uint32_t)clampd(y1 + 3.0 - p.y0, 0.0, p.h);
    if (wx1 <= wx0 || wy1 <= wy0) wx0 = wx1 = wy0 = wy1 = 0;   // the scene is off screen
}

extern "C" {

// diagnostic (tests, tools; not part of include/rtgo.h): what rtgo_set_scene's grid build came to -- {has one, nx, ny, nz, list entries, bytes}
extern "C" int rtgo_debug_grid(rtgo_ctx* c, int32_t out[6])
{
    if (!c || !out) return RTGO_E_INVALID;
    out[0] = c->grid.have ? 1 : 0;
    out[1] = c->grid.gp.nx;
    out[2] = c->grid.gp.ny;
    out[3] = c->grid.gp.nz;
    out[4] = c->grid.entries;
    out[5] = c->grid.n_nodes * 32;
    return RTGO_OK;
}

#ifdef RTGO_CMPWALK
// diagnostic build only (tools/cmp_walks.py): rays on which the canonical and the fast walk disagreed since the last call
extern "C" int rtgo_debug_cmpwalk(rtgo_ctx* c, void* host, size_t bytes)
{
    if (!c || !c->d_cmp) return -1;
    if (rtgo_sync(c)) return -1;
    const size_t n = 256 * 16 * sizeof(float);
    if (hipMemcpy(host, c->d_cmp, n < bytes ? n : bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemset(c->d_cmp, 0, n) != hipSuccess) return -1;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return -1;
    return 0;
}
#endif

#ifdef RTGO_TIMELINE
// diagnostic build only (tools/timeline.py): per-wave records of the last launch; returns the number of waves
extern "C" int rtgo_debug_timeline(rtgo_ctx* c, void* host, size_t bytes)
{
    if (!c || !c->d_timeline) return -1;
    if (rtgo_sync(c)) return -1;
    const size_t n = (size_t)c->timeline_waves * 128;
    if (hipMemcpy(host, c->d_timeline, n < bytes ? n : bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)c->timeline_waves;
}
#endif

// Diagnostic builds whose launches do not produce product results (the whitted tile timer overwrites accum.w and reuses the
// V/T/h counters for ticks; the timeline build records per-wave clocks) answer with a tagged version, so that no test suite or
// driver passes on one of them unnoticed (tests/test_capi_symbols.py asserts the plain number).
#if defined(RTGO_WHITTED_TIMING) || defined(RTGO_TIMELINE) || defined(RTGO_STREAM_STATS)
uint32_t rtgo_abi_version(void) { return RTGO_ABI_VERSION | 0x0D1A6000u; }
#else
uint32_t rtgo_abi_version(void) { return RTGO_ABI_VERSION; }
#endif

const char* rtgo_last_error(const rtgo_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

uint32_t rtgo_local_rows(uint32_t h, uint32_t band_h, uint32_t n_ranks, uint32_t rank)
{
    if (n_ranks <= 1) return h;
    if (band_h == 0) band_h = 4;
    const uint32_t bands = (h + band_h - 1) / band_h;
    uint32_t rows = 0;
    for (uint32_t b = rank; b < bands; b += n_ranks) {
        const uint32_t r0 = b * band_h;
        rows += (r0 + band_h <= h) ? band_h : (h - r0);
    }
    return rows;
}

int rtgo_create(int device, rtgo_ctx** out)
{
    if (!out) return fail(nullptr, RTGO_E_INVALID, "rtgo_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, RTGO_E_NO_DEVICE, "rtgo_create: no HIP device (this library has no CPU path)");
    if (device < 0 || device >= count) return fail(nullptr, RTGO_E_INVALID, "rtgo_create: bad device index");
    RTGO_HIP(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    RTGO_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, RTGO_E_NO_DEVICE,
                    std::string("rtgo_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    rtgo_ctx* c = new rtgo_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    hipError_t err = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    for (int i = 0; i < rtgo_ctx::kEvRing && err == hipSuccess; ++i) {
        err = hipEventCreate(&c->ev_start[i]);
        if (err == hipSuccess) err = hipEventCreate(&c->ev_stop[i]);
    }
    if (err == hipSuccess) err = hipMalloc(&c->d_queue, 2 * kQueues * kQueueStride * sizeof(unsigned int));
    if (err == hipSuccess) err = hipMemset(c->d_queue, 0, 2 * kQueues * kQueueStride * sizeof(unsigned int));
    if (err == hipSuccess) err = hipMalloc(&c->d_counters, 8 * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMemset(c->d_counters, 0, 8 * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMalloc(&c->d_lights, kMaxLights * sizeof(LightRec));
    if (err == hipSuccess) err = hipMemset(c->d_lights, 0, kMaxLights * sizeof(LightRec));
    if (err == hipSuccess) err = hipMalloc(&c->d_meta, 16 * sizeof(int));
    // the megakernel may use most of the 160 KiB LDS of a CU
    const int max_lds = 160 * 1024;
    for (const RenderKernelEntry& e : kRenderKernels)
        if (err == hipSuccess) err = hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    // (the build kernel holds ~58 KB of static LDS; its dynamic part is the fast walk's tree under construction)
    if (err == hipSuccess) err = hipFuncSetAttribute((const void*)build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBuildDynLds);
    if (err == hipSuccess) err = hipFuncSetAttribute((const void*)whitted::sah_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (err == hipSuccess) err = hipFuncSetAttribute((const void*)whitted::build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(whitted::kMaxTriangles * sizeof(unsigned long long)));
    if (err == hipSuccess) err = hipFuncSetAttribute((const void*)whitted::render_kernel<whitted::kAllInL2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess) err = hipFuncSetAttribute((const void*)whitted::render_kernel<whitted::kRecordsInLds>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess) err = hipFuncSetAttribute((const void*)whitted::render_kernel<whitted::kAllInLds>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err == hipSuccess) err = hipDeviceSynchronize();  // the null-stream memsets above must land before any launch
    if (err != hipSuccess) {
        std::string m = std::string("rtgo_create: ") + hipGetErrorString(err);
        rtgo_destroy(c);
        return fail(nullptr, RTGO_E_HIP_BASE + (int)err, m);
    }
    c->stream = c->own_stream;
    *out = c;
    return RTGO_OK;
}

int rtgo_destroy(rtgo_ctx* c)
{
    if (!c) return RTGO_OK;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    (void)hipFree(c->d_prims_in);
    (void)hipFree(c->d_aabb);
    (void)hipFree(c->d_nodes);
    (void)hipFree(c->d_prims);
    (void)hipFree(c->d_fnodes);
    (void)hipFree(c->d_fprims);
    (void)hipFree(c->alt.d_fnodes);
    (void)hipFree(c->alt.d_fprims);
    (void)hipFree(c->d_meta_alt);
    (void)hipFree(c->d_frames);
    (void)hipFree(c->d_meta);
    (void)hipFree(c->d_lights);
    if (c->own_output) {
        (void)hipFree(c->d_accum);
        (void)hipFree(c->d_image);
    }
    (void)hipFree(c->d_queue);
    (void)hipFree(c->d_counters);
    (void)hipFree(c->d_tight);
    (void)hipFree(c->d_mask);
    for (int k = 0; k < 2; ++k) {
        if (c->h_mask[k]) (void)hipHostFree(c->h_mask[k]);
        if (c->mask_copied[k]) (void)hipEventDestroy(c->mask_copied[k]);
    }
    (void)hipFree(c->w_positions);
    (void)hipFree(c->w_normals);
    (void)hipFree(c->w_indices);
    (void)hipFree(c->w_tri_material);
    (void)hipFree(c->w_materials);
    (void)hipFree(c->w_texcoords);
    (void)hipFree(c->w_mat_tex);
    for (void* t : c->w_texels) (void)hipFree(t);
    (void)hipFree(c->w_lights);
    (void)hipFree(c->w_tile_counters);
    (void)hipFree(c->w_nodes);
    (void)hipFree(c->w_recs);
    (void)hipFree(c->w_tris);
    (void)hipFree(c->w_qrecs);
    (void)hipFree(c->w_tidx);
    (void)hipFree(c->w_scratch);
    for (int i = 0; i < rtgo_ctx::kEvRing; ++i) {
        if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
        if (c->ev_stop[i]) (void)hipEventDestroy(c->ev_stop[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RTGO_OK;
}

int rtgo_set_stream(rtgo_ctx* c, void* hip_stream)
{
    if (!c) return RTGO_E_INVALID;
    // launches are ordered by the stream they run on (accumulation buffer, the two alternating sets of queue heads): finish the
    // work on the old stream before moving
    const int rc = rtgo_sync(c);
    if (rc) return rc;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RTGO_OK;
}

// The uniform grid of rtgo::fast_grid over structure 0's small primitives (fprims [0, n_small)), from the boxes the fast walk culls
// with (c->tight: build_kernel's, SBT order).  Every box is grown by `pad` before it is binned, and fast_grid stops `pad / 2` (in t)
// late: the walk's own rounding (entry point, 96 accumulated steps: <= ~2e-5 of the rays' reach) stays an order of magnitude inside.
// Cells: <= 32 per axis; table, cell records and lists within 40 KB of LDS; no grid for fewer than 64 small primitives (RTGO_GRID_MIN) or when
// every resolution with at least half as many cells as primitives lists more than 3 entries per primitive (RTGO_GRID_MAX_DUP; a few big shapes among small ones: the tree's job).
static int build_grid(rtgo_ctx* c, uint32_t n)
{
    c->grid.have = false;
    const int ns = c->n_small;
    if (ns < (int)env_uint("RTGO_GRID_MIN", 64) || std::getenv("RTGO_NO_GRID")) return RTGO_OK;
    std::vector<float4> fp((size_t)n * 4);
    RTGO_HIP(c, hipMemcpyAsync(fp.data(), c->d_fprims, fp.size() * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<const float*> box((size_t)ns);
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f}, scene_reach = 0.0f;
    for (int pos = 0; pos < ns; ++pos) {
        int orig;
        std::memcpy(&orig, &fp[4 * (size_t)pos + 3].y, sizeof orig);
        if (orig < 0 || orig >= (int)n) return fail(c, RTGO_E_UNSUPPORTED, "rtgo_set_scene: fast-walk record without a primitive");
        box[pos] = &c->tight[6 * (size_t)orig];
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::fmin(lo[a], box[pos][a]);
            hi[a] = std::fmax(hi[a], box[pos][3 + a]);
        }
    }
    for (int a = 0; a < 6; ++a) scene_reach = std::fmax(scene_reach, std::fabs(c->bounds[a]));
    const float reach_max = 4.0f * scene_reach;
    float ext[3], max_ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
        ext[a] = hi[a] - lo[a];
        max_ext = std::fmax(max_ext, ext[a]);
    }
    if (!(max_ext > 0.0f)) return RTGO_OK;
    // Resolution: the (nx, ny, nz) that minimises the classic cost of a grid walk -- a ray meets cells in proportion to their surface
    // area, pays one step per cell and one test per list entry: cost = A_cell (n_cells + kTest entries) -- over every resolution
    // up to 32 per axis, with the entries counted exactly (a primitive's span of cells is separable per axis).  The entries term is what
    // matters: a resolution that cuts through the shapes lists them two to eight times (balls, profiles/r03r: 16 x 4 x 16, one sphere per
    // column, 321 entries, 10.1 ms; 13 x 6 x 13, 754 entries, 13.8 ms; 32 x 1 x 32, 1024 entries, 19.1 ms; the tree 13.5).  kTest: 1.7 fitted there; 1.0 since the walk tests a cell's box before its list (16 x 3 x 16, 9.7 ms).
    const float kTest = env_float("RTGO_GRID_KTEST", 1.0f), max_dup = env_float("RTGO_GRID_MAX_DUP", 3.0f);
    const float pad0 = 2e-3f * (max_ext / 8.0f) + 1e-4f * reach_max;   // (the pad of the binning below depends on the cell size: close enough for counting)
    std::vector<uint8_t> span[3];
    for (int a = 0; a < 3; ++a) {
        span[a].assign((size_t)33 * ns, 1);
        for (int m = 1; m <= 32; ++m) {
            const float g0 = lo[a] - 2.0f * pad0, ics = (float)m / (ext[a] + 4.0f * pad0);
            for (int pos = 0; pos < ns; ++pos) {
                int c0 = (int)std::floor((box[pos][a] - pad0 - g0) * ics), c1 = (int)std::floor((box[pos][3 + a] + pad0 - g0) * ics);
                c0 = c0 < 0 ? 0 : c0;
                c1 = c1 > m - 1 ? m - 1 : c1;
                span[a][(size_t)m * ns + pos] = (uint8_t)(c1 - c0 + 1);
            }
        }
    }
    int dim[3] = {1, 1, 1};
    double best_cost = 1e300;
    std::vector<uint32_t> sxy((size_t)ns);
    for (int mx = 1; mx <= 32; ++mx)
        for (int my = 1; my <= 32; ++my) {
            if ((mx + 2) * (my + 2) * 3 * 4 > 32 * 1024) continue;
            for (int pos = 0; pos < ns; ++pos) sxy[pos] = (uint32_t)span[0][(size_t)mx * ns + pos] * span[1][(size_t)my * ns + pos];
            for (int mz = 1; mz <= 32; ++mz) {
                const size_t words = (size_t)(mx + 2) * (my + 2) * (mz + 2);
                if (words * 4 > 30 * 1024) break;
                size_t entries = 0;
                const uint8_t* sz = &span[2][(size_t)mz * ns];
                for (int pos = 0; pos < ns; ++pos) entries += (size_t)sxy[pos] * sz[pos];
                const size_t listing = entries < (size_t)mx * my * mz ? entries : (size_t)mx * my * mz;   // (at most this many cells carry a record)
                if ((float)entries > max_dup * (float)ns || words * 4 + listing * 32 + entries * 2 > 38 * 1024) continue;
                if (2 * mx * my * mz < ns) continue;   // (fewer cells than half the primitives: lists, not a grid)
                const double cx = (ext[0] + 4.0 * pad0) / mx, cy = (ext[1] + 4.0 * pad0) / my, cz = (ext[2] + 4.0 * pad0) / mz;
                const double cost = 2.0 * (cx * cy + cy * cz + cx * cz) * ((double)mx * my * mz + (double)kTest * (double)entries);
                if (cost < best_cost) {
                    best_cost = cost;
                    dim[0] = mx; dim[1] = my; dim[2] = mz;
                }
            }
        }
    if (best_cost >= 1e300) return RTGO_OK;
    if (const char* want = std::getenv("RTGO_GRID_DIMS")) {   // "nx,ny,nz": experiments
        int d[3];
        if (std::sscanf(want, "%d,%d,%d", &d[0], &d[1], &d[2]) == 3)
            for (int a = 0; a < 3; ++a) dim[a] = d[a] < 1 ? 1 : (d[a] > 32 ? 32 : d[a]);
    }
    rtgo::GridParams g = {};
    float max_cs = 0.0f;
    for (int a = 0; a < 3; ++a) max_cs = std::fmax(max_cs, ext[a] / (float)dim[a]);
    const float pad = 2e-3f * max_cs + 1e-4f * reach_max;
    float gmin[3], gcs[3], gics[3];
    for (int a = 0; a < 3; ++a) {
        gmin[a] = lo[a] - 2.0f * pad;
        gcs[a] = (ext[a] + 4.0f * pad) / (float)dim[a];
        gics[a] = 1.0f / gcs[a];
    }
    g.min_x = gmin[0]; g.min_y = gmin[1]; g.min_z = gmin[2];
    g.cs_x = gcs[0]; g.cs_y = gcs[1]; g.cs_z = gcs[2];
    g.ics_x = gics[0]; g.ics_y = gics[1]; g.ics_z = gics[2];
    g.nx = dim[0]; g.ny = dim[1]; g.nz = dim[2];
    // the table carries a border of empty cells (fast_grid steps into it when it leaves the grid)
    const int NX = dim[0] + 2, NY = dim[1] + 2, NZ = dim[2] + 2;
    g.n_cells = NX * NY * NZ;
    g.margin = 0.5f * pad;
    std::vector<std::vector<uint16_t>> lists((size_t)g.n_cells);
    size_t total = 0;
    for (int pos = 0; pos < ns; ++pos) {
        int a0[3], a1[3];
        for (int a = 0; a < 3; ++a) {
            a0[a] = (int)std::floor((box[pos][a] - pad - gmin[a]) * gics[a]);
            a1[a] = (int)std::floor((box[pos][3 + a] + pad - gmin[a]) * gics[a]);
            a0[a] = a0[a] < 0 ? 0 : a0[a];
            a1[a] = a1[a] > dim[a] - 1 ? dim[a] - 1 : a1[a];
        }
        for (int z = a0[2]; z <= a1[2]; ++z)
            for (int y = a0[1]; y <= a1[1]; ++y)
                for (int x = a0[0]; x <= a1[0]; ++x) {
                    lists[((size_t)(z + 1) * NY + (y + 1)) * NX + (x + 1)].push_back((uint16_t)pos);
                    ++total;
                }
    }
    if ((float)total > (max_dup + 0.5f) * (float)ns || total > 60000) return RTGO_OK;   // (the binning's pad is a little larger than the count's)
    // image: [table, one word per cell][records, 2 float4 per listing cell][items, 16 bit each], each part on a 16-byte boundary
    size_t n_rec = 0;
    for (const std::vector<uint16_t>& l : lists) n_rec += l.empty() ? 0 : 1;
    const size_t table_bytes = ((size_t)g.n_cells * 4 + 15) / 16 * 16, rec_bytes = n_rec * 32;
    const size_t bytes = (table_bytes + rec_bytes + total * 2 + 31) / 32 * 32;
    if (bytes > 40 * 1024) return RTGO_OK;
    g.rec_off4 = (int)(table_bytes / 16);
    g.items_off4 = (int)((table_bytes + rec_bytes) / 16);
    std::vector<unsigned char> img(bytes, 0);
    uint32_t* cells = reinterpret_cast<uint32_t*>(img.data());
    float* recs = reinterpret_cast<float*>(img.data() + table_bytes);
    uint16_t* items = reinterpret_cast<uint16_t*>(img.data() + table_bytes + rec_bytes);
    size_t at = 0, rec = 0;
    for (int k = 0; k < g.n_cells; ++k) {
        const std::vector<uint16_t>& l = lists[(size_t)k];
        if (l.empty()) continue;
        cells[k] = (uint32_t)(rec + 1);
        float* q = recs + 8 * rec;
        for (int a = 0; a < 3; ++a) {
            q[a] = 1e30f;
            q[4 + a] = -1e30f;
        }
        for (uint16_t v : l) {
            for (int a = 0; a < 3; ++a) {   // the box around what the cell lists: the shapes' own boxes grown by the pad ...
                q[a] = std::fmin(q[a], box[v][a] - pad);
                q[4 + a] = std::fmax(q[4 + a], box[v][3 + a] + pad);
            }
            items[at++] = v;
        }
        {   // ... cut to the cell, itself grown by the pad: a hit point lies (within the walk's rounding) in a cell the walk visits, that
            // cell lists the shape, and the point is inside this box of it -- so the test happens there at the latest
            const int kx = k % NX - 1, ky = (k / NX) % NY - 1, kz = k / (NX * NY) - 1;
            const int kk[3] = {kx, ky, kz};
            for (int a = 0; a < 3; ++a) {
                q[a] = std::fmax(q[a], gmin[a] + gcs[a] * (float)kk[a] - pad);
                q[4 + a] = std::fmin(q[4 + a], gmin[a] + gcs[a] * (float)(kk[a] + 1) + pad);
            }
        }
        const uint32_t fc = (uint32_t)(at - l.size()) | ((uint32_t)l.size() << 16);
        std::memcpy(&q[3], &fc, 4);
        q[7] = 0.0f;
        ++rec;
    }
    RTGO_HIP(c, hipMalloc(&c->grid.d, bytes));
    RTGO_HIP(c, hipMemcpyAsync(c->grid.d, img.data(), bytes, hipMemcpyHostToDevice, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    c->grid.n_nodes = (int)(bytes / 32);
    c->grid.entries = (int)total;
    c->grid.gp = g;
    c->grid.reach_max = reach_max;
    c->grid.have = true;
    if (std::getenv("RTGO_DEBUG"))
        std::fprintf(stderr, "rtgo_set_scene: grid %d x %d x %d over %d primitives, %zu list entries, %zu bytes, pad %g, for rays within %g\n", dim[0], dim[1], dim[2], ns,
                     total, bytes, pad, reach_max);
    return RTGO_OK;
}

int rtgo_set_scene(rtgo_ctx* c, const rtgo_prim* prims, const rtgo_aabb* aabbs, uint32_t n)
{
    if (!c || !prims) return fail(c, RTGO_E_INVALID, "rtgo_set_scene: NULL argument");
    if (n == 0 || n > RTGO_MAX_PRIMS)
        return fail(c, RTGO_E_UNSUPPORTED, "rtgo_set_scene: primitive count must be in [1, " + std::to_string(RTGO_MAX_PRIMS) + "]");
    for (uint32_t i = 0; i < n; ++i) {
        const rtgo_prim& q = prims[i];
        if (q.type > RTGO_SPHERE) return fail(c, RTGO_E_INVALID, "rtgo_set_scene: unknown primitive type");
        // the intersection programs work in object space through M^-1 (kernel.cu:125-135): M must be finite and invertible
        bool finite = std::isfinite(q.specularity);
        for (int k = 0; k < 16; ++k) finite = finite && std::isfinite(q.model[k]);
        for (int k = 0; k < 3; ++k) finite = finite && std::isfinite(q.kd[k]) && std::isfinite(q.kr[k]) && std::isfinite(q.Le[k]);
        const double a = q.model[0], b = q.model[1], d3 = q.model[2], e = q.model[4], g = q.model[5], h = q.model[6], k2 = q.model[8],
                     l = q.model[9], m = q.model[10];
        const double det = a * (g * m - h * l) - b * (e * m - h * k2) + d3 * (e * l - g * k2);
        if (!finite || !std::isfinite(det) || std::fabs(det) < 1e-30)
            return fail(c, RTGO_E_INVALID, "rtgo_set_scene: primitive " + std::to_string(i) + " has a non-finite or singular model matrix / material");
        if (aabbs) {
            const rtgo_aabb& bb = aabbs[i];
            if (!(bb.minX <= bb.maxX && bb.minY <= bb.maxY && bb.minZ <= bb.maxZ) || !std::isfinite(bb.minX + bb.minY + bb.minZ + bb.maxX + bb.maxY + bb.maxZ))
                return fail(c, RTGO_E_INVALID, "rtgo_set_scene: box " + std::to_string(i) + " is empty or not finite");
        }
    }
    RTGO_HIP(c, hipSetDevice(c->device));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_prims_in);
    (void)hipFree(c->d_aabb);
    (void)hipFree(c->d_nodes);
    (void)hipFree(c->d_prims);
    (void)hipFree(c->d_fnodes);
    (void)hipFree(c->d_fprims);
    (void)hipFree(c->alt.d_fnodes);
    (void)hipFree(c->alt.d_fprims);
    c->alt = rtgo_ctx::FastTree();
    c->have_alt = false;
    (void)hipFree(c->grid.d);
    c->grid = rtgo_ctx::Grid();
    (void)hipFree(c->d_frames);
    (void)hipFree(c->d_tight);
    c->d_frames = nullptr;
    c->d_tight = nullptr;
    c->mask_key.clear();
    c->trial = rtgo_ctx::Trial();
    c->d_fnodes = nullptr;
    c->d_fprims = nullptr;
    c->d_prims_in = nullptr;
    c->d_aabb = nullptr;
    c->d_nodes = nullptr;
    c->d_prims = nullptr;
    c->n_prims = 0;
    RTGO_HIP(c, hipMalloc(&c->d_prims_in, n * sizeof(PrimIn)));
    RTGO_HIP(c, hipMalloc(&c->d_aabb, n * 6 * sizeof(float)));
    RTGO_HIP(c, hipMalloc(&c->d_nodes, (2 * n - 1) * 2 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->d_prims, n * 6 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->d_fnodes, (2 * n - 1) * 2 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->d_fprims, n * 4 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->d_frames, n * 2 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->d_tight, n * 6 * sizeof(float)));
    if (const char* k = std::getenv("RTGO_LEAF_BUDGET")) {  // tuning knob for experiments; results do not depend on it
        const int v = std::atoi(k);
        if (v >= 0 && v <= 16 * kMaxPrims) c->leaf_budget = v;
    }
    RTGO_HIP(c, hipMemcpyAsync(c->d_prims_in, prims, n * sizeof(PrimIn), hipMemcpyHostToDevice, c->stream));
    if (aabbs) RTGO_HIP(c, hipMemcpyAsync(c->d_aabb, aabbs, n * sizeof(rtgo_aabb), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(build_kernel, dim3(1), dim3(kMaxPrims), kBuildDynLds, c->stream, c->d_prims_in, c->d_aabb, aabbs ? 1 : 0, (int)n,
                       c->d_nodes, c->d_prims, c->d_fnodes, c->d_fprims, c->leaf_budget,
                       (float)env_uint("RTGO_BIG_PERCENT", 36) * 0.01f, c->d_meta, c->d_tight, std::getenv("RTGO_NO_CUBOID") ? 0 : 1, c->d_frames);
    RTGO_HIP(c, hipGetLastError());
    int meta[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    RTGO_HIP(c, hipMemcpyAsync(meta, c->d_meta, sizeof meta, hipMemcpyDeviceToHost, c->stream));
    c->tight.assign((size_t)n * 6, 0.0f);
    RTGO_HIP(c, hipMemcpyAsync(c->tight.data(), c->d_tight, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    const int depth = meta[0];
    c->lbvh_depth = depth;
    c->fast_depth = meta[1];
    c->n_small = meta[2];
    c->n_big_pairs = meta[9] & 0xFF;
    c->list_cub = meta[9] >> 8;
    c->cuboid_groups = meta[13] + (c->list_cub ? 1 : 0);
    c->tree_spheres = (c->n_small > 0 && meta[14] == (1 << 3) && !std::getenv("RTGO_NO_SPHERE_LEAVES")) ? 1 : 0;   // (type 3 = sphere)
    std::memcpy(&c->cub_a, &meta[11], sizeof(float));
    std::memcpy(&c->cub_b, &meta[12], sizeof(float));
    c->n_fnodes = meta[10];
    if (c->n_fnodes < 0 || c->n_fnodes > 2 * (int)n - 1 || (c->n_small > 0 && c->n_fnodes < 1))
        return fail(c, RTGO_E_UNSUPPORTED, "rtgo_set_scene: the fast walk's tree has " + std::to_string(c->n_fnodes) + " nodes");
    std::memcpy(c->bounds, &meta[3], sizeof c->bounds);
    if (!std::getenv("RTGO_BIG_PERCENT") && !std::getenv("RTGO_ONE_TREE")) {
        // the alternative structure: big_frac 15 % (the canonical outputs, boxes and frames are rewritten with the same values)
        RTGO_HIP(c, hipMalloc(&c->alt.d_fnodes, (2 * n - 1) * 2 * sizeof(float4)));
        RTGO_HIP(c, hipMalloc(&c->alt.d_fprims, n * 4 * sizeof(float4)));
        if (!c->d_meta_alt) RTGO_HIP(c, hipMalloc(&c->d_meta_alt, 16 * sizeof(int)));
        hipLaunchKernelGGL(build_kernel, dim3(1), dim3(kMaxPrims), kBuildDynLds, c->stream, c->d_prims_in, c->d_aabb, 1, (int)n,
                           c->d_nodes, c->d_prims, c->alt.d_fnodes, c->alt.d_fprims, c->leaf_budget, 0.15f, c->d_meta_alt, c->d_tight,
                           std::getenv("RTGO_NO_CUBOID") ? 0 : 1, c->d_frames);
        RTGO_HIP(c, hipGetLastError());
        int m2[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        RTGO_HIP(c, hipMemcpyAsync(m2, c->d_meta_alt, sizeof m2, hipMemcpyDeviceToHost, c->stream));
        RTGO_HIP(c, hipStreamSynchronize(c->stream));
        rtgo_ctx::FastTree& a = c->alt;
        a.fast_depth = m2[1];
        a.n_small = m2[2];
        a.n_big_pairs = m2[9] & 0xFF;
        a.list_cub = m2[9] >> 8;
        a.cuboid_groups = m2[13] + (a.list_cub ? 1 : 0);
        a.tree_spheres = (a.n_small > 0 && m2[14] == (1 << 3) && !std::getenv("RTGO_NO_SPHERE_LEAVES")) ? 1 : 0;
        std::memcpy(&a.cub_a, &m2[11], sizeof(float));
        std::memcpy(&a.cub_b, &m2[12], sizeof(float));
        a.n_fnodes = m2[10];
        const bool sane = m2[0] == meta[0] && a.n_fnodes >= 0 && a.n_fnodes <= 2 * (int)n - 1 && !(a.n_small > 0 && a.n_fnodes < 1);
        // (the same split of primitives = the same structure: nothing to try)
        c->have_alt = sane && !(a.n_small == c->n_small && a.n_fnodes == c->n_fnodes && a.n_big_pairs == c->n_big_pairs && a.list_cub == c->list_cub);
    }
    if (const int rc = build_grid(c, n)) return rc;
    if (std::getenv("RTGO_DEBUG"))
        std::fprintf(stderr, "rtgo_set_scene: %d primitives, %d in the fast walk's tree (%d nodes, depth %d), %d up front (%d pairs, cuboid certificate %d), %d cuboid leaves, margin coefficients %g %g, canonical LBVH depth %d\n",
                     (int)n, c->n_small, c->n_fnodes, c->fast_depth, (int)n - c->n_small, c->n_big_pairs, c->list_cub, meta[13], c->cub_a, c->cub_b, depth);
    if (depth > kStackDepth)
        return fail(c, RTGO_E_UNSUPPORTED, "rtgo_set_scene: LBVH depth " + std::to_string(depth) + " exceeds the per-lane LDS stack (" +
                                               std::to_string(kStackDepth) + ")");
    c->n_prims = n;
    c->quadrics.clear();
    for (uint32_t i = 0; i < n; ++i) {
        const rtgo_prim& q = prims[i];
        if (q.type != RTGO_SPHERE && q.type != RTGO_CYLINDER) continue;
        // axis scales = column norms of the model matrix' 3x3 (exact for translate * rotate * scale; a cylinder's quadratic lives in x, z)
        double s[3];
        for (int k = 0; k < 3; ++k) s[k] = std::sqrt((double)q.model[k] * q.model[k] + (double)q.model[4 + k] * q.model[4 + k] + (double)q.model[8 + k] * q.model[8 + k]);
        double smin = s[0] < s[2] ? s[0] : s[2], smax = s[0] > s[2] ? s[0] : s[2];
        if (q.type == RTGO_SPHERE) {
            smin = s[1] < smin ? s[1] : smin;
            smax = s[1] > smax ? s[1] : smax;
        }
        rtgo_ctx::Quadric e;
        e.c[0] = q.model[3];
        e.c[1] = q.model[7];
        e.c[2] = q.model[11];
        e.w = (float)(smax / (smin * smin));
        c->quadrics.push_back(e);
    }
    return RTGO_OK;
}

int rtgo_set_camera(rtgo_ctx* c, const float eye[3], const float U[3], const float V[3], const float W[3])
{
    if (!c || !eye || !U || !V || !W) return fail(c, RTGO_E_INVALID, "rtgo_set_camera: NULL argument");
    c->eye = v3{eye[0], eye[1], eye[2]};
    c->U = v3{U[0], U[1], U[2]};
    c->V = v3{V[0], V[1], V[2]};
    c->W = v3{W[0], W[1], W[2]};
    c->have_camera = true;
    return RTGO_OK;
}

int rtgo_set_background(rtgo_ctx* c, const float rgb[3])
{
    if (!c || !rgb) return fail(c, RTGO_E_INVALID, "rtgo_set_background: NULL argument");
    c->bg = v3{rgb[0], rgb[1], rgb[2]};
    return RTGO_OK;
}

int rtgo_set_lights(rtgo_ctx* c, const rtgo_light* lights, int n)
{
    if (!c || n < 0 || (n > 0 && !lights)) return fail(c, RTGO_E_INVALID, "rtgo_set_lights: bad argument");
    if (n > RTGO_MAX_LIGHTS) n = RTGO_MAX_LIGHTS;  // Renderer::WriteLights copies at most MAX_LIGHTS (renderer.cpp:661)
    RTGO_HIP(c, hipSetDevice(c->device));
    if (n > 0) RTGO_HIP(c, hipMemcpyAsync(c->d_lights, lights, n * sizeof(rtgo_light), hipMemcpyHostToDevice, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    c->n_lights = n;
    return RTGO_OK;
}

int rtgo_resize(rtgo_ctx* c, size_t pixels)
{
    if (!c || pixels == 0) return fail(c, RTGO_E_INVALID, "rtgo_resize: bad argument");
    RTGO_HIP(c, hipSetDevice(c->device));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    if (c->own_output) {
        (void)hipFree(c->d_accum);
        (void)hipFree(c->d_image);
    }
    c->d_accum = nullptr;
    c->d_image = nullptr;
    c->own_output = false;
    c->pixels = 0;
    RTGO_HIP(c, hipMalloc(&c->d_accum, pixels * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->d_image, pixels * sizeof(uchar4)));
    RTGO_HIP(c, hipMemsetAsync(c->d_accum, 0, pixels * sizeof(float4), c->stream));
    RTGO_HIP(c, hipMemsetAsync(c->d_image, 0, pixels * sizeof(uchar4), c->stream));
    c->own_output = true;
    c->pixels = pixels;
    return RTGO_OK;
}

int rtgo_bind_output(rtgo_ctx* c, void* d_accum, void* d_image, size_t pixels)
{
    if (!c || !d_accum || !d_image || pixels == 0) return fail(c, RTGO_E_INVALID, "rtgo_bind_output: bad argument");
    if (((uintptr_t)d_accum & 15u) || ((uintptr_t)d_image & 3u))
        return fail(c, RTGO_E_INVALID, "rtgo_bind_output: accum must be 16-byte aligned, image 4-byte aligned");
    if (c->own_output) {
        (void)hipFree(c->d_accum);
        (void)hipFree(c->d_image);
    }
    c->d_accum = (float4*)d_accum;
    c->d_image = (uchar4*)d_image;
    c->own_output = false;
    c->pixels = pixels;
    return RTGO_OK;
}

int rtgo_launch(rtgo_ctx* c, const rtgo_frame* f)
{
    if (!c || !f) return fail(c, RTGO_E_INVALID, "rtgo_launch: NULL argument");
    if (c->n_prims == 0) return fail(c, RTGO_E_STATE, "rtgo_launch: no scene (call rtgo_set_scene)");
    if (!c->have_camera) return fail(c, RTGO_E_STATE, "rtgo_launch: no camera (call rtgo_set_camera)");
    if (!c->d_accum || !c->d_image) return fail(c, RTGO_E_STATE, "rtgo_launch: no output (call rtgo_resize or rtgo_bind_output)");
    if (f->image_width == 0 || f->image_height == 0 || f->sqrt_spp <= 0)
        return fail(c, RTGO_E_INVALID, "rtgo_launch: image size and sqrt_spp must be positive");
    if (f->max_trace_depth < 0 || f->max_trace_depth > kMaxLevels)
        return fail(c, RTGO_E_UNSUPPORTED, "rtgo_launch: max_trace_depth must be in [0, 5] (reference uses 5, renderer.cpp:616)");
    if (!f->path_tracing && c->n_lights < 1)
        return fail(c, RTGO_E_STATE, "rtgo_launch: distributed mode needs at least one surface light");
    LaunchParams p;
    std::memset(&p, 0, sizeof p);
    p.W = f->image_width;
    p.H = f->image_height;
    p.x0 = f->x0;
    p.y0 = f->y0;
    p.w = f->w ? f->w : f->image_width;
    p.h = f->h ? f->h : f->image_height;
    if ((uint64_t)p.x0 + p.w > p.W || (uint64_t)p.y0 + p.h > p.H) return fail(c, RTGO_E_INVALID, "rtgo_launch: window outside the image");
    p.band_h = f->band_h ? f->band_h : 4;
    p.n_ranks = f->n_ranks ? f->n_ranks : 1;
    p.rank = f->rank;
    if (p.rank >= p.n_ranks) return fail(c, RTGO_E_INVALID, "rtgo_launch: rank >= n_ranks");
    p.local_rows = rtgo_local_rows(p.h, p.band_h, p.n_ranks, p.rank);
    if ((size_t)p.local_rows * p.w > c->pixels) return fail(c, RTGO_E_INVALID, "rtgo_launch: output buffer too small for this window");
    p.eye = c->eye;
    p.U = c->U;
    p.V = c->V;
    p.Wv = c->W;
    p.bg = c->bg;
    const bool path = f->path_tracing != 0, stats = f->collect_stats != 0;
    bool frames = false;   // (set once the walk is chosen)
    // scheduling: units of 64 paths = the N*N samples of `unit_px` neighbouring pixels of one row; the queue hands out STRIPS of
    // `grab` units side by side (<= 64 pixels) from the rectangle that can contain geometry.  Strips are long when there is
    // plenty of work (their pixel seeds are hashed once per strip) and short when units are scarce (small windows, one GPU's
    // share of a tiled frame), so that every resident wave still gets >= ~32 turns
    // (the last strips in flight set the tail of the launch: cornell 1080p spp 16 runs 6 % faster on 1-unit strips than on 4-unit ones).
    const uint32_t nn = (uint32_t)f->sqrt_spp * (uint32_t)f->sqrt_spp;
    const uint32_t unit_px = 64u / (nn < (uint32_t)kSamplesPerPass ? nn : (uint32_t)kSamplesPerPass);
    {
        // the same float additions, in the same order, as the kernel's in-order sum over samples that all miss (kernel.cu:232-237)
        volatile float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (uint32_t k = 0; k < nn; ++k) {
            sx = sx + c->bg.x;
            sy = sy + c->bg.y;
            sz = sz + c->bg.z;
        }
        const float inv = 1.0f / (float)nn;
        p.bg_pixel = v3{sx * inv, sy * inv, sz * inv};
    }
    uint32_t wx0 = 0, wx1 = p.w, wy0 = 0, wy1 = p.h;
    // collect_stats 1: the instrumented kernel traces every pixel (V, T, h over ALL rays); 2: it culls like the timed
    // kernel, so that the counters describe the traversed rays only
    const bool cull = f->collect_stats != 1 && !std::getenv("RTGO_NO_CULL");
    if (cull) box_screen_rect(c->bounds, p, wx0, wx1, wy0, wy1);
    // The fast walk's tight boxes carry 1e-3 of padding against the rounding of the intersection programs, which grows with
    // the coordinates involved (~1e-7 of them for a rectangle's hit point).  Beyond 500 units -- the reference's scenes stay
    // within 20, its camera at 14 -- the launch takes the canonical walk instead: slower, and equal to it by definition.
    bool canon = stats;
    {
        float reach = 0.0f;
        for (int k = 0; k < 6; ++k) reach = std::fabs(c->bounds[k]) > reach ? std::fabs(c->bounds[k]) : reach;
        const float e[3] = {p.eye.x, p.eye.y, p.eye.z};
        for (int k = 0; k < 3; ++k) reach = std::fabs(e[k]) > reach ? std::fabs(e[k]) : reach;
        // ... and a sphere's or cylinder's reported hit leaves its surface as the ray origin recedes: b^2 - 4ac cancels to the last
        // bits of b^2 ~ D^2 / s^4, i.e. the hit lies up to ~2^-25 D^2 smax / smin^2 off the surface (D: origin to the primitive, s: its
        // axis scales) -- outside the reference's own box when that exceeds AABB_EPSILON.  From there on no two traversals agree on
        // grazing rays (the canonical LBVH culls such a hit by the primitive's box, a multi-primitive leaf's box lets it through, and
        // OptiX promises neither), so what is bounded is Q = max over quadrics of D^2 smax / smin^2, D over the eye and the scene's
        // tight bounds (where bounce rays start).  Thresholds: kGuardReach / kGuardQuadric, set from tools/fuzz_farfield.py's table
        // (profiles/r03a) with the safety factors stated at their definition.
        float quad = 0.0f;
        for (const rtgo_ctx::Quadric& qd : c->quadrics) {
            float d2 = 0.0f, e2 = 0.0f;
            for (int k = 0; k < 3; ++k) {
                const float lo = std::fabs(c->bounds[k] - qd.c[k]), hi = std::fabs(c->bounds[3 + k] - qd.c[k]);
                const float far_k = lo > hi ? lo : hi;
                d2 += far_k * far_k;
                e2 += (e[k] - qd.c[k]) * (e[k] - qd.c[k]);
            }
            const float q2 = (d2 > e2 ? d2 : e2) * qd.w;
            quad = q2 > quad ? q2 : quad;
        }
        c->guard_reach = reach;
        c->guard_quadric = quad;
        static const float guard_reach_max = env_float("RTGO_GUARD_REACH", kGuardReach), guard_quadric_max = env_float("RTGO_GUARD_QUADRIC", kGuardQuadric);
        if (!(reach <= guard_reach_max) || !(quad <= guard_quadric_max)) canon = true;
    }
    // ---- which loop and which structure (rtgo_ctx::Trial).  More than 16 spp = several passes per pixel: the streaming variant
    // (render_kernel, STREAM) lets a lane start its next sample when its path has ended instead of waiting for the wave's longest path,
    // pass after pass; and where rtgo_set_scene's two builds differ, either structure can be the faster one.  Candidate k = loop (k & 1:
    // 0 = streaming when there is a choice) | structure (k >> 1 when both loops are candidates, else k).
    bool stream = false;
    int structure = 0;   // 0 / 1: the trees of 36 % / 15 %, 2: the grid
    unsigned char trial_tag = 0;
    if (!canon) {
        const bool multi_pass = passes_of(nn) > 1;
        const char* force_loop = std::getenv("RTGO_STREAM");   // "0" / "1": experiment and test knobs, no trial over that dimension
        const char* force_tree = std::getenv("RTGO_TREE");     // "0" / "1" / "2": the 36 % tree / the 15 % tree / the grid (when the scene has it)
        // (the grid: where rtgo_set_scene built one, for rays that start within the reach its pad was sized for, and in the instantiations
        // that exist -- not the flat-primitives one)
        const bool flat_only = path && c->quadrics.empty() && !std::getenv("RTGO_NO_FRAMES");
        const bool grid_ok = c->grid.have && c->guard_reach <= c->grid.reach_max && !flat_only;
        int structs[3], n_structs = 0;
        structs[n_structs++] = 0;
        if (c->have_alt) structs[n_structs++] = 1;
        if (grid_ok) structs[n_structs++] = 2;
        if (force_tree) {
            const int want = force_tree[0] - '0';
            structure = 0;
            for (int k = 0; k < n_structs; ++k)
                if (structs[k] == want) structure = want;
            n_structs = 1;
            structs[0] = structure;
        }
        const bool loops = multi_pass && !force_loop;
        if (multi_pass && force_loop) stream = force_loop[0] != '0';
        const int n_loops = loops ? 2 : 1;
        const int n_cand = n_loops * n_structs;
        auto decode = [&](int k) {
            if (loops) stream = (k % n_loops) == 0;
            structure = structs[k / n_loops];
        };
        if (n_cand > 1) {
            rtgo_ctx::Trial& t = c->trial;
            const std::vector<uint32_t> key = {p.W, p.H, p.x0, p.y0, p.w, p.h, p.band_h, p.n_ranks, p.rank, nn, (uint32_t)path, (uint32_t)f->max_trace_depth,
                                               (uint32_t)(f->use_ambient != 0), (uint32_t)n_cand, (uint32_t)stream, (uint32_t)structure, (uint32_t)grid_ok};
            if (key != t.key) {
                // (event tags of an unfinished trial of the old key stay where they are: they are counted into the old minima nobody reads)
                t = rtgo_ctx::Trial();
                t.key = key;
                t.n_cand = n_cand;
                for (unsigned char& tag : c->ev_tag) tag = 0;
            }
            if (t.choice < 0 && t.issued >= 2 * n_cand) {
                // all are in flight or done: WAIT for them.  A caller that enqueues a whole job without synchronising (bench.py's spin-up,
                // a batch render) would otherwise run it to the end on whatever stands in for an undecided trial -- profiles/r03p caught
                // 90 of 100 launches of C4 on its slowest candidate that way.  One stall of at most 2 * n_cand launches per job.
                while (t.done < 2 * n_cand && c->ev_pending > 0) {
                    const int rc = harvest_events(c, 1);
                    if (rc) return rc;
                }
                if (t.done >= 2 * n_cand) {
                    t.choice = 0;
                    for (int k = 1; k < n_cand; ++k)
                        if (t.best[k] < t.best[t.choice]) t.choice = k;
                }
            }
            if (t.choice >= 0) decode(t.choice);
            else if (t.issued < 2 * n_cand) {
                const int k = t.issued % n_cand;
                decode(k);
                trial_tag = (unsigned char)(k + 1);
                t.issued++;
            } else decode(0);   // (the trial's events were lost to a key change: start over with the first candidate)
        } else if (n_structs == 1) structure = structs[0];
    }
#ifdef RTGO_CMPWALK
    if (canon)   // (diagnostic build: the instrumented launch also runs the pinned fast structure on every ray, rtgo_ray_trace.inc)
        if (const char* want = std::getenv("RTGO_TREE")) {
            if (want[0] == '1' && c->have_alt) structure = 1;
            if (want[0] == '2' && c->grid.have && c->guard_reach <= c->grid.reach_max) structure = 2;
        }
#endif
    const bool use_alt = structure == 1, use_grid = structure == 2;
    // the structure this launch walks
    const rtgo_ctx::FastTree ft = use_alt ? c->alt : [&] {
        rtgo_ctx::FastTree m;
        m.d_fnodes = c->d_fnodes; m.d_fprims = c->d_fprims; m.fast_depth = c->fast_depth; m.n_small = c->n_small; m.n_fnodes = c->n_fnodes;
        m.cuboid_groups = c->cuboid_groups; m.tree_spheres = c->tree_spheres; m.list_cub = c->list_cub; m.n_big_pairs = c->n_big_pairs;
        m.cub_a = c->cub_a; m.cub_b = c->cub_b;
        return m;
    }();
    {
        float reach = c->guard_reach;
        // cuboid_range's margin, in the object-space y units of a face g: the certificate's tolerance plus the rounding of what is
        // compared -- the reference's (u, v) on a face f, carried into y_g units by L_fg, and y_g(t_f) itself.  Each is a handful of
        // float operations on terms no larger than |row| (|o| + t |d|) + |w| <= |row|_1 * 3 reach + |w| (origins within `reach`, hit
        // points within the scene: t |d| <= 2 reach), i.e. <= 12 * 2^-24 of them; K = 64 * 2^-24 leaves five times that, and the
        // build's A, B = max over (f, g) of L_fg |row_f|_1 + |row_g|_1 and of L_fg |w_f| + |w_g|.
        p.cub_mu = kCuboidTol + 64.0f * 5.9604645e-8f * (ft.cub_a * 3.0f * reach + ft.cub_b);
        p.list_cub = ft.list_cub;
        p.tree_spheres = ft.tree_spheres;
        if (!(p.cub_mu < 0.02f)) {   // (tiny faces far from the origin: the margin would let two faces through too often to pay)
            p.list_cub = 0;
            p.cub_mu = -1.0f;        // tree leaves: cuboid_range is not taken either (see render_kernel)
        }
    }
    const uint32_t lr0 = owned_rows_below(wy0, p.band_h, p.n_ranks, p.rank), lr1 = owned_rows_below(wy1, p.band_h, p.n_ranks, p.rank);
    const uint64_t units_hot = (uint64_t)((wx1 - wx0 + unit_px - 1) / unit_px) * (lr1 - lr0);
    {
        const uint64_t waves_guess = (uint64_t)c->num_cus * 16u;
        uint32_t grab = (uint32_t)(units_hot / (waves_guess * 32u));
        const uint32_t grab_cap = env_uint("RTGO_GRAB_MAX", (uint32_t)kUnitsPerGrab);
        const uint32_t grab_max = (64u / unit_px) < grab_cap ? (64u / unit_px) : grab_cap;
        const uint32_t grab_min = env_uint("RTGO_GRAB_MIN", 1u);   // (experiment knob)
        grab = grab < grab_min ? grab_min : grab;
        grab = grab < 1u ? 1u : (grab > grab_max ? grab_max : grab);
        p.grab = grab;
    }
    const uint32_t strip_px = unit_px * p.grab;
    p.hot_x0 = wx0 / strip_px;
    p.hot_w = (wx1 + strip_px - 1) / strip_px - p.hot_x0;
    p.hot_y0 = lr0;
    p.hot_h = lr1 - lr0;
    if (p.hot_w == 0 || p.hot_h == 0) p.hot_x0 = p.hot_y0 = p.hot_w = p.hot_h = 0;
    if ((uint64_t)p.hot_w * p.hot_h > 0x7FFFFF00ull) return fail(c, RTGO_E_UNSUPPORTED, "rtgo_launch: window too large");
    p.n_hot = p.hot_w * p.hot_h;
    p.cold_x0 = p.hot_x0 * strip_px;
    p.cold_x1 = (p.hot_x0 + p.hot_w) * strip_px < p.w ? (p.hot_x0 + p.hot_w) * strip_px : p.w;
    p.rows_above = p.local_rows - p.hot_y0 - p.hot_h;
    p.segs_full = (p.w + 63u) / 64u;
    p.segs_l = p.n_hot ? (p.cold_x0 + 63u) / 64u : 0u;
    p.segs_r = p.n_hot ? (p.w - p.cold_x1 + 63u) / 64u : 0u;
    const uint64_t cold_segs = (uint64_t)(p.hot_y0 + p.rows_above) * p.segs_full + (uint64_t)p.hot_h * (p.segs_l + p.segs_r);
    if (cold_segs > 0x7FFFFF00ull) return fail(c, RTGO_E_UNSUPPORTED, "rtgo_launch: window too large");
    p.n_cold_segs = (uint32_t)cold_segs;
    // cold chunks: about two per resident wave
    p.cold_cs = (uint32_t)((cold_segs + (uint64_t)c->num_cus * 32u - 1) / ((uint64_t)c->num_cus * 32u));
    if (p.cold_cs == 0) p.cold_cs = 1;
    const uint32_t n_cold = (p.n_cold_segs + p.cold_cs - 1) / p.cold_cs;
    p.n_tiles = p.n_hot + n_cold;
    // Inside the rectangle, strip by strip: does any primitive's own screen rectangle (its box as the fast walk culls with it,
    // through the same pinhole projection, padded) reach the strip?  Scenes that do not fill their rectangle -- plateau: a plate and
    // a few objects -- have most of it empty.  The mask depends on the launch geometry only; it is rebuilt when that changes.
    p.hot_mask = nullptr;
    unsigned long long mask_cold_pixels = 0;
    if (cull && p.n_hot > 0 && !std::getenv("RTGO_NO_MASK")) {
        std::vector<uint32_t> key = {p.W, p.H, p.x0, p.y0, p.w, p.h, p.band_h, p.n_ranks, p.rank, p.grab, strip_px, p.hot_x0, p.hot_y0, p.hot_w, p.hot_h};
        const float cam[12] = {p.eye.x, p.eye.y, p.eye.z, p.U.x, p.U.y, p.U.z, p.V.x, p.V.y, p.V.z, p.Wv.x, p.Wv.y, p.Wv.z};
        for (float v : cam) {
            uint32_t bits;
            std::memcpy(&bits, &v, 4);
            key.push_back(bits);
        }
        const size_t words = ((size_t)p.n_hot + 31) / 32;
        if (key != c->mask_key) {
            std::vector<uint32_t> mask(words, 0u);
            bool all_hot = false;
            for (uint32_t i = 0; i < c->n_prims && !all_hot; ++i) {
                uint32_t rx0, rx1, ry0, ry1;
                box_screen_rect(&c->tight[6 * (size_t)i], p, rx0, rx1, ry0, ry1);
                if (rx0 == 0 && rx1 == p.w && ry0 == 0 && ry1 == p.h) {   // a primitive whose rectangle is the whole window (or unknown)
                    all_hot = true;
                    break;
                }
                if (rx1 <= rx0 || ry1 <= ry0) continue;
                const uint32_t sa = rx0 / strip_px, sb = (rx1 - 1) / strip_px;   // strip columns the rectangle touches
                const uint32_t ca = sa > p.hot_x0 ? sa : p.hot_x0, cb = sb < p.hot_x0 + p.hot_w - 1 ? sb : p.hot_x0 + p.hot_w - 1;
                if (ca > cb) continue;
                for (uint32_t wrow = ry0; wrow < ry1; ++wrow) {
                    if (p.n_ranks > 1 && (wrow / p.band_h) % p.n_ranks != p.rank) continue;
                    const uint32_t lrow = owned_rows_below(wrow, p.band_h, p.n_ranks, p.rank);
                    if (lrow < p.hot_y0 || lrow >= p.hot_y0 + p.hot_h) continue;
                    const size_t base = (size_t)(lrow - p.hot_y0) * p.hot_w;
                    for (uint32_t sc = ca; sc <= cb; ++sc) {
                        const size_t bit = base + (sc - p.hot_x0);
                        mask[bit >> 5] |= 1u << (bit & 31u);
                    }
                }
            }
            unsigned long long cold_px = 0;
            if (!all_hot) {
                size_t hot_bits = 0;
                for (size_t b = 0; b < (size_t)p.n_hot; ++b) {
                    if ((mask[b >> 5] >> (b & 31u)) & 1u) {
                        ++hot_bits;
                    } else {
                        const uint32_t sc = p.hot_x0 + (uint32_t)(b % p.hot_w);
                        const uint32_t xa = sc * strip_px, xb = xa + strip_px < p.w ? xa + strip_px : p.w;
                        cold_px += xb > xa ? xb - xa : 0;
                    }
                }
                all_hot = hot_bits == (size_t)p.n_hot;
            }
            if (!all_hot) {
                if (words > c->mask_capacity) {
                    (void)hipFree(c->d_mask);
                    c->d_mask = nullptr;
                    c->mask_capacity = 0;
                    RTGO_HIP(c, hipMalloc(&c->d_mask, words * sizeof(uint32_t)));
                    c->mask_capacity = words;
                }
                // (the previous launch may still be reading the old mask: stream order takes care of it.)  The copy leaves from pinned
                // memory the context keeps, so nothing here waits for the stream; a slot is reused two rebuilds later, by when its copy
                // has long completed (the event wait is a formality)
                const int slot = c->mask_slot;
                c->mask_slot = 1 - slot;
                if (!c->mask_copied[slot]) RTGO_HIP(c, hipEventCreateWithFlags(&c->mask_copied[slot], hipEventDisableTiming));
                else RTGO_HIP(c, hipEventSynchronize(c->mask_copied[slot]));
                if (words > c->h_mask_capacity[slot]) {
                    if (c->h_mask[slot]) (void)hipHostFree(c->h_mask[slot]);
                    c->h_mask[slot] = nullptr;
                    c->h_mask_capacity[slot] = 0;
                    RTGO_HIP(c, hipHostMalloc((void**)&c->h_mask[slot], words * sizeof(uint32_t), hipHostMallocDefault));
                    c->h_mask_capacity[slot] = words;
                }
                std::memcpy(c->h_mask[slot], mask.data(), words * sizeof(uint32_t));
                RTGO_HIP(c, hipMemcpyAsync(c->d_mask, c->h_mask[slot], words * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                RTGO_HIP(c, hipEventRecord(c->mask_copied[slot], c->stream));
            }
            c->mask_all_hot = all_hot;
            c->mask_cold_pixels = all_hot ? 0 : cold_px;
            c->mask_key = key;
        }
        if (!c->mask_all_hot) {
            p.hot_mask = c->d_mask;
            mask_cold_pixels = c->mask_cold_pixels;
        }
    }
    p.nodes = c->d_nodes;
    p.prims = c->d_prims;
    p.fnodes = use_grid ? (const float4*)c->grid.d : ft.d_fnodes;
    p.n_fnodes = use_grid ? c->grid.n_nodes : ft.n_fnodes;
    p.grid = rtgo::GridParams();
    if (use_grid) p.grid = c->grid.gp;
    p.fprims = ft.d_fprims;
    p.frames = c->d_frames;
    p.n_small = ft.n_small;
    p.n_big_pairs = ft.n_big_pairs;
    p.stack_depth = canon ? kStackDepth : ((ft.fast_depth > 0 && !use_grid) ? ft.fast_depth : 1) + 1;   // (+1: fast_tree writes the slot past the top before it knows whether it pushes)
    p.lights = c->d_lights;
    p.accum = c->d_accum;
    p.image = c->d_image;
    p.queue = c->d_queue + (size_t)c->queue_set * kQueues * kQueueStride;
    p.queue_next = c->d_queue + (size_t)(1 - c->queue_set) * kQueues * kQueueStride;
    p.counters = c->d_counters;
#ifdef RTGO_CMPWALK
    if (!c->d_cmp) {
        RTGO_HIP(c, hipMalloc(&c->d_cmp, 256 * 16 * sizeof(float)));
        RTGO_HIP(c, hipMemset(c->d_cmp, 0, 256 * 16 * sizeof(float)));
        RTGO_HIP(c, hipStreamSynchronize(nullptr));   // (null-stream memset: the launch stream does not wait for it)
    }
    p.cmp = c->d_cmp;
#endif
#ifdef RTGO_TIMELINE
    if (!c->d_timeline) RTGO_HIP(c, hipMalloc(&c->d_timeline, 16384 * 128));
    p.timeline = c->d_timeline;
#endif
    p.n_prims = (int)c->n_prims;
    p.n_nodes = 2 * (int)c->n_prims - 1;
    p.n_lights = c->n_lights;
    p.sqrt_spp = f->sqrt_spp;
    p.max_depth = f->max_trace_depth;
    p.frame = f->frame_count;
    p.ambient = f->use_ambient ? 1 : 0;
    p.count_stats = stats ? 1 : 0;
    if (p.n_tiles == 0) return RTGO_OK;  // this rank owns no rows

    // LDS image of the chosen kernel (see render_kernel): canonical = nodes + 6/prim; fast = fnodes + 4/prim + 3/prim.
    // The scene copy is per workgroup and the stack per lane, so bigger scenes want bigger workgroups: pick the size that
    // puts the most waves on a CU (at most 16 = 4 per SIMD, what the kernel's VGPR budget admits), smallest size on ties.
    const int fast_nodes = p.n_fnodes;   // (the tree's nodes, or the grid in their place)
    frames = path && !canon && c->quadrics.empty() && !std::getenv("RTGO_NO_FRAMES");   // scenes of flat primitives only: N and the sampling tangent from LDS
    const size_t scene_lds = (size_t)(2 * (canon ? p.n_nodes : fast_nodes) + (canon ? 6 : 7) * p.n_prims + (frames ? 2 * p.n_prims : 0) /* shading frames */) * sizeof(float4) +
                             (size_t)kMaxLights * sizeof(LightRec) + 16 * sizeof(float) +   // + the raygen constants
                             (size_t)(nn < (uint32_t)kSampleTab ? nn : (uint32_t)kSampleTab) * sizeof(uint4);   // + the per-sample start table
    int block = 0, blocks_per_cu = 0, best_waves = 0, wpe = 4;
    size_t lds = 0;
    // Waves per SIMD the kernel variant is compiled for: 4 (<= 128 VGPRs) or 5 (<= 96, level records in LDS; 1-2 spilled dwords).
    // More resident waves fill more of the vector issue slots (cornell 1080p: 1.32 ms at 4, 1.22 at 5), but every wave then runs
    // slower and the launch ends one unit-duration after the queue runs dry: with few units per wave the shorter tail of fewer
    // waves wins (a 1/16 share: 0.127 / 0.137 ms at 4 / 5).  A 6-waves variant (<= 80 VGPRs) ran 1 % faster still (1.209 ms) but
    // spilled 25-32 registers to scratch -- 472 MB of HBM traffic per launch against 94 MB at 5 waves and 74.6 MB of framebuffer
    // (profiles/r02d) -- and was dropped: the kernel should not pay HBM for registers.
    const uint64_t units_per_wave4 = units_hot * (passes_of(nn)) / ((uint64_t)c->num_cus * 16u);
    const int max_wpe_work = units_per_wave4 >= 3 ? 5 : 4;
    int max_wpe = canon ? 4 : (int)env_uint("RTGO_MAX_WPE", (unsigned int)max_wpe_work);   // (experiment knob, clamped to what exists)
    max_wpe = max_wpe < 4 ? 4 : (max_wpe > 5 ? 5 : max_wpe);
    int min_block = (int)env_uint("RTGO_MIN_BLOCK", 256);   // (experiment knob: 256 / 512 / 1024)
    min_block = min_block >= 1024 ? 1024 : (min_block >= 512 ? 512 : 256);
    for (int w = 4; w <= max_wpe; ++w)
        for (int b = min_block; b <= kMaxBlock; b *= 2) {
            const size_t l = scene_lds + (stream ? (size_t)(b / 64) * 192 * kStreamWindow * sizeof(float) : 0) + (size_t)p.stack_depth * b * (canon ? sizeof(float2) : sizeof(unsigned int)) + (w >= 5 ? (size_t)b * (path ? 3 : 4) * kMaxLevels * sizeof(float) : 0);
            int per_cu = (int)((160 * 1024) / l);
            if (per_cu * (b / 64) > 4 * w) per_cu = (4 * w) / (b / 64);
            const int waves = per_cu * (b / 64);
            if (waves > best_waves) {
                best_waves = waves;
                block = b;
                blocks_per_cu = per_cu;
                lds = l;
                wpe = w;
            }
        }
    if (best_waves == 0) return fail(c, RTGO_E_UNSUPPORTED, "rtgo_launch: scene does not fit in LDS");
    int cus = c->num_cus - (int)(f->reserve_cus < (uint32_t)c->num_cus / 2 ? f->reserve_cus : (uint32_t)c->num_cus / 2);
    unsigned int grid = (unsigned int)(cus * blocks_per_cu);
    const unsigned int need = (p.n_tiles + (block / 64) - 1) / (block / 64);
    if (grid > need) grid = need;

    if (std::getenv("RTGO_DEBUG"))
        std::fprintf(stderr, "rtgo_launch: %s walk%s, grid %u x %d threads, %zu B LDS, %d waves/SIMD variant, %d workgroups/CU, %u strips of %u px (%u x %u at %u,%u), %u cold segments in chunks of %u, stack %d, cuboid margin %g, guard reach %g quadric %g\n",
                     canon ? "canonical" : "fast", (canon && !stats) ? " (beyond the far-field guard)" : "", grid, block, lds, wpe, blocks_per_cu, p.n_hot, strip_px, p.hot_w, p.hot_h, p.hot_x0, p.hot_y0, p.n_cold_segs, p.cold_cs, p.stack_depth, p.cub_mu, c->guard_reach, c->guard_quadric);
#ifdef RTGO_TIMELINE
    c->timeline_waves = grid * (unsigned int)(block / 64);
    if (c->timeline_waves > 16384) return fail(c, RTGO_E_UNSUPPORTED, "timeline buffer too small");
#endif
    RTGO_HIP(c, hipSetDevice(c->device));
    if (c->ev_pending == rtgo_ctx::kEvRing) {
        int rc = harvest_events(c, 1);
        if (rc) return rc;
    }
    const int slot = c->ev_head;
    c->ev_tag[slot] = trial_tag;
    RTGO_HIP(c, hipEventRecord(c->ev_start[slot], c->stream));
    const float4* fp = (const float4*)ft.d_fprims;
    const RenderKernel kernel = find_kernel(path, canon, wpe, stream, stats, frames, use_grid);
    if (!kernel) return fail(c, RTGO_E_UNSUPPORTED, "rtgo_launch: no kernel variant for this configuration");
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, c->stream, p, fp);
    RTGO_HIP(c, hipGetLastError());
    RTGO_HIP(c, hipEventRecord(c->ev_stop[slot], c->stream));
    c->queue_set = 1 - c->queue_set;
    c->rays_culled += ((unsigned long long)p.local_rows * p.w - (unsigned long long)p.hot_h * (p.cold_x1 - p.cold_x0) + mask_cold_pixels) * nn;
    c->ev_head = (c->ev_head + 1) % rtgo_ctx::kEvRing;
    c->ev_pending++;
    c->launches++;
    if (canon) c->launches_canonical++;
    if (trial_tag) c->launches_trial++;
    c->last_variant = (stream ? 1u : 0u) | (use_alt ? 2u : 0u) | (canon ? 4u : 0u) | (trial_tag ? 8u : 0u) | (use_grid ? 16u : 0u);
    return RTGO_OK;
}

int rtgo_assemble_bands(rtgo_ctx* c, void* hip_stream, const void* d_gathered, void* d_full, uint32_t w, uint32_t h, uint32_t band_h,
                        uint32_t n_ranks, uint32_t rows_pad, uint32_t elem_bytes)
{
    if (!c || !d_gathered || !d_full) return fail(c, RTGO_E_INVALID, "rtgo_assemble_bands: NULL argument");
    if (w == 0 || h == 0 || n_ranks == 0 || (elem_bytes != 4 && elem_bytes != 16))
        return fail(c, RTGO_E_INVALID, "rtgo_assemble_bands: empty window, no ranks, or element size not 4 / 16");
    if (band_h == 0) band_h = 4;
    for (uint32_t g = 0; g < n_ranks; ++g)
        if (rtgo_local_rows(h, band_h, n_ranks, g) > rows_pad) return fail(c, RTGO_E_INVALID, "rtgo_assemble_bands: rows_pad smaller than a rank's share");
    RTGO_HIP(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const uint64_t row_bytes = (uint64_t)w * elem_bytes;
    const bool wide = (row_bytes % 16 == 0) && (((uintptr_t)d_gathered | (uintptr_t)d_full) % 16 == 0);
    const uint64_t units = (wide ? row_bytes / 16 : row_bytes / 4) * h;
    uint64_t blocks = (units + 255) / 256;
    const uint64_t cap = (uint64_t)c->num_cus * 16;
    if (blocks > cap) blocks = cap;
    if (wide)
        hipLaunchKernelGGL(assemble_bands_kernel<uint4>, dim3((unsigned int)blocks), dim3(256), 0, st, (const uint4*)d_gathered, (uint4*)d_full,
                           (unsigned int)(row_bytes / 16), h, band_h, n_ranks, rows_pad);
    else
        hipLaunchKernelGGL(assemble_bands_kernel<unsigned int>, dim3((unsigned int)blocks), dim3(256), 0, st, (const unsigned int*)d_gathered,
                           (unsigned int*)d_full, (unsigned int)(row_bytes / 4), h, band_h, n_ranks, rows_pad);
    RTGO_HIP(c, hipGetLastError());
    return RTGO_OK;
}

int rtgo_whitted_set_mesh(rtgo_ctx* c, const float* positions, const float* normals, uint32_t n_vertices, const uint32_t* indices,
                          const uint32_t* material_of_triangle, uint32_t n_triangles, const rtgo_pbr* materials, uint32_t n_materials)
{
    if (!c || !positions || !indices || !materials) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_mesh: NULL argument");
    if (n_triangles == 0 || n_triangles > RTGO_MAX_TRIANGLES || n_vertices == 0 || n_materials == 0)
        return fail(c, RTGO_E_UNSUPPORTED, "rtgo_whitted_set_mesh: triangle count must be in [1, " + std::to_string(RTGO_MAX_TRIANGLES) + "], vertices and materials non-empty");
    for (uint32_t i = 0; i < 3 * n_triangles; ++i)
        if (indices[i] >= n_vertices) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_mesh: index beyond the vertex array");
    for (uint32_t i = 0; i < 3 * n_vertices; ++i)
        if (!std::isfinite(positions[i]) || (normals && !std::isfinite(normals[i]))) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_mesh: non-finite vertex data");
    if (material_of_triangle)
        for (uint32_t i = 0; i < n_triangles; ++i)
            if (material_of_triangle[i] >= n_materials) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_mesh: material index beyond the material array");
    RTGO_HIP(c, hipSetDevice(c->device));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->w_positions);
    (void)hipFree(c->w_normals);
    (void)hipFree(c->w_indices);
    (void)hipFree(c->w_tri_material);
    (void)hipFree(c->w_materials);
    (void)hipFree(c->w_texcoords);
    (void)hipFree(c->w_mat_tex);
    for (void* t : c->w_texels) (void)hipFree(t);
    c->w_texels.clear();
    c->w_mat_tex_host.clear();
    c->w_texcoords = nullptr;
    c->w_mat_tex = nullptr;
    (void)hipFree(c->w_nodes);
    (void)hipFree(c->w_recs);
    (void)hipFree(c->w_tris);
    (void)hipFree(c->w_qrecs);
    (void)hipFree(c->w_tidx);
    (void)hipFree(c->w_scratch);
    c->w_positions = c->w_normals = nullptr;
    c->w_indices = c->w_tri_material = nullptr;
    c->w_materials = nullptr;
    c->w_nodes = nullptr;
    c->w_recs = c->w_tris = nullptr;
    c->w_qrecs = nullptr;
    c->w_tidx = nullptr;
    c->w_scratch = nullptr;
    c->w_triangles = 0;
    const size_t vb = (size_t)n_vertices * 3 * sizeof(float), ib = (size_t)n_triangles * 3 * sizeof(unsigned int);
    RTGO_HIP(c, hipMalloc(&c->w_positions, vb));
    RTGO_HIP(c, hipMemcpyAsync(c->w_positions, positions, vb, hipMemcpyHostToDevice, c->stream));
    if (normals) {
        RTGO_HIP(c, hipMalloc(&c->w_normals, vb));
        RTGO_HIP(c, hipMemcpyAsync(c->w_normals, normals, vb, hipMemcpyHostToDevice, c->stream));
    }
    RTGO_HIP(c, hipMalloc(&c->w_indices, ib));
    RTGO_HIP(c, hipMemcpyAsync(c->w_indices, indices, ib, hipMemcpyHostToDevice, c->stream));
    if (material_of_triangle) {
        RTGO_HIP(c, hipMalloc(&c->w_tri_material, (size_t)n_triangles * sizeof(unsigned int)));
        RTGO_HIP(c, hipMemcpyAsync(c->w_tri_material, material_of_triangle, (size_t)n_triangles * sizeof(unsigned int), hipMemcpyHostToDevice, c->stream));
    }
    RTGO_HIP(c, hipMalloc(&c->w_materials, (size_t)n_materials * sizeof(whitted::Pbr)));
    RTGO_HIP(c, hipMemcpyAsync(c->w_materials, materials, (size_t)n_materials * sizeof(whitted::Pbr), hipMemcpyHostToDevice, c->stream));
    RTGO_HIP(c, hipMalloc(&c->w_nodes, (size_t)(2 * n_triangles - 1) * 2 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->w_recs, (size_t)n_triangles * 4 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->w_tris, (size_t)n_triangles * 3 * sizeof(float4)));
    RTGO_HIP(c, hipMalloc(&c->w_qrecs, (size_t)n_triangles * 2 * sizeof(uint4)));
    RTGO_HIP(c, hipMalloc(&c->w_tidx, (size_t)n_triangles * sizeof(uint2)));
    RTGO_HIP(c, hipMalloc(&c->w_scratch, (size_t)(6 * n_triangles + 16 + 32 * n_triangles) * sizeof(int)));   // parent [2n-1], visit, first, count, record [n each], meta, sah_kernel's 32 n
    int* parent = c->w_scratch;
    int* visit = parent + (2 * n_triangles - 1);
    int* first_of = visit + n_triangles;
    int* count_of = first_of + n_triangles;
    int* rec_of = count_of + n_triangles;
    int* meta = rec_of + n_triangles;
    const size_t keys_lds = (size_t)whitted::kMaxTriangles * sizeof(unsigned long long);
    hipLaunchKernelGGL(whitted::build_kernel, dim3(1), dim3(whitted::kBuildThreads), keys_lds, c->stream, c->w_positions, c->w_indices, (int)n_triangles, c->w_nodes,
                       parent, visit, first_of, count_of, rec_of, c->w_recs, c->w_tris, c->w_qrecs, c->w_tidx, meta);
    RTGO_HIP(c, hipGetLastError());
    // the records over the same leaves, rebuilt top-down with the surface-area heuristic (leaf boxes, links, order arrays in LDS: 33 B per
    // triangle, so meshes beyond ~4650 triangles keep the Morton records)
    const size_t sah_lds = (size_t)n_triangles * (6 * sizeof(float) + sizeof(int) + 2 * sizeof(short) + 1) + 16;
    const bool sah = !std::getenv("RTGO_WHITTED_NO_SAH") && sah_lds <= 150 * 1024;
    if (sah) {
        hipLaunchKernelGGL(whitted::sah_kernel, dim3(1), dim3(whitted::kBuildThreads), sah_lds, c->stream, (int)n_triangles, (const float4*)c->w_nodes, (const int*)parent,
                           (const int*)first_of, (const int*)count_of, meta + 16, c->w_recs, c->w_qrecs, meta);
        RTGO_HIP(c, hipGetLastError());
    }
    int m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    RTGO_HIP(c, hipMemcpyAsync(m, meta, sizeof m, hipMemcpyDeviceToHost, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    if (m[2] > whitted::kMaxWalkDepth && sah) {
        // the surface-area tree came out deeper than the walk's stack (it has no depth bound of its own): back to the Morton records,
        // whose depth is bounded by the code length
        hipLaunchKernelGGL(whitted::build_kernel, dim3(1), dim3(whitted::kBuildThreads), keys_lds, c->stream, c->w_positions, c->w_indices, (int)n_triangles, c->w_nodes,
                           parent, visit, first_of, count_of, rec_of, c->w_recs, c->w_tris, c->w_qrecs, c->w_tidx, meta);
        RTGO_HIP(c, hipGetLastError());
        RTGO_HIP(c, hipMemcpyAsync(m, meta, sizeof m, hipMemcpyDeviceToHost, c->stream));
        RTGO_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (m[0] > 2 * whitted::kStack)
        return fail(c, RTGO_E_UNSUPPORTED, "rtgo_whitted_set_mesh: triangle LBVH depth " + std::to_string(m[0]) + " exceeds what the build handles (" +
                                               std::to_string(2 * whitted::kStack) + ")");
    if (m[2] > whitted::kMaxWalkDepth)
        return fail(c, RTGO_E_UNSUPPORTED, "rtgo_whitted_set_mesh: the walk needs " + std::to_string(m[2]) + " stack entries (limit " +
                                               std::to_string(whitted::kMaxWalkDepth) + ")");
    c->w_n_recs = m[1];
    c->w_n_vertices = (int)n_vertices;
    std::memcpy(&c->w_grid_lo, &m[3], 3 * sizeof(float));
    std::memcpy(&c->w_grid_step, &m[6], 3 * sizeof(float));
    c->w_walk_depth = m[2] < 1 ? 1 : m[2];
    if (!c->w_tile_counters) {
        const size_t heads_bytes = 2 * (size_t)whitted::kTileHeads * whitted::kTileHeadStride * sizeof(unsigned int);
        RTGO_HIP(c, hipMalloc(&c->w_tile_counters, heads_bytes));
        RTGO_HIP(c, hipMemsetAsync(c->w_tile_counters, 0, heads_bytes, c->stream));
        RTGO_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->w_triangles = (int)n_triangles;
    c->w_n_materials = (int)n_materials;
    return RTGO_OK;
}

int rtgo_whitted_set_texcoords(rtgo_ctx* c, const float* uv, uint32_t n_vertices)
{
    if (!c) return RTGO_E_INVALID;
    if (c->w_triangles == 0) return fail(c, RTGO_E_STATE, "rtgo_whitted_set_texcoords: no mesh (call rtgo_whitted_set_mesh first)");
    if (uv && n_vertices != (uint32_t)c->w_n_vertices) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_texcoords: one (u, v) per vertex of the mesh");
    RTGO_HIP(c, hipSetDevice(c->device));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->w_texcoords);
    c->w_texcoords = nullptr;
    if (uv) {
        for (size_t k = 0; k < (size_t)n_vertices * 2; ++k)
            if (!std::isfinite(uv[k])) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_texcoords: non-finite coordinate");
        RTGO_HIP(c, hipMalloc(&c->w_texcoords, (size_t)n_vertices * 2 * sizeof(float)));
        RTGO_HIP(c, hipMemcpyAsync(c->w_texcoords, uv, (size_t)n_vertices * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        RTGO_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RTGO_OK;
}

int rtgo_whitted_set_material_textures(rtgo_ctx* c, uint32_t material, const rtgo_texture* base_color, const rtgo_texture* metallic_roughness,
                                       const rtgo_texture* normal)
{
    if (!c) return RTGO_E_INVALID;
    if (c->w_triangles == 0) return fail(c, RTGO_E_STATE, "rtgo_whitted_set_material_textures: no mesh (call rtgo_whitted_set_mesh first)");
    if (material >= (uint32_t)c->w_n_materials) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_material_textures: material index beyond the table");
    const rtgo_texture* in[3] = {base_color, metallic_roughness, normal};
    for (const rtgo_texture* t : in)
        if (t && (!t->rgba8 || t->width == 0 || t->height == 0 || t->width > 16384 || t->height > 16384))
            return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_material_textures: a texture needs texels and a size in [1, 16384]^2");
    RTGO_HIP(c, hipSetDevice(c->device));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    if (c->w_mat_tex_host.empty()) c->w_mat_tex_host.assign((size_t)c->w_n_materials, whitted::MatTex{{nullptr, 0, 0}, {nullptr, 0, 0}, {nullptr, 0, 0}});
    whitted::Tex out[3];
    for (int k = 0; k < 3; ++k) {
        out[k] = whitted::Tex{nullptr, 0, 0};
        if (!in[k]) continue;
        const size_t bytes = (size_t)in[k]->width * in[k]->height * 4;
        void* d = nullptr;
        RTGO_HIP(c, hipMalloc(&d, bytes));
        c->w_texels.push_back(d);
        RTGO_HIP(c, hipMemcpyAsync(d, in[k]->rgba8, bytes, hipMemcpyHostToDevice, c->stream));
        out[k] = whitted::Tex{(const uchar4*)d, in[k]->width, in[k]->height};
    }
    c->w_mat_tex_host[material] = whitted::MatTex{out[0], out[1], out[2]};   // (texels of a replaced entry stay allocated until the next set_mesh)
    if (!c->w_mat_tex) RTGO_HIP(c, hipMalloc(&c->w_mat_tex, (size_t)c->w_n_materials * sizeof(whitted::MatTex)));
    RTGO_HIP(c, hipMemcpyAsync(c->w_mat_tex, c->w_mat_tex_host.data(), (size_t)c->w_n_materials * sizeof(whitted::MatTex), hipMemcpyHostToDevice, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    return RTGO_OK;
}

int rtgo_whitted_set_lights(rtgo_ctx* c, const rtgo_point_light* lights, uint32_t n)
{
    if (!c || (n > 0 && !lights)) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_lights: bad argument");
    if (n > RTGO_MAX_LIGHTS) return fail(c, RTGO_E_UNSUPPORTED, "rtgo_whitted_set_lights: at most " + std::to_string(RTGO_MAX_LIGHTS) + " lights");
    RTGO_HIP(c, hipSetDevice(c->device));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    if (!c->w_lights) RTGO_HIP(c, hipMalloc(&c->w_lights, RTGO_MAX_LIGHTS * sizeof(whitted::PointLight)));
    if (n > 0) RTGO_HIP(c, hipMemcpyAsync(c->w_lights, lights, n * sizeof(whitted::PointLight), hipMemcpyHostToDevice, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    c->w_n_lights = (int)n;
    return RTGO_OK;
}

int rtgo_whitted_set_miss_color(rtgo_ctx* c, const float rgb[3])
{
    if (!c || !rgb) return fail(c, RTGO_E_INVALID, "rtgo_whitted_set_miss_color: NULL argument");
    c->w_miss = v3{rgb[0], rgb[1], rgb[2]};
    return RTGO_OK;
}

int rtgo_whitted_launch(rtgo_ctx* c, uint32_t width, uint32_t height, uint32_t subframe_index)
{
    if (!c) return RTGO_E_INVALID;
    if (c->w_triangles == 0) return fail(c, RTGO_E_STATE, "rtgo_whitted_launch: no mesh (call rtgo_whitted_set_mesh)");
    if (!c->have_camera) return fail(c, RTGO_E_STATE, "rtgo_whitted_launch: no camera (call rtgo_set_camera)");
    if (!c->d_accum || !c->d_image) return fail(c, RTGO_E_STATE, "rtgo_whitted_launch: no output (call rtgo_resize or rtgo_bind_output)");
    if (width == 0 || height == 0 || (uint64_t)width * height > c->pixels) return fail(c, RTGO_E_INVALID, "rtgo_whitted_launch: image empty or larger than the output buffers");
    RTGO_HIP(c, hipSetDevice(c->device));
    if (!c->w_lights) RTGO_HIP(c, hipMalloc(&c->w_lights, RTGO_MAX_LIGHTS * sizeof(whitted::PointLight)));
    whitted::Params p;
    std::memset(&p, 0, sizeof p);
    p.recs = c->w_recs;
    p.tris = c->w_tris;
    p.n_recs = c->w_n_recs;
    p.qrecs = c->w_qrecs;
    p.tidx = c->w_tidx;
    p.n_vertices = c->w_n_vertices;
    p.grid_lo = c->w_grid_lo;
    p.grid_step = c->w_grid_step;
    p.stack_depth = c->w_walk_depth;
    p.tile_counter = c->w_tile_counters + (size_t)c->w_launch_parity * whitted::kTileHeads * whitted::kTileHeadStride;
    p.tile_counter_next = c->w_tile_counters + (size_t)(1 - c->w_launch_parity) * whitted::kTileHeads * whitted::kTileHeadStride;
    p.tiles_x = (width + 7) / 8;
    p.tiles_y = (height + 7) / 8;
    {
        const uint64_t nt = (uint64_t)p.tiles_x * p.tiles_y;
        auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
        uint64_t stride = (uint64_t)((double)nt * 0.6180339887498949);
        if (stride < 1) stride = 1;
        while (gcd(stride, nt) != 1) ++stride;   // (terminates: nt - 1 and 1 are coprime to nt)
        p.tile_stride = (unsigned int)(stride % (nt > 1 ? nt : 2));
        if (p.tile_stride == 0) p.tile_stride = 1;
    }
    p.positions = c->w_positions;
    p.normals = c->w_normals;
    p.indices = c->w_indices;
    p.tri_material = c->w_tri_material;
    p.texcoords = c->w_texcoords;
    p.mat_tex = c->w_mat_tex;
    p.materials = c->w_materials;
    p.lights = c->w_lights;
    p.n_triangles = c->w_triangles;
    p.n_lights = c->w_n_lights;
    p.accum = c->d_accum;
    p.image = c->d_image;
    p.width = width;
    p.height = height;
    p.subframe = subframe_index;
    p.eye = c->eye;
    p.U = c->U;
    p.V = c->V;
    p.W = c->W;
    p.miss = c->w_miss;
    p.counters = c->d_counters;
    if (c->ev_pending == rtgo_ctx::kEvRing) {
        int rc = harvest_events(c, 1);
        if (rc) return rc;
    }
    const int slot = c->ev_head;
    RTGO_HIP(c, hipEventRecord(c->ev_start[slot], c->stream));
    // one persistent workgroup per CU.  Its LDS holds, beside the lanes' stacks, as much of the structure as fits: everything in
    // its compact form (quantised records, vertices, 16-bit vertex indices), or the fp32 records alone, or nothing
    const size_t stack_bytes = (size_t)whitted::kRenderBlock * (size_t)p.stack_depth * sizeof(unsigned short);
    const size_t rec_bytes = (size_t)p.n_recs * 4 * sizeof(float4);
    const size_t compact_bytes = (size_t)p.n_recs * 2 * sizeof(uint4) + (size_t)p.n_vertices * sizeof(float4) + (size_t)p.n_triangles * sizeof(uint2);
    const size_t lds_cap = 160 * 1024;
    const int mode_cap = (int)env_uint("RTGO_WHITTED_MODE", (unsigned int)whitted::kAllInLds);   // (test and experiment knob: 0 / 1 / 2 = at most that much in LDS)
    int mode = whitted::kAllInL2;
    if (mode_cap >= whitted::kAllInLds && p.n_vertices <= 65535 && compact_bytes + stack_bytes <= lds_cap) mode = whitted::kAllInLds;
    else if (mode_cap >= whitted::kRecordsInLds && rec_bytes + stack_bytes <= lds_cap) mode = whitted::kRecordsInLds;
    const size_t lds = stack_bytes + (mode == whitted::kAllInLds ? compact_bytes : (mode == whitted::kRecordsInLds ? rec_bytes : 0));
    const unsigned int n_tiles = p.tiles_x * p.tiles_y;
    unsigned int blocks = (n_tiles + (whitted::kRenderBlock / 64) - 1) / (whitted::kRenderBlock / 64);
    if (blocks > (unsigned int)c->num_cus) blocks = (unsigned int)c->num_cus;
    if (mode == whitted::kAllInLds) hipLaunchKernelGGL(whitted::render_kernel<whitted::kAllInLds>, dim3(blocks), dim3(whitted::kRenderBlock), lds, c->stream, p);
    else if (mode == whitted::kRecordsInLds) hipLaunchKernelGGL(whitted::render_kernel<whitted::kRecordsInLds>, dim3(blocks), dim3(whitted::kRenderBlock), lds, c->stream, p);
    else hipLaunchKernelGGL(whitted::render_kernel<whitted::kAllInL2>, dim3(blocks), dim3(whitted::kRenderBlock), lds, c->stream, p);
    RTGO_HIP(c, hipGetLastError());
    c->w_launch_parity = 1 - c->w_launch_parity;   // (only once the launch that zeroes the other head is in the stream)
    RTGO_HIP(c, hipEventRecord(c->ev_stop[slot], c->stream));
    c->ev_head = (c->ev_head + 1) % rtgo_ctx::kEvRing;
    c->ev_pending++;
    c->launches++;
    return RTGO_OK;
}

int rtgo_sync(rtgo_ctx* c)
{
    if (!c) return RTGO_E_INVALID;
    RTGO_HIP(c, hipSetDevice(c->device));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    return harvest_events(c, c->ev_pending);
}

static int copy_out(rtgo_ctx* c, void* host, const void* dev, size_t bytes, size_t elem)
{
    if (!c || !host) return fail(c, RTGO_E_INVALID, "rtgo_read: NULL argument");
    if (!dev) return fail(c, RTGO_E_STATE, "rtgo_read: no output buffer");
    if (bytes > c->pixels * elem) return fail(c, RTGO_E_INVALID, "rtgo_read: more bytes than the output holds");
    int rc = rtgo_sync(c);
    if (rc) return rc;
    RTGO_HIP(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    return RTGO_OK;
}

int rtgo_read_image(rtgo_ctx* c, void* host, size_t bytes) { return copy_out(c, host, c ? c->d_image : nullptr, bytes, sizeof(uchar4)); }
int rtgo_read_accum(rtgo_ctx* c, void* host, size_t bytes) { return copy_out(c, host, c ? c->d_accum : nullptr, bytes, sizeof(float4)); }

int rtgo_write_accum(rtgo_ctx* c, const void* host, size_t bytes)
{
    if (!c || !host) return fail(c, RTGO_E_INVALID, "rtgo_write_accum: NULL argument");
    if (!c->d_accum) return fail(c, RTGO_E_STATE, "rtgo_write_accum: no output buffer");
    if (bytes > c->pixels * sizeof(float4)) return fail(c, RTGO_E_INVALID, "rtgo_write_accum: too many bytes");
    int rc = rtgo_sync(c);
    if (rc) return rc;
    RTGO_HIP(c, hipMemcpyAsync(c->d_accum, host, bytes, hipMemcpyHostToDevice, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    return RTGO_OK;
}

int rtgo_get_stats(rtgo_ctx* c, rtgo_stats* out)
{
    if (!c || !out) return fail(c, RTGO_E_INVALID, "rtgo_get_stats: NULL argument");
    int rc = rtgo_sync(c);
    if (rc) return rc;
    unsigned long long h[8];
    RTGO_HIP(c, hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    out->rays_total = h[0];
    out->rays_occlusion = h[1];
    out->node_visits = h[2];
    out->prim_tests = h[3];
    out->hits = h[4];
    out->last_launch_ms = c->last_ms;
    out->total_launch_ms = c->total_ms;
    out->launches = c->launches;
    out->lbvh_depth = (uint32_t)c->lbvh_depth;
    out->dbg_fast_boxes = h[5];
    out->dbg_fast_tests = h[6];
    out->rays_culled = c->rays_culled;
    out->launches_canonical = c->launches_canonical;
    out->cuboid_groups = (uint32_t)c->cuboid_groups;
    out->guard_reach = c->guard_reach;
    out->guard_quadric = c->guard_quadric;
    out->last_variant = c->last_variant;
    out->launches_trial = c->launches_trial;
    return RTGO_OK;
}

int rtgo_reset_stats(rtgo_ctx* c)
{
    if (!c) return RTGO_E_INVALID;
    int rc = rtgo_sync(c);
    if (rc) return rc;
    // on the launch stream: a memset on the null stream is not ordered against a non-blocking stream's kernels
    RTGO_HIP(c, hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream));
    RTGO_HIP(c, hipStreamSynchronize(c->stream));
    c->total_ms = 0.0f;
    c->last_ms = 0.0f;
    c->launches = 0;
    c->launches_canonical = 0;
    c->launches_trial = 0;
    c->rays_culled = 0;
    return RTGO_OK;
}

int rtgo_read_bvh(rtgo_ctx* c, void* host_nodes, size_t node_bytes, void* host_inv, size_t inv_bytes, void* host_aabbs, size_t aabb_bytes)
{
    if (!c) return RTGO_E_INVALID;
    if (c->n_prims == 0) return fail(c, RTGO_E_STATE, "rtgo_read_bvh: no scene");
    const size_t n = c->n_prims;
    if ((host_nodes && node_bytes != (2 * n - 1) * 32) || (host_inv && inv_bytes != n * 48) || (host_aabbs && aabb_bytes != n * 24))
        return fail(c, RTGO_E_INVALID, "rtgo_read_bvh: buffer sizes must be (2n-1)*32, n*48, n*24");
    int rc = rtgo_sync(c);
    if (rc) return rc;
    if (host_nodes) RTGO_HIP(c, hipMemcpy(host_nodes, c->d_nodes, node_bytes, hipMemcpyDeviceToHost));
    if (host_inv) {
        std::vector<float4> tmp(6 * n);
        RTGO_HIP(c, hipMemcpy(tmp.data(), c->d_prims, 6 * n * sizeof(float4), hipMemcpyDeviceToHost));
        float* o = (float*)host_inv;
        for (size_t i = 0; i < n; ++i) std::memcpy(o + 12 * i, &tmp[6 * i], 48);
    }
    if (host_aabbs) RTGO_HIP(c, hipMemcpy(host_aabbs, c->d_aabb, aabb_bytes, hipMemcpyDeviceToHost));
    return RTGO_OK;
}

}  // extern "C"
