// rtgo_device.h -- gfx950 device code of the RayTracinGO hot path: the scene-preparation/LBVH-build kernel and the
// render megakernel that replaces the OptiX pipeline of engine/kernel.cu (raygen + traversal + 4 intersection
// programs + 2 closest-hit programs + miss + accumulation), all in one launch.
//
// Written for CDNA4 only (wave64, LDS-resident scene, per-lane LDS traversal stack).  No MFMA: there is no dense
// contraction on this path.  Arithmetic follows the reference's IEEE-float32 statement operation by operation (this
// translation unit is compiled with -ffp-contract=off); the few places where the hoisted form differs from the
// literal one only in the sign of a zero are noted.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtgo {

constexpr int kMaxBlock = 1024;      // workgroup = 256, 512 or 1024 threads: chosen per scene so that 16 waves fit a CU's LDS
constexpr int kStackDepth = 24;      // per-lane traversal stack entries of the canonical walk (LBVH depth is checked against it at build)
constexpr int kDefaultLeafBudget = 6;   // fast walk: LBVH subtrees whose leaf-test cost is <= this many rectangle tests become one leaf (6 = one cube)
constexpr int kMaxPrims = 512;
constexpr int kMaxLights = 10;
constexpr int kSamplesPerPass = 16;   // samples of one pixel that run side by side (the in-order sum costs this many lane exchanges)
constexpr int kQueues = 32;          // work-queue heads (power of two <= 64; 8 / 16 / 32 heads: a 1/8 frame share takes 0.261 / 0.248 / 0.244 ms)
constexpr unsigned int kQueueStride = 16;   // words between two heads: one 64-byte line each
constexpr int kUnitsPerGrab = 4;     // most units (64 paths each) in one strip = one queue entry (longer strips: seeds cheaper, balance worse)
constexpr int kSampleTab = 256;     // samples per pixel whose start (LCG skip, jitter cell) comes from a table in LDS (16 B each); beyond: computed
constexpr int kMaxLevels = 5;        // bounce records kept per path (maxTraceDepth <= 5)
constexpr float kPi = 3.14159265358979323846f;  // M_PIf, sutil/vec_math.h:43

struct v3 {
    float x, y, z;
};

// ---- LDS/global scene layout ----------------------------------------------------------------------------------
// node record, 32 B = 2 x float4:  q0 = (bmin.xyz, bits(left)), q1 = (bmax.xyz, bits(right))
//   internal nodes [0, n-2], leaves [n-1, 2n-2]; leaf: left = primitive (SBT) index, right = -1
// primitive record, 96 B = 6 x float4:
//   q0..q2 = rows 0..2 of M^-1 (row 3 is never needed: TransformRay/TransformNormal drop w, kernel.cu:125-142)
//   q3 = (kd.xyz, specularity)   q4 = (kr.xyz, bits(type))   q5 = (Le.xyz, 0)
struct LightRec {
    float corner[3], v1[3], v2[3], normal[3], color[3], falloff;  // device::SurfaceLight, params.h:73-87
};

// The fast walk's third structure (GRID instantiations): a uniform grid over the small primitives, walked cell by cell (fast_grid).
// It travels in the buffer and the LDS region of the tree it replaces: n_cells words (first item | count << 16), then the items (16-bit
// positions into fprims).
struct GridParams {
    float min_x, min_y, min_z, cs_x, cs_y, cs_z, ics_x, ics_y, ics_z;   // lower corner, cell size, 1 / cell size
    int nx, ny, nz;                 // cells per axis (<= 32 each)
    int n_cells;                    // words of the table: (nx + 2) (ny + 2) (nz + 2), a border of empty cells around the grid
    int rec_off4, items_off4;       // where the cell records and the item lists start, in float4 units from the table's start
    float margin;                   // fast_grid stops once the closest hit lies this far (in t) before the exit of the cell it is in
};

struct LaunchParams {
    const float4* nodes;            // canonical LBVH, 2 float4 per node
    const float4* prims;            // 6 float4 per primitive, SBT order
    const float4* fnodes;           // the fast walk's tree: 2 float4 per node, root 0; leaf: left = first record, right = -(count | pairs << 12)
    int n_fnodes;
    const float4* fprims;           // 4 float4 per primitive in Morton order: rows 0..2 of M^-1, (bits(type), bits(SBT index), 0, 0)
    const float4* frames;           // 2 float4 per primitive, SBT order: the shading frame of a flat primitive (see build_kernel), w of the first = 1 when it has one
    int stack_depth;                // per-lane LDS stack entries this launch needs
    int n_small;                    // fast walk: fprims [0, n_small) are in the tree, [n_small, n_prims) are tested up front
    int n_big_pairs;                // ... of which the first 2*n_big_pairs records are pairs of opposite rectangles (pair_test)
    int list_cub;                   // 1 / 2: the up-front list starts with three pairs certified as one box / one room (cuboid_range), 0: it does not
    float cub_mu;                   // cuboid_range's margin for this launch (object-space units of a face's y axis)
    int tree_spheres;               // 1: every primitive of the fast walk's tree is a sphere (balls): leaves go straight to the sphere test
    GridParams grid;                // GRID instantiations: fnodes holds the grid instead of a tree
    const LightRec* lights;
    float4* accum;
    uchar4* image;
    unsigned int* queue;            // kQueues work-queue heads, kQueueStride words apart, all zero when the launch starts
    unsigned int* queue_next;       // the other set of heads: zeroed by this launch for the next one (no memset between frames)
    unsigned long long* counters;   // [0] rays_total [1] rays_occlusion [2] node_visits [3] prim_tests [4] hits
    int n_prims, n_nodes, n_lights;
    unsigned int W, H;              // full image
    int sqrt_spp, max_depth;
    unsigned int frame;
    int ambient;
    int count_stats;                // canonical walk only: flush the V/T/h counters (0 when the walk is a fallback, not a request)
    unsigned int x0, y0, w, h;      // window
    unsigned int band_h, n_ranks, rank, local_rows;
    unsigned int n_tiles;            // work-queue entries of this launch = n_hot strips, then n_cold chunks
    // The strips that can contain geometry form a rectangle of hot_w x hot_h strips at (strip column hot_x0, local row hot_y0):
    // the screen bounds of the scene, computed by the host.  Queue entries [0, n_hot) are those strips, row-major.
    unsigned int hot_x0, hot_y0, hot_w, hot_h, n_hot;
    // Every pixel outside it is background whatever its samples' jitter (no primary ray can reach the scene's bounds), so the
    // timed kernel writes those without tracing (the instrumented kernel, which defines V/T/h, traces everything: the host
    // then makes the rectangle the whole window).  They are cut into 64-pixel row segments, enumerated region by region
    // (rows below the rectangle, rows above, left of it, right of it); queue entry n_hot + c is cold chunk c = segments
    // [c*cold_cs, (c+1)*cold_cs).  They sit at the END of the queue: cheap filler for the waves that run out of strips first.
    unsigned int cold_x0, cold_x1;   // window columns [cold_x0, cold_x1) belong to the rectangle's strips
    unsigned int rows_above;         // local rows over the rectangle (those below it: hot_y0)
    unsigned int segs_full, segs_l, segs_r;   // 64-pixel segments per full row, per row left / right of the rectangle
    unsigned int n_cold_segs, cold_cs;
    unsigned int grab;               // units per strip (1..kUnitsPerGrab, strip <= 64 pixels)
    // One bit per strip of the rectangle (bit = strip index, row-major): 0 = no primitive's screen rectangle reaches the strip, so
    // its pixels are background like those outside the rectangle and are written without tracing.  Null when every strip is hot
    // (scenes that fill their rectangle: nothing to look up then).
    const unsigned int* hot_mask;
#ifdef RTGO_TIMELINE
    unsigned long long* timeline;    // diagnostic build: 8 words per wave, see tools/timeline.py
#endif
#ifdef RTGO_CMPWALK
    float* cmp;                      // diagnostic build: [0] = number of rays on which the two walks disagree, then 16-float records
#endif
    v3 eye, U, V, Wv, bg;
    v3 bg_pixel;                     // ((0 + bg) + bg + ... N*N times) * (1/(N*N)) in float: the value of a pixel whose samples all miss
};

// ---- float3 helpers, same operation order as sutil/vec_math.h ----------------------------------------------------
__device__ __forceinline__ v3 mk(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 vmul(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 vscale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 vneg(v3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec_math.h:523-526
__device__ __forceinline__ v3 vcross(v3 a, v3 b)                                                   // vec_math.h:529-532
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Correctly rounded a / b and sqrt(x) as the compiler's own expansions compute them, without the range plumbing around them.
// hipcc expands an IEEE f32 division into v_div_scale x 2, v_rcp, six v_fma / v_mul, v_div_fmas, v_div_fixup (12 instructions): the
// scale / fmas / fixup steps only act when an operand is zero, infinite, NaN or subnormal, when |a / b| leaves [2^-126, 2^96) or when
// |a| < 2^-103 (V_DIV_SCALE_F32's rules); otherwise they pass their operands through and the quotient is the value of the eight steps
// below, bit for bit.  Likewise sqrtf: v_sqrt_f32 and two residual tests against the neighbouring floats are the correctly rounded
// root for normal x in [2^-96, inf); the other nine instructions rescale subnormal inputs and pass 0 / inf / NaN through.
// Where these run (ray parameters t = -o.y / d.y of front-facing planes with o.y > 0, sphere roots, lengths of directions and
// normals) a quotient outside those ranges is rejected by the comparisons that follow whatever its value (|t| >= 2^95 or < 2^-126
// against 1e-4 < t < tmax <= 1e16), and operands are float combinations of scene coordinates (|.| <= 500, granularity ~1e-7 of
// them): zero, or nowhere near 2^-103.  The canonical walk keeps the plain operators, so every fast == canonical test -- the suite,
// tools/cmp_walks.py ray by ray, the fuzzers -- holds the lean forms to IEEE on every ray traced.  -DRTGO_IEEE_OPS: plain operators.
__device__ __forceinline__ float div_cr(float a, float b)
{
#ifdef RTGO_IEEE_OPS
    return a / b;
#else
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, y0, 1.0f);
    const float y1 = fmaf(e, y0, y0);
    const float q0 = a * y1;
    const float r0 = fmaf(-b, q0, a);
    const float q1 = fmaf(r0, y1, q0);
    const float r1 = fmaf(-b, q1, a);
    return fmaf(r1, y1, q1);
#endif
}
__device__ __forceinline__ float sqrt_cr(float x)
{
#ifdef RTGO_IEEE_OPS
    return sqrtf(x);
#else
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sdn = __uint_as_float(__float_as_uint(s) - 1u), sup = __uint_as_float(__float_as_uint(s) + 1u);
    const float vdn = fmaf(-sdn, s, x), vup = fmaf(-sup, s, x);
    float r = vdn <= 0.0f ? sdn : s;
    r = vup > 0.0f ? sup : r;
    return r;
#endif
}

// sincosf(x) for 0 <= x < 131072, finite: the device library's own small-argument path (three-constant Cody-Waite reduction by pi/2, its
// two minimax polynomials, the quadrant logic) without what it wraps around it -- the test for the Payne-Hanek path, the sign of x, the
// inf / NaN class test on both results.  GetRayOnHemisphere's angles are acos(..) in [0, pi/2] and 2 pi r in [0, 2 pi).
// tools/lean_ops_probe.hip runs it against sincosf on EVERY float of [0, 8): identical bits.  -DRTGO_IEEE_OPS: the library call.
__device__ __forceinline__ void sincos_cr(float x, float* s_out, float* c_out)
{
#ifdef RTGO_IEEE_OPS
    sincosf(x, s_out, c_out);
#else
    const float n = rintf(x * __uint_as_float(0x3f22f983u));                  // x * 2/pi, to nearest even
    const int ni = (int)n;
    float r = fmaf(n, __uint_as_float(0xbfc90fdau), x);
    r = fmaf(n, __uint_as_float(0xb3a22168u), r);
    r = fmaf(n, __uint_as_float(0xa7c234c4u), r);
    const float r2 = r * r;
    float t = fmaf(__uint_as_float(0xb94c1982u), r2, __uint_as_float(0x3c0881c4u));
    t = fmaf(r2, t, __uint_as_float(0xbe2aaa9du));
    t = r2 * t;
    const float sr = fmaf(r, t, r);
    float u = fmaf(__uint_as_float(0x37d75334u), r2, __uint_as_float(0xbab64f3bu));
    u = fmaf(r2, u, __uint_as_float(0x3d2aabf7u));
    u = fmaf(r2, u, __uint_as_float(0xbf000004u));
    const float cr = fmaf(r2, u, 1.0f);
    const bool even = (ni & 1) == 0;
    const unsigned int flip = ((unsigned int)ni << 30) & 0x80000000u;          // quadrants 2 and 3
    *s_out = __uint_as_float(__float_as_uint(even ? sr : cr) ^ flip);
    *c_out = __uint_as_float(__float_as_uint(even ? cr : -sr) ^ flip);
#endif
}

__device__ __forceinline__ float vlength(v3 v) { return sqrt_cr(vdot(v, v)); }  // vec_math.h:535-538
__device__ __forceinline__ v3 vnormalize(v3 v)                               // vec_math.h:541-545
{
    float invLen = div_cr(1.0f, sqrt_cr(vdot(v, v)));
    return vscale(v, invLen);
}

// ---- RNG: cuda/random.h:30-66 -------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int tea16(unsigned int v0, unsigned int v1)
{
    unsigned int s0 = 0;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
// the LCG advanced by m draws in O(log m): f(s) = a*s + c, f^m(s) = A*s + C by repeated squaring of the affine map
__device__ __forceinline__ unsigned int lcg_skip(unsigned int s, unsigned int m)
{
    unsigned int A = 1u, C = 0u, a = 1664525u, c = 1013904223u;
    while (m) {
        if (m & 1u) {
            A = A * a;
            C = C * a + c;
        }
        c = (a + 1u) * c;
        a = a * a;
        m >>= 1;
    }
    return A * s + C;
}
__device__ __forceinline__ float rnd(unsigned int& prev)
{
    prev = 1664525u * prev + 1013904223u;
    return (float)(prev & 0x00FFFFFFu) / (float)0x01000000;
}

// ---- intersection programs (kernel.cu:250-416) with M^-1 hoisted to scene upload -------------------------------------
// TransformRay (kernel.cu:125-135).  The w = 0 / w = 1 products of the 4-term operator* (Matrix.h:472-493) are
// dropped / folded: x*1 is exact and +(m*0) can only change the sign of a zero.
__device__ __forceinline__ v3 xf_dir(const float4 r0, const float4 r1, const float4 r2, v3 d)
{
    return mk(r0.x * d.x + r0.y * d.y + r0.z * d.z, r1.x * d.x + r1.y * d.y + r1.z * d.z,
              r2.x * d.x + r2.y * d.y + r2.z * d.z);
}
__device__ __forceinline__ v3 xf_point(const float4 r0, const float4 r1, const float4 r2, v3 o)
{
    return mk(r0.x * o.x + r0.y * o.y + r0.z * o.z + r0.w, r1.x * o.x + r1.y * o.y + r1.z * o.z + r1.w,
              r2.x * o.x + r2.y * o.y + r2.z * o.z + r2.w);
}
// TransformNormal (kernel.cu:138-142): transpose(M^-1) * (n, 0)
__device__ __forceinline__ v3 xf_normal(const float4 r0, const float4 r1, const float4 r2, v3 n)
{
    return mk(r0.x * n.x + r1.x * n.y + r2.x * n.z, r0.y * n.x + r1.y * n.y + r2.y * n.z,
              r0.z * n.x + r1.z * n.y + r2.z * n.z);
}

// returns true when the program would call optixReportIntersection(t, n)
__device__ __forceinline__ bool intersect_prim(int type, const float4 r0, const float4 r1, const float4 r2, v3 wo, v3 wd,
                                               float& t_out, v3& n_out)
{
    const v3 d = xf_dir(r0, r1, r2, wd);
    const v3 o = xf_point(r0, r1, r2, wo);
    if (type == 2) {  // __intersection__rectangle, kernel.cu:372-416 (one-sided: only rays going down in object space)
        const float divisor = d.y;
        if (divisor != 0.0f) {
            const float t = (0.0f - o.y) / divisor;
            if (t > 0.0001f) {
                const float px = o.x + t * d.x, pz = o.z + t * d.z;
                const float u = px + 0.5f;      // dot(p - p0, a), a = (1,0,0), p0 = (-1/2, 0, 1/2)
                const float v = -(pz - 0.5f);   // dot(p - p0, b), b = (0,0,-1)
                if (0.0f < u && u < 1.0f && 0.0f < v && v < 1.0f && d.y < 0.0f) {
                    t_out = t;
                    n_out = xf_normal(r0, r1, r2, mk(0.0f, 1.0f, 0.0f));
                    return true;
                }
            }
        }
        return false;
    }
    if (type == 3) {  // __intersection__sphere, kernel.cu:250-287 (near root only)
        const float a = vdot(d, d);
        const float b = 2.0f * vdot(d, o);
        const float c = vdot(o, o) - 1.0f;
        const float discr = b * b - 4.0f * a * c;
        if (discr > 0.0f) {
            const float sdiscr = sqrtf(discr);
            const float t = (-b - sdiscr) / (2.0f * a);
            if (t > 0.0001f) {
                const v3 n = vnormalize(vadd(o, vscale(d, t)));
                t_out = t;
                n_out = xf_normal(r0, r1, r2, n);
                return true;
            }
        }
        return false;
    }
    if (type == 0) {  // __intersection__cylinder + GetTMinCylinder, kernel.cu:290-331, 152-181
        const float a = d.x * d.x + d.z * d.z;
        const float b = 2.0f * (o.x * d.x + o.z * d.z);
        const float c = o.x * o.x + o.z * o.z - 1.0f;
        const float discr = b * b - 4.0f * a * c;
        if (discr > 0.001f) {
            const float sdiscr = sqrtf(discr);
            const float t0 = (-b + sdiscr) / (2.0f * a);
            const float t1 = (-b - sdiscr) / (2.0f * a);
            float t = 1e16f;
            bool valid = false;
            if (t0 > 0.001f) {
                const float py = o.y + t0 * d.y;
                if (py > -1.0f && py < 1.0f) {
                    t = t0;
                    valid = true;
                }
            }
            if (t1 > 0.001f && t1 < t) {
                const float py = o.y + t1 * d.y;
                if (py > -1.0f && py < 1.0f) {
                    t = t1;
                    valid = true;
                }
            }
            if (valid) {
                const float px = o.x + t * d.x, pz = o.z + t * d.z;
                t_out = t;
                n_out = xf_normal(r0, r1, r2, mk(px, 0.0f, pz));
                return true;
            }
        }
        return false;
    }
    {  // __intersection__disk, kernel.cu:334-369 (two-sided, |d.y| >= 0.01)
        const float divisor = d.y;
        if (!(divisor > 0.0f - 0.01f && divisor < 0.0f + 0.01f)) {
            const float t = (-o.y) / divisor;
            if (t > 0.0001f) {
                const v3 p = vadd(o, vscale(d, t));
                if (vdot(p, p) < 1.0f) {
                    t_out = t;
                    n_out = xf_normal(r0, r1, r2, mk(0.0f, 1.0f, 0.0f));
                    return true;
                }
            }
        }
        return false;
    }
}

// slab test of one node box against [tmin, tmax]; identical, operation for operation, to oracle box_test()
__device__ __forceinline__ bool box_test(const float4 q0, const float4 q1, v3 o, v3 id, float tmin, float tmax, float& tn_out)
{
    float t0 = (q0.x - o.x) * id.x, t1 = (q1.x - o.x) * id.x;
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = (q0.y - o.y) * id.y;
    t1 = (q1.y - o.y) * id.y;
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    t0 = (q0.z - o.z) * id.z;
    t1 = (q1.z - o.z) * id.z;
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    tn = fmaxf(tn, tmin);
    tf = fminf(tf, tmax);
    tn_out = tn;
    return tn <= tf;
}

struct Hit {
    float t;
    v3 n;      // world-space normal, not normalised (TransformNormal); not filled in for flat winners of the fast walk: their frame is in LDS
    int prim;
};

// optixTrace's traversal over the LDS-resident canonical LBVH: nearest child first, far child pushed on the per-lane LDS
// stack with its entry distance and culled against the current closest hit when popped.  A hit is accepted iff
// tmin < t < current tmax; ties keep the lower SBT index.
template <bool STATS>
__device__ __forceinline__ bool closest_hit(const float4* __restrict__ s_nodes, const float4* __restrict__ s_prims,
                                            float2* __restrict__ s_stack, int bshift, v3 o, v3 d, float tmin, float tmax, Hit& best,
                                            unsigned int& c_nodes, unsigned int& c_tests)
{
    best.prim = -1;
    best.t = tmax;
    best.n = mk(0.0f, 0.0f, 0.0f);
    const v3 id = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float tn;
    float4 q0 = s_nodes[0], q1 = s_nodes[1];
    if (STATS) c_nodes += 1;
    if (!box_test(q0, q1, o, id, tmin, best.t, tn)) return false;
    int sp = 0;
    int left = __float_as_int(q0.w), right = __float_as_int(q1.w);
    for (;;) {
        bool pop = false;
        if (right < 0) {
            const int i = left;
            const float4 r0 = s_prims[6 * i + 0], r1 = s_prims[6 * i + 1], r2 = s_prims[6 * i + 2];
            const int type = __float_as_int(s_prims[6 * i + 4].w);
            float t;
            v3 n;
            if (STATS) c_tests += 1;
            if (intersect_prim(type, r0, r1, r2, o, d, t, n) && t > tmin &&
                (t < best.t || (t == best.t && best.prim >= 0 && i < best.prim))) {
                best.t = t;
                best.n = n;
                best.prim = i;
            }
            pop = true;
        } else {
            const float4 l0 = s_nodes[2 * left], l1 = s_nodes[2 * left + 1];
            const float4 h0 = s_nodes[2 * right], h1 = s_nodes[2 * right + 1];
            float tl, tr;
            if (STATS) c_nodes += 2;
            const bool hl = box_test(l0, l1, o, id, tmin, best.t, tl);
            const bool hr = box_test(h0, h1, o, id, tmin, best.t, tr);
            if (hl && hr) {
                const bool swap = tr < tl;
                // push the far child
                const int far_idx = swap ? left : right;
                const float far_t = swap ? tl : tr;
                s_stack[sp << bshift] = make_float2(far_t, __int_as_float(far_idx));
                ++sp;
                if (swap) {
                    left = __float_as_int(h0.w);
                    right = __float_as_int(h1.w);
                } else {
                    left = __float_as_int(l0.w);
                    right = __float_as_int(l1.w);
                }
            } else if (hl) {
                left = __float_as_int(l0.w);
                right = __float_as_int(l1.w);
            } else if (hr) {
                left = __float_as_int(h0.w);
                right = __float_as_int(h1.w);
            } else {
                pop = true;
            }
        }
        if (pop) {
            bool found = false;
            while (sp > 0) {
                --sp;
                const float2 e = s_stack[sp << bshift];
                if (e.x <= best.t) {
                    const int idx = __float_as_int(e.y);
                    const float4 n0 = s_nodes[2 * idx], n1 = s_nodes[2 * idx + 1];
                    left = __float_as_int(n0.w);
                    right = __float_as_int(n1.w);
                    found = true;
                    break;
                }
            }
            if (!found) break;
        }
    }
    return best.prim >= 0;
}

// =====================================================================================================================
// Fast walk (the kernel that is timed).  Same closest hit, bit for bit, as the canonical walk above -- every accepted
// candidate goes through the same intersection arithmetic and the same acceptance rule -- but organised for the SIMDs:
//   * LBVH subtrees whose leaf-test cost fits a budget are collapsed into one leaf whose primitives sit contiguously (Morton order), so
//     a wave spends its time in short uniform primitive loops instead of divergent one-primitive leaves;
//   * while-while structure: all lanes first descend to their next leaf, then all lanes with a leaf test it;
//   * the slab test is 6 FMAs on a reciprocal direction (v_rcp_f32): it only steers culling, which stays conservative
//     because every reference AABB is padded by 1e-3 (primitive.cpp:16,62-67), three orders above the rounding at stake;
//   * rejections that need no division come first (rectangle: d.y >= 0 or o.y <= 0 in object space can never pass
//     kernel.cu:394-400), and the normal is transformed once, for the winner only.
// =====================================================================================================================
struct FastHit {
    float t;
    int pos;   // Morton position of the winner | kFlat when it is a rectangle or a disk (object-space normal (0,1,0): kernel.cu:345,388)
    int orig;  // its SBT index (tie-break + material lookup)
};
constexpr int kFlat = 0x10000;
constexpr float kCuboidTol = 1e-4f;   // cuboid certificate: how far (in a face's object-space y) another face's corner may be on the wrong side of its plane

// the acceptance rule of SURVEY a14 (tmin < t < current closest; ties keep the lower SBT index), without branches
__device__ __forceinline__ bool closer(float t, int orig, float tmin, const FastHit& best)
{
    return (t > tmin) & ((t < best.t) | ((t == best.t) & (best.orig >= 0) & (orig < best.orig)));
}

// __intersection__rectangle (kernel.cu:372-416) on M^-1 rows, as straight-line code: every lane evaluates the whole test and the
// result is committed through selects.  The lanes of a wave carry unrelated rays once paths have bounced, so some lane needs every
// stage of the test anyway; nesting the stages in branches then only adds exec-mask bookkeeping and idle lanes.  `facing` (d.y < 0 in
// object space, evaluated by the caller) and the other conditions of the reference enter as predicates.
// second half of the rectangle test for the lanes in `c` (everything up to t > 0.0001 and the closest-hit rule has passed): the hit
// point against the unit square, and the commit through selects
template <typename Ptr>
__device__ __forceinline__ void rect_finish(Ptr rec, float t, bool c, int pos, int orig, v3 wo, v3 wd, FastHit& best)
{
    if (__ballot(c) == 0ull) return;   // (coherent waves -- primary rays -- often leave here together)
    const float4 r0 = rec[0], r2 = rec[2];
    const float dx = r0.x * wd.x + r0.y * wd.y + r0.z * wd.z, dz = r2.x * wd.x + r2.y * wd.y + r2.z * wd.z;
    const float ox = r0.x * wo.x + r0.y * wo.y + r0.z * wo.z + r0.w, oz = r2.x * wo.x + r2.y * wo.y + r2.z * wo.z + r2.w;
    const float px = ox + t * dx, pz = oz + t * dz;
    const float u = px + 0.5f, v = -(pz - 0.5f);
    c = c & (0.0f < u) & (u < 1.0f) & (0.0f < v) & (v < 1.0f);
    best.t = c ? t : best.t;
    best.pos = c ? (pos | kFlat) : best.pos;
    best.orig = c ? orig : best.orig;
}

template <typename Ptr>
__device__ __forceinline__ void rect_commit(Ptr rec, const float4 r1, float dy, bool facing, int pos, int orig, v3 wo, v3 wd, float tmin, FastHit& best)
{
    const float oy = r1.x * wo.x + r1.y * wo.y + r1.z * wo.z + r1.w;
    // oy > 0 with d.y < 0 is the only way to t > 0 (so d.y != 0 and t > 1e-4 can hold); lanes that fail carry garbage in t
    const float t = div_cr(0.0f - oy, dy);
    const bool c = facing & (oy > 0.0f) & (t > 0.0001f) & closer(t, orig, tmin, best);
    rect_finish(rec, t, c, pos, orig, wo, wd, best);
}

template <typename Ptr>
__device__ __forceinline__ void leaf_test(Ptr s_fprims, int pos, v3 wo, v3 wd, float tmin, FastHit& best)
{
    const float4 r1 = s_fprims[4 * pos + 1];
    const float4 meta = s_fprims[4 * pos + 3];
    const int type = __float_as_int(meta.x), orig = __float_as_int(meta.y);
    if (type == 2) {  // rectangle
        const float dy = r1.x * wd.x + r1.y * wd.y + r1.z * wd.z;
        rect_commit(s_fprims + 4 * pos, r1, dy, dy < 0.0f, pos, orig, wo, wd, tmin, best);
        return;
    }
    const float4 r0 = s_fprims[4 * pos + 0], r2 = s_fprims[4 * pos + 2];
    const v3 d = xf_dir(r0, r1, r2, wd);
    const v3 o = xf_point(r0, r1, r2, wo);
    if (type == 3) {  // sphere
        const float a = vdot(d, d);
        const float b = 2.0f * vdot(d, o);
        const float c = vdot(o, o) - 1.0f;
        const float discr = b * b - 4.0f * a * c;
        if (discr > 0.0f) {
            const float sdiscr = sqrt_cr(discr);
            const float t = div_cr(-b - sdiscr, 2.0f * a);
            if (t > 0.0001f && closer(t, orig, tmin, best)) {
                best.t = t;
                best.pos = pos;
                best.orig = orig;
            }
        }
    } else if (type == 0) {  // cylinder
        const float a = d.x * d.x + d.z * d.z;
        const float b = 2.0f * (o.x * d.x + o.z * d.z);
        const float c = o.x * o.x + o.z * o.z - 1.0f;
        const float discr = b * b - 4.0f * a * c;
        if (discr > 0.001f) {
            const float sdiscr = sqrt_cr(discr);
            const float t0 = div_cr(-b + sdiscr, 2.0f * a);
            const float t1 = div_cr(-b - sdiscr, 2.0f * a);
            float t = 1e16f;
            bool valid = false;
            if (t0 > 0.001f) {
                const float py = o.y + t0 * d.y;
                if (py > -1.0f && py < 1.0f) {
                    t = t0;
                    valid = true;
                }
            }
            if (t1 > 0.001f && t1 < t) {
                const float py = o.y + t1 * d.y;
                if (py > -1.0f && py < 1.0f) {
                    t = t1;
                    valid = true;
                }
            }
            if (valid && closer(t, orig, tmin, best)) {
                best.t = t;
                best.pos = pos;
                best.orig = orig;
            }
        }
    } else {  // disk
        const float divisor = d.y;
        if (!(divisor > 0.0f - 0.01f && divisor < 0.0f + 0.01f)) {
            const float t = div_cr(-o.y, divisor);
            if (t > 0.0001f && closer(t, orig, tmin, best)) {
                const v3 p = vadd(o, vscale(d, t));
                if (vdot(p, p) < 1.0f) {
                    best.t = t;
                    best.pos = pos | kFlat;
                    best.orig = orig;
                }
            }
        }
    }
}

// Two rectangles that the build has put side by side because their normals are opposite (the two faces of a box, floor and
// ceiling, left and right wall).  A rectangle is one-sided: its test starts with d.y < 0 in object space (kernel.cu:394-400),
// and d.y is the ray direction against the world normal, so at most one of the two can get past that first test -- the
// rest of the test then runs once, on whichever it is, instead of twice.  Exactly the tests leaf_test() would make, on the same
// values: when rounding lets BOTH through (a wave-level vote; rare), those lanes run both.
template <bool LIST, typename Ptr>
__device__ __forceinline__ void pair_test(Ptr fp, const float4* __restrict__ lds, int pos, v3 wo, v3 wd, float tmin, FastHit& best)
{
    // fp: where the two normals come from (the up-front list reads them through scalar loads); lds: the LDS copy of the same
    // records, from which each lane then reads the ONE rectangle it goes on with (a per-lane address costs one ds_read; choosing
    // between two rows held in SGPRs costs three VALU instructions per float)
    const float4 r1a = fp[4 * pos + 1], r1b = fp[4 * pos + 5];
    const float dya = r1a.x * wd.x + r1a.y * wd.y + r1a.z * wd.z;
    const float dyb = r1b.x * wd.x + r1b.y * wd.y + r1b.z * wd.z;
    const bool fa = dya < 0.0f, fb = dyb < 0.0f;
    if (__ballot(fa & fb) != 0ull) {
        if (fa & fb) {
            leaf_test(lds, pos, wo, wd, tmin, best);
            leaf_test(lds, pos + 1, wo, wd, tmin, best);
        }
    }
    const int sel = fa ? pos : pos + 1;
    const float4 r1 = LIST ? lds[4 * sel + 1] : make_float4(fa ? r1a.x : r1b.x, fa ? r1a.y : r1b.y, fa ? r1a.z : r1b.z, fa ? r1a.w : r1b.w);
    const int orig = __float_as_int(lds[4 * sel + 3].y);
    rect_commit(lds + 4 * sel, r1, fa ? dya : dyb, fa != fb, sel, orig, wo, wd, tmin, best);
}

// the records [first, first + cnt) of one leaf (or of the up-front list): npairs pairs first, then single primitives
template <bool LIST, typename Ptr>
__device__ __forceinline__ void leaf_range(Ptr fp, const float4* __restrict__ lds, int first, int cnt, int npairs, v3 wo, v3 wd, float tmin, FastHit& best)
{
#pragma unroll 1
    for (int k = 0; k < npairs; ++k) pair_test<LIST>(fp, lds, first + 2 * k, wo, wd, tmin, best);
    if (LIST) {
#pragma unroll 2
        for (int k = 2 * npairs; k < cnt; ++k) leaf_test(fp, first + k, wo, wd, tmin, best);
    } else {
        for (int k = 2 * npairs; k < cnt; ++k) leaf_test(fp, first + k, wo, wd, tmin, best);
    }
}

// Three rectangle pairs that the build has certified as the faces of ONE cuboid -- a box seen from outside (rectangles facing away
// from it: every point of face f lies at or below the plane of every face g of the other two pairs, y_g <= tol in g's object space) or a
// room seen from inside (facing into it: y_g >= -tol).  pair_test's argument says at most one face of a pair is front-facing; this one
// says at most one of the three front-facing faces can be HIT: the point where the ray meets face f's plane has to lie on the inner
// side of the other front-facing planes, or it is outside f's square.  So the first halves of the three tests run as in
// kernel.cu:372-400 (same operations: d.y, o.y, t = -o.y / d.y for the front-facing face of each pair), then a face is dropped when
// that point, o + t_f d, is beyond another front-facing plane by more than `mu` -- y_g(t_f) = o.y_g + t_f d.y_g from the values
// already at hand -- and the second half (the other two rows of M^-1, the unit-square test, kernel.cu:401-414) runs ONCE, on the face
// that is left, instead of three times.  Near an edge two faces can be left: a wave-level vote runs the second half again for those.
// mu covers the certificate's tolerance and the rounding of the reference's u, v and of y_g (rtgo_capi.hip, cub_mu): a face the
// reference accepts is never dropped, and a face that is not dropped gets the reference's whole test, so the closest hit is the same.
// mu_out / mu_in: drop when y > mu_out or y < mu_in (box: (mu, -inf); room: (+inf, -mu)).
template <bool LIST, typename Ptr>
__device__ __forceinline__ void cuboid_range(Ptr fp, const float4* __restrict__ lds, int first, float mu_out, float mu_in, v3 wo, v3 wd, float tmin, FastHit& best)
{
    float t[3], dy[3];
    bool front[3], ok[3], second[3];   // second: the pair's front-facing face is its second record
    bool both = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pos = first + 2 * k;
        const float4 r1a = fp[4 * pos + 1], r1b = fp[4 * pos + 5];
        const float dya = r1a.x * wd.x + r1a.y * wd.y + r1a.z * wd.z;
        const float dyb = r1b.x * wd.x + r1b.y * wd.y + r1b.z * wd.z;
        const bool fa = dya < 0.0f, fb = dyb < 0.0f;
        both = both | (fa & fb);
        second[k] = !fa;
        const float4 r1 = lds[4 * (fa ? pos : pos + 1) + 1];   // (a per-lane read of the one row: cheaper than holding both for selects)
        dy[k] = fa ? dya : dyb;
        const float oyk = r1.x * wo.x + r1.y * wo.y + r1.z * wo.z + r1.w;
        t[k] = div_cr(0.0f - oyk, dy[k]);
        front[k] = fa != fb;
        ok[k] = front[k] & (oyk > 0.0f);
        both = both | (front[k] & !(fabsf(t[k]) < 1e30f));   // (t overflowed: y_g below needs it finite; never seen, handled like `both`)
    }
    // rounding can let both faces of a pair through the sign test (the ray all but parallel to them): those lanes take the six
    // single tests and nothing else
    if (__ballot(both) != 0ull) {
        if (both) {
#pragma unroll 1
            for (int k = 0; k < 6; ++k) leaf_test(lds, first + k, wo, wd, tmin, best);
        }
    }
    auto beyond = [&](int f, int g) {
        // y_g at the point where the ray meets f's plane: o.y_g + t_f d.y_g = d.y_g (t_f - t_g) up to 2^-24 of o.y_g (t_g = -o.y_g / d.y_g
        // correctly rounded), which the margin covers -- and three registers fewer than keeping the o.y
        const float y = dy[g] * (t[f] - t[g]);
        return front[g] & ((y > mu_out) | (y < mu_in));
    };
    bool s0 = ok[0] & !both & !beyond(0, 1) & !beyond(0, 2);
    bool s1 = ok[1] & !both & !beyond(1, 0) & !beyond(1, 2);
    bool s2 = ok[2] & !both & !beyond(2, 0) & !beyond(2, 1);
#pragma unroll 1
    for (;;) {
        const bool any = s0 | s1 | s2;
        if (__ballot(any) == 0ull) break;
        const float tt = s0 ? t[0] : (s1 ? t[1] : t[2]);
        const int ps = first + (s0 ? (second[0] ? 1 : 0) : (s1 ? (second[1] ? 3 : 2) : (second[2] ? 5 : 4)));
        s2 = s2 & (s0 | s1);
        s1 = s1 & s0;
        s0 = false;
        const int orig = __float_as_int(lds[4 * ps + 3].y);
        const bool c = any & (tt > 0.0001f) & closer(tt, orig, tmin, best);
        rect_finish(lds + 4 * ps, tt, c, ps, orig, wo, wd, best);
    }
}

// conservative slab test: t = fma(b, 1/d, -o/d) with the hardware reciprocal
__device__ __forceinline__ bool box_fast(const float4 q0, const float4 q1, v3 id, v3 noid, float tmin, float tmax, float& tn_out)
{
    float t0 = fmaf(q0.x, id.x, noid.x), t1 = fmaf(q1.x, id.x, noid.x);
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = fmaf(q0.y, id.y, noid.y);
    t1 = fmaf(q1.y, id.y, noid.y);
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    t0 = fmaf(q0.z, id.z, noid.z);
    t1 = fmaf(q1.z, id.z, noid.z);
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    tn = fmaxf(tn, tmin);
    tf = fminf(tf, tmax);
    tn_out = tn;
#ifdef RTGO_NO_WIDEN
    return tn <= tf;
#else
    // widen by a few ulps so that the reciprocal's rounding can never drop a box the exact test keeps
    return tn <= tf * 1.000002f + 1e-7f;
#endif
}


// The fast walk in three parts (closest_hit_fast below runs them back to back; an experiment of round 2 ran the middle one in another
// lane than the other two: profiles/r02f/README.md).  fast_list: the up-front list, which also gives the ray its first closest-hit bound.
__device__ __forceinline__ void fast_list(const float4* __restrict__ s_fprims, const float4* __restrict__ g_fprims, int n_small, int n_prims, int n_big_pairs,
                                          int list_cub, float cub_mu, v3 o, v3 d, float tmin, FastHit& best)
{
    // the few "big" primitives (walls, floors; the whole scene when it is tiny) first.  The loop index is wave-uniform and
    // g_fprims is a read-only kernel argument, so the records arrive by scalar loads (s_load_dwordx4) into SGPRs: no LDS
    // traffic, no VGPRs for the matrices, and the loads of the next primitives overlap the tests of the current ones.
    // It also gives every ray a closest-hit bound before it enters the tree.
    if (list_cub != 0) {
        // the room's six walls (or one big box) as a cuboid, the rest of the list after them
        cuboid_range<true>(g_fprims, s_fprims, n_small, list_cub == 1 ? cub_mu : INFINITY, list_cub == 1 ? -INFINITY : -cub_mu, o, d, tmin, best);
        leaf_range<true>(g_fprims, s_fprims, n_small + 6, n_prims - n_small - 6, 0, o, d, tmin, best);
    } else {
        leaf_range<true>(g_fprims, s_fprims, n_small, n_prims - n_small, n_big_pairs, o, d, tmin, best);
    }
}

// A leaf of a tree that holds nothing but spheres (a scene-wide fact the build reports): __intersection__sphere (kernel.cu:250-287)
// without the per-lane dispatch on the primitive's type that leaf_test opens with -- three exec-masked regions per leaf phase that
// a wave of sphere leaves walks through for nothing
__device__ __forceinline__ void sphere_leaf(const float4* __restrict__ s_fprims, int first, int cnt, v3 wo, v3 wd, float tmin, FastHit& best)
{
    for (int k = 0; k < cnt; ++k) {
        const int pos = first + k;
        const float4 r0 = s_fprims[4 * pos + 0], r1 = s_fprims[4 * pos + 1], r2 = s_fprims[4 * pos + 2];
        const int orig = __float_as_int(s_fprims[4 * pos + 3].y);
        const v3 d = xf_dir(r0, r1, r2, wd);
        const v3 o = xf_point(r0, r1, r2, wo);
        const float a = vdot(d, d);
        const float b = 2.0f * vdot(d, o);
        const float c = vdot(o, o) - 1.0f;
        const float discr = b * b - 4.0f * a * c;
        if (discr > 0.0f) {
            const float sdiscr = sqrt_cr(discr);
            const float t = div_cr(-b - sdiscr, 2.0f * a);
            if (t > 0.0001f && closer(t, orig, tmin, best)) {
                best.t = t;
                best.pos = pos;
                best.orig = orig;
            }
        }
    }
}

// fast_tree: the walk proper -- everything in the tree that can beat `best`
__device__ __forceinline__ void fast_tree(const float4* __restrict__ s_fnodes, const float4* __restrict__ s_fprims, unsigned int* __restrict__ s_stack, int bshift,
                                          int n_small, float cub_mu, v3 o, v3 d, float tmin, FastHit& best, unsigned int& dbg_boxes, unsigned int& dbg_tests, bool tree_spheres)
{
    // 1/d for the slab tests.  A direction component that is exactly zero is not rare: the hemisphere sample has sin(phi) = 0
    // whenever its random number is 0 (one ray in 2^24 per bounce, a few per 1080p frame), and cameras can be axis-aligned.
    // rcp(0) = inf would turn fma(b, 1/d, -o/d) into inf - inf = NaN on one side of the origin and -inf on the other, and a
    // box straddling zero would be dropped; a huge FINITE reciprocal keeps both products finite and the slab's sign logic
    // intact (inside: (-huge, +huge); outside: both ends on one side).
    auto safe_rcp = [](float x) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(x), -1e30f, 1e30f); };   // (rcp(+-0) = +-inf and anything beyond 1e30 end up at +-1e30)
    const v3 id = mk(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    const v3 noid = mk(-(o.x * id.x), -(o.y * id.y), -(o.z * id.z));
    float tn;
    float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0;
    bool have = n_small > 0;
    if (have) {
        q0 = s_fnodes[0];
        q1 = s_fnodes[1];
        have = box_fast(q0, q1, id, noid, tmin, best.t, tn);
    }
    int left = __float_as_int(q0.w), right = __float_as_int(q1.w);
    int sp = 0;
    // A stack entry is ONE word: the far child's entry distance cut to its upper 16 bits (toward zero: a lower bound of a
    // positive number, so culling at pop time stays conservative) | its node index (< 2 * kMaxPrims).
    auto pop = [&]() -> bool {
        while (sp > 0) {
            --sp;
            const unsigned int e = s_stack[sp << bshift];
            if (__uint_as_float(e & 0xFFFF0000u) <= best.t) {
                const int idx = (int)(e & 0xFFFFu);
                left = __float_as_int(s_fnodes[2 * idx].w);
                right = __float_as_int(s_fnodes[2 * idx + 1].w);
                return true;
            }
        }
        return false;
    };
    while (have) {
        while (have && right >= 0) {
            const float4 l0 = s_fnodes[2 * left], l1 = s_fnodes[2 * left + 1];
            const float4 h0 = s_fnodes[2 * right], h1 = s_fnodes[2 * right + 1];
            float tl, tr;
#ifdef RTGO_FAST_COUNTERS
#if RTGO_FAST_COUNTERS == 2   /* wave-level: one count per executed node step / leaf phase, whatever the number of live lanes */
            dbg_boxes += (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) ? 1u : 0u;
#else
            dbg_boxes += 2;
#endif
#endif
            const bool hl = box_fast(l0, l1, id, noid, tmin, best.t, tl);
            const bool hr = box_fast(h0, h1, id, noid, tmin, best.t, tr);
            // the step as selects: go to the right child when only it is hit, or when both are and it is nearer; the other one
            // of two hit children waits on the stack
            const bool go_r = hr & (!hl | (tr < tl));
            // (the entry is written whether or not it is needed -- the slot past the top is scratch: the launch allots one entry more -- and
            // only the stack pointer depends on the vote: one exec-masked region less per step; balls -1.7 %, checkered -1.3 %)
            s_stack[sp << bshift] = (__float_as_uint(go_r ? tl : tr) & 0xFFFF0000u) | (unsigned int)(go_r ? left : right);
            sp += (hl & hr) ? 1 : 0;
            if (hl | hr) {
                left = __float_as_int(go_r ? h0.w : l0.w);
                right = __float_as_int(go_r ? h1.w : l1.w);
            } else {
                have = pop();
            }
        }
        if (have) {
            const int first = left, cnt = (-right) & 0xFFF, npairs = ((-right) >> 12) & 0xFF;   // leaf link = -(count | pairs << 12 | cuboid << 20)
#ifdef RTGO_FAST_COUNTERS
#if RTGO_FAST_COUNTERS == 2
            dbg_tests += (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) ? 1u : 0u;
#else
            dbg_tests += (unsigned int)cnt;
#endif
#endif
            if (tree_spheres) sphere_leaf(s_fprims, first, cnt, o, d, tmin, best);
            else if (((-right) >> 20) != 0 && cub_mu > 0.0f) cuboid_range<false>(s_fprims, s_fprims, first, cub_mu, -INFINITY, o, d, tmin, best);
            else leaf_range<false>(s_fprims, s_fprims, first, cnt, npairs, o, d, tmin, best);
            have = pop();
        }
    }
}

// fast_grid: the same job as fast_tree over a uniform grid (Amanatides & Woo's walk, one lane = one ray).  The host bins every small
// primitive into the cells its box -- grown by a pad that is far above the rounding of the walk -- overlaps; a lane steps from cell to
// cell along its ray, tests what the cell lists through the same leaf tests as the tree (so an accepted hit is the tree's and the
// canonical walk's, bit for bit; a primitive met again in the next cell changes nothing: closer() is strict), and stops once its closest
// hit lies before the exit of the cell it is in (everything that could beat it is listed in a cell already visited).  No stack, no box
// tests: ~20 vector instructions per cell where the tree pays ~50 per node pair; scenes of many small, evenly spread primitives (balls).
__device__ __forceinline__ void fast_grid(const float4* __restrict__ s_grid, const float4* __restrict__ s_fprims, const GridParams g, bool spheres,
                                          v3 o, v3 d, float tmin, FastHit& best, unsigned int& dbg_boxes, unsigned int& dbg_tests)
{
    // (everything of `g` into scalars first: it is wave-uniform, and a struct member read inside the loop through a reference went to scratch)
    const float gx = g.min_x, gy = g.min_y, gz = g.min_z, csx = g.cs_x, csy = g.cs_y, csz = g.cs_z, margin = g.margin;
    const int nx = g.nx, ny = g.ny, nz = g.nz, n_cells = g.n_cells;
    // the table: one word per cell, 0 = lists nothing, else 1 + the index of the cell's record = two float4: the box around everything the
    // cell lists (min.xyz | first item + count << 16, max.xyz | 0); then the items, 16-bit positions into fprims
    const unsigned int* __restrict__ cells = reinterpret_cast<const unsigned int*>(s_grid);
    const float4* __restrict__ recs = s_grid + g.rec_off4;
    const unsigned short* __restrict__ items = reinterpret_cast<const unsigned short*>(s_grid + g.items_off4);
    const float idx = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e30f, 1e30f);   // (see fast_tree on 1 / 0)
    const float idy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e30f, 1e30f);
    const float idz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e30f, 1e30f);
    // the ray's stretch inside the grid's bounds
    float t0 = tmin, t1 = best.t;
    {
        const float ax = (gx - o.x) * idx, bx = (gx + csx * (float)nx - o.x) * idx;
        const float ay = (gy - o.y) * idy, by = (gy + csy * (float)ny - o.y) * idy;
        const float az = (gz - o.z) * idz, bz = (gz + csz * (float)nz - o.z) * idz;
        t0 = fmaxf(fmaxf(t0, fminf(ax, bx)), fmaxf(fminf(ay, by), fminf(az, bz)));
        t1 = fminf(fminf(t1, fmaxf(ax, bx)), fminf(fmaxf(ay, by), fmaxf(az, bz)));
    }
    bool live = (n_cells > 0) & (t0 <= t1 * 1.000002f + margin);
    // the cell of the entry point (clamped: the point may sit a rounding outside), the parameter at which the ray leaves it on each
    // axis and the parameter per cell.  The table has a border of empty cells around the nx x ny x nz that list something: the step
    // out of the grid lands there and the walk ends on the parameter test alone, without a count of cells per axis (a ray leaves through
    // a face, an edge or a corner: at most one step per axis beyond t1, all inside the border).
    const int cx = min(max((int)floorf((o.x + d.x * t0 - gx) * g.ics_x), 0), nx - 1);
    const int cy = min(max((int)floorf((o.y + d.y * t0 - gy) * g.ics_y), 0), ny - 1);
    const int cz = min(max((int)floorf((o.z + d.z * t0 - gz) * g.ics_z), 0), nz - 1);
    const bool fx = idx >= 0.0f, fy = idy >= 0.0f, fz = idz >= 0.0f;
    float tmx = (gx + csx * (float)(cx + (fx ? 1 : 0)) - o.x) * idx;
    float tmy = (gy + csy * (float)(cy + (fy ? 1 : 0)) - o.y) * idy;
    float tmz = (gz + csz * (float)(cz + (fz ? 1 : 0)) - o.z) * idz;
    const float tdx = csx * fabsf(idx), tdy = csy * fabsf(idy), tdz = csz * fabsf(idz);
    const int NX = nx + 2, NXY = NX * (ny + 2);
    const int sx = fx ? 1 : -1, sy = fy ? NX : -NX, sz = fz ? NXY : -NXY;
    int cell = (cz + 1) * NXY + (cy + 1) * NX + cx + 1;
    // A ray this walk cannot take -- an infinite or NaN component (its reciprocal, and with it the parameter per cell, is zero or no number:
    // te would never grow), or a direction so long that a cell is crossed in less than the margin (the border argument above needs
    // td > margin) -- is dropped here, once (1.4 % of balls' frame; a range test per step cost 4 %): every step then adds a positive td to the
    // parameter it compares, so the loop ends, and it ends inside the table.  Such rays only exist where the arithmetic has broken down
    // (far beyond the far-field guard, where the launch walks the canonical tree anyway).
    live = live & (tdx > 2.0f * margin) & (tdy > 2.0f * margin) & (tdz > 2.0f * margin) & (fabsf(tmx + tmy + tmz) < 3e38f);
    float tstop = fminf(best.t, t1) + margin;   // the walk goes on while the current cell's exit lies before this
    int last = -1;                              // the record tested last: a shape that straddles two cells along the ray is listed in both
    // A lane stops at a cell only when its ray meets the box around what the cell lists (the conservative slab test of the tree walk):
    // a sphere fills a tenth of its cell, and every stop is a test phase in which the lanes that did not stop wait.
    const v3 id = mk(idx, idy, idz);
    const v3 noid = mk(-(o.x * idx), -(o.y * idy), -(o.z * idz));
    unsigned int fc = 0u;                       // first item | count << 16 of the cell this lane has stopped at, 0 = it has not
    auto look = [&]() {
        const unsigned int c = cells[cell];
        if (c != 0u) {
            const float4 q0 = recs[2u * c - 2u], q1 = recs[2u * c - 1u];
            float tn;
            fc = box_fast(q0, q1, id, noid, tmin, best.t, tn) ? __float_as_uint(q0.w) : 0u;
        }
    };
    if (live) look();
    while (live) {
        // to the next cell that lists something in the ray's way: out of the current one through the nearest of its three far planes
        while (live && fc == 0u) {
#ifdef RTGO_FAST_COUNTERS
#if RTGO_FAST_COUNTERS == 2   /* wave-level: one count per executed iteration, whatever the number of live lanes */
            dbg_boxes += (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) ? 1u : 0u;
#else
            dbg_boxes += 1;
#endif
#endif
            const bool ux = (tmx <= tmy) & (tmx <= tmz), uy = !ux & (tmy <= tmz);
            const float te = ux ? tmx : (uy ? tmy : tmz);
            live = te <= tstop;
            tmx += ux ? tdx : 0.0f;
            tmy += uy ? tdy : 0.0f;
            tmz += (ux | uy) ? 0.0f : tdz;
            cell += ux ? sx : (uy ? sy : sz);
            if (live) look();
        }
        if (live) {
            const int first = (int)(fc & 0xFFFFu), cnt = (int)(fc >> 16);
            for (int k = 0; k < cnt; ++k) {
                const int pos = (int)items[first + k];
#if defined(RTGO_FAST_COUNTERS) && RTGO_FAST_COUNTERS == 2
                dbg_tests += (__ffsll((long long)__ballot(true)) - 1 == (int)(threadIdx.x & 63u)) ? 1u : 0u;
#endif
                if (pos != last) {
#if defined(RTGO_FAST_COUNTERS) && RTGO_FAST_COUNTERS != 2
                    dbg_tests += 1;
#endif
                    if (spheres) sphere_leaf(s_fprims, pos, 1, o, d, tmin, best);
                    else leaf_test(s_fprims, pos, o, d, tmin, best);
                }
                last = pos;
            }
            tstop = fminf(best.t, t1) + margin;
            fc = 0u;   // (back into the stepping loop)
        }
    }
}

// fast_winner: the closest hit's record for the closest-hit program (t, SBT index, world normal)
__device__ __forceinline__ bool fast_winner(const float4* __restrict__ s_fprims, v3 o, v3 d, float tmax, const FastHit& best, Hit& out)
{
    out.prim = -1;
    out.t = tmax;
    out.n = mk(0.0f, 0.0f, 0.0f);
    if (best.pos < 0) return false;
    const int wpos = best.pos & (kFlat - 1);
    const bool flat = (best.pos & kFlat) != 0;
    const float4 r0 = s_fprims[4 * wpos + 0], r1 = s_fprims[4 * wpos + 1], r2 = s_fprims[4 * wpos + 2];
    out.t = best.t;
    // the object-space normal of the winner only (kernel.cu:270 sphere, :315 cylinder, :345/:388 disk and rectangle), from the same
    // object-space ray and the same t as its test: candidates that were overtaken never needed one
    v3 nobj = mk(0.0f, 1.0f, 0.0f);
    if (!flat) {
        const v3 od = xf_dir(r0, r1, r2, d);
        const v3 oo = xf_point(r0, r1, r2, o);
        const bool sphere = __float_as_int(s_fprims[4 * wpos + 3].x) == 3;
        const v3 at = sphere ? vadd(oo, vscale(od, best.t)) : mk(oo.x + best.t * od.x, 0.0f, oo.z + best.t * od.z);
        nobj = sphere ? vnormalize(at) : at;
    }
    out.n = xf_normal(r0, r1, r2, nobj);   // (flat winners: the closest-hit code takes the primitive's precomputed frame instead and this is dead code there)
    out.prim = best.orig;
    return true;
}

template <bool GRID>
__device__ __forceinline__ bool closest_hit_fast(const float4* __restrict__ s_fnodes, const float4* __restrict__ s_fprims,
                                                 const float4* __restrict__ g_fprims, const GridParams grid,
 unsigned int* __restrict__ s_stack, int bshift,
                                                 int n_small, int n_prims, int n_big_pairs, int list_cub, float cub_mu, bool tree_spheres, v3 o, v3 d, float tmin, float tmax, Hit& out,
                                                 unsigned int& dbg_boxes, unsigned int& dbg_tests
#ifdef RTGO_TIMELINE
                                                 , unsigned long long& tl_big, unsigned long long& tl_tree
#endif
)
{
#ifdef RTGO_TIMELINE
    const unsigned long long tl_s0 = wall_clock64();
#endif
    FastHit best;
    best.t = tmax;
    best.pos = -1;
    best.orig = -1;
    fast_list(s_fprims, g_fprims, n_small, n_prims, n_big_pairs, list_cub, cub_mu, o, d, tmin, best);
#if defined(RTGO_FAST_COUNTERS) && RTGO_FAST_COUNTERS != 2
    dbg_tests += (unsigned int)(n_prims - n_small);
#endif
#ifdef RTGO_TIMELINE
    const unsigned long long tl_s1 = wall_clock64() + (best.pos == 12345 ? 1 : 0);
    tl_big += tl_s1 - tl_s0;
#endif
    if constexpr (GRID) fast_grid(s_fnodes, s_fprims, grid, tree_spheres, o, d, tmin, best, dbg_boxes, dbg_tests);
    else fast_tree(s_fnodes, s_fprims, s_stack, bshift, n_small, cub_mu, o, d, tmin, best, dbg_boxes, dbg_tests, tree_spheres);
#ifdef RTGO_TIMELINE
    tl_tree += wall_clock64() + (best.pos == 12345 ? 1 : 0) - tl_s1;
#endif
    return fast_winner(s_fprims, o, d, tmax, best, out);
}

// acos(pow(base, expo)) with both steps in f64, each rounded to float like the reference's float calls.  Kept out of line:
// the f64 libm bodies need ~80 VGPRs that would otherwise be charged to every wave of the megakernel.
__device__ __attribute__((noinline)) float glossy_theta(float base, float expo)
{
    // x^e as exp(e * log x): each f64 call is good to ~1e-16 relative, |e log x| < 20, so the product carries ~1e-14 -- far
    // inside the 3e-8 half-ulp of the float it is rounded to, at about half the cost of the extended-precision f64 pow
    const float c = (float)exp((double)expo * log((double)base));
    return (float)acos((double)c);
}

// GetRayOnHemisphere, kernel.cu:101-122
// UNIT_DIR: the caller's `direction` is a unit vector already (a normalised normal): kernel.cu:103 normalises it again, which moves it by
// an ulp at most; the default build takes it as it is.
// Diffuse lobe (coefficient 0: every path-mode bounce, kernel.cu:467): theta = acos(1 - r2) and then cos(theta), sin(theta) -- i.e.
// cos = 1 - r2 (exact in float: r2 = k / 2^24) and sin = sqrt(r2 (2 - r2)), each within an ulp or two of what acosf / sinf / cosf
// return, for a fifth of the instructions.  Both are inside the tolerance the contract states (1e-4 on >= 99 % of the pixels;
// the reference's own build is --use_fast_math, CMakeLists.txt:165-170); the oracle keeps the libm calls.  Both walks share this code,
// so fast == canonical stays bit for bit.  -DRTGO_LITERAL_SHADING: the literal forms (acosf, the second normalisation), as in round 2.
// LEAN is set by path mode only: a diffuse bounce forgets the incoming direction, so an ulp in the hit point stays an ulp.  Distributed
// mode keeps the literal forms: its mirror and glossy bounces off small spheres multiply a direction's last bit by ~2 d / r per bounce
// (balls: 3.5 % of the pixels of a 96 x 64 frame left the tolerance when it ran lean, profiles/r03c).
// have_x / Xpre: the tangent X of this direction is known already (a flat primitive's frame from build_kernel, bit for bit the value
// computed here): lanes that have it skip the normalisation (a wave vote skips it altogether when every lane has).
constexpr int kHemisphereMaxTries = 1024;   // (= ORACLE_HEMISPHERE_MAX_TRIES)
template <bool LEAN_IN>
__device__ __forceinline__ v3 hemisphere(v3 normal, v3 direction, float coefficient, unsigned int& seed, bool have_x = false, v3 Xpre = v3{0.0f, 0.0f, 0.0f})
{
    v3 ray;
#ifdef RTGO_LITERAL_SHADING
    constexpr bool LEAN = false;
#else
    constexpr bool LEAN = LEAN_IN;
#endif
    const v3 Y = LEAN ? direction : vnormalize(direction);   // (lean callers pass a unit vector: a normalised normal)
    v3 X;
    if (have_x) X = Xpre;   // (wave-uniform: LaunchParams::all_flat)
    else X = vnormalize(mk(Y.y - Y.z, -Y.x, Y.x));
    const v3 Z = vcross(Y, X);
    const float expo = div_cr(1.f, coefficient + 1.f);
    // The reference's rejection loop (kernel.cu:109-120) is unbounded, and it never ends when the lobe lies wholly below the horizon of
    // `normal`: kernel.cu:443-447 flips N by V = normalize(origin - x), which is rounding noise when t is tiny against the coordinates, and
    // Rr = reflect about a wrongly flipped N (:508-510) points into the surface -- a mirror's lobe around it never passes the test and the
    // launch hangs the GPU (tools/fuzz_farfield.py, random scene 45 seen 270 units off the origin).  kHemisphereMaxTries draws, then the
    // last one stands; the oracle does the same, and wherever the reference's loop ends within that many draws nothing changes.
    int tries = 0;
    do {
        const float r1 = rnd(seed);
        const float r2 = rnd(seed);
        const float phi = 2.f * kPi * r1;
        const float base = 1.f - r2;
        float st, ct, sp, cp;
        if (LEAN && expo == 1.0f) {
            ct = base;
            st = sqrt_cr(r2 * (1.0f + base));
        } else {
            float theta;
            if (expo == 1.0f) {
                // diffuse lobe: powf(x, 1) == x exactly in any sound libm
                theta = acosf(base);
            } else if (expo == 0.5f) {
                // specularity 1 (every cornell surface in distributed mode): pow(x, 1/2) is sqrt(x), which IS correctly rounded on
                // the device, and the lobe is as wide as the diffuse one, so float acosf is as benign here as it is there
                theta = acosf(sqrt_cr(base));
            } else {
                // glossy lobe: acos(pow(x, 1/(coef+1))) sits at the ill-conditioned end of acos (argument within 1e-4 of 1),
                // where one ulp of pow moves theta by ~1e-3 relative.  Evaluate both in f64 and round, which reproduces a
                // correctly rounded float libm (tools/libm_probe: <0.02 % differing results vs 12 % / 28 % for the f32 forms).
                theta = glossy_theta(base, expo);
            }
            // sincosf shares the range reduction and returns bit for bit what sinf and cosf return (tools/sincos_probe.hip); sincos_cr is
            // its small-argument path alone
            sincos_cr(theta, &st, &ct);
        }
        sincos_cr(phi, &sp, &cp);
        ray = vsub(vadd(vscale(X, st * cp), vscale(Y, ct)), vscale(Z, st * sp));
    } while (vdot(normal, ray) < 0.f && ++tries < kHemisphereMaxTries);
    return ray;
}

// `dot(N, normalize(w)) < 0` (kernel.cu:437-441: V = normalize(ray.origin - x); if (dot(N, V) < 0) N = -N) without the
// normalisation when the sign is beyond doubt.  V_i = fl(w_i * inv) with inv > 0 carries one rounding, each product and each of
// the two additions one more: the computed dot differs from dot(N, w) * inv by less than 5e-7 of sum |N_i w_i| * inv, so
// beyond 1e-6 of that sum its sign is the sign of dot(N, w).  Closer calls (grazing within a microradian, NaN, w = 0) are
// decided by the reference's own expression.  (Saves a sqrt and a division per hit.)
__device__ __forceinline__ bool faces_away(v3 N, v3 w)
{
    const float ax = N.x * w.x, ay = N.y * w.y, az = N.z * w.z;
    const float s = ax + ay + az;
    const float sa = fabsf(ax) + fabsf(ay) + fabsf(az);
    if (fabsf(s) > 1e-6f * sa) return s < 0.0f;
    return vdot(N, vnormalize(w)) < 0.0f;
}

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }  // vec_math.h:115-118

__device__ __forceinline__ unsigned int wave_sum(unsigned int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// running average + 8-bit image of one pixel: kernel.cu:236-246
__device__ __forceinline__ void write_pixel(const LaunchParams& p, size_t idx, v3 cur)
{
    if (p.frame > 0) {
        const float4 prev4 = p.accum[idx];
        const v3 prev = mk(prev4.x, prev4.y, prev4.z);
        const float ratio = 1.0f / (float)(p.frame + 1);
        cur = vadd(prev, vscale(vsub(cur, prev), ratio));  // lerp, vec_math.h:496-499
    }
    p.accum[idx] = make_float4(cur.x, cur.y, cur.z, 1.0f);
    // make_color, kernel.cu:90-98
    p.image[idx] = make_uchar4((unsigned char)(clampf(cur.x, 0.0f, 1.0f) * 255.0f), (unsigned char)(clampf(cur.y, 0.0f, 1.0f) * 255.0f),
                               (unsigned char)(clampf(cur.z, 0.0f, 1.0f) * 255.0f), 255u);
}

// Cold segment s -> (local row, first window column, column limit); see LaunchParams.  Wave-uniform.
__device__ __forceinline__ void cold_segment(const LaunchParams& p, unsigned int s, unsigned int& lr, unsigned int& x, unsigned int& lim)
{
    unsigned int j = s;
    const unsigned int below = p.hot_y0 * p.segs_full;
    if (j < below) {
        lr = j / p.segs_full;
        x = (j - lr * p.segs_full) * 64u;
        lim = p.w;
        return;
    }
    j -= below;
    const unsigned int above = p.rows_above * p.segs_full;
    if (j < above) {
        const unsigned int r = j / p.segs_full;
        lr = p.hot_y0 + p.hot_h + r;
        x = (j - r * p.segs_full) * 64u;
        lim = p.w;
        return;
    }
    j -= above;
    const unsigned int left = p.hot_h * p.segs_l;
    if (j < left) {
        const unsigned int r = j / p.segs_l;
        lr = p.hot_y0 + r;
        x = (j - r * p.segs_l) * 64u;
        lim = p.cold_x0;
        return;
    }
    j -= left;
    const unsigned int r = j / p.segs_r;
    lr = p.hot_y0 + r;
    x = p.cold_x1 + (j - r * p.segs_r) * 64u;
    lim = p.w;
}

// =====================================================================================================================
// The render megakernel.  Persistent workgroups; one lane = one PATH (see the decomposition note inside); each loop iteration
// traces exactly ONE ray per live lane (primary, bounce or shadow), so the 64 lanes of a wave stay at the same bounce and
// share the traversal and shading code.  Replaces __raygen__rg + optixTrace + the closest-hit/miss programs of kernel.cu.
// PATH = Params::enablePathTracing.  STATS = false: the fast walk (timed kernel).  STATS = true: the canonical LBVH walk with
// the V/T/h counters that define the roofline's algorithmic bytes; both produce the same pixels bit for bit.
// =====================================================================================================================
// WPE = waves per SIMD the register allocation targets: 4 (<= 128 VGPRs) for scenes whose LDS image limits a CU to 16 waves
// anyway, 5 (<= 96 VGPRs, per-level path records in LDS) for small scenes, where the fifth wave buys more than the tighter
// budget costs (rtgo_capi.hip picks per launch; kRenderKernels there lists every instantiation).
#ifndef RTGO_STREAM_WINDOW
#define RTGO_STREAM_WINDOW 4
#endif
constexpr int kStreamWindow = RTGO_STREAM_WINDOW;   // STREAM: passes a lane may run ahead of the oldest pass that is still open (192 floats of LDS per wave each)

// COUNT: the canonical walk's V/T/h counters (collect_stats launches); the same walk without them serves launches beyond the
// far-field guard, where it is the product path.
// FRAMES: the scene holds flat primitives only (cornell, checkered): closest-hit takes N and the sampling tangent from the frames
// build_kernel computed (bit for bit the per-hit values); an instantiation of its own, so that the other scenes' code is untouched.
template <bool PATH, bool STATS, int WPE, bool STREAM, bool COUNT = STATS, bool FRAMES = false, bool GRID = false>
__global__ __launch_bounds__(kMaxBlock) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void render_kernel(const LaunchParams p, const float4* __restrict__ g_fprims)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS image.  STATS (canonical, instrumented walk): [nodes 2/node][prims 6/prim, SBT order][frames 2/prim][stack][lights]
    //             fast walk (the timed kernel):          [fnodes 2/node][fprims 4/prim, Morton order][materials 3/prim][frames 2/prim][stack, 4 B/entry][lights]
    constexpr int MS = STATS ? 6 : 3;  // float4 stride between two primitives' material rows (kd|spec, kr|type, Le)
    const int n_nodes = STATS ? p.n_nodes : p.n_fnodes;
    float4* s_nodes = reinterpret_cast<float4*>(smem);
    float4* s_prims = s_nodes + 2 * n_nodes;
    float4* s_mat_w = STATS ? s_prims + 3 : s_prims + 4 * p.n_prims;
    float4* s_frame_w = STATS ? s_prims + 6 * p.n_prims : s_mat_w + 3 * p.n_prims;   // shading frames of the flat primitives, 2 float4 per primitive, SBT order
    float4* s_end = s_frame_w + (FRAMES ? 2 * p.n_prims : 0);   // (only the instantiation that uses them pays for them)
    const float4* s_frame = s_frame_w;
    float2* s_stack_base = reinterpret_cast<float2*>(s_end);
    const int stack_depth = STATS ? kStackDepth : p.stack_depth;
    const int kBlock = (int)blockDim.x;           // 256, 512 or 1024
    const int bshift = 31 - __clz(kBlock);        // per-lane stack entry e lives at [e << bshift]
    // (entries x workgroup size x 8 bytes for the canonical walk, x 4 for the fast walk's packed words: a multiple of 1 KiB either way)
    LightRec* s_lights = reinterpret_cast<LightRec*>(reinterpret_cast<unsigned char*>(s_stack_base) + (size_t)stack_depth * kBlock * (STATS ? 8 : 4));
    const float4* s_mat = s_mat_w;
    // small scenes (the 5-waves-per-SIMD variant): per-level path records live in LDS, [4 words x kMaxLevels][lane], instead of
    // 15-20 VGPRs -- that is what lets the allocation fit 96 registers without spilling to scratch
    constexpr bool LVLDS = (WPE >= 5);
    constexpr int LVW = PATH ? 3 : 4;   // words per level record: the weight (path) / the term and the primitive (distributed)
    // raygen constants (eye, U, V, W, image size, sample step): read from LDS where a sample starts, instead of sitting in
    // registers through the ray loop
    float* s_cam = reinterpret_cast<float*>(s_lights + kMaxLights);
    // per sample k of a pixel (the first kSampleTab of them): the LCG's 2k-step map (A, C) -- the sample's jitter starts from A * seed + C,
    // the pixel's tea<16> seed advanced by 2k draws -- and the sample's cell (i, j) = (k / N, k % N) of the N x N jitter grid.  They depend on
    // k alone: a table instead of ~70 instructions of squaring loop and an integer division wherever a sample starts (most of a primary ray's
    // cost where primary rays are most rays: plateau 4K spp 256 17.2 -> 16.4 ms, mirror_spheres 4K spp 64 14.6 -> 14.3, cornell -1 %)
    uint4* s_tab = reinterpret_cast<uint4*>(s_cam + 16);
    const unsigned int n_tab = (unsigned int)(p.sqrt_spp * p.sqrt_spp) < (unsigned int)kSampleTab ? (unsigned int)(p.sqrt_spp * p.sqrt_spp) : (unsigned int)kSampleTab;
    float* s_lv = s_cam + 16 + 4 * n_tab + threadIdx.x;
    // STREAM: the payload window of this wave (4 passes x 3 channels x 64 lanes), behind the level records
    float* s_win_base = s_cam + 16 + 4 * n_tab + (LVLDS ? (int)blockDim.x * LVW * kMaxLevels : 0) + 192 * kStreamWindow * (threadIdx.x >> 6);

    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < kQueues) p.queue_next[kQueueStride * (unsigned int)tid] = 0u;
#ifdef RTGO_TIMELINE
    const unsigned long long tl_t0 = wall_clock64();
    unsigned long long tl_t1 = 0, tl_first = 0, tl_lanes = 0, tl_qwait = 0, tl_cold = 0;
    unsigned int tl_hot = 0;
    unsigned long long tl_a = 0, tl_b = 0, tl_c = 0, tl_d = 0, tl_big = 0, tl_tree = 0, tl_loop = 0;
    unsigned long long tls_regen = 0, tls_trace = 0, tls_shade = 0, tls_lanes_trace = 0, tls_lanes_regen = 0, tls_regens = 0;   // streaming loop (tools/timeline_stream.py)
    unsigned int tl_units = 0, tl_iters = 0;
#endif
    if (STATS) {
        for (int i = tid; i < 2 * p.n_nodes; i += kBlock) s_nodes[i] = p.nodes[i];
        for (int i = tid; i < 6 * p.n_prims; i += kBlock) s_prims[i] = p.prims[i];
    } else {
        for (int i = tid; i < 2 * n_nodes; i += kBlock) s_nodes[i] = p.fnodes[i];
        for (int i = tid; i < 4 * p.n_prims; i += kBlock) s_prims[i] = p.fprims[i];
        for (int i = tid; i < 3 * p.n_prims; i += kBlock) s_mat_w[i] = p.prims[6 * (i / 3) + 3 + (i % 3)];
    }
    if (FRAMES)
        for (int i = tid; i < 2 * p.n_prims; i += kBlock) s_frame_w[i] = p.frames[i];
    for (unsigned int k = threadIdx.x; k < n_tab; k += blockDim.x) {
        const unsigned int si = k / (unsigned int)p.sqrt_spp;
        s_tab[k] = make_uint4(lcg_skip(1u, 2u * k) - lcg_skip(0u, 2u * k), lcg_skip(0u, 2u * k), si, k - si * (unsigned int)p.sqrt_spp);   // A = map(1) - map(0), C = map(0)
    }
    if (threadIdx.x == 0) {
        s_cam[0] = p.eye.x; s_cam[1] = p.eye.y; s_cam[2] = p.eye.z; s_cam[3] = (float)p.W;
        s_cam[4] = p.U.x; s_cam[5] = p.U.y; s_cam[6] = p.U.z; s_cam[7] = (float)p.H;
        s_cam[8] = p.V.x; s_cam[9] = p.V.y; s_cam[10] = p.V.z; s_cam[11] = 1.0f / (float)p.sqrt_spp;
        s_cam[12] = p.Wv.x; s_cam[13] = p.Wv.y; s_cam[14] = p.Wv.z; s_cam[15] = 0.0f;
    }
    {
        const float* src = reinterpret_cast<const float*>(p.lights);
        float* dst = reinterpret_cast<float*>(s_lights);
        for (int i = tid; i < p.n_lights * 16; i += kBlock) dst[i] = src[i];
    }
    __syncthreads();

#ifdef RTGO_TIMELINE
    tl_t1 = wall_clock64();
#endif
    float2* s_stack = s_stack_base + tid;                                                 // canonical walk: (distance, node) entries
    unsigned int* s_stack4 = reinterpret_cast<unsigned int*>(s_stack_base) + tid;         // fast walk: one packed word per entry
    const int lane = tid & 63;
    const unsigned int nn = (unsigned int)(p.sqrt_spp * p.sqrt_spp);
    // Work decomposition: ONE LANE = ONE PATH.  A wave takes "units" of 64/nn_eff neighbouring pixels of a row and runs
    // nn_eff = min(nn, 16) samples of each side by side, in ceil(nn / nn_eff) passes.  The samples of a pixel are independent
    // given the LCG state their jitter starts from (trace passes the seed by value, kernel.cu:46-79), which is the pixel's
    // tea<16> seed advanced by 2k draws; their results are then summed IN SAMPLE ORDER (kernel.cu:232), so the pixel is bit
    // for bit what the reference's sequential loop gives.  Against one-lane-per-pixel this keeps the 64 lanes at the same
    // bounce, and makes the unit of scheduling 16x smaller at 16 spp (the frame has only ~2 in-scene 64-pixel tiles per
    // resident wave: whole waves idle behind the last ones, and with the frame split over 8 GPUs most waves never get one).
    const unsigned int nn_eff = nn < (unsigned int)kSamplesPerPass ? nn : (unsigned int)kSamplesPerPass;  // samples of one pixel that share a pass
    const unsigned int P = 64u / nn_eff;                        // pixels per unit
    const unsigned int passes = (nn + nn_eff - 1u) / nn_eff;
    const unsigned int pl = (unsigned int)lane / nn_eff;        // this lane's pixel within the unit
    const unsigned int kl = (unsigned int)lane - pl * nn_eff;   // this lane's sample within the pass
    const unsigned int group_base = (pl < P ? pl : 0u) * nn_eff;

    unsigned int c_rays = 0, c_occl = 0, c_nodes = 0, c_tests = 0, c_hits = 0;
#ifdef RTGO_STREAM_STATS
    unsigned long long ss_iter = 0, ss_trace = 0, ss_shade_rounds = 0, ss_shade = 0, ss_stall = 0;   // diagnostic: streaming-loop census
#endif

    // Work queue: kQueues heads, 64 bytes apart; head q serves the entries u with u % kQueues == q.  A wave pulls from the head
    // blockIdx % kQueues and, when that runs dry, from the others.  One head saturates at ~88 dequeues/us chip-wide
    // (MI355X_MICROARCH "dequeue"), which 5120 waves on 64-path units exceed several times over; heads on different lines
    // proceed side by side.  Results do not depend on who takes what.
    unsigned int q = blockIdx.x % (unsigned int)kQueues;
    // The FIRST strip of every wave is assigned statically (its rank among the waves of its queue): 5120 waves pulling at once
    // would queue up behind the heads for 10-30 us.  Head q therefore counts from n_static(q) = the waves on queue q.
    const unsigned int wpb = (unsigned int)kBlock >> 6;
    auto n_static = [&](unsigned int qq) { return ((gridDim.x + (unsigned int)kQueues - 1u - qq) / (unsigned int)kQueues) * wpb; };
    // the pull for the NEXT strip is issued before the current one is processed, so its ~1-2 us round trip hides behind work
    // (the head's offset is added when the value is USED: arithmetic on it here would make the wave wait for the atomic at once)
    unsigned int pending = (blockIdx.x / (unsigned int)kQueues) * wpb + ((unsigned int)tid >> 6), pending_off = 0u;
    for (;;) {
        const unsigned int q_count = (p.n_tiles + (unsigned int)kQueues - 1u - q) / (unsigned int)kQueues;   // units in queue q
#ifdef RTGO_TIMELINE
        const unsigned long long tl_q0 = wall_clock64();
#endif
        const unsigned int first = __builtin_amdgcn_readfirstlane(pending) + pending_off;
#ifdef RTGO_TIMELINE
        tl_qwait += wall_clock64() - tl_q0;
        if (tl_a == 0) tl_a = wall_clock64();
#endif
        if (first >= q_count) {
            // own head is past its end: look at all heads at once (one load, lanes 0..7) and move to one that still has work.
            // Heads only grow, so "none has work" is final: the wave leaves and the grid drains.
            unsigned int head = 0xFFFFFFFFu, cnt_l = 0u;
            if (lane < kQueues) {
                head = __hip_atomic_load(p.queue + kQueueStride * (unsigned int)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + n_static((unsigned int)lane);
                cnt_l = (p.n_tiles + (unsigned int)kQueues - 1u - (unsigned int)lane) / (unsigned int)kQueues;
            }
            const unsigned long long open = __ballot(head < cnt_l);
            if (open == 0ull) break;
            q = (unsigned int)(__ffsll((long long)open) - 1);
            // (rare path: consume the result at once, so that no write to its register is pending where the paths join --
            // the compiler would otherwise wait for the common path's prefetch there as well)
            unsigned int stolen = 0;
            if (lane == 0) stolen = atomicAdd(p.queue + kQueueStride * q, 1u);
            pending = __builtin_amdgcn_readfirstlane(stolen);
            pending_off = n_static(q);
            continue;
        }
        if (lane == 0) pending = atomicAdd(p.queue + kQueueStride * q, 1u);
        pending_off = n_static(q);
        // one queue entry = one STRIP: p.grab units side by side on a row (at most 64 pixels).  The strip's tea<16> pixel seeds
        // are computed once, one pixel per lane (the hash is 16 dependent rounds: ~160 instructions whether 4 or 64 lanes need
        // it), and handed to the units by lane exchange.
        const unsigned int strip = first * (unsigned int)kQueues + q;
        if (strip >= p.n_hot) {
            // cold chunk: pixels no primary ray of which can reach the scene's bounds.  Each of their N*N samples is one ray that
            // misses (__miss__ms, kernel.cu:419-423), so the pixel is the in-order sum of N*N background colours / (N*N): p.bg_pixel.
            const unsigned int s0 = (strip - p.n_hot) * p.cold_cs;
            const unsigned int s1 = s0 + p.cold_cs < p.n_cold_segs ? s0 + p.cold_cs : p.n_cold_segs;
            for (unsigned int s = s0; s < s1; ++s) {
                unsigned int clr, cx, clim;
                cold_segment(p, s, clr, cx, clim);
                const unsigned int lx = cx + (unsigned int)lane;
                if (lx < clim) {
                    write_pixel(p, (size_t)clr * p.w + lx, p.bg_pixel);
                    c_rays += nn;
                }
            }
            continue;
        }
        const unsigned int lr = p.hot_y0 + strip / p.hot_w;   // local (compact) row
        const unsigned int sx = p.hot_x0 + (strip - (lr - p.hot_y0) * p.hot_w);   // strip column
        if (p.hot_mask != nullptr && ((p.hot_mask[strip >> 5] >> (strip & 31u)) & 1u) == 0u) {
            // a strip of the rectangle that no primitive's own screen rectangle reaches (the word is wave-uniform: a scalar load)
            const unsigned int lx0 = sx * p.grab * P + (unsigned int)lane;
            if ((unsigned int)lane < p.grab * P && lx0 < p.w) {
                write_pixel(p, (size_t)lr * p.w + lx0, p.bg_pixel);
                c_rays += nn;
            }
            continue;
        }
        // local row -> window row under the band interleave
        const unsigned int band = lr / p.band_h;
        const unsigned int wrow = (band * p.n_ranks + p.rank) * p.band_h + (lr - band * p.band_h);
        const unsigned int gy = p.y0 + wrow;
        const float fy = (float)gy;
        const unsigned int strip_x0 = sx * p.grab * P;
        const unsigned int strip_seed = tea16(p.W * gy + (p.x0 + strip_x0 + (unsigned int)lane), p.frame);
#ifdef RTGO_TIMELINE
        if (tl_b == 0) tl_b = wall_clock64() + (strip_seed == 0x12345u ? 1 : 0);
#endif
#pragma unroll 1
        for (unsigned int ui = 0; ui < p.grab; ++ui) {
        const unsigned int lx = strip_x0 + ui * P + pl;
        if (strip_x0 + ui * P >= p.w) break;
        const bool in_range = pl < P && lx < p.w;
        const unsigned int gx = p.x0 + lx;
        const float fx = (float)gx;

        // __raygen__rg (kernel.cu:184-247)
        const unsigned int pix0 = (unsigned int)__shfl((int)strip_seed, (int)((ui * P + pl) & 63u), 64);
        v3 color = mk(0.0f, 0.0f, 0.0f);
        if constexpr (!STREAM) {
#pragma unroll 1
        for (unsigned int pass = 0; pass < passes; ++pass) {
        const unsigned int k = pass * nn_eff + kl;   // sample index: i-major, k = i*N + j (kernel.cu:206-208)
        bool active;
        int depth;
        int phase;                     // distributed mode: 0 = radiance ray in flight, 1 = shadow ray in flight
        unsigned int seed;
        v3 ro, rd;
        float tmin, tmax;
        v3 result;                     // payload of this sample's primary ray
        bool any_hit;
        // per-level records, folded innermost-first when the path ends
        v3 lvA[kMaxLevels];            // PATH: w_k = dot(N,Ra)*kd ; distributed: a_k = falloff*diffuse
        int lvPrim[kMaxLevels];        // distributed: primitive of level k (kr, kd re-read at fold time)
        // distributed: state kept across the shadow ray
        v3 sN, sRr;
        float sDist;
        int sPrim, sLight;
#include "rtgo_start_sample.inc"

#ifdef RTGO_TIMELINE
        if (tl_units++ == 0) tl_first = wall_clock64();
        if (ui == 0) {
            if (strip < p.n_hot) tl_hot += 1;
            else if (tl_cold == 0) tl_cold = wall_clock64();
        }
#endif
        while (__ballot(active) != 0ull) {
#ifdef RTGO_TIMELINE
            if (tl_iters == 1 && tl_c == 0) tl_c = wall_clock64();
            tl_iters += 1;
            const unsigned long long tl_i0 = wall_clock64();
            tl_lanes += (unsigned long long)__popcll(__ballot(active));
#endif
            if (active) {
#include "rtgo_ray_trace.inc"
#include "rtgo_ray_shade.inc"
            }
#ifdef RTGO_TIMELINE
            tl_loop += wall_clock64() + (result.x == 12345.0f ? 1 : 0) - tl_i0;
#endif
        }
        // color += payload, in sample order (kernel.cu:232): every lane of a pixel's group walks the group's results
        const unsigned int cnt = (nn - pass * nn_eff) < nn_eff ? (nn - pass * nn_eff) : nn_eff;
        if (__ballot(any_hit) == 0ull) {
            // every primary ray of the unit missed: all payloads are the background colour, no lane exchange needed
            for (unsigned int q = 0; q < cnt; ++q) color = vadd(color, p.bg);
        } else {
            for (unsigned int q = 0; q < cnt; ++q) {
                const int src = (int)(group_base + q);
                color = vadd(color, mk(__shfl(result.x, src, 64), __shfl(result.y, src, 64), __shfl(result.z, src, 64)));
            }
        }
        }  // pass
        } else {
            // STREAM (frames of more than 16 spp).  The unit's work is a list of TASKS, one per (pixel, sample): task t is sample
            // t / npx of the unit's pixel t % npx.  Any lane whose path has ended takes the next task -- lanes are not tied to a pixel
            // or to a sample slot -- so the samples of the one pixel of the unit that sees the scene spread over all 64 lanes while
            // the pixels that see the background cost one ray each (lock-step: the wave traces the longest path of every pass, sixteen
            // passes at 256 spp, with most lanes idle after the first ray).  New tasks start in batches (>= kRegenBatch idle lanes
            // by __ballot, or nobody active) so that the raygen code runs with lanes to fill it.  A finished path parks its payload
            // in a ring in LDS (slot t % kRing; the slot holds a "pending" pattern from the moment the task is taken); the lanes
            // 0 .. npx-1 own one pixel each and add the payloads of its tasks in task order = sample order (kernel.cu:232) as the
            // completed prefix of the list grows, so the pixel is bit for bit the lock-step one.
            constexpr unsigned int kRing = 64u * (unsigned int)kStreamWindow, kRegenBatch = 4u;
            constexpr unsigned int kPending = 0x7FC0DEADu;   // a NaN no computation produces
            float* s_ring = s_win_base;                      // x at [slot], y at [kRing + slot], z at [2 kRing + slot]
            const unsigned int px_left = p.w - (strip_x0 + ui * P);
            const unsigned int npx = px_left < P ? px_left : P;
            const unsigned int n_tasks = nn * npx;
            unsigned int t_next = 0u, fold_ptr = 0u;   // wave-uniform: tasks taken so far, tasks folded so far
            unsigned int my_slot = 0u;
        bool active;
        int depth;
        int phase;                     // distributed mode: 0 = radiance ray in flight, 1 = shadow ray in flight
        unsigned int seed;
        v3 ro, rd;
        float tmin, tmax;
        v3 result;                     // payload of this sample's primary ray
        bool any_hit;
        // per-level records, folded innermost-first when the path ends
        v3 lvA[kMaxLevels];            // PATH: w_k = dot(N,Ra)*kd ; distributed: a_k = falloff*diffuse
        int lvPrim[kMaxLevels];        // distributed: primitive of level k (kr, kd re-read at fold time)
        // distributed: state kept across the shadow ray
        v3 sN, sRr;
        float sDist;
        int sPrim, sLight;
            active = false;
            depth = 0;
            phase = 0;
            seed = 0u;
            ro = rd = result = sN = sRr = mk(0, 0, 0);
            tmin = tmax = sDist = 0.0f;
            any_hit = false;
            sPrim = sLight = 0;
#pragma unroll
            for (int q = 0; q < kMaxLevels; ++q) {
                lvA[q] = mk(0, 0, 0);
                lvPrim[q] = 0;
            }
#ifdef RTGO_TIMELINE
            unsigned long long ts_regen = 0, ts_trace = 0, ts_shade = 0, ts_fold = 0, ts_lanes_trace = 0, ts_lanes_regen = 0, ts_regens = 0;
#endif
            while (fold_ptr < n_tasks) {
#ifdef RTGO_TIMELINE
                const unsigned long long ts0 = wall_clock64();
                tl_iters += 1;
#endif
                // ---- idle lanes take the next tasks, in lane order
                const unsigned long long m_idle = __builtin_amdgcn_ballot_w64(!active), m_act = ~m_idle;
                const unsigned int room = kRing - (t_next - fold_ptr);
                unsigned int n_take = (unsigned int)__popcll(m_idle);
                n_take = n_take < n_tasks - t_next ? n_take : n_tasks - t_next;
                n_take = n_take < room ? n_take : room;
                if (n_take != 0u && (m_act == 0ull || n_take >= kRegenBatch || t_next + n_take == n_tasks)) {
                    const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m_idle, 0u));
                    // (every lane computes its would-be task: the seed exchange below needs the whole wave)
                    const unsigned int t = t_next + rank;
                    const unsigned int k = npx == 4u ? (t >> 2) : t / npx;   // (the usual unit: four pixels)
                    const unsigned int j = t - k * npx;
                    const unsigned int pix0 = (unsigned int)__shfl((int)strip_seed, (int)((ui * P + j) & 63u), 64);
                    if (!active && rank < n_take) {
                        const float fx = (float)(p.x0 + strip_x0 + ui * P + j);
                        const bool in_range = true;
                        my_slot = t & (kRing - 1u);
                        s_ring[my_slot] = __uint_as_float(kPending);
#include "rtgo_start_sample.inc"
                    }
                    t_next += n_take;
#ifdef RTGO_TIMELINE
                    ts_lanes_regen += n_take;
                    ts_regens += 1;
#endif
                }
#ifdef RTGO_TIMELINE
                const unsigned long long ts1 = wall_clock64() + (seed == 0x12345u ? 1 : 0);
                ts_regen += ts1 - ts0;
                unsigned long long ts3 = ts1;
#endif
                const bool run = active;
                // ---- one ray for every lane that runs
                if (__builtin_amdgcn_ballot_w64(run) != 0ull) {
#ifdef RTGO_TIMELINE
                    ts_lanes_trace += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(run));
#endif
#ifdef RTGO_STREAM_STATS
                    ss_iter += 1;
                    ss_trace += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(active));
                    if (room == 0u && m_idle != 0ull && t_next < n_tasks) ss_stall += 1;
#endif
                    const bool was = run;
                    if (run) {
#include "rtgo_ray_trace.inc"
#ifdef RTGO_STREAM_STATS
                        if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
                            ss_shade_rounds += 1;
                            ss_shade += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(hit));
                        }
#endif
#include "rtgo_ray_shade.inc"
                    }
                    if (was && !active) {
                        s_ring[kRing + my_slot] = result.y;
                        s_ring[2u * kRing + my_slot] = result.z;
                        s_ring[my_slot] = result.x;
                    }
#ifdef RTGO_TIMELINE
                    ts3 = wall_clock64() + (result.x == 12345.0f ? 1 : 0);
                    ts_trace += ts3 - ts1;   // (trace + shading + the ring write of this iteration)
#endif
                }
                // ---- the completed prefix of the task list goes into the pixels
                {
                    const unsigned int probe = fold_ptr + (unsigned int)lane;
                    const bool ready = probe < t_next && __float_as_uint(s_ring[probe & (kRing - 1u)]) != kPending;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(ready);
                    const unsigned int n_ready = m == ~0ull ? 64u : (unsigned int)__builtin_ctzll(~m);
                    const bool must = t_next == n_tasks || (t_next - fold_ptr) + 64u > kRing;   // nothing left to take / the ring is filling up
                    if (n_ready >= 32u || (must && n_ready != 0u)) {
                        if ((unsigned int)lane < npx) {
                            unsigned int t = fold_ptr + (((unsigned int)lane + npx - fold_ptr % npx) % npx);   // this pixel's first task in the prefix
                            for (; t < fold_ptr + n_ready; t += npx) {
                                const unsigned int slot = t & (kRing - 1u);
                                color = vadd(color, mk(s_ring[slot], s_ring[kRing + slot], s_ring[2u * kRing + slot]));
                            }
                        }
                        fold_ptr += n_ready;
                    }
                }
#ifdef RTGO_TIMELINE
                ts_shade += wall_clock64() + (color.x == 12345.0f ? 1 : 0) - ts3;   // (the fold of the completed prefix)
#endif
            }
#ifdef RTGO_TIMELINE
            tls_regen += ts_regen; tls_trace += ts_trace; tls_shade += ts_shade; tls_lanes_trace += ts_lanes_trace; tls_lanes_regen += ts_lanes_regen; tls_regens += ts_regens;
            tl_units += 1;
#endif
            if ((unsigned int)lane < npx) write_pixel(p, (size_t)lr * p.w + strip_x0 + ui * P + (unsigned int)lane, vscale(color, 1.0f / (float)nn));
        }

        if constexpr (!STREAM)
        if (in_range && kl == 0) {
            // kernel.cu:236-246.  float3 / float multiplies by the reciprocal (vec_math.h:479-483)
            write_pixel(p, (size_t)lr * p.w + lx, vscale(color, 1.0f / (float)nn));
        }
#ifdef RTGO_TIMELINE
        if (tl_d == 0) tl_d = wall_clock64();
#endif
        }  // unit
    }

#ifdef RTGO_TIMELINE
    if (lane == 0) {
        unsigned long long* r = p.timeline + 16ull * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
        r[8] = tl_a; r[9] = tl_b; r[10] = tl_c; r[11] = tl_d; r[12] = tl_big; r[13] = tl_tree; r[14] = tl_loop;
        if constexpr (STREAM) {
            r[8] = tls_regen; r[9] = tls_trace; r[10] = tls_shade; r[11] = tls_lanes_trace; r[14] = tls_regens; r[15] = tls_lanes_regen;
        }
        r[0] = tl_t0; r[1] = tl_t1; r[2] = tl_first; r[3] = wall_clock64(); r[4] = tl_units | ((unsigned long long)tl_hot << 32); r[5] = tl_iters; r[6] = tl_lanes;
        r[7] = tl_qwait | ((tl_cold ? tl_cold - tl_t0 : 0ull) << 32);
    }
#endif
    // one atomic per wave per counter
    c_rays = wave_sum(c_rays);
    c_occl = wave_sum(c_occl);
    if (COUNT) {
        c_nodes = wave_sum(c_nodes);
        c_tests = wave_sum(c_tests);
        c_hits = wave_sum(c_hits);
    }
#ifdef RTGO_FAST_COUNTERS
    // diagnostic build only: what the FAST walk itself visits (boxes tested, leaf tests incl. the up-front list)
    if (!STATS) {
        c_nodes = wave_sum(c_nodes);
        c_tests = wave_sum(c_tests);
        if (lane == 0) {
            atomicAdd(&p.counters[5], (unsigned long long)c_nodes);
            atomicAdd(&p.counters[6], (unsigned long long)c_tests);
        }
    }
#endif
#ifdef RTGO_STREAM_STATS
    if (lane == 0 && !STATS) {
        atomicAdd(&p.counters[2], ss_iter);          // -> node_visits
        atomicAdd(&p.counters[3], ss_trace);         // -> prim_tests
        atomicAdd(&p.counters[4], ss_shade_rounds);  // -> hits
        atomicAdd(&p.counters[5], ss_shade);         // -> dbg_fast_boxes
        atomicAdd(&p.counters[6], ss_stall);         // -> dbg_fast_tests
    }
#endif
    if (lane == 0) {
        atomicAdd(&p.counters[0], (unsigned long long)c_rays);
        if (!PATH) atomicAdd(&p.counters[1], (unsigned long long)c_occl);
        if (COUNT && p.count_stats) {
            atomicAdd(&p.counters[2], (unsigned long long)c_nodes);
            atomicAdd(&p.counters[3], (unsigned long long)c_tests);
            atomicAdd(&p.counters[4], (unsigned long long)c_hits);
        }
    }
}

// =====================================================================================================================
// Scene preparation + canonical LBVH build, one workgroup (n <= 512): replaces optixAccelBuild (renderer.cpp:514-611) and
// hoists Matrix4x4::inverse() (Matrix.h:591-635) out of the intersection programs.
// =====================================================================================================================
struct PrimIn {  // = rtgo_prim
    unsigned int type;
    float M[16];
    float kd[3], kr[3], spec, Le[3];
};

__device__ __forceinline__ float det4(const float* m)
{
    // Matrix.h:591-608, term order and product association preserved
    return m[0] * m[5] * m[10] * m[15] - m[0] * m[5] * m[11] * m[14] + m[0] * m[9] * m[14] * m[7] - m[0] * m[9] * m[6] * m[15] +
           m[0] * m[13] * m[6] * m[11] - m[0] * m[13] * m[10] * m[7] - m[4] * m[1] * m[10] * m[15] + m[4] * m[1] * m[11] * m[14] -
           m[4] * m[9] * m[14] * m[3] + m[4] * m[9] * m[2] * m[15] - m[4] * m[13] * m[2] * m[11] + m[4] * m[13] * m[10] * m[3] +
           m[8] * m[1] * m[6] * m[15] - m[8] * m[1] * m[14] * m[7] + m[8] * m[5] * m[14] * m[3] - m[8] * m[5] * m[2] * m[15] +
           m[8] * m[13] * m[2] * m[7] - m[8] * m[13] * m[6] * m[3] - m[12] * m[1] * m[6] * m[11] + m[12] * m[1] * m[10] * m[7] -
           m[12] * m[5] * m[10] * m[3] + m[12] * m[5] * m[2] * m[11] - m[12] * m[9] * m[2] * m[7] + m[12] * m[9] * m[6] * m[3];
}

// one cofactor group of Matrix.h:612-635: a*(b*c - d*e)
#define RTGO_G(a, b, c, d, e) (m[a] * (m[b] * m[c] - m[d] * m[e]))

__device__ __forceinline__ void inverse_rows012(const float* m, float* o)
{
    const float d = 1.0f / det4(m);
    o[0] = d * (RTGO_G(5, 10, 15, 14, 11) + RTGO_G(9, 14, 7, 6, 15) + RTGO_G(13, 6, 11, 10, 7));
    o[4] = d * (RTGO_G(6, 8, 15, 12, 11) + RTGO_G(10, 12, 7, 4, 15) + RTGO_G(14, 4, 11, 8, 7));
    o[8] = d * (RTGO_G(7, 8, 13, 12, 9) + RTGO_G(11, 12, 5, 4, 13) + RTGO_G(15, 4, 9, 8, 5));
    o[1] = d * (RTGO_G(9, 2, 15, 14, 3) + RTGO_G(13, 10, 3, 2, 11) + RTGO_G(1, 14, 11, 10, 15));
    o[5] = d * (RTGO_G(10, 0, 15, 12, 3) + RTGO_G(14, 8, 3, 0, 11) + RTGO_G(2, 12, 11, 8, 15));
    o[9] = d * (RTGO_G(11, 0, 13, 12, 1) + RTGO_G(15, 8, 1, 0, 9) + RTGO_G(3, 12, 9, 8, 13));
    o[2] = d * (RTGO_G(13, 2, 7, 6, 3) + RTGO_G(1, 6, 15, 14, 7) + RTGO_G(5, 14, 3, 2, 15));
    o[6] = d * (RTGO_G(14, 0, 7, 4, 3) + RTGO_G(2, 4, 15, 12, 7) + RTGO_G(6, 12, 3, 0, 15));
    o[10] = d * (RTGO_G(15, 0, 5, 4, 1) + RTGO_G(3, 4, 13, 12, 5) + RTGO_G(7, 12, 1, 0, 13));
    o[3] = d * (RTGO_G(1, 10, 7, 6, 11) + RTGO_G(5, 2, 11, 10, 3) + RTGO_G(9, 6, 3, 2, 7));
    o[7] = d * (RTGO_G(2, 8, 7, 4, 11) + RTGO_G(6, 0, 11, 8, 3) + RTGO_G(10, 4, 3, 0, 7));
    o[11] = d * (RTGO_G(3, 8, 5, 4, 9) + RTGO_G(7, 0, 9, 8, 1) + RTGO_G(11, 4, 1, 0, 5));
}
#undef RTGO_G

// Primitive::GetAabb / CubeBox::TransformAndAlign (primitive.cpp:35-79, 100-115): the 8 corners of [-1,1]^3 through the
// 4-term matrix product (sum seeded with 0.0f, Matrix.h:344-360), min/max seeded with +-50, +-1e-3 pad.
__device__ __forceinline__ void cube_aabb(const float* M, float* bb)
{
    float mn[3] = {50.0f, 50.0f, 50.0f}, mx[3] = {-50.0f, -50.0f, -50.0f};
    // corner order of CubeBox::face0/face1 columns: x = {-1,-1,1,1}, z = {-1,1,-1,1}, y = -1 (face0) / +1 (face1)
    const float cxs[4] = {-1.f, -1.f, 1.f, 1.f}, czs[4] = {-1.f, 1.f, -1.f, 1.f};
    for (int i = 0; i < 4; ++i)
        for (int a = 0; a < 3; ++a) {
            const float* r = M + 4 * a;
            float p0 = 0.0f, p1 = 0.0f;
            p0 += r[0] * cxs[i];
            p0 += r[1] * -1.f;
            p0 += r[2] * czs[i];
            p0 += r[3] * 1.f;
            p1 += r[0] * cxs[i];
            p1 += r[1] * 1.f;
            p1 += r[2] * czs[i];
            p1 += r[3] * 1.f;
            float t = (p0 < mn[a]) ? p0 : mn[a];
            mn[a] = (p1 < t) ? p1 : t;
            t = (mx[a] < p0) ? p0 : mx[a];
            mx[a] = (t < p1) ? p1 : t;
        }
    for (int a = 0; a < 3; ++a) {
        bb[a] = mn[a] - 0.001f;
        bb[3 + a] = mx[a] + 0.001f;
    }
}

__device__ __forceinline__ unsigned int expand_bits(unsigned int v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ int lbvh_delta(const unsigned long long* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned int a = (unsigned int)(keys[i] >> 32), b = (unsigned int)(keys[j] >> 32);
    if (a == b) return 32 + __clz((unsigned int)i ^ (unsigned int)j);
    return __clz(a ^ b);
}

// Outputs.  out_nodes: (2n-1) x 2 float4 canonical LBVH; out_prims: n x 6 float4 (SBT order); aabb_io: n x 6 floats (read when
// have_aabb, else written); out_fnodes / out_fprims: the fast walk's tree (2*n_small-1 nodes) and Morton-ordered records
// (small primitives first, then the "big" ones that are tested up front);
// out_tight: n x 6 floats, the fast walk's box of every primitive (SBT order);
// out_meta = {canonical depth, fast-walk stack depth, n_small, tight scene bounds (6 floats as bits), pairs in the up-front list
// | cuboid certificate of the list << 8, nodes of the fast walk's tree, the two coefficients of cuboid_range's margin (float bits)}.
__global__ __launch_bounds__(kMaxPrims) void build_kernel(const PrimIn* __restrict__ prims, float* __restrict__ aabb_io,
                                                          int have_aabb, int n, float4* __restrict__ out_nodes,
                                                          float4* __restrict__ out_prims, float4* __restrict__ out_fnodes,
                                                          float4* __restrict__ out_fprims, int leaf_budget, float big_frac, int* __restrict__ out_meta,
                                                          float* __restrict__ out_tight, int cuboids, float4* __restrict__ out_frames)
{
    __shared__ float s_box[kMaxPrims][6];               // per primitive: reference AABB, later the tight box
    __shared__ unsigned long long s_keys[kMaxPrims];
    __shared__ float s_nbox[2 * kMaxPrims][6];
    __shared__ int s_left[kMaxPrims], s_right[kMaxPrims];
    __shared__ int s_parent[2 * kMaxPrims];
    __shared__ int s_visit[kMaxPrims];
    __shared__ int s_wt[2 * kMaxPrims];                 // fast walk: cost weight of each subtree
    __shared__ short s_lo[kMaxPrims], s_hi[kMaxPrims];  // Morton range covered by each internal node
    __shared__ unsigned char s_flag[kMaxPrims];         // fast walk: 1 = "big" primitive kept out of the tree
    __shared__ unsigned short s_order[kMaxPrims];       // fast walk: primitive at each record position (pairs side by side)
    __shared__ unsigned char s_used[kMaxPrims];
    __shared__ int s_depth, s_count, s_tmask;   // s_tmask: primitive types present in the fast walk's tree (bit = type)
    // the bounds reductions run while s_nbox is not in use: borrow its storage (keeps static LDS under 64 KiB)
    float(*s_red)[kMaxPrims] = reinterpret_cast<float(*)[kMaxPrims]>(&s_nbox[0][0]);

    const int i = threadIdx.x;

    // min/max over the boxes of the primitives selected by `take` -> s_red[0..5][0] (exact, order-independent)
    auto reduce_bounds = [&](bool take) {
        for (int a = 0; a < 3; ++a) {
            s_red[a][i] = take ? s_box[i][a] : INFINITY;
            s_red[3 + a][i] = take ? s_box[i][3 + a] : -INFINITY;
        }
        __syncthreads();
        for (int stride = kMaxPrims / 2; stride > 0; stride >>= 1) {
            if (i < stride)
                for (int a = 0; a < 3; ++a) {
                    s_red[a][i] = fminf(s_red[a][i], s_red[a][i + stride]);
                    s_red[3 + a][i] = fmaxf(s_red[3 + a][i], s_red[3 + a][i + stride]);
                }
            __syncthreads();
        }
    };
    // 30-bit Morton code of primitive i's box centre normalised to the bounds in s_red[..][0]
    auto morton_of = [&](int first = -1, int count = 1, bool cubic = false) -> unsigned int {   // centre of the union of the boxes [first, first + count)
        if (first < 0) first = i;
        unsigned int q[3];
        // cubic: one scale for the three axes (the longest extent), so that a Morton cell is a cube and not a slab
        const float emax = fmaxf(fmaxf(s_red[3][0] - s_red[0][0], s_red[4][0] - s_red[1][0]), s_red[5][0] - s_red[2][0]);
        for (int a = 0; a < 3; ++a) {
            float lo = s_box[first][a], hi = s_box[first][3 + a];
            for (int k = 1; k < count; ++k) {
                lo = fminf(lo, s_box[first + k][a]);
                hi = fmaxf(hi, s_box[first + k][3 + a]);
            }
            const float c = (lo + hi) * 0.5f;
            const float ext = cubic ? emax : s_red[3 + a][0] - s_red[a][0];
            const float u = ext > 0.0f ? (c - s_red[a][0]) / ext : 0.0f;
            q[a] = (unsigned int)fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        }
        return (expand_bits(q[0]) << 2) | (expand_bits(q[1]) << 1) | expand_bits(q[2]);
    };
    // bitonic sort of the kMaxPrims keys in LDS; keys are unique, so the result is THE (code, index) order
    auto sort_keys = [&]() {
        __syncthreads();
        for (int k = 2; k <= kMaxPrims; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s_keys[i] = b;
                        s_keys[ixj] = a;
                    }
                }
                __syncthreads();
            }
    };
    // Karras 2012 over the first m sorted keys + bottom-up fit of boxes (and cost weights).  Leaves are nodes [m-1, 2m-2].
    auto build_tree = [&](int m, bool with_weights) {
        const int leaf0 = m - 1;
        if (i < m) {
            const int prim = (int)(s_keys[i] & 0xFFFFFFFFu);
            for (int a = 0; a < 6; ++a) s_nbox[leaf0 + i][a] = s_box[prim][a];
            s_visit[i] = 0;
        }
        if (i < 2 * m - 1) s_parent[i] = -1;
        if (i + kMaxPrims < 2 * m - 1) s_parent[i + kMaxPrims] = -1;
        __syncthreads();
        if (i < m - 1) {
            const int d = (lbvh_delta(s_keys, m, i, i + 1) - lbvh_delta(s_keys, m, i, i - 1)) >= 0 ? 1 : -1;
            const int dmin = lbvh_delta(s_keys, m, i, i - d);
            int lmax = 2;
            while (lbvh_delta(s_keys, m, i, i + lmax * d) > dmin) lmax *= 2;
            int l = 0;
            for (int t = lmax / 2; t >= 1; t /= 2)
                if (lbvh_delta(s_keys, m, i, i + (l + t) * d) > dmin) l += t;
            const int j = i + l * d;
            const int dnode = lbvh_delta(s_keys, m, i, j);
            int s = 0, t = l;
            do {
                t = (t + 1) / 2;
                if (lbvh_delta(s_keys, m, i, i + (s + t) * d) > dnode) s += t;
            } while (t > 1);
            const int gamma = i + s * d + (d < 0 ? -1 : 0);
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int left = (lo == gamma) ? leaf0 + gamma : gamma;
            const int right = (hi == gamma + 1) ? leaf0 + gamma + 1 : gamma + 1;
            s_left[i] = left;
            s_right[i] = right;
            s_lo[i] = (short)lo;
            s_hi[i] = (short)hi;
            s_parent[left] = i;
            s_parent[right] = i;
        }
        __syncthreads();
        // the second arrival at a node (LDS atomic) owns it
        if (i < m && m > 1) {
            int pnode = s_parent[leaf0 + i];
            while (pnode >= 0) {
                __threadfence_block();
                if (atomicAdd(&s_visit[pnode], 1) == 0) break;
                __threadfence_block();
                const int L = s_left[pnode], R = s_right[pnode];
                for (int a = 0; a < 3; ++a) {
                    s_nbox[pnode][a] = fminf(s_nbox[L][a], s_nbox[R][a]);
                    s_nbox[pnode][3 + a] = fmaxf(s_nbox[L][3 + a], s_nbox[R][3 + a]);
                }
                if (with_weights) s_wt[pnode] = s_wt[L] + s_wt[R];
                pnode = s_parent[pnode];
            }
        }
        __syncthreads();
    };

    if (i == 0) {
        s_depth = 0;
        s_count = 0;
        s_tmask = 0;
    }
    // ---- per primitive: inverse, record, reference AABB
    PrimIn P;
    if (i < n) {
        P = prims[i];
        float inv[12];
        inverse_rows012(P.M, inv);
        out_prims[6 * i + 0] = make_float4(inv[0], inv[1], inv[2], inv[3]);
        out_prims[6 * i + 1] = make_float4(inv[4], inv[5], inv[6], inv[7]);
        out_prims[6 * i + 2] = make_float4(inv[8], inv[9], inv[10], inv[11]);
        out_prims[6 * i + 3] = make_float4(P.kd[0], P.kd[1], P.kd[2], P.spec);
        out_prims[6 * i + 4] = make_float4(P.kr[0], P.kr[1], P.kr[2], __int_as_float((int)P.type));
        out_prims[6 * i + 5] = make_float4(P.Le[0], P.Le[1], P.Le[2], 0.0f);
        // Shading frame of a FLAT primitive (rectangle, disk: object-space normal (0,1,0), kernel.cu:345,388): what the closest-hit
        // program computes from it on every hit -- N = normalize(TransformNormal(0,1,0)) (kernel.cu:428) and the tangent of
        // GetRayOnHemisphere for direction N, X = normalize(N.y - N.z, -N.x, N.x) (kernel.cu:105) -- depends on the primitive alone.
        // Computed here ONCE with the same device functions on the same values, so the bits are those of the per-hit computation;
        // a flipped normal flips both exactly (the expressions are odd in N, negation is exact), and Z = N x X is unchanged.
        {
            const bool flat = P.type == 1u || P.type == 2u;
            v3 fn = mk(0.0f, 0.0f, 0.0f), fx = mk(0.0f, 0.0f, 0.0f);
            if (flat) {
                fn = vnormalize(xf_normal(make_float4(inv[0], inv[1], inv[2], inv[3]), make_float4(inv[4], inv[5], inv[6], inv[7]), make_float4(inv[8], inv[9], inv[10], inv[11]), mk(0.0f, 1.0f, 0.0f)));
                fx = vnormalize(mk(fn.y - fn.z, -fn.x, fn.x));
            }
            out_frames[2 * i + 0] = make_float4(fn.x, fn.y, fn.z, flat ? 1.0f : 0.0f);
            out_frames[2 * i + 1] = make_float4(fx.x, fx.y, fx.z, 0.0f);
        }
        float bb[6];
        if (have_aabb) {
            for (int a = 0; a < 6; ++a) bb[a] = aabb_io[6 * i + a];
        } else {
            cube_aabb(P.M, bb);
            for (int a = 0; a < 6; ++a) aabb_io[6 * i + a] = bb[a];
        }
        for (int a = 0; a < 6; ++a) s_box[i][a] = bb[a];
    }
    __syncthreads();

    // ================= canonical LBVH: every primitive, the reference's AABBs =================
    reduce_bounds(i < n);
    s_keys[i] = (i < n) ? (((unsigned long long)morton_of() << 32) | (unsigned int)i) : ~0ull;
    sort_keys();
    build_tree(n, false);
    {
        const int leaf0 = n - 1;
        if (i < n) {
            int dep = 0;
            int q = s_parent[leaf0 + i];
            while (q >= 0) {
                ++dep;
                q = s_parent[q];
            }
            atomicMax(&s_depth, dep);
        }
        for (int k = i; k < 2 * n - 1; k += kMaxPrims) {
            int left, right;
            if (k >= leaf0) {
                left = (int)(s_keys[k - leaf0] & 0xFFFFFFFFu);
                right = -1;
            } else {
                left = s_left[k];
                right = s_right[k];
            }
            out_nodes[2 * k + 0] = make_float4(s_nbox[k][0], s_nbox[k][1], s_nbox[k][2], __int_as_float(left));
            out_nodes[2 * k + 1] = make_float4(s_nbox[k][3], s_nbox[k][4], s_nbox[k][5], __int_as_float(right));
        }
    }
    __syncthreads();
    if (i == 0) out_meta[0] = s_depth;

    // ================= fast walk's structure =================
    // Any conservative structure returns the same closest hit, so this one is built for speed:
    //  * TIGHT per-shape boxes for rectangles and disks (the reference's CubeBox boxes span a whole cube around a flat shape);
    //  * "big" primitives (box spanning >= 36 % of the scene on two axes: room walls, floors) stay out of the tree and are
    //    tested first, which also gives every ray an early closest-hit bound for culling the tree;
    //  * LBVH over the rest, subtrees collapsed into multi-primitive leaves by a cost budget.
    // Spheres and cylinders keep the box the canonical walk uses (the caller's / the CubeBox one): their quadratic loses its
    // digits with distance (b*b - 4ac at |o| ~ 2000 radii is good to ~0.1 radius), so from far away the intersection program
    // reports hits up to tenths of a unit OFF the surface -- inside the reference's loose box, outside a tight one -- and the
    // closest hit must be the reference's arithmetic, not the geometry (tools/fuzz_cameras.py found it: a camera 1200 units
    // from the slide scene).  Rectangles and disks divide once (error ~1e-7 of the distance): their tight boxes stand.
    if (i < n && (P.type == 2 || P.type == 1)) {
        const float* M = P.M;
        for (int a = 0; a < 3; ++a) {
            const float mx = M[4 * a + 0], mz = M[4 * a + 2], c = M[4 * a + 3];
            float e;  // half extent of the unit shape's image along world axis a
            if (P.type == 2) e = 0.5f * fabsf(mx) + 0.5f * fabsf(mz);   // rectangle |x|,|z| <= 1/2, y = 0
            else e = sqrtf(mx * mx + mz * mz);                          // disk, radius 1 in y = 0
            e = e * 1.00001f + 0.001f;  // rounding headroom + the reference's own pad (AABB_EPSILON)
            s_box[i][a] = c - e;
            s_box[i][3 + a] = c + e;
        }
    }
    __syncthreads();
    if (i < n)   // per primitive: the box the fast walk culls with (the host projects these onto the screen: LaunchParams::hot_mask)
        for (int a = 0; a < 6; ++a) out_tight[6 * i + a] = s_box[i][a];
    reduce_bounds(i < n);
    if (i < 6) out_meta[3 + i] = __float_as_int(s_red[i][0]);  // tight scene bounds: min xyz, max xyz
    bool big = false;
    if (i < n) {
        int wide = 0;
        for (int a = 0; a < 3; ++a)
            if (s_box[i][3 + a] - s_box[i][a] >= big_frac * (s_red[3 + a][0] - s_red[a][0])) ++wide;
        big = wide >= 2;
        s_flag[i] = big ? 1 : 0;
        if (!big) {
            atomicAdd(&s_count, 1);
            atomicOr(&s_tmask, 1 << (int)(P.type & 3u));
        }
    }
    __syncthreads();
    const int n_small = s_count;
    reduce_bounds(i < n && !big);
    // Boxes: six consecutive small rectangles that pair up by opposite normals (ShapeFactory::CreateCube emits a cube's faces
    // consecutively, shapefactory.cpp) share ONE Morton code, that of the box centre, so that the Karras hierarchy keeps them
    // in one subtree and the cost budget (6) turns exactly that subtree into a leaf: whole-box leaves, which pair_test halves.
    // Left to the face centroids, Morton order cuts across touching boxes (checkered: leaves of every mix of pairs and singles).
    if (i < n) s_used[i] = 0xFF;   // first primitive of the box this one belongs to, relative: 0..5, or 0xFF
    __syncthreads();
    if (i == 0) {
        int a = 0;
        while (a + 6 <= n) {
            bool ok = true;
            for (int k = 0; k < 6 && ok; ++k) ok = prims[a + k].type == 2u && !s_flag[a + k];
            if (ok) {
                unsigned int paired = 0;
                for (int k = 0; k < 6; ++k) {
                    if (paired & (1u << k)) continue;
                    const float4 ra = out_prims[6 * (a + k) + 1];
                    const float la = sqrtf(ra.x * ra.x + ra.y * ra.y + ra.z * ra.z);
                    for (int m = k + 1; m < 6; ++m) {
                        if (paired & (1u << m)) continue;
                        const float4 rb = out_prims[6 * (a + m) + 1];
                        const float lb = sqrtf(rb.x * rb.x + rb.y * rb.y + rb.z * rb.z);
                        if (ra.x * rb.x + ra.y * rb.y + ra.z * rb.z < -0.9999f * la * lb) {
                            paired |= (1u << k) | (1u << m);
                            break;
                        }
                    }
                }
                ok = paired == 0x3Fu;
            }
            if (ok) {
                for (int k = 0; k < 6; ++k) s_used[a + k] = (unsigned char)k;
                a += 6;
            } else {
                a += 1;
            }
        }
    }
    __syncthreads();
    // small primitives sort by Morton code; big ones after them, in SBT order
    {
        const bool boxed = i < n && s_used[i] != 0xFF;
        const bool cubic = true;   // (balls -1 %, plateau -1 %, slide -2.5 % against per-axis scaling; nothing lost elsewhere)
        const unsigned int code = (i < n && !big) ? (boxed ? morton_of(i - (int)s_used[i], 6, cubic) : morton_of(-1, 1, cubic)) : 0u;
        s_keys[i] = (i < n) ? ((big ? (0xFFFFFFFEull << 32) : ((unsigned long long)code << 32)) | (unsigned int)i) : ~0ull;
    }
    sort_keys();
    if (i < n_small) {
        const int prim = (int)(s_keys[i] & 0xFFFFFFFFu);
        const unsigned int type = prims[prim].type;
        // relative cost of one leaf test vs one box test: rectangles reject on two signs, quadrics need the full transform
        s_wt[n_small - 1 + i] = (type == 2) ? 1 : (type == 1 ? 4 : 32);
    }
    if (i == 0) s_depth = 0;
    __syncthreads();
    if (n_small > 0) build_tree(n_small, true);
    {
        const int leaf0 = n_small - 1;
        if (i < n_small) {
            // only ancestors that stay internal (cost above the budget) can push on the fast walk's stack
            int fdep = 0;
            int q = s_parent[leaf0 + i];
            while (q >= 0) {
                if (s_wt[q] > leaf_budget) ++fdep;
                q = s_parent[q];
            }
            atomicMax(&s_depth, fdep);
        }
        // Record order inside every group the walk scans linearly -- the up-front list and each maximal collapsed leaf:
        // rectangles with opposite normals side by side (pair_test), pairs first, the rest after them.  One thread per group.
        // Leaves are only paired when EVERY multi-record leaf of the scene pairs up completely (whole boxes): the lanes of a wave
        // scan different leaves side by side, and with leaves of both kinds they take turns in the pair loop and the single
        // loop (checkered, whose Morton leaves cut across its 64 cubes: +9 %).  The up-front list is scanned by all lanes
        // together and is always paired.
        if (i < n) {
            s_order[i] = (unsigned short)(s_keys[i] & 0xFFFFFFFFu);
            s_used[i] = 0;
            s_visit[i] = 0;   // (build_tree's arrival counters are no longer needed: pairs per node from here on)
        }
        __shared__ int s_cubA, s_cubB, s_cubN;   // cuboid_range's margin coefficients (positive floats as bits: integer max = float max)
        if (i == 0) {
            out_meta[9] = 0;
            s_count = 0;   // leaves that do not pair up completely
            s_cubA = 0;
            s_cubB = 0;
            s_cubN = 0;   // leaves certified as cuboids
        }
        __syncthreads();
        int g_lo = 0, g_hi = -1;
        {
            const bool list = (i == kMaxPrims - 1);
            if (list) {
                g_lo = n_small;
                g_hi = n - 1;
            } else if (i < leaf0 && s_wt[i] <= leaf_budget && (s_parent[i] < 0 || s_wt[s_parent[i]] > leaf_budget)) {
                g_lo = s_lo[i];
                g_hi = s_hi[i];
            }
            if (g_hi > g_lo) {
                auto prim_at = [&](int pos) { return (int)(s_keys[pos] & 0xFFFFFFFFu); };
                auto is_rect = [&](int pos) { return prims[prim_at(pos)].type == 2u; };
                auto normal_of = [&](int pos) {   // world normal of a rectangle = row 1 of M^-1 (TransformNormal of (0,1,0))
                    const float4 r = out_prims[6 * prim_at(pos) + 1];
                    const float l = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z);
                    return l > 0.0f ? mk(r.x / l, r.y / l, r.z / l) : mk(0.0f, 0.0f, 0.0f);
                };
                int out = g_lo;
                for (int a = g_lo; a <= g_hi; ++a) {
                    if (s_used[a] || !is_rect(a)) continue;
                    const v3 na = normal_of(a);
                    int bsel = -1;
                    float bdot = -0.9999f;
                    for (int b = a + 1; b <= g_hi; ++b) {
                        if (s_used[b] || !is_rect(b)) continue;
                        const float dt = vdot(na, normal_of(b));
                        if (dt < bdot) {
                            bdot = dt;
                            bsel = b;
                        }
                    }
                    if (bsel >= 0) {
                        s_used[a] = 1;
                        s_used[bsel] = 1;
                        s_order[out] = (unsigned short)prim_at(a);
                        s_order[out + 1] = (unsigned short)prim_at(bsel);
                        out += 2;
                    }
                }
                const int npairs = (out - g_lo) / 2;
                // Cuboid certificate (cuboid_range): three pairs -- a whole leaf, or the pairs of the up-front list -- are the faces
                // of one box seen from outside when the four corners of every face f lie at or below the plane of every face g of
                // the other two pairs (y_g <= tol in g's object space; y_g is affine, so the whole face does), of one room seen from
                // inside when they lie at or above it.  Checked on the matrices themselves: whatever passes is safe, whatever the
                // shapes were meant to be.  L = how far y_g varies over face f: it carries the rounding of the reference's (u, v) on
                // f into y_g units; A, B: margin = tol + K (A R + B) for rays within R of the origin (rtgo_capi.hip).
                int cert = 0;
                if (cuboids && npairs == 3 && (list || g_hi - g_lo + 1 == 6)) {
                    bool outw = true, inw = true;
                    float A = 0.0f, B = 0.0f;
                    auto n1 = [](const float4 r) { return fabsf(r.x) + fabsf(r.y) + fabsf(r.z); };
                    for (int f = 0; f < 6; ++f) {
                        const int pf = (int)s_order[g_lo + f];
                        const float* M = prims[pf].M;
                        const float4 f0 = out_prims[6 * pf + 0], f2 = out_prims[6 * pf + 2];
                        const float n1f = fmaxf(n1(f0), n1(f2)), wf = fmaxf(fabsf(f0.w), fabsf(f2.w));
                        for (int g = 0; g < 6; ++g) {
                            if ((g >> 1) == (f >> 1)) continue;
                            const float4 r1 = out_prims[6 * (int)s_order[g_lo + g] + 1];
                            float ymax = -INFINITY, ymin = INFINITY;
                            for (int c = 0; c < 4; ++c) {
                                const float sx = (c & 1) ? 0.5f : -0.5f, sz = (c & 2) ? 0.5f : -0.5f;
                                const float cx = M[0] * sx + M[2] * sz + M[3], cy = M[4] * sx + M[6] * sz + M[7], cz = M[8] * sx + M[10] * sz + M[11];
                                const float y = r1.x * cx + r1.y * cy + r1.z * cz + r1.w;
                                ymax = fmaxf(ymax, y);
                                ymin = fminf(ymin, y);
                                if (!(y == y)) outw = inw = false;
                            }
                            outw = outw && ymax <= kCuboidTol;
                            inw = inw && ymin >= -kCuboidTol;
                            const float L = ymax - ymin;
                            A = fmaxf(A, L * n1f + n1(r1));
                            B = fmaxf(B, L * wf + fabsf(r1.w));
                        }
                    }
                    cert = outw ? 1 : ((inw && list) ? 2 : 0);   // (rooms are big: only the list can hold one)
                    if (!(A < 1e30f && B < 1e30f)) cert = 0;
                    if (cert) {
                        atomicMax(&s_cubA, __float_as_int(A));
                        atomicMax(&s_cubB, __float_as_int(B));
                    }
                }
                if (list || 2 * npairs == g_hi - g_lo + 1) {
                    for (int a = g_lo; a <= g_hi; ++a)
                        if (!s_used[a]) s_order[out++] = (unsigned short)prim_at(a);
                    if (list) out_meta[9] = npairs | (cert << 8);
                    else s_visit[i] = npairs | (cert << 8);
                } else {
                    atomicAdd(&s_count, 1);
                }
            }
        }
        __syncthreads();
        if (s_count > 0 && i != kMaxPrims - 1 && g_hi > g_lo) {   // mixed scene: leave every leaf as it was
            for (int a = g_lo; a <= g_hi; ++a) s_order[a] = (unsigned short)(s_keys[a] & 0xFFFFFFFFu);
            s_visit[i] = 0;
        }
        if (i == 0) {
            out_meta[11] = s_cubA;
            out_meta[12] = s_cubB;
        }
        __syncthreads();
        if (i < n) {
            // traversal record (inverse rows were written to out_prims by the primitive's own thread above)
            const int prim = (int)s_order[i];
            out_fprims[4 * i + 0] = out_prims[6 * prim + 0];
            out_fprims[4 * i + 1] = out_prims[6 * prim + 1];
            out_fprims[4 * i + 2] = out_prims[6 * prim + 2];
            out_fprims[4 * i + 3] = make_float4(__int_as_float((int)prims[prim].type), __int_as_float(prim), 0.0f, 0.0f);
        }
        // ---- the tree the walk uses: a top-down surface-area-heuristic build over the walk's UNITS.  A unit is a maximal collapsed
        // subtree of the Morton hierarchy above (cost <= budget: one multi-record leaf whose records are contiguous) or a single
        // primitive; the hierarchy only serves to form them.  Above the units the Morton prefixes are a poor guide for rays (they
        // know nothing of box areas), and any tree over the same leaves returns the same closest hit, so the topology is rebuilt:
        // every node is split where A(left) * W(left) + A(right) * W(right) is smallest over the three axes and every position of
        // the units sorted by centroid (W = leaf-test cost weights).  All 512 threads walk one task queue together: a rank sort per
        // axis in parallel, the sweep by one thread.  (Rotations of the Morton tree gave balls -4.5 %; this build ... see DESIGN.)
        short* s_unit = reinterpret_cast<short*>(s_keys);    // (the sort keys are no longer needed) [kMaxPrims] binary node of unit u
        short* s_perm = s_unit + kMaxPrims;                  // [kMaxPrims] the units in the current task order
        short* s_tmp = s_perm + kMaxPrims;                   // [kMaxPrims]
        float* s_sfx = &s_box[0][0];                         // (the primitive boxes are no longer needed) [kMaxPrims][7]: suffix box + weight
        short* s_tq_node = reinterpret_cast<short*>(s_parent);   // task queue, <= 2 * units - 1 entries: node, lo, hi, depth
        short* s_tq_lo = s_tq_node + 2 * kMaxPrims;
        short* s_tq_hi = reinterpret_cast<short*>(s_wt);         // (s_wt is read until the units are formed; see the barrier below)
        short* s_tq_dep = s_tq_hi + 2 * kMaxPrims;
        __shared__ int s_qtail, s_best_axis, s_best_pos, s_units;
        // the tree under construction (dynamic LDS, 2 * kMaxPrims nodes): box, links, parent -- rotated below, then written out
        extern __shared__ __attribute__((aligned(16))) unsigned char build_dyn[];
        float (*t_box)[6] = reinterpret_cast<float (*)[6]>(build_dyn);
        int* t_left = reinterpret_cast<int*>(t_box + 2 * kMaxPrims);
        int* t_right = t_left + 2 * kMaxPrims;
        int* t_parent = t_right + 2 * kMaxPrims;
        auto leafish = [&](int k) { return k >= leaf0 || s_wt[k] <= leaf_budget; };
        __syncthreads();
        if (i < n_small) s_left[i] = -1;   // (the binary links are no longer needed) unit that starts at Morton position i
        __syncthreads();
        for (int k = i; k < 2 * n_small - 1; k += kMaxPrims)
            if (leafish(k) && (s_parent[k] < 0 || s_wt[s_parent[k]] > leaf_budget)) s_left[k >= leaf0 ? k - leaf0 : s_lo[k]] = k;
        __syncthreads();
        if (i == 0) {
            int L = 0;
            for (int pos = 0; pos < n_small; ++pos)
                if (s_left[pos] >= 0) s_unit[L++] = (short)s_left[pos];
            s_units = L;
        }
        __syncthreads();
        const int L = s_units;
        // per unit: weight (kept in s_right, an int array that is free now) -- after this barrier s_wt and s_parent are reused
        if (i < L) {
            const int k = s_unit[i];
            s_right[i] = s_wt[k] < 1 ? 1 : s_wt[k];
            s_perm[i] = (short)i;
        }
        __syncthreads();
        if (i == 0) {
            s_tq_node[0] = 0;
            s_tq_lo[0] = 0;
            s_tq_hi[0] = (short)L;
            s_tq_dep[0] = 0;
            s_qtail = L > 0 ? 1 : 0;
            s_count = L > 0 ? 1 : 0;   // nodes allocated
            s_depth = 0;
            t_parent[0] = -1;
        }
        __syncthreads();
        auto ubox = [&](int u, int c) { return s_nbox[s_unit[u]][c]; };
        for (int qi = 0; qi < 2 * kMaxPrims; ++qi) {
            __syncthreads();
            if (qi >= s_qtail) break;   // (uniform: every thread reads the same word after the barrier)
            const int lo = s_tq_lo[qi], hi = s_tq_hi[qi], node = s_tq_node[qi], dep = s_tq_dep[qi], m = hi - lo;
            if (m == 1) {
                if (i == 0) {
                    const int k = s_unit[s_perm[lo]];
                    const int first = k >= leaf0 ? k - leaf0 : (int)s_lo[k];
                    const int cnt = k >= leaf0 ? 1 : (int)s_hi[k] - (int)s_lo[k] + 1;
                    const int npairs = k >= leaf0 ? 0 : s_visit[k];
                    for (int c = 0; c < 6; ++c) t_box[node][c] = s_nbox[k][c];
                    t_left[node] = first;
                    t_right[node] = -(cnt | (npairs << 12));
                }
                continue;
            }
            if (i == 0) {
                s_best_axis = -1;
                s_best_pos = m / 2;
            }
            float best_cost = INFINITY;   // (thread 0's)
            for (int pass = 0; pass < 4; ++pass) {
                // passes 0..2: try axis `pass`; pass 3: put the range back in the order of the best axis
                __syncthreads();
                const int axis = pass < 3 ? pass : s_best_axis;
                if (pass == 3 && (axis < 0 || axis == 2)) break;   // (uniform) no finite cost at all, or already in z order
                if (i < m) {
                    const int me = s_perm[lo + i];
                    const float key = ubox(me, axis) + ubox(me, 3 + axis);
                    int rank = 0;
                    for (int j = 0; j < m; ++j) {
                        const int other = s_perm[lo + j];
                        const float kj = ubox(other, axis) + ubox(other, 3 + axis);
                        rank += (kj < key || (kj == key && other < me)) ? 1 : 0;
                    }
                    s_tmp[lo + rank] = (short)me;
                }
                __syncthreads();
                if (i < m) s_perm[lo + i] = s_tmp[lo + i];
                __syncthreads();
                if (pass < 3 && i == 0) {
                    // suffix boxes and weights from the right, then the sweep from the left
                    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    int w = 0;
                    for (int j = m - 1; j >= 1; --j) {
                        const int u = s_perm[lo + j];
                        for (int c = 0; c < 3; ++c) {
                            b[c] = fminf(b[c], ubox(u, c));
                            b[3 + c] = fmaxf(b[3 + c], ubox(u, 3 + c));
                        }
                        w += s_right[u];
                        const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
                        s_sfx[2 * j + 0] = ex * ey + ey * ez + ez * ex;
                        s_sfx[2 * j + 1] = (float)w;
                    }
                    float a[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    int wl = 0;
                    for (int j = 1; j < m; ++j) {   // left = [0, j), right = [j, m)
                        const int u = s_perm[lo + j - 1];
                        for (int c = 0; c < 3; ++c) {
                            a[c] = fminf(a[c], ubox(u, c));
                            a[3 + c] = fmaxf(a[3 + c], ubox(u, 3 + c));
                        }
                        wl += s_right[u];
                        const float ex = a[3] - a[0], ey = a[4] - a[1], ez = a[5] - a[2];
                        const float cost = (ex * ey + ey * ez + ez * ex) * (float)wl + s_sfx[2 * j] * s_sfx[2 * j + 1];
                        if (cost < best_cost) {
                            best_cost = cost;
                            s_best_axis = pass;
                            s_best_pos = j;
                        }
                    }
                }
            }
            __syncthreads();
            if (i == 0) {
                // this node: box of its range, two children appended to the queue
                float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                for (int j = lo; j < hi; ++j) {
                    const int u = s_perm[j];
                    for (int c = 0; c < 3; ++c) {
                        b[c] = fminf(b[c], ubox(u, c));
                        b[3 + c] = fmaxf(b[3 + c], ubox(u, 3 + c));
                    }
                }
                const int cl = s_count, cr = s_count + 1;
                s_count += 2;
                for (int c = 0; c < 6; ++c) t_box[node][c] = b[c];
                t_left[node] = cl;
                t_right[node] = cr;
                t_parent[cl] = node;
                t_parent[cr] = node;
                const int mid = lo + s_best_pos;
                const int t = s_qtail;
                s_tq_node[t] = (short)cl; s_tq_lo[t] = (short)lo;  s_tq_hi[t] = (short)mid; s_tq_dep[t] = (short)(dep + 1);
                s_tq_node[t + 1] = (short)cr; s_tq_lo[t + 1] = (short)mid; s_tq_hi[t + 1] = (short)hi; s_tq_dep[t + 1] = (short)(dep + 1);
                s_qtail = t + 2;
            }
        }
        // Tree rotations (Kensler 2008) as a second pass: the top-down build is greedy, and a node may still gain from trading one
        // child for a grandchild on the other side when that shrinks the grandchild's parent.  One thread; a handful of sweeps.
        __syncthreads();
        const int n_nodes = s_count;
        if (i == 0 && n_nodes > 3) {
            auto internal = [&](int k) { return t_right[k] >= 0; };
            auto area2 = [&](int a, int b) {
                float e[3];
                for (int ax = 0; ax < 3; ++ax) e[ax] = fmaxf(t_box[a][3 + ax], t_box[b][3 + ax]) - fminf(t_box[a][ax], t_box[b][ax]);
                return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
            };
            auto refit = [&](int k) {
                const int A = t_left[k], B = t_right[k];
                for (int ax = 0; ax < 3; ++ax) {
                    t_box[k][ax] = fminf(t_box[A][ax], t_box[B][ax]);
                    t_box[k][3 + ax] = fmaxf(t_box[A][3 + ax], t_box[B][3 + ax]);
                }
            };
            for (int pass = 0; pass < 6; ++pass) {
                int changed = 0;
                for (int N = 0; N < n_nodes; ++N) {
                    if (!internal(N)) continue;
                    const int A = t_left[N], B = t_right[N];
                    float best = 0.0f;
                    int which = 0;   // 1: B <-> left(A), 2: B <-> right(A), 3: A <-> left(B), 4: A <-> right(B)
                    if (internal(A)) {
                        const float a0 = area2(A, A);
                        const float g1 = a0 - area2(B, t_right[A]), g2 = a0 - area2(t_left[A], B);
                        if (g1 > best) { best = g1; which = 1; }
                        if (g2 > best) { best = g2; which = 2; }
                    }
                    if (internal(B)) {
                        const float a0 = area2(B, B);
                        const float g3 = a0 - area2(A, t_right[B]), g4 = a0 - area2(t_left[B], A);
                        if (g3 > best) { best = g3; which = 3; }
                        if (g4 > best) { best = g4; which = 4; }
                    }
                    if (which == 0 || !(best > 1e-6f * area2(N, N))) continue;
                    if (which <= 2) {
                        const int g = which == 1 ? t_left[A] : t_right[A];   // the grandchild that moves up
                        if (which == 1) t_left[A] = B; else t_right[A] = B;
                        t_parent[B] = A;
                        t_right[N] = g;
                        t_parent[g] = N;
                        refit(A);
                    } else {
                        const int g = which == 3 ? t_left[B] : t_right[B];
                        if (which == 3) t_left[B] = A; else t_right[B] = A;
                        t_parent[A] = B;
                        t_left[N] = g;
                        t_parent[g] = N;
                        refit(B);
                    }
                    ++changed;
                }
                if (!changed) break;
            }
            s_depth = 0;
        }
        __syncthreads();
        for (int k = i; k < n_nodes; k += kMaxPrims) {
            out_fnodes[2 * k + 0] = make_float4(t_box[k][0], t_box[k][1], t_box[k][2], __int_as_float(t_left[k]));
            out_fnodes[2 * k + 1] = make_float4(t_box[k][3], t_box[k][4], t_box[k][5], __int_as_float(t_right[k]));
            if (t_right[k] < 0) {   // a leaf: internal nodes above it = stack entries the walk can need on the way
                if (((-t_right[k]) >> 20) != 0) atomicAdd(&s_cubN, 1);
                int d = 0;
                for (int q = t_parent[k]; q >= 0; q = t_parent[q]) ++d;
                atomicMax(&s_depth, d);
            }
        }
        __syncthreads();
        if (i == 0) out_meta[13] = s_cubN;
    }
    __syncthreads();
    if (i == 0) {
        out_meta[1] = s_depth;
        out_meta[2] = n_small;
        out_meta[10] = s_count;   // nodes of the walk's tree (2 * units - 1)
        out_meta[14] = s_tmask;
    }
}

// =====================================================================================================================
// Presentation step of the multi-GPU driver: the root holds n_ranks compact band buffers back to back (rows_pad rows each) and
// scatters their rows to the rows of the full window they belong to under the band interleave.  T = uint4 (16-byte units) or
// unsigned int (4-byte units); row_units = units per row.  Pure copy: HBM-bound, one unit per thread, coalesced both ways.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void assemble_bands_kernel(const T* __restrict__ gathered, T* __restrict__ full, unsigned int row_units,
                                                            unsigned int h, unsigned int band_h, unsigned int n_ranks, unsigned int rows_pad)
{
    const unsigned long long total = (unsigned long long)row_units * h;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned int r = (unsigned int)(i / row_units), x = (unsigned int)(i - (unsigned long long)r * row_units);
        const unsigned int band = r / band_h, g = band % n_ranks;
        const unsigned int k = (band / n_ranks) * band_h + (r - band * band_h);   // rank g's local row
        full[i] = gathered[((unsigned long long)g * rows_pad + k) * row_units + x];
    }
}

}  // namespace rtgo
