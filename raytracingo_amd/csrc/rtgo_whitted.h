// rtgo_whitted.h -- gfx950 device code of the "whitted" triangle path: the programs of the reference's cuda/whitted.cu
// (__raygen__pinhole :183-240, __miss__constant_radiance :243-246, __closesthit__occlusion :249-252, __closesthit__radiance
// :255-337 with cuda/LocalGeometry.h:55-141) over triangle meshes, without textures.  The reference hands triangle
// intersection and the acceleration structure to OptiX (built-in triangles, optixAccelBuild in sutil/Scene.cpp); here both are
// written out: a Moeller-Trumbore test with a fixed operation order (shared with the oracle, so the two agree bit for bit) and
// an LBVH over the triangles built on the device by one workgroup (Morton codes -> bitonic sort in LDS -> Karras hierarchy ->
// bottom-up fit -> subtrees of at most four triangles collapsed into leaves, one 64-byte record per remaining node holding
// BOTH children's boxes).  The render kernel is persistent like the analytic path's: one workgroup of 1024 threads per CU keeps
// the records in LDS (when they fit: always below ~2000 triangles, usually up to the limit), its waves pull 8 x 8-pixel tiles
// from a counter; one lane = one pixel, one launch per subframe: a primary ray plus one shadow ray per point light, no
// recursion (whitted.cu traces none either).
//
// Compiled with -ffp-contract=off like the rest of the library: one IEEE rounding per operation.
#pragma once

#include "rtgo_device.h"

namespace rtgo {
namespace whitted {

constexpr int kMaxTriangles = 8192;   // one workgroup sorts the Morton keys in (dynamic) LDS, 8 B per key; leaf codes and 2-byte stack entries hold 13 bits of triangle position
constexpr int kLeafShift = 13;        // leaf code = first sorted triangle | (count - 1) << kLeafShift   (count <= 4: 15 bits, what a stack entry holds beside its flag)
constexpr int kBuildThreads = 1024;
constexpr int kStack = 32;            // the bottom-up fit runs 2 * kStack + 2 = 66 passes: enough for any Karras hierarchy over 30-bit codes of <= 8192 triangles (depth <= 43)
constexpr int kRenderBlock = 1024;    // one workgroup per CU shares one LDS copy of the records
#ifndef RTGO_LEAF_TRIS
#define RTGO_LEAF_TRIS 4
#endif
constexpr int kLeafTris = RTGO_LEAF_TRIS;          // a subtree of at most this many triangles is one leaf of the walk
constexpr int kMaxWalkDepth = 40;
// where the walk reads from: records and triangles in L2 / fp32 records in LDS, triangles in L2 / quantised records, vertices and
// triangle indices all in LDS
constexpr int kAllInL2 = 0, kRecordsInLds = 1, kAllInLds = 2;
constexpr int kTileHeads = 32;        // tile-queue heads (<= 64), kTileHeadStride words apart
constexpr unsigned int kTileHeadStride = 16;     // words between two heads: one 64-byte line each

struct PointLight {   // Light::Point, cuda/Light.h:47-53
    float color[3];
    float intensity;
    float position[3];
    int falloff;      // never read by whitted.cu
};

struct Pbr {          // MaterialData::Pbr without its texture handles, cuda/MaterialData.h:43-52
    float base_color[4];
    float metallic, roughness;
};

// one texture of a material (cudaTextureObject_t of sutil::Scene::addSampler, sutil/Scene.cpp:505-538: RGBA8 read as normalised floats,
// normalised coordinates, and -- because addSampler compares its CUDA enums with GL constants -- always wrap addressing and linear filtering)
struct Tex {
    const uchar4* px;           // row 0 first; null: the material has no such texture
    unsigned int w, h;
};
struct MatTex {                 // MaterialData::Pbr's three handles, cuda/MaterialData.h:43-52
    Tex base_color, metallic_roughness, normal;
};

struct Params {       // whitted::LaunchParams, cuda/whitted.h:59-74
    const float4* recs;         // the walk's records, 4 float4 each: (left min, left link) (left max, -) (right min, right link) (right max, -);
                                //   link >= 0: a record; link < 0: a leaf, -1 - (first sorted triangle | (count - 1) << kLeafShift)
    const float4* tris;         // 3 float4 per triangle in Morton order: (P0, original index) (P1, -) (P2, -)
    // the compact form of the same structure, for meshes that fit a CU's LDS whole (kAllInLds):
    const uint4* qrecs;         // 2 uint4 per record: (x, y, z planes of the left box as lo | hi << 16 on a 16-bit grid, left link), same for the right box
    const uint2* tidx;          // per triangle in Morton order: (v0 | v1 << 16, v2 | original index << 16)
    int n_vertices;
    v3 grid_lo, grid_step;      // world = grid_lo + cell * grid_step
    int n_recs;                 // 0: the whole mesh is one leaf (at most kLeafTris triangles)
    int stack_depth;            // per-lane stack entries
    unsigned int* tile_counter; // this launch's tile queue heads (kTileHeads of them, zero at launch) and the set it zeroes for the next launch
    unsigned int* tile_counter_next;
    unsigned int tiles_x, tiles_y;
    unsigned int tile_stride;   // coprime to tiles_x * tiles_y
    const float* positions;     // 3 floats per vertex
    const float* normals;       // 3 floats per vertex, or null (then N = Ng, LocalGeometry.h:113-116)
    const unsigned int* indices;    // 3 per triangle
    const unsigned int* tri_material;
    const float* texcoords;     // 2 floats per vertex, or null (then UV = the barycentrics, LocalGeometry.h:97-102)
    const MatTex* mat_tex;      // per material, or null when no material has a texture
    const Pbr* materials;
    const PointLight* lights;
    int n_triangles, n_lights;
    float4* accum;
    uchar4* image;
    unsigned int width, height, subframe;
    v3 eye, U, V, W, miss;
    unsigned long long* counters;   // [0] rays_total [1] rays_occlusion
};

// tea<4>, cuda/random.h:30-45 with N = 4
__device__ __forceinline__ unsigned int tea4(unsigned int v0, unsigned int v1)
{
    unsigned int s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

__device__ __forceinline__ v3 ld3(const float* p, unsigned int i) { return mk(p[3 * i + 0], p[3 * i + 1], p[3 * i + 2]); }

// Moeller-Trumbore, two-sided (OptiX built-in triangles cull nothing unless asked to); the operation order is the contract
// between this kernel and the oracle (oracle_tri_intersect).  Accepts tmin < t < tmax like the analytic path.
__device__ __forceinline__ bool tri_intersect(v3 p0, v3 p1, v3 p2, v3 o, v3 d, float tmin, float tmax, float& t_out, float& u_out, float& v_out)
{
    const v3 e1 = vsub(p1, p0), e2 = vsub(p2, p0);
    const v3 pv = vcross(d, e2);
    const float det = vdot(e1, pv);
    if (det == 0.0f) return false;
    const float inv = div_cr(1.0f, det);   // (rtgo_device.h: the compiler's own correctly rounded steps; |det| is 0 or far above 2^-96 for triangles of any sane size)
    const v3 tv = vsub(o, p0);
    const float u = vdot(tv, pv) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    const v3 qv = vcross(tv, e1);
    const float v = vdot(d, qv) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = vdot(e2, qv) * inv;
    if (!(t > tmin && t < tmax)) return false;
    t_out = t;
    u_out = u;
    v_out = v;
    return true;
}

// slab test of a node box (exact reciprocal: the boxes carry a small pad, see build)
__device__ __forceinline__ bool node_hit(const float4 q0, const float4 q1, v3 o, v3 id, float tmin, float tmax, float& tn)
{
    float t0 = (q0.x - o.x) * id.x, t1 = (q1.x - o.x) * id.x;
    float a = fminf(t0, t1), b = fmaxf(t0, t1);
    t0 = (q0.y - o.y) * id.y;
    t1 = (q1.y - o.y) * id.y;
    a = fmaxf(a, fminf(t0, t1));
    b = fminf(b, fmaxf(t0, t1));
    t0 = (q0.z - o.z) * id.z;
    t1 = (q1.z - o.z) * id.z;
    a = fmaxf(a, fminf(t0, t1));
    b = fminf(b, fmaxf(t0, t1));
    a = fmaxf(a, tmin);
    b = fminf(b, tmax);
    tn = a;
    return a <= b * 1.000002f + 1e-7f;
}

// the triangles [first, first + cnt) of one leaf, in Morton order (contiguous 48-byte records).  All of them are asked for before
// the first one is tested: one round trip to L2 per leaf instead of one per triangle.
template <bool ANY>
__device__ __forceinline__ bool leaf_tris(const float4* __restrict__ tris, int first, int cnt, v3 o, v3 d, float tmin, float tmax, int& best, int& best_pos,
                                          float& bt, float& bu, float& bv)
{
    float4 a[kLeafTris], b[kLeafTris], c[kLeafTris];
#pragma unroll
    for (int k = 0; k < kLeafTris; ++k) {
        a[k] = b[k] = c[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (k < cnt) {   // (lanes whose leaf is shorter ask for nothing: the memory pipeline is paced by addresses, not by instructions)
            a[k] = tris[3 * (first + k) + 0];
            b[k] = tris[3 * (first + k) + 1];
            c[k] = tris[3 * (first + k) + 2];
        }
    }
#pragma unroll
    for (int k = 0; k < kLeafTris; ++k) {
        if (k < cnt) {
            const int tri = __float_as_int(a[k].w);
            float t, u, v;
            if (tri_intersect(mk(a[k].x, a[k].y, a[k].z), mk(b[k].x, b[k].y, b[k].z), mk(c[k].x, c[k].y, c[k].z), o, d, tmin, tmax, t, u, v) &&
                (t < bt || (t == bt && best >= 0 && tri < best))) {
                bt = t;
                bu = u;
                bv = v;
                best = tri;
                best_pos = first + k;
                if (ANY) return true;
            }
        }
    }
    return false;
}

// closest hit (ANY = false: smallest t, lowest triangle index on ties) or any hit (ANY = true: the occlusion ray's
// OPTIX_RAY_FLAG_TERMINATE_ON_FIRST_HIT, whitted.cu:140-151) over the triangle LBVH.  A step reads ONE record -- both
// children's boxes, 64 contiguous bytes -- descends into the nearer child that is hit and parks the other one on the lane's
// stack (2-byte entries: a record index, or 0x8000 | leaf code).
template <bool ANY, typename Recs>
__device__ __forceinline__ bool trace(const Params& p, Recs recs, unsigned short* __restrict__ s_stack, int stride, v3 o, v3 d, float tmin, float tmax,
                                      int& tri_out, int& pos_out, float& t_out, float& u_out, float& v_out)
{
    auto safe_inv = [](float x) { return fabsf(x) < 1e-30f ? copysignf(1e30f, x) : 1.0f / x; };
    const v3 id = mk(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    int best = -1, best_pos = 0;
    float bt = tmax, bu = 0.0f, bv = 0.0f;
    if (p.n_recs == 0) {
        leaf_tris<ANY>(p.tris, 0, p.n_triangles, o, d, tmin, tmax, best, best_pos, bt, bu, bv);
    } else {
        int sp = 0;
        int cur = 0;   // >= 0: a record; < 0: a leaf code
        for (;;) {
            bool pop = true;
            if (cur >= 0) {
                const float4 a0 = recs[4 * cur + 0], a1 = recs[4 * cur + 1], b0 = recs[4 * cur + 2], b1 = recs[4 * cur + 3];
                float tl, tr;
                const bool hl = node_hit(a0, a1, o, id, tmin, bt, tl);
                const bool hr = node_hit(b0, b1, o, id, tmin, bt, tr);
                const int ll = __float_as_int(a0.w), lr = __float_as_int(b0.w);
                const bool go_r = hr && (!hl || tr < tl);
                if (hl && hr) {
                    const int far = go_r ? ll : lr;
                    s_stack[sp * stride] = (unsigned short)(far >= 0 ? far : (0x8000 | (-1 - far)));
                    ++sp;
                }
                if (hl || hr) {
                    cur = go_r ? lr : ll;
                    pop = false;
                }
            } else {
                const int code = -1 - cur;
                if (leaf_tris<ANY>(p.tris, code & ((1 << kLeafShift) - 1), (code >> kLeafShift) + 1, o, d, tmin, tmax, best, best_pos, bt, bu, bv)) break;   // (true only when ANY)
            }
            if (pop) {
                if (sp == 0) break;
                --sp;
                const int e = (int)s_stack[sp * stride];
                cur = (e & 0x8000) ? -1 - (e & 0x7FFF) : e;
            }
        }
    }
    tri_out = best;
    pos_out = best_pos;
    t_out = bt;
    u_out = bu;
    v_out = bv;
    return best >= 0;
}

// whitted.cu:49-80 (powf / sqrtf are the device libm's)
__device__ __forceinline__ v3 schlick(v3 spec, float VdotH)
{
    const float k = powf(1.0f - VdotH, 5.0f);
    return vadd(spec, vscale(vsub(mk(1.0f, 1.0f, 1.0f), spec), k));
}
__device__ __forceinline__ float vis(float NdotL, float NdotV, float alpha)
{
    const float a2 = alpha * alpha;
    const float g0 = NdotL * sqrt_cr(NdotV * NdotV * (1.0f - a2) + a2);
    const float g1 = NdotV * sqrt_cr(NdotL * NdotL * (1.0f - a2) + a2);
    return 2.0f * NdotL * NdotV / (g0 + g1);
}
__device__ __forceinline__ float ggx_normal(float NdotH, float alpha)
{
    const float a2 = alpha * alpha;
    const float n2 = NdotH * NdotH;
    const float x = n2 * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * x * x);
}

// tex2D<float4>( tex, u, v ) of a cudaReadModeNormalizedFloat / normalizedCoords / cudaAddressModeWrap / cudaFilterModeLinear texture, as
// the CUDA programming guide states it (appendix "Texture Fetching", linear filtering): x = u N, xB = x - 0.5, i = floor(xB),
// alpha = frac(xB) kept in 1.8 fixed point (8 fractional bits; rounded to nearest here -- the hardware's rounding is not published:
// parity unpinned), tex = (1-a)(1-b) T[i,j] + a (1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1] with indices wrapped.
// The operation order is the contract between this kernel and the oracle (oracle_tex2d).
__device__ __forceinline__ float4 tex2d(const Tex t, float u, float v)
{
    const float xb = u * (float)t.w - 0.5f, yb = v * (float)t.h - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float a = floorf((xb - fx) * 256.0f + 0.5f) * (1.0f / 256.0f), b = floorf((yb - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    // wrap: floored modulo of the integer texel index (|u|, |v| stay far below 2^24 / N)
    const int w = (int)t.w, h = (int)t.h;
    int i0 = (int)fx % w, j0 = (int)fy % h;
    i0 += i0 < 0 ? w : 0;
    j0 += j0 < 0 ? h : 0;
    const int i1 = i0 + 1 == w ? 0 : i0 + 1, j1 = j0 + 1 == h ? 0 : j0 + 1;
    const uchar4 t00 = t.px[(size_t)j0 * t.w + i0], t10 = t.px[(size_t)j0 * t.w + i1], t01 = t.px[(size_t)j1 * t.w + i0], t11 = t.px[(size_t)j1 * t.w + i1];
    const float k = 1.0f / 255.0f;
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    auto mix = [&](unsigned char c00, unsigned char c10, unsigned char c01, unsigned char c11) {
        return w00 * ((float)c00 * k) + w10 * ((float)c10 * k) + w01 * ((float)c01 * k) + w11 * ((float)c11 * k);
    };
    return make_float4(mix(t00.x, t10.x, t01.x, t11.x), mix(t00.y, t10.y, t01.y, t11.y), mix(t00.z, t10.z, t01.z, t11.z), mix(t00.w, t10.w, t01.w, t11.w));
}

// ---- the walk over the LDS-resident compact form (kAllInLds) -------------------------------------------------------------
// slab test of a quantised box: world plane = grid_lo + cell * step, so t = fma(cell, step / d, (grid_lo - o) / d) -- one FMA per
// plane on two per-ray constants.  The build rounds lower planes down and upper planes up by one extra cell (3e-5 of the scene), which
// swallows the rounding of these products: the boxes only grow.
__device__ __forceinline__ bool qbox_hit(const uint4 q, v3 sid, v3 snoid, float tmin, float tmax, float& tn)
{
    float t0 = fmaf((float)(q.x & 0xFFFFu), sid.x, snoid.x), t1 = fmaf((float)(q.x >> 16), sid.x, snoid.x);
    float a = fminf(t0, t1), b = fmaxf(t0, t1);
    t0 = fmaf((float)(q.y & 0xFFFFu), sid.y, snoid.y);
    t1 = fmaf((float)(q.y >> 16), sid.y, snoid.y);
    a = fmaxf(a, fminf(t0, t1));
    b = fminf(b, fmaxf(t0, t1));
    t0 = fmaf((float)(q.z & 0xFFFFu), sid.z, snoid.z);
    t1 = fmaf((float)(q.z >> 16), sid.z, snoid.z);
    a = fmaxf(a, fminf(t0, t1));
    b = fminf(b, fmaxf(t0, t1));
    a = fmaxf(a, tmin);
    b = fminf(b, tmax);
    tn = a;
    return a <= b * 1.000002f + 1e-7f;
}

template <bool ANY>
__device__ __forceinline__ bool leaf_tris_lds(const float4* __restrict__ s_verts, const uint2* __restrict__ s_tidx, int first, int cnt, v3 o, v3 d, float tmin,
                                              float tmax, int& best, int& best_pos, float& bt, float& bu, float& bv)
{
    uint2 ti[kLeafTris];
#pragma unroll
    for (int k = 0; k < kLeafTris; ++k) ti[k] = s_tidx[first + (k < cnt ? k : 0)];
    float4 a[kLeafTris], b[kLeafTris], c[kLeafTris];
#pragma unroll
    for (int k = 0; k < kLeafTris; ++k) {
        a[k] = s_verts[ti[k].x & 0xFFFFu];
        b[k] = s_verts[ti[k].x >> 16];
        c[k] = s_verts[ti[k].y & 0xFFFFu];
    }
#pragma unroll
    for (int k = 0; k < kLeafTris; ++k) {
        if (k < cnt) {
            const int tri = (int)(ti[k].y >> 16);
            float t, u, v;
            if (tri_intersect(mk(a[k].x, a[k].y, a[k].z), mk(b[k].x, b[k].y, b[k].z), mk(c[k].x, c[k].y, c[k].z), o, d, tmin, tmax, t, u, v) &&
                (t < bt || (t == bt && best >= 0 && tri < best))) {
                bt = t;
                bu = u;
                bv = v;
                best = tri;
                best_pos = first + k;
                if (ANY) return true;
            }
        }
    }
    return false;
}

template <bool ANY>
__device__ __forceinline__ bool trace_lds(const Params& p, const uint4* __restrict__ s_q, const float4* __restrict__ s_verts, const uint2* __restrict__ s_tidx,
                                          unsigned short* __restrict__ s_stack, int stride, v3 o, v3 d, float tmin, float tmax, int& tri_out, int& pos_out,
                                          float& t_out, float& u_out, float& v_out)
{
    auto safe_inv = [](float x) { return fabsf(x) < 1e-30f ? copysignf(1e30f, x) : 1.0f / x; };
    const v3 id = mk(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    const v3 sid = mk(p.grid_step.x * id.x, p.grid_step.y * id.y, p.grid_step.z * id.z);
    const v3 snoid = mk((p.grid_lo.x - o.x) * id.x, (p.grid_lo.y - o.y) * id.y, (p.grid_lo.z - o.z) * id.z);
    int best = -1, best_pos = 0;
    float bt = tmax, bu = 0.0f, bv = 0.0f;
    if (p.n_recs == 0) {
        leaf_tris_lds<ANY>(s_verts, s_tidx, 0, p.n_triangles, o, d, tmin, tmax, best, best_pos, bt, bu, bv);
    } else {
        // while-while: every lane first descends to its next leaf (or runs out of work), then the lanes that hold a leaf test it
        int sp = 0;
        int cur = 0;
        bool have = true;
        auto pop = [&]() {
            have = sp > 0;
            if (have) {
                --sp;
                const int e = (int)s_stack[sp * stride];
                cur = (e & 0x8000) ? -1 - (e & 0x7FFF) : e;
            }
        };
        while (have) {
            while (have && cur >= 0) {
                const uint4 L = s_q[2 * cur + 0], R = s_q[2 * cur + 1];
                float tl, tr;
                const bool hl = qbox_hit(L, sid, snoid, tmin, bt, tl);
                const bool hr = qbox_hit(R, sid, snoid, tmin, bt, tr);
                const int ll = (int)L.w, lr = (int)R.w;
                const bool go_r = hr && (!hl || tr < tl);
                if (hl && hr) {
                    const int far = go_r ? ll : lr;
                    s_stack[sp * stride] = (unsigned short)(far >= 0 ? far : (0x8000 | (-1 - far)));
                    ++sp;
                }
                if (hl || hr) cur = go_r ? lr : ll;
                else pop();
            }
            if (have) {
                const int code = -1 - cur;
                if (leaf_tris_lds<ANY>(s_verts, s_tidx, code & ((1 << kLeafShift) - 1), (code >> kLeafShift) + 1, o, d, tmin, tmax, best, best_pos, bt, bu, bv)) break;   // (true only when ANY)
                pop();
            }
        }
    }
    tri_out = best;
    pos_out = best_pos;
    t_out = bt;
    u_out = bu;
    v_out = bv;
    return best >= 0;
}

template <int MODE>
__global__ __launch_bounds__(kRenderBlock) void render_kernel(const Params p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char w_smem[];
#ifdef RTGO_WHITTED_TIMING
    const unsigned long long wt0 = wall_clock64();
    unsigned long long wt_tiles = 0, wt_n = 0, wt_max = 0;
#endif
    // LDS image.  kRecordsInLds: [fp32 records, 64 B each][stacks];  kAllInLds: [quantised records, 32 B][vertices, 16 B][triangle
    // indices, 8 B][stacks];  kAllInL2: [stacks]
    float4* s_recs = reinterpret_cast<float4*>(w_smem);
    uint4* s_q = reinterpret_cast<uint4*>(w_smem);
    float4* s_verts = reinterpret_cast<float4*>(s_q + 2 * p.n_recs);
    uint2* s_tidx = reinterpret_cast<uint2*>(s_verts + p.n_vertices);
    const int stride = (int)blockDim.x;
    unsigned short* s_stack = (MODE == kAllInLds ? reinterpret_cast<unsigned short*>(s_tidx + p.n_triangles)
                                                 : reinterpret_cast<unsigned short*>(s_recs + (MODE == kRecordsInLds ? 4 * p.n_recs : 0))) + threadIdx.x;   // entry e at [e * stride]
    if (MODE == kRecordsInLds) {
        for (int i = (int)threadIdx.x; i < 4 * p.n_recs; i += stride) s_recs[i] = p.recs[i];
        __syncthreads();
    }
    if (MODE == kAllInLds) {
        for (int i = (int)threadIdx.x; i < 2 * p.n_recs; i += stride) s_q[i] = p.qrecs[i];
        for (int i = (int)threadIdx.x; i < p.n_vertices; i += stride) s_verts[i] = make_float4(p.positions[3 * i + 0], p.positions[3 * i + 1], p.positions[3 * i + 2], 0.0f);
        for (int i = (int)threadIdx.x; i < p.n_triangles; i += stride) s_tidx[i] = p.tidx[i];
        __syncthreads();
    }
#ifdef RTGO_WHITTED_TIMING
    const unsigned long long wt1 = wall_clock64();
#endif
    if (blockIdx.x == 0 && threadIdx.x < (unsigned int)kTileHeads) p.tile_counter_next[kTileHeadStride * threadIdx.x] = 0u;
    auto pick = [&]() {
        if constexpr (MODE == kRecordsInLds) return static_cast<const float4*>(s_recs);
        else return p.recs;
    };
    const auto recs = pick();
    // closest hit / any hit through whichever form of the structure this instantiation walks
    auto closest = [&](v3 o, v3 d, float t0, float t1, int& tri, int& pos, float& t, float& u, float& v) -> bool {
        if constexpr (MODE == kAllInLds) return trace_lds<false>(p, s_q, s_verts, s_tidx, s_stack, stride, o, d, t0, t1, tri, pos, t, u, v);
        else return trace<false>(p, recs, s_stack, stride, o, d, t0, t1, tri, pos, t, u, v);
    };
    auto occluded = [&](v3 o, v3 d, float t0, float t1) -> bool {
        int tri, pos;
        float t, u, v;
        if constexpr (MODE == kAllInLds) return trace_lds<true>(p, s_q, s_verts, s_tidx, s_stack, stride, o, d, t0, t1, tri, pos, t, u, v);
        else return trace<true>(p, recs, s_stack, stride, o, d, t0, t1, tri, pos, t, u, v);
    };
    const unsigned int lane = threadIdx.x & 63u;
    const unsigned int n_tiles = p.tiles_x * p.tiles_y;
    unsigned int rays = 0, occl = 0;
    // Tile queue: kTileHeads counters 64 bytes apart, head h serves the tiles h, h + kTileHeads, ...  (one counter hands out ~88
    // entries per microsecond chip-wide: a 1080p subframe is 32 k tiles).  A wave starts on head blockIdx % kTileHeads; when that is
    // dry it looks at all heads with one load and moves to an open one.  The pull for the next tile is in flight while the
    // current one is rendered.
    auto tiles_of = [&](unsigned int h) { return (n_tiles + (unsigned int)kTileHeads - 1u - h) / (unsigned int)kTileHeads; };
    unsigned int head = blockIdx.x % (unsigned int)kTileHeads;
    unsigned int pending = 0u;
    if (lane == 0u) pending = atomicAdd(p.tile_counter + kTileHeadStride * head, 1u);
    for (;;) {
        unsigned int pos = (unsigned int)__builtin_amdgcn_readfirstlane((int)pending);
        if (pos >= tiles_of(head)) {
            // this head is dry: any other one still open?
            unsigned int v = 0xFFFFFFFFu;
            if (lane < (unsigned int)kTileHeads) v = __hip_atomic_load(p.tile_counter + kTileHeadStride * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long open = __builtin_amdgcn_ballot_w64(lane < (unsigned int)kTileHeads && v < tiles_of(lane));
            if (open == 0ull) break;
            // the first open head after this one (waves spread over the open heads instead of all falling on the lowest)
            const unsigned long long above = open & ~((2ull << head) - 1ull);
            head = (unsigned int)(__ffsll((long long)(above ? above : open)) - 1);
            if (lane == 0u) pending = atomicAdd(p.tile_counter + kTileHeadStride * head, 1u);
            continue;
        }
        // queue entry -> tile through a fixed permutation (multiplication by a number coprime to the tile count, near the golden
        // section of it): in row-major order every wave of the chip reaches the dense part of the mesh at the same time and they all
        // queue at the vector memory pipeline for its triangles; scattered, tiles that wait for triangles overlap tiles that do not
        const unsigned int tile = (unsigned int)(((unsigned long long)(pos * (unsigned int)kTileHeads + head) * p.tile_stride) % n_tiles);
        if (lane == 0u) pending = atomicAdd(p.tile_counter + kTileHeadStride * head, 1u);
#ifdef RTGO_WHITTED_TIMING
        const unsigned long long wt2 = wall_clock64();
        wt_n += 1;
#endif
        const unsigned int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
        const unsigned int x = tx * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
        if (x < p.width && y < p.height) {
            const unsigned int idx = y * p.width + x;
            // __raygen__pinhole, whitted.cu:183-240
            unsigned int seed = tea4(y * p.width + x, p.subframe);
            float jx = 0.0f, jy = 0.0f;
            if (p.subframe != 0) {
                jx = rnd(seed) - 0.5f;   // x first (source order)
                jy = rnd(seed) - 0.5f;
            }
            const float dx = 2.0f * div_cr((float)x + jx, (float)p.width) - 1.0f;
            const float dy = 2.0f * div_cr((float)y + jy, (float)p.height) - 1.0f;
            const v3 rd = vnormalize(vadd(vadd(vscale(p.U, dx), vscale(p.V, dy)), p.W));
            const v3 ro = p.eye;
            v3 result = p.miss;   // __miss__constant_radiance, :243-246
            int tri, tpos;
            float t, bu, bv;
            rays += 1;
            if (closest(ro, rd, 0.01f, 1e16f, tri, tpos, t, bu, bv)) {
                // __closesthit__radiance, :255-337, with getLocalGeometry (LocalGeometry.h:55-141) for a mesh in world space.
                // (the corners come from the Morton-ordered copy the walk read them from: same values, no trip through the index array)
                float4 c0, c1, c2;
                unsigned int i0 = 0, i1 = 0, i2 = 0;   // vertex indices (for the normals)
                if constexpr (MODE == kAllInLds) {
                    const uint2 ti = s_tidx[tpos];
                    i0 = ti.x & 0xFFFFu;
                    i1 = ti.x >> 16;
                    i2 = ti.y & 0xFFFFu;
                    c0 = s_verts[i0];
                    c1 = s_verts[i1];
                    c2 = s_verts[i2];
                } else {
                    c0 = p.tris[3 * tpos + 0];
                    c1 = p.tris[3 * tpos + 1];
                    c2 = p.tris[3 * tpos + 2];
                    if (p.normals || p.texcoords) {
                        i0 = p.indices[3 * tri + 0];
                        i1 = p.indices[3 * tri + 1];
                        i2 = p.indices[3 * tri + 2];
                    }
                }
                const v3 P0 = mk(c0.x, c0.y, c0.z), P1 = mk(c1.x, c1.y, c1.z), P2 = mk(c2.x, c2.y, c2.z);
                const float w0 = 1.0f - bu - bv;
                const v3 P = vadd(vadd(vscale(P0, w0), vscale(P1, bu)), vscale(P2, bv));
                const v3 Ng = vnormalize(vcross(vsub(P1, P0), vsub(P2, P0)));
                v3 N = Ng;
                if (p.normals) {
                    const v3 N0 = ld3(p.normals, i0), N1 = ld3(p.normals, i1), N2 = ld3(p.normals, i2);
                    N = vnormalize(vadd(vadd(vscale(N0, w0), vscale(N1, bu)), vscale(N2, bv)));
                }
                const unsigned int mi = p.tri_material ? p.tri_material[tri] : 0u;
                const Pbr m = p.materials[mi];
                v3 base = mk(m.base_color[0], m.base_color[1], m.base_color[2]);
                float mr_y = 1.0f, mr_z = 1.0f;   // the (1,1,1,1) of an absent metallic-roughness texture, whitted.cu:271
                if (p.mat_tex) {
                    const MatTex mt = p.mat_tex[mi];
                    if (mt.base_color.px || mt.metallic_roughness.px || mt.normal.px) {
                        // getLocalGeometry's UV and dp/du, dp/dv (LocalGeometry.h:88-135)
                        float2 UV0 = make_float2(0.0f, 0.0f), UV1 = make_float2(0.0f, 1.0f), UV2 = make_float2(1.0f, 0.0f), UV = make_float2(bu, bv);
                        if (p.texcoords) {
                            UV0 = make_float2(p.texcoords[2 * i0], p.texcoords[2 * i0 + 1]);
                            UV1 = make_float2(p.texcoords[2 * i1], p.texcoords[2 * i1 + 1]);
                            UV2 = make_float2(p.texcoords[2 * i2], p.texcoords[2 * i2 + 1]);
                            UV = make_float2(w0 * UV0.x + bu * UV1.x + bv * UV2.x, w0 * UV0.y + bu * UV1.y + bv * UV2.y);
                        }
                        if (mt.base_color.px) {
                            // base_color *= linearize( tex2D ), whitted.cu:78-85, 264-267
                            const float4 tc = tex2d(mt.base_color, UV.x, UV.y);
                            base = vmul(base, mk(powf(tc.x, 2.2f), powf(tc.y, 2.2f), powf(tc.z, 2.2f)));
                        }
                        if (mt.metallic_roughness.px) {
                            const float4 tc = tex2d(mt.metallic_roughness, UV.x, UV.y);   // (occlusion, roughness, metallic), :272-276
                            mr_y = tc.y;
                            mr_z = tc.z;
                        }
                        if (mt.normal.px) {
                            // whitted.cu:288-292 over LocalGeometry.h:118-134
                            const float du1 = UV0.x - UV2.x, du2 = UV1.x - UV2.x, dv1 = UV0.y - UV2.y, dv2 = UV1.y - UV2.y;
                            const v3 dp1 = vsub(P0, P2), dp2 = vsub(P1, P2);
                            const float det = du1 * dv2 - dv1 * du2;
                            const float invdet = 1.0f / det;
                            const v3 dpdu = vscale(vsub(vscale(dp1, dv2), vscale(dp2, dv1)), invdet);
                            const v3 dpdv = vscale(vadd(vscale(dp1, -du2), vscale(dp2, du1)), invdet);
                            const float4 tc = tex2d(mt.normal, UV.x, UV.y);
                            const float nx = 2.0f * tc.x - 1.0f, ny = 2.0f * tc.y - 1.0f, nz = 2.0f * tc.z - 1.0f;
                            N = vnormalize(vadd(vadd(vscale(vnormalize(dpdu), nx), vscale(vnormalize(dpdv), ny)), vscale(N, nz)));
                        }
                    }
                }
                const float metallic = m.metallic * mr_z, roughness = m.roughness * mr_y;   // :269-276
                const float F0 = 0.04f;
                const v3 diff_color = vscale(vscale(base, 1.0f - F0), 1.0f - metallic);
                // lerp(a, b, t) = a + t * (b - a), vec_math.h:496-499
                const v3 spec_color = vadd(mk(F0, F0, F0), vscale(vsub(base, mk(F0, F0, F0)), metallic));
                const float alpha = roughness * roughness;
                result = mk(0.0f, 0.0f, 0.0f);
                for (int l = 0; l < p.n_lights; ++l) {
                    const PointLight L = p.lights[l];
                    const v3 toL = vsub(mk(L.position[0], L.position[1], L.position[2]), P);
                    const float Ldist = vlength(toL);
                    const v3 Lv = vscale(toL, 1.0f / Ldist);   // float3 / float multiplies by the reciprocal (vec_math.h:479-483)
                    const v3 Vv = vneg(vnormalize(rd));
                    const v3 H = vnormalize(vadd(Lv, Vv));
                    const float NdotL = vdot(N, Lv), NdotV = vdot(N, Vv), NdotH = vdot(N, H), VdotH = vdot(Vv, H);
                    if (NdotL > 0.0f && NdotV > 0.0f) {
                        rays += 1;
                        occl += 1;
                        if (!occluded(P, Lv, 0.001f, Ldist - 0.001f)) {
                            const v3 F = schlick(spec_color, VdotH);
                            const float G = vis(NdotL, NdotV, alpha);
                            const float D = ggx_normal(NdotH, alpha);
                            const v3 one_minus_F = vsub(mk(1.0f, 1.0f, 1.0f), F);
                            const v3 dd = vmul(one_minus_F, diff_color);
                            const float ip = 1.0f / kPi;
                            const v3 diff = vscale(dd, ip);   // float3 / float
                            const v3 spec = vscale(vscale(F, G), D);
                            const v3 lc = vscale(mk(L.color[0], L.color[1], L.color[2]), L.intensity);
                            result = vadd(result, vmul(vscale(lc, NdotL), vadd(diff, spec)));
                        }
                    }
                }
            }
            // whitted.cu:226-239
            v3 acc = result;
            if (p.subframe > 0) {
                const float a = 1.0f / (float)(p.subframe + 1);
                const float4 prev = p.accum[idx];
                acc = vadd(mk(prev.x, prev.y, prev.z), vscale(vsub(acc, mk(prev.x, prev.y, prev.z)), a));
            }
            p.accum[idx] = make_float4(acc.x, acc.y, acc.z, 1.0f);
            // make_color, whitted.cu:164-173: gamma 2.2
            const float g = (float)(1.0 / 2.2f);
            p.image[idx] = make_uchar4((unsigned char)(powf(clampf(acc.x, 0.0f, 1.0f), g) * 255.0f), (unsigned char)(powf(clampf(acc.y, 0.0f, 1.0f), g) * 255.0f),
                                       (unsigned char)(powf(clampf(acc.z, 0.0f, 1.0f), g) * 255.0f), 255u);
        }
#ifdef RTGO_WHITTED_TIMING
        const unsigned long long wt3 = wall_clock64() + (rays == 0xFFFFFFFFu ? 1 : 0) - wt2;
        wt_tiles += wt3;
        if (x < p.width && y < p.height) p.accum[y * p.width + x].w = (float)wt3;   // (diagnostic: ticks of the tile)
        wt_max = wt3 > wt_max ? wt3 : wt_max;
#endif
    }
    rays = wave_sum(rays);
    occl = wave_sum(occl);
    if (lane == 0u) {
        atomicAdd(&p.counters[0], (unsigned long long)rays);
        atomicAdd(&p.counters[1], (unsigned long long)occl);
#ifdef RTGO_WHITTED_TIMING
        // diagnostic build (tools/whitted_perf.py prints them): 10 ns ticks summed over the waves
        atomicAdd(&p.counters[2], wall_clock64() - wt0);   // wave lifetime          -> rtgo_stats.node_visits
        atomicAdd(&p.counters[4], wt_tiles);               // inside tiles           -> hits
        atomicAdd(&p.counters[5], wt_n);                   // tiles                  -> dbg_fast_boxes
        atomicMax(&p.counters[6], wt_max);                 // the longest tile       -> dbg_fast_tests
#endif
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LBVH over the triangles, one workgroup: replaces optixAccelBuild over OPTIX_BUILD_INPUT_TYPE_TRIANGLES (sutil/Scene.cpp,
// buildMeshAccels).  Same recipe as the analytic path's canonical tree: 30-bit Morton code of the centroid normalised to the
// scene bounds, stable order by (code, triangle index), Karras 2012, one triangle per leaf, bottom-up fit.
// Boxes are padded by 1e-4 of the scene's extent + 1e-6: the slab test rounds, the triangle test must never be cut off.
// out_meta = {depth of the hierarchy, records of the walk, stack entries the walk needs, grid origin xyz, grid step xyz (float bits)}.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBuildThreads) void build_kernel(const float* __restrict__ positions, const unsigned int* __restrict__ indices, int n,
                                                              float4* __restrict__ nodes, int* __restrict__ parent, int* __restrict__ visit,
                                                              int* __restrict__ first_of, int* __restrict__ count_of, int* __restrict__ rec_of,
                                                              float4* __restrict__ recs, float4* __restrict__ tris, uint4* __restrict__ qrecs,
                                                              uint2* __restrict__ tidx, int* __restrict__ out_meta)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char build_keys_dyn[];   // kMaxTriangles x 8 B: the launch passes the size
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(build_keys_dyn);
    __shared__ float s_red[6][kBuildThreads];
    __shared__ int s_depth, s_nrec, s_wdepth;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_depth = 0;
        s_nrec = 1;     // record 0 is the root's
        s_wdepth = 0;
    }
    // scene bounds
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += kBuildThreads)
        for (int k = 0; k < 3; ++k) {
            const unsigned int vi = indices[3 * i + k];
            for (int a = 0; a < 3; ++a) {
                const float c = positions[3 * vi + a];
                lo[a] = fminf(lo[a], c);
                hi[a] = fmaxf(hi[a], c);
            }
        }
    for (int a = 0; a < 3; ++a) {
        s_red[a][tid] = lo[a];
        s_red[3 + a][tid] = hi[a];
    }
    __syncthreads();
    for (int stride = kBuildThreads / 2; stride > 0; stride >>= 1) {
        if (tid < stride)
            for (int a = 0; a < 3; ++a) {
                s_red[a][tid] = fminf(s_red[a][tid], s_red[a][tid + stride]);
                s_red[3 + a][tid] = fmaxf(s_red[3 + a][tid], s_red[3 + a][tid + stride]);
            }
        __syncthreads();
    }
    float blo[3], ext[3], maxext = 0.0f;
    for (int a = 0; a < 3; ++a) {
        blo[a] = s_red[a][0];
        ext[a] = s_red[3 + a][0] - s_red[a][0];
        maxext = fmaxf(maxext, ext[a]);
    }
    const float pad = maxext * 1e-4f + 1e-6f;
    // Morton keys
    for (int i = tid; i < kMaxTriangles; i += kBuildThreads) {
        unsigned long long key = ~0ull;
        if (i < n) {
            unsigned int q[3];
            for (int a = 0; a < 3; ++a) {
                const float c = (positions[3 * indices[3 * i + 0] + a] + positions[3 * indices[3 * i + 1] + a] + positions[3 * indices[3 * i + 2] + a]) * (1.0f / 3.0f);
                const float u = ext[a] > 0.0f ? (c - blo[a]) / ext[a] : 0.0f;
                q[a] = (unsigned int)fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
            }
            key = ((unsigned long long)((expand_bits(q[0]) << 2) | (expand_bits(q[1]) << 1) | expand_bits(q[2])) << 32) | (unsigned int)i;
        }
        s_keys[i] = key;
    }
    __syncthreads();
    // bitonic sort of kMaxTriangles keys (unique: the index is part of the key)
    for (int k = 2; k <= kMaxTriangles; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < kMaxTriangles; i += kBuildThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s_keys[i] = b;
                        s_keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    auto delta = [&](int i, int j) -> int {
        if (j < 0 || j >= n) return -1;
        const unsigned int a = (unsigned int)(s_keys[i] >> 32), b = (unsigned int)(s_keys[j] >> 32);
        if (a == b) return 32 + __clz((unsigned int)i ^ (unsigned int)j);
        return __clz(a ^ b);
    };
    const int leaf0 = n - 1;
    // leaves
    for (int i = tid; i < n; i += kBuildThreads) {
        const int tri = (int)(s_keys[i] & 0xFFFFFFFFu);
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) {
                const float c = positions[3 * indices[3 * tri + k] + a];
                l[a] = fminf(l[a], c);
                h[a] = fmaxf(h[a], c);
            }
        nodes[2 * (leaf0 + i) + 0] = make_float4(l[0] - pad, l[1] - pad, l[2] - pad, __int_as_float(tri));
        nodes[2 * (leaf0 + i) + 1] = make_float4(h[0] + pad, h[1] + pad, h[2] + pad, __int_as_float(-1));
    }
    for (int i = tid; i < 2 * n - 1; i += kBuildThreads) parent[i] = -1;
    __syncthreads();
    // Karras 2012
    for (int i = tid; i < n - 1; i += kBuildThreads) {
        const int d = (delta(i, i + 1) - delta(i, i - 1)) >= 0 ? 1 : -1;
        const int dmin = delta(i, i - d);
        int lmax = 2;
        while (delta(i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = delta(i, j);
        int s = 0, t = l;
        do {
            t = (t + 1) / 2;
            if (delta(i, i + (s + t) * d) > dnode) s += t;
        } while (t > 1);
        const int gamma = i + s * d + (d < 0 ? -1 : 0);
        const int lo_i = i < j ? i : j, hi_i = i < j ? j : i;
        const int left = (lo_i == gamma) ? leaf0 + gamma : gamma;
        const int right = (hi_i == gamma + 1) ? leaf0 + gamma + 1 : gamma + 1;
        nodes[2 * i + 0].w = __int_as_float(left);
        nodes[2 * i + 1].w = __int_as_float(right);
        parent[left] = i;
        parent[right] = i;
        first_of[i] = lo_i;               // the node's triangles: sorted positions [lo_i, hi_i]
        count_of[i] = hi_i - lo_i + 1;
    }
    __syncthreads();
    // bottom-up fit, level by level: a node is fitted in the pass after both of its children (visit[] = 1 once a node is done;
    // leaves are done from the start).  A workgroup barrier orders the passes, global memory included.
    for (int i = tid; i < n; i += kBuildThreads) visit[i] = 0;   // (per internal node)
    __syncthreads();
    for (int pass = 0; pass < 2 * kStack + 2; ++pass) {
        int fitted[kMaxTriangles / kBuildThreads], nf = 0;   // nodes this thread fits in this pass
        for (int i = tid; i < n - 1; i += kBuildThreads) {
            if (visit[i]) continue;
            const int L = __float_as_int(nodes[2 * i + 0].w), R = __float_as_int(nodes[2 * i + 1].w);
            const bool ldone = L >= leaf0 || visit[L] == 1, rdone = R >= leaf0 || visit[R] == 1;
            if (ldone && rdone) {
                const float4 a0 = nodes[2 * L], a1 = nodes[2 * L + 1], b0 = nodes[2 * R], b1 = nodes[2 * R + 1];
                nodes[2 * i + 0] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), __int_as_float(L));
                nodes[2 * i + 1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), __int_as_float(R));
                fitted[nf++] = i;   // marked after the barrier: a node fitted in THIS pass must not feed its parent before it
            }
        }
        __syncthreads();
        for (int k = 0; k < nf; ++k) visit[fitted[k]] = 1;
        __syncthreads();
        if (n < 2 || visit[0] == 1) break;   // the root is done (uniform: every thread reads the same word after the barrier)
    }
    for (int i = tid; i < n; i += kBuildThreads) {
        int dd = 0;
        for (int q = n > 1 ? parent[leaf0 + i] : -1; q >= 0; q = parent[q]) ++dd;
        atomicMax(&s_depth, dd);
    }
    __syncthreads();
    // ---- the structure the render kernel walks --------------------------------------------------------------------
    // Subtrees of at most kLeafTris triangles become leaves (their triangles are neighbours in Morton order); every node above
    // them gets one record with BOTH children's boxes, so that a step of the walk reads 64 contiguous bytes.
    for (int i = tid; i < n - 1; i += kBuildThreads) rec_of[i] = count_of[i] > kLeafTris ? (i == 0 ? 0 : atomicAdd(&s_nrec, 1)) : -1;
    __syncthreads();
    // grid of the compact form: cells 1 .. 65534 span the padded scene box
    float glo[3], gstep[3];
    for (int a = 0; a < 3; ++a) {
        glo[a] = blo[a] - pad;
        gstep[a] = (ext[a] + 2.0f * pad) * (1.0f / 65533.0f);
    }
    auto link_of = [&](int child) -> int {
        if (child >= leaf0) return -1 - (child - leaf0);                                         // one triangle
        if (count_of[child] <= kLeafTris) return -1 - (first_of[child] | ((count_of[child] - 1) << kLeafShift));
        return rec_of[child];
    };
    for (int i = tid; i < n - 1; i += kBuildThreads) {
        const int r = rec_of[i];
        if (r < 0) continue;
        const int L = __float_as_int(nodes[2 * i + 0].w), R = __float_as_int(nodes[2 * i + 1].w);
        const float4 a0 = nodes[2 * L], a1 = nodes[2 * L + 1], b0 = nodes[2 * R], b1 = nodes[2 * R + 1];
        recs[4 * r + 0] = make_float4(a0.x, a0.y, a0.z, __int_as_float(link_of(L)));
        recs[4 * r + 1] = make_float4(a1.x, a1.y, a1.z, 0.0f);
        recs[4 * r + 2] = make_float4(b0.x, b0.y, b0.z, __int_as_float(link_of(R)));
        recs[4 * r + 3] = make_float4(b1.x, b1.y, b1.z, 0.0f);
        // the same record on the 16-bit grid over the padded scene box: lower planes one cell below their floor, upper planes one
        // above their ceiling
        auto cell = [&](float v, int a, bool up) -> unsigned int {
            const float c = (v - glo[a]) / gstep[a];
            const float q = up ? ceilf(c) + 1.0f : floorf(c) - 1.0f;
            return (unsigned int)fminf(fmaxf(q, 0.0f), 65535.0f);
        };
        qrecs[2 * r + 0] = make_uint4(cell(a0.x, 0, false) | (cell(a1.x, 0, true) << 16), cell(a0.y, 1, false) | (cell(a1.y, 1, true) << 16),
                                      cell(a0.z, 2, false) | (cell(a1.z, 2, true) << 16), (unsigned int)link_of(L));
        qrecs[2 * r + 1] = make_uint4(cell(b0.x, 0, false) | (cell(b1.x, 0, true) << 16), cell(b0.y, 1, false) | (cell(b1.y, 1, true) << 16),
                                      cell(b0.z, 2, false) | (cell(b1.z, 2, true) << 16), (unsigned int)link_of(R));
        int dd = 1;   // stack entries a walk can hold below this record: one per record on the way down, its own included
        for (int q = parent[i]; q >= 0; q = parent[q]) ++dd;
        atomicMax(&s_wdepth, dd);
    }
    for (int i = tid; i < n; i += kBuildThreads) {
        const unsigned int tri = (unsigned int)(s_keys[i] & 0xFFFFFFFFu);
        const unsigned int i0 = indices[3 * tri + 0], i1 = indices[3 * tri + 1], i2 = indices[3 * tri + 2];
        tris[3 * i + 0] = make_float4(positions[3 * i0 + 0], positions[3 * i0 + 1], positions[3 * i0 + 2], __int_as_float((int)tri));
        tris[3 * i + 1] = make_float4(positions[3 * i1 + 0], positions[3 * i1 + 1], positions[3 * i1 + 2], 0.0f);
        tris[3 * i + 2] = make_float4(positions[3 * i2 + 0], positions[3 * i2 + 1], positions[3 * i2 + 2], 0.0f);
        tidx[i] = make_uint2((i0 & 0xFFFFu) | (i1 << 16), (i2 & 0xFFFFu) | (tri << 16));   // (used only when every vertex index fits 16 bits)
    }
    __syncthreads();
    if (tid == 0) {
        out_meta[0] = s_depth;
        out_meta[1] = n > kLeafTris ? s_nrec : 0;   // 0: the mesh is one leaf, no records
        out_meta[2] = s_wdepth;
        for (int a = 0; a < 3; ++a) {
            out_meta[3 + a] = __float_as_int(glo[a]);
            out_meta[6 + a] = __float_as_int(gstep[a]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The walk's records rebuilt over the same leaves with the surface-area heuristic (second session of round 2).  build_kernel's
// Morton hierarchy forms the leaves (subtrees of at most kLeafTris triangles: neighbours in Morton order, contiguous in `tris`);
// above them bit prefixes know nothing of box areas, and any tree over the same leaves returns the same hits, so the topology is
// rebuilt top-down like the analytic path's (rtgo_device.h, build_kernel): every node is split where
// A(left) * T(left) + A(right) * T(right) is smallest over the three axes and every position of its leaves sorted by centroid
// (T = triangles).  One workgroup: all threads walk one task queue together -- a rank sort per axis in parallel, the sweep by one
// thread.  What sets the whitted launch is one wave's serial walk over the finely tessellated part: fewer steps there.
// Overwrites recs / qrecs and out_meta[1] (records), [2] (stack entries); scratch: 32 n ints.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBuildThreads) void sah_kernel(int n, const float4* __restrict__ nodes, const int* __restrict__ parent, const int* __restrict__ first_of,
                                                            const int* __restrict__ count_of, int* __restrict__ scratch, float4* __restrict__ recs,
                                                            uint4* __restrict__ qrecs, int* __restrict__ out_meta)
{
    if (n <= kLeafTris) return;   // the mesh is one leaf: no records
    extern __shared__ __attribute__((aligned(16))) unsigned char sah_dyn[];
    float (*u_box)[6] = reinterpret_cast<float (*)[6]>(sah_dyn);      // [n] leaf boxes
    int* u_link = reinterpret_cast<int*>(u_box + n);                  // [n] the leaf's link in the walk's records
    short* perm = reinterpret_cast<short*>(u_link + n);               // [n] leaves in the current task order
    short* tmp = perm + n;                                            // [n]
    unsigned char* u_w = reinterpret_cast<unsigned char*>(tmp + n);   // [n] triangles of the leaf
    __shared__ int s_U, s_qtail, s_count, s_best_axis, s_best_pos, s_wdepth, s_nrec;
    // global scratch (one thread group, barriers order it)
    int* t_left = scratch;               // [2n] first child / the leaf's link
    int* t_right = t_left + 2 * n;       // [2n] second child, -1: a leaf
    int* t_parent = t_right + 2 * n;     // [2n]
    int* tq_node = t_parent + 2 * n;     // [2n] task queue
    int* tq_lo = tq_node + 2 * n;
    int* tq_hi = tq_lo + 2 * n;
    float* t_box = reinterpret_cast<float*>(tq_hi + 2 * n);   // [2n][6]
    int* start = reinterpret_cast<int*>(t_box + 12 * n);      // [n] node of the leaf that starts at a Morton position, -1: none
    int* rec_of = start + n;                                  // [2n]
    float* sfx = reinterpret_cast<float*>(rec_of + 2 * n);    // [n][2] suffix area and triangles of the sweep
    const int tid = threadIdx.x, leaf0 = n - 1;
    for (int pos = tid; pos < n; pos += kBuildThreads) start[pos] = -1;
    __syncthreads();
    for (int k = tid; k < 2 * n - 1; k += kBuildThreads) {
        const int cnt = k >= leaf0 ? 1 : count_of[k];
        const bool is_leaf = cnt <= kLeafTris && k != 0 && count_of[parent[k]] > kLeafTris;
        if (is_leaf) start[k >= leaf0 ? k - leaf0 : first_of[k]] = k;
    }
    __syncthreads();
    if (tid == 0) {
        int U = 0;
        for (int pos = 0; pos < n; ++pos)
            if (start[pos] >= 0) rec_of[U++] = start[pos];   // (rec_of borrowed: leaf u's node)
        s_U = U;
        s_wdepth = 0;
    }
    __syncthreads();
    const int U = s_U;
    for (int u = tid; u < U; u += kBuildThreads) {
        const int k = rec_of[u];
        const float4 b0 = nodes[2 * k], b1 = nodes[2 * k + 1];
        u_box[u][0] = b0.x; u_box[u][1] = b0.y; u_box[u][2] = b0.z;
        u_box[u][3] = b1.x; u_box[u][4] = b1.y; u_box[u][5] = b1.z;
        const int first = k >= leaf0 ? k - leaf0 : first_of[k], cnt = k >= leaf0 ? 1 : count_of[k];
        u_link[u] = -1 - (first | ((cnt - 1) << kLeafShift));
        u_w[u] = (unsigned char)cnt;
        perm[u] = (short)u;
    }
    if (tid == 0) {
        tq_node[0] = 0; tq_lo[0] = 0; tq_hi[0] = U;
        s_qtail = 1;
        s_count = 1;
        t_parent[0] = -1;
    }
    __syncthreads();
    for (int qi = 0; qi < 2 * n; ++qi) {
        __syncthreads();
        if (qi >= s_qtail) break;   // (uniform: every thread reads the same word after the barrier)
        const int lo = tq_lo[qi], hi = tq_hi[qi], node = tq_node[qi], m = hi - lo;
        if (m == 1) {
            if (tid == 0) {
                const int u = perm[lo];
                for (int c = 0; c < 6; ++c) t_box[6 * node + c] = u_box[u][c];
                t_left[node] = u_link[u];
                t_right[node] = -1;
            }
            continue;
        }
        if (tid == 0) {
            s_best_axis = -1;
            s_best_pos = m / 2;
        }
        float best_cost = INFINITY;   // (thread 0's)
        for (int pass = 0; pass < 4; ++pass) {
            // passes 0..2: try axis `pass`; pass 3: put the range back in the order of the best axis
            __syncthreads();
            const int axis = pass < 3 ? pass : s_best_axis;
            if (pass == 3 && (axis < 0 || axis == 2)) break;   // (uniform) no finite cost at all, or already in z order
            for (int e = tid; e < m; e += kBuildThreads) {
                const int me = perm[lo + e];
                const float key = u_box[me][axis] + u_box[me][3 + axis];
                int rank = 0;
                for (int j = 0; j < m; ++j) {
                    const int other = perm[lo + j];
                    const float kj = u_box[other][axis] + u_box[other][3 + axis];
                    rank += (kj < key || (kj == key && other < me)) ? 1 : 0;
                }
                tmp[lo + rank] = (short)me;
            }
            __syncthreads();
            for (int e = tid; e < m; e += kBuildThreads) perm[lo + e] = tmp[lo + e];
            __syncthreads();
            if (pass < 3 && tid == 0) {
                float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int w = 0;
                for (int j = m - 1; j >= 1; --j) {
                    const int u = perm[lo + j];
                    for (int c = 0; c < 3; ++c) {
                        b[c] = fminf(b[c], u_box[u][c]);
                        b[3 + c] = fmaxf(b[3 + c], u_box[u][3 + c]);
                    }
                    w += u_w[u];
                    const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
                    sfx[2 * j + 0] = ex * ey + ey * ez + ez * ex;
                    sfx[2 * j + 1] = (float)w;
                }
                float a[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int wl = 0;
                for (int j = 1; j < m; ++j) {   // left = [0, j), right = [j, m)
                    const int u = perm[lo + j - 1];
                    for (int c = 0; c < 3; ++c) {
                        a[c] = fminf(a[c], u_box[u][c]);
                        a[3 + c] = fmaxf(a[3 + c], u_box[u][3 + c]);
                    }
                    wl += u_w[u];
                    const float ex = a[3] - a[0], ey = a[4] - a[1], ez = a[5] - a[2];
                    const float cost = (ex * ey + ey * ez + ez * ex) * (float)wl + sfx[2 * j] * sfx[2 * j + 1];
                    // ties go to the split nearer the median: a range of leaves with one and the same box (coincident or duplicated
                    // triangles) ties at every position, and "first wins" would peel one leaf per level -- a chain as deep as the range
                    const int dj = j > m / 2 ? j - m / 2 : m / 2 - j, db = s_best_pos > m / 2 ? s_best_pos - m / 2 : m / 2 - s_best_pos;
                    if (cost < best_cost || (cost == best_cost && dj < db)) {
                        best_cost = cost;
                        s_best_axis = pass;
                        s_best_pos = j;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int j = lo; j < hi; ++j) {
                const int u = perm[j];
                for (int c = 0; c < 3; ++c) {
                    b[c] = fminf(b[c], u_box[u][c]);
                    b[3 + c] = fmaxf(b[3 + c], u_box[u][3 + c]);
                }
            }
            const int cl = s_count, cr = s_count + 1;
            s_count += 2;
            for (int c = 0; c < 6; ++c) t_box[6 * node + c] = b[c];
            t_left[node] = cl;
            t_right[node] = cr;
            t_parent[cl] = node;
            t_parent[cr] = node;
            const int mid = lo + s_best_pos;
            const int t = s_qtail;
            tq_node[t] = cl; tq_lo[t] = lo; tq_hi[t] = mid;
            tq_node[t + 1] = cr; tq_lo[t + 1] = mid; tq_hi[t + 1] = hi;
            s_qtail = t + 2;
        }
    }
    __syncthreads();
    const int n_nodes = s_count;
    if (tid == 0) {
        int r = 0;
        for (int k = 0; k < n_nodes; ++k) rec_of[k] = t_right[k] >= 0 ? r++ : -1;   // (the root is node 0 and record 0)
        s_nrec = r;
    }
    __syncthreads();
    float glo[3], gstep[3];
    for (int a = 0; a < 3; ++a) {
        glo[a] = __int_as_float(out_meta[3 + a]);
        gstep[a] = __int_as_float(out_meta[6 + a]);
    }
    for (int k = tid; k < n_nodes; k += kBuildThreads) {
        const int r = rec_of[k];
        if (r < 0) continue;
        const int L = t_left[k], R = t_right[k];
        const int linkL = t_right[L] >= 0 ? rec_of[L] : t_left[L], linkR = t_right[R] >= 0 ? rec_of[R] : t_left[R];
        const float* a = t_box + 6 * L;
        const float* b = t_box + 6 * R;
        recs[4 * r + 0] = make_float4(a[0], a[1], a[2], __int_as_float(linkL));
        recs[4 * r + 1] = make_float4(a[3], a[4], a[5], 0.0f);
        recs[4 * r + 2] = make_float4(b[0], b[1], b[2], __int_as_float(linkR));
        recs[4 * r + 3] = make_float4(b[3], b[4], b[5], 0.0f);
        auto cell = [&](float v, int ax, bool up) -> unsigned int {
            const float c = (v - glo[ax]) / gstep[ax];
            const float q = up ? ceilf(c) + 1.0f : floorf(c) - 1.0f;
            return (unsigned int)fminf(fmaxf(q, 0.0f), 65535.0f);
        };
        qrecs[2 * r + 0] = make_uint4(cell(a[0], 0, false) | (cell(a[3], 0, true) << 16), cell(a[1], 1, false) | (cell(a[4], 1, true) << 16),
                                      cell(a[2], 2, false) | (cell(a[5], 2, true) << 16), (unsigned int)linkL);
        qrecs[2 * r + 1] = make_uint4(cell(b[0], 0, false) | (cell(b[3], 0, true) << 16), cell(b[1], 1, false) | (cell(b[4], 1, true) << 16),
                                      cell(b[2], 2, false) | (cell(b[5], 2, true) << 16), (unsigned int)linkR);
        int dd = 1;   // stack entries a walk can hold below this record: one per record on the way down, its own included
        for (int q = t_parent[k]; q >= 0; q = t_parent[q]) ++dd;
        atomicMax(&s_wdepth, dd);
    }
    __syncthreads();
    if (tid == 0) {
        out_meta[1] = s_nrec;
        out_meta[2] = s_wdepth;
    }
}

}  // namespace whitted
}  // namespace rtgo
