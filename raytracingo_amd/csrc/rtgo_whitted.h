// rtgo_whitted.h -- gfx950 device code of the "whitted" triangle path: the programs of the reference's cuda/whitted.cu
// (__raygen__pinhole :183-240, __miss__constant_radiance :243-246, __closesthit__occlusion :249-252, __closesthit__radiance
// :255-337 with cuda/LocalGeometry.h:55-141) over triangle meshes, without textures.  The reference hands triangle
// intersection and the acceleration structure to OptiX (built-in triangles, optixAccelBuild in sutil/Scene.cpp); here both are
// written out: a Moeller-Trumbore test with a fixed operation order (shared with the oracle, so the two agree bit for bit) and
// an LBVH over the triangles built on the device by one workgroup (Morton codes -> bitonic sort in LDS -> Karras hierarchy ->
// bottom-up fit), kept in HBM / L2 and walked with a per-thread stack in LDS.  One thread per pixel, one launch per subframe:
// a primary ray plus one shadow ray per point light, no recursion (whitted.cu traces none either).
//
// Compiled with -ffp-contract=off like the rest of the library: one IEEE rounding per operation.
#pragma once

#include "rtgo_device.h"

namespace rtgo {
namespace whitted {

constexpr int kMaxTriangles = 4096;   // one workgroup sorts the Morton keys in LDS (8 B per key)
constexpr int kBuildThreads = 1024;
constexpr int kStack = 32;            // per-thread traversal stack entries (LBVH depth is checked against it at build)
constexpr int kBlock = 256;

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

struct Params {       // whitted::LaunchParams, cuda/whitted.h:59-74
    const float4* nodes;        // LBVH: 2 float4 per node, (bmin, left), (bmax, right); internal [0, n-2], leaves [n-1, 2n-2]: left = triangle, right = -1
    const float* positions;     // 3 floats per vertex
    const float* normals;       // 3 floats per vertex, or null (then N = Ng, LocalGeometry.h:113-116)
    const unsigned int* indices;    // 3 per triangle
    const unsigned int* tri_material;
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
    const float inv = 1.0f / det;
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

// closest hit (ANY = false: smallest t, lowest triangle index on ties) or any hit (ANY = true: the occlusion ray's
// OPTIX_RAY_FLAG_TERMINATE_ON_FIRST_HIT, whitted.cu:140-151) over the triangle LBVH
template <bool ANY>
__device__ __forceinline__ bool trace(const Params& p, unsigned int* __restrict__ s_stack, v3 o, v3 d, float tmin, float tmax, int& tri_out,
                                      float& t_out, float& u_out, float& v_out)
{
    auto safe_inv = [](float x) { return fabsf(x) < 1e-30f ? copysignf(1e30f, x) : 1.0f / x; };
    const v3 id = mk(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    int best = -1;
    float bt = tmax, bu = 0.0f, bv = 0.0f;
    int sp = 0;
    int node = 0;
    if (p.n_triangles == 1) node = 0;   // a single leaf at index n - 1 = 0
    float tn;
    {
        const float4 q0 = p.nodes[0], q1 = p.nodes[1];
        if (!node_hit(q0, q1, o, id, tmin, bt, tn)) return false;
    }
    for (;;) {
        const float4 q0 = p.nodes[2 * node], q1 = p.nodes[2 * node + 1];
        const int left = __float_as_int(q0.w), right = __float_as_int(q1.w);
        bool popped = false;
        if (right < 0) {
            const unsigned int i0 = p.indices[3 * left + 0], i1 = p.indices[3 * left + 1], i2 = p.indices[3 * left + 2];
            float t, u, v;
            if (tri_intersect(ld3(p.positions, i0), ld3(p.positions, i1), ld3(p.positions, i2), o, d, tmin, tmax, t, u, v) &&
                (t < bt || (t == bt && best >= 0 && left < best))) {
                bt = t;
                bu = u;
                bv = v;
                best = left;
                if (ANY) break;
            }
            popped = true;
        } else {
            const float4 l0 = p.nodes[2 * left], l1 = p.nodes[2 * left + 1];
            const float4 h0 = p.nodes[2 * right], h1 = p.nodes[2 * right + 1];
            float tl, tr;
            const bool hl = node_hit(l0, l1, o, id, tmin, bt, tl);
            const bool hr = node_hit(h0, h1, o, id, tmin, bt, tr);
            const bool go_r = hr & (!hl | (tr < tl));
            if (hl & hr) {
                s_stack[sp * kBlock] = (unsigned int)(go_r ? left : right);
                ++sp;
            }
            if (hl | hr) node = go_r ? right : left;
            else popped = true;
        }
        if (popped) {
            if (sp == 0) break;
            --sp;
            node = (int)s_stack[sp * kBlock];
        }
    }
    tri_out = best;
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
    const float g0 = NdotL * sqrtf(NdotV * NdotV * (1.0f - a2) + a2);
    const float g1 = NdotV * sqrtf(NdotL * NdotL * (1.0f - a2) + a2);
    return 2.0f * NdotL * NdotV / (g0 + g1);
}
__device__ __forceinline__ float ggx_normal(float NdotH, float alpha)
{
    const float a2 = alpha * alpha;
    const float n2 = NdotH * NdotH;
    const float x = n2 * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * x * x);
}

__global__ __launch_bounds__(kBlock) void render_kernel(const Params p)
{
    __shared__ unsigned int s_stack_all[kStack * kBlock];
    unsigned int* s_stack = s_stack_all + threadIdx.x;
    const unsigned int idx = blockIdx.x * kBlock + threadIdx.x;
    unsigned int rays = 0, occl = 0;
    if (idx < p.width * p.height) {
        const unsigned int y = idx / p.width, x = idx - y * p.width;
        // __raygen__pinhole, whitted.cu:183-240
        unsigned int seed = tea4(y * p.width + x, p.subframe);
        float jx = 0.0f, jy = 0.0f;
        if (p.subframe != 0) {
            jx = rnd(seed) - 0.5f;   // x first (source order)
            jy = rnd(seed) - 0.5f;
        }
        const float dx = 2.0f * (((float)x + jx) / (float)p.width) - 1.0f;
        const float dy = 2.0f * (((float)y + jy) / (float)p.height) - 1.0f;
        const v3 rd = vnormalize(vadd(vadd(vscale(p.U, dx), vscale(p.V, dy)), p.W));
        const v3 ro = p.eye;
        v3 result = p.miss;   // __miss__constant_radiance, :243-246
        int tri;
        float t, bu, bv;
        rays += 1;
        if (trace<false>(p, s_stack, ro, rd, 0.01f, 1e16f, tri, t, bu, bv)) {
            // __closesthit__radiance, :255-337, with getLocalGeometry (LocalGeometry.h:55-141) for a mesh in world space
            const unsigned int i0 = p.indices[3 * tri + 0], i1 = p.indices[3 * tri + 1], i2 = p.indices[3 * tri + 2];
            const v3 P0 = ld3(p.positions, i0), P1 = ld3(p.positions, i1), P2 = ld3(p.positions, i2);
            const float w0 = 1.0f - bu - bv;
            const v3 P = vadd(vadd(vscale(P0, w0), vscale(P1, bu)), vscale(P2, bv));
            const v3 Ng = vnormalize(vcross(vsub(P1, P0), vsub(P2, P0)));
            v3 N = Ng;
            if (p.normals) {
                const v3 N0 = ld3(p.normals, i0), N1 = ld3(p.normals, i1), N2 = ld3(p.normals, i2);
                N = vnormalize(vadd(vadd(vscale(N0, w0), vscale(N1, bu)), vscale(N2, bv)));
            }
            const Pbr m = p.materials[p.tri_material ? p.tri_material[tri] : 0u];
            const v3 base = mk(m.base_color[0], m.base_color[1], m.base_color[2]);
            const float metallic = m.metallic * 1.0f, roughness = m.roughness * 1.0f;   // (x the (1,1,1,1) of an absent texture, :270-275)
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
                    int ot;
                    float tt, uu, vv;
                    rays += 1;
                    occl += 1;
                    if (!trace<true>(p, s_stack, P, Lv, 0.001f, Ldist - 0.001f, ot, tt, uu, vv)) {
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
    rays = wave_sum(rays);
    occl = wave_sum(occl);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&p.counters[0], (unsigned long long)rays);
        atomicAdd(&p.counters[1], (unsigned long long)occl);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LBVH over the triangles, one workgroup: replaces optixAccelBuild over OPTIX_BUILD_INPUT_TYPE_TRIANGLES (sutil/Scene.cpp,
// buildMeshAccels).  Same recipe as the analytic path's canonical tree: 30-bit Morton code of the centroid normalised to the
// scene bounds, stable order by (code, triangle index), Karras 2012, one triangle per leaf, bottom-up fit.
// Boxes are padded by 1e-4 of the scene's extent + 1e-6: the slab test rounds, the triangle test must never be cut off.
// out_meta = {depth}.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBuildThreads) void build_kernel(const float* __restrict__ positions, const unsigned int* __restrict__ indices, int n,
                                                              float4* __restrict__ nodes, int* __restrict__ parent, int* __restrict__ visit,
                                                              int* __restrict__ out_meta)
{
    __shared__ unsigned long long s_keys[kMaxTriangles];
    __shared__ float s_red[6][kBuildThreads];
    __shared__ int s_depth;
    const int tid = threadIdx.x;
    if (tid == 0) s_depth = 0;
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
    if (tid == 0) out_meta[0] = s_depth;
}

}  // namespace whitted
}  // namespace rtgo
