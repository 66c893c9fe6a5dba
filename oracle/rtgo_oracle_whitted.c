/*
 * rtgo_oracle_whitted.c -- CPU ORACLE of the "whitted" triangle path.  TEST INFRASTRUCTURE ONLY (see rtgo_oracle.h).
 *
 * Plain-C restatement of the reference's cuda/whitted.cu -- __raygen__pinhole (:183-240), __miss__constant_radiance
 * (:243-246), __closesthit__occlusion (:249-252), __closesthit__radiance (:255-337), the GGX helpers (:49-89), make_color
 * (:164-173) -- of getLocalGeometry for triangle meshes (cuda/LocalGeometry.h:55-141), and of tex2D<float4> on the texture objects
 * sutil::Scene::addSampler makes (sutil/Scene.cpp:505-538; the filtering arithmetic is the CUDA programming guide's, "Texture Fetching").
 *
 * PINNING STATUS: tea<4> / rnd are held to the reference's own cuda/random.h through oracle/_ref (tests/golden/ref_blocks.json).
 * Everything else here is PARITY UNPINNED: whitted.cu includes <optix.h> (absent from this image), no program of the reference
 * ever launches it (engine/ never instantiates sutil::Scene), and the reference holds no fixture for it.  Triangle
 * intersection itself is OptiX's built-in (closed): the Moeller-Trumbore statement below is this project's definition, shared
 * operation for operation with the device code (raytracingo_amd/csrc/rtgo_whitted.h).  Traversal is brute force in triangle
 * order: the closest hit is the smallest t, the lowest triangle index on ties.
 */
#include "rtgo_oracle.h"

#include <math.h>
#include <string.h>

typedef struct { float x, y, z; } w3;
static inline w3 W3(float x, float y, float z) { w3 r = { x, y, z }; return r; }
static inline w3 wadd(w3 a, w3 b) { return W3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline w3 wsub(w3 a, w3 b) { return W3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline w3 wmul(w3 a, w3 b) { return W3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline w3 wscale(w3 a, float s) { return W3(a.x * s, a.y * s, a.z * s); }
static inline w3 wneg(w3 a) { return W3(-a.x, -a.y, -a.z); }
static inline float wdot(w3 a, w3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                 /* vec_math.h:523-526 */
static inline w3 wcross(w3 a, w3 b) { return W3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } /* :529-532 */
static inline float wlength(w3 v) { return sqrtf(wdot(v, v)); }                                    /* :535-538 */
static inline w3 wnormalize(w3 v) { float inv = 1.0f / sqrtf(wdot(v, v)); return wscale(v, inv); } /* :541-545 */
static inline float wclamp(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }            /* :115-118 */
static inline w3 wld(const float* p, uint32_t i) { return W3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

/* tea<4>, cuda/random.h:30-45 */
uint32_t oracle_tea4(uint32_t v0, uint32_t v1)
{
    uint32_t s0 = 0;
    for (int n = 0; n < 4; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

/* Moeller-Trumbore, two-sided; hit iff tmin < t < tmax.  p0..p2, o, d: 3 floats each. */
int oracle_tri_intersect(const float* p0, const float* p1, const float* p2, const float* o, const float* d, float tmin, float tmax,
                         float* t_out, float* u_out, float* v_out)
{
    const w3 P0 = wld(p0, 0), P1 = wld(p1, 0), P2 = wld(p2, 0), O = wld(o, 0), D = wld(d, 0);
    const w3 e1 = wsub(P1, P0), e2 = wsub(P2, P0);
    const w3 pv = wcross(D, e2);
    const float det = wdot(e1, pv);
    if (det == 0.0f) return 0;
    const float inv = 1.0f / det;
    const w3 tv = wsub(O, P0);
    const float u = wdot(tv, pv) * inv;
    if (u < 0.0f || u > 1.0f) return 0;
    const w3 qv = wcross(tv, e1);
    const float v = wdot(D, qv) * inv;
    if (v < 0.0f || u + v > 1.0f) return 0;
    const float t = wdot(e2, qv) * inv;
    if (!(t > tmin && t < tmax)) return 0;
    *t_out = t;
    *u_out = u;
    *v_out = v;
    return 1;
}

/* tex2D<float4>( tex, u, v ): cudaReadModeNormalizedFloat, normalizedCoords, and -- addSampler compares its CUDA enum arguments with GL
   constants (Scene.cpp:517-524), so always -- cudaAddressModeWrap and cudaFilterModeLinear.  CUDA programming guide, linear filtering:
   xB = u N - 0.5, i = floor(xB), alpha = frac(xB) in 1.8 fixed point (rounded to nearest here: the hardware's rounding is not published,
   PARITY UNPINNED), tex = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1], indices wrapped.  Operation order
   shared with the device code (rtgo_whitted.h, tex2d). */
void oracle_tex2d(const oracle_tex* t, float u, float v, float* rgba)
{
    const float xb = u * (float)t->w - 0.5f, yb = v * (float)t->h - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    const float a = floorf((xb - fx) * 256.0f + 0.5f) * (1.0f / 256.0f), b = floorf((yb - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
    const int w = (int)t->w, h = (int)t->h;
    int i0 = (int)fx % w, j0 = (int)fy % h;
    i0 += i0 < 0 ? w : 0;
    j0 += j0 < 0 ? h : 0;
    const int i1 = i0 + 1 == w ? 0 : i0 + 1, j1 = j0 + 1 == h ? 0 : j0 + 1;
    const uint8_t* t00 = t->px + 4 * ((size_t)j0 * t->w + i0);
    const uint8_t* t10 = t->px + 4 * ((size_t)j0 * t->w + i1);
    const uint8_t* t01 = t->px + 4 * ((size_t)j1 * t->w + i0);
    const uint8_t* t11 = t->px + 4 * ((size_t)j1 * t->w + i1);
    const float k = 1.0f / 255.0f;
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    for (int c = 0; c < 4; ++c)
        rgba[c] = w00 * ((float)t00[c] * k) + w10 * ((float)t10[c] * k) + w01 * ((float)t01[c] * k) + w11 * ((float)t11[c] * k);
}

static int trace(const oracle_whitted_scene* s, w3 o, w3 d, float tmin, float tmax, int any, int* tri, float* t, float* u, float* v)
{
    const float O[3] = { o.x, o.y, o.z }, D[3] = { d.x, d.y, d.z };
    int best = -1;
    float bt = tmax, bu = 0.0f, bv = 0.0f;
    for (uint32_t i = 0; i < s->n_triangles; ++i) {
        float tt, uu, vv;
        const uint32_t* ix = s->indices + 3 * i;
        if (oracle_tri_intersect(s->positions + 3 * ix[0], s->positions + 3 * ix[1], s->positions + 3 * ix[2], O, D, tmin, tmax, &tt, &uu, &vv) && tt < bt) {
            bt = tt;
            bu = uu;
            bv = vv;
            best = (int)i;
            if (any) break;
        }
    }
    *tri = best;
    *t = bt;
    *u = bu;
    *v = bv;
    return best >= 0;
}

/* whitted.cu:49-80 */
static w3 schlick(w3 spec, float VdotH)
{
    const float k = powf(1.0f - VdotH, 5.0f);
    return wadd(spec, wscale(wsub(W3(1.0f, 1.0f, 1.0f), spec), k));
}
static float vis(float NdotL, float NdotV, float alpha)
{
    const float a2 = alpha * alpha;
    const float g0 = NdotL * sqrtf(NdotV * NdotV * (1.0f - a2) + a2);
    const float g1 = NdotV * sqrtf(NdotL * NdotL * (1.0f - a2) + a2);
    return 2.0f * NdotL * NdotV / (g0 + g1);
}
static float ggx_normal(float NdotH, float alpha)
{
    const float a2 = alpha * alpha;
    const float n2 = NdotH * NdotH;
    const float x = n2 * (a2 - 1.0f) + 1.0f;
    return a2 / (3.14159265358979323846f * x * x);
}

/* one subframe: accum (float4 per pixel, read when subframe > 0) and image (uchar4) are updated in place; rays[0] += rays
   traced, rays[1] += occlusion rays among them */
int oracle_whitted_render(const oracle_whitted_scene* s, uint32_t width, uint32_t height, uint32_t subframe, float* accum, uint8_t* image,
                          uint64_t* rays, int threads)
{
    if (!s || !accum || !image || width == 0 || height == 0 || s->n_triangles == 0) return -1;
    uint64_t n_rays = 0, n_occl = 0;
    const w3 eye = wld(s->eye, 0), U = wld(s->U, 0), V = wld(s->V, 0), Wv = wld(s->W, 0);
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : n_rays, n_occl)
    for (int64_t yy = 0; yy < (int64_t)height; ++yy)
        for (uint32_t x = 0; x < width; ++x) {
            const uint32_t y = (uint32_t)yy, idx = y * width + x;
            /* __raygen__pinhole, whitted.cu:183-240 */
            uint32_t seed = oracle_tea4(y * width + x, subframe);
            float jx = 0.0f, jy = 0.0f;
            if (subframe != 0) {
                jx = oracle_rnd(&seed) - 0.5f; /* x first: source order */
                jy = oracle_rnd(&seed) - 0.5f;
            }
            const float dx = 2.0f * (((float)x + jx) / (float)width) - 1.0f;
            const float dy = 2.0f * (((float)y + jy) / (float)height) - 1.0f;
            const w3 rd = wnormalize(wadd(wadd(wscale(U, dx), wscale(V, dy)), Wv));
            w3 result = wld(s->miss, 0); /* __miss__constant_radiance */
            int tri;
            float t, bu, bv;
            n_rays += 1;
            if (trace(s, eye, rd, 0.01f, 1e16f, 0, &tri, &t, &bu, &bv)) {
                /* __closesthit__radiance (:255-337) + getLocalGeometry (LocalGeometry.h:55-141), mesh in world space */
                const uint32_t* ix = s->indices + 3 * (uint32_t)tri;
                const w3 P0 = wld(s->positions, ix[0]), P1 = wld(s->positions, ix[1]), P2 = wld(s->positions, ix[2]);
                const float w0 = 1.0f - bu - bv;
                const w3 P = wadd(wadd(wscale(P0, w0), wscale(P1, bu)), wscale(P2, bv));
                const w3 Ng = wnormalize(wcross(wsub(P1, P0), wsub(P2, P0)));
                w3 N = Ng;
                if (s->normals) {
                    const w3 N0 = wld(s->normals, ix[0]), N1 = wld(s->normals, ix[1]), N2 = wld(s->normals, ix[2]);
                    N = wnormalize(wadd(wadd(wscale(N0, w0), wscale(N1, bu)), wscale(N2, bv)));
                }
                const uint32_t mi = s->tri_material ? s->tri_material[tri] : 0u;
                const oracle_pbr* m = s->materials + mi;
                w3 base = W3(m->base_color[0], m->base_color[1], m->base_color[2]);
                float mr_y = 1.0f, mr_z = 1.0f; /* the (1,1,1,1) of an absent metallic-roughness texture, whitted.cu:271 */
                if (s->mat_tex) {
                    const oracle_mat_tex* mt = s->mat_tex + mi;
                    if (mt->base_color.px || mt->metallic_roughness.px || mt->normal.px) {
                        /* getLocalGeometry's UV and dp/du, dp/dv (LocalGeometry.h:88-135) */
                        float UV0[2] = { 0.0f, 0.0f }, UV1[2] = { 0.0f, 1.0f }, UV2[2] = { 1.0f, 0.0f }, UV[2] = { bu, bv };
                        if (s->texcoords) {
                            for (int k = 0; k < 2; ++k) {
                                UV0[k] = s->texcoords[2 * ix[0] + k];
                                UV1[k] = s->texcoords[2 * ix[1] + k];
                                UV2[k] = s->texcoords[2 * ix[2] + k];
                                UV[k] = w0 * UV0[k] + bu * UV1[k] + bv * UV2[k];
                            }
                        }
                        float tc[4];
                        if (mt->base_color.px) { /* base_color *= linearize( tex2D ), whitted.cu:78-85, 264-267 */
                            oracle_tex2d(&mt->base_color, UV[0], UV[1], tc);
                            base = wmul(base, W3(powf(tc[0], 2.2f), powf(tc[1], 2.2f), powf(tc[2], 2.2f)));
                        }
                        if (mt->metallic_roughness.px) { /* (occlusion, roughness, metallic), :272-276 */
                            oracle_tex2d(&mt->metallic_roughness, UV[0], UV[1], tc);
                            mr_y = tc[1];
                            mr_z = tc[2];
                        }
                        if (mt->normal.px) { /* whitted.cu:288-292 over LocalGeometry.h:118-134 */
                            const float du1 = UV0[0] - UV2[0], du2 = UV1[0] - UV2[0], dv1 = UV0[1] - UV2[1], dv2 = UV1[1] - UV2[1];
                            const w3 dp1 = wsub(P0, P2), dp2 = wsub(P1, P2);
                            const float det = du1 * dv2 - dv1 * du2;
                            const float invdet = 1.0f / det;
                            const w3 dpdu = wscale(wsub(wscale(dp1, dv2), wscale(dp2, dv1)), invdet);
                            const w3 dpdv = wscale(wadd(wscale(dp1, -du2), wscale(dp2, du1)), invdet);
                            oracle_tex2d(&mt->normal, UV[0], UV[1], tc);
                            const float nx = 2.0f * tc[0] - 1.0f, ny = 2.0f * tc[1] - 1.0f, nz = 2.0f * tc[2] - 1.0f;
                            N = wnormalize(wadd(wadd(wscale(wnormalize(dpdu), nx), wscale(wnormalize(dpdv), ny)), wscale(N, nz)));
                        }
                    }
                }
                const float metallic = m->metallic * mr_z, roughness = m->roughness * mr_y; /* :269-276 */
                const float F0 = 0.04f;
                const w3 diff_color = wscale(wscale(base, 1.0f - F0), 1.0f - metallic);
                const w3 spec_color = wadd(W3(F0, F0, F0), wscale(wsub(base, W3(F0, F0, F0)), metallic)); /* lerp, vec_math.h:496-499 */
                const float alpha = roughness * roughness;
                result = W3(0.0f, 0.0f, 0.0f);
                for (uint32_t l = 0; l < s->n_lights; ++l) {
                    const oracle_point_light* L = s->lights + l;
                    const w3 toL = wsub(W3(L->position[0], L->position[1], L->position[2]), P);
                    const float Ldist = wlength(toL);
                    const w3 Lv = wscale(toL, 1.0f / Ldist); /* float3 / float: vec_math.h:479-483 */
                    const w3 Vv = wneg(wnormalize(rd));
                    const w3 H = wnormalize(wadd(Lv, Vv));
                    const float NdotL = wdot(N, Lv), NdotV = wdot(N, Vv), NdotH = wdot(N, H), VdotH = wdot(Vv, H);
                    if (NdotL > 0.0f && NdotV > 0.0f) {
                        int ot;
                        float tt, uu, vv;
                        n_rays += 1;
                        n_occl += 1;
                        if (!trace(s, P, Lv, 0.001f, Ldist - 0.001f, 1, &ot, &tt, &uu, &vv)) {
                            const w3 F = schlick(spec_color, VdotH);
                            const float G = vis(NdotL, NdotV, alpha);
                            const float D = ggx_normal(NdotH, alpha);
                            const w3 diff = wscale(wmul(wsub(W3(1.0f, 1.0f, 1.0f), F), diff_color), 1.0f / 3.14159265358979323846f);
                            const w3 spec = wscale(wscale(F, G), D);
                            const w3 lc = wscale(W3(L->color[0], L->color[1], L->color[2]), L->intensity);
                            result = wadd(result, wmul(wscale(lc, NdotL), wadd(diff, spec)));
                        }
                    }
                }
            }
            /* whitted.cu:226-239 */
            w3 acc = result;
            if (subframe > 0) {
                const float a = 1.0f / (float)(subframe + 1);
                const w3 prev = W3(accum[4 * idx], accum[4 * idx + 1], accum[4 * idx + 2]);
                acc = wadd(prev, wscale(wsub(acc, prev), a));
            }
            accum[4 * idx + 0] = acc.x;
            accum[4 * idx + 1] = acc.y;
            accum[4 * idx + 2] = acc.z;
            accum[4 * idx + 3] = 1.0f;
            const float g = (float)(1.0 / 2.2f); /* make_color, :164-173 */
            image[4 * idx + 0] = (uint8_t)(powf(wclamp(acc.x, 0.0f, 1.0f), g) * 255.0f);
            image[4 * idx + 1] = (uint8_t)(powf(wclamp(acc.y, 0.0f, 1.0f), g) * 255.0f);
            image[4 * idx + 2] = (uint8_t)(powf(wclamp(acc.z, 0.0f, 1.0f), g) * 255.0f);
            image[4 * idx + 3] = 255u;
        }
    if (rays) {
        rays[0] += n_rays;
        rays[1] += n_occl;
    }
    return 0;
}
