/*
 * rtgo_oracle.c -- CPU ORACLE (test infrastructure, see rtgo_oracle.h).  Plain C restatement of the reference's
 * arithmetic; NOT part of the product.  Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math).
 *
 * Evaluation-order convention: wherever the reference calls rnd(seed) several times inside one
 * expression (kernel.cu:214-217, :492; scene.cpp:524,533,602-605) the draws are made in SOURCE ORDER, left to right,
 * in explicit statements.
 */
#include "rtgo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_F 3.14159265358979323846f /* M_PIf, vec_math.h:43 */

/* ------------------------------------------------------------------------------------------------ RNG */

uint32_t oracle_tea16(uint32_t v0, uint32_t v1)
{
    /* cuda/random.h:30-45, N = 16 */
    uint32_t s0 = 0;
    for (int n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

uint32_t oracle_lcg(uint32_t* prev)
{
    /* cuda/random.h:48-54 */
    *prev = 1664525u * (*prev) + 1013904223u;
    return *prev & 0x00FFFFFFu;
}

float oracle_rnd(uint32_t* prev)
{
    /* cuda/random.h:63-66 */
    return (float)oracle_lcg(prev) / (float)0x01000000;
}

/* ------------------------------------------------------------------------------------------------ vec3 */

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); } /* vec_math.h:455-461 */
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* vec_math.h:523-526 */
static inline v3 vcross(v3 a, v3 b) /* vec_math.h:529-532 */
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlength(v3 v) { return sqrtf(vdot(v, v)); } /* vec_math.h:535-538 */
static inline v3 vnormalize(v3 v) /* vec_math.h:541-545 */
{
    float invLen = 1.0f / sqrtf(vdot(v, v));
    return vscale(v, invLen);
}
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); } /* vec_math.h:115-118 */
static inline v3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

void oracle_normalize3(const float* v, float* out) { st3(out, vnormalize(ld3(v))); }

/* ------------------------------------------------------------------------------------------------ Matrix4x4 */

void oracle_mat_identity(float* m)
{
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}

void oracle_mat_mul(const float* a, const float* b, float* out)
{
    /* Matrix.h:344-360: sum starts at 0.0f and accumulates k = 0..3 in order */
    float tmp[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float sum = 0.0f;
            for (int k = 0; k < 4; ++k) sum += a[i * 4 + k] * b[k * 4 + j];
            tmp[i * 4 + j] = sum;
        }
    memcpy(out, tmp, sizeof tmp);
}

void oracle_mat_vec4(const float* m, const float* v, float* out)
{
    /* Matrix.h:472-493 */
    float t[4];
    for (int r = 0; r < 4; ++r) t[r] = m[4 * r + 0] * v[0] + m[4 * r + 1] * v[1] + m[4 * r + 2] * v[2] + m[4 * r + 3] * v[3];
    memcpy(out, t, sizeof t);
}

void oracle_mat_transpose(const float* m, float* out)
{
    /* Matrix.h:570-577 */
    float t[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) t[c * 4 + r] = m[r * 4 + c];
    memcpy(out, t, sizeof t);
}

/* the 24 signed products of Matrix.h:591-608, in source order; each product is ((a*b)*c)*d */
static const signed char DET_TERMS[24][5] = {
    { +1, 0, 5, 10, 15 }, { -1, 0, 5, 11, 14 }, { +1, 0, 9, 14, 7 },  { -1, 0, 9, 6, 15 },  { +1, 0, 13, 6, 11 },
    { -1, 0, 13, 10, 7 }, { -1, 4, 1, 10, 15 }, { +1, 4, 1, 11, 14 }, { -1, 4, 9, 14, 3 },  { +1, 4, 9, 2, 15 },
    { -1, 4, 13, 2, 11 }, { +1, 4, 13, 10, 3 }, { +1, 8, 1, 6, 15 },  { -1, 8, 1, 14, 7 },  { +1, 8, 5, 14, 3 },
    { -1, 8, 5, 2, 15 },  { +1, 8, 13, 2, 7 },  { -1, 8, 13, 6, 3 },  { -1, 12, 1, 6, 11 }, { +1, 12, 1, 10, 7 },
    { -1, 12, 5, 10, 3 }, { +1, 12, 5, 2, 11 }, { -1, 12, 9, 2, 7 },  { +1, 12, 9, 6, 3 },
};

float oracle_mat_det(const float* m)
{
    float d = 0.0f;
    for (int k = 0; k < 24; ++k) {
        const signed char* t = DET_TERMS[k];
        float p = m[t[1]] * m[t[2]] * m[t[3]] * m[t[4]];
        if (k == 0)
            d = p; /* the expression starts with the first product, not with 0 + */
        else if (t[0] > 0)
            d = d + p;
        else
            d = d - p;
    }
    return d;
}

/* Matrix.h:612-635: dst[k] = d * ( m[a]*(m[b]*m[c] - m[d]*m[e]) + ... three such groups ), in the order the
   reference assigns them (irrelevant for the values). */
static const unsigned char INV_ROWS[16][16] = {
    /* dst, then 3 x (a, b, c, d, e) */
    { 0, 5, 10, 15, 14, 11, 9, 14, 7, 6, 15, 13, 6, 11, 10, 7 },
    { 4, 6, 8, 15, 12, 11, 10, 12, 7, 4, 15, 14, 4, 11, 8, 7 },
    { 8, 7, 8, 13, 12, 9, 11, 12, 5, 4, 13, 15, 4, 9, 8, 5 },
    { 12, 4, 13, 10, 9, 14, 8, 5, 14, 13, 6, 12, 9, 6, 5, 10 },
    { 1, 9, 2, 15, 14, 3, 13, 10, 3, 2, 11, 1, 14, 11, 10, 15 },
    { 5, 10, 0, 15, 12, 3, 14, 8, 3, 0, 11, 2, 12, 11, 8, 15 },
    { 9, 11, 0, 13, 12, 1, 15, 8, 1, 0, 9, 3, 12, 9, 8, 13 },
    { 13, 8, 13, 2, 1, 14, 12, 1, 10, 9, 2, 0, 9, 14, 13, 10 },
    { 2, 13, 2, 7, 6, 3, 1, 6, 15, 14, 7, 5, 14, 3, 2, 15 },
    { 6, 14, 0, 7, 4, 3, 2, 4, 15, 12, 7, 6, 12, 3, 0, 15 },
    { 10, 15, 0, 5, 4, 1, 3, 4, 13, 12, 5, 7, 12, 1, 0, 13 },
    { 14, 12, 5, 2, 1, 6, 0, 13, 6, 5, 14, 4, 1, 14, 13, 2 },
    { 3, 1, 10, 7, 6, 11, 5, 2, 11, 10, 3, 9, 6, 3, 2, 7 },
    { 7, 2, 8, 7, 4, 11, 6, 0, 11, 8, 3, 10, 4, 3, 0, 7 },
    { 11, 3, 8, 5, 4, 9, 7, 0, 9, 8, 1, 11, 4, 1, 0, 5 },
    { 15, 0, 5, 10, 9, 6, 4, 9, 2, 1, 10, 8, 1, 6, 5, 2 },
};

void oracle_mat_inverse(const float* m, float* out)
{
    float dst[16];
    const float d = 1.0f / oracle_mat_det(m);
    for (int r = 0; r < 16; ++r) {
        const unsigned char* q = INV_ROWS[r];
        float g0 = m[q[1]] * (m[q[2]] * m[q[3]] - m[q[4]] * m[q[5]]);
        float g1 = m[q[6]] * (m[q[7]] * m[q[8]] - m[q[9]] * m[q[10]]);
        float g2 = m[q[11]] * (m[q[12]] * m[q[13]] - m[q[14]] * m[q[15]]);
        dst[q[0]] = d * (g0 + g1 + g2);
    }
    memcpy(out, dst, sizeof dst);
}

void oracle_mat_rotate(float radians, float ux, float uy, float uz, float* m)
{
    /* Matrix.h:640-672; the axis is NOT normalised (scene.cpp:262,359,362,410 pass non-unit axes) */
    float s = sinf(radians);
    float c = cosf(radians);
    oracle_mat_identity(m);
    m[0] = ux * ux + c * (1 - ux * ux);
    m[1] = ux * uy * (1 - c) - uz * s;
    m[2] = uz * ux * (1 - c) + uy * s;
    m[3] = 0;
    m[4] = ux * uy * (1 - c) + uz * s;
    m[5] = uy * uy + c * (1 - uy * uy);
    m[6] = uy * uz * (1 - c) - ux * s;
    m[7] = 0;
    m[8] = uz * ux * (1 - c) - uy * s;
    m[9] = uy * uz * (1 - c) + ux * s;
    m[10] = uz * uz + c * (1 - uz * uz);
    m[11] = 0;
    m[12] = 0;
    m[13] = 0;
    m[14] = 0;
    m[15] = 1;
}

void oracle_mat_translate(float x, float y, float z, float* m)
{
    /* Matrix.h:677-688 */
    oracle_mat_identity(m);
    m[3] = x;
    m[7] = y;
    m[11] = z;
}

void oracle_mat_scale(float x, float y, float z, float* m)
{
    /* Matrix.h:691-702 */
    oracle_mat_identity(m);
    m[0] = x;
    m[5] = y;
    m[10] = z;
}

/* ------------------------------------------------------------------------------------------------ camera / AABB / light */

void oracle_camera_uvw(const float* eye, const float* lookat, const float* up, float fovy, float aspect, float* Uo,
                       float* Vo, float* Wo)
{
    /* sutil/Camera.cpp:34-45 */
    v3 W = vsub(ld3(lookat), ld3(eye));
    float wlen = vlength(W);
    v3 U = vnormalize(vcross(W, ld3(up)));
    v3 V = vnormalize(vcross(U, W));
    float vlen = wlen * tanf(0.5f * fovy * PI_F / 180.0f);
    V = vscale(V, vlen);
    float ulen = vlen * aspect;
    U = vscale(U, ulen);
    st3(Uo, U);
    st3(Vo, V);
    st3(Wo, W);
}

void oracle_prim_aabb(const float* M, float* bb)
{
    /* primitive.cpp:20-79 (CubeBox) + :100-115 (GetAabb). std::min/std::max semantics: min(a,b) = b<a ? b : a */
    static const float F0[16] = { -1, -1, 1, 1, -1, -1, -1, -1, -1, 1, -1, 1, 1, 1, 1, 1 };
    static const float F1[16] = { -1, -1, 1, 1, 1, 1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1 };
    float f0[16], f1[16];
    oracle_mat_mul(M, F0, f0);
    oracle_mat_mul(M, F1, f1);
    float mn[3] = { 50.0f, 50.0f, 50.0f }, mx[3] = { -50.0f, -50.0f, -50.0f }; /* SCENE_MAX_BOUND */
    for (int i = 0; i < 4; ++i)
        for (int a = 0; a < 3; ++a) {
            float p0 = f0[i + 4 * a], p1 = f1[i + 4 * a];
            float t = (p0 < mn[a]) ? p0 : mn[a];
            mn[a] = (p1 < t) ? p1 : t;
            t = (mx[a] < p0) ? p0 : mx[a];
            mx[a] = (t < p1) ? p1 : t;
        }
    for (int a = 0; a < 3; ++a) {
        bb[a] = mn[a] - 0.001f; /* AABB_EPSILON */
        bb[3 + a] = mx[a] + 0.001f;
    }
}

void oracle_light_from_matrix(const float* M, const float* color, float falloff, oracle_light* out)
{
    /* light.cpp:9-28; normal = glm::normalize(glm::cross(v1, v2)) (glm func_geometric.inl:68-90) */
    static const float ORIGIN[4] = { -0.5f, 0.0f, 0.5f, 1.0f };
    static const float XAXIS[4] = { 1.0f, 0.0f, 0.0f, 0.0f };
    static const float ZAXIS[4] = { 0.0f, 0.0f, -1.0f, 0.0f };
    float c[4], a[4], b[4];
    oracle_mat_vec4(M, ORIGIN, c);
    oracle_mat_vec4(M, XAXIS, a);
    oracle_mat_vec4(M, ZAXIS, b);
    v3 v1 = ld3(a), v2 = ld3(b);
    /* glm::cross: (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y) */
    v3 cr = V3(v1.y * v2.z - v2.y * v1.z, v1.z * v2.x - v2.z * v1.x, v1.x * v2.y - v2.x * v1.y);
    float inv = 1.0f / sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z); /* glm::inversesqrt(dot(v,v)) */
    v3 n = vscale(cr, inv);
    st3(out->corner, ld3(c));
    st3(out->v1, v1);
    st3(out->v2, v2);
    st3(out->normal, n);
    st3(out->color, ld3(color));
    out->falloff = falloff;
}

/* ------------------------------------------------------------------------------------------------ intersection programs */

/* TransformRay, kernel.cu:125-135: o' = Minv*(o,1), d' = Minv*(d,0) through the 4-term operator* (Matrix.h:472-493) */
static inline void transform_ray(const float* inv, v3* o, v3* d)
{
    float ho[4] = { o->x, o->y, o->z, 1.0f }, hd[4] = { d->x, d->y, d->z, 0.0f }, ro[4], rd[4];
    oracle_mat_vec4(inv, hd, rd);
    oracle_mat_vec4(inv, ho, ro);
    *d = ld3(rd);
    *o = ld3(ro);
}

/* TransformNormal, kernel.cu:138-142: n = transpose(Minv) * (n,0) */
static inline v3 transform_normal(const float* inv, v3 n)
{
    float t[16], hn[4] = { n.x, n.y, n.z, 0.0f }, r[4];
    oracle_mat_transpose(inv, t);
    oracle_mat_vec4(t, hn, r);
    return ld3(r);
}

static int isect_sphere(const float* inv, v3 o, v3 d, float* tout, v3* nout)
{
    /* kernel.cu:250-287 */
    transform_ray(inv, &o, &d);
    const float a = vdot(d, d);
    const float b = 2.0f * vdot(d, o);
    const float c = vdot(o, o) - 1.0f; /* GENERIC_SPHERE_RADIUS */
    const float discr = b * b - 4.0f * a * c;
    if (discr > 0.0f) {
        const float sdiscr = sqrtf(discr);
        const float t = (-b - sdiscr) / (2.0f * a);
        v3 n = vnormalize(vadd(o, vscale(d, t)));
        n = transform_normal(inv, n);
        if (t > 0.0001f) {
            *tout = t;
            *nout = n;
            return 1;
        }
    }
    return 0;
}

static int cyl_tmin(v3 o, v3 d, float t0, float t1, float* out_t)
{
    /* GetTMinCylinder, kernel.cu:152-181 */
    float t = 1e16f;
    const float t_epsilon = 0.001f;
    int valid = 0;
    const float halfHeight = 2.0f / 2.0f; /* GENERIC_CYLINDER_HEIGHT / 2 */
    if (t0 > t_epsilon) {
        v3 p = vadd(o, vscale(d, t0));
        if (p.y > -halfHeight && p.y < halfHeight) {
            t = t0;
            valid = 1;
        }
    }
    if (t1 > t_epsilon && t1 < t) {
        v3 p = vadd(o, vscale(d, t1));
        if (p.y > -halfHeight && p.y < halfHeight) {
            t = t1;
            valid = 1;
        }
    }
    *out_t = t;
    return valid;
}

static int isect_cylinder(const float* inv, v3 o, v3 d, float* tout, v3* nout)
{
    /* kernel.cu:290-331 */
    transform_ray(inv, &o, &d);
    const float a = d.x * d.x + d.z * d.z;
    const float b = 2.0f * (o.x * d.x + o.z * d.z);
    const float c = o.x * o.x + o.z * o.z - 1.0f; /* GENERIC_CYLINDER_RADIUS */
    const float discr = b * b - 4.0f * a * c;
    if (discr > 0.001f) {
        const float sdiscr = sqrtf(discr);
        const float t0 = (-b + sdiscr) / (2.0f * a);
        const float t1 = (-b - sdiscr) / (2.0f * a);
        float t;
        if (cyl_tmin(o, d, t0, t1, &t)) {
            v3 p = vadd(o, vscale(d, t));
            v3 n = V3(p.x, 0.0f, p.z);
            *nout = transform_normal(inv, n);
            *tout = t;
            return 1;
        }
    }
    return 0;
}

static int isect_disk(const float* inv, v3 o, v3 d, float* tout, v3* nout)
{
    /* kernel.cu:334-369; equalFloat :146-149 */
    transform_ray(inv, &o, &d);
    v3 n = V3(0.0f, 1.0f, 0.0f);
    const float divisor = vdot(d, n);
    if (!(divisor > 0.0f - 0.01f && divisor < 0.0f + 0.01f)) {
        const float t = vdot(vneg(o), n) / divisor;
        if (t > 0.0001f) {
            v3 p = vadd(o, vscale(d, t));
            if (vdot(p, p) < 1.0f) { /* GENERIC_DISK_RADIUS */
                *nout = transform_normal(inv, n);
                *tout = t;
                return 1;
            }
        }
    }
    return 0;
}

static int isect_rectangle(const float* inv, v3 o, v3 d, float* tout, v3* nout)
{
    /* kernel.cu:372-416 */
    transform_ray(inv, &o, &d);
    const float width = 1.0f; /* GENERIC_RECTANGLE_WIDTH */
    const v3 p0 = V3(-width / 2.0f, 0.0f, width / 2.0f);
    const v3 a = V3(width, 0.0f, 0.0f);
    const v3 b = V3(0.0f, 0.0f, -width);
    v3 n = V3(0.0f, 1.0f, 0.0f);
    float divisor = vdot(d, n);
    if (divisor != 0.0f) {
        const float t = vdot(vsub(p0, o), n) / divisor;
        if (t > 0.0001f) {
            v3 p = vadd(o, vscale(d, t));
            v3 q = vsub(p, p0);
            if (0.0f < vdot(q, a) && vdot(q, a) < width && 0.0f < vdot(q, b) && vdot(q, b) < width) {
                if (vdot(n, d) < 0.0f) {
                    *nout = transform_normal(inv, n);
                    *tout = t;
                    return 1;
                }
            }
        }
    }
    return 0;
}

static inline int isect_dispatch(int type, const float* inv, v3 o, v3 d, float* t, v3* n)
{
    switch (type) {
    case ORACLE_CYLINDER: return isect_cylinder(inv, o, d, t, n);
    case ORACLE_DISK: return isect_disk(inv, o, d, t, n);
    case ORACLE_RECTANGLE: return isect_rectangle(inv, o, d, t, n);
    default: return isect_sphere(inv, o, d, t, n);
    }
}

int oracle_intersect(const oracle_prim* p, const float* o, const float* d, float* t, float* n)
{
    float inv[16];
    v3 nn = V3(0, 0, 0);
    oracle_mat_inverse(p->M, inv); /* modelMatrix.inverse() per call: kernel.cu:254,295,339,378 */
    int h = isect_dispatch(p->type, inv, ld3(o), ld3(d), t, &nn);
    if (h) st3(n, nn);
    return h;
}

/* ------------------------------------------------------------------------------------------------ hemisphere */

#define ORACLE_HEMISPHERE_MAX_TRIES 1024
static v3 hemisphere_x(v3 normal, v3 direction, float coefficient, uint32_t* seed, int double_trig)
{
    /* GetRayOnHemisphere, kernel.cu:101-122. sin/cos/acos/pow on float arguments are the float overloads in device code.
       ONE deliberate difference: the reference's rejection loop is unbounded, and it never ends when the lobe lies wholly below the horizon of
       `normal` -- which kernel.cu:443-447, 508-510 can produce: V = normalize(origin - x) is rounding noise when t is tiny against the
       coordinates, the flip of N then goes by noise, and Rr = reflect about that N points into the surface; a mirror's lobe around it never
       yields dot(normal, ray) >= 0 (tools/fuzz_farfield.py found the launch: random scene 45, 270 units off the origin -- a hung GPU).  The
       product stops after ORACLE_HEMISPHERE_MAX_TRIES draws and keeps the last one, and so does this restatement; wherever the reference's
       loop ends within that many draws (acceptance >= 1 % => all but 3e-5 of those) nothing changes. */
    v3 ray;
    int tries = 0;
    do {
        float r1 = oracle_rnd(seed);
        float r2 = oracle_rnd(seed);
        float phi = 2.f * PI_F * r1;
        float theta = acosf(powf((1.f - r2), 1.f / (coefficient + 1.f)));
        const v3 Y = vnormalize(direction);
        const v3 X = vnormalize(V3(Y.y - Y.z, -Y.x, Y.x));
        const v3 Z = vcross(Y, X);
        if (!double_trig)
            ray = vsub(vadd(vscale(X, sinf(theta) * cosf(phi)), vscale(Y, cosf(theta))), vscale(Z, sinf(theta) * sinf(phi)));
        else /* host-build artefact, see oracle_frame::host_double_trig */
            ray = vsub(vadd(vscale(X, (float)(sin((double)theta) * cos((double)phi))), vscale(Y, (float)cos((double)theta))),
                       vscale(Z, (float)(sin((double)theta) * sin((double)phi))));
    } while (vdot(normal, ray) < 0.f && ++tries < ORACLE_HEMISPHERE_MAX_TRIES);
    return ray;
}

void oracle_hemisphere(const float* normal, const float* direction, float coef, uint32_t* seed, float* out)
{
    st3(out, hemisphere_x(ld3(normal), ld3(direction), coef, seed, 0));
}

/* ------------------------------------------------------------------------------------------------ canonical LBVH */

static inline uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

/* delta(i,j) of Karras 2012 on keys made unique by their sorted position */
static inline int lbvh_delta(const uint32_t* code, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    uint32_t a = code[i], b = code[j];
    if (a == b) return 32 + clz32((uint32_t)i ^ (uint32_t)j);
    return clz32(a ^ b);
}

int oracle_lbvh_build(const float (*aabb)[6], int n, oracle_node* nodes, uint32_t* code, int32_t* order)
{
    /* 30-bit Morton code of each AABB centroid normalised to the scene bounds, stable sort by
       (code, primitive index), Karras-2012 hierarchy, one primitive per leaf, node boxes = union of reference AABBs. */
    float smin[3] = { INFINITY, INFINITY, INFINITY }, smax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            smin[a] = fminf(smin[a], aabb[i][a]);
            smax[a] = fmaxf(smax[a], aabb[i][3 + a]);
        }
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            float c = (aabb[i][a] + aabb[i][3 + a]) * 0.5f;
            float ext = smax[a] - smin[a];
            float u = ext > 0.0f ? (c - smin[a]) / ext : 0.0f;
            q[a] = (uint32_t)fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        }
        uint32_t m = (expand_bits(q[0]) << 2) | (expand_bits(q[1]) << 1) | expand_bits(q[2]);
        keys[i] = ((uint64_t)m << 32) | (uint32_t)i;
    }
    /* keys are unique -> any sort gives the (code, index) order; insertion sort keeps this dependency-free */
    for (int i = 1; i < n; ++i) {
        uint64_t k = keys[i];
        int j = i - 1;
        while (j >= 0 && keys[j] > k) {
            keys[j + 1] = keys[j];
            --j;
        }
        keys[j + 1] = k;
    }
    for (int i = 0; i < n; ++i) {
        code[i] = (uint32_t)(keys[i] >> 32);
        order[i] = (int32_t)(keys[i] & 0xFFFFFFFFu);
    }
    free(keys);

    const int leaf0 = n - 1;
    int* parent = (int*)malloc(sizeof(int) * (size_t)(2 * n));
    for (int i = 0; i < 2 * n - 1; ++i) parent[i] = -1;
    for (int j = 0; j < n; ++j) {
        oracle_node* L = &nodes[leaf0 + j];
        for (int a = 0; a < 3; ++a) {
            L->bmin[a] = aabb[order[j]][a];
            L->bmax[a] = aabb[order[j]][3 + a];
        }
        L->left = order[j];
        L->right = -1;
    }
    for (int i = 0; i < n - 1; ++i) {
        int d = (lbvh_delta(code, n, i, i + 1) - lbvh_delta(code, n, i, i - 1)) >= 0 ? 1 : -1;
        int dmin = lbvh_delta(code, n, i, i - d);
        int lmax = 2;
        while (lbvh_delta(code, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lbvh_delta(code, n, i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = lbvh_delta(code, n, i, j);
        int s = 0;
        int t = l;
        do {
            t = (t + 1) / 2; /* ceil(l / 2^k) */
            if (lbvh_delta(code, n, i, i + (s + t) * d) > dnode) s += t;
        } while (t > 1);
        int gamma = i + s * d + (d < 0 ? -1 : 0);
        int lo = i < j ? i : j, hi = i < j ? j : i;
        int left = (lo == gamma) ? leaf0 + gamma : gamma;
        int right = (hi == gamma + 1) ? leaf0 + gamma + 1 : gamma + 1;
        nodes[i].left = left;
        nodes[i].right = right;
        parent[left] = i;
        parent[right] = i;
    }
    /* bottom-up fit: min/max are exact, so the visiting order cannot change the result */
    if (n > 1) {
        int* visits = (int*)calloc((size_t)n, sizeof(int));
        for (int j = 0; j < n; ++j) {
            int p = parent[leaf0 + j];
            while (p >= 0) {
                if (++visits[p] < 2) break;
                const oracle_node *A = &nodes[nodes[p].left], *B = &nodes[nodes[p].right];
                for (int a = 0; a < 3; ++a) {
                    nodes[p].bmin[a] = fminf(A->bmin[a], B->bmin[a]);
                    nodes[p].bmax[a] = fmaxf(A->bmax[a], B->bmax[a]);
                }
                p = parent[p];
            }
        }
        free(visits);
    }
    free(parent);
    return 0; /* root: node 0 (for n == 1 node 0 is the single leaf since leaf0 == 0) */
}

/* ------------------------------------------------------------------------------------------------ tracing context */

typedef struct {
    const oracle_scene* sc;
    const oracle_frame* fr;
    const float (*inv)[16];   /* mode 1: hoisted inverses (bit-identical to the per-call inverse()) */
    const oracle_node* nodes; /* mode 1 */
    oracle_counters c;
} tctx;

typedef struct {
    float t;
    v3 n;
    int prim;
} hit_t;

/* box test shared (bit for bit) with the HIP kernel: slab test with IEEE 1/d, fminf/fmaxf NaN-dropping */
static inline int box_test(const oracle_node* nd, v3 o, v3 id, float tmin, float tmax, float* tn_out)
{
    float t0 = (nd->bmin[0] - o.x) * id.x, t1 = (nd->bmax[0] - o.x) * id.x;
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = (nd->bmin[1] - o.y) * id.y;
    t1 = (nd->bmax[1] - o.y) * id.y;
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    t0 = (nd->bmin[2] - o.z) * id.z;
    t1 = (nd->bmax[2] - o.z) * id.z;
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    tn = fmaxf(tn, tmin);
    tf = fminf(tf, tmax);
    *tn_out = tn;
    return tn <= tf;
}

/* optixTrace's traversal + optixReportIntersection acceptance: accepted iff tmin < t < current tmax,
   closest wins, ties keep the lower SBT index. returns 1 when something was hit. */
static int closest_hit(tctx* cx, v3 o, v3 d, float tmin, float tmax, hit_t* best)
{
    const oracle_scene* sc = cx->sc;
    best->prim = -1;
    best->t = tmax;
    if (cx->fr->mode == 0) {
        for (int i = 0; i < sc->n_prims; ++i) {
            float inv[16], t;
            v3 n;
            oracle_mat_inverse(sc->prims[i].M, inv);
            cx->c.prim_tests++;
            if (isect_dispatch(sc->prims[i].type, inv, o, d, &t, &n) && t > tmin && t < best->t) {
                best->t = t;
                best->n = n;
                best->prim = i;
            }
        }
        return best->prim >= 0;
    }
    /* mode 1: canonical traversal. visit = fetch one 32-B node record and test its box. nearest child first,
       far child pushed with its entry distance and culled against the current tmax when popped. */
    const oracle_node* nodes = cx->nodes;
    const v3 id = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int stack_n[64];
    float stack_t[64];
    int sp = 0;
    float tn;
    cx->c.node_visits++;
    if (!box_test(&nodes[0], o, id, tmin, best->t, &tn)) return 0;
    int node = 0;
    for (;;) {
        const oracle_node* nd = &nodes[node];
        if (nd->right < 0) {
            const int i = nd->left;
            float t;
            v3 n;
            cx->c.prim_tests++;
            if (isect_dispatch(sc->prims[i].type, cx->inv[i], o, d, &t, &n) && t > tmin &&
                (t < best->t || (t == best->t && best->prim >= 0 && i < best->prim))) {
                best->t = t;
                best->n = n;
                best->prim = i;
            }
            node = -1;
        } else {
            float tl, tr;
            cx->c.node_visits += 2;
            int hl = box_test(&nodes[nd->left], o, id, tmin, best->t, &tl);
            int hr = box_test(&nodes[nd->right], o, id, tmin, best->t, &tr);
            if (hl && hr) {
                int nearc = nd->left, farc = nd->right;
                float tfar = tr;
                if (tr < tl) {
                    nearc = nd->right;
                    farc = nd->left;
                    tfar = tl;
                }
                stack_n[sp] = farc;
                stack_t[sp] = tfar;
                ++sp;
                node = nearc;
            } else if (hl)
                node = nd->left;
            else if (hr)
                node = nd->right;
            else
                node = -1;
        }
        while (node < 0) {
            if (sp == 0) return best->prim >= 0;
            --sp;
            if (stack_t[sp] <= best->t) node = stack_n[sp]; /* entry distance still inside [tmin, current tmax] */
        }
    }
}

/* forward */
static void trace_radiance(tctx* cx, v3 o, v3 d, float tmin, float tmax, v3* prd, int depth, uint32_t seed);

static void trace_occlusion(tctx* cx, v3 o, v3 d, float tmin, float tmax, v3* prd)
{
    /* trace(..., RAY_TYPE_OCCLUSION, ...) kernel.cu:46-79 + __closesthit__full_occlusion :539-549.
       miss: missSBTIndex 1 is out of bounds in the reference; de-facto the payload keeps its initial value. */
    hit_t h;
    cx->c.rays_occlusion++;
    cx->c.rays_total++;
    if (closest_hit(cx, o, d, tmin, tmax, &h)) {
        const oracle_prim* p = &cx->sc->prims[h.prim];
        cx->c.hits++;
        *prd = V3(fminf(p->Le[0], 1.0f), fminf(p->Le[1], 1.0f), fminf(p->Le[2], 1.0f));
    }
}

static void closesthit_ch(tctx* cx, const hit_t* h, v3 origin, v3 direction, v3* payload, int depth, uint32_t seed)
{
    /* __closesthit__ch, kernel.cu:426-536 */
    const oracle_frame* fr = cx->fr;
    const oracle_scene* sc = cx->sc;
    const oracle_prim* mat = &sc->prims[h->prim];
    const int dt = fr->host_double_trig;
    v3 N = vnormalize(h->n);
    const float t = h->t;
    const float rayEpsilon = 1e-6f * fmaxf(t * t, 1.0f);
    const v3 x = vadd(origin, vscale(vnormalize(direction), t));
    const v3 V = vnormalize(vsub(origin, x));
    if (vdot(N, V) < 0.0f) N = vscale(N, -1.0f);
    const v3 kd = ld3(mat->kd);
    const v3 Le = ld3(mat->Le);
    v3 color = V3(0, 0, 0);
    v3 prd = V3(0, 0, 0);
    if (fr->path_tracing) {
        if (Le.x > 0.01f) {
            *payload = Le;
        } else {
            if (depth < fr->max_depth) {
                const v3 Ra = hemisphere_x(N, N, 0.0f, &seed, dt);
                ++depth;
                trace_radiance(cx, x, Ra, rayEpsilon, 1e6f, &prd, depth, seed);
                color = vadd(color, vscale(kd, vdot(N, Ra)));
                color = vmul(color, prd);
            }
            *payload = color;
        }
    } else {
        if (vlength(Le) > 0.01f) {
            *payload = V3(1.0f, 1.0f, 1.0f);
            return;
        }
        const int l = (int)(oracle_rnd(&seed) * (float)(sc->n_lights - 1));
        const oracle_light* L = &sc->lights[l];
        const v3 lightNormal = ld3(L->normal), v1 = ld3(L->v1), v2 = ld3(L->v2), corner = ld3(L->corner);
        const float ra = oracle_rnd(&seed); /* Q1: source order, v1 factor first */
        const float rb = oracle_rnd(&seed);
        const v3 samplingPos = vadd(vadd(corner, vscale(v1, ra)), vscale(v2, rb));
        const v3 Lm = vnormalize(vsub(samplingPos, x));
        const float lightDistance = vlength(vsub(samplingPos, x));
        v3 illumination = V3(1.0f, 1.0f, 1.0f);
        trace_occlusion(cx, x, Lm, rayEpsilon, lightDistance - rayEpsilon, &illumination);
        const v3 lightColor = vscale(illumination, fabsf(vdot(Lm, lightNormal)));
        const float falloff = 1.0f / (1.0f + L->falloff * lightDistance);
        const v3 compDiffuse =
            vdot(N, V) < 0.f ? V3(0, 0, 0) : vmul(vscale(lightColor, fmaxf(vdot(N, Lm), 0.0f)), kd);
        color = vadd(color, vscale(compDiffuse, falloff));
        if (depth < fr->max_depth) {
            const v3 omega = vneg(vnormalize(direction));
            const v3 Rr = vadd(vneg(omega), vscale(N, 2 * vdot(N, omega)));
            const v3 kr = ld3(mat->kr);
            if (fr->use_ambient) {
                if (mat->specularity > 0.5f) {
                    ++depth;
                    const v3 r = hemisphere_x(N, Rr, mat->specularity, &seed, dt);
                    trace_radiance(cx, x, r, rayEpsilon, 1e6f, &prd, depth, seed);
                    color = vadd(color, vmul(kr, prd));
                }
                color = vadd(color, vmul(kd, V3(0.1f, 0.1f, 0.1f)));
            } else {
                ++depth;
                const v3 r = mat->specularity < 0.5f ? hemisphere_x(N, N, 0.0f, &seed, dt)
                                                     : hemisphere_x(N, Rr, mat->specularity, &seed, dt);
                trace_radiance(cx, x, r, rayEpsilon, 1e6f, &prd, depth, seed);
                color = vadd(color, vmul(kr, prd));
            }
        }
        *payload = color;
    }
}

static void trace_radiance(tctx* cx, v3 o, v3 d, float tmin, float tmax, v3* prd, int depth, uint32_t seed)
{
    /* trace(..., RAY_TYPE_RADIANCE, ...) kernel.cu:46-79: depth and seed travel by value */
    hit_t h;
    cx->c.rays_radiance[depth < 7 ? depth : 7]++;
    cx->c.rays_total++;
    if (closest_hit(cx, o, d, tmin, tmax, &h)) {
        cx->c.hits++;
        closesthit_ch(cx, &h, o, d, prd, depth, seed);
    } else {
        *prd = ld3(cx->sc->bg); /* __miss__ms, kernel.cu:419-423 */
    }
}

uint32_t oracle_local_rows(uint32_t h, uint32_t band_h, uint32_t n_ranks, uint32_t rank)
{
    if (n_ranks <= 1) return h;
    uint32_t rows = 0;
    for (uint32_t r = 0; r < h; ++r)
        if ((r / band_h) % n_ranks == rank) ++rows;
    return rows;
}

static void raygen_pixel(tctx* cx, uint32_t px, uint32_t py, float* accum4, uint8_t* image4)
{
    /* __raygen__rg, kernel.cu:184-247, for launch index (px, py) of a width x height launch */
    const oracle_frame* fr = cx->fr;
    const oracle_scene* sc = cx->sc;
    const float x = (float)px, y = (float)py;
    const float dimX = (float)fr->width, dimY = (float)fr->height;
    const v3 U = ld3(sc->U), Vv = ld3(sc->V), W = ld3(sc->W);
    v3 color = V3(0.0f, 0.0f, 0.0f);
    const uint32_t n = (uint32_t)fr->sqrt_spp;
    const uint32_t image_index = fr->width * py + px;
    uint32_t seed = oracle_tea16(image_index, fr->frame_count);
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            const float inc = 1.0f / (float)n;
            const float fi = (float)i, fj = (float)j;
            const float r0 = oracle_rnd(&seed); /* Q1: x jitter first */
            const float r1 = oracle_rnd(&seed);
            const float dx = 2.0f * ((x + (fi + r0) * inc) / dimX) - 1.0f;
            const float dy = 2.0f * ((y + (fj + r1) * inc) / dimY) - 1.0f;
            const v3 origin = ld3(sc->eye);
            const v3 direction = vnormalize(vadd(vadd(vscale(U, dx), vscale(Vv, dy)), W));
            v3 payload = V3(0.5f, 0.5f, 0.5f);
            trace_radiance(cx, origin, direction, 0.05f, 1e16f, &payload, 0, seed);
            color = vadd(color, payload);
        }
    /* float3 / float = multiply by the reciprocal (vec_math.h:479-483) */
    v3 cur = vscale(color, 1.0f / (float)(n * n));
    if (fr->frame_count > 0) {
        const v3 prev = ld3(accum4);
        const float ratio = 1.0f / (float)(fr->frame_count + 1);
        cur = vadd(prev, vscale(vsub(cur, prev), ratio)); /* lerp, vec_math.h:496-499 */
    }
    accum4[0] = cur.x;
    accum4[1] = cur.y;
    accum4[2] = cur.z;
    accum4[3] = 1.0f;
    /* make_color, kernel.cu:90-98 */
    image4[0] = (uint8_t)(clampf(cur.x, 0.0f, 1.0f) * 255.0f);
    image4[1] = (uint8_t)(clampf(cur.y, 0.0f, 1.0f) * 255.0f);
    image4[2] = (uint8_t)(clampf(cur.z, 0.0f, 1.0f) * 255.0f);
    image4[3] = 255u;
}

int oracle_render(const oracle_scene* sc, const oracle_frame* fr, float* accum, uint8_t* image, oracle_counters* ctr)
{
    const int n = sc->n_prims;
    float(*inv)[16] = NULL;
    oracle_node* nodes = NULL;
    if (fr->mode == 1) {
        inv = (float(*)[16])malloc(sizeof(float[16]) * (size_t)n);
        nodes = (oracle_node*)malloc(sizeof(oracle_node) * (size_t)(2 * n));
        uint32_t* code = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
        int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
        for (int i = 0; i < n; ++i) oracle_mat_inverse(sc->prims[i].M, inv[i]);
        oracle_lbvh_build(sc->aabb, n, nodes, code, order);
        free(code);
        free(order);
    }
    const uint32_t n_ranks = fr->n_ranks ? fr->n_ranks : 1, band_h = fr->band_h ? fr->band_h : 1;
    /* local row table */
    uint32_t* rows = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(fr->h ? fr->h : 1));
    uint32_t nrows = 0;
    for (uint32_t r = 0; r < fr->h; ++r)
        if (n_ranks == 1 || (r / band_h) % n_ranks == fr->rank) rows[nrows++] = r;
    oracle_counters total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    int nth = fr->threads > 0 ? fr->threads : omp_get_max_threads();
#pragma omp parallel num_threads(nth)
#endif
    {
        tctx cx;
        memset(&cx, 0, sizeof cx);
        cx.sc = sc;
        cx.fr = fr;
        cx.inv = (const float(*)[16])inv;
        cx.nodes = nodes;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (uint32_t lr = 0; lr < nrows; ++lr) {
            const uint32_t gy = fr->y0 + rows[lr];
            for (uint32_t lx = 0; lx < fr->w; ++lx) {
                const size_t idx = (size_t)lr * fr->w + lx;
                raygen_pixel(&cx, fr->x0 + lx, gy, accum + 4 * idx, image + 4 * idx);
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            for (int k = 0; k < 8; ++k) total.rays_radiance[k] += cx.c.rays_radiance[k];
            total.rays_occlusion += cx.c.rays_occlusion;
            total.rays_total += cx.c.rays_total;
            total.node_visits += cx.c.node_visits;
            total.prim_tests += cx.c.prim_tests;
            total.hits += cx.c.hits;
        }
    }
    if (ctr) *ctr = total;
    free(rows);
    free(inv);
    free(nodes);
    return 0;
}
