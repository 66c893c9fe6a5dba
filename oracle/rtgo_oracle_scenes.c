/*
 * rtgo_oracle_scenes.c -- CPU ORACLE (test infrastructure): restatement of the reference's eight procedural scenes
 * (engine/scene.cpp:29-671), shape factory (shapefactory.cpp:9-76), Shape/Primitive transform rule (shape.cpp:7-19,
 * primitive.cpp:120-123), material constants (materials.h:13-283) and the flattening done by the renderer
 * (renderer.cpp:321-336 camera, :386-398 miss, :400-453 primitives in shape order, :655-677 lights).
 *
 * PARITY UNPINNED for the tables themselves (scene.cpp cannot be compiled here: it needs <optix.h>); what IS pinned:
 * every matrix operation used below (oracle/_ref golden vectors) and the facts SURVEY.md records (primitive counts,
 * first ball position, image means).  rnd() draws follow source order, left to right.
 */
#include "rtgo_oracle.h"

#include <math.h>
#include <string.h>

#define PI_F 3.14159265358979323846f

typedef struct {
    float kd[3], kr[3], le[3];
    float spec;
} omat;

#define MAT(name, kdr, kdg, kdb, krr, krg, krb, ler, leg, leb, sp) \
    static const omat name = { { kdr, kdg, kdb }, { krr, krg, krb }, { ler, leg, leb }, sp }

/* materials.h -- constructor argument order is (kd, kr, le, specularity) (basicmaterial.cpp:17-27) */
MAT(m_grey, 0.29f, 0.29f, 0.29f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);                           /* :28-33 */
MAT(m_cream, 1.0f, 0.941f, 0.729f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);                         /* :36-41 */
MAT(m_msBlackMirror, 0.05f, 0.05f, 0.05f, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 100000.0f);             /* :55-60 */
MAT(m_msGround, 0.9765f, 0.651f, 0.6549f, 0.7f, 0.7f, 0.7f, 0.0f, 0.0f, 0.0f, 100000.0f);             /* :63-68 */
MAT(m_msOrange, 0.8549f, 0.4078f, 0.0588f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 10000.0f);             /* :71-76 */
MAT(m_msSilver, 0.7529f, 0.7529f, 0.7529f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 10000.0f);             /* :79-84 */
MAT(m_plateGold, 0.83f, 0.69f, 0.22f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 10000.0f);                  /* :89-94 */
MAT(m_platePurple, 1.0f, 0.0f, 1.0f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 1000.0f);                    /* :97-102 */
MAT(m_plateCyan, 0.1f, 1.0f, 1.0f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 5000.0f);                      /* :105-110 */
MAT(m_plateGreen, 0.16f, 0.83f, 0.18f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 1000.0f);                  /* :113-118 */
MAT(m_plateDarkRed, 0.5f, 0.0f, 0.0f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 100.0f);                    /* :121-126 */
MAT(m_plateYellow, 1.0f, 1.0f, 0.0f, 0.7f, 0.7f, 0.7f, 0.0f, 0.0f, 0.0f, 0.0f);                       /* :129-134 */
MAT(m_cornellWhite, 0.8f, 0.8f, 0.8f, 0.3f, 0.3f, 0.3f, 0.0f, 0.0f, 0.0f, 1.0f);                      /* :155-160 */
MAT(m_cornellBlue, 0.0f, 0.0f, 1.0f, 0.3f, 0.3f, 0.3f, 0.0f, 0.0f, 0.0f, 1.0f);                       /* :163-168 */
MAT(m_cornellRed, 1.0f, 0.0f, 0.0f, 0.3f, 0.3f, 0.3f, 0.0f, 0.0f, 0.0f, 1.0f);                        /* :171-176 */
MAT(m_cornellLight, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 15.f, 15.f, 15.f, 1.0f);                      /* :179-184 */
MAT(m_light12, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 12.0f, 12.0f, 12.0f, 1.0f); /* Checkered/Balls/WindowLight :253-272 */
MAT(m_windowWhite, 0.9f, 0.9f, 0.9f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f, 100.0f);                     /* :275-280 */

/* softMirrorsMirror0..7 (materials.h:188-249): same kd/kr, specularity table */
static const float SOFT_MIRROR_SPEC[8] = { 500000.0f, 100000.0f, 50000.0f, 10000.0f, 5000.0f, 1000.0f, 500.0f, 100.0f };

/* materials.h constants no scene at HEAD uses, kept so that the whole table is held to the reference (tests/golden/ref_blocks.json) */
MAT(m_blackMirror, 0.1f, 0.1f, 0.1f, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f);                       /* :13-18 */
MAT(m_blue, 0.0f, 0.549f, 0.988f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);                          /* :21-26 */
MAT(m_white, 0.8f, 0.8f, 0.8f, 0.3f, 0.3f, 0.3f, 0.0f, 0.0f, 0.0f, 1.0f);                             /* :44-49 */
MAT(m_plateLight, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 15.f, 15.f, 15.f, 0.0f);                        /* :137-142 */
MAT(m_cornellMirror, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f);                     /* :147-152 */

/* the table by the reference's names: kd[3], kr[3], Le[3], specularity.  returns 0, or -1 for an unknown name */
int oracle_material(const char* name, float* out10)
{
    static const struct { const char* name; const omat* m; } table[] = {
        { "blackMirror", &m_blackMirror }, { "blue", &m_blue }, { "grey", &m_grey }, { "cream", &m_cream }, { "white", &m_white },
        { "mirrorSpheresBlackMirror", &m_msBlackMirror }, { "mirrorSpheresGroundMat", &m_msGround },
        { "mirrorSpheresMetallicOrange", &m_msOrange }, { "mirrorSpheresSilver", &m_msSilver },
        { "plateMetallicGold", &m_plateGold }, { "platePurple", &m_platePurple }, { "plateCyan", &m_plateCyan },
        { "platePrettyGreen", &m_plateGreen }, { "plateDarkRed", &m_plateDarkRed }, { "plateYellow", &m_plateYellow },
        { "plateLight", &m_plateLight }, { "cornellMirror", &m_cornellMirror }, { "cornellWhite", &m_cornellWhite },
        { "cornellBlue", &m_cornellBlue }, { "cornellRed", &m_cornellRed }, { "cornellLight", &m_cornellLight },
        { "CheckeredLight", &m_light12 }, { "BallsLight", &m_light12 }, { "WindowLight", &m_light12 }, { "windowWhite", &m_windowWhite },
    };
    omat m;
    int found = 0;
    for (size_t i = 0; i < sizeof table / sizeof table[0] && !found; ++i)
        if (strcmp(name, table[i].name) == 0) {
            m = *table[i].m;
            found = 1;
        }
    if (!found && strncmp(name, "softMirrorsMirror", 17) == 0 && name[17] >= '0' && name[17] <= '7' && name[18] == 0) {
        m = m_msBlackMirror;                       /* materials.h:188-249: kd 0.05, kr 1, Le 0 */
        m.spec = SOFT_MIRROR_SPEC[name[17] - '0'];
        found = 1;
    }
    if (!found) return -1;
    memcpy(out10 + 0, m.kd, sizeof m.kd);
    memcpy(out10 + 3, m.kr, sizeof m.kr);
    memcpy(out10 + 6, m.le, sizeof m.le);
    out10[9] = m.spec;
    return 0;
}

typedef struct {
    int n;
    oracle_prim p[8];
} oshape;

typedef struct { float m[16]; } M4;

static M4 T(float x, float y, float z) { M4 r; oracle_mat_translate(x, y, z, r.m); return r; }       /* scene.cpp:679-682 */
static M4 S(float x, float y, float z) { M4 r; oracle_mat_scale(x, y, z, r.m); return r; }           /* scene.cpp:684-687 */
static M4 R(float a, float x, float y, float z) { M4 r; oracle_mat_rotate(a, x, y, z, r.m); return r; } /* :689-692 */
static M4 I(void) { M4 r; oracle_mat_identity(r.m); return r; }
static M4 mul(M4 a, M4 b) { M4 r; oracle_mat_mul(a.m, b.m, r.m); return r; }
static M4 mul3(M4 a, M4 b, M4 c) { return mul(mul(a, b), c); } /* a * b * c associates left to right */

static oracle_prim make_prim(int type, M4 m, const omat* mat)
{
    oracle_prim p;
    memset(&p, 0, sizeof p);
    p.type = type;
    memcpy(p.M, m.m, sizeof p.M);
    memcpy(p.kd, mat->kd, sizeof p.kd);
    memcpy(p.kr, mat->kr, sizeof p.kr);
    memcpy(p.Le, mat->le, sizeof p.Le);
    p.specularity = mat->spec;
    return p;
}

/* Shape::Transform (shape.cpp:13-19) -> Primitive::Transform (primitive.cpp:120-123): m = transform * m */
static void shape_transform(oshape* s, M4 x)
{
    for (int i = 0; i < s->n; ++i) oracle_mat_mul(x.m, s->p[i].M, s->p[i].M);
}

/* Shape ctor (shape.cpp:7-11) */
static oshape shape_from(const oracle_prim* prims, int n, M4 model)
{
    oshape s;
    s.n = n;
    memcpy(s.p, prims, sizeof(oracle_prim) * (size_t)n);
    shape_transform(&s, model);
    return s;
}

static oshape single(int type, M4 model, const omat* mat) /* shapefactory.cpp:9-14,16-21,34-39,41-46 */
{
    oracle_prim p = make_prim(type, I(), mat);
    return shape_from(&p, 1, model);
}
static oshape mk_rectangle(M4 m, const omat* mat) { return single(ORACLE_RECTANGLE, m, mat); }
static oshape mk_open_cylinder(M4 m, const omat* mat) { return single(ORACLE_CYLINDER, m, mat); }
static oshape mk_disk(M4 m, const omat* mat) { return single(ORACLE_DISK, m, mat); }
static oshape mk_sphere(M4 m, const omat* mat) { return single(ORACLE_SPHERE, m, mat); }

static oshape mk_closed_cylinder(M4 m, const omat* mat) /* shapefactory.cpp:23-32 */
{
    oracle_prim p[3];
    p[0] = make_prim(ORACLE_CYLINDER, I(), mat);
    p[1] = make_prim(ORACLE_DISK, T(0.0f, 1.0f, 0.0f), mat);
    p[2] = make_prim(ORACLE_DISK, T(0.0f, -1.0f, 0.0f), mat);
    return shape_from(p, 3, m);
}

static oshape mk_cube(M4 m, const omat* mat) /* shapefactory.cpp:48-68 */
{
    oracle_prim p[6];
    p[0] = make_prim(ORACLE_RECTANGLE, mul(T(0.5f, 0.0f, 0.0f), R(-PI_F / 2.0f, 0.0f, 0.0f, 1.0f)), mat);
    p[1] = make_prim(ORACLE_RECTANGLE, mul(T(-0.5f, 0.0f, 0.0f), R(PI_F / 2.0f, 0.0f, 0.0f, 1.0f)), mat);
    p[2] = make_prim(ORACLE_RECTANGLE, T(0.0f, 0.5f, 0.0f), mat);
    p[3] = make_prim(ORACLE_RECTANGLE, mul(T(0.0f, -0.5f, 0.0f), R((float)M_PI, 0.0f, 0.0f, 1.0f)), mat); /* M_PI (double) -> float */
    p[4] = make_prim(ORACLE_RECTANGLE, mul(T(0.0f, 0.0f, 0.5f), R(PI_F / 2.0f, 1.0f, 0.0f, 0.0f)), mat);
    p[5] = make_prim(ORACLE_RECTANGLE, mul(T(0.0f, 0.0f, -0.5f), R(-PI_F / 2.0f, 1.0f, 0.0f, 0.0f)), mat);
    return shape_from(p, 6, m);
}

/* the "triangle" custom shape shared by soft_mirrors/plateau/slide (scene.cpp:93-128, 203-238, 368-403).
   cos()/sin() of a float argument: see note in triangle_prims. */
static void triangle_prims(oracle_prim* p, const omat* mat)
{
    /* scene.cpp has no <cmath> using-declaration of its own; with MSVC (the reference toolchain) ::cos(float) is the
       float overload. cosf/sinf and (float)cos((double)x) agree bit for bit for these two arguments (tests check). */
    const float c = cosf(PI_F / 3.0f), s = sinf(PI_F / 3.0f);
    p[0] = make_prim(ORACLE_CYLINDER, mul3(T(c, s, 0.0f), R(PI_F / 6.0f, 0.0f, 0.0f, 1.0f), S(0.4f, 1.0f, 0.4f)), mat);
    p[1] = make_prim(ORACLE_CYLINDER, mul3(T(-c, s, 0.0f), R(-PI_F / 6.0f, 0.0f, 0.0f, 1.0f), S(0.4f, 1.0f, 0.4f)), mat);
    p[2] = make_prim(ORACLE_CYLINDER, mul(R(PI_F / 2.0f, 0.0f, 0.0f, 1.0f), S(0.4f, 1.0f, 0.4f)), mat);
    p[3] = make_prim(ORACLE_SPHERE, mul(T(-1.0f, 0.0f, 0.0f), S(0.4f, 0.4f, 0.4f)), mat);
    p[4] = make_prim(ORACLE_SPHERE, mul(T(1.0f, 0.0f, 0.0f), S(0.4f, 0.4f, 0.4f)), mat);
    p[5] = make_prim(ORACLE_SPHERE, mul(T(0.0f, sqrtf(3.0f), 0.0f), S(0.4f, 0.4f, 0.4f)), mat);
}

/* ---- scene accumulation (Scene::AddObject scene.cpp:673-677 + flattening renderer.cpp:414-441) ---- */

static int add_object(oracle_scene* sc, const oshape* s)
{
    for (int i = 0; i < s->n; ++i) {
        if (sc->n_prims >= ORACLE_MAX_PRIMS) return -1;
        sc->prims[sc->n_prims] = s->p[i];
        oracle_prim_aabb(s->p[i].M, sc->aabb[sc->n_prims]);
        sc->n_prims++;
    }
    return 0;
}

static void add_light(oracle_scene* sc, const oshape* lightObj, float r, float g, float b, float falloff)
{
    /* SurfaceLight(primitive.GetType(), primitive.GetModelMatrix(), color, falloff) + Renderer::WriteLights */
    float color[3] = { r, g, b };
    oracle_light_from_matrix(lightObj->p[0].M, color, falloff, &sc->lights[sc->n_lights]);
    sc->n_lights++;
}

/* the six room walls used by cornell / mirror_spheres (scale 8) */
static void room8(const omat* back_m, const omat* front_m, const omat* left_m, const omat* right_m, const omat* top_m,
                  const omat* bottom_m, oshape* back, oshape* front, oshape* left, oshape* right, oshape* top,
                  oshape* bottom)
{
    *back = mk_rectangle(mul3(T(0.0f, 0.0f, -4.0f), R(PI_F / 2.0f, 1.0f, 0.0f, 0.0f), S(8.0f, 1.0f, 8.0f)), back_m);
    *front = mk_rectangle(mul3(T(0.0f, 0.0f, 4.0f), R(-PI_F / 2.0f, 1.0f, 0.0f, 0.0f), S(8.0f, 1.0f, 8.0f)), front_m);
    *left = mk_rectangle(mul3(T(-4.0f, 0.0f, 0.0f), R(-PI_F / 2.0f, 0.0f, 0.0f, 1.0f), S(8.0f, 1.0f, 8.0f)), left_m);
    *right = mk_rectangle(mul3(T(4.0f, 0.0f, 0.0f), R(PI_F / 2.0f, 0.0f, 0.0f, 1.0f), S(8.0f, 1.0f, 8.0f)), right_m);
    *top = mk_rectangle(mul3(T(0.0f, 4.0f, 0.0f), R(PI_F, 0.0f, 0.0f, 1.0f), S(8.0f, 1.0f, 8.0f)), top_m);
    *bottom = mk_rectangle(mul(T(0.0f, -4.0f, 0.0f), S(8.0f, 1.0f, 8.0f)), bottom_m);
}

/* the five/six walls of window / checkered / balls (scale 16 x 8) */
static void room16(const omat* back_m, const omat* front_m, const omat* left_m, const omat* right_m, const omat* top_m,
                   oshape* back, oshape* front, oshape* left, oshape* right, oshape* top)
{
    *back = mk_rectangle(mul3(T(0.0f, 0.0f, -8.0f), R(PI_F / 2.0f, 1.0f, 0.0f, 0.0f), S(16.0f, 1.0f, 8.0f)), back_m);
    *front = mk_rectangle(mul3(T(0.0f, 0.0f, 8.0f), R(-PI_F / 2.0f, 1.0f, 0.0f, 0.0f), S(16.0f, 1.0f, 8.0f)), front_m);
    *left = mk_rectangle(mul3(T(-8.0f, 0.0f, 0.0f), R(-PI_F / 2.0f, 0.0f, 0.0f, 1.0f), S(8.0f, 1.0f, 16.0f)), left_m);
    *right = mk_rectangle(mul3(T(8.0f, 0.0f, 0.0f), R(PI_F / 2.0f, 0.0f, 0.0f, 1.0f), S(8.0f, 1.0f, 16.0f)), right_m);
    *top = mk_rectangle(mul3(T(0.0f, 4.0f, 0.0f), R(PI_F, 0.0f, 0.0f, 1.0f), S(16.0f, 1.0f, 16.0f)), top_m);
}

static void create_mirror_spheres(oracle_scene* sc) /* scene.cpp:29-88 */
{
    oshape back, front, left, right, top, bottom;
    room8(&m_msBlackMirror, &m_msBlackMirror, &m_msBlackMirror, &m_msBlackMirror, &m_msBlackMirror, &m_msGround, &back,
          &front, &left, &right, &top, &bottom);
    oshape orange = mk_sphere(I(), &m_msOrange);
    oshape silver = mk_sphere(T(1.0f, 1.0f, -1.0f), &m_msSilver);
    add_object(sc, &back);
    add_object(sc, &front);
    add_object(sc, &left);
    add_object(sc, &bottom);
    add_object(sc, &top);
    add_object(sc, &right);
    add_object(sc, &silver);
    add_object(sc, &orange);
    oshape light = mk_rectangle(mul3(T(0.0f, 3.95f, 0.0f), R(PI_F, 1.0f, 0.0f, 0.0f), S(4.0f, 1.0f, 4.0f)), &m_cornellLight);
    add_light(sc, &light, 1.0f, 1.0f, 1.0f, 0.1f);
    add_object(sc, &light);
}

static void create_soft_mirrors(oracle_scene* sc) /* scene.cpp:90-198 */
{
    oracle_prim tp[6];
    triangle_prims(tp, &m_plateGold);
    oshape tri = shape_from(tp, 6, T(0.0f, -1.5f, 0.0f));
    add_object(sc, &tri);
    omat mm = m_msBlackMirror; /* kd 0.05, kr 1, Le 0: identical to softMirrorsMirrorK apart from specularity */
    mm.spec = SOFT_MIRROR_SPEC[0];
    oshape mirror0 = mk_rectangle(mul3(T(0.0f, 0.0f, -5.0f), R(PI_F / 2.0f, 1.0f, 0.0f, 0.0f), S(4.0f, 1.0f, 4.0f)), &mm);
    add_object(sc, &mirror0);
    M4 m0;
    memcpy(m0.m, mirror0.p[0].M, sizeof m0.m);
    const float ang[8] = { 0.0f, PI_F / 4.0f, PI_F / 2.0f, 3.0f * PI_F / 4.0f, PI_F,
                           5.0f * PI_F / 4.0f, 3.0f * PI_F / 2.0f, 7.0f * PI_F / 4.0f };
    for (int k = 1; k < 8; ++k) {
        mm.spec = SOFT_MIRROR_SPEC[k];
        oshape mk = mk_rectangle(mul(R(ang[k], 0.0f, 1.0f, 0.0f), m0), &mm);
        add_object(sc, &mk);
    }
    oshape ground = mk_disk(mul(T(0.0f, -2.0f, 0.0f), S(4.0f, 1.0f, 4.0f)), &m_platePurple);
    add_object(sc, &ground);
    oshape light = mk_rectangle(mul3(T(0.0f, 6.0f, 0.0f), R(PI_F, 1.0f, 0.0f, 0.0f), S(4.0f, 1.0f, 4.0f)), &m_cornellLight);
    add_light(sc, &light, 15.f, 15.f, 15.f, 0.0f); /* color = GetLe() */
    add_object(sc, &light);
}

static void create_fun_plate(oracle_scene* sc) /* scene.cpp:200-277 */
{
    oracle_prim tp[6];
    triangle_prims(tp, &m_plateGold);
    oshape tri = shape_from(tp, 6, I());
    shape_transform(&tri, R(-PI_F / 2.0f, 1.0f, 0.0f, 0.0f));
    shape_transform(&tri, T(0.0f, 2.5f, 0.5f));
    add_object(sc, &tri);
    oshape disk0 = mk_disk(S(4.0f, 1.0f, 4.0f), &m_platePurple);
    shape_transform(&disk0, T(0.0f, -1.0f, 0.0f));
    add_object(sc, &disk0);
    oshape sphere0 = mk_sphere(T(-2.5f, 1.0f, -0.5f), &m_plateCyan);
    add_object(sc, &sphere0);
    oshape sphere1 = mk_sphere(S(1.0f, 2.0f, 1.0f), &m_plateGreen);
    shape_transform(&sphere1, T(1.0f, 1.0f, -2.5f));
    add_object(sc, &sphere1);
    oshape cyl0 = mk_closed_cylinder(T(-0.5f, 0.1f, 1.0f), &m_plateDarkRed);
    add_object(sc, &cyl0);
    oshape cube0 = mk_cube(R(PI_F / 4.0f, 1.0f, 1.0f, 1.0f), &m_plateYellow);
    shape_transform(&cube0, T(2.0f, 0.25f, 0.5f));
    add_object(sc, &cube0);
    oshape light = mk_rectangle(mul3(T(0.0f, 6.0f, 0.0f), R(PI_F, 1.0f, 0.0f, 0.0f), S(4.0f, 1.0f, 4.0f)), &m_cornellLight);
    add_light(sc, &light, 15.f, 15.f, 15.f, 0.0f);
    add_object(sc, &light);
}

static void create_cornell(oracle_scene* sc) /* scene.cpp:279-336 */
{
    oshape back, front, left, right, top, bottom;
    room8(&m_cornellWhite, &m_cornellWhite, &m_cornellRed, &m_cornellBlue, &m_cornellWhite, &m_cornellWhite, &back,
          &front, &left, &right, &top, &bottom);
    oshape box1 = mk_cube(mul3(T(1.3f, -3.0f, 1.3f), R(-PI_F / 6.0f, 0.0f, 1.0f, 0.0f), S(2.0f, 2.0f, 2.0f)), &m_cornellWhite);
    oshape box2 = mk_cube(mul3(T(-1.3f, -2.0f, -1.3f), R(PI_F / 8.0f, 0.0f, 1.0f, 0.0f), S(2.0f, 4.0f, 2.0f)), &m_cornellWhite);
    add_object(sc, &back);
    add_object(sc, &front);
    add_object(sc, &left);
    add_object(sc, &right);
    add_object(sc, &top);
    add_object(sc, &bottom);
    add_object(sc, &box1);
    add_object(sc, &box2);
    oshape light = mk_rectangle(mul3(T(0.0f, 3.95f, 0.0f), R(PI_F, 1.0f, 0.0f, 0.0f), S(4.0f, 1.0f, 4.0f)), &m_cornellLight);
    add_light(sc, &light, 1.0f, 1.0f, 1.0f, 0.0f);
    add_object(sc, &light);
}

static void create_slide(oracle_scene* sc) /* scene.cpp:338-425 */
{
    oshape s;
    s = mk_rectangle(mul(T(0.0f, -1.0f, 0.0f), S(150.0f, 1.0f, 8.0f)), &m_grey);
    add_object(sc, &s);
    s = mk_sphere(T(12.0f, 0.0f, 0.0f), &m_plateCyan);
    add_object(sc, &s);
    s = mk_cube(mul(T(6.0f, 0.0f, 0.0f), S(2.0f, 2.0f, 2.0f)), &m_plateCyan);
    add_object(sc, &s);
    s = mk_closed_cylinder(T(0.0f, 0.0f, 0.0f), &m_plateCyan);
    add_object(sc, &s);
    s = mk_disk(T(-6.0f, 0.0f, 0.0f), &m_plateCyan);
    add_object(sc, &s);
    s = mk_rectangle(mul(T(-12.0f, 0.0f, 0.0f), S(2.0f, 1.0f, 2.0f)), &m_plateCyan);
    add_object(sc, &s);
    s = mk_sphere(mul3(T(-20.0f, 2.0f, 0.0f), R(PI_F / 2.0f, 1.0f, 1.0f, .0f), S(3.0f, 2.0f, 2.0f)), &m_cornellBlue);
    add_object(sc, &s);
    s = mk_cube(mul3(T(-30.0f, 2.0f, 0.0f), R(PI_F / 4.0f, 0.f, 1.f, 1.f), S(2.0f, 2.0f, 2.0f)), &m_plateGreen);
    add_object(sc, &s);
    s = mk_open_cylinder(mul3(T(-40.0f, 0.3f, 0.0f), R(PI_F / 2.0f, 1.f, 0.f, 0.f), S(1.0f, 2.0f, 1.0f)), &m_plateGold);
    add_object(sc, &s);
    oracle_prim tp[6];
    triangle_prims(tp, &m_cornellRed);
    oshape tri = shape_from(tp, 6, I());
    shape_transform(&tri, S(1.5f, 1.5f, 1.5f));
    shape_transform(&tri, R(-PI_F / 6.0f, 1.0f, 0.0f, 0.0f));
    shape_transform(&tri, T(-50.0f, 1.2f, 0.7f));
    add_object(sc, &tri);
    s = mk_cube(mul3(T(-60.0f, 2.5f, 0.0f), R(PI_F / 4.0f, 0.f, 1.f, 1.f), S(1.0f, 6.0f, 0.4f)), &m_cream);
    add_object(sc, &s);
    oshape light = mk_rectangle(mul3(T(0.0f, 100.0f, 10.0f), R(PI_F, 1.0f, 0.0f, 0.0f), S(0.001f, 1.0f, 0.001f)), &m_cornellLight);
    add_light(sc, &light, 1.0f, 1.0f, 1.0f, 0.0f);
    add_object(sc, &light);
}

static void create_window(oracle_scene* sc) /* scene.cpp:427-488 */
{
    oshape back, front, left, right, top;
    room16(&m_cornellRed, &m_windowWhite, &m_cornellRed, &m_cornellRed, &m_windowWhite, &back, &front, &left, &right, &top);
    oshape bottom = mk_rectangle(mul(T(0.0f, -4.0f, 0.0f), S(16.0f, 1.0f, 16.0f)), &m_windowWhite);
    oshape wall = mk_cube(mul(T(-1.2f, 0.0f, -7.0f), S(0.5f, 8.0f, 2.0f)), &m_cornellRed);
    oshape sphere0 = mk_sphere(mul(T(2.0f, (float)-3.0, -5.5f), S(1.0f, 1.0f, 1.0f)), &m_plateGreen);
    oshape sphere1 = mk_sphere(mul(T(5.5f, (float)-3.0, -5.5f), S(1.0f, 1.0f, 1.0f)), &m_cornellBlue);
    add_object(sc, &back);
    add_object(sc, &front);
    add_object(sc, &left);
    add_object(sc, &right);
    add_object(sc, &top);
    add_object(sc, &bottom);
    add_object(sc, &sphere0);
    add_object(sc, &sphere1);
    add_object(sc, &wall);
    oshape light = mk_rectangle(mul3(T(-5.0f, 0.0f, -7.99f), R(PI_F / 2.0f, 1.0f, 0.0f, 0.0f), S(4.0f, 1.0f, 4.0f)), &m_light12);
    add_light(sc, &light, 1.0f, 1.0f, 1.0f, 0.02f);
    add_object(sc, &light);
}

static void create_checkered(oracle_scene* sc) /* scene.cpp:490-560 */
{
    oshape back, front, left, right, top;
    room16(&m_windowWhite, &m_windowWhite, &m_windowWhite, &m_windowWhite, &m_windowWhite, &back, &front, &left, &right, &top);
    uint32_t seed = oracle_tea16(12, 1234567);
    for (int i = 0; i < 8; i++) {
        int red = (i % 2 == 0) ? 1 : 0;
        for (int j = 0; j < 8; j++) {
            const float r = oracle_rnd(&seed);
            oshape floor = mk_cube(mul(T(-7.0f + (i * 2.0f), -5.0f + r, -7.0f + (j * 2.0f)), S(2.0f, 2.0f, 2.0f)),
                                   red ? &m_cornellRed : &m_cornellBlue);
            add_object(sc, &floor);
            red = !red;
        }
    }
    add_object(sc, &back);
    add_object(sc, &front);
    add_object(sc, &left);
    add_object(sc, &right);
    add_object(sc, &top);
    oshape light = mk_rectangle(mul3(T(0.0f, 3.96f, 0.0f), R(PI_F, 1.0f, 0.0f, 0.0f), S(6.0f, 1.0f, 6.0f)), &m_light12);
    add_light(sc, &light, 1.0f, 1.0f, 1.0f, 0.01f);
    add_object(sc, &light);
}

static void create_balls(oracle_scene* sc) /* scene.cpp:562-627 */
{
    oshape back, front, left, right, top;
    room16(&m_windowWhite, &m_windowWhite, &m_windowWhite, &m_windowWhite, &m_windowWhite, &back, &front, &left, &right, &top);
    uint32_t seed = oracle_tea16(12, 1234567);
    oshape bottom = mk_rectangle(mul(T(0.0f, -4.0f, 0.0f), S(16.0f, 1.0f, 16.0f)), &m_windowWhite);
    add_object(sc, &bottom);
    const omat* mats[5] = { &m_plateGold, &m_plateCyan, &m_platePurple, &m_plateGreen, &m_windowWhite };
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            /* Q1: four draws in source order */
            const float r0 = oracle_rnd(&seed);
            const float r1 = oracle_rnd(&seed);
            const float r2 = oracle_rnd(&seed);
            const float r3 = oracle_rnd(&seed);
            const float tx = -7.5f + (i * 1.0f) + (-0.2f + (0.4f * r0));
            const float ty = -3.5f + (6.0f * r1);
            const float tz = -7.5f + (j * 1.0f) + (-0.2f + (0.4f * r2));
            oshape sphere = mk_sphere(mul(T(tx, ty, tz), S(0.25f, 0.25f, 0.25f)), mats[(int)(5 * r3)]);
            add_object(sc, &sphere);
        }
    add_object(sc, &back);
    add_object(sc, &front);
    add_object(sc, &left);
    add_object(sc, &right);
    add_object(sc, &top);
    oshape light = mk_rectangle(mul3(T(0.0f, 3.96f, 0.0f), R(PI_F, 1.0f, 0.0f, 0.0f), S(6.0f, 1.0f, 6.0f)), &m_light12);
    add_light(sc, &light, 1.0f, 1.0f, 1.0f, 0.01f);
    add_object(sc, &light);
}

int oracle_scene_create(const char* name, uint32_t width, uint32_t height, oracle_scene* sc)
{
    memset(sc, 0, sizeof *sc);
    if (!strcmp(name, "cornell")) create_cornell(sc);
    else if (!strcmp(name, "slide")) create_slide(sc);
    else if (!strcmp(name, "mirror_spheres")) create_mirror_spheres(sc);
    else if (!strcmp(name, "plateau")) create_fun_plate(sc);
    else if (!strcmp(name, "window")) create_window(sc);
    else if (!strcmp(name, "checkered")) create_checkered(sc);
    else if (!strcmp(name, "balls")) create_balls(sc);
    else if (!strcmp(name, "soft_mirrors")) create_soft_mirrors(sc);
    else return -1;
    /* Scene::SetupCamera, scene.cpp:660-671 + Renderer::CreateRayGen, renderer.cpp:327-331 */
    const float eye[3] = { 0.0f, 0.0f, 14.0f }, lookat[3] = { 0.0f, 0.0f, 0.0f }, up[3] = { 0.0f, 1.0f, 0.0f };
    memcpy(sc->eye, eye, sizeof eye);
    oracle_camera_uvw(eye, lookat, up, 60.0f, (float)width / (float)height, sc->U, sc->V, sc->W);
    sc->bg[0] = sc->bg[1] = sc->bg[2] = 0.0f;
    return 0;
}
