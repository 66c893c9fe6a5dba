// ref_probe.cpp -- ORACLE-SIDE ONLY (test infrastructure).
//
// Thin extern "C" wrapper compiled AGAINST THE REFERENCE'S OWN HEADERS where they lie under /root/reference
// (cuda/random.h, sutil/Matrix.h, sutil/vec_math.h, sutil/Camera.{h,cpp}, support/glm), by oracle/Makefile, into
// oracle/_ref/libref_probe.so.  It contains no reference code: it only calls it, so that oracle/gen_golden.py can
// record golden vectors (tests/golden/ref_*.json) that pin the restatement in oracle/rtgo_oracle.c bit for bit.
//
// Also compiled in: engine/materials.h + engine/basicmaterial.{h,cpp} (glm only): the 33 material constants of the scenes.
// These are the only parts of the hot path's arithmetic that compile here without OptiX: the rest of engine/*.cpp, params.h
// and kernel.cu all include <optix.h> (absent from this image) and are therefore unbuildable -- see DESIGN.md.
// vector_types.h / vector_functions.h come from the CUDA headers this image ships inside the triton wheel.
#include <sutil/Matrix.h>
#include <sutil/vec_math.h>
#include <sutil/Camera.h>
#include <random.h>
#include <glm/glm.hpp>
#include <materials.h>   // engine/materials.h:13-283 (includes <basicmaterial.h>)

#include <cstring>
#include <cstdint>

extern "C" {

uint32_t ref_tea16(uint32_t a, uint32_t b) { return tea<16>(a, b); }
uint32_t ref_tea4(uint32_t a, uint32_t b) { return tea<4>(a, b); }   // whitted.cu:196

uint32_t ref_lcg(uint32_t* s)
{
    unsigned int p = *s;
    unsigned int r = lcg(p);
    *s = p;
    return r;
}

float ref_rnd(uint32_t* s)
{
    unsigned int p = *s;
    float r = rnd(p);
    *s = p;
    return r;
}

static void put(const sutil::Matrix4x4& m, float* out) { std::memcpy(out, m.getData(), 16 * sizeof(float)); }

void ref_mat_mul(const float* a, const float* b, float* out) { put(sutil::Matrix4x4(a) * sutil::Matrix4x4(b), out); }
void ref_mat_inverse(const float* a, float* out) { put(sutil::Matrix4x4(a).inverse(), out); }
float ref_mat_det(const float* a) { return sutil::Matrix4x4(a).det(); }
void ref_mat_transpose(const float* a, float* out) { put(sutil::Matrix4x4(a).transpose(), out); }
void ref_mat_rotate(float rad, float x, float y, float z, float* out)
{
    put(sutil::Matrix4x4::rotate(rad, make_float3(x, y, z)), out);
}
void ref_mat_translate(float x, float y, float z, float* out) { put(sutil::Matrix4x4::translate(make_float3(x, y, z)), out); }
void ref_mat_scale(float x, float y, float z, float* out) { put(sutil::Matrix4x4::scale(make_float3(x, y, z)), out); }
void ref_mat_vec4(const float* m, const float* v, float* out)
{
    float4 r = sutil::Matrix4x4(m) * make_float4(v[0], v[1], v[2], v[3]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void ref_normalize3(const float* v, float* out)
{
    float3 r = normalize(make_float3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_cross3(const float* a, const float* b, float* out)
{
    float3 r = cross(make_float3(a[0], a[1], a[2]), make_float3(b[0], b[1], b[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_dot3(const float* a, const float* b) { return dot(make_float3(a[0], a[1], a[2]), make_float3(b[0], b[1], b[2])); }
float ref_length3(const float* a) { return length(make_float3(a[0], a[1], a[2])); }
void ref_lerp3(const float* a, const float* b, float t, float* out)
{
    float3 r = lerp(make_float3(a[0], a[1], a[2]), make_float3(b[0], b[1], b[2]), t);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_div3(const float* a, float s, float* out)
{
    float3 r = make_float3(a[0], a[1], a[2]) / s;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_clamp(float f, float a, float b) { return clamp(f, a, b); }

void ref_camera_uvw(const float* eye, const float* lookat, const float* up, float fovy, float aspect, float* U, float* V,
                    float* W)
{
    sutil::Camera cam(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                      make_float3(up[0], up[1], up[2]), fovy, aspect);
    float3 u, v, w;
    cam.UVWFrame(u, v, w);
    U[0] = u.x; U[1] = u.y; U[2] = u.z;
    V[0] = v.x; V[1] = v.y; V[2] = v.z;
    W[0] = w.x; W[1] = w.y; W[2] = w.z;
}

// glm::normalize(glm::cross(v1, v2)) as used by SurfaceLight (engine/light.cpp:27)
void ref_glm_normal(const float* v1, const float* v2, float* out)
{
    glm::vec3 n = glm::normalize(glm::cross(glm::vec3(v1[0], v1[1], v1[2]), glm::vec3(v2[0], v2[1], v2[2])));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

// engine/materials.h: the material constants by name -> kd[3], kr[3], Le[3], specularity (BasicMaterial getters, basicmaterial.h:36-50)
int ref_material_count(void) { return 33; }
int ref_material(int index, char* name_out, int name_cap, float* out10)
{
    namespace M = engine::host::materials;
    static const struct { const char* name; const engine::host::BasicMaterial* m; } table[] = {
        {"blackMirror", &M::blackMirror},
        {"blue", &M::blue},
        {"grey", &M::grey},
        {"cream", &M::cream},
        {"white", &M::white},
        {"mirrorSpheresBlackMirror", &M::mirrorSpheresBlackMirror},
        {"mirrorSpheresGroundMat", &M::mirrorSpheresGroundMat},
        {"mirrorSpheresMetallicOrange", &M::mirrorSpheresMetallicOrange},
        {"mirrorSpheresSilver", &M::mirrorSpheresSilver},
        {"plateMetallicGold", &M::plateMetallicGold},
        {"platePurple", &M::platePurple},
        {"plateCyan", &M::plateCyan},
        {"platePrettyGreen", &M::platePrettyGreen},
        {"plateDarkRed", &M::plateDarkRed},
        {"plateYellow", &M::plateYellow},
        {"plateLight", &M::plateLight},
        {"cornellMirror", &M::cornellMirror},
        {"cornellWhite", &M::cornellWhite},
        {"cornellBlue", &M::cornellBlue},
        {"cornellRed", &M::cornellRed},
        {"cornellLight", &M::cornellLight},
        {"softMirrorsMirror0", &M::softMirrorsMirror0},
        {"softMirrorsMirror1", &M::softMirrorsMirror1},
        {"softMirrorsMirror2", &M::softMirrorsMirror2},
        {"softMirrorsMirror3", &M::softMirrorsMirror3},
        {"softMirrorsMirror4", &M::softMirrorsMirror4},
        {"softMirrorsMirror5", &M::softMirrorsMirror5},
        {"softMirrorsMirror6", &M::softMirrorsMirror6},
        {"softMirrorsMirror7", &M::softMirrorsMirror7},
        {"CheckeredLight", &M::CheckeredLight},
        {"BallsLight", &M::BallsLight},
        {"WindowLight", &M::WindowLight},
        {"windowWhite", &M::windowWhite}
    };
    if (index < 0 || index >= (int)(sizeof table / sizeof table[0])) return -1;
    std::strncpy(name_out, table[index].name, (size_t)name_cap - 1);
    name_out[name_cap - 1] = 0;
    const glm::vec3 kd = table[index].m->GetKd(), kr = table[index].m->GetKr(), le = table[index].m->GetLe();
    const float v[10] = {kd.x, kd.y, kd.z, kr.x, kr.y, kr.z, le.x, le.y, le.z, table[index].m->GetSpecularity()};
    std::memcpy(out10, v, sizeof v);
    return 0;
}

} // extern "C"
