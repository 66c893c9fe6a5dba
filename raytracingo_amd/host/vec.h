// vec.h -- the few vector types the host surface needs, without CUDA or glm headers.
// float3/float4 + make_float3 stand in for CUDA's vector_types.h; glm::vec3 for the colour triples of
// BasicMaterial / SurfaceLight (engine/basicmaterial.h, engine/light.h use glm::vec3 in their signatures).
// Operation order follows sutil/vec_math.h so host tables are bit-identical to the reference's (IEEE float32).
#pragma once
#include <cmath>

struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

#ifndef M_PIf
#define M_PIf 3.14159265358979323846f  // sutil/vec_math.h:43
#endif

namespace rtgo_vec {
inline float3 sub(const float3& a, const float3& b) { return float3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 mul(const float3& a, float s) { return float3{a.x * s, a.y * s, a.z * s}; }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // vec_math.h:523-526
inline float3 cross(const float3& a, const float3& b)                                              // vec_math.h:529-532
{
    return float3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(const float3& v) { return sqrtf(dot(v, v)); }                                  // vec_math.h:535-538
inline float3 normalize(const float3& v) { return mul(v, 1.0f / sqrtf(dot(v, v))); }               // vec_math.h:541-545
}  // namespace rtgo_vec

namespace glm {
struct vec3 {
    union { float x, r; };
    union { float y, g; };
    union { float z, b; };
    vec3() : x(0.0f), y(0.0f), z(0.0f) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b_, float c) : x(a), y(b_), z(c) {}
};
// glm/detail/func_geometric.inl:68-90: cross and normalize = v * (1 / sqrt(dot(v, v)))
inline vec3 cross(const vec3& p, const vec3& q) { return vec3(p.y * q.z - q.y * p.z, p.z * q.x - q.z * p.x, p.x * q.y - q.x * p.y); }
inline vec3 normalize(const vec3& v)
{
    const float inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return vec3(v.x * inv, v.y * inv, v.z * inv);
}
}  // namespace glm
