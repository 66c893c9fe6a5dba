// light.h -- engine::host::SurfaceLight (engine/light.h:12-62, light.cpp:9-28): a rectangular area light described by
// the corner and the two edge vectors of a transformed unit rectangle.
#pragma once
#include "primitive.h"

namespace engine {
namespace host {
class SurfaceLight {
public:
    /// \param falloff distance attenuation constant, should be < 1
    SurfaceLight(PRIMITIVE_TYPE type, const sutil::Matrix4x4& modelMatrix, const glm::vec3& color, const float falloff)
        : m_type(type), m_modelMatrix(modelMatrix), m_color(color), m_falloff(falloff)
    {
        auto xyz = [](const float4& v) { return glm::vec3(v.x, v.y, v.z); };
        m_corner = xyz(modelMatrix * make_float4(-0.5f, 0.0f, 0.5f, 1.0f));
        m_v1 = xyz(modelMatrix * make_float4(1.0f, 0.0f, 0.0f, 0.0f));
        m_v2 = xyz(modelMatrix * make_float4(0.0f, 0.0f, -1.0f, 0.0f));
        m_normal = glm::normalize(glm::cross(m_v1, m_v2));
    }
    ~SurfaceLight() = default;
    glm::vec3 GetCorner() const { return m_corner; }
    glm::vec3 GetV1() const { return m_v1; }
    glm::vec3 GetV2() const { return m_v2; }
    glm::vec3 GetNormal() const { return m_normal; }
    glm::vec3 GetColor() const { return m_color; }
    float GetFalloff() const { return m_falloff; }
    PRIMITIVE_TYPE GetType() const { return m_type; }

private:
    PRIMITIVE_TYPE m_type;
    sutil::Matrix4x4 m_modelMatrix;
    glm::vec3 m_corner, m_v1, m_v2, m_normal, m_color;
    float m_falloff;
};
}  // namespace host
}  // namespace engine
