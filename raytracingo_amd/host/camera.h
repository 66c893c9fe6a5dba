// camera.h -- pinhole camera of sutil/Camera.{h,cpp}: the 48 bytes the raygen record needs (eye + UVW frame).
#pragma once
#include "vec.h"

namespace sutil {
class Camera {
public:
    Camera() : m_eye{1.0f, 1.0f, 1.0f}, m_lookat{0.0f, 0.0f, 0.0f}, m_up{0.0f, 1.0f, 0.0f}, m_fovY(35.0f), m_aspectRatio(1.0f) {}
    Camera(const float3& eye, const float3& lookat, const float3& up, float fovY, float aspectRatio)
        : m_eye(eye), m_lookat(lookat), m_up(up), m_fovY(fovY), m_aspectRatio(aspectRatio) {}

    const float3& eye() const { return m_eye; }
    void setEye(const float3& v) { m_eye = v; }
    const float3& lookat() const { return m_lookat; }
    void setLookat(const float3& v) { m_lookat = v; }
    const float3& up() const { return m_up; }
    void setUp(const float3& v) { m_up = v; }
    const float& fovY() const { return m_fovY; }
    void setFovY(const float& v) { m_fovY = v; }
    const float& aspectRatio() const { return m_aspectRatio; }
    void setAspectRatio(const float& v) { m_aspectRatio = v; }

    // UVW is orthogonal but not orthonormal: |W| is the focal length (Camera.cpp:34-45)
    void UVWFrame(float3& U, float3& V, float3& W) const
    {
        using namespace rtgo_vec;
        W = sub(m_lookat, m_eye);
        const float wlen = length(W);
        U = normalize(cross(W, m_up));
        V = normalize(cross(U, W));
        const float vlen = wlen * tanf(0.5f * m_fovY * M_PIf / 180.0f);
        V = mul(V, vlen);
        const float ulen = vlen * m_aspectRatio;
        U = mul(U, ulen);
    }

private:
    float3 m_eye, m_lookat, m_up;
    float m_fovY, m_aspectRatio;
};
}  // namespace sutil
