// matrix.h -- row-major 4x4 float matrix with the subset of sutil::Matrix4x4 the scene surface uses
// (sutil/Matrix.h:344-360 operator*, :472-493 M*float4, :640-702 rotate/translate/scale).
// Own implementation; the summation order of every product is the reference's, so tables match bit for bit.
#pragma once
#include "vec.h"
#include <cstring>

namespace sutil {
class Matrix4x4 {
public:
    Matrix4x4() { std::memset(m_, 0, sizeof m_); }
    explicit Matrix4x4(const float* d) { std::memcpy(m_, d, sizeof m_); }

    static Matrix4x4 identity()
    {
        Matrix4x4 r;
        r.m_[0] = r.m_[5] = r.m_[10] = r.m_[15] = 1.0f;
        return r;
    }
    static Matrix4x4 translate(const float3& v)
    {
        Matrix4x4 r = identity();
        r.m_[3] = v.x; r.m_[7] = v.y; r.m_[11] = v.z;
        return r;
    }
    static Matrix4x4 scale(const float3& v)
    {
        Matrix4x4 r = identity();
        r.m_[0] = v.x; r.m_[5] = v.y; r.m_[10] = v.z;
        return r;
    }
    // Rodrigues form with an axis that is used as given, NOT normalised (the scenes rely on that)
    static Matrix4x4 rotate(const float radians, const float3& axis)
    {
        Matrix4x4 r = identity();
        const float s = sinf(radians), c = cosf(radians);
        const float x = axis.x, y = axis.y, z = axis.z;
        float* m = r.m_;
        m[0] = x * x + c * (1 - x * x);
        m[1] = x * y * (1 - c) - z * s;
        m[2] = z * x * (1 - c) + y * s;
        m[4] = x * y * (1 - c) + z * s;
        m[5] = y * y + c * (1 - y * y);
        m[6] = y * z * (1 - c) - x * s;
        m[8] = z * x * (1 - c) - y * s;
        m[9] = y * z * (1 - c) + x * s;
        m[10] = z * z + c * (1 - z * z);
        return r;
    }

    float* getData() { return m_; }
    const float* getData() const { return m_; }
    float& operator[](unsigned i) { return m_[i]; }
    float operator[](unsigned i) const { return m_[i]; }

private:
    float m_[16];
};

inline Matrix4x4 operator*(const Matrix4x4& a, const Matrix4x4& b)
{
    Matrix4x4 out;
    for (unsigned i = 0; i < 4; ++i)
        for (unsigned j = 0; j < 4; ++j) {
            float sum = 0.0f;
            for (unsigned k = 0; k < 4; ++k) sum += a[i * 4 + k] * b[k * 4 + j];
            out[i * 4 + j] = sum;
        }
    return out;
}

inline float4 operator*(const Matrix4x4& m, const float4& v)
{
    float4 t;
    t.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w;
    t.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * v.w;
    t.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * v.w;
    t.w = m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w;
    return t;
}
}  // namespace sutil
