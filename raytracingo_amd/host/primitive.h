// primitive.h -- engine::host::Primitive (engine/primitive.h:16-117): one analytic unit shape under a model matrix.
#pragma once
#include "basicmaterial.h"
#include "matrix.h"
#include <rtgo.h>
#include <string>

namespace engine {
namespace host {
/// order matters: it is the `type` field of rtgo_prim (RTGO_CYLINDER.. in include/rtgo.h)
enum class PRIMITIVE_TYPE { CYLINDER, DISK, RECTANGLE, SPHERE };

class Primitive {
public:
    Primitive() = delete;
    Primitive(PRIMITIVE_TYPE type, const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material);
    ~Primitive() = default;

    /// Fill one hit-group record (the reference fills device::HitGroupData; here the ABI's rtgo_prim)
    void CopyToDevice(rtgo_prim& data) const;
    /// m = transform * m
    void Transform(const sutil::Matrix4x4& transform);
    /// World-space box of the transformed [-1,1]^3 cube, clipped-seeded at +-50 and padded by 1e-3
    rtgo_aabb GetAabb() const;
    /// Name of the OptiX intersection program the reference binds; kept for source compatibility
    const char* GetIntersectionProgram() const { return m_intersectionProgram.c_str(); }
    sutil::Matrix4x4 GetModelMatrix() const { return m_modelMatrix; }
    PRIMITIVE_TYPE GetType() const { return m_type; }
    BasicMaterial GetMaterial() const { return m_material; }

private:
    PRIMITIVE_TYPE m_type;
    sutil::Matrix4x4 m_modelMatrix;
    BasicMaterial m_material;
    std::string m_intersectionProgram;
};
}  // namespace host
}  // namespace engine
