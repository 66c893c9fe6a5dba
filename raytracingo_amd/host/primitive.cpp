// primitive.cpp -- Primitive: transform accumulation, AABB and record packing (engine/primitive.cpp:20-138).
#include "primitive.h"

namespace engine {
namespace host {
namespace {
const char* const kPrograms[4] = {"__intersection__cylinder", "__intersection__disk", "__intersection__rectangle",
                                  "__intersection__sphere"};
constexpr float kAabbEpsilon = 0.001f;   // AABB_EPSILON
constexpr float kSceneMaxBound = 50.0f;  // SCENE_MAX_BOUND: seeds of the min/max search
}  // namespace

Primitive::Primitive(PRIMITIVE_TYPE type, const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material)
    : m_type(type), m_modelMatrix(modelMatrix), m_material(material), m_intersectionProgram(kPrograms[static_cast<int>(type)])
{
}

void Primitive::Transform(const sutil::Matrix4x4& transform) { m_modelMatrix = transform * m_modelMatrix; }

rtgo_aabb Primitive::GetAabb() const
{
    // The reference pushes the corners of the unit cube through two 4x4 products (one per y face) and scans the
    // columns; corner i of a face is (sx[i], y, sz[i]).  Each coordinate is the 4-term sum seeded with 0.0f.
    static const float sx[4] = {-1.f, -1.f, 1.f, 1.f}, sz[4] = {-1.f, 1.f, -1.f, 1.f};
    float lo[3] = {kSceneMaxBound, kSceneMaxBound, kSceneMaxBound};
    float hi[3] = {-kSceneMaxBound, -kSceneMaxBound, -kSceneMaxBound};
    for (int i = 0; i < 4; ++i)
        for (int axis = 0; axis < 3; ++axis) {
            const float* row = m_modelMatrix.getData() + 4 * axis;
            float below = 0.0f, above = 0.0f;  // y = -1 face, y = +1 face
            const float c[2][4] = {{sx[i], -1.f, sz[i], 1.f}, {sx[i], 1.f, sz[i], 1.f}};
            for (int k = 0; k < 4; ++k) {
                below += row[k] * c[0][k];
                above += row[k] * c[1][k];
            }
            lo[axis] = std::fmin(std::fmin(lo[axis], below), above);
            hi[axis] = std::fmax(std::fmax(hi[axis], below), above);
        }
    rtgo_aabb bb;
    bb.minX = lo[0] - kAabbEpsilon;
    bb.minY = lo[1] - kAabbEpsilon;
    bb.minZ = lo[2] - kAabbEpsilon;
    bb.maxX = hi[0] + kAabbEpsilon;
    bb.maxY = hi[1] + kAabbEpsilon;
    bb.maxZ = hi[2] + kAabbEpsilon;
    return bb;
}

void Primitive::CopyToDevice(rtgo_prim& data) const
{
    const glm::vec3 kd = m_material.GetKd(), kr = m_material.GetKr(), le = m_material.GetLe();
    data.type = static_cast<uint32_t>(m_type);
    for (int i = 0; i < 16; ++i) data.model[i] = m_modelMatrix[i];
    data.kd[0] = kd.r; data.kd[1] = kd.g; data.kd[2] = kd.b;
    data.kr[0] = kr.r; data.kr[1] = kr.g; data.kr[2] = kr.b;
    data.Le[0] = le.r; data.Le[1] = le.g; data.Le[2] = le.b;
    data.specularity = m_material.GetSpecularity();
}
}  // namespace host
}  // namespace engine
