// shape.h -- engine::host::Shape (engine/shape.h:12-36): a rigid group of primitives.
#pragma once
#include "primitive.h"
#include <vector>

namespace engine {
namespace host {
class Shape {
public:
    Shape() = default;
    /// every primitive's matrix is pre-multiplied by modelMatrix
    Shape(const std::vector<Primitive>& primitives, const sutil::Matrix4x4& modelMatrix) : m_primitives(primitives) { Transform(modelMatrix); }
    virtual ~Shape() = default;
    virtual void Transform(const sutil::Matrix4x4& transform)
    {
        for (Primitive& p : m_primitives) p.Transform(transform);
    }
    std::vector<Primitive> GetPrimitives() const { return m_primitives; }

protected:
    std::vector<Primitive> m_primitives;
};
}  // namespace host
}  // namespace engine
