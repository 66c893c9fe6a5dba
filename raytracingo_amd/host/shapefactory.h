// shapefactory.h -- engine::host::ShapeFactory (engine/shapefactory.h:10-59): the seven factory shapes.
// Every Create* returns {shape, number of primitives in it}.
#pragma once
#include "shape.h"
#include <memory>
#include <utility>

namespace engine {
namespace host {
class ShapeFactory {
public:
    using Result = std::pair<std::shared_ptr<Shape>, int>;
    ShapeFactory() = default;
    ~ShapeFactory() = default;
    Result CreateRectangle(const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material) const;
    Result CreateOpenCylinder(const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material) const;
    Result CreateClosedCylinder(const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material) const;
    Result CreateDisk(const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material) const;
    Result CreateSphere(const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material) const;
    Result CreateCube(const sutil::Matrix4x4& modelMatrix, const BasicMaterial& material) const;
    Result CreateCustom(const std::vector<Primitive>& primitives, const sutil::Matrix4x4& modelMatrix) const;
};
}  // namespace host
}  // namespace engine
