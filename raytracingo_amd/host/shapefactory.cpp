// shapefactory.cpp -- factory shapes (engine/shapefactory.cpp:9-76) built from the four unit primitives.
#include "shapefactory.h"

namespace engine {
namespace host {
namespace {
using M = sutil::Matrix4x4;
ShapeFactory::Result wrap(const std::vector<Primitive>& prims, const M& model)
{
    return {std::make_shared<Shape>(prims, model), static_cast<int>(prims.size())};
}
ShapeFactory::Result one(PRIMITIVE_TYPE t, const M& model, const BasicMaterial& mat)
{
    return wrap({Primitive(t, M::identity(), mat)}, model);
}
}  // namespace

ShapeFactory::Result ShapeFactory::CreateRectangle(const M& m, const BasicMaterial& mat) const { return one(PRIMITIVE_TYPE::RECTANGLE, m, mat); }
ShapeFactory::Result ShapeFactory::CreateOpenCylinder(const M& m, const BasicMaterial& mat) const { return one(PRIMITIVE_TYPE::CYLINDER, m, mat); }
ShapeFactory::Result ShapeFactory::CreateDisk(const M& m, const BasicMaterial& mat) const { return one(PRIMITIVE_TYPE::DISK, m, mat); }
ShapeFactory::Result ShapeFactory::CreateSphere(const M& m, const BasicMaterial& mat) const { return one(PRIMITIVE_TYPE::SPHERE, m, mat); }

ShapeFactory::Result ShapeFactory::CreateClosedCylinder(const M& m, const BasicMaterial& mat) const
{
    // tube + a disk on each end (y = +1, y = -1)
    return wrap({Primitive(PRIMITIVE_TYPE::CYLINDER, M::identity(), mat),
                 Primitive(PRIMITIVE_TYPE::DISK, M::translate(make_float3(0.0f, 1.0f, 0.0f)), mat),
                 Primitive(PRIMITIVE_TYPE::DISK, M::translate(make_float3(0.0f, -1.0f, 0.0f)), mat)},
                m);
}

ShapeFactory::Result ShapeFactory::CreateCube(const M& m, const BasicMaterial& mat) const
{
    // six one-sided unit rectangles facing outwards: +x, -x, +y, -y, +z, -z (rectangles face +y in object space)
    const float3 X = make_float3(1.0f, 0.0f, 0.0f), Z = make_float3(0.0f, 0.0f, 1.0f);
    const float quarter = M_PIf / 2.0f;
    const M faces[6] = {
        M::translate(make_float3(0.5f, 0.0f, 0.0f)) * M::rotate(-quarter, Z),
        M::translate(make_float3(-0.5f, 0.0f, 0.0f)) * M::rotate(quarter, Z),
        M::translate(make_float3(0.0f, 0.5f, 0.0f)),
        M::translate(make_float3(0.0f, -0.5f, 0.0f)) * M::rotate(static_cast<float>(M_PI), Z),  // the reference passes the double M_PI here
        M::translate(make_float3(0.0f, 0.0f, 0.5f)) * M::rotate(quarter, X),
        M::translate(make_float3(0.0f, 0.0f, -0.5f)) * M::rotate(-quarter, X),
    };
    std::vector<Primitive> prims;
    for (const M& f : faces) prims.emplace_back(PRIMITIVE_TYPE::RECTANGLE, f, mat);
    return wrap(prims, m);
}

ShapeFactory::Result ShapeFactory::CreateCustom(const std::vector<Primitive>& primitives, const M& m) const { return wrap(primitives, m); }
}  // namespace host
}  // namespace engine
