// scene.cpp -- the eight procedural scenes of RayTracinGO (engine/scene.cpp:29-671), rebuilt from their geometric
// description.  What must hold (tests/test_host_scene.py): the flattened primitive tables -- order, model matrices,
// materials, boxes, light frame, camera -- equal the oracle's restatement of the reference bit for bit, because they
// ARE the benchmark inputs.  Random placements draw rnd() in source order, left to right.
#include "scene.h"
#include "materials.h"
#include "random.h"

namespace engine {
namespace host {
namespace {
using M = sutil::Matrix4x4;
constexpr float kPi = M_PIf;
const float3 kX = {1.0f, 0.0f, 0.0f}, kY = {0.0f, 1.0f, 0.0f}, kZ = {0.0f, 0.0f, 1.0f};
}  // namespace

Scene::Scene(SceneModel sceneModel, const unsigned int& camWidth, const unsigned int& camHeight)
    : m_nbObjects(0), m_cameraWidth(camWidth), m_cameraHeight(camHeight), m_sceneModel(sceneModel)
{
    SetupObjects();
    SetupCamera();
}

M Scene::GetTranslate(float tx, float ty, float tz) const { return M::translate(make_float3(tx, ty, tz)); }
M Scene::GetScale(float sx, float sy, float sz) const { return M::scale(make_float3(sx, sy, sz)); }
M Scene::GetRotate(float a, float vx, float vy, float vz) const { return M::rotate(a, make_float3(vx, vy, vz)); }

void Scene::AddObject(const ShapeFactory::Result& object)
{
    m_shapes.push_back(object.first);
    m_nbObjects += object.second;
}

void Scene::AddLight(const M& model, const BasicMaterial& material, const glm::vec3& color, float falloff)
{
    const ShapeFactory::Result obj = m_factory.CreateRectangle(model, material);
    const Primitive prim = obj.first->GetPrimitives()[0];
    m_surfaceLights.push_back(SurfaceLight(prim.GetType(), prim.GetModelMatrix(), color, falloff));
    AddObject(obj);
}

std::vector<Primitive> Scene::TrianglePrimitives(const BasicMaterial& mat) const
{
    // an equilateral triangle of side 2 in the xy plane: three thin cylinders for the edges, three balls on the corners
    const float c = cosf(kPi / 3.0f), s = sinf(kPi / 3.0f);
    const M thin = GetScale(0.4f, 1.0f, 0.4f), ball = GetScale(0.4f, 0.4f, 0.4f);
    return {
        Primitive(PRIMITIVE_TYPE::CYLINDER, GetTranslate(c, s, 0.0f) * GetRotate(kPi / 6.0f, 0.0f, 0.0f, 1.0f) * thin, mat),
        Primitive(PRIMITIVE_TYPE::CYLINDER, GetTranslate(-c, s, 0.0f) * GetRotate(-kPi / 6.0f, 0.0f, 0.0f, 1.0f) * thin, mat),
        Primitive(PRIMITIVE_TYPE::CYLINDER, GetRotate(kPi / 2.0f, 0.0f, 0.0f, 1.0f) * thin, mat),
        Primitive(PRIMITIVE_TYPE::SPHERE, GetTranslate(-1.0f, 0.0f, 0.0f) * ball, mat),
        Primitive(PRIMITIVE_TYPE::SPHERE, GetTranslate(1.0f, 0.0f, 0.0f) * ball, mat),
        Primitive(PRIMITIVE_TYPE::SPHERE, GetTranslate(0.0f, sqrtf(3.0f), 0.0f) * ball, mat),
    };
}

// ---------------------------------------------------------------------------------------------------------------------
// A box room seen from +z: walls are one-sided unit rectangles (normal +y in object space) turned to face inwards.
// `sx, sy, sz` are the room's extents along x, y(height) and z; each wall is scaled by the two extents it spans.
namespace {
struct Room {
    M back, front, left, right, top, bottom;
};
Room make_room(float halfX, float halfY, float halfZ, float sx, float sy, float sz)
{
    const auto T = [](float x, float y, float z) { return M::translate(make_float3(x, y, z)); };
    const auto S = [](float x, float y, float z) { return M::scale(make_float3(x, y, z)); };
    Room r;
    r.back = T(0.0f, 0.0f, -halfZ) * M::rotate(kPi / 2.0f, kX) * S(sx, 1.0f, sy);
    r.front = T(0.0f, 0.0f, halfZ) * M::rotate(-kPi / 2.0f, kX) * S(sx, 1.0f, sy);
    r.left = T(-halfX, 0.0f, 0.0f) * M::rotate(-kPi / 2.0f, kZ) * S(sy, 1.0f, sz);
    r.right = T(halfX, 0.0f, 0.0f) * M::rotate(kPi / 2.0f, kZ) * S(sy, 1.0f, sz);
    r.top = T(0.0f, halfY, 0.0f) * M::rotate(kPi, kZ) * S(sx, 1.0f, sz);
    r.bottom = T(0.0f, -halfY, 0.0f) * S(sx, 1.0f, sz);
    return r;
}
}  // namespace

void Scene::CreateCornellBox()
{
    using namespace materials;
    const Room room = make_room(4.0f, 4.0f, 4.0f, 8.0f, 8.0f, 8.0f);
    AddObject(m_factory.CreateRectangle(room.back, cornellWhite));
    AddObject(m_factory.CreateRectangle(room.front, cornellWhite));
    AddObject(m_factory.CreateRectangle(room.left, cornellRed));
    AddObject(m_factory.CreateRectangle(room.right, cornellBlue));
    AddObject(m_factory.CreateRectangle(room.top, cornellWhite));
    AddObject(m_factory.CreateRectangle(room.bottom, cornellWhite));
    // short box front right, tall box back left
    AddObject(m_factory.CreateCube(GetTranslate(1.3f, -3.0f, 1.3f) * GetRotate(-kPi / 6.0f, 0.0f, 1.0f, 0.0f) * GetScale(2.0f, 2.0f, 2.0f), cornellWhite));
    AddObject(m_factory.CreateCube(GetTranslate(-1.3f, -2.0f, -1.3f) * GetRotate(kPi / 8.0f, 0.0f, 1.0f, 0.0f) * GetScale(2.0f, 4.0f, 2.0f), cornellWhite));
    AddLight(GetTranslate(0.0f, 3.95f, 0.0f) * GetRotate(kPi, 1.0f, 0.0f, 0.0f) * GetScale(4.0f, 1.0f, 4.0f), cornellLight, glm::vec3(1.0f), 0.0f);
}

void Scene::CreateMirrorSpheres()
{
    using namespace materials;
    const Room room = make_room(4.0f, 4.0f, 4.0f, 8.0f, 8.0f, 8.0f);
    // note the insertion order: bottom comes before top and right last
    AddObject(m_factory.CreateRectangle(room.back, mirrorSpheresBlackMirror));
    AddObject(m_factory.CreateRectangle(room.front, mirrorSpheresBlackMirror));
    AddObject(m_factory.CreateRectangle(room.left, mirrorSpheresBlackMirror));
    AddObject(m_factory.CreateRectangle(room.bottom, mirrorSpheresGroundMat));
    AddObject(m_factory.CreateRectangle(room.top, mirrorSpheresBlackMirror));
    AddObject(m_factory.CreateRectangle(room.right, mirrorSpheresBlackMirror));
    AddObject(m_factory.CreateSphere(GetTranslate(1.0f, 1.0f, -1.0f), mirrorSpheresSilver));
    AddObject(m_factory.CreateSphere(M::identity(), mirrorSpheresMetallicOrange));
    AddLight(GetTranslate(0.0f, 3.95f, 0.0f) * GetRotate(kPi, 1.0f, 0.0f, 0.0f) * GetScale(4.0f, 1.0f, 4.0f), cornellLight, glm::vec3(1.0f), 0.1f);
}

void Scene::CreateSoftMirrors()
{
    using namespace materials;
    AddObject(m_factory.CreateCustom(TrianglePrimitives(plateMetallicGold), GetTranslate(0.0f, -1.5f, 0.0f)));
    // eight mirrors on a circle of radius 5 around the ornament, sharper to blurrier going round
    const BasicMaterial* mirrors[8] = {&softMirrorsMirror0, &softMirrorsMirror1, &softMirrorsMirror2, &softMirrorsMirror3,
                                       &softMirrorsMirror4, &softMirrorsMirror5, &softMirrorsMirror6, &softMirrorsMirror7};
    const float turn[8] = {0.0f, kPi / 4.0f, kPi / 2.0f, 3.0f * kPi / 4.0f, kPi, 5.0f * kPi / 4.0f, 3.0f * kPi / 2.0f, 7.0f * kPi / 4.0f};
    const ShapeFactory::Result first =
        m_factory.CreateRectangle(GetTranslate(0.0f, 0.0f, -5.0f) * GetRotate(kPi / 2.0f, 1.0f, 0.0f, 0.0f) * GetScale(4.0f, 1.0f, 4.0f), *mirrors[0]);
    AddObject(first);
    const M base = first.first->GetPrimitives()[0].GetModelMatrix();
    for (int k = 1; k < 8; ++k) AddObject(m_factory.CreateRectangle(GetRotate(turn[k], 0.0f, 1.0f, 0.0f) * base, *mirrors[k]));
    AddObject(m_factory.CreateDisk(GetTranslate(0.0f, -2.0f, 0.0f) * GetScale(4.0f, 1.0f, 4.0f), platePurple));
    AddLight(GetTranslate(0.0f, 6.0f, 0.0f) * GetRotate(kPi, 1.0f, 0.0f, 0.0f) * GetScale(4.0f, 1.0f, 4.0f), cornellLight, cornellLight.GetLe(), 0.0f);
}

void Scene::CreateFunPlate()
{
    using namespace materials;
    ShapeFactory::Result tri = m_factory.CreateCustom(TrianglePrimitives(plateMetallicGold), M::identity());
    tri.first->Transform(GetRotate(-kPi / 2.0f, 1.0f, 0.0f, 0.0f));
    tri.first->Transform(GetTranslate(0.0f, 2.5f, 0.5f));
    AddObject(tri);
    ShapeFactory::Result plate = m_factory.CreateDisk(GetScale(4.0f, 1.0f, 4.0f), platePurple);
    plate.first->Transform(GetTranslate(0.0f, -1.0f, 0.0f));
    AddObject(plate);
    AddObject(m_factory.CreateSphere(GetTranslate(-2.5f, 1.0f, -0.5f), plateCyan));
    ShapeFactory::Result egg = m_factory.CreateSphere(GetScale(1.0f, 2.0f, 1.0f), platePrettyGreen);
    egg.first->Transform(GetTranslate(1.0f, 1.0f, -2.5f));
    AddObject(egg);
    AddObject(m_factory.CreateClosedCylinder(GetTranslate(-0.5f, 0.1f, 1.0f), plateDarkRed));
    ShapeFactory::Result die = m_factory.CreateCube(GetRotate(kPi / 4.0f, 1.0f, 1.0f, 1.0f), plateYellow);
    die.first->Transform(GetTranslate(2.0f, 0.25f, 0.5f));
    AddObject(die);
    AddLight(GetTranslate(0.0f, 6.0f, 0.0f) * GetRotate(kPi, 1.0f, 0.0f, 0.0f) * GetScale(4.0f, 1.0f, 4.0f), cornellLight, cornellLight.GetLe(), 0.0f);
}

void Scene::CreateSlide()
{
    using namespace materials;
    // a long strip with one specimen of every shape every few units along -x
    AddObject(m_factory.CreateRectangle(GetTranslate(0.0f, -1.0f, 0.0f) * GetScale(150.0f, 1.0f, 8.0f), grey));
    AddObject(m_factory.CreateSphere(GetTranslate(12.0f, 0.0f, 0.0f), plateCyan));
    AddObject(m_factory.CreateCube(GetTranslate(6.0f, 0.0f, 0.0f) * GetScale(2.0f, 2.0f, 2.0f), plateCyan));
    AddObject(m_factory.CreateClosedCylinder(GetTranslate(0.0f, 0.0f, 0.0f), plateCyan));
    AddObject(m_factory.CreateDisk(GetTranslate(-6.0f, 0.0f, 0.0f), plateCyan));
    AddObject(m_factory.CreateRectangle(GetTranslate(-12.0f, 0.0f, 0.0f) * GetScale(2.0f, 1.0f, 2.0f), plateCyan));
    AddObject(m_factory.CreateSphere(GetTranslate(-20.0f, 2.0f, 0.0f) * GetRotate(kPi / 2.0f, 1.0f, 1.0f, .0f) * GetScale(3.0f, 2.0f, 2.0f), cornellBlue));
    AddObject(m_factory.CreateCube(GetTranslate(-30.0f, 2.0f, 0.0f) * GetRotate(kPi / 4.0f, 0.f, 1.f, 1.f) * GetScale(2.0f, 2.0f, 2.0f), platePrettyGreen));
    AddObject(m_factory.CreateOpenCylinder(GetTranslate(-40.0f, 0.3f, 0.0f) * GetRotate(kPi / 2.0f, 1.f, 0.f, 0.f) * GetScale(1.0f, 2.0f, 1.0f), plateMetallicGold));
    ShapeFactory::Result tri = m_factory.CreateCustom(TrianglePrimitives(cornellRed), M::identity());
    tri.first->Transform(GetScale(1.5f, 1.5f, 1.5f));
    tri.first->Transform(GetRotate(-kPi / 6.0f, 1.0f, 0.0f, 0.0f));
    tri.first->Transform(GetTranslate(-50.0f, 1.2f, 0.7f));
    AddObject(tri);
    AddObject(m_factory.CreateCube(GetTranslate(-60.0f, 2.5f, 0.0f) * GetRotate(kPi / 4.0f, 0.f, 1.f, 1.f) * GetScale(1.0f, 6.0f, 0.4f), cream));
    // a near-point light far above
    AddLight(GetTranslate(0.0f, 100.0f, 10.0f) * GetRotate(kPi, 1.0f, 0.0f, 0.0f) * GetScale(0.001f, 1.0f, 0.001f), cornellLight, glm::vec3(1.0f), 0.0f);
}

void Scene::CreateWindowScene()
{
    using namespace materials;
    const Room room = make_room(8.0f, 4.0f, 8.0f, 16.0f, 8.0f, 16.0f);
    AddObject(m_factory.CreateRectangle(room.back, cornellRed));
    AddObject(m_factory.CreateRectangle(room.front, windowWhite));
    AddObject(m_factory.CreateRectangle(room.left, cornellRed));
    AddObject(m_factory.CreateRectangle(room.right, cornellRed));
    AddObject(m_factory.CreateRectangle(room.top, windowWhite));
    AddObject(m_factory.CreateRectangle(room.bottom, windowWhite));
    AddObject(m_factory.CreateSphere(GetTranslate(2.0f, -3.0, -5.5f) * GetScale(1.0f, 1.0f, 1.0f), platePrettyGreen));
    AddObject(m_factory.CreateSphere(GetTranslate(5.5f, -3.0, -5.5f) * GetScale(1.0f, 1.0f, 1.0f), cornellBlue));
    AddObject(m_factory.CreateCube(GetTranslate(-1.2f, 0.0f, -7.0f) * GetScale(0.5f, 8.0f, 2.0f), cornellRed));
    // the "window": a bright panel just in front of the back wall
    AddLight(GetTranslate(-5.0f, 0.0f, -7.99f) * GetRotate(kPi / 2.0f, 1.0f, 0.0f, 0.0f) * GetScale(4.0f, 1.0f, 4.0f), WindowLight, glm::vec3(1.0f), 0.02f);
}

void Scene::CreateCheckeredFloor()
{
    using namespace materials;
    const Room room = make_room(8.0f, 4.0f, 8.0f, 16.0f, 8.0f, 16.0f);
    // 8 x 8 cubes of side 2 whose tops sit at a random height in [-4, -3): red where i + j is even
    unsigned int seed = tea<16>(12, 1234567);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const float lift = rnd(seed);
            const BasicMaterial& mat = ((i + j) % 2 == 0) ? cornellRed : cornellBlue;
            AddObject(m_factory.CreateCube(GetTranslate(-7.0f + (i * 2.0f), -5.0f + lift, -7.0f + (j * 2.0f)) * GetScale(2.0f, 2.0f, 2.0f), mat));
        }
    AddObject(m_factory.CreateRectangle(room.back, windowWhite));
    AddObject(m_factory.CreateRectangle(room.front, windowWhite));
    AddObject(m_factory.CreateRectangle(room.left, windowWhite));
    AddObject(m_factory.CreateRectangle(room.right, windowWhite));
    AddObject(m_factory.CreateRectangle(room.top, windowWhite));
    AddLight(GetTranslate(0.0f, 3.96f, 0.0f) * GetRotate(kPi, 1.0f, 0.0f, 0.0f) * GetScale(6.0f, 1.0f, 6.0f), CheckeredLight, glm::vec3(1.0f), 0.01f);
}

void Scene::CreateBalls()
{
    using namespace materials;
    const Room room = make_room(8.0f, 4.0f, 8.0f, 16.0f, 8.0f, 16.0f);
    unsigned int seed = tea<16>(12, 1234567);
    AddObject(m_factory.CreateRectangle(room.bottom, windowWhite));
    // 16 x 16 balls of radius 1/4 on a jittered unit grid, random height, one of five materials
    const BasicMaterial* palette[5] = {&plateMetallicGold, &plateCyan, &platePurple, &platePrettyGreen, &windowWhite};
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            const float jx = rnd(seed);    // four draws, in the order they appear in the reference's expression
            const float hy = rnd(seed);
            const float jz = rnd(seed);
            const float pick = rnd(seed);
            const float x = -7.5f + (i * 1.0f) + (-0.2f + (0.4f * jx));
            const float y = -3.5f + (6.0f * hy);
            const float z = -7.5f + (j * 1.0f) + (-0.2f + (0.4f * jz));
            AddObject(m_factory.CreateSphere(GetTranslate(x, y, z) * GetScale(0.25f, 0.25f, 0.25f), *palette[(int)(5 * pick)]));
        }
    AddObject(m_factory.CreateRectangle(room.back, windowWhite));
    AddObject(m_factory.CreateRectangle(room.front, windowWhite));
    AddObject(m_factory.CreateRectangle(room.left, windowWhite));
    AddObject(m_factory.CreateRectangle(room.right, windowWhite));
    AddObject(m_factory.CreateRectangle(room.top, windowWhite));
    AddLight(GetTranslate(0.0f, 3.96f, 0.0f) * GetRotate(kPi, 1.0f, 0.0f, 0.0f) * GetScale(6.0f, 1.0f, 6.0f), BallsLight, glm::vec3(1.0f), 0.01f);
}

void Scene::SetupObjects()
{
    switch (m_sceneModel) {
    case SceneModel::CORNELL: CreateCornellBox(); break;
    case SceneModel::SLIDE: CreateSlide(); break;
    case SceneModel::MIRROR_SPHERES: CreateMirrorSpheres(); break;
    case SceneModel::PLATE: CreateFunPlate(); break;
    case SceneModel::WINDOW: CreateWindowScene(); break;
    case SceneModel::CHECKERED: CreateCheckeredFloor(); break;
    case SceneModel::BALLS: CreateBalls(); break;
    case SceneModel::SOFT_MIRRORS: CreateSoftMirrors(); break;
    }
}

void Scene::SetupCamera()
{
    // every scene is viewed from (0,0,14) towards the origin with a 60 degree vertical field of view on black
    m_backgroundColor = glm::vec3(0.0f);
    m_camera.reset(new sutil::Camera(make_float3(0.0f, 0.0f, 14.0f), make_float3(0.0f, 0.0f, 0.0f), make_float3(0.0f, 1.0f, 0.0f), 60.0f,
                                     static_cast<float>(m_cameraWidth) / static_cast<float>(m_cameraHeight)));
}
}  // namespace host
}  // namespace engine
