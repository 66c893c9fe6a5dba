// scene.h -- engine::host::Scene (engine/scene.h:17-155): one of the eight procedural scenes + its camera.
// Public surface identical to the reference's; the renderer only needs GetShapes / GetSurfaceLights / GetCamera /
// GetBackgroundColor / GetNbObjects.
#pragma once
#include "camera.h"
#include "light.h"
#include "shapefactory.h"
#include <memory>
#include <vector>

namespace engine {
namespace host {
enum class SceneModel { SLIDE, CORNELL, PLATE, WINDOW, BALLS, CHECKERED, MIRROR_SPHERES, SOFT_MIRRORS };

class Scene {
public:
    Scene(SceneModel sceneModel, const unsigned int& camWidth, const unsigned int& camHeight);
    ~Scene() = default;

    std::vector<std::shared_ptr<Shape>> GetShapes() const { return m_shapes; }
    std::vector<SurfaceLight> GetSurfaceLights() const { return m_surfaceLights; }
    std::shared_ptr<sutil::Camera> GetCamera() const { return m_camera; }
    unsigned int GetCameraWidth() const { return m_cameraWidth; }
    unsigned int GetCameraHeight() const { return m_cameraHeight; }
    glm::vec3 GetBackgroundColor() const { return m_backgroundColor; }
    int GetNbObjects() const { return m_nbObjects; }

private:
    std::vector<std::shared_ptr<Shape>> m_shapes;
    std::vector<SurfaceLight> m_surfaceLights;
    std::shared_ptr<sutil::Camera> m_camera;
    glm::vec3 m_backgroundColor;
    ShapeFactory m_factory;
    int m_nbObjects;
    unsigned int m_cameraWidth, m_cameraHeight;
    SceneModel m_sceneModel;

    void SetupCamera();
    void SetupObjects();
    void CreateFunPlate();
    void CreateSlide();
    void CreateWindowScene();
    void CreateCheckeredFloor();
    void CreateBalls();
    void CreateCornellBox();
    void CreateMirrorSpheres();
    void CreateSoftMirrors();

    void AddObject(const ShapeFactory::Result& object);
    /// emissive rectangle that is both geometry and the scene's area light
    void AddLight(const sutil::Matrix4x4& model, const BasicMaterial& material, const glm::vec3& color, float falloff);
    /// the three-cylinder / three-sphere "triangle" ornament shared by three scenes
    std::vector<Primitive> TrianglePrimitives(const BasicMaterial& material) const;
    sutil::Matrix4x4 GetScale(float sx, float sy, float sz) const;
    sutil::Matrix4x4 GetTranslate(float tx, float ty, float tz) const;
    sutil::Matrix4x4 GetRotate(float angleRad, float vx, float vy, float vz) const;
};
}  // namespace host
}  // namespace engine
