// basicmaterial.h -- engine::host::BasicMaterial (engine/basicmaterial.h:10-57): kd, kr, Le, specularity.
#pragma once
#include "vec.h"

namespace engine {
namespace host {
class BasicMaterial {
public:
    BasicMaterial() : m_kd(0.0f), m_kr(0.0f), m_Le(0.0f), m_specularity(0.0f) {}
    /// \param kd diffuse colour  \param kr reflected proportion  \param le emission  \param specularity lobe exponent
    BasicMaterial(const glm::vec3& kd, const glm::vec3& kr, const glm::vec3& le, const float& specularity)
        : m_kd(kd), m_kr(kr), m_Le(le), m_specularity(specularity) {}
    virtual ~BasicMaterial() = default;

    glm::vec3 GetKd() const { return m_kd; }
    glm::vec3 GetKr() const { return m_kr; }
    glm::vec3 GetLe() const { return m_Le; }
    float GetSpecularity() const { return m_specularity; }

private:
    glm::vec3 m_kd, m_kr, m_Le;
    float m_specularity;
};
}  // namespace host
}  // namespace engine
