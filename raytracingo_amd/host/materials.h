// materials.h -- the material constants of engine/materials.h:13-283 (same names), as one compact table.
// Argument order is BasicMaterial(kd, kr, Le, specularity); kr and Le are grey (one value for the three channels).
#pragma once
#include "basicmaterial.h"

namespace engine {
namespace host {
namespace materials {
namespace detail {
inline BasicMaterial make(float kdr, float kdg, float kdb, float k, float le, float spec)
{
    return BasicMaterial(glm::vec3(kdr, kdg, kdb), glm::vec3(k), glm::vec3(le), spec);
}
}  // namespace detail

// name                                                              kd                          kr    Le     specularity
inline const BasicMaterial blackMirror                 = detail::make(0.1f, 0.1f, 0.1f,          1.0f, 0.0f,  0.0f);
inline const BasicMaterial blue                        = detail::make(0.0f, 0.549f, 0.988f,      0.0f, 0.0f,  1.0f);
inline const BasicMaterial grey                        = detail::make(0.29f, 0.29f, 0.29f,       0.0f, 0.0f,  1.0f);
inline const BasicMaterial cream                       = detail::make(1.0f, 0.941f, 0.729f,      0.0f, 0.0f,  1.0f);
inline const BasicMaterial white                       = detail::make(0.8f, 0.8f, 0.8f,          0.3f, 0.0f,  1.0f);
// mirror_spheres
inline const BasicMaterial mirrorSpheresBlackMirror    = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  100000.0f);
inline const BasicMaterial mirrorSpheresGroundMat      = detail::make(0.9765f, 0.651f, 0.6549f,  0.7f, 0.0f,  100000.0f);
inline const BasicMaterial mirrorSpheresMetallicOrange = detail::make(0.8549f, 0.4078f, 0.0588f, 0.5f, 0.0f,  10000.0f);
inline const BasicMaterial mirrorSpheresSilver         = detail::make(0.7529f, 0.7529f, 0.7529f, 0.5f, 0.0f,  10000.0f);
// plateau
inline const BasicMaterial plateMetallicGold           = detail::make(0.83f, 0.69f, 0.22f,       0.5f, 0.0f,  10000.0f);
inline const BasicMaterial platePurple                 = detail::make(1.0f, 0.0f, 1.0f,          0.5f, 0.0f,  1000.0f);
inline const BasicMaterial plateCyan                   = detail::make(0.1f, 1.0f, 1.0f,          0.5f, 0.0f,  5000.0f);
inline const BasicMaterial platePrettyGreen            = detail::make(0.16f, 0.83f, 0.18f,       0.5f, 0.0f,  1000.0f);
inline const BasicMaterial plateDarkRed                = detail::make(0.5f, 0.0f, 0.0f,          0.5f, 0.0f,  100.0f);
inline const BasicMaterial plateYellow                 = detail::make(1.0f, 1.0f, 0.0f,          0.7f, 0.0f,  0.0f);
inline const BasicMaterial plateLight                  = detail::make(0.0f, 0.0f, 0.0f,          0.0f, 15.f,  0.0f);
// cornell
inline const BasicMaterial cornellMirror               = detail::make(0.0f, 0.0f, 0.0f,          1.0f, 0.0f,  0.0f);
inline const BasicMaterial cornellWhite                = detail::make(0.8f, 0.8f, 0.8f,          0.3f, 0.0f,  1.0f);
inline const BasicMaterial cornellBlue                 = detail::make(0.0f, 0.0f, 1.0f,          0.3f, 0.0f,  1.0f);
inline const BasicMaterial cornellRed                  = detail::make(1.0f, 0.0f, 0.0f,          0.3f, 0.0f,  1.0f);
inline const BasicMaterial cornellLight                = detail::make(0.0f, 0.0f, 0.0f,          0.0f, 15.f,  1.0f);
// soft_mirrors: eight mirrors that differ only by the width of their glossy lobe
inline const BasicMaterial softMirrorsMirror0          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  500000.0f);
inline const BasicMaterial softMirrorsMirror1          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  100000.0f);
inline const BasicMaterial softMirrorsMirror2          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  50000.0f);
inline const BasicMaterial softMirrorsMirror3          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  10000.0f);
inline const BasicMaterial softMirrorsMirror4          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  5000.0f);
inline const BasicMaterial softMirrorsMirror5          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  1000.0f);
inline const BasicMaterial softMirrorsMirror6          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  500.0f);
inline const BasicMaterial softMirrorsMirror7          = detail::make(0.05f, 0.05f, 0.05f,       1.0f, 0.0f,  100.0f);
// lights and walls of checkered / balls / window
inline const BasicMaterial CheckeredLight              = detail::make(0.0f, 0.0f, 0.0f,          0.0f, 12.0f, 1.0f);
inline const BasicMaterial BallsLight                  = detail::make(0.0f, 0.0f, 0.0f,          0.0f, 12.0f, 1.0f);
inline const BasicMaterial WindowLight                 = detail::make(0.0f, 0.0f, 0.0f,          0.0f, 12.0f, 1.0f);
inline const BasicMaterial windowWhite                 = detail::make(0.9f, 0.9f, 0.9f,          0.5f, 0.0f,  100.0f);
}  // namespace materials
}  // namespace host
}  // namespace engine
