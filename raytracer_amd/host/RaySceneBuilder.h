// Procedural demo scenes made of analytic primitives
// (reference src/RaySceneBuilder.h:8-13).
#pragma once
#include "Material.h"

#include "glm/glm.hpp"
#include "glm/gtc/quaternion.hpp"

class RayScene;

// All members add objects and materials to `scene`; none keeps state.
struct RaySceneBuilder {
    // Five spheres of increasing roughness over a quad floor.
    static void RoughnessMetallicDemo(RayScene &scene);

    // Three coloured spheres and an emissive one.
    static void ThreeColorDemo(RayScene &scene);

    // Quad-walled box of edge `boxLen` with its floor centre at `pos` and a square quad light of
    // edge `lightLen` under the ceiling.
    static void BuildCornellBox(RayScene &scene,
                                glm::vec3 pos,
                                float boxLen,
                                float lightLen,
                                float emissionStrenth,
                                bool includedFront = false);

    // Six quads of one material: a box of `size` centred at `pos`, rotated by `q`.
    static void BuildBox(RayScene &scene,
                         const Material &material,
                         glm::vec3 size,
                         glm::vec3 pos,
                         glm::quat q);
};
