// Scene contents (positions, sizes, colours, insertion ORDER — which fixes the
// primitive and material indices) follow reference src/RaySceneBuilder.cpp:6-221.
#include "RaySceneBuilder.h"
#include "RayScene.h"
#include "TraceableObject.h"
#include "glm/gtc/matrix_transform.hpp"

#include <cmath>

namespace {
const glm::vec3 kFloorQ{-5, -0.1f, 0}, kFloorU{10, 0, 0}, kFloorV{0, 0, 10};

Material emitter(glm::vec3 colour, float strength) {
    Material m;
    m.emissionColor = colour;
    m.emissionStrength = strength;
    return m;
}
}  // namespace

// 11 x 2 red spheres sweeping roughness (columns) and metallic (rows), a floor
// quad and one bright sphere light.
void RaySceneBuilder::RoughnessMetallicDemo(RayScene &scene) {
    Material ball;
    ball.albedo = glm::vec3(.65f, .05f, .05f);
    for (int col = 0; col <= 10; ++col) {
        for (int row = 0; row <= 1; ++row) {
            ball.roughness = static_cast<f32>(col / 10.0);
            ball.metallic = static_cast<f32>(row * (1 - col / 10.0));
            scene.addObject(Sphere{glm::vec3{col * 0.3 - 10 * 0.5 * 0.3, 0, 2 - row * 0.5}, 0.1f}, ball);
        }
    }
    scene.addObject(Quad{kFloorQ, kFloorU, kFloorV}, Material());
    scene.addObject(Sphere{glm::vec3{-5, 8, -15}, 1.5f}, emitter({1, 1, 1}, 100));
}

// A white sphere on a floor lit by three small coloured sphere lights.
void RaySceneBuilder::ThreeColorDemo(RayScene &scene) {
    scene.addObject(Sphere{glm::vec3{0, 0, 1}, 0.12f}, Material());
    scene.addObject(Quad{kFloorQ, kFloorU, kFloorV}, Material());

    const float l = 0.3f;
    scene.addObject(Sphere{glm::vec3{l, 0.5, 1.0 - l}, 0.03f}, emitter({1, 0.2f, 0.2f}, 140));
    scene.addObject(Sphere{glm::vec3{-l, 0.5, 1.0 - l}, 0.03f}, emitter({0.2f, 0.2f, 1}, 140));
    scene.addObject(Sphere{glm::vec3{0, 0.5, 1 + l * std::sqrt(2) - 0.1}, 0.03f}, emitter({0.2f, 1.0f, 0.2f}, 140));
}

// Order: light, left (red), right (green), floor, ceiling, back, [front, culled].
void RaySceneBuilder::BuildCornellBox(RayScene &scene, glm::vec3 pos, float boxLen, float lightLen, float emissionStrenth, bool includedFront) {
    const glm::vec3 ex{boxLen, 0, 0}, ey{0, boxLen, 0}, ez{0, 0, boxLen};

    Material lamp = emitter(glm::vec3(1.0f), emissionStrenth);
    lamp.albedo = glm::vec3(0.0f);
    scene.addObject(Quad{pos + glm::vec3{(boxLen - lightLen) * 0.5, boxLen - 1e-4, boxLen * 0.5 - lightLen}, glm::vec3{lightLen, 0, 0},
                         glm::vec3{0, 0, lightLen}},
                    lamp);

    Material wall;
    wall.albedo = glm::vec3(.65f, .05f, .05f);
    scene.addObject(Quad{pos, ey, ez}, wall);
    wall.albedo = glm::vec3(.12f, .45f, .15f);
    scene.addObject(Quad{pos + ex, ey, ez}, wall);

    const Material white;
    scene.addObject(Quad{pos, ex, ez}, white);
    scene.addObject(Quad{pos + ey, ex, ez}, white);
    scene.addObject(Quad{pos + ez, ey, ex}, white);
    if (includedFront) scene.addObject(Quad{pos, ex, ey, true}, white);
}

// Six quads of an oriented box; corners are (±size/2) * R with R = mat3(q)
// applied as a row-vector product.
void RaySceneBuilder::BuildBox(RayScene &scene, const Material &material, glm::vec3 size, glm::vec3 pos, glm::quat q) {
    const glm::mat3 R = glm::mat4_cast(q);
    glm::vec3 c[8];
    for (int k = 0; k < 8; ++k) {
        const glm::vec3 corner{(k & 1) ? -size.x : size.x, (k & 2) ? -size.y : size.y, (k & 4) ? -size.z : size.z};
        c[k] = corner * 0.5f * R;
    }
    // (origin corner, corner along u, corner along v): top, bottom, left, right, back, front.
    const int faces[6][3] = {{0, 1, 4}, {2, 3, 6}, {1, 5, 3}, {0, 2, 4}, {0, 1, 2}, {4, 6, 5}};
    for (const auto &f : faces) scene.addObject(Quad{pos + c[f[0]], c[f[1]] - c[f[0]], c[f[2]] - c[f[0]]}, material);
}
