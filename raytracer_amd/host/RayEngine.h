// Facade tying camera, scene and integrator together
// (reference src/RayEngine.h:14-35; the GL framebuffer/quad members are gone).
#pragma once

#include "RayCamera.h"
#include "RayScene.h"
#include "RayTracer.h"

class RayEngine {
public:
    RayEngine() = default;

    bool initialize(const RayCamera &camera, RayTracer::Type raytracerType, int device = 0);
    void changeResolution(glm::ivec2 resolution);
    // One progressive frame (what the reference's full-screen draw computes), or
    // `frames` of them traced as one batch.
    void render(u32 frames = 1);

    RayScene &getScene() { return m_rayScene; }
    RayTracer &getRayTracer() { return m_rayTracer; }
    RayCamera &getCamera() { return m_camera; }

private:
    RayCamera m_camera;
    RayScene m_rayScene;
    RayTracer m_rayTracer;
};
