#include "RayEngine.h"

// Order differs from reference src/RayEngine.cpp:9-21 in one respect: the
// integrator is initialised first because it owns the backend context the
// scene's storages upload into.
bool RayEngine::initialize(const RayCamera &camera, RayTracer::Type raytracerType, int device) {
    m_camera = camera;
    m_rayTracer.setDevice(device);
    if (!m_rayTracer.initialize(camera.resolution, raytracerType)) return false;
    m_rayScene.initialize(m_rayTracer.backend());
    return true;
}

void RayEngine::changeResolution(glm::ivec2 resolution) {
    getRayTracer().changeResolution(resolution);
    m_camera.resolution = resolution;
}

void RayEngine::render(u32 frames) { m_rayTracer.renderToTexture(m_camera, m_rayScene, frames); }
