#include "RayTracer.h"

#include "RayCamera.h"
#include "RayScene.h"
#include "rt_hip.h"

#include <iostream>

RayTracer::~RayTracer() {
    if (m_ctx != nullptr) rt_destroy(m_ctx);
}

bool RayTracer::initialize(glm::ivec2 resolution, Type type) {
    m_type = type;
    if (type != Type::RGB && type != Type::Spectral) {
        std::cout << "Unsupported raytracer type: " << static_cast<int>(type) << std::endl;
        ASSERT(false);
    }
    int rc = rt_create(&m_ctx, m_device);
    if (rc != RT_OK) {
        std::cout << "rt_create: " << rt_strerror(rc) << std::endl;
        return false;
    }
    changeResolution(resolution);
    return m_status == RT_OK;
}

// Uniform set of reference src/RayTracer.cpp:1436-1451, then the frames.
void RayTracer::renderToTexture(const RayCamera &camera, const RayScene &scene, u32 frames) {
    if (frames == 0) return;
    m_status = scene.bind();
    if (m_status != RT_OK) {
        std::cout << "scene commit: " << rt_strerror(m_status) << std::endl;
        return;
    }

    rt_camera cam;
    cam.fov = camera.fov;
    cam.resolution[0] = static_cast<float>(camera.resolution.x);
    cam.resolution[1] = static_cast<float>(camera.resolution.y);
    for (int k = 0; k < 3; ++k) {
        cam.position[k] = camera.position[k];
        cam.forward[k] = camera.forward[k];
        cam.right[k] = camera.right[k];
        cam.up[k] = camera.up[k];
    }
    cam.bounces = camera.bounces;
    cam.rayPerPixel = camera.rayPerPixel;

    const glm::vec3 sky = scene.getLinearSkyColor();
    const float skyv[3] = {sky.x, sky.y, sky.z};
    m_status = rt_set_frame(m_ctx, &cam, skyv, m_type == Type::RGB ? RT_TYPE_RGB : RT_TYPE_SPECTRAL);
    if (m_status == RT_OK) m_status = rt_render(m_ctx, m_frameCount, frames, 1);
    if (m_status != RT_OK) {   // nothing was rendered: the frame counter stays where it is
        std::cout << "render: " << rt_strerror(m_status) << " " << rt_last_hip_error(m_ctx) << std::endl;
        return;
    }

    if (frames & 1u) m_frameIndex = !m_frameIndex;
    m_frameCount += frames;
}

// Like the reference (src/RayTracer.cpp:1454-1465) this re-creates the frame
// storage but leaves the frame counter alone.
void RayTracer::changeResolution(glm::ivec2 resolution) {
    m_frame.width = resolution.x;
    m_frame.height = resolution.y;
    m_frame.rgba.assign(static_cast<size_t>(resolution.x) * resolution.y * 4, 0.0f);
    if (m_ctx != nullptr) {
        m_status = rt_resize(m_ctx, resolution.x, resolution.y);
    }
}

void RayTracer::reset() {
    m_frameCount = 1;
    m_frameIndex = 0;
    if (m_ctx != nullptr) rt_reset_accum(m_ctx);
}

RayFrame &RayTracer::getCurrentFrame() {
    u32 frames = 0;
    m_status = rt_readback(m_ctx, m_frame.rgba.data(), m_frame.rgba.size(), &frames);
    if (m_status != RT_OK) {
        m_frame.rgba.assign(m_frame.rgba.size(), 0.0f);
        return m_frame;
    }
    if (frames > 0) {
        const f32 n = static_cast<f32>(frames);
        for (size_t i = 0; i < m_frame.rgba.size(); i += 4) {
            m_frame.rgba[i + 0] /= n;
            m_frame.rgba[i + 1] /= n;
            m_frame.rgba[i + 2] /= n;
            m_frame.rgba[i + 3] = 1.0f;
        }
    }
    return m_frame;
}
