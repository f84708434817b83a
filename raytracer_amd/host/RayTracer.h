// Integrator front end: owns the backend context, the progressive frame
// counter and the accumulated frame (reference src/RayTracer.h:15-43).
#pragma once
#include "glm/glm.hpp"
#include "util.h"

#include <vector>

struct rt_ctx;
struct RayCamera;
class RayScene;

// Host-readable stand-in for the gl::Texture2D the reference returns from
// getCurrentFrame(): RGBA32F, row 0 at the bottom.
struct RayFrame {
    i32 width = 0, height = 0;
    std::vector<f32> rgba;  // progressive mean, filled by RayTracer::download()
    i32 getWidth() const { return width; }
    i32 getHeight() const { return height; }
};

class RayTracer {
public:
    enum class Type { RGB, Spectral };

private:
    rt_ctx *m_ctx = nullptr;
    RayFrame m_frame;
    u32 m_frameCount = 1;  // value of the frameCount uniform for the NEXT frame
    i32 m_frameIndex = 0;
    Type m_type = Type::RGB;
    int m_device = 0;
    int m_status = 0;      // backend status of the last call that reached it

public:
    explicit RayTracer() = default;
    ~RayTracer();
    RayTracer(const RayTracer &) = delete;
    RayTracer &operator=(const RayTracer &) = delete;

    void setDevice(int device) { m_device = device; }
    // Creates the backend context and the accumulation frame.  false when no
    // device/back end is available (the reference returns false on shader errors).
    bool initialize(glm::ivec2 resolution, Type type);
    // Sets the per-frame uniforms and queues `frames` progressive frames (the
    // reference draws one per call; several frames in one call are traced as one
    // batch and give the same accumulated image as that many single calls).
    void renderToTexture(const RayCamera &camera, const RayScene &scene, u32 frames = 1);
    void changeResolution(glm::ivec2 resolution);
    void reset();

    Type getType() { return m_type; }
    // Synchronises, downloads and returns the progressive mean.
    RayFrame &getCurrentFrame();
    RayFrame &getPreviousFrame() { return getCurrentFrame(); }
    u32 getFrameCount() const { return m_frameCount; }
    rt_ctx *backend() const { return m_ctx; }
    // 0, or the rt_status of the last backend call that failed (device out of memory, inconsistent
    // scene streams ...).  The reference traps on a GL error; an embedding process gets a code here.
    int status() const { return m_status; }
};
