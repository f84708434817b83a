// Scene container: materials, analytic primitives, models; serialises them into
// the five float streams and hands them to gpu::Storage objects
// (reference src/RayScene.h:21-73).
#pragma once
#include "Material.h"
#include "TraceableObject.h"
#include "glm/glm.hpp"
#include "gpu/Buffer.h"
#include "gpu/Storage.h"

#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

struct rt_ctx;
struct RayCamera;

class RayScene {
private:
    glm::vec3 m_skyColor = {0.5f, 0.7f, 1.0f};
    glm::vec3 m_linearSkyColor = glm::pow(m_skyColor, glm::vec3(2.2f));

    // One serialised stream per backend slot (RT_SLOT_* 1..5; entry 0 unused): the words being
    // collected and the storage object that receives them at submit().
    struct Stream {
        std::unique_ptr<gpu::Buffer> words;
        std::unique_ptr<gpu::Storage> storage;
    };
    Stream m_streams[6];

    std::vector<Material> m_materials;
    std::vector<std::unique_ptr<TraceableObject>> m_traceableObjects;
    std::vector<std::unique_ptr<Model>> m_modelObjects;

    // Models that carry emissive triangles; they are serialised FIRST, in this
    // container's iteration order (the light sampler indexes them 0..count-1).
    std::unordered_set<i32> lightSourcesMark;

    i32 m_textureTotalSize = 0;  // words already claimed in the texture stream
    i32 lightSourcesCount = 0;
    rt_ctx *m_backend = nullptr;

    void registerModel(Model &&model);

public:
    explicit RayScene() = default;

    // backend == nullptr keeps the streams on the host only (gpu::HostStorage).
    void initialize(rt_ctx *backend = nullptr);
    // Counterpart of bindShader(gl::ShaderProgram&): names the five slots and
    // commits object/model/light counts to the backend.
    // Returns the backend's status (RT_OK, or the first failing upload / the commit's code).
    int bind() const;
    // Serialises everything and uploads.  One-shot: the buffers are never cleared.
    void submit();

    void setSkyColor(glm::vec3 skyColor);
    glm::vec3 getSkyColor() const;
    glm::vec3 getLinearSkyColor() const;
    u32 getObjectsCount() const { return static_cast<u32>(m_traceableObjects.size()); }
    u32 getModelsCount() const { return static_cast<u32>(m_modelObjects.size()); }
    i32 getLightSourcesCount() const { return lightSourcesCount; }

    void addModel(const std::string &modelPath);
    void addModel(MeshData mesh);  // in-memory mesh (procedural scenes)
    void addObject(const TraceableObject &obj, const Material &material);
    void addObject(const TraceableObject &obj, int materialIndex);

    // The reference returns GLSL text here; the HIP backend has no shader source
    // to generate, so this returns an identifying tag only.
    static std::string genShaderCode();

    // Host copies of the submitted streams (slot numbers 1..5 as in bind()).
    const std::vector<f32> &streamWords(int slot) const;
    const std::vector<std::unique_ptr<Model>> &models() const { return m_modelObjects; }
    const std::vector<Material> &materials() const { return m_materials; }
};
