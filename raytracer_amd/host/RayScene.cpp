#include "RayScene.h"

#include "gpu/FloatBuffer.h"
#include "gpu/HipStorage.h"
#include "gpu/HostStorage.h"
#include "rt_hip.h"

#include <utility>

namespace {
std::unique_ptr<gpu::Storage> makeStorage(rt_ctx *backend, int slot) {
    if (backend != nullptr) return std::make_unique<gpu::HipStorage>(backend, slot);
    return std::make_unique<gpu::HostStorage>();
}
const std::vector<f32> &wordsOf(const gpu::Storage &s) {
    if (auto *hip = dynamic_cast<const gpu::HipStorage *>(&s)) return hip->words();
    return dynamic_cast<const gpu::HostStorage &>(s).words();
}
}  // namespace

void RayScene::initialize(rt_ctx *backend) {
    m_backend = backend;
    for (int slot = RT_SLOT_OBJECTS; slot <= RT_SLOT_TEXTURES; ++slot) {
        m_streams[slot].words = std::make_unique<gpu::FloatBuffer>();
        m_streams[slot].storage = makeStorage(backend, slot);
    }

    // Material 0 is the default material (reference src/RayScene.cpp:25-27).
    m_materials.push_back(Material());
}

int RayScene::bind() const {
    for (int slot = RT_SLOT_OBJECTS; slot <= RT_SLOT_TEXTURES; ++slot) {
        m_streams[slot].storage->bindToUnit(slot);
        if (m_streams[slot].storage->status() != RT_OK) return m_streams[slot].storage->status();
    }
    if (m_backend == nullptr) return RT_OK;
    return rt_commit_scene(m_backend, static_cast<int>(getObjectsCount()), static_cast<int>(getModelsCount()), lightSourcesCount);
}

// Stream order and layouts follow reference src/RayScene.cpp:54-101.
void RayScene::submit() {
    for (const auto &prim : m_traceableObjects) {
        m_streams[RT_SLOT_OBJECTS].words->push(static_cast<i32>(prim->getType()));
        m_streams[RT_SLOT_OBJECTS].words->push(static_cast<i32>(prim->getMaterialIndex()));
        prim->serialize(*m_streams[RT_SLOT_OBJECTS].words);
    }
    m_streams[RT_SLOT_OBJECTS].storage->upload(*m_streams[RT_SLOT_OBJECTS].words);

    auto emitModel = [this](const Model &model) {
        model.info.serialize(*m_streams[RT_SLOT_MODEL_INFO].words);
        // 4th info word: where this model's blob starts, in words.
        m_streams[RT_SLOT_MODEL_INFO].words->push(static_cast<i32>(m_streams[RT_SLOT_MODELS].words->size() / sizeof(f32)));
        model.serialize(*m_streams[RT_SLOT_MODELS].words);
    };
    for (i32 key : lightSourcesMark) emitModel(*m_modelObjects[key]);
    for (i32 i = 0; i < static_cast<i32>(m_modelObjects.size()); ++i)
        if (lightSourcesMark.find(i) == lightSourcesMark.end()) emitModel(*m_modelObjects[i]);
    m_streams[RT_SLOT_MODELS].storage->upload(*m_streams[RT_SLOT_MODELS].words);
    m_streams[RT_SLOT_MODEL_INFO].storage->upload(*m_streams[RT_SLOT_MODEL_INFO].words);

    for (const auto &model : m_modelObjects)
        for (const auto &tex : model->meshData.textures) tex.serialize(*m_streams[RT_SLOT_TEXTURES].words);
    m_streams[RT_SLOT_TEXTURES].storage->upload(*m_streams[RT_SLOT_TEXTURES].words);

    for (const auto &mat : m_materials) mat.serialize(*m_streams[RT_SLOT_MATERIALS].words);
    m_streams[RT_SLOT_MATERIALS].storage->upload(*m_streams[RT_SLOT_MATERIALS].words);
}

void RayScene::setSkyColor(glm::vec3 skyColor) {
    m_skyColor = skyColor;
    m_linearSkyColor = glm::pow(m_skyColor, glm::vec3(2.2f));
}
glm::vec3 RayScene::getSkyColor() const { return m_skyColor; }
glm::vec3 RayScene::getLinearSkyColor() const { return m_linearSkyColor; }

void RayScene::addModel(const std::string &modelPath) {
    Model loaded{modelPath};
    if (loaded.meshData.vertices.empty()) return;  // failed loads are skipped silently
    registerModel(std::move(loaded));
}

void RayScene::addModel(MeshData mesh) {
    if (mesh.vertices.empty()) return;
    registerModel(Model{std::move(mesh)});
}

// Index fix-ups of reference src/RayScene.cpp:124-190.
void RayScene::registerModel(Model &&incoming) {
    if (incoming.meshData.lightSourcesCount > 0) {
        lightSourcesCount++;
        lightSourcesMark.insert(static_cast<i32>(m_modelObjects.size()));
    }
    m_modelObjects.push_back(std::make_unique<Model>(std::move(incoming)));
    MeshData &mesh = m_modelObjects.back()->meshData;

    // Model-local material indices -> scene-wide; a mesh without materials uses material 0.
    const i32 base = static_cast<i32>(m_materials.size());
    for (auto &id : mesh.identifiers) id.materialIndex = mesh.materials.empty() ? 0 : base + id.materialIndex;

    // Model-local texture numbers -> word offset of the texture header in the scene stream.
    std::vector<i32> headerAt(mesh.textures.size());
    i32 words = 0;
    for (size_t i = 0; i < headerAt.size(); ++i) {
        headerAt[i] = words;
        words += 5 + static_cast<i32>(mesh.textures[i].data.size());
    }
    for (auto &mat : mesh.materials) {
        i32 *refs[6] = {&mat.texture.normalTexture,   &mat.texture.baseColorTexture,    &mat.texture.metallicRoughnessTexture,
                        &mat.texture.emissiveTexture, &mat.texture.transmissionTexture, &mat.texture.occlusionTexture};
        for (i32 *r : refs)
            if (*r != -1) *r = headerAt[*r] + m_textureTotalSize;
    }
    m_materials.insert(m_materials.end(), mesh.materials.begin(), mesh.materials.end());
    m_textureTotalSize += words;
}

void RayScene::addObject(const TraceableObject &obj, const Material &material) {
    m_traceableObjects.push_back(obj.clone());
    m_traceableObjects.back()->m_materialIndex = static_cast<i32>(m_materials.size());
    m_materials.push_back(material);
}

void RayScene::addObject(const TraceableObject &obj, int materialIndex) {
    ASSERT(materialIndex >= 0 && materialIndex < static_cast<int>(m_materials.size()));
    m_traceableObjects.push_back(obj.clone());
    m_traceableObjects.back()->m_materialIndex = materialIndex;
}

std::string RayScene::genShaderCode() { return "hip:gfx950"; }

const std::vector<f32> &RayScene::streamWords(int slot) const {
    // Slots outside 1..5 read the texture stream.
    const int s = (slot >= RT_SLOT_OBJECTS && slot <= RT_SLOT_TEXTURES) ? slot : RT_SLOT_TEXTURES;
    return wordsOf(*m_streams[s].storage);
}
