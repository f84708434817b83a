#include "MeshData.h"
#include <iostream>

void MeshData::adoptLights(const MeshData &lights) {
    if (lights.identifiers.empty()) return;
    lightIdentifiers = lights.identifiers;
    lightSourcesCount = static_cast<i32>(lights.identifiers.size());

    const int shift = static_cast<int>(lights.vertices.size());
    for (auto &id : identifiers) id.index = id.index + glm::ivec3(shift);

    vertices.insert(vertices.begin(), lights.vertices.begin(), lights.vertices.end());
    identifiers.insert(identifiers.begin(), lights.identifiers.begin(), lights.identifiers.end());
}

void MeshData::deriveTextureFlags() {
    for (auto &id : identifiers) {
        if (materials.empty()) { id.hasTextures = false; continue; }
        const MaterialTexture &t = materials[id.materialIndex].texture;
        // transmissionTexture is deliberately not part of the test (src/MeshData.cpp:463-467).
        id.hasTextures = t.baseColorTexture != -1 || t.metallicRoughnessTexture != -1 || t.normalTexture != -1 ||
                         t.emissiveTexture != -1 || t.occlusionTexture != -1;
    }
}

void MeshData::Identifier::serialize(gpu::Buffer &out) const {
    out.push(index);
    out.push(materialIndex);
    out.push(hasTextures);
}

void MeshData::Texture::serialize(gpu::Buffer &out) const {
    for (i32 h : {width, height, channels, wrapS, wrapT}) out.push(h);
    out.push(reinterpret_cast<const char *>(data.data()), data.size() * sizeof(f32));
}
