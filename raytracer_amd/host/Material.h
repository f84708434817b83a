// Surface description and its 23-word wire record (reference src/Material.h:10-63,
// decoded by loadMaterial in src/RayScene.cpp:383-415).
#pragma once

#include "glm/glm.hpp"
#include "util.h"
#include "gpu/Serializable.h"

using GVec3 = glm::vec<3, f32, glm::packed>;

// Texture references: -1 = none.  After RayScene::addModel they hold the float
// offset of the texture header inside the scene-wide texture stream.
struct MaterialTexture {
    i32 normalTexture = -1, baseColorTexture = -1, metallicRoughnessTexture = -1;
    i32 emissiveTexture = -1, transmissionTexture = -1, occlusionTexture = -1;
};

struct Material : public gpu::Serializable {
    // Emission: colour and a scale; a material emits when both are non-zero.
    GVec3 emissionColor = {0, 0, 0};
    f32 emissionStrength = 0;

    // Disney-style surface terms.
    GVec3 albedo = {1, 1, 1};
    f32 subsurface = 0, roughness = 1.0f, metallic = 0;
    f32 specular = 0.5f, specularTint = 0;

    // Refraction: share of transmitted light and index of refraction.
    f32 transmission = 0, ior = 1.5f;

    // Texture modifiers: alpha below alphaCut is a miss; normalScale < 0 means "no normal map".
    f32 alphaCut = 0;
    f32 normalScale = -1, occlusionStrength = 1.0f;

    MaterialTexture texture;

    static constexpr int kWords = 23;

    // Word order: 6 texture refs, emission rgb + strength, albedo rgb, subsurface,
    // roughness, metallic, specular, specularTint, transmission, ior, alphaCut,
    // normalScale, occlusionStrength.
    void serialize(gpu::Buffer &out) const override {
        const i32 refs[6] = {texture.normalTexture,   texture.baseColorTexture,    texture.metallicRoughnessTexture,
                             texture.emissiveTexture, texture.transmissionTexture, texture.occlusionTexture};
        for (i32 r : refs) out.push(r);
        out.push(emissionColor);
        out.push(emissionStrength);
        out.push(albedo);
        const f32 tail[10] = {subsurface, roughness, metallic,    specular,    specularTint,
                              transmission, ior,     alphaCut,    normalScale, occlusionStrength};
        for (f32 f : tail) out.push(f);
    }
};
