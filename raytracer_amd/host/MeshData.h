// In-memory triangle soup of one model (reference src/MeshData.h:10-42).
#pragma once

#include "gpu/Serializable.h"
#include "Material.h"
#include "glm/glm.hpp"

#include <string>
#include <vector>

struct MeshData {
    // 8 words on the wire: position, normal, texture coordinate.
    struct Vertex {
        glm::vec3 position, normal;
        glm::vec2 uv;
    };
    std::vector<Vertex> vertices;

    // One triangle: three vertex indices + material; 5 words on the wire.
    struct Identifier : public gpu::Serializable {
        glm::ivec3 index;             // into `vertices`
        i32 materialIndex = 0;        // into `materials` until the scene rebases it
        bool hasTextures = false;     // the material has at least one texture
        void serialize(gpu::Buffer &out) const override;
    };

    // The model's triangles, the materials they name, and the textures those name.
    std::vector<Identifier> identifiers;
    std::vector<Material> materials;

    // Emissive triangles, also present at the FRONT of `identifiers`/`vertices`.
    std::vector<Identifier> lightIdentifiers;

    // Texel data already converted to f32; header is 5 words on the wire.
    struct Texture : public gpu::Serializable {
        i32 width = 0, height = 0, channels = 0;
        i32 channelSize = 1;                    // bytes per channel in the source image
        i32 wrapS = 10497, wrapT = 10497;       // glTF sampler codes; 10497 = repeat
        std::vector<f32> data;
        void serialize(gpu::Buffer &out) const override;
    };
    std::vector<Texture> textures;

    i32 lightSourcesCount = 0;

    // Moves `lights` to the front of this mesh the way the glTF loader does
    // (reference src/MeshData.cpp:444-454) and derives Identifier::hasTextures
    // from the materials (src/MeshData.cpp:461-473).
    void adoptLights(const MeshData &lights);
    void deriveTextureFlags();

    // Binary glTF (.glb) ingestion (reference src/MeshData.cpp:419-590), implemented in
    // GltfLoader.cpp without tinygltf/stb_image.  An unreadable path yields an empty mesh
    // exactly like a failed load in the reference (src/MeshData.cpp:429-433).
    static MeshData LoadMeshData(std::string modelPath);
};
