// Scene object types and their wire layouts (reference src/TraceableObject.h:14-146).
#pragma once

#include "util.h"
#include "glm/glm.hpp"
#include "AABB.h"
#include "gpu/Serializable.h"

#include <memory>
#include <string>
#include <vector>

// The numeric value is the `type` word of the primitive stream (src/RayScene.cpp:56,861-881).
enum class TraceableType { Sphere, Quad, Triangle, Model, Count };

class RayScene;
class TraceableObject : public gpu::Serializable {
    friend class RayScene;

private:
    i32 m_materialIndex = -1;
    TraceableType m_type = TraceableType::Count;

protected:
    AABB boundingBox;

public:
    explicit TraceableObject(TraceableType type) : m_type(type) {}
    virtual ~TraceableObject() = default;
    virtual std::unique_ptr<TraceableObject> clone() const = 0;
    virtual bool inAABB(const AABB &box) const = 0;

    TraceableType getType() const { return m_type; }
    i32 getMaterialIndex() const { return m_materialIndex; }
    virtual AABB getAABB() const { return boundingBox; }
};

// Wire: center.xyz, radius.
struct Sphere : public TraceableObject {
    Sphere(glm::vec3 center, f32 radius);
    bool inAABB(const AABB &box) const override;
    void serialize(gpu::Buffer &out) const override;
    std::unique_ptr<TraceableObject> clone() const override { return std::make_unique<Sphere>(*this); }

    glm::vec3 center = {0, 0, 0};
    f32 radius = 1.0f;
};

// Parallelogram q + a*u + b*v, a,b in [0,1].  Wire: q, u, v, cullFace(1.0/0.0).
struct Quad : public TraceableObject {
    Quad(glm::vec3 q, glm::vec3 u, glm::vec3 v, bool cullFace = false);
    bool inAABB(const AABB &box) const override;
    void serialize(gpu::Buffer &out) const override;
    std::unique_ptr<TraceableObject> clone() const override { return std::make_unique<Quad>(*this); }

    glm::vec3 q, u, v;
    bool cullFace;
};

// Wire: posA, posB, posC, normA, normB, normC.
struct Triangle : public TraceableObject {
    Triangle(glm::vec3 posA, glm::vec3 posB, glm::vec3 posC, glm::vec3 normA = {}, glm::vec3 normB = {},
             glm::vec3 normC = {});
    bool inAABB(const AABB &box) const override;
    void serialize(gpu::Buffer &out) const override;
    std::unique_ptr<TraceableObject> clone() const override { return std::make_unique<Triangle>(*this); }
    float area() const { return 0.5f * glm::length(glm::cross(posB - posA, posC - posA)); }

    glm::vec3 posA, posB, posC;
    glm::vec3 normA, normB, normC;
};

#include "BVHTree.h"
#include "MeshData.h"

// A triangle mesh with its BVH.  Wire blob (src/TraceableObject.cpp:202-228):
// [light identifiers L x 5][nodes N x 9][identifiers T x 5][vertices V x 8].
struct Model : public TraceableObject {
    Triangle getTriangleFromIdentifier(int idenIndex) const { return getTriangleFromIdentifier(meshData.identifiers[idenIndex]); }
    Triangle getTriangleFromIdentifier(const MeshData::Identifier &iden) const {
        const auto &a = meshData.vertices[iden.index.x], &b = meshData.vertices[iden.index.y], &c = meshData.vertices[iden.index.z];
        return Triangle{a.position, b.position, c.position, a.normal, b.normal, c.normal};
    }

    // Loads a .glb (see MeshData::LoadMeshData) and builds the BVH.
    explicit Model(std::string modelPath);
    // Builds the BVH over an in-memory mesh (no reference counterpart: the
    // reference can only construct a Model from a file path).
    explicit Model(MeshData mesh);
    Model(Model &&) = default;
    Model(const Model &other) : TraceableObject(TraceableType::Model), meshData(other.meshData), bvh(other.bvh), info(other.info) {
        boundingBox = other.boundingBox;
    }

    bool inAABB(const AABB &box) const override;
    void serialize(gpu::Buffer &out) const override;
    std::unique_ptr<TraceableObject> clone() const override { return std::make_unique<Model>(*this); }

    // Depth/leaf statistics of the built tree; `valid` is the reference's
    // leaf-containment check (src/TraceableObject.cpp:138-143).
    struct TreeStats {
        i32 minHeight = 0, maxHeight = 0, leafCount = 0, emptyLeaves = 0, minLeafTris = 0, maxLeafTris = 0;
        f32 avgHeight = 0;
        bool valid = true;
    };
    TreeStats treeStats() const;

    MeshData meshData;
    BVHTree bvh;

    // 3 words on the wire: identifiersCount, lightSourcesCount, nodesCount
    // (RayScene::submit appends the blob location as the 4th).
    struct Info : gpu::Serializable {
        u32 nodesCount = 0, identifiersCount = 0, lightSourcesCount = 0;
        void serialize(gpu::Buffer &out) const override;
    } info;

private:
    void finishConstruction();
};
