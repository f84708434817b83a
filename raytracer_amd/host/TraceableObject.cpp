#include "TraceableObject.h"

#include <algorithm>
#include <climits>
#include <iostream>
#include <utility>

// ---- Sphere -------------------------------------------------------------------
Sphere::Sphere(glm::vec3 c, f32 r) : TraceableObject(TraceableType::Sphere), center(c), radius(r) {
    boundingBox = AABB(center - glm::vec3(radius), center + glm::vec3(radius));
}

bool Sphere::inAABB(const AABB &box) const {
    for (int k = 0; k < 3; ++k)
        if (center[k] - radius < box.min[k] || center[k] + radius > box.max[k]) return false;
    return true;
}

void Sphere::serialize(gpu::Buffer &out) const {
    out.push(center);
    out.push(radius);
}

// ---- Quad ---------------------------------------------------------------------
namespace {
void quadExtent(const Quad &quad, glm::vec3 &lo, glm::vec3 &hi) {
    const glm::vec3 p0 = quad.q, p1 = quad.q + quad.u, p2 = quad.q + quad.v, p3 = quad.q + quad.u + quad.v;
    lo = glm::min(glm::min(p0, p1), glm::min(p2, p3));
    hi = glm::max(glm::max(p0, p1), glm::max(p2, p3));
}
}  // namespace

Quad::Quad(glm::vec3 q_, glm::vec3 u_, glm::vec3 v_, bool cull) : TraceableObject(TraceableType::Quad), q(q_), u(u_), v(v_), cullFace(cull) {
    glm::vec3 lo, hi;
    quadExtent(*this, lo, hi);
    boundingBox = AABB(lo, hi);
}

// Overlap (not containment) test, as the reference (src/TraceableObject.cpp:42-54).
bool Quad::inAABB(const AABB &box) const {
    glm::vec3 lo, hi;
    quadExtent(*this, lo, hi);
    for (int k = 0; k < 3; ++k)
        if (hi[k] < box.min[k] || lo[k] > box.max[k]) return false;
    return true;
}

void Quad::serialize(gpu::Buffer &out) const {
    out.push(q);
    out.push(u);
    out.push(v);
    out.push(static_cast<f32>(cullFace));
}

// ---- Triangle -----------------------------------------------------------------
Triangle::Triangle(glm::vec3 a, glm::vec3 b, glm::vec3 c, glm::vec3 na, glm::vec3 nb, glm::vec3 nc)
    : TraceableObject(TraceableType::Triangle), posA(a), posB(b), posC(c), normA(na), normB(nb), normC(nc) {
    boundingBox = AABB(glm::min(posA, glm::min(posB, posC)), glm::max(posA, glm::max(posB, posC)));
}

bool Triangle::inAABB(const AABB &box) const {
    for (const glm::vec3 &p : {posA, posB, posC})
        for (int k = 0; k < 3; ++k)
            if (p[k] < box.min[k] || p[k] > box.max[k]) return false;
    return true;
}

void Triangle::serialize(gpu::Buffer &out) const {
    for (const glm::vec3 &p : {posA, posB, posC, normA, normB, normC}) out.push(p);
}

// ---- Model --------------------------------------------------------------------
Model::Model(std::string modelPath) : TraceableObject(TraceableType::Model), meshData(MeshData::LoadMeshData(std::move(modelPath))), bvh(*this) {
    finishConstruction();
}

Model::Model(MeshData mesh) : TraceableObject(TraceableType::Model), meshData(std::move(mesh)), bvh(*this) {
    finishConstruction();
}

void Model::finishConstruction() {
    const int n = static_cast<int>(meshData.identifiers.size());
    if (n > 0) {
        boundingBox = getTriangleFromIdentifier(0).getAABB();
        for (int i = 1; i < n; ++i) boundingBox = AABB(boundingBox, getTriangleFromIdentifier(i).getAABB());
    }
    if (!treeStats().valid) std::cout << "Invalid BVH\n";

    info.identifiersCount = static_cast<u32>(n);
    info.lightSourcesCount = static_cast<u32>(meshData.lightSourcesCount);
    info.nodesCount = static_cast<u32>(bvh.getNodes().size());
}

Model::TreeStats Model::treeStats() const {
    TreeStats s;
    const auto &nodes = bvh.getNodes();
    if (nodes.empty()) return s;
    s.minHeight = INT_MAX;
    s.minLeafTris = INT_MAX;
    s.maxLeafTris = INT_MIN;
    i64 heightSum = 0;

    std::vector<std::pair<i32, i32>> todo{{0, 1}};  // (node, depth)
    while (!todo.empty()) {
        auto [idx, depth] = todo.back();
        todo.pop_back();
        s.maxHeight = std::max(s.maxHeight, depth);
        s.minHeight = std::min(s.minHeight, depth);
        const BVHNode &node = nodes[idx];
        if (!node.isLeaf) {
            todo.push_back({node.rightIndex, depth + 1});
            todo.push_back({node.leftIndex, depth + 1});
            continue;
        }
        const i32 count = node.rightIndex - node.leftIndex;
        s.leafCount++;
        s.emptyLeaves += (count == 0);
        heightSum += depth;
        s.maxLeafTris = std::max(s.maxLeafTris, count);
        s.minLeafTris = std::min(s.minLeafTris, count);
        for (i32 i = node.leftIndex; i < node.rightIndex; ++i)
            if (!getTriangleFromIdentifier(i).inAABB(node.box)) s.valid = false;
    }
    s.avgHeight = static_cast<f32>(heightSum) / static_cast<f32>(s.leafCount);
    return s;
}

bool Model::inAABB(const AABB &box) const {
    return glm::max(boundingBox.min, box.min) == boundingBox.min && glm::min(boundingBox.max, box.max) == boundingBox.max;
}

void Model::Info::serialize(gpu::Buffer &out) const {
    out.push(static_cast<i32>(identifiersCount));
    out.push(static_cast<i32>(lightSourcesCount));
    out.push(static_cast<i32>(nodesCount));
}

void Model::serialize(gpu::Buffer &out) const {
    for (const auto &id : meshData.lightIdentifiers) id.serialize(out);

    for (const BVHNode &node : bvh.getNodes()) {
        out.push(node.box.min);
        out.push(node.box.max);
        out.push(node.leftIndex);
        out.push(node.rightIndex);
        out.push(node.isLeaf);
    }

    for (const auto &id : meshData.identifiers) id.serialize(out);

    for (const auto &v : meshData.vertices) {
        out.push(v.position);
        out.push(v.normal);
        out.push(v.uv);
    }
}
