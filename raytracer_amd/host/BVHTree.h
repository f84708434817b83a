// Host-side median-split BVH over a Model's triangles (reference src/BVHTree.h:6-24).
// The tree is kept exactly as the reference builds it because its shape and
// the order of triangles inside leaves decide which hit wins a distance tie.
#pragma once
#include "util.h"
#include "AABB.h"
#include <vector>

struct BVHNode {
    AABB box;
    // Inner node: child node indices.  Leaf: [leftIndex, rightIndex) into the
    // model's (re-ordered) identifier list.
    i32 leftIndex = -1, rightIndex = -1;
    bool isLeaf = false;

    BVHNode() = default;
};

struct Model;
class BVHTree {
    std::vector<BVHNode> m_nodes;

public:
    // Builds the tree and re-orders model.meshData.identifiers into leaf order.
    explicit BVHTree(Model &model);

    const std::vector<BVHNode> &getNodes() const { return m_nodes; }
};
