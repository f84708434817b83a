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
    i32 leftIndex = -1;
    i32 rightIndex = -1;
    bool isLeaf = false;
};

struct Model;

class BVHTree {
public:
    // Reference: the reference's median split, the only mode whose images are the reference's.
    // SAH: binned surface-area-heuristic splits (no reference counterpart; opt-in).  Same node
    // format, same padded boxes, leaves of <= 2 triangles, so the traversal code is unchanged,
    // but the visiting order differs: hits at exactly equal distance may resolve to another
    // triangle, and rays open about a third fewer nodes.
    enum class Mode { Reference, SAH };
    // Mode used by constructors that do not name one; starts as Reference unless the
    // environment variable RT_BVH_MODE is "sah".
    static Mode defaultMode();
    static void setDefaultMode(Mode mode);

    // Builds the tree and re-orders model.meshData.identifiers into leaf order.
    explicit BVHTree(Model &model);
    BVHTree(Model &model, Mode mode);

    // Pre-order: node 0 is the root, children come after their parent.
    const std::vector<BVHNode> &getNodes() const { return m_nodes; }

private:
    std::vector<BVHNode> m_nodes;
};
