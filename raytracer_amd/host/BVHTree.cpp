// Median-split construction with the reference's exact decisions
// (reference src/BVHTree.cpp:6-85):
//   * node box = running union of the padded triangle boxes in range order
//     (each union pads again, src/AABB.cpp:8-16);
//   * leaf when the range holds <= 2 triangles;
//   * the split axis is advanced BEFORE it is used, starting from the longest
//     extent of the whole mesh; mid = start + (end-start)*0.5f;
//   * std::nth_element on the centroid coordinate; children are laid out in
//     pre-order (left = parent+1, right = first index after the left subtree).
// Unlike the reference, triangle boxes and centroids are computed once and the
// selection runs on an index permutation; nth_element only observes comparator
// results, so the resulting order is the same.
#include "BVHTree.h"
#include "TraceableObject.h"

#include <algorithm>
#include <numeric>

namespace {

struct Builder {
    std::vector<BVHNode> &nodes;
    std::vector<u32> &order;            // current permutation of original triangle ids
    const std::vector<AABB> &triBox;    // padded box per original triangle
    const std::vector<glm::vec3> &centroid;

    void build(i32 axis, i32 start, i32 end) {
        AABB box = triBox[order[start]];
        for (i32 i = start + 1; i < end; ++i) box = AABB(box, triBox[order[i]]);

        const i32 self = static_cast<i32>(nodes.size());
        nodes.emplace_back();
        nodes[self].box = box;

        if (end - start <= 2) {
            nodes[self].leftIndex = start;
            nodes[self].rightIndex = end;
            nodes[self].isLeaf = true;
            return;
        }

        axis = (axis + 1) % 3;
        const i32 mid = static_cast<i32>(start + (end - start) * 0.5f);
        const std::vector<glm::vec3> &c = centroid;
        std::nth_element(order.begin() + start, order.begin() + mid, order.begin() + end,
                         [&c, axis](u32 a, u32 b) { return c[a][axis] < c[b][axis]; });

        nodes[self].leftIndex = self + 1;
        build(axis, start, mid);
        nodes[self].rightIndex = static_cast<i32>(nodes.size());
        build(axis, mid, end);
    }
};

}  // namespace

BVHTree::BVHTree(Model &model) {
    MeshData &mesh = model.meshData;
    const size_t n = mesh.identifiers.size();
    if (n == 0) return;

    m_nodes.reserve(n * 2);

    std::vector<AABB> triBox(n);
    std::vector<glm::vec3> centroid(n);
    for (size_t i = 0; i < n; ++i) {
        Triangle t = model.getTriangleFromIdentifier(static_cast<int>(i));
        triBox[i] = t.getAABB();
        centroid[i] = (t.posA + t.posB + t.posC) / 3.0f;
    }

    AABB all = triBox[0];
    for (size_t i = 1; i < n; ++i) all = AABB(all, triBox[i]);

    const f32 xr = all.max.x - all.min.x, yr = all.max.y - all.min.y, zr = all.max.z - all.min.z;
    const f32 longest = std::max({xr, yr, zr});
    const i32 axis = (longest == xr) ? 0 : (longest == yr) ? 1 : 2;

    std::vector<u32> order(n);
    std::iota(order.begin(), order.end(), 0u);

    Builder b{m_nodes, order, triBox, centroid};
    b.build(axis, 0, static_cast<i32>(n));

    std::vector<MeshData::Identifier> leafOrder(n);
    for (size_t i = 0; i < n; ++i) leafOrder[i] = mesh.identifiers[order[i]];
    mesh.identifiers = std::move(leafOrder);
}
