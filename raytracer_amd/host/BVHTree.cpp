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
#include <cstdlib>
#include <cstring>
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

// Binned SAH over the same triangle boxes.  Boxes are accumulated exactly like the median
// build (running padded union in range order), so they stay conservative in the same way.
struct SahBuilder {
    static constexpr int kBins = 16;
    static constexpr int kMaxDepth = 30;   // the traversal stack holds 32 entries
    std::vector<BVHNode> &nodes;
    std::vector<u32> &order;
    const std::vector<AABB> &triBox;
    const std::vector<glm::vec3> &centroid;

    static f32 halfArea(const glm::vec3 &lo, const glm::vec3 &hi) {
        const f32 dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
        return dx * dy + dy * dz + dz * dx;
    }

    void build(i32 start, i32 end, int depth) {
        AABB box = triBox[order[start]];
        for (i32 i = start + 1; i < end; ++i) box = AABB(box, triBox[order[i]]);
        const i32 self = static_cast<i32>(nodes.size());
        nodes.emplace_back();
        nodes[self].box = box;
        const i32 count = end - start;
        if (count <= 2) {
            nodes[self].leftIndex = start;
            nodes[self].rightIndex = end;
            nodes[self].isLeaf = true;
            return;
        }

        i32 mid = -1;
        glm::vec3 cLo = centroid[order[start]], cHi = cLo;
        for (i32 i = start + 1; i < end; ++i) {
            cLo = glm::min(cLo, centroid[order[i]]);
            cHi = glm::max(cHi, centroid[order[i]]);
        }
        // Subtrees that would no longer fit the depth budget with uneven splits fall back to the median.
        int levelsNeeded = 0;
        while ((1 << levelsNeeded) < count) ++levelsNeeded;
        const bool balancedOnly = depth + levelsNeeded >= kMaxDepth;
        int bestAxis = -1, bestBin = -1;
        if (!balancedOnly) {
            f32 bestCost = 1e30f;
            for (int axis = 0; axis < 3; ++axis) {
                const f32 extent = cHi[axis] - cLo[axis];
                if (!(extent > 0.0f)) continue;
                const f32 scale = static_cast<f32>(kBins) / extent;
                i32 binCount[kBins] = {};
                glm::vec3 binLo[kBins], binHi[kBins];
                for (int b = 0; b < kBins; ++b) { binLo[b] = glm::vec3(1e30f); binHi[b] = glm::vec3(-1e30f); }
                for (i32 i = start; i < end; ++i) {
                    const u32 t = order[i];
                    int b = static_cast<int>((centroid[t][axis] - cLo[axis]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    binCount[b]++;
                    binLo[b] = glm::min(binLo[b], glm::vec3(triBox[t].min));
                    binHi[b] = glm::max(binHi[b], glm::vec3(triBox[t].max));
                }
                f32 leftArea[kBins], rightArea[kBins];
                i32 leftCount[kBins], rightCount[kBins];
                glm::vec3 lo(1e30f), hi(-1e30f);
                i32 n = 0;
                for (int b = 0; b < kBins; ++b) {
                    if (binCount[b]) { lo = glm::min(lo, binLo[b]); hi = glm::max(hi, binHi[b]); }
                    n += binCount[b];
                    leftCount[b] = n;
                    leftArea[b] = n ? halfArea(lo, hi) : 0.0f;
                }
                lo = glm::vec3(1e30f); hi = glm::vec3(-1e30f); n = 0;
                for (int b = kBins - 1; b >= 0; --b) {
                    if (binCount[b]) { lo = glm::min(lo, binLo[b]); hi = glm::max(hi, binHi[b]); }
                    n += binCount[b];
                    rightCount[b] = n;
                    rightArea[b] = n ? halfArea(lo, hi) : 0.0f;
                }
                for (int b = 0; b + 1 < kBins; ++b) {   // split between bin b and b+1
                    if (leftCount[b] == 0 || rightCount[b + 1] == 0) continue;
                    const f32 cost = leftArea[b] * static_cast<f32>(leftCount[b]) + rightArea[b + 1] * static_cast<f32>(rightCount[b + 1]);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
                }
            }
        }
        if (bestAxis >= 0) {
            const f32 scale = static_cast<f32>(kBins) / (cHi[bestAxis] - cLo[bestAxis]);
            const f32 lo = cLo[bestAxis];
            const int axis = bestAxis, bin = bestBin;
            const std::vector<glm::vec3> &c = centroid;
            auto it = std::stable_partition(order.begin() + start, order.begin() + end, [&](u32 t) {
                int b = static_cast<int>((c[t][axis] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= bin;
            });
            mid = static_cast<i32>(it - order.begin());
        }
        if (mid <= start || mid >= end) {   // no usable SAH split: median on the widest centroid axis
            int axis = 0;
            if (cHi[1] - cLo[1] > cHi[axis] - cLo[axis]) axis = 1;
            if (cHi[2] - cLo[2] > cHi[axis] - cLo[axis]) axis = 2;
            mid = start + count / 2;
            const std::vector<glm::vec3> &c = centroid;
            std::nth_element(order.begin() + start, order.begin() + mid, order.begin() + end,
                             [&c, axis](u32 a, u32 b) { return c[a][axis] < c[b][axis] || (c[a][axis] == c[b][axis] && a < b); });
        }
        nodes[self].leftIndex = self + 1;
        build(start, mid, depth + 1);
        nodes[self].rightIndex = static_cast<i32>(nodes.size());
        build(mid, end, depth + 1);
    }
};

BVHTree::Mode g_defaultMode = [] {
    const char *env = std::getenv("RT_BVH_MODE");
    return (env != nullptr && std::strcmp(env, "sah") == 0) ? BVHTree::Mode::SAH : BVHTree::Mode::Reference;
}();

}  // namespace

BVHTree::Mode BVHTree::defaultMode() { return g_defaultMode; }
void BVHTree::setDefaultMode(Mode mode) { g_defaultMode = mode; }

BVHTree::BVHTree(Model &model) : BVHTree(model, g_defaultMode) {}

BVHTree::BVHTree(Model &model, Mode mode) {
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

    if (mode == Mode::SAH) {
        SahBuilder b{m_nodes, order, triBox, centroid};
        b.build(0, static_cast<i32>(n), 1);
    } else {
        Builder b{m_nodes, order, triBox, centroid};
        b.build(axis, 0, static_cast<i32>(n));
    }

    std::vector<MeshData::Identifier> leafOrder(n);
    for (size_t i = 0; i < n; ++i) leafOrder[i] = mesh.identifiers[order[i]];
    mesh.identifiers = std::move(leafOrder);
}
