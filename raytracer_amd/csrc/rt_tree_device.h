// rt_tree_device.h — binary trees of triangle models built on the device (rt_tree_device.hip): Morton order of
// the triangle centres, then neighbouring clusters of that order merged bottom-up by least surface.  The commit-time
// code (rt_commit.cpp) collapses these trees into wide nodes exactly as it does its own.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "rt_commit.h"

namespace rttree {

// One model's triangles as the uploaded model stream holds them (`models` is the device copy of the stream):
// identifier t names its three vertices at models[identBase + t * kIdentWords + 0..2], vertex v sits at
// models[vertBase + v * 8 + 0..2].  The stream must have passed rtc::repackScene (every index in range).
struct ModelTriangles {
    const float *models;
    int32_t identBase, vertBase, count;
    int32_t triBase;   // index of the model's first triangle in the packed triangle array (goes into the leaf references)
};

// Builds the tree on `stream` and copies it to the host: `nodes` receives the nodes, root first, child numbers
// relative to the root, leaves of one or two triangles (entries no node refers to are empty leaves);
// `order[i]` = the model's triangle that stands at place i of the leaf order (a leaf's reference names places).
// Returns RT_OK, RT_ERR_SCENE for a coordinate that is not finite or beyond 1e15 (as the host build declines
// such a model), RT_ERR_NOMEM / RT_ERR_HIP otherwise (`error` then says which call).
int buildBinaryTree(hipStream_t stream, const ModelTriangles &model, std::vector<rtc::BinNode> &nodes, std::vector<int32_t> &order, std::string &error);

// The four-wide trees of all models and the triangle pairs of their leaves, made on the device from the same
// binary trees (the cut of at most four descendants per wide node, the 8-bit child boxes and the pair records
// are those rtc::buildWideNodes / rtc::pairLeaves make on the host).  `nodes` and `pairs` (or, with
// leavesAsTriangles, `leafTris`: what scenes with alpha-cut textures walk) are hipMalloc'ed, owned by the receiver; rootRefs[m] = child reference of model m's root (kNoRoot for a model without triangles).
struct WideTrees {
    rtd::WideNode *nodes = nullptr;
    rtd::TriPair *pairs = nullptr;        // leaves as pairs: a leaf reference names one pair
    rtd::PackedTri *leafTris = nullptr;   // leaves as triangles: a leaf reference names its first record and their number
    size_t nodeCount = 0, pairCount = 0, leafTriCount = 0;
    std::vector<int32_t> rootRefs;
    int stackNeed = 0;   // most stack entries a ray can hold
    void release();
    WideTrees() = default;
    WideTrees(const WideTrees &) = delete;
    WideTrees &operator=(const WideTrees &) = delete;
    ~WideTrees() { release(); }
};
// `tris`: the device copy of the packed triangles in upload order (ModelTriangles::triBase indexes it).
// RT_ERR_SCENE for coordinates the 8-bit boxes cannot hold (beyond 1e15, not finite).
int buildWideTrees(hipStream_t stream, const std::vector<ModelTriangles> &models, const rtd::PackedTri *tris, bool leavesAsTriangles, WideTrees &out,
                   std::string &error);

}  // namespace rttree
