// rt_commit.h — interface between the HIP translation unit (rt_hip.hip: context, uploads, launches)
// and the commit-time host work in rt_commit.cpp (plain C++: stream validation, re-pack, tree builds).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <memory>
#include <utility>
#include <vector>

#include "rt_hip.h"      // status codes, slot numbers
#include "rt_device.h"   // record layouts shared with the kernels

namespace rtc {
// Arrays of the re-packed records: their elements are written by the threads that fill them, so growing one does
// not clear the new elements first (a cleared half-gigabyte is a pass over memory by one thread).
// Large arrays start on a 2 MiB boundary and ask for huge pages: first touch of fresh memory by 4 KiB pages is
// page-fault bound, whatever the number of threads.
void *allocateRecords(size_t bytes);
template <typename T> struct Uncleared : std::allocator<T> {
    template <typename U> struct rebind { using other = Uncleared<U>; };
    T *allocate(size_t n) { return static_cast<T *>(allocateRecords(n * sizeof(T))); }
    void deallocate(T *p, size_t) noexcept { std::free(p); }
    template <typename U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename A, typename... More> void construct(U *p, A &&a, More &&...more) { ::new (static_cast<void *>(p)) U(std::forward<A>(a), std::forward<More>(more)...); }
};
using PackedTris = std::vector<rtd::PackedTri, Uncleared<rtd::PackedTri>>;
using PackedShades = std::vector<rtd::PackedShade, Uncleared<rtd::PackedShade>>;
using PackedUVs = std::vector<rtd::PackedUV, Uncleared<rtd::PackedUV>>;
using PackedNodes = std::vector<rtd::PackedNode, Uncleared<rtd::PackedNode>>;
// A node of a model's binary tree as the collapse into wide nodes reads it: box, children (node numbers), or the
// leaf reference ~((first triangle << 2) | triangle count).
struct BinNode { float box[6]; int32_t left, right; int32_t leafRef; bool leaf; };
// What a visit of a wide node / of a leaf costs, per unit of its surface, when the collapse chooses which binary
// nodes become the children of a wide node (buildWideNodes; the device build uses the same figures).
constexpr double kWideNodeCost = 1.0, kWideLeafCost = 0.9;

// The cheapest way to stand for a binary subtree with 1, 2 or 3 slots of a four-wide parent (cost[0..2]) and how
// the slots are shared out between its two children (split[j], j = 2..4: slots given to the left child when j
// are shared out; bit 7 marks "one slot fewer is as cheap").  Solved bottom-up: a leaf costs its surface x
// kWideLeafCost whatever the slots; an inner node with one slot becomes a wide node of its own (its surface x
// kWideNodeCost plus the cheapest way to stand for both children with four slots), with more slots it may hand
// them to its children instead.  Shared by the host collapse and the device build.
RT_FN double boxHalfArea(const float *b) {
    const double ex = (double)b[3] - b[0], ey = (double)b[4] - b[1], ez = (double)b[5] - b[2];
    return ex * ey + ey * ez + ez * ex;
}
RT_FN void leafCutCost(const float *box, bool empty, float cost[3]) {
    cost[0] = cost[1] = cost[2] = empty ? 0.0f : (float)(boxHalfArea(box) * kWideLeafCost);
}
RT_FN void innerCutCost(const float *box, const float lcost[3], const float rcost[3], float cost[3], uint8_t split[5]) {
    double share[5] = {0, 0, 0, 0, 0};
    split[0] = split[1] = 0;
    for (int j = 2; j <= 4; ++j) {
        double best = 0.0;
        int bestK = 0;
        for (int k = 1; k < j; ++k) {
            if (k > 3 || j - k > 3) continue;
            const double v = (double)lcost[k - 1] + (double)rcost[j - k - 1];
            if (bestK == 0 || v < best) { best = v; bestK = k; }
        }
        share[j] = best;
        split[j] = (uint8_t)bestK;
    }
    cost[0] = (float)(boxHalfArea(box) * kWideNodeCost + share[4]);
    for (int i = 1; i < 3; ++i) {
        if ((float)share[i + 1] < cost[i - 1]) cost[i] = (float)share[i + 1];
        else { cost[i] = cost[i - 1]; split[i + 1] = (uint8_t)(split[i + 1] | 0x80); }
    }
}
}  // namespace rtc

// The part of a context the commit-time code reads and writes; rt_ctx derives from it.
struct rt_scene_host {
    std::vector<float> host[6];     // uploaded words per slot (1..5)
    int objectCount = 0, modelsCount = 0, lightSourcesCount = 0;
    int wideBvh = 3;                // asked for: 0 off, 1 derived from the uploaded trees, 2 the library's own trees, 3 = 2 + ties re-traced in reference order, 4 = 3 on eight-wide trees
    // results of repackScene
    std::vector<int32_t> hPrimAt;   // word offset of every analytic primitive's record in the objects stream
    int maxTreeDepth = 0;           // deepest uploaded tree
    bool hasAlphaTest = false;      // some triangle needs the alpha-cut texture test
    // Set by the owner of a device: builds one model's binary tree there instead of on the host threads (child
    // numbers relative to the root, leaves naming places of `order`; see rt_tree_device.h).  Null: host build.
    int (*deviceTree)(rt_scene_host *ctx, const rtd::ModelDesc &model, std::vector<rtc::BinNode> &nodes, std::vector<int32_t> &order) = nullptr;
    // result of buildWideNodes / buildWide8
    int wideStackNeed = 0;          // most stack entries a ray can hold in the wide tree
};

namespace rtc {

// What packTextures did, for rt_get_info.
struct TexturePack {
    std::vector<float> words;        // the texture buffer the kernels read (headers + texels, see packTextures)
    std::vector<float> materials;    // the materials stream with its texture references pointing into `words`
    uint64_t packed8 = 0, kept32 = 0;       // textures stored as RGBA8 words / as fp32 words
    uint64_t sourceBytes = 0;               // bytes of the uploaded texture stream
};

// Device copy of the textures (the reference's loader turns 8-bit images into one fp32 word per channel, four
// times their size, src/MeshData.cpp:556-606).  Every texture a material refers to is copied into `out.words`;
// a three- or four-channel texture whose texels all equal (float)k / 255.0f bit for bit, k = 0..255, is stored
// as one 32-bit word per texel holding the k of its channels (channel c in byte c) and marked kTexPacked8 in
// its header's channel word; the kernels turn k back into the same float through a 256-entry table of
// k / 255.0f.  The reference's alpha fetch reads a fourth word whatever the channel count
// (src/RayScene.cpp:571-572): a three-channel texel keeps the next texel's red in its fourth byte, and the word
// that followed the last texel in the upload is kept as it was behind the packed texels.  Other textures
// keep their fp32 words, followed by the three words that followed them in the upload.  `enable` false copies
// the uploaded stream and materials unchanged.
void packTextures(const rt_scene_host *ctx, bool enable, TexturePack &out);

// table[k] = (float)k / 255.0f and table[256 + k] = srgb_to_linear of it (src/RayScene.cpp:268-277), computed
// with the same pinned arithmetic the kernels would run per fetch.
void colourTables8(float table[512]);

// Builds the traversal re-pack from the uploaded model streams and validates every offset the
// kernels will follow.  RT_ERR_SCENE for a stream the kernels could not walk safely.
int repackScene(rt_scene_host *ctx, PackedNodes &nodes, PackedTris &tris, PackedShades &shade,
                PackedUVs &uvs, std::vector<rtd::LightTri> &lightTris, std::vector<rtd::ModelDesc> &descs);
// Four-wide trees of every model (ctx->wideBvh 1: collapsed from the uploaded trees; >= 2: from the library's own).
int buildWideNodes(rt_scene_host *ctx, std::vector<rtd::ModelDesc> &descs, std::vector<rtd::WideNode> &wnodes, const PackedTris &tris,
                   PackedTris &wtris);
// Re-packs the leaves of four-wide trees as triangle pairs; leaf references then count pairs.
void pairLeaves(std::vector<rtd::WideNode> &wnodes, std::vector<rtd::ModelDesc> &descs, const PackedTris &wtris, std::vector<rtd::TriPair> &wpairs);
// Eight-wide trees over the library's own binary trees.
int buildWide8(rt_scene_host *ctx, std::vector<rtd::ModelDesc> &descs, std::vector<rtd::Wide8Node> &w8, const PackedTris &tris,
               PackedTris &w8tris);

}  // namespace rtc
