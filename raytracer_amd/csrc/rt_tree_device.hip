// rt_tree_device.hip — the models' trees built on the device (gfx950).
//
// Steps, all on one stream:
//   1. k_triangle_boxes  bounds of every triangle (the exact minima / maxima of its vertices, as the host build
//                        takes them) and the bounds of the box centres (one atomic per wave and bound)
//   2. k_morton_keys     63-bit Morton key of every centre inside those bounds (21 bits per axis)
//   3. hipcub radix sort of (key, triangle)
//   4. rounds of k_nearest, k_mark_pairs, a prefix sum and k_merge: neighbouring clusters of the sorted order that
//      choose each other are merged until one is left (see below); the host reads two counts per round
//   5a. buildWideTrees: the four-wide nodes and the triangle pairs of their leaves are made on the device too, level
//       by level from the roots (k_wide_plan, a prefix sum, k_wide_write): nothing but counts goes to the host
//   5b. buildBinaryTree: k_emit writes the binary tree as rtc::BinNode records, root first, and the triangles' leaf
//       order for the host's collapse (eight-wide trees, scenes with alpha-cut textures)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "rt_tree_device.h"

namespace {

using rtc::BinNode;
using rttree::ModelTriangles;

constexpr int kThreads = 256;

// Floats as unsigned integers of the same order, for atomicMin / atomicMax.
__device__ __forceinline__ uint32_t orderedBits(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fromOrdered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ float waveMin(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float waveMax(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// boxes[t] = (min.xyz, max.xyz) of triangle t; centreBounds[0..2] / [3..5] = ordered bits of the least / greatest
// box centre per axis; *bad is set for a coordinate that is not finite or beyond 1e15.
__global__ __launch_bounds__(kThreads) void k_triangle_boxes(ModelTriangles m, float *boxes, uint32_t *centreBounds, int *bad) {
    const int t = (int)(blockIdx.x * kThreads + threadIdx.x);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool fine = true;
    if (t < m.count) {
        const float *ident = m.models + (size_t)m.identBase + (size_t)t * rtd::kIdentWords;
        for (int c = 0; c < 3; ++c) {
            const float *v = m.models + (size_t)m.vertBase + (size_t)__float_as_int(ident[c]) * 8;
            for (int k = 0; k < 3; ++k) {
                const float x = v[k];
                fine = fine && (fabsf(x) <= 1e15f);
                lo[k] = fminf(lo[k], x);
                hi[k] = fmaxf(hi[k], x);
            }
        }
        for (int k = 0; k < 3; ++k) {
            boxes[(size_t)t * 6 + k] = lo[k];
            boxes[(size_t)t * 6 + 3 + k] = hi[k];
        }
    }
    if (!fine) *bad = 1;
    const bool lead = (threadIdx.x & 63u) == 0u;
    for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * lo[k] + 0.5f * hi[k];   // +-inf or nan for a lane without a triangle: ignored by fmin / fmax below
        const float least = waveMin(t < m.count ? c : INFINITY), most = waveMax(t < m.count ? c : -INFINITY);
        if (lead && least <= most) {
            atomicMin(&centreBounds[k], orderedBits(least));
            atomicMax(&centreBounds[3 + k], orderedBits(most));
        }
    }
}

// Spreads the low 21 bits of x so that two zero bits follow each.
__device__ __forceinline__ uint64_t spread3(uint32_t x) {
    uint64_t v = x & 0x1fffffu;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(kThreads) void k_morton_keys(int count, const float *boxes, const uint32_t *centreBounds, uint64_t *keys, uint32_t *values) {
    const int t = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (t >= count) return;
    uint64_t key = 0;
    for (int k = 0; k < 3; ++k) {
        const float least = fromOrdered(centreBounds[k]), most = fromOrdered(centreBounds[3 + k]);
        const float c = 0.5f * boxes[(size_t)t * 6 + k] + 0.5f * boxes[(size_t)t * 6 + 3 + k];
        const double extent = (double)most - (double)least;
        uint32_t cell = 0;
        if (extent > 0.0) {
            const double at = ((double)c - (double)least) / extent * 2097152.0;
            cell = at >= 2097151.0 ? 2097151u : at > 0.0 ? (uint32_t)at : 0u;
        }
        key |= spread3(cell) << (2 - k);
    }
    keys[t] = key;
    values[t] = (uint32_t)t;
}

// ---- Agglomerative clustering along the Morton order ------------------------------------------------------------
// (D. Meister, J. Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", TVCG 2018.)
// The clusters stand in Morton order; in every round each cluster looks kRadius places to either side for the
// neighbour whose union with it has the smallest surface, and two clusters that chose each other are merged into
// one node, which takes the place of the first.  Rounds repeat until one cluster is left.  Node numbers: the
// triangle at place p of the sorted order is node p; merged nodes are numbered from `count` upwards in the order of
// their places within a round (from a prefix sum, so the numbering does not depend on timing).  Two single
// triangles merge into a leaf of two; every other merge makes an inner node.
constexpr int kRadius = 16;

struct BuildNode {
    float box[6];
    int32_t left, right;   // node numbers (of two single triangles for a leaf of two)
    int32_t parent;        // -1 for the root
    int32_t triangles;     // triangles below
    // The cheapest way to stand for this subtree with 1, 2 or 3 slots of a four-wide parent, and how the slots are
    // shared out between the two children: the same bottom-up solution as rtc::buildWideNodes computes on the host
    // (see there), filled in when the node is made -- its children are complete by then.
    float cost[3];
    uint8_t split[8];
};

__device__ __forceinline__ void leafCut(BuildNode &n) {
    rtc::leafCutCost(n.box, false, n.cost);
    for (int k = 0; k < 8; ++k) n.split[k] = 0;
}
__device__ __forceinline__ void innerCut(BuildNode &n, const BuildNode &l, const BuildNode &r) {
    for (int k = 0; k < 8; ++k) n.split[k] = 0;
    rtc::innerCutCost(n.box, l.cost, r.cost, n.cost, n.split);
}

__device__ __forceinline__ float unionHalfArea(const float *a, const float *b) {
    const float ex = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), ey = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), ez = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return ex * ey + ey * ez + ez * ex;
}

// The single-triangle nodes: place p of the sorted order.
__global__ __launch_bounds__(kThreads) void k_first_clusters(int count, const float *boxes, const uint32_t *order, BuildNode *nodes, int32_t *cluster, float *clusterBox) {
    const int p = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (p >= count) return;
    BuildNode n;
    for (int a = 0; a < 6; ++a) n.box[a] = clusterBox[(size_t)p * 6 + a] = boxes[(size_t)order[p] * 6 + a];
    n.left = n.right = -1;
    n.parent = -1;
    n.triangles = 1;
    leafCut(n);
    nodes[p] = n;
    cluster[p] = p;
}

// nearest[i] = the place within `radius` of i whose union with cluster i has the least surface.  Equal surfaces are
// told apart by the pair of places (lo, hi), the same way from both sides, so the best pair of all always chooses
// each other and every round merges something; the order among equals -- pairs that start at an even place first,
// then the closer ones -- makes a run of identical boxes pair up (0,1), (2,3), ... at once instead of one pair per
// round.  With `forced` every cluster simply takes its neighbour of that pairing: the host asks for it after a round
// that merged next to nothing (surfaces that grow steadily along the order chain every choice to the previous place).
__global__ __launch_bounds__(kThreads) void k_nearest(int clusters, int radius, int forced, const float *clusterBox, int32_t *nearest) {
    const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (i >= clusters) return;
    if (forced) {
        const int j = i ^ 1;
        nearest[i] = j < clusters ? j : -1;
        return;
    }
    float mine[6];
    for (int a = 0; a < 6; ++a) mine[a] = clusterBox[(size_t)i * 6 + a];
    const int from = i - radius < 0 ? 0 : i - radius, to = i + radius >= clusters ? clusters - 1 : i + radius;
    float best = INFINITY;
    int bestOdd = 2, bestGap = 0x7fffffff, bestLo = 0x7fffffff, at = -1;
    for (int j = from; j <= to; ++j) {
        if (j == i) continue;
        const float area = unionHalfArea(mine, clusterBox + (size_t)j * 6);
        const int lo = i < j ? i : j, gap = i < j ? j - i : i - j, odd = lo & 1;
        const bool better = at < 0 || area < best || (area == best && (odd < bestOdd || (odd == bestOdd && (gap < bestGap || (gap == bestGap && lo < bestLo)))));
        if (better) {
            best = area;
            bestOdd = odd;
            bestGap = gap;
            bestLo = lo;
            at = j;
        }
    }
    nearest[i] = at;
}

// flags[i]: low word 1 when place i survives the round (it is not the second of a merged pair), high word 1 when
// it is the first of a merged pair.
__global__ __launch_bounds__(kThreads) void k_mark_pairs(int clusters, const int32_t *nearest, uint64_t *flags) {
    const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (i >= clusters) return;
    const int j = nearest[i];
    const bool paired = j >= 0 && nearest[j] == i;
    flags[i] = (uint64_t)((paired && j < i) ? 0u : 1u) | ((uint64_t)((paired && i < j) ? 1u : 0u) << 32);
}

// sums = exclusive prefix sums of flags.  Writes the next round's clusters and the merged nodes; totals[0] / [1] =
// clusters left / nodes made, written by the last place.
__global__ __launch_bounds__(kThreads) void k_merge(int clusters, int count, int32_t nextNode, const int32_t *nearest, const uint64_t *flags, const uint64_t *sums,
                                                    const int32_t *cluster, const float *clusterBox, int32_t *clusterOut, float *clusterBoxOut, BuildNode *nodes, uint32_t *totals) {
    const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (i >= clusters) return;
    const uint64_t flag = flags[i], sum = sums[i];
    if (i == clusters - 1) {
        totals[0] = (uint32_t)(sum & 0xffffffffu) + (uint32_t)(flag & 0xffffffffu);
        totals[1] = (uint32_t)(sum >> 32) + (uint32_t)(flag >> 32);
    }
    if ((flag & 1u) == 0u) return;
    const uint32_t place = (uint32_t)(sum & 0xffffffffu);
    if ((flag >> 32) == 0u) {
        clusterOut[place] = cluster[i];
        for (int a = 0; a < 6; ++a) clusterBoxOut[(size_t)place * 6 + a] = clusterBox[(size_t)i * 6 + a];
        return;
    }
    const int j = nearest[i];
    const int32_t id = nextNode + (int32_t)(sum >> 32), l = cluster[i], r = cluster[j];
    BuildNode n;
    for (int a = 0; a < 3; ++a) {
        n.box[a] = fminf(clusterBox[(size_t)i * 6 + a], clusterBox[(size_t)j * 6 + a]);
        n.box[3 + a] = fmaxf(clusterBox[(size_t)i * 6 + 3 + a], clusterBox[(size_t)j * 6 + 3 + a]);
    }
    n.left = l;
    n.right = r;
    n.parent = -1;
    const BuildNode ln = nodes[l], rn = nodes[r];
    n.triangles = ln.triangles + rn.triangles;
    if (l < count && r < count) leafCut(n);
    else innerCut(n, ln, rn);
    nodes[id] = n;
    nodes[l].parent = id;
    nodes[r].parent = id;
    clusterOut[place] = id;
    for (int a = 0; a < 6; ++a) clusterBoxOut[(size_t)place * 6 + a] = n.box[a];
}

// The tree as the host reads it: node `id` of `total` goes to place total - 1 - id (the root was made last), a leaf's
// triangles to consecutive places of the leaf order -- the number of triangles to the left of the leaf in the tree,
// found by climbing to the root.  Single triangles inside a leaf of two keep an entry nobody refers to.
__global__ __launch_bounds__(kThreads) void k_emit(int count, int total, int32_t triBase, const BuildNode *nodes, const uint32_t *order, BinNode *out, int32_t *leafOrder) {
    const int id = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (id >= total) return;
    const BuildNode n = nodes[id];
    BinNode b;
    for (int a = 0; a < 6; ++a) b.box[a] = n.box[a];
    const bool single = id < count, pairLeaf = !single && n.left < count && n.right < count;
    b.leaf = single || pairLeaf;
    b.left = b.right = -1;
    b.leafRef = 0;
    if (!b.leaf) {
        b.left = total - 1 - n.left;
        b.right = total - 1 - n.right;
    } else if (single && n.parent >= 0 && nodes[n.parent].left < count && nodes[n.parent].right < count) {
        b.leafRef = ~(int32_t)0;   // part of a leaf of two: an empty leaf nobody refers to
    } else {
        int32_t before = 0;
        for (int32_t c = id, p = n.parent; p >= 0; c = p, p = nodes[p].parent)
            if (nodes[p].right == c) before += nodes[nodes[p].left].triangles;
        if (single) {
            leafOrder[before] = (int32_t)order[id];
        } else {
            leafOrder[before] = (int32_t)order[n.left];
            leafOrder[before + 1] = (int32_t)order[n.right];
        }
        b.leafRef = ~(int32_t)((((uint32_t)(triBase + before)) << 2) | (uint32_t)(single ? 1 : 2));
    }
    out[total - 1 - id] = b;
}

// ---- Four-wide nodes, level by level ---------------------------------------------------------------------------------
// A frontier entry is a binary node that becomes a wide node: its number, the wide node's index and the stack
// entries its ancestors keep (as rtc::buildWideNodes counts them).
struct Frontier { int32_t *node, *wide, *held; };

__device__ __forceinline__ bool isLeaf(const BuildNode *nodes, int count, int32_t id) { return id < count || (nodes[id].left < count && nodes[id].right < count); }

// The binary nodes that become the children of the wide node standing for n (rtc::buildWideNodes, cutChildren).
__device__ __forceinline__ int cutChildren(const BuildNode *nodes, int count, int32_t n, int32_t *kids) {
    int nk = 0, sp = 0;
    int32_t stackNode[8];
    int stackSlots[8];
    const int k4 = nodes[n].split[4] & 0x7f;
    stackNode[sp] = nodes[n].right; stackSlots[sp++] = 4 - k4;
    stackNode[sp] = nodes[n].left; stackSlots[sp++] = k4;
    while (sp > 0) {
        --sp;
        const int32_t node = stackNode[sp];
        int slots = stackSlots[sp];
        const bool leaf = isLeaf(nodes, count, node);
        while (!leaf && slots > 1 && (nodes[node].split[slots] & 0x80)) --slots;
        if (leaf || slots == 1) { kids[nk++] = node; continue; }
        const int k = nodes[node].split[slots] & 0x7f;
        stackNode[sp] = nodes[node].right; stackSlots[sp++] = slots - k;
        stackNode[sp] = nodes[node].left; stackSlots[sp++] = k;
    }
    return nk;
}

// family[4 i .. 4 i + 3] = the children of frontier entry i (-1: none); flags[i] = inner children | leaf children << 32.
__global__ __launch_bounds__(kThreads) void k_wide_plan(int entries, int count, const BuildNode *nodes, const int32_t *frontierNode, int32_t *family, uint64_t *flags) {
    const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (i >= entries) return;
    int32_t kids[4] = {-1, -1, -1, -1};
    const int nk = cutChildren(nodes, count, frontierNode[i], kids);
    uint32_t inner = 0, leaves = 0;
    for (int k = 0; k < 4; ++k) {
        family[(size_t)i * 4 + k] = kids[k];
        if (k < nk) { if (isLeaf(nodes, count, kids[k])) ++leaves; else ++inner; }
    }
    flags[i] = (uint64_t)inner | ((uint64_t)leaves << 32);
}

// One axis of a wide node's child boxes as 8-bit plane numbers (rtc::buildWideNodes, quantise): origin below every
// box less the margin, the smallest power-of-two step with which every upper plane fits in 255, planes rounded
// outwards and checked against the exact bounds.  False for bounds that are no numbers or out of order.
__device__ bool quantiseAxis(int nk, const float *lo, const float *hi, double pad, float &origin, float &stepOut, uint32_t &wlo, uint32_t &whi) {
    double lower = 0.0, upper = 0.0;
    for (int i = 0; i < nk; ++i) {
        const double mn = lo[i], mx = hi[i];
        if (!(fabs(mn) <= 1e15) || !(fabs(mx) <= 1e15) || mn > mx) return false;
        if (i == 0 || mn < lower) lower = mn;
        if (i == 0 || mx > upper) upper = mx;
    }
    float o = (float)(lower - pad);
    if ((double)o > lower - pad) o = nextafterf(o, -INFINITY);
    const double extent = (upper + pad) - (double)o;
    int e = -60;
    if (extent > 0.0) { const int need = (int)ceil(log2(extent / 255.0)); e = need > -60 ? need : -60; }
    for (;; ++e) {
        const double step = ldexp(1.0, e);
        bool fits = true;
        wlo = whi = 0;
        for (int i = 0; i < 4 && fits; ++i) {
            uint32_t qlo = 255, qhi = 0;   // unused slot: inverted
            if (i < nk) {
                const double mn = (double)lo[i] - pad, mx = (double)hi[i] + pad;
                long long ql = (long long)floor((mn - (double)o) / step), qh = (long long)ceil((mx - (double)o) / step);
                ql = ql < 0 ? 0 : ql > 255 ? 255 : ql;
                while (ql > 0 && (double)o + (double)ql * step > mn) --ql;
                qh = qh < 0 ? 0 : qh;
                while (qh <= 255 && (double)o + (double)qh * step < mx) ++qh;
                if (qh > 255) { fits = false; break; }
                qlo = (uint32_t)ql;
                qhi = (uint32_t)qh;
            }
            wlo |= qlo << (8 * i);
            whi |= qhi << (8 * i);
        }
        if (fits) {
            origin = o;
            stepOut = (float)step;
            return true;
        }
    }
}

// The pair record of a leaf (one or two triangles); `sorted` maps a single-triangle node to the model's triangle.
__device__ __forceinline__ void writePair(rtd::TriPair *out, const BuildNode *nodes, int count, int32_t leaf, const uint32_t *sorted, const rtd::PackedTri *tris, int32_t triBase) {
    rtd::TriPair tp;
    for (int h = 0; h < 2; ++h) {
        tp.ax[h] = tp.ay[h] = tp.az[h] = tp.abx[h] = tp.aby[h] = tp.abz[h] = tp.acx[h] = tp.acy[h] = tp.acz[h] = 0.0f;
        tp.ident[h] = (int32_t)rtd::kMiss;
    }
    const int n = leaf < count ? 1 : 2;
    for (int h = 0; h < n; ++h) {
        const int32_t single = leaf < count ? leaf : (h == 0 ? nodes[leaf].left : nodes[leaf].right);
        const int32_t at = triBase + (int32_t)sorted[single];
        const rtd::PackedTri t = tris[at];
        tp.ax[h] = t.a[0]; tp.ay[h] = t.a[1]; tp.az[h] = t.a[2];
        tp.abx[h] = t.ab[0]; tp.aby[h] = t.ab[1]; tp.abz[h] = t.ab[2];
        tp.acx[h] = t.ac[0]; tp.acy[h] = t.ac[1]; tp.acz[h] = t.ac[2];
        tp.ident[h] = at;
    }
    *out = tp;
}

// The same leaf as two triangle records in leaf order (scenes with alpha-cut textures are walked triangle by
// triangle): record 2 * pair and, for a leaf of two, 2 * pair + 1; `ident` names the triangle in `tris`.
__device__ __forceinline__ void writeLeafTriangles(rtd::PackedTri *out, const BuildNode *nodes, int count, int32_t leaf, const uint32_t *sorted, const rtd::PackedTri *tris, int32_t triBase) {
    const int n = leaf < count ? 1 : 2;
    for (int h = 0; h < 2; ++h) {
        rtd::PackedTri t;
        if (h < n) {
            const int32_t single = leaf < count ? leaf : (h == 0 ? nodes[leaf].left : nodes[leaf].right);
            const int32_t at = triBase + (int32_t)sorted[single];
            t = tris[at];
            t.ident = at;
        } else {
            for (int k = 0; k < 3; ++k) t.a[k] = t.ab[k] = t.ac[k] = 0.0f;
            t.material = 0;
            t.model = 0;
            t.ident = (int32_t)rtd::kMiss;   // (no leaf reference reaches it)
        }
        out[h] = t;
    }
}

// Writes the wide node of every frontier entry: inner children get the next wide indices of this level (wideNext +
// the entry's prefix sum) and form the next frontier in the same order, leaf children the next pair records.
// results[0] / [1] = inner / leaf children of the level (from the last entry), [2] = deepest stack (max), [3] = bad boxes.
__global__ __launch_bounds__(kThreads) void k_wide_write(int entries, int count, const BuildNode *nodes, Frontier now, Frontier next, const int32_t *family, const uint64_t *flags,
                                                         const uint64_t *sums, uint32_t wideNext, uint32_t pairNext, double pad, const uint32_t *sorted,
                                                         const rtd::PackedTri *tris, int32_t triBase, rtd::WideNode *wide, rtd::TriPair *pairs, rtd::PackedTri *leafTris,
                                                         uint32_t *results) {
    const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (i >= entries) return;
    const uint64_t flag = flags[i], sum = sums[i];
    if (i == entries - 1) {
        results[0] = (uint32_t)(sum & 0xffffffffu) + (uint32_t)(flag & 0xffffffffu);
        results[1] = (uint32_t)(sum >> 32) + (uint32_t)(flag >> 32);
    }
    int32_t kids[4];
    int nk = 0;
    for (int k = 0; k < 4; ++k) {
        kids[k] = family[(size_t)i * 4 + k];
        if (kids[k] >= 0) nk = k + 1;
    }
    const int held = now.held[i] + (nk > 1 ? nk - 1 : 0);
    atomicMax(&results[2], (uint32_t)held);
    uint32_t innerAt = (uint32_t)(sum & 0xffffffffu), pairAt = pairNext + (uint32_t)(sum >> 32);
    int32_t refs[4] = {-1, -1, -1, -1};
    float lo[3][4], hi[3][4];
    for (int k = 0; k < nk; ++k) {
        for (int a = 0; a < 3; ++a) { lo[a][k] = nodes[kids[k]].box[a]; hi[a][k] = nodes[kids[k]].box[3 + a]; }
        if (isLeaf(nodes, count, kids[k])) {
            if (leafTris != nullptr) {
                writeLeafTriangles(&leafTris[2u * pairAt], nodes, count, kids[k], sorted, tris, triBase);
                refs[k] = ~(int32_t)(((2u * pairAt) << 2) | (kids[k] < count ? 1u : 2u));
            } else {
                writePair(&pairs[pairAt], nodes, count, kids[k], sorted, tris, triBase);
                refs[k] = ~(int32_t)((pairAt << 2) | 1u);
            }
            ++pairAt;
        } else {
            refs[k] = (int32_t)(wideNext + innerAt);
            next.node[innerAt] = kids[k];
            next.wide[innerAt] = refs[k];
            next.held[innerAt] = held;
            ++innerAt;
        }
    }
    rtd::WideNode wn;
    wn.c0 = refs[0]; wn.c1 = refs[1]; wn.c2 = refs[2]; wn.c3 = refs[3];
    for (int a = 0; a < 3; ++a)
        if (!quantiseAxis(nk, lo[a], hi[a], pad, wn.o[a], wn.s[a], wn.lo[a], wn.hi[a])) {
            results[3] = 1u;
            wn.o[a] = 0.0f; wn.s[a] = 1.0f; wn.lo[a] = 0xffffffffu; wn.hi[a] = 0u;
        }
    wide[now.wide[i]] = wn;
}

// The pair record (or the two triangle records) of a model that is one leaf.
__global__ void k_root_leaf(int count, const BuildNode *nodes, int32_t leaf, const uint32_t *sorted, const rtd::PackedTri *tris, int32_t triBase, rtd::TriPair *pair,
                            rtd::PackedTri *leafTris) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (leafTris != nullptr) writeLeafTriangles(leafTris, nodes, count, leaf, sorted, tris, triBase);
    else writePair(pair, nodes, count, leaf, sorted, tris, triBase);
}

struct DeviceArena {
    void *base = nullptr;
    size_t used = 0, size = 0;
    DeviceArena() = default;
    DeviceArena(const DeviceArena &) = delete;
    DeviceArena &operator=(const DeviceArena &) = delete;
    ~DeviceArena() { if (base) (void)hipFree(base); }
    template <typename T> T *take(size_t count) {
        used = (used + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(static_cast<char *>(base) + used);
        used += count * sizeof(T);
        return p;
    }
};

}  // namespace

namespace rttree {

#define RT_TREE_HIP(call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            error = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP;                       \
        }                                                                                       \
    } while (0)

namespace {

// What the clustering of one model leaves on the device.
struct ModelBuild {
    DeviceArena arena;
    int count = 0, total = 0;          // triangles, nodes (the root is node total - 1)
    BuildNode *build = nullptr;
    uint32_t *sorted = nullptr;        // triangle of the model at every place of the Morton order
    float *spare[2] = {nullptr, nullptr};   // 6 * count words each, free once the clustering is done
    int32_t *leafOrder = nullptr;
    BinNode *binNodes = nullptr;
    uint64_t *flags = nullptr, *sums = nullptr;
    uint32_t *header = nullptr;        // centre bounds (6), bad flag, -, counts of a round or level (4)
    void *scratch = nullptr;
    size_t scanBytes = 0;
};

struct Laps {   // RT_COMMIT_TIMING=1 prints where the build spends its time (each lap ends with a wait for the stream)
    hipStream_t stream;
    bool on;
    std::chrono::steady_clock::time_point mark = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "rt_commit_scene:     %-26s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - mark).count());
        mark = now;
    }
};

dim3 blocksFor(int items) { return dim3((unsigned)((items + kThreads - 1) / kThreads)); }

// Steps 1 to 4 for one model (count >= 1).
int clusterModel(hipStream_t stream, const ModelTriangles &model, ModelBuild &mb, std::string &error) {
    const int count = model.count;
    const size_t n = (size_t)count, mostNodes = 2 * n - 1;
    Laps lap{stream, std::getenv("RT_COMMIT_TIMING") != nullptr && count >= 65536};
    size_t sortBytes = 0;
    RT_TREE_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, count, 0, 63, stream));
    RT_TREE_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, mb.scanBytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, count, stream));
    const size_t scratchBytes = sortBytes > mb.scanBytes ? sortBytes : mb.scanBytes;
    DeviceArena &arena = mb.arena;
    arena.size = n * (24 + 8 + 8 + 4 + 4 + 2 * 4 + 2 * 24 + 4 + 8 + 8 + 4) + mostNodes * (sizeof(BuildNode) + sizeof(BinNode)) + scratchBytes + 32 * 256;
    RT_TREE_HIP(hipMalloc(&arena.base, arena.size));
    float *boxes = arena.take<float>(n * 6);
    uint64_t *keys = arena.take<uint64_t>(n), *sortedKeys = arena.take<uint64_t>(n);
    uint32_t *values = arena.take<uint32_t>(n);
    mb.sorted = arena.take<uint32_t>(n);
    int32_t *cluster[2] = {arena.take<int32_t>(n), arena.take<int32_t>(n)};
    float *clusterBox[2] = {arena.take<float>(n * 6), arena.take<float>(n * 6)};
    int32_t *nearest = arena.take<int32_t>(n);
    mb.flags = arena.take<uint64_t>(n);
    mb.sums = arena.take<uint64_t>(n);
    mb.leafOrder = arena.take<int32_t>(n);
    mb.build = arena.take<BuildNode>(mostNodes);
    mb.binNodes = arena.take<BinNode>(mostNodes);
    mb.header = arena.take<uint32_t>(16);
    mb.scratch = arena.take<char>(scratchBytes);
    mb.spare[0] = clusterBox[0];
    mb.spare[1] = clusterBox[1];
    mb.count = count;
    if (arena.used > arena.size) { error = "tree build: scratch layout"; return RT_ERR_HIP; }

    const uint32_t init[16] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    RT_TREE_HIP(hipMemcpyAsync(mb.header, init, sizeof init, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_triangle_boxes, blocksFor(count), dim3(kThreads), 0, stream, model, boxes, mb.header, reinterpret_cast<int *>(mb.header + 6));
    hipLaunchKernelGGL(k_morton_keys, blocksFor(count), dim3(kThreads), 0, stream, count, boxes, mb.header, keys, values);
    RT_TREE_HIP(hipcub::DeviceRadixSort::SortPairs(mb.scratch, sortBytes, keys, sortedKeys, values, mb.sorted, count, 0, 63, stream));
    uint32_t bad = 0;
    RT_TREE_HIP(hipMemcpyAsync(&bad, mb.header + 6, sizeof bad, hipMemcpyDeviceToHost, stream));
    RT_TREE_HIP(hipStreamSynchronize(stream));
    lap("device tree: boxes, sort");
    if (bad != 0u) return RT_ERR_SCENE;   // the surfaces compared below would not be numbers
    hipLaunchKernelGGL(k_first_clusters, blocksFor(count), dim3(kThreads), 0, stream, count, boxes, mb.sorted, mb.build, cluster[0], clusterBox[0]);
    int radius = kRadius;
    if (const char *env = std::getenv("RT_TREE_RADIUS")) radius = std::min(std::max(std::atoi(env), 1), 256);   // (for measurements)
    static const bool alwaysForced = std::getenv("RT_TREE_FORCE_PAIRS") != nullptr;   // (a testing aid: every round pairs neighbours)
    int clusters = count, made = 0, side = 0, rounds = 0, forced = alwaysForced ? 1 : 0;
    while (clusters > 1) {
        ++rounds;
        hipLaunchKernelGGL(k_nearest, blocksFor(clusters), dim3(kThreads), 0, stream, clusters, radius, forced, clusterBox[side], nearest);
        hipLaunchKernelGGL(k_mark_pairs, blocksFor(clusters), dim3(kThreads), 0, stream, clusters, nearest, mb.flags);
        RT_TREE_HIP(hipcub::DeviceScan::ExclusiveSum(mb.scratch, mb.scanBytes, mb.flags, mb.sums, clusters, stream));
        hipLaunchKernelGGL(k_merge, blocksFor(clusters), dim3(kThreads), 0, stream, clusters, count, count + made, nearest, mb.flags, mb.sums, cluster[side], clusterBox[side],
                           cluster[side ^ 1], clusterBox[side ^ 1], mb.build, mb.header + 8);
        uint32_t totals[2];
        RT_TREE_HIP(hipMemcpyAsync(totals, mb.header + 8, sizeof totals, hipMemcpyDeviceToHost, stream));
        RT_TREE_HIP(hipStreamSynchronize(stream));
        if (totals[1] == 0u || (int)totals[0] != clusters - (int)totals[1]) { error = "tree build: a clustering round merged nothing"; return RT_ERR_HIP; }
        // A round that merged less than a sixteenth of its clusters is followed by one that pairs neighbours whatever
        // their surfaces, so the number of rounds stays logarithmic on any input.
        forced = (alwaysForced || (int)totals[1] * 16 < clusters) ? 1 : 0;
        clusters = (int)totals[0];
        made += (int)totals[1];
        side ^= 1;
    }
    mb.total = count + made;
    RT_TREE_HIP(hipGetLastError());
    lap("device tree: clustering");
    if (lap.on) std::fprintf(stderr, "rt_commit_scene:     %d triangles, %d rounds, %d nodes\n", count, rounds, mb.total);
    return RT_OK;
}

}  // namespace

int buildBinaryTree(hipStream_t stream, const ModelTriangles &model, std::vector<rtc::BinNode> &nodes, std::vector<int32_t> &order, std::string &error) {
    nodes.clear();
    order.clear();
    if (model.count <= 0) return RT_OK;
    ModelBuild mb;
    const int rc = clusterModel(stream, model, mb, error);
    if (rc != RT_OK) return rc;
    Laps lap{stream, std::getenv("RT_COMMIT_TIMING") != nullptr && model.count >= 65536};
    hipLaunchKernelGGL(k_emit, blocksFor(mb.total), dim3(kThreads), 0, stream, mb.count, mb.total, model.triBase, mb.build, mb.sorted, mb.binNodes, mb.leafOrder);
    RT_TREE_HIP(hipGetLastError());
    nodes.resize((size_t)mb.total);
    order.resize((size_t)mb.count);
    RT_TREE_HIP(hipMemcpyAsync(nodes.data(), mb.binNodes, (size_t)mb.total * sizeof(BinNode), hipMemcpyDeviceToHost, stream));
    RT_TREE_HIP(hipMemcpyAsync(order.data(), mb.leafOrder, (size_t)mb.count * 4, hipMemcpyDeviceToHost, stream));
    RT_TREE_HIP(hipStreamSynchronize(stream));
    lap("device tree: emit, download");
    return RT_OK;
}

void WideTrees::release() {
    if (nodes) (void)hipFree(nodes);
    if (pairs) (void)hipFree(pairs);
    if (leafTris) (void)hipFree(leafTris);
    nodes = nullptr;
    pairs = nullptr;
    leafTris = nullptr;
    nodeCount = pairCount = leafTriCount = 0;
}

int buildWideTrees(hipStream_t stream, const std::vector<ModelTriangles> &models, const rtd::PackedTri *tris, bool leavesAsTriangles, WideTrees &out, std::string &error) {
    out.release();
    out.rootRefs.assign(models.size(), rtd::kNoRoot);
    out.stackNeed = 0;
    // Binary trees of all models first: the margin of the quantised boxes depends on the largest coordinate of any.
    std::vector<ModelBuild> builds(models.size());
    size_t triangles = 0;
    double reach = 0.0;
    for (size_t m = 0; m < models.size(); ++m) {
        if (models[m].count <= 0) continue;
        const int rc = clusterModel(stream, models[m], builds[m], error);
        if (rc != RT_OK) return rc;
        triangles += (size_t)models[m].count;
        BuildNode root;
        RT_TREE_HIP(hipMemcpy(&root, builds[m].build + (builds[m].total - 1), sizeof root, hipMemcpyDeviceToHost));
        for (int k = 0; k < 6; ++k) reach = std::max(reach, (double)std::fabs(root.box[k]));
    }
    if (triangles == 0) return RT_OK;
    if (!(reach <= 1e15)) return RT_ERR_SCENE;
    const double pad = std::ldexp(reach, -21);   // as rtc::buildWideNodes

    // Room for the most there can be (a wide node per inner binary node, a pair per triangle); copied to size below.
    rtd::WideNode *wide = nullptr;
    rtd::TriPair *pairs = nullptr;
    RT_TREE_HIP(hipMalloc(reinterpret_cast<void **>(&wide), triangles * sizeof(rtd::WideNode)));
    struct Guard { void *p; ~Guard() { if (p) (void)hipFree(p); } } wideGuard{wide};
    rtd::PackedTri *leafTris = nullptr;
    if (leavesAsTriangles) RT_TREE_HIP(hipMalloc(reinterpret_cast<void **>(&leafTris), 2 * triangles * sizeof(rtd::PackedTri)));
    else RT_TREE_HIP(hipMalloc(reinterpret_cast<void **>(&pairs), triangles * sizeof(rtd::TriPair)));
    Guard pairGuard{pairs}, leafGuard{leafTris};
    uint32_t wideNext = 0, pairNext = 0, stackNeed = 0;
    for (size_t m = 0; m < models.size(); ++m) {
        ModelBuild &mb = builds[m];
        if (mb.count <= 0) continue;
        Laps lap{stream, std::getenv("RT_COMMIT_TIMING") != nullptr && mb.count >= 65536};
        const int count = mb.count;
        const size_t n = (size_t)count;
        int32_t *ints = reinterpret_cast<int32_t *>(mb.spare[0]);
        Frontier frontier[2] = {{ints, ints + n, ints + 2 * n}, {ints + 3 * n, ints + 4 * n, ints + 5 * n}};
        int32_t *family = reinterpret_cast<int32_t *>(mb.spare[1]);
        uint32_t *results = mb.header + 8;
        const int32_t rootNode = mb.total - 1;
        if (mb.total == 1 || mb.total == 3) {
            // One or two triangles: the root is a leaf, the model's root reference names its pair record.
            // (k_wide_write would need a parent; the pair is written by a one-entry launch of its own.)
            RT_TREE_HIP(hipMemsetAsync(results, 0, 16, stream));
            hipLaunchKernelGGL(k_root_leaf, dim3(1), dim3(64), 0, stream, count, mb.build, rootNode, mb.sorted, tris, models[m].triBase,
                               leavesAsTriangles ? nullptr : pairs + pairNext, leavesAsTriangles ? leafTris + 2 * (size_t)pairNext : nullptr);
            out.rootRefs[m] = leavesAsTriangles ? ~(int32_t)(((2u * pairNext) << 2) | (uint32_t)count) : ~(int32_t)((pairNext << 2) | 1u);
            ++pairNext;
            continue;
        }
        const int32_t first[3] = {rootNode, (int32_t)wideNext, 0};
        RT_TREE_HIP(hipMemcpyAsync(frontier[0].node, &first[0], 4, hipMemcpyHostToDevice, stream));
        RT_TREE_HIP(hipMemcpyAsync(frontier[0].wide, &first[1], 4, hipMemcpyHostToDevice, stream));
        RT_TREE_HIP(hipMemcpyAsync(frontier[0].held, &first[2], 4, hipMemcpyHostToDevice, stream));
        RT_TREE_HIP(hipMemsetAsync(results, 0, 16, stream));
        out.rootRefs[m] = (int32_t)wideNext;
        ++wideNext;
        int entries = 1, side = 0, levels = 0;
        while (entries > 0) {
            ++levels;
            hipLaunchKernelGGL(k_wide_plan, blocksFor(entries), dim3(kThreads), 0, stream, entries, count, mb.build, frontier[side].node, family, mb.flags);
            RT_TREE_HIP(hipcub::DeviceScan::ExclusiveSum(mb.scratch, mb.scanBytes, mb.flags, mb.sums, entries, stream));
            hipLaunchKernelGGL(k_wide_write, blocksFor(entries), dim3(kThreads), 0, stream, entries, count, mb.build, frontier[side], frontier[side ^ 1], family, mb.flags, mb.sums,
                               wideNext, pairNext, pad, mb.sorted, tris, models[m].triBase, wide, pairs, leafTris, results);
            uint32_t got[4];
            RT_TREE_HIP(hipMemcpyAsync(got, results, sizeof got, hipMemcpyDeviceToHost, stream));
            RT_TREE_HIP(hipStreamSynchronize(stream));
            if (got[3] != 0u) return RT_ERR_SCENE;
            wideNext += got[0];
            pairNext += got[1];
            stackNeed = std::max(stackNeed, got[2]);
            entries = (int)got[0];
            side ^= 1;
        }
        RT_TREE_HIP(hipGetLastError());
        lap("device tree: wide nodes");
        if (lap.on) std::fprintf(stderr, "rt_commit_scene:     %d levels, %u wide nodes, %u pairs so far\n", levels, wideNext, pairNext);
    }
    // (a wide node stands for an inner binary node, a pair for a leaf: neither can pass the number of triangles)
    RT_TREE_HIP(hipMalloc(reinterpret_cast<void **>(&out.nodes), std::max<size_t>(wideNext, 1) * sizeof(rtd::WideNode)));
    RT_TREE_HIP(hipMemcpyAsync(out.nodes, wide, (size_t)wideNext * sizeof(rtd::WideNode), hipMemcpyDeviceToDevice, stream));
    if (leavesAsTriangles) {
        RT_TREE_HIP(hipMalloc(reinterpret_cast<void **>(&out.leafTris), std::max<size_t>(2 * (size_t)pairNext, 1) * sizeof(rtd::PackedTri)));
        RT_TREE_HIP(hipMemcpyAsync(out.leafTris, leafTris, 2 * (size_t)pairNext * sizeof(rtd::PackedTri), hipMemcpyDeviceToDevice, stream));
        out.leafTriCount = 2 * (size_t)pairNext;
    } else {
        RT_TREE_HIP(hipMalloc(reinterpret_cast<void **>(&out.pairs), std::max<size_t>(pairNext, 1) * sizeof(rtd::TriPair)));
        RT_TREE_HIP(hipMemcpyAsync(out.pairs, pairs, (size_t)pairNext * sizeof(rtd::TriPair), hipMemcpyDeviceToDevice, stream));
        out.pairCount = pairNext;
    }
    RT_TREE_HIP(hipStreamSynchronize(stream));
    out.nodeCount = wideNext;
    out.stackNeed = (int)stackNeed;
    return RT_OK;
}

}  // namespace rttree
