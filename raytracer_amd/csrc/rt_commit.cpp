// rt_commit.cpp — the commit-time host work of the HIP backend, as plain C++ (no device code, no HIP calls):
// validation of the five uploaded word streams and their re-pack into the traversal records
// (repackScene), the library's own binned-SAH binary trees (sahTree), their collapse into four-wide
// and eight-wide nodes (buildWideNodes, buildWide8) and the pairing of leaf triangles (pairLeaves).
// Kept out of the .hip translation unit so that g++ alone builds it: the sanitizer and fuzz targets of
// the Makefile (asan, tsan, tools/fuzz_host.cpp) link this file without a GPU toolchain.
//
// Build with -ffp-contract=off (see rt_math.h): the shading records computed here must equal what the
// kernels would compute on every hit.

#include "rt_commit.h"

#include <algorithm>
#include <array>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

using namespace rtd;

namespace rtc {

// RT_COMMIT_TIMING=1: prints the host time since `since` under the name `what`, and restarts the clock.
void commitLap(const char *what, std::chrono::steady_clock::time_point &since) {
    static const bool on = std::getenv("RT_COMMIT_TIMING") != nullptr;
    const auto now = std::chrono::steady_clock::now();
    if (on) std::fprintf(stderr, "rt_commit_scene:   %-26s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - since).count());
    since = now;
}

// Host threads for building trees at commit: RT_BUILD_THREADS, else the machine's, at most 16.
int buildThreads() {
    if (const char *env = std::getenv("RT_BUILD_THREADS")) return std::min(std::max(std::atoi(env), 1), 64);
    return (int)std::min(std::max(std::thread::hardware_concurrency(), 1u), 16u);
}

// Joins every thread it holds when it goes out of scope, so that an exception thrown while threads
// are running (an allocation failure in the spawning loop or in the caller's own share of the work)
// unwinds past joined threads instead of ending the process in ~thread.
struct ThreadGroup {
    std::vector<std::thread> threads;
    explicit ThreadGroup(size_t expected) { threads.reserve(expected); }
    ~ThreadGroup() { join(); }
    void join() {
        for (std::thread &t : threads)
            if (t.joinable()) t.join();
    }
    // False when the system has no thread to give: the caller does that piece itself.
    template <typename Fn, typename... Args> bool spawn(Fn &&fn, Args &&...args) {
        try {
            threads.emplace_back(std::forward<Fn>(fn), std::forward<Args>(args)...);
            return true;
        } catch (const std::system_error &) {
            return false;
        }
    }
};

void *allocateRecords(size_t bytes) {
    void *p = nullptr;
    if (bytes >= ((size_t)4 << 20)) {
        if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) throw std::bad_alloc();
        (void)madvise(p, bytes, MADV_HUGEPAGE);   // advice only: without huge pages the array is an ordinary one
    } else {
        p = std::malloc(bytes ? bytes : 1);
        if (p == nullptr) throw std::bad_alloc();
    }
    return p;
}

// Builds the traversal re-pack from the uploaded model streams and validates
// every offset the kernels will follow.
int repackScene(rt_scene_host *ctx, PackedNodes &nodes, PackedTris &tris, PackedShades &shade, PackedUVs &uvs, std::vector<LightTri> &lightTris, std::vector<ModelDesc> &descs) {
    const std::vector<float> &info = ctx->host[RT_SLOT_MODEL_INFO];
    const std::vector<float> &mdl = ctx->host[RT_SLOT_MODELS];
    const std::vector<float> &mats = ctx->host[RT_SLOT_MATERIALS];
    const std::vector<float> &tex = ctx->host[RT_SLOT_TEXTURES];
    const std::vector<float> &obj = ctx->host[RT_SLOT_OBJECTS];
    const int materialCount = (int)(mats.size() / kMaterialWords);
    if (materialCount < 1) return RT_ERR_SCENE;   // material 0 must exist

    // Texture references of every material must point at a complete texture.
    for (int m = 0; m < materialCount; ++m) {
        for (int k = 0; k < 6; ++k) {
            int ref = f2i(mats[(size_t)m * kMaterialWords + k]);
            if (ref == -1) continue;
            if (ref < 0 || (size_t)ref + 5 > tex.size()) return RT_ERR_SCENE;
            int w = f2i(tex[ref]), h = f2i(tex[ref + 1]), c = f2i(tex[ref + 2]);
            if (w <= 0 || h <= 0 || c <= 0 || (size_t)ref + 5 + (size_t)w * h * c > tex.size()) return RT_ERR_SCENE;
        }
    }

    // Primitive stream: objectCount well-formed records.  Where each starts is kept for the shading kernels, which
    // would otherwise walk the variable-length records up to the one that was hit.
    ctx->hPrimAt.clear();
    {
        size_t at = 0;
        for (int i = 0; i < ctx->objectCount; ++i) {
            if (at + 2 > obj.size()) return RT_ERR_SCENE;
            int type = f2i(obj[at]), material = f2i(obj[at + 1]);
            if (material < 0 || material >= materialCount) return RT_ERR_SCENE;
            ctx->hPrimAt.push_back((int32_t)at);
            at += 2;
            at += (type == 0) ? 4 : (type == 1) ? 10 : (type == 2) ? 18 : 0;
            if (type < 0 || type > 2 || at > obj.size()) return RT_ERR_SCENE;
        }
    }

    ctx->maxTreeDepth = 0;
    ctx->hasAlphaTest = false;
    if ((size_t)ctx->modelsCount * 4 > info.size()) return RT_ERR_SCENE;
    if (ctx->lightSourcesCount < 0 || ctx->lightSourcesCount > ctx->modelsCount) return RT_ERR_SCENE;
    descs.resize((size_t)ctx->modelsCount);
    {   // room for every model's records at once: growing model by model would copy the earlier ones again and again
        uint64_t triangles = 0, treeNodes = 0;
        for (int m = 0; m < ctx->modelsCount; ++m) {
            triangles += (uint64_t)std::max(f2i(info[(size_t)m * 4 + 0]), 0);
            treeNodes += (uint64_t)std::max(f2i(info[(size_t)m * 4 + 2]), 0);
        }
        // (every identifier and node the counts promise must be in the stream: the reservations below are sized by them)
        if (triangles >= (1u << 29) || triangles * kIdentWords > mdl.size() || treeNodes * kNodeWords > mdl.size()) return RT_ERR_SCENE;
        tris.reserve((size_t)triangles);
        shade.reserve((size_t)triangles);
        uvs.reserve((size_t)triangles);
        nodes.reserve((size_t)treeNodes / 2 + 1);
    }

    for (int m = 0; m < ctx->modelsCount; ++m) {
        ModelDesc d{};
        d.identifiersCount = f2i(info[(size_t)m * 4 + 0]);
        d.lightSourcesCount = f2i(info[(size_t)m * 4 + 1]);
        d.nodesCount = f2i(info[(size_t)m * 4 + 2]);
        d.lightBase = f2i(info[(size_t)m * 4 + 3]);
        if (d.identifiersCount < 0 || d.lightSourcesCount < 0 || d.nodesCount < 0 || d.lightBase < 0) return RT_ERR_SCENE;
        if (m < ctx->lightSourcesCount && d.lightSourcesCount <= 0) return RT_ERR_SCENE;  // light models come first
        // Section ends in 64 bits: counts are caller data, and a product such as identifiersCount * 5
        // must not wrap into a small offset that passes the range test.
        const int64_t nodeBase = (int64_t)d.lightBase + (int64_t)d.lightSourcesCount * kIdentWords;
        const int64_t identBase = nodeBase + (int64_t)d.nodesCount * kNodeWords;
        const int64_t vertBase = identBase + (int64_t)d.identifiersCount * kIdentWords;
        if (vertBase > (int64_t)mdl.size() || vertBase > 0x7fffffff) return RT_ERR_SCENE;   // every section lies inside the stream
        d.nodeBase = (int32_t)nodeBase;
        d.identBase = (int32_t)identBase;
        d.vertBase = (int32_t)vertBase;
        d.triBase = (int32_t)tris.size();
        // Leaf references keep the triangle index in 29 bits below the sign.
        if (tris.size() + (size_t)d.identifiersCount >= (1u << 29)) return RT_ERR_SCENE;
        const int vertexCapacity = (int)((mdl.size() - (size_t)d.vertBase) / 8);

        auto checkIdent = [&](int wordAt) -> bool {
            for (int k = 0; k < 3; ++k) {
                int vi = f2i(mdl[(size_t)wordAt + k]);
                if (vi < 0 || vi >= vertexCapacity) return false;
            }
            int mat = f2i(mdl[(size_t)wordAt + 3]);
            return mat >= 0 && mat < materialCount;
        };
        d.lightTriBase = (int32_t)lightTris.size();
        for (int l = 0; l < d.lightSourcesCount; ++l) {
            LightTri lt{};
            float *corner[3] = {lt.p0, lt.p1, lt.p2};
            for (int k = 0; k < 3; ++k) {
                int vi = f2i(mdl[(size_t)d.lightBase + (size_t)l * kIdentWords + k]);
                if (vi < 0 || vi >= vertexCapacity) return RT_ERR_SCENE;
                for (int c = 0; c < 3; ++c) corner[k][c] = mdl[(size_t)d.vertBase + (size_t)vi * 8 + (size_t)c];
            }
            const V3 p0 = v3(lt.p0[0], lt.p0[1], lt.p0[2]), p1 = v3(lt.p1[0], lt.p1[1], lt.p1[2]), p2 = v3(lt.p2[0], lt.p2[1], lt.p2[2]);
            lt.area = 0.5f * length(cross(p1 - p0, p2 - p0));   // as sampleRandomPointFromLightSouces computes it, src/RayScene.cpp:900-935
            lightTris.push_back(lt);
        }

        // Triangles in identifier (leaf) order; a large model's are shared among host threads (every triangle's
        // records depend on its own identifier and vertices only).
        auto mark = std::chrono::steady_clock::now();
        {
            const size_t first = tris.size(), count = (size_t)d.identifiersCount;
            tris.resize(first + count);
            shade.resize(first + count);
            uvs.resize(first + count);
            std::atomic<bool> malformed{false}, alphaTest{false};
            auto pack = [&](size_t t0, size_t t1) {
                for (size_t t = t0; t < t1; ++t) {
                    const int at = d.identBase + (int)t * kIdentWords;
                    if (!checkIdent(at)) { malformed = true; return; }
                    const int idx[3] = {f2i(mdl[at]), f2i(mdl[at + 1]), f2i(mdl[at + 2])};
                    const int material = f2i(mdl[at + 3]);
                    const bool hasTextures = mdl[at + 4] != 0.0f;
                    const float *A = &mdl[(size_t)d.vertBase + (size_t)idx[0] * 8];
                    const float *B = &mdl[(size_t)d.vertBase + (size_t)idx[1] * 8];
                    const float *C = &mdl[(size_t)d.vertBase + (size_t)idx[2] * 8];
                    PackedTri pt;
                    for (int k = 0; k < 3; ++k) {
                        pt.a[k] = A[k];
                        pt.ab[k] = B[k] - A[k];
                        pt.ac[k] = C[k] - A[k];
                    }
                    pt.material = material;
                    pt.model = m;
                    pt.ident = (int32_t)t;
                    if (hasTextures) {
                        const int baseColor = f2i(mats[(size_t)material * kMaterialWords + 1]);
                        const float alphaCut = mats[(size_t)material * kMaterialWords + 20];
                        if (baseColor != -1 && alphaCut > 0.0f) {
                            pt.model |= kTriAlphaTest;
                            alphaTest = true;
                        }
                    }
                    tris[first + t] = pt;
                    PackedShade ps{};
                    for (int k = 0; k < 3; ++k) {
                        ps.n0[k] = A[3 + k];
                        ps.n1[k] = B[3 + k];
                        ps.n2[k] = C[3 + k];
                    }
                    // (the same single-precision operations, in the same order, as the kernels would run on every hit)
                    const V3 geometric = cross(v3(pt.ab[0], pt.ab[1], pt.ab[2]), v3(pt.ac[0], pt.ac[1], pt.ac[2]));
                    ps.geometric[0] = geometric.x; ps.geometric[1] = geometric.y; ps.geometric[2] = geometric.z;
                    ps.area = 0.5f * length(geometric);
                    ps.material = material;
                    ps.model = m;
                    ps.hasTextures = hasTextures ? 1 : 0;
                    shade[first + t] = ps;
                    PackedUV pu{};
                    for (int k = 0; k < 2; ++k) {
                        pu.uv0[k] = A[6 + k];
                        pu.uv1[k] = B[6 + k];
                        pu.uv2[k] = C[6 + k];
                    }
                    uvs[first + t] = pu;
                }
            };
            const int workers = count >= 65536 ? buildThreads() : 1;
            {
                ThreadGroup pool((size_t)workers);
                for (int w = 1; w < workers; ++w) {
                    const size_t t0 = count * (size_t)w / (size_t)workers, t1 = count * (size_t)(w + 1) / (size_t)workers;
                    if (!pool.spawn(pack, t0, t1)) pack(t0, t1);
                }
                pack(0, count / (size_t)workers);
                pool.join();
            }
            if (malformed) return RT_ERR_SCENE;
            if (alphaTest) ctx->hasAlphaTest = true;
        }
        if (d.identifiersCount >= 65536) commitLap("triangle records", mark);

        // Inner nodes get consecutive packed indices in original node order.  One pass in node order checks that
        // the stream holds a tree, not a graph -- children come after their parent (no cycles) and every node has at
        // most one parent (a node shared by two parents would make a 32-level graph 2^32 nodes large for anything that
        // walks it as a tree) -- and finds every node's depth (a child's from its parent's, which precedes it); the
        // traversal stack holds at most depth entries.
        std::vector<int32_t> packedIndex((size_t)d.nodesCount, -1);
        const int32_t innerBase = (int32_t)nodes.size();
        int32_t innerCount = 0;
        auto nodeWord = [&](int n, int w) { return mdl[(size_t)d.nodeBase + (size_t)n * kNodeWords + w]; };
        {
            std::vector<uint8_t> depth((size_t)d.nodesCount, 0);   // 0: not reachable from node 0; 255 stands for "too deep"
            std::vector<char> hasParent((size_t)d.nodesCount, 0);
            if (d.nodesCount > 0) depth[0] = 1;
            int maxDepth = d.nodesCount > 0 ? 1 : 0;
            for (int n = 0; n < d.nodesCount; ++n) {
                if (nodeWord(n, 8) != 0.0f) continue;
                packedIndex[(size_t)n] = innerBase + innerCount++;
                const int l = f2i(nodeWord(n, 6)), r = f2i(nodeWord(n, 7));
                if (l <= n || r <= n || l >= d.nodesCount || r >= d.nodesCount || l == r) return RT_ERR_SCENE;
                if (hasParent[(size_t)l] || hasParent[(size_t)r]) return RT_ERR_SCENE;
                hasParent[(size_t)l] = hasParent[(size_t)r] = 1;
                if (depth[(size_t)n] != 0) {
                    const int below = std::min((int)depth[(size_t)n] + 1, 255);
                    depth[(size_t)l] = depth[(size_t)r] = (uint8_t)below;
                    maxDepth = std::max(maxDepth, below);
                }
            }
            if (maxDepth > kStackDepth) return RT_ERR_SCENE;
            ctx->maxTreeDepth = std::max(ctx->maxTreeDepth, maxDepth);
        }
        auto childRef = [&](int n, int32_t &ref) -> bool {
            if (n < 0 || n >= d.nodesCount) return false;
            if (nodeWord(n, 8) != 0.0f) {
                int first = f2i(nodeWord(n, 6)), last = f2i(nodeWord(n, 7));
                if (first < 0 || last < first || last > d.identifiersCount) return false;   // (before the subtraction: both are caller data)
                int count = last - first;
                if (count > 3) return false;
                ref = ~(int32_t)((((uint32_t)(d.triBase + first)) << 2) | (uint32_t)count);
            } else {
                ref = packedIndex[(size_t)n];
            }
            return true;
        };
        nodes.resize((size_t)innerBase + (size_t)innerCount);
        // The packed nodes: every inner node's record depends on its own two children only.
        {
            std::atomic<bool> malformed{false};
            auto packNodes = [&](int n0, int n1) {
                for (int n = n0; n < n1; ++n) {
                    if (nodeWord(n, 8) != 0.0f) continue;
                    const int l = f2i(nodeWord(n, 6)), r = f2i(nodeWord(n, 7));
                    PackedNode pn{};
                    if (!childRef(l, pn.left) || !childRef(r, pn.right)) { malformed = true; return; }
                    for (int k = 0; k < 3; ++k) {
                        pn.lmin[k] = nodeWord(l, k);
                        pn.lmax[k] = nodeWord(l, 3 + k);
                        pn.rmin[k] = nodeWord(r, k);
                        pn.rmax[k] = nodeWord(r, 3 + k);
                    }
                    nodes[(size_t)packedIndex[(size_t)n]] = pn;
                }
            };
            const int workers = d.nodesCount >= 65536 ? buildThreads() : 1;
            ThreadGroup pool((size_t)workers);
            for (int w = 1; w < workers; ++w) {
                const int n0 = (int)((int64_t)d.nodesCount * w / workers), n1 = (int)((int64_t)d.nodesCount * (w + 1) / workers);
                if (!pool.spawn(packNodes, n0, n1)) packNodes(n0, n1);
            }
            packNodes(0, (int)((int64_t)d.nodesCount / workers));
            pool.join();
            if (malformed) return RT_ERR_SCENE;
        }
        if (d.nodesCount > 0 && !childRef(0, d.rootRef)) return RT_ERR_SCENE;
        if (d.identifiersCount >= 65536) commitLap("uploaded tree", mark);
        descs[(size_t)m] = d;
    }
    return RT_OK;
}

void colourTables8(float table[512]) {
    for (int k = 0; k < 256; ++k) {
        table[k] = (float)k / 255.0f;
        table[256 + k] = srgbToLinear(table[k]);
    }
}

void packTextures(const rt_scene_host *ctx, bool enable, TexturePack &out) {
    const std::vector<float> &mats = ctx->host[RT_SLOT_MATERIALS];
    const std::vector<float> &tex = ctx->host[RT_SLOT_TEXTURES];
    out.words.clear();
    out.materials = mats;
    out.packed8 = out.kept32 = 0;
    out.sourceBytes = tex.size() * sizeof(float);
    auto keepUpload = [&]() {
        out.words = tex;
        out.materials = mats;
        out.packed8 = out.kept32 = 0;
    };
    if (!enable) { keepUpload(); return; }
    float unorm[512];
    colourTables8(unorm);
    uint32_t unormBits[256];
    for (int k = 0; k < 256; ++k) std::memcpy(&unormBits[k], &unorm[k], 4);

    // Textures in use, by the word offset of their header (repackScene has validated every reference).
    const int materialCount = (int)(mats.size() / kMaterialWords);
    std::vector<int32_t> refs;
    for (int m = 0; m < materialCount; ++m)
        for (int k = 0; k < 6; ++k) {
            const int32_t ref = f2i(mats[(size_t)m * kMaterialWords + k]);
            if (ref != -1) refs.push_back(ref);
        }
    std::sort(refs.begin(), refs.end());
    refs.erase(std::unique(refs.begin(), refs.end()), refs.end());
    // References that overlap each other (a stream nobody's loader writes) could multiply the data: such a
    // scene keeps the upload as it is.
    const size_t budget = 2 * tex.size() + 64 * refs.size() + 64;
    std::vector<int32_t> movedTo(refs.size(), -1);
    for (size_t i = 0; i < refs.size(); ++i) {
        const size_t ref = (size_t)refs[i];
        const int w = f2i(tex[ref]), h = f2i(tex[ref + 1]), c = f2i(tex[ref + 2]);
        const size_t texels = (size_t)w * (size_t)h, count = texels * (size_t)c;
        const float *src = &tex[ref + 5];
        bool eightBit = c == 4 || c == 3;
        // (every word must be one of the 256 values exactly: the kernels return table[k])
        std::vector<uint32_t> packed;
        if (eightBit) {
            packed.resize(texels);
            for (size_t t = 0; t < texels && eightBit; ++t) {
                uint32_t word = 0;
                // a three-channel texel carries the red of the texel after it in its fourth byte: the word the
                // reference's alpha fetch reads there (what follows the LAST texel is kept as a raw word instead)
                const int take = (c == 3 && t + 1 == texels) ? 3 : 4;
                for (int ch = 0; ch < take; ++ch) {
                    const float f = src[t * (size_t)c + (size_t)ch];
                    uint32_t bits;
                    std::memcpy(&bits, &f, 4);
                    const int k = (f >= 0.0f && f <= 1.0f) ? (int)(f * 255.0f + 0.5f) : -1;
                    if (k < 0 || k > 255 || unormBits[k] != bits) { eightBit = false; break; }
                    word |= (uint32_t)k << (8 * ch);
                }
                packed[t] = word;
            }
        }
        if (out.words.size() + 5 + (eightBit ? texels + 1 : count + 3) > budget) { keepUpload(); return; }
        movedTo[i] = (int32_t)out.words.size();
        for (int k = 0; k < 5; ++k) out.words.push_back(tex[ref + (size_t)k]);
        if (eightBit) {
            const int32_t marked = c | (int32_t)kTexPacked8;
            std::memcpy(&out.words[out.words.size() - 3], &marked, 4);
            const size_t at = out.words.size();
            out.words.resize(at + texels);
            std::memcpy(&out.words[at], packed.data(), texels * 4);
            out.words.push_back(ref + 5 + count < tex.size() ? tex[ref + 5 + count] : 0.0f);   // the word behind the last texel, as uploaded
            ++out.packed8;
        } else {
            out.words.insert(out.words.end(), src, src + count);
            for (size_t k = 0; k < 3; ++k) out.words.push_back(ref + 5 + count + k < tex.size() ? tex[ref + 5 + count + k] : 0.0f);
            ++out.kept32;
        }
    }
    for (int m = 0; m < materialCount; ++m)
        for (int k = 0; k < 6; ++k) {
            const int32_t ref = f2i(mats[(size_t)m * kMaterialWords + k]);
            if (ref == -1) continue;
            const size_t i = (size_t)(std::lower_bound(refs.begin(), refs.end(), ref) - refs.begin());
            std::memcpy(&out.materials[(size_t)m * kMaterialWords + k], &movedTo[i], 4);
        }
}

// Binary tree over one model's triangles by binned surface-area splits (16 bins per axis on the
// triangle centres), for RT_OPT_WIDE_BVH 2 and 3.  Nodes are emitted (root first) through `emit(box,
// left, right, leafRef, leaf)` with child numbers relative to the model's first node; leaves hold one
// or two triangles (three where equal centres leave no split).  `wtris` receives the model's packed
// triangles in leaf order.  Boxes are the exact bounds of the vertices; the
// margin the device arithmetic needs is added when they are quantised.
template <typename Emit>
bool sahTree(const std::vector<float> &mdl, const ModelDesc &d, const PackedTris &tris, PackedTris &wtris, Emit emit) {
    const int T = d.identifiersCount;
    auto mark = std::chrono::steady_clock::now();
    struct Prim { float box[6]; float centre[3]; };
    std::vector<Prim> prims((size_t)T);
    for (int t = 0; t < T; ++t) {
        const int at = d.identBase + t * kIdentWords;
        Prim &p = prims[(size_t)t];
        for (int k = 0; k < 3; ++k) { p.box[k] = INFINITY; p.box[3 + k] = -INFINITY; }
        for (int c = 0; c < 3; ++c) {
            const float *v = &mdl[(size_t)d.vertBase + (size_t)f2i(mdl[(size_t)at + c]) * 8];
            for (int k = 0; k < 3; ++k) {
                if (!(std::fabs(v[k]) <= 1e15f)) return false;
                p.box[k] = std::min(p.box[k], v[k]);
                p.box[3 + k] = std::max(p.box[3 + k], v[k]);
            }
        }
        for (int k = 0; k < 3; ++k) p.centre[k] = 0.5f * p.box[k] + 0.5f * p.box[3 + k];
    }
    std::vector<int32_t> order((size_t)T);
    for (int t = 0; t < T; ++t) order[(size_t)t] = t;
    if (T >= 65536) commitLap("triangle bounds", mark);

    struct Node { float box[6]; int32_t left, right, lo, hi; };
    auto bounds = [&](int lo, int hi, float *box) {
        for (int k = 0; k < 3; ++k) { box[k] = INFINITY; box[3 + k] = -INFINITY; }
        for (int i = lo; i < hi; ++i)
            for (int k = 0; k < 3; ++k) {
                box[k] = std::min(box[k], prims[(size_t)order[(size_t)i]].box[k]);
                box[3 + k] = std::max(box[3 + k], prims[(size_t)order[(size_t)i]].box[3 + k]);
            }
    };
    auto halfArea = [](const float *b) {
        const double ex = (double)b[3] - b[0], ey = (double)b[4] - b[1], ez = (double)b[5] - b[2];
        return ex * ey + ey * ez + ez * ex;
    };
    constexpr int kBins = 16;
    const int workers = T >= 65536 ? buildThreads() : 1;
    // Runs fn(part, a, b) over `parts` equal pieces of [lo, hi), piece 0 on this thread.
    auto forked = [](int parts, int lo, int hi, auto fn) {
        const long long count = hi - lo;
        ThreadGroup pool((size_t)parts);
        for (int t = 1; t < parts; ++t) {
            const int a = lo + (int)(count * t / parts), b = lo + (int)(count * (t + 1) / parts);
            if (!pool.spawn(fn, t, a, b)) fn(t, a, b);   // no thread to be had: do the piece here
        }
        fn(0, lo, lo + (int)(count / parts));
        pool.join();
    };
    struct Bins {
        int counts[3][kBins];
        float boxes[3][kBins][6];
        float cb[6];
        void clear() {
            for (int k = 0; k < 3; ++k) { cb[k] = INFINITY; cb[3 + k] = -INFINITY; }
            for (auto &axis : counts) for (int &c : axis) c = 0;
            for (auto &axis : boxes) for (auto &bb : axis) for (int k = 0; k < 3; ++k) { bb[k] = INFINITY; bb[3 + k] = -INFINITY; }
        }
    };
    // Where to cut order[lo, hi): partitions the range and returns the first index of the right
    // part, or -1 for a leaf.  Touches nothing outside the range.  With mayFork the scans of a long
    // range are shared among threads; minima, maxima and counts merge exactly, so the cut is the same.
    auto split = [&](int lo, int hi, int depth, bool mayFork) -> int {
        const int count = hi - lo;
        if (count <= 2) return -1;
        const int parts = mayFork ? std::max(1, std::min(workers, count / 8192)) : 1;
        std::vector<Bins> partial((size_t)parts);
        for (Bins &pb : partial) pb.clear();
        forked(parts, lo, hi, [&](int part, int a, int b) {
            float *cb = partial[(size_t)part].cb;
            for (int i = a; i < b; ++i)
                for (int k = 0; k < 3; ++k) {
                    cb[k] = std::min(cb[k], prims[(size_t)order[(size_t)i]].centre[k]);
                    cb[3 + k] = std::max(cb[3 + k], prims[(size_t)order[(size_t)i]].centre[k]);
                }
        });
        float cb[6];
        for (int k = 0; k < 3; ++k) { cb[k] = INFINITY; cb[3 + k] = -INFINITY; }
        for (const Bins &pb : partial)
            for (int k = 0; k < 3; ++k) { cb[k] = std::min(cb[k], pb.cb[k]); cb[3 + k] = std::max(cb[3 + k], pb.cb[3 + k]); }
        double extents[3];
        for (int axis = 0; axis < 3; ++axis) extents[axis] = (double)cb[3 + axis] - cb[axis];
        double bestCost = INFINITY;
        int bestAxis = -1, bestBin = -1;
        if (depth < 48) {
            forked(parts, lo, hi, [&](int part, int a, int b) {
                Bins &mine = partial[(size_t)part];
                for (int i = a; i < b; ++i) {
                    const Prim &pr = prims[(size_t)order[(size_t)i]];
                    for (int axis = 0; axis < 3; ++axis) {
                        if (!(extents[axis] > 0.0)) continue;
                        int bin = (int)(kBins * (((double)pr.centre[axis] - cb[axis]) / extents[axis]));
                        bin = std::min(std::max(bin, 0), kBins - 1);
                        mine.counts[axis][bin]++;
                        float *bb = mine.boxes[axis][bin];
                        for (int k = 0; k < 3; ++k) { bb[k] = std::min(bb[k], pr.box[k]); bb[3 + k] = std::max(bb[3 + k], pr.box[3 + k]); }
                    }
                }
            });
            Bins &all = partial[0];
            for (size_t t = 1; t < partial.size(); ++t)
                for (int axis = 0; axis < 3; ++axis)
                    for (int bin = 0; bin < kBins; ++bin) {
                        all.counts[axis][bin] += partial[t].counts[axis][bin];
                        for (int k = 0; k < 3; ++k) {
                            all.boxes[axis][bin][k] = std::min(all.boxes[axis][bin][k], partial[t].boxes[axis][bin][k]);
                            all.boxes[axis][bin][3 + k] = std::max(all.boxes[axis][bin][3 + k], partial[t].boxes[axis][bin][3 + k]);
                        }
                    }
            for (int axis = 0; axis < 3; ++axis) {
                if (!(extents[axis] > 0.0)) continue;
                const int *counts = all.counts[axis];
                float (*boxes)[6] = all.boxes[axis];
                double rightArea[kBins];
                int rightCount[kBins];
                float acc[6];
                for (int k = 0; k < 3; ++k) { acc[k] = INFINITY; acc[3 + k] = -INFINITY; }
                int c = 0;
                for (int bin = kBins - 1; bin > 0; --bin) {
                    for (int k = 0; k < 3; ++k) { acc[k] = std::min(acc[k], boxes[bin][k]); acc[3 + k] = std::max(acc[3 + k], boxes[bin][3 + k]); }
                    c += counts[bin];
                    rightArea[bin] = c ? halfArea(acc) : 0.0;
                    rightCount[bin] = c;
                }
                for (int k = 0; k < 3; ++k) { acc[k] = INFINITY; acc[3 + k] = -INFINITY; }
                c = 0;
                for (int bin = 0; bin < kBins - 1; ++bin) {
                    for (int k = 0; k < 3; ++k) { acc[k] = std::min(acc[k], boxes[bin][k]); acc[3 + k] = std::max(acc[3 + k], boxes[bin][3 + k]); }
                    c += counts[bin];
                    if (c == 0 || rightCount[bin + 1] == 0) continue;
                    const double cost = halfArea(acc) * c + rightArea[bin + 1] * rightCount[bin + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = bin; }
                }
            }
        }
        int mid = -1;
        if (bestAxis >= 0) {
            const double extent = (double)cb[3 + bestAxis] - cb[bestAxis];
            auto first = order.begin() + lo, last = order.begin() + hi;
            auto cut = std::partition(first, last, [&](int32_t t) {
                int bin = (int)(kBins * (((double)prims[(size_t)t].centre[bestAxis] - cb[bestAxis]) / extent));
                return std::min(std::max(bin, 0), kBins - 1) <= bestBin;
            });
            mid = (int)(cut - order.begin());
        }
        if (mid <= lo || mid >= hi) mid = count > 3 ? lo + count / 2 : -1;   // equal centres: halve, or keep three together
        return mid;
    };
    // Builds the subtree over order[lo, hi) into `out` (child numbers index `out`): a node is
    // numbered when it is taken off the stack, its left subtree follows it directly, and the
    // parent's right link is patched when that child is numbered.  Ranges of at most `defer`
    // triangles are not expanded but listed in `later` (with the node that stands for them).
    struct Todo { int lo, hi, parent; bool isRight; int depth; };
    struct Deferred { int lo, hi, depth; int32_t node; };
    auto build = [&](int lo0, int hi0, int depth0, std::vector<Node> &out, int defer, std::vector<Deferred> *later) {
        std::vector<Todo> todo{{lo0, hi0, -1, false, depth0}};
        while (!todo.empty()) {
            const Todo it = todo.back();
            todo.pop_back();
            const int32_t self = (int32_t)out.size();
            out.emplace_back();
            if (it.parent >= 0) (it.isRight ? out[(size_t)it.parent].right : out[(size_t)it.parent].left) = self;
            Node n{};
            n.lo = it.lo;
            n.hi = it.hi;
            n.left = n.right = -1;
            if (later != nullptr && it.hi - it.lo <= defer && it.hi - it.lo > 2) {
                later->push_back({it.lo, it.hi, it.depth, self});
                out[(size_t)self] = n;
                continue;
            }
            const int pieces = later != nullptr ? std::max(1, std::min(workers, (it.hi - it.lo) / 8192)) : 1;
            if (pieces > 1) {   // a long range near the top: the scan is shared, minima and maxima merge exactly
                std::vector<std::array<float, 6>> partial((size_t)pieces);
                forked(pieces, it.lo, it.hi, [&](int part, int a, int b) { bounds(a, b, partial[(size_t)part].data()); });
                for (int k = 0; k < 3; ++k) { n.box[k] = INFINITY; n.box[3 + k] = -INFINITY; }
                for (const auto &pbox : partial)
                    for (int k = 0; k < 3; ++k) { n.box[k] = std::min(n.box[k], pbox[(size_t)k]); n.box[3 + k] = std::max(n.box[3 + k], pbox[(size_t)(3 + k)]); }
            } else {
                bounds(it.lo, it.hi, n.box);
            }
            const int mid = split(it.lo, it.hi, it.depth, later != nullptr);
            out[(size_t)self] = n;
            if (mid > 0) {
                todo.push_back({mid, it.hi, self, true, it.depth + 1});
                todo.push_back({it.lo, mid, self, false, it.depth + 1});
            }
        }
    };
    // Large models: the top of the tree is cut on this thread until the ranges are small, then the
    // ranges (disjoint parts of `order`) are built side by side and appended in range order, so the
    // result does not depend on the number of threads.
    std::vector<Node> nodes;
    nodes.reserve((size_t)T);
    if (workers <= 1) {
        build(0, T, 0, nodes, 0, nullptr);
    } else {
        std::vector<Deferred> later;
        build(0, T, 0, nodes, std::max(T / (workers * 4), 8192), &later);
        std::vector<std::vector<Node>> parts(later.size());
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};   // an allocation failure must not end the process from inside a thread
        auto work = [&]() {
            try {
                for (size_t k = next.fetch_add(1); k < later.size(); k = next.fetch_add(1))
                    build(later[k].lo, later[k].hi, later[k].depth, parts[k], 0, nullptr);
            } catch (...) {
                failed = true;
            }
        };
        {
            ThreadGroup pool((size_t)workers);
            for (int w = 1; w < workers; ++w)
                if (!pool.spawn(work)) break;   // the threads that did start, and this one, share the ranges
            work();
        }
        if (failed) throw std::bad_alloc();   // reported as RT_ERR_NOMEM by the caller
        for (size_t k = 0; k < later.size(); ++k) {
            // the part's root takes the place of the node that stood for the range; the rest is appended
            const int32_t shift = (int32_t)nodes.size() - 1;
            auto moved = [&](int32_t c) { return c < 0 ? c : c + shift; };
            const std::vector<Node> &part = parts[k];
            Node root = part[0];
            root.left = moved(root.left);
            root.right = moved(root.right);
            nodes[(size_t)later[k].node] = root;
            for (size_t i = 1; i < part.size(); ++i) {
                Node n = part[i];
                n.left = moved(n.left);
                n.right = moved(n.right);
                nodes.push_back(n);
            }
        }
    }
    if (T >= 65536) commitLap("binary tree", mark);
    // The triangle records in this tree's leaf order, each naming its place in `tris`; then the nodes.
    for (int i = 0; i < T; ++i) {
        const int32_t at = d.triBase + order[(size_t)i];
        PackedTri pt = tris[(size_t)at];
        pt.ident = at;
        wtris[(size_t)d.triBase + (size_t)i] = pt;
    }
    for (const Node &n : nodes) {
        const bool leaf = n.left < 0;
        const int32_t leafRef = leaf ? ~(int32_t)((((uint32_t)(d.triBase + n.lo)) << 2) | (uint32_t)(n.hi - n.lo)) : 0;
        emit(n.box, n.left, n.right, leafRef, leaf);
    }
    if (T >= 65536) commitLap("leaf-order copy + emit", mark);
    return true;
}

// Binary trees of all models in one array (children are indices into the array).
struct BinaryTrees {
    std::vector<BinNode> bin;
    std::vector<std::pair<int32_t, ModelDesc *>> roots;   // root node and model of every tree that is more than one leaf
    double reach = 0.0;                                    // largest coordinate magnitude of any root box
};

// Fills `trees` with every model's binary tree: the library's own binned-SAH tree when `rebuild` (then
// `wtris` receives the triangle records in that tree's leaf order), else the uploaded one.  A model
// whose tree is a single leaf gets that leaf's reference in wideRootRef and no entry in `roots`.
int binaryTreesOfModels(rt_scene_host *ctx, std::vector<ModelDesc> &descs, const PackedTris &tris, PackedTris &wtris, bool rebuild, BinaryTrees &trees) {
    const std::vector<float> &mdl = ctx->host[RT_SLOT_MODELS];
    std::vector<BinNode> &bin = trees.bin;
    auto &roots = trees.roots;
    double &reach = trees.reach;
    // Triangles in the order of the trees walked, each naming its index in `tris` (what hit records
    // and shading use): the uploaded order unless a model's tree is rebuilt below.
    auto mark = std::chrono::steady_clock::now();
    wtris = tris;
    for (size_t i = 0; i < wtris.size(); ++i) wtris[i].ident = (int32_t)i;
    commitLap("triangle copy", mark);
    using Bin = BinNode;
    // The library's own trees (binned surface-area splits, leaves of one or two triangles, a copy of the triangle
    // records in leaf order) are built side by side, one model per thread on top of each build's own threads
    // (a single build does not keep sixteen cores busy), each into nodes of its own with child numbers relative
    // to its root; they are strung together in model order below.
    std::vector<std::vector<Bin>> own(descs.size());
    std::vector<char> failed(descs.size(), 0);
    if (rebuild) {
        auto build = [&](size_t m) {
            const ModelDesc &d = descs[m];
            std::vector<Bin> &nodes = own[m];
            if (!sahTree(mdl, d, tris, wtris, [&](const float *box, int32_t left, int32_t right, int32_t leafRef, bool leaf) {
                    Bin b{};
                    for (int k = 0; k < 6; ++k) b.box[k] = box[k];
                    b.left = leaf ? -1 : left;
                    b.right = leaf ? -1 : right;
                    b.leafRef = leafRef;
                    b.leaf = leaf;
                    nodes.push_back(b);
                })) failed[m] = 1;
        };
        std::vector<size_t> jobs;
        for (size_t m = 0; m < descs.size(); ++m)
            if (descs[m].nodesCount > 0 && descs[m].identifiersCount > 0) jobs.push_back(m);
        if (ctx->deviceTree != nullptr) {
            // Trees from the device, one model after the other (each build fills the device by itself); the
            // triangle records are put in the leaf order it reports.
            std::vector<int32_t> order;
            for (size_t m : jobs) {
                const ModelDesc &d = descs[m];
                const int rc = ctx->deviceTree(ctx, d, own[m], order);
                if (rc == RT_ERR_NOMEM) {   // no room on the device for the build's scratch: this model and the rest on the host
                    own[m].clear();
                    break;
                }
                if (rc != RT_OK) return rc;
                if (order.size() != (size_t)d.identifiersCount) return RT_ERR_SCENE;
                for (int i = 0; i < d.identifiersCount; ++i) {
                    const int32_t at = d.triBase + order[(size_t)i];
                    PackedTri pt = tris[(size_t)at];
                    pt.ident = at;
                    wtris[(size_t)d.triBase + (size_t)i] = pt;
                }
            }
            jobs.erase(std::remove_if(jobs.begin(), jobs.end(), [&](size_t m) { return !own[m].empty(); }), jobs.end());
            commitLap("binary trees on the device", mark);
        }
        ThreadGroup pool(jobs.size());
        const bool sideBySide = jobs.size() > 1 && buildThreads() > 1;
        for (size_t j = 1; j < jobs.size(); ++j)
            if (!sideBySide || !pool.spawn(build, jobs[j])) build(jobs[j]);
        if (!jobs.empty()) build(jobs[0]);
        pool.join();
        for (size_t m = 0; m < descs.size(); ++m)
            if (failed[m]) return RT_ERR_SCENE;
    }
    for (size_t m = 0; m < descs.size(); ++m) {
        ModelDesc &d = descs[m];
        d.wideRootRef = d.rootRef;   // a tree that is one leaf: same reference, wtris being a copy of tris there
        if (d.nodesCount <= 0) continue;
        if (rebuild && d.identifiersCount > 0) {
            const int32_t base = (int32_t)bin.size();
            for (Bin b : own[m]) {
                if (!b.leaf) { b.left += base; b.right += base; }
                bin.push_back(b);
            }
            std::vector<Bin>().swap(own[m]);
            if (bin[(size_t)base].leaf) {
                d.wideRootRef = bin[(size_t)base].leafRef;
                continue;
            }
            roots.push_back({base, &d});
            for (int k = 0; k < 6; ++k) reach = std::max(reach, (double)std::fabs(bin[(size_t)base].box[k]));
            continue;
        }
        const int32_t base = (int32_t)bin.size();
        bin.resize(bin.size() + (size_t)d.nodesCount);
        for (int n = 0; n < d.nodesCount; ++n) {
            const float *w = &mdl[(size_t)d.nodeBase + (size_t)n * kNodeWords];
            Bin &b = bin[(size_t)base + n];
            for (int k = 0; k < 6; ++k) b.box[k] = w[k];
            b.leaf = w[8] != 0.0f;
            if (b.leaf) {
                const int first = f2i(w[6]), count = f2i(w[7]) - first;
                b.left = b.right = -1;
                b.leafRef = ~(int32_t)((((uint32_t)(d.triBase + first)) << 2) | (uint32_t)count);
            } else {
                b.left = base + f2i(w[6]);
                b.right = base + f2i(w[7]);
                b.leafRef = 0;
            }
        }
        if (bin[(size_t)base].leaf) continue;   // rootRef already names the leaf
        roots.push_back({base, &d});
        for (int k = 0; k < 6; ++k) reach = std::max(reach, (double)std::fabs(bin[(size_t)base].box[k]));
    }
    commitLap("binary trees, all models", mark);
    return RT_OK;
}

// Four-wide tree of every model, collapsed from the binary tree of the model stream: a node adopts
// the children of its child with the largest box until it has four (or only leaves are left).
// `descs` must have passed repackScene; wideRootRef receives the new tree's root.
int buildWideNodes(rt_scene_host *ctx, std::vector<ModelDesc> &descs, std::vector<WideNode> &wnodes, const PackedTris &tris, PackedTris &wtris) {
    ctx->wideStackNeed = 0;
    BinaryTrees trees;
    const int treesRc = binaryTreesOfModels(ctx, descs, tris, wtris, ctx->wideBvh >= 2, trees);
    if (treesRc != RT_OK) return treesRc;
    std::vector<BinNode> &bin = trees.bin;
    auto &roots = trees.roots;
    const double reach = trees.reach;
    auto mark = std::chrono::steady_clock::now();
    if (roots.empty()) return RT_OK;
    if (!(reach <= 1e15)) return RT_ERR_SCENE;   // also catches NaN
    // Margin around every box for the rounding of the plane arithmetic on the device, whose operands
    // are as large as the distance from the ray origin to the scene: four units in the last place of
    // the scene's largest coordinate.
    const double pad = std::ldexp(reach, -21);

    auto leafCount = [&](int32_t n) { return (int)((~(uint32_t)bin[(size_t)n].leafRef) & 3u); };
    auto area = [&](int32_t n) {
        const float *b = bin[(size_t)n].box;
        const double ex = (double)b[3] - b[0], ey = (double)b[4] - b[1], ez = (double)b[5] - b[2];
        return ex * ey + ey * ez + ez * ex;
    };
    // Which binary nodes become the children of a wide node.  Default: the cut of at most four descendants that
    // minimises the expected cost of walking the collapsed tree (surface-area heuristic over WIDE nodes, solved
    // bottom-up: cost[n][i] = cheapest way to stand for subtree n with at most i + 1 slots of its parent; a
    // visit of a wide node costs kNodeCost x its area, of a leaf kLeafCost x its area).  RT_COLLAPSE=greedy
    // keeps the earlier rule: a node adopts the children of its child with the largest box until it has four.
    // (Every leaf is visited through exactly one slot whatever the cut, so the leaves' share of the cost is the same
    // for all cuts: only the areas of the nodes that become wide nodes decide.  Measured: any leaf weight, same trees.)
    static const bool greedyCollapse = std::getenv("RT_COLLAPSE") != nullptr && std::strcmp(std::getenv("RT_COLLAPSE"), "greedy") == 0;
    struct Cut { float cost[3]; uint8_t split[5]; };   // split[j], j = 2..4: slots given to the left child when j are shared out; 0 = use j - 1
    std::vector<Cut> cuts;
    if (!greedyCollapse) {
        cuts.resize(bin.size());
        // (one tree per thread: a tree's nodes are its own)
        auto solve = [&](int32_t root) {
        std::vector<std::pair<int32_t, bool>> walk{{root, false}};   // (node, children done)
        while (!walk.empty()) {
            const auto [n, done] = walk.back();
            walk.pop_back();
            const BinNode &b = bin[(size_t)n];
            Cut &c = cuts[(size_t)n];
            if (b.leaf) {
                leafCutCost(b.box, leafCount(n) == 0, c.cost);
                continue;
            }
            if (!done) {
                walk.push_back({n, true});
                walk.push_back({b.left, false});
                walk.push_back({b.right, false});
                continue;
            }
            const Cut &l = cuts[(size_t)b.left], &r = cuts[(size_t)b.right];
            innerCutCost(b.box, l.cost, r.cost, c.cost, c.split);
        }
        };
        ThreadGroup pool(roots.size());
        const bool threaded = bin.size() >= 65536 && buildThreads() > 1;
        for (size_t k = 1; k < roots.size(); ++k)
            if (!threaded || !pool.spawn(solve, roots[k].first)) solve(roots[k].first);
        solve(roots[0].first);
        pool.join();
    }
    // The children of the wide node that stands for binary node n, by the cuts above.
    auto cutChildren = [&](int32_t n, int32_t *kids) -> int {
        int nk = 0;
        struct Share { int32_t node; int slots; };
        Share stack[8];
        int sp = 0;
        const BinNode &top = bin[(size_t)n];
        const int k4 = cuts[(size_t)n].split[4] & 0x7f;
        stack[sp++] = {top.right, 4 - k4};
        stack[sp++] = {top.left, k4};
        while (sp > 0) {
            Share s = stack[--sp];
            const BinNode &b = bin[(size_t)s.node];
            // (a marked split says the subtree is as cheap with one slot fewer: step down until a real split or one slot)
            while (!b.leaf && s.slots > 1 && (cuts[(size_t)s.node].split[s.slots] & 0x80)) --s.slots;
            if (b.leaf || s.slots == 1) { kids[nk++] = s.node; continue; }
            const int k = cuts[(size_t)s.node].split[s.slots] & 0x7f;
            stack[sp++] = {b.right, s.slots - k};
            stack[sp++] = {b.left, k};
        }
        return nk;
    };
    struct Item { int32_t node; int32_t wide; int held; };   // binary node to expand, its slot, entries its ancestors keep stacked
    std::vector<Item> todo;
    struct Family { int32_t kids[4]; int count; };   // the binary nodes a wide node's children stand for
    std::vector<Family> families(wnodes.size());
    for (auto &rootOf : roots) {
        rootOf.second->wideRootRef = (int32_t)wnodes.size();
        wnodes.emplace_back();
        families.emplace_back();
        todo.push_back({rootOf.first, rootOf.second->wideRootRef, 0});
    }
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        int32_t kids[4] = {bin[(size_t)it.node].left, bin[(size_t)it.node].right, -1, -1};
        int nk = 2;
        if (!greedyCollapse) nk = cutChildren(it.node, kids);
        while (greedyCollapse && nk < 4) {
            int best = -1;
            double bestArea = -1.0;
            for (int i = 0; i < nk; ++i) {
                if (bin[(size_t)kids[i]].leaf) continue;
                const double a = area(kids[i]);
                if (best < 0 || a > bestArea) { best = i; bestArea = a; }
            }
            if (best < 0) break;
            const int32_t n = kids[best];
            kids[best] = bin[(size_t)n].left;
            kids[nk++] = bin[(size_t)n].right;
        }
        // Leaves without triangles need no slot.
        int kept = 0;
        for (int i = 0; i < nk; ++i)
            if (!(bin[(size_t)kids[i]].leaf && leafCount(kids[i]) == 0)) kids[kept++] = kids[i];
        nk = kept;
        Family fam{{-1, -1, -1, -1}, nk};
        int32_t refs[4] = {-1, -1, -1, -1};   // ~0: a leaf without triangles
        const int held = it.held + std::max(nk - 1, 0);
        ctx->wideStackNeed = std::max(ctx->wideStackNeed, held);
        for (int i = 0; i < nk; ++i) {
            fam.kids[i] = kids[i];
            if (bin[(size_t)kids[i]].leaf) {
                refs[i] = bin[(size_t)kids[i]].leafRef;
            } else {
                refs[i] = (int32_t)wnodes.size();
                wnodes.emplace_back();
                families.emplace_back();
                todo.push_back({kids[i], refs[i], held});
            }
        }
        WideNode &wn = wnodes[(size_t)it.wide];
        wn.c0 = refs[0]; wn.c1 = refs[1]; wn.c2 = refs[2]; wn.c3 = refs[3];
        families[(size_t)it.wide] = fam;
    }
    commitLap("four-wide structure", mark);

    // The child boxes of every node as 8-bit plane numbers: independent per node, shared among threads.
    std::atomic<bool> bad{false};
    auto quantise = [&](size_t first, size_t last) {
        for (size_t w = first; w < last; ++w) {
            const Family &fam = families[w];
            const int nk = fam.count;
            const int32_t *kids = fam.kids;
            WideNode &wn = wnodes[w];
            for (int i = 0; i < nk; ++i)
                for (int k = 0; k < 6; ++k)
                    if (!(std::fabs(bin[(size_t)kids[i]].box[k]) <= 1e15f)) { bad = true; return; }
            for (int a = 0; a < 3; ++a) {
                double lower = 0.0, upper = 0.0;
                for (int i = 0; i < nk; ++i) {
                    const double mn = bin[(size_t)kids[i]].box[a], mx = bin[(size_t)kids[i]].box[3 + a];
                    if (mn > mx) { bad = true; return; }
                    if (i == 0 || mn < lower) lower = mn;
                    if (i == 0 || mx > upper) upper = mx;
                }
                float o = (float)(lower - pad);
                if ((double)o > lower - pad) o = std::nextafterf(o, -INFINITY);
                const double extent = (upper + pad) - (double)o;
                int e = -60;
                if (extent > 0.0) e = std::max(-60, (int)std::ceil(std::log2(extent / 255.0)));
                for (;; ++e) {
                    const double step = std::ldexp(1.0, e);
                    bool fits = true;
                    uint32_t wlo = 0, whi = 0;
                    for (int i = 0; i < 4 && fits; ++i) {
                        uint32_t qlo = 255, qhi = 0;   // unused slot: inverted
                        if (i < nk) {
                            const double mn = (double)bin[(size_t)kids[i]].box[a] - pad, mx = (double)bin[(size_t)kids[i]].box[3 + a] + pad;
                            long ql = (long)std::floor((mn - (double)o) / step), qh = (long)std::ceil((mx - (double)o) / step);
                            ql = std::max(0l, std::min(ql, 255l));
                            while (ql > 0 && (double)o + (double)ql * step > mn) --ql;
                            qh = std::max(qh, 0l);
                            while (qh <= 255 && (double)o + (double)qh * step < mx) ++qh;
                            if (qh > 255) { fits = false; break; }
                            qlo = (uint32_t)ql;
                            qhi = (uint32_t)qh;
                        }
                        wlo |= qlo << (8 * i);
                        whi |= qhi << (8 * i);
                    }
                    if (fits) {
                        wn.o[a] = o;
                        wn.s[a] = (float)step;
                        wn.lo[a] = wlo;
                        wn.hi[a] = whi;
                        break;
                    }
                }
            }
        }
    };
    {
        const size_t total = wnodes.size();
        const int workers = total >= 32768 ? buildThreads() : 1;
        ThreadGroup pool((size_t)workers);
        for (int t = 1; t < workers; ++t) {
            const size_t first = total * (size_t)t / (size_t)workers, last = total * (size_t)(t + 1) / (size_t)workers;
            if (!pool.spawn(quantise, first, last)) quantise(first, last);
        }
        quantise(0, total / (size_t)workers);
        pool.join();
    }
    if (bad) return RT_ERR_SCENE;
    commitLap("child boxes to 8 bits", mark);
    return RT_OK;
}

// Re-packs the leaves of the four-wide trees as triangle pairs (TriPair) and rewrites every leaf
// reference, in the nodes and in the models' roots, from (first triangle, triangles) to (first pair, pairs).
void pairLeaves(std::vector<WideNode> &wnodes, std::vector<ModelDesc> &descs, const PackedTris &wtris, std::vector<TriPair> &wpairs) {
    wpairs.clear();
    wpairs.reserve(wtris.size() * 2 / 3 + 1);
    auto paired = [&](int32_t ref) -> int32_t {
        if (ref >= 0) return ref;
        const uint32_t code = ~(uint32_t)ref, first = code >> 2, count = code & 3u;
        if (count == 0u) return ref;
        const uint32_t at = (uint32_t)wpairs.size(), pairs = (count + 1u) / 2u;
        for (uint32_t p = 0; p < pairs; ++p) {
            TriPair tp;
            std::memset(&tp, 0, sizeof tp);
            tp.ident[0] = tp.ident[1] = (int32_t)kMiss;
            for (uint32_t h = 0; h < 2u && 2u * p + h < count; ++h) {
                const PackedTri &t = wtris[(size_t)first + 2u * p + h];
                tp.ax[h] = t.a[0]; tp.ay[h] = t.a[1]; tp.az[h] = t.a[2];
                tp.abx[h] = t.ab[0]; tp.aby[h] = t.ab[1]; tp.abz[h] = t.ab[2];
                tp.acx[h] = t.ac[0]; tp.acy[h] = t.ac[1]; tp.acz[h] = t.ac[2]; tp.ident[h] = t.ident;
            }
            wpairs.push_back(tp);
        }
        return ~(int32_t)((at << 2) | pairs);
    };
    for (WideNode &wn : wnodes) {
        wn.c0 = paired(wn.c0); wn.c1 = paired(wn.c1); wn.c2 = paired(wn.c2); wn.c3 = paired(wn.c3);
    }
    for (ModelDesc &d : descs)
        if (d.nodesCount > 0) d.wideRootRef = paired(d.wideRootRef);
}

// Eight-wide tree of every model (RT_OPT_WIDE_BVH 4), collapsed from the library's binary trees: a
// node adopts the children of its child with the largest box until it has eight.  Layout and child
// encoding: Wide8Node in rt_device.h.  The inner children of a node get consecutive node numbers and
// the triangles of its leaf children consecutive places in `w8tris`, both in slot order; slots are
// dealt so that a child lying towards (+x, +y, +z) of the node's centre sits in the slot whose number
// has those bits set, which lets the device visit children front to back by slot number alone.
int buildWide8(rt_scene_host *ctx, std::vector<ModelDesc> &descs, std::vector<Wide8Node> &w8, const PackedTris &tris, PackedTris &w8tris) {
    ctx->wideStackNeed = 0;
    BinaryTrees trees;
    PackedTris leafTris;   // in the binary trees' leaf order
    const int treesRc = binaryTreesOfModels(ctx, descs, tris, leafTris, true, trees);
    if (treesRc != RT_OK) return treesRc;
    const std::vector<BinNode> &bin = trees.bin;
    auto mark = std::chrono::steady_clock::now();
    if (!(trees.reach <= 1e15)) return RT_ERR_SCENE;   // also catches NaN
    const double pad = std::ldexp(trees.reach, -21);   // see buildWideNodes

    w8tris.clear();
    w8tris.reserve(tris.size());
    auto leafFirst = [&](int32_t n) { return (~(uint32_t)bin[(size_t)n].leafRef) >> 2; };
    auto leafCount = [&](int32_t n) { return (int)((~(uint32_t)bin[(size_t)n].leafRef) & 3u); };
    // Appends the triangles of a leaf reference (into leafTris) and returns the reference into w8tris.
    auto moveLeaf = [&](int32_t ref) -> int32_t {
        const uint32_t code = ~(uint32_t)ref, first = code >> 2, count = code & 3u;
        const uint32_t at = (uint32_t)w8tris.size();
        for (uint32_t k = 0; k < count; ++k) w8tris.push_back(leafTris[(size_t)first + k]);
        return ~(int32_t)((at << 2) | count);
    };
    // Models whose tree is one leaf keep a leaf reference as their root; a leaf without triangles
    // (and a model without triangles) has nothing to walk.
    for (ModelDesc &d : descs) {
        if (d.nodesCount <= 0) continue;
        bool hasTree = false;
        for (const auto &r : trees.roots) hasTree = hasTree || r.second == &d;
        if (hasTree) continue;
        const uint32_t count = (~(uint32_t)d.wideRootRef) & 3u;
        d.wideRootRef = (d.wideRootRef < 0 && count > 0) ? moveLeaf(d.wideRootRef) : kNoRoot;
    }
    auto area = [&](int32_t n) {
        const float *b = bin[(size_t)n].box;
        const double ex = (double)b[3] - b[0], ey = (double)b[4] - b[1], ez = (double)b[5] - b[2];
        return ex * ey + ey * ez + ez * ex;
    };
    struct Family { int32_t kids[8]; };   // binary node standing behind each slot, -1 = empty
    std::vector<Family> families;
    struct Item { int32_t node; uint32_t wide; int depth; };
    std::vector<Item> todo;
    auto newNode = [&]() -> uint32_t {
        w8.emplace_back();
        families.push_back(Family{{-1, -1, -1, -1, -1, -1, -1, -1}});
        return (uint32_t)(w8.size() - 1);
    };
    for (auto &rootOf : trees.roots) {
        const uint32_t root = newNode();
        rootOf.second->wideRootRef = (int32_t)root;
        todo.push_back({rootOf.first, root, 0});
    }
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        ctx->wideStackNeed = std::max(ctx->wideStackNeed, it.depth);   // one stack entry per ancestor at most
        int32_t kids[8] = {bin[(size_t)it.node].left, bin[(size_t)it.node].right, -1, -1, -1, -1, -1, -1};
        int nk = 2;
        while (nk < 8) {
            int best = -1;
            double bestArea = -1.0;
            for (int i = 0; i < nk; ++i) {
                if (bin[(size_t)kids[i]].leaf) continue;
                const double a = area(kids[i]);
                if (best < 0 || a > bestArea) { best = i; bestArea = a; }
            }
            if (best < 0) break;
            const int32_t n = kids[best];
            kids[best] = bin[(size_t)n].left;
            kids[nk++] = bin[(size_t)n].right;
        }
        int kept = 0;
        for (int i = 0; i < nk; ++i)
            if (!(bin[(size_t)kids[i]].leaf && leafCount(kids[i]) == 0)) kids[kept++] = kids[i];
        nk = kept;
        // Slots: greedily the (child, slot) pair with the largest projection of the child's centre
        // (relative to the centre of all children) on the slot's diagonal, until every child sits.
        double centre[3] = {0.0, 0.0, 0.0}, lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        for (int i = 0; i < nk; ++i)
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], (double)bin[(size_t)kids[i]].box[a]);
                hi[a] = std::max(hi[a], (double)bin[(size_t)kids[i]].box[3 + a]);
            }
        for (int a = 0; a < 3; ++a) centre[a] = 0.5 * (lo[a] + hi[a]);
        double cost[8][8];
        for (int i = 0; i < nk; ++i) {
            double rel[3];
            for (int a = 0; a < 3; ++a) rel[a] = 0.5 * ((double)bin[(size_t)kids[i]].box[a] + (double)bin[(size_t)kids[i]].box[3 + a]) - centre[a];
            for (int sl = 0; sl < 8; ++sl) cost[i][sl] = ((sl & 1) ? rel[0] : -rel[0]) + ((sl & 2) ? rel[1] : -rel[1]) + ((sl & 4) ? rel[2] : -rel[2]);
        }
        Family fam{{-1, -1, -1, -1, -1, -1, -1, -1}};
        bool seated[8] = {false, false, false, false, false, false, false, false};
        for (int round = 0; round < nk; ++round) {
            int bi = -1, bs = -1;
            for (int i = 0; i < nk; ++i) {
                if (seated[i]) continue;
                for (int sl = 0; sl < 8; ++sl)
                    if (fam.kids[sl] < 0 && (bi < 0 || cost[i][sl] > cost[bi][bs])) { bi = i; bs = sl; }
            }
            seated[bi] = true;
            fam.kids[bs] = kids[bi];
        }
        Wide8Node wn{};
        uint32_t imask = 0, triCount = 0;
        uint8_t meta[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t triBase = (uint32_t)w8tris.size();
        for (int sl = 0; sl < 8; ++sl) {
            const int32_t kid = fam.kids[sl];
            if (kid < 0) continue;
            if (bin[(size_t)kid].leaf) {
                const int count = leafCount(kid);
                meta[sl] = (uint8_t)((((1u << count) - 1u) << 5) | triCount);
                for (int k = 0; k < count; ++k) w8tris.push_back(leafTris[(size_t)leafFirst(kid) + (size_t)k]);
                triCount += (uint32_t)count;
            } else {
                meta[sl] = (uint8_t)(0x38u + (uint32_t)sl);   // one bit, at place 24 + slot
                imask |= 1u << sl;
            }
        }
        wn.childBase = (uint32_t)w8.size();
        wn.triBase = triBase;
        wn.meta[0] = (uint32_t)meta[0] | ((uint32_t)meta[1] << 8) | ((uint32_t)meta[2] << 16) | ((uint32_t)meta[3] << 24);
        wn.meta[1] = (uint32_t)meta[4] | ((uint32_t)meta[5] << 8) | ((uint32_t)meta[6] << 16) | ((uint32_t)meta[7] << 24);
        wn.expImask = imask << 24;
        // inner children: consecutive numbers in slot order; expanded in that order too (depth first)
        std::vector<Item> mine;
        for (int sl = 0; sl < 8; ++sl) {
            const int32_t kid = fam.kids[sl];
            if (kid < 0 || bin[(size_t)kid].leaf) continue;
            const uint32_t child = newNode();
            mine.push_back({kid, child, it.depth + 1});
        }
        for (size_t k = mine.size(); k-- > 0;) todo.push_back(mine[k]);
        w8[(size_t)it.wide] = wn;
        families[(size_t)it.wide] = fam;
    }
    commitLap("eight-wide structure", mark);

    // Child boxes as 8-bit plane numbers on a power-of-two grid per axis, rounded outwards.
    std::atomic<bool> bad{false};
    auto quantise = [&](size_t first, size_t last) {
        for (size_t w = first; w < last; ++w) {
            const int32_t *kids = families[w].kids;
            Wide8Node &wn = w8[w];
            for (int a = 0; a < 3; ++a) {
                double lower = 0.0, upper = 0.0;
                bool any = false;
                for (int i = 0; i < 8; ++i) {
                    if (kids[i] < 0) continue;
                    const double mn = bin[(size_t)kids[i]].box[a], mx = bin[(size_t)kids[i]].box[3 + a];
                    if (!(std::fabs(mn) <= 1e15) || !(std::fabs(mx) <= 1e15) || mn > mx) { bad = true; return; }
                    if (!any || mn < lower) lower = mn;
                    if (!any || mx > upper) upper = mx;
                    any = true;
                }
                float o = (float)(lower - pad);
                if ((double)o > lower - pad) o = std::nextafterf(o, -INFINITY);
                const double extent = (upper + pad) - (double)o;
                int e = -60;
                if (extent > 0.0) e = std::max(-60, (int)std::ceil(std::log2(extent / 255.0)));
                for (;; ++e) {
                    const double step = std::ldexp(1.0, e);
                    bool fits = true;
                    uint32_t wlo[2] = {0, 0}, whi[2] = {0, 0};
                    for (int i = 0; i < 8 && fits; ++i) {
                        uint32_t qlo = 255, qhi = 0;   // unused slot: inverted
                        if (kids[i] >= 0) {
                            const double mn = (double)bin[(size_t)kids[i]].box[a] - pad, mx = (double)bin[(size_t)kids[i]].box[3 + a] + pad;
                            long ql = (long)std::floor((mn - (double)o) / step), qh = (long)std::ceil((mx - (double)o) / step);
                            ql = std::max(0l, std::min(ql, 255l));
                            while (ql > 0 && (double)o + (double)ql * step > mn) --ql;
                            qh = std::max(qh, 0l);
                            while (qh <= 255 && (double)o + (double)qh * step < mx) ++qh;
                            if (qh > 255) { fits = false; break; }
                            qlo = (uint32_t)ql;
                            qhi = (uint32_t)qh;
                        }
                        wlo[i >> 2] |= qlo << (8 * (i & 3));
                        whi[i >> 2] |= qhi << (8 * (i & 3));
                    }
                    if (fits) {
                        if (e + 127 < 1 || e + 127 > 254) { bad = true; return; }
                        wn.o[a] = o;
                        wn.expImask |= (uint32_t)(e + 127) << (8 * a);
                        wn.lo[a][0] = wlo[0]; wn.lo[a][1] = wlo[1];
                        wn.hi[a][0] = whi[0]; wn.hi[a][1] = whi[1];
                        break;
                    }
                }
            }
        }
    };
    {
        const size_t total = w8.size();
        const int workers = total >= 32768 ? buildThreads() : 1;
        ThreadGroup pool((size_t)workers);
        for (int t = 1; t < workers; ++t) {
            const size_t first = total * (size_t)t / (size_t)workers, last = total * (size_t)(t + 1) / (size_t)workers;
            if (!pool.spawn(quantise, first, last)) quantise(first, last);
        }
        quantise(0, total / (size_t)workers);
        pool.join();
    }
    if (bad) return RT_ERR_SCENE;
    if (w8tris.size() >= (1u << 29) || w8.size() >= (1u << 30)) return RT_ERR_SCENE;
    commitLap("child boxes to 8 bits", mark);
    return RT_OK;
}

}  // namespace rtc
