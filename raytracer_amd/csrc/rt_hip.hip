// rt_hip.hip — wavefront path tracer for gfx950 (MI355X) behind the C ABI of
// include/rt_hip.h.
//
// Pipeline per batch of camera samples (all kernels on one stream, queue
// lengths stay on the device, no host round trip inside a batch):
//
//   k_generate         camera rays + path state                  (1 thread / path)
//   for bounce b = 0..bounces
//     k_trace          closest hit for {extension rays of b} U {shadow rays of b-1}
//     k_connect        NEE contribution of the shadow rays of b-1
//     k_shade_*        material, emission/MIS, lobe sampling, NEE ray, Russian roulette;
//                      survivors are compacted into the next queue (wave64 ballot +
//                      one atomic per wave)
//   k_trace/k_connect  shadow rays of the last bounce
//   k_resolve          per pixel, samples added in sample order -> sum buffer
//
// k_trace is a persistent kernel: each wave pulls 64 rays at a time from the
// queue, walks the per-model BVH2 with 64-byte node loads and keeps its
// traversal stack in LDS ([entry][lane], conflict free).  Traversal order,
// strictness of every comparison and the push-time-only culling follow
// hitModel (reference src/RayScene.cpp:681-736), so the winner of a distance
// tie is the one the reference would report.
//
// This file is the host side (context, scene commit and re-pack, launches, RCCL); the kernels are in
// rt_kernels_frame.h, rt_kernels_trace.h and rt_kernels_shade.h, their shared state in
// rt_wavefront.h, scene decode and BSDFs in rt_device.h / rt_bsdf.h, pinned math in rt_math.h.
//
// Build with -ffp-contract=off (see rt_math.h).

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "rt_hip.h"
#include "rt_bsdf.h"

using namespace rtd;

#include "rt_wavefront.h"
#include "rt_kernels_frame.h"
#include "rt_kernels_trace.h"
#include "rt_kernels_trace8.h"
#include "rt_kernels_shade.h"

// ============================================================================================
// Host side
// ============================================================================================

// Staging area of rt_display_rgba8_noalloc: the RGBA8 frame between the display kernel and the copy
// into the caller's buffer.  One copy per GPU, held until the process ends; host code never uses
// the array's name as an address (it is a host-side stand-in there) but the per-device address
// obtained with hipGetSymbolAddress in rt_create.
constexpr size_t kDisplayScratchBytes = 32u << 20;   // 8 Mi pixels: frames up to 3840x2160
static __device__ __align__(256) unsigned char scratch[kDisplayScratchBytes];

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string hipError;
    uchar4 *displayScratch = nullptr;   // this device's copy of `scratch` (hipGetSymbolAddress)

    std::vector<float> host[6];     // uploaded words per slot (1..5)
    float *dStream[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t dStreamWords[6] = {0, 0, 0, 0, 0, 0};
    bool dirty = true;

    int objectCount = 0, modelsCount = 0, lightSourcesCount = 0;
    bool committed = false;
    PackedNode *dNodes = nullptr;
    WideNode *dWNodes = nullptr;    // four-wide tree (wideBvh only)
    Wide8Node *dW8Nodes = nullptr;  // eight-wide tree (wideBvh 4)
    TriPair *dWPairs = nullptr;     // leaves of the four-wide trees as triangle pairs (scenes without alpha-cut textures)
    int wideBvh = 3;                // asked for: 0 off, 1 derived from the uploaded trees, 2 the library's own trees, 3 = 2 + ties re-traced in reference order, 4 = 3 on eight-wide trees
    bool wideChosen = false;        // by option or environment: then a scene the four-wide build declines is an error
    int wideNow = 0;                // in effect for the committed scene (0 when the default was declined for it)
    PackedTri *dWTris = nullptr;    // triangles in wide-tree leaf order
    int wideStackNeed = 0;          // most stack entries a ray can hold in the wide tree
    uint32_t overflowWords = 0;     // 4-byte words per lane allocated in pb.overflow
    PackedTri *dTris = nullptr;
    PackedShade *dShade = nullptr;
    PackedUV *dUVs = nullptr;
    LightTri *dLightTris = nullptr;
    int32_t *dPrimAt = nullptr;     // word offset of every analytic primitive's record in the objects stream
    std::vector<int32_t> hPrimAt;
    ModelDesc *dModelDesc = nullptr;
    std::vector<PackedTri> hTris;   // kept for id decoding in rt_trace_primary
    SceneView view{};

    int W = 0, H = 0;
    float4 *dAccum = nullptr;       // owned unless external
    bool accumExternal = false;
    float4 *dPartial = nullptr;
    uint32_t framesAccum = 0;

    FrameParams fp{};
    int type = RT_TYPE_RGB;
    bool frameSet = false;

    uint32_t maxPaths = kDefaultMaxPaths;
    uint32_t P = 0;                 // allocated path capacity
    PathBuffers pb{};
    bool countersOn = false;
    int numCUs = 256;
    int maxTreeDepth = 0;
    bool hasAlphaTest = false;
    bool shadowEarlyOut = true;
    bool spectralPdfAsWritten = false;
    bool heroWavelength = false;
    bool nee = true;
    bool stratified = true;
    uint32_t tileOffset = 0, tileStride = 1;
    bool planesInner = true;   // tile-major path order inside a batch
    int refillLanes = 32;   // idle lanes that trigger a refill for stages > 0
    bool refillChosen = false;   // set through the option or the environment
    int innerShare = 48;    // inner phase runs while >= innerShare/64 of the active lanes sit on inner nodes

    // RCCL communicator of rt_comm_init (library loaded on first use), one rank per context.
    void *comm = nullptr;
    int commRank = 0, commWorld = 1;
    unsigned long long *dFrameCount = nullptr;

    // Frame pipeline.  A render call of few paths (one progressive frame of a viewer) cannot fill the
    // chip: every kernel of its chain lasts as long as one ray batch's lifetime.  Such calls are queued
    // on kPipes streams in turn, each with path buffers of its own, so that consecutive calls overlap.
    // Everything that touches the sum buffer is chained through `accumFence` (the event recorded after
    // the latest such operation, on whichever stream ran it), so the buffer still receives the frames
    // in call order and holds the same bits.  Pipe 0 is (stream, pb, P) above.
    struct Pipe {
        hipStream_t stream = nullptr;
        PathBuffers pb{};
        uint32_t P = 0;
        uint32_t overflowWords = 0;
        hipEvent_t fence = nullptr;
    };
#ifndef RT_PIPES
#define RT_PIPES 3
#endif
    static constexpr int kPipes = RT_PIPES;
    Pipe pipes[kPipes - 1];
    hipEvent_t fence = nullptr;        // pipe 0's
    hipEvent_t accumFence = nullptr;   // null: nothing pending on the sum buffer
    // Asynchronous screen pass (rt_display_rgba8_begin/_end).  The pass and its copy are queued on the stream
    // of the latest render call: behind that frame, and ahead only of the call that comes round to the same
    // stream again (a stream of its own would be one more than the process has hardware queues for).
    struct DisplaySlot {
        uchar4 *staging = nullptr;
        size_t bytes = 0;
        hipEvent_t readDone = nullptr;   // the pass has read the sum buffer
        hipEvent_t copied = nullptr;     // the caller's buffer holds the frame
        bool busy = false;
    };
    static constexpr int kDisplaySlots = 4;
    DisplaySlot displaySlots[kDisplaySlots];
    hipStream_t lastRenderStream = nullptr;   // null: the context's own
    int nextPipe = 0;
    bool pipelineOn = true;            // RT_OPT_PIPELINE / env RT_PIPELINE
    uint64_t pipelinePaths = 1u << 20; // calls of at most this many paths are pipelined (a 1280x720 frame; a call of 4 Mi paths fills the chip
                                       // alone and only loses by rotating through three sets of buffers: 3.6 -> 4.6 ms)

    std::vector<std::pair<hipEvent_t, hipEvent_t>> traceEvents, renderEvents;
    std::vector<hipEvent_t> eventPool;
    double traceMs = 0.0, totalMs = 0.0;
    uint64_t traceLaunches = 0, cameraSamples = 0;
};

namespace {

#define RT_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->hipError = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return RT_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

template <typename T> int devAlloc(rt_ctx *ctx, T **ptr, size_t count) {
    if (*ptr != nullptr) {
        RT_HIP(hipFree(*ptr));
        *ptr = nullptr;
    }
    if (count == 0) count = 1;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(ptr), count * sizeof(T)));
    return RT_OK;
}

hipEvent_t takeEvent(rt_ctx *ctx) {
    if (!ctx->eventPool.empty()) {
        hipEvent_t e = ctx->eventPool.back();
        ctx->eventPool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Folds finished event pairs into the millisecond totals.  Caller has synchronised.
void drainEvents(rt_ctx *ctx) {
    auto fold = [ctx](std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &acc) {
        for (auto &p : v) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) acc += ms;
            ctx->eventPool.push_back(p.first);
            ctx->eventPool.push_back(p.second);
        }
        v.clear();
    };
    fold(ctx->traceEvents, ctx->traceMs);
    fold(ctx->renderEvents, ctx->totalMs);
}

// The stream now queueing work waits (on the device) for the latest operation on the sum buffer.
void accumWait(rt_ctx *ctx) {
    if (ctx->accumFence != nullptr) hipStreamWaitEvent(ctx->stream, ctx->accumFence, 0);
}
// ... and what it has queued since becomes the latest operation.
void accumMark(rt_ctx *ctx) {
    if (ctx->fence == nullptr && hipEventCreateWithFlags(&ctx->fence, hipEventDisableTiming) != hipSuccess) ctx->fence = nullptr;
    if (ctx->fence != nullptr && hipEventRecord(ctx->fence, ctx->stream) == hipSuccess) ctx->accumFence = ctx->fence;
    else hipStreamSynchronize(ctx->stream);   // no event to be had: the host waits instead
}
// Host-side wait for every stream of the context.
hipError_t syncAllStreams(rt_ctx *ctx) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (rt_ctx::Pipe &p : ctx->pipes)
        if (p.stream != nullptr) {
            const hipError_t ep = hipStreamSynchronize(p.stream);
            if (e == hipSuccess) e = ep;
        }
    if (e == hipSuccess) ctx->accumFence = nullptr;
    return e;
}
// Makes pipe k (1..kPipes-1) the context's current stream and path buffers for the scope's lifetime.
struct PipeScope {
    rt_ctx *ctx;
    rt_ctx::Pipe *pipe;
    PipeScope(rt_ctx *c, int k) : ctx(c), pipe(k > 0 ? &c->pipes[k - 1] : nullptr) { swapIn(); }
    ~PipeScope() { swapIn(); }
    void swapIn() {
        if (pipe == nullptr) return;
        std::swap(ctx->stream, pipe->stream);
        std::swap(ctx->pb, pipe->pb);
        std::swap(ctx->P, pipe->P);
        std::swap(ctx->overflowWords, pipe->overflowWords);
        std::swap(ctx->fence, pipe->fence);
    }
};

// Keeps the number of outstanding timing events bounded for callers that render many times
// without ever synchronising: folds the pairs that have already completed and, if too many are
// still pending, waits for the stream.
void boundEvents(rt_ctx *ctx) {
    constexpr size_t kReapAt = 1024, kSyncAt = 8192;
    if (ctx->traceEvents.size() + ctx->renderEvents.size() < kReapAt) return;
    auto reap = [ctx](std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, double &acc) {
        size_t done = 0;
        while (done < v.size() && hipEventQuery(v[done].second) == hipSuccess) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, v[done].first, v[done].second) == hipSuccess) acc += ms;
            ctx->eventPool.push_back(v[done].first);
            ctx->eventPool.push_back(v[done].second);
            ++done;
        }
        v.erase(v.begin(), v.begin() + (ptrdiff_t)done);
    };
    reap(ctx->traceEvents, ctx->traceMs);
    reap(ctx->renderEvents, ctx->totalMs);
    if (ctx->traceEvents.size() + ctx->renderEvents.size() >= kSyncAt && syncAllStreams(ctx) == hipSuccess) drainEvents(ctx);
}

int gridFor(const rt_ctx *ctx, uint64_t items, int blocksPerCU) {
    uint64_t want = (items + kBlock - 1) / kBlock;
    uint64_t cap = (uint64_t)ctx->numCUs * (uint64_t)blocksPerCU;
    want = std::min(std::max<uint64_t>(want, 1), cap);
    // Producer/consumer kernels map block b to sub-queue b % kSubQueues.
    return (int)((want + kSubQueues - 1) / kSubQueues * kSubQueues);
}

// Builds the traversal re-pack from the uploaded model streams and validates
// every offset the kernels will follow.
int repackScene(rt_ctx *ctx, std::vector<PackedNode> &nodes, std::vector<PackedTri> &tris, std::vector<PackedShade> &shade, std::vector<PackedUV> &uvs, std::vector<LightTri> &lightTris, std::vector<ModelDesc> &descs) {
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

        // Triangles in identifier (leaf) order.
        for (int t = 0; t < d.identifiersCount; ++t) {
            int at = d.identBase + t * kIdentWords;
            if (!checkIdent(at)) return RT_ERR_SCENE;
            int idx[3] = {f2i(mdl[at]), f2i(mdl[at + 1]), f2i(mdl[at + 2])};
            int material = f2i(mdl[at + 3]);
            bool hasTextures = mdl[at + 4] != 0.0f;
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
            pt.ident = t;
            if (hasTextures) {
                int baseColor = f2i(mats[(size_t)material * kMaterialWords + 1]);
                float alphaCut = mats[(size_t)material * kMaterialWords + 20];
                if (baseColor != -1 && alphaCut > 0.0f) {
                    pt.model |= kTriAlphaTest;
                    ctx->hasAlphaTest = true;
                }
            }
            tris.push_back(pt);
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
            shade.push_back(ps);
            PackedUV pu{};
            for (int k = 0; k < 2; ++k) {
                pu.uv0[k] = A[6 + k];
                pu.uv1[k] = B[6 + k];
                pu.uv2[k] = C[6 + k];
            }
            uvs.push_back(pu);
        }

        // Inner nodes get consecutive packed indices in original node order.
        std::vector<int32_t> packedIndex((size_t)d.nodesCount, -1);
        const int32_t innerBase = (int32_t)nodes.size();
        int32_t innerCount = 0;
        auto nodeWord = [&](int n, int w) { return mdl[(size_t)d.nodeBase + (size_t)n * kNodeWords + w]; };
        for (int n = 0; n < d.nodesCount; ++n)
            if (nodeWord(n, 8) == 0.0f) packedIndex[(size_t)n] = innerBase + innerCount++;
        auto childRef = [&](int n, int32_t &ref) -> bool {
            if (n < 0 || n >= d.nodesCount) return false;
            if (nodeWord(n, 8) != 0.0f) {
                int first = f2i(nodeWord(n, 6)), last = f2i(nodeWord(n, 7));
                int count = last - first;
                if (first < 0 || count < 0 || count > 3 || last > d.identifiersCount) return false;
                ref = ~(int32_t)((((uint32_t)(d.triBase + first)) << 2) | (uint32_t)count);
            } else {
                ref = packedIndex[(size_t)n];
            }
            return true;
        };
        nodes.resize((size_t)innerBase + (size_t)innerCount);
        // A tree, not a graph: children come after their parent (no cycles) and every node has at
        // most one parent (a node shared by two parents would make a 32-level graph 2^32 nodes
        // large for anything that walks it as a tree).
        std::vector<char> hasParent((size_t)d.nodesCount, 0);
        for (int n = 0; n < d.nodesCount; ++n) {
            if (nodeWord(n, 8) != 0.0f) continue;
            int l = f2i(nodeWord(n, 6)), r = f2i(nodeWord(n, 7));
            if (l <= n || r <= n || l >= d.nodesCount || r >= d.nodesCount || l == r) return RT_ERR_SCENE;
            if (hasParent[(size_t)l] || hasParent[(size_t)r]) return RT_ERR_SCENE;
            hasParent[(size_t)l] = hasParent[(size_t)r] = 1;
            PackedNode pn{};
            if (!childRef(l, pn.left) || !childRef(r, pn.right)) return RT_ERR_SCENE;
            for (int k = 0; k < 3; ++k) {
                pn.lmin[k] = nodeWord(l, k);
                pn.lmax[k] = nodeWord(l, 3 + k);
                pn.rmin[k] = nodeWord(r, k);
                pn.rmax[k] = nodeWord(r, 3 + k);
            }
            nodes[(size_t)packedIndex[(size_t)n]] = pn;
        }
        if (d.nodesCount > 0) {
            if (!childRef(0, d.rootRef)) return RT_ERR_SCENE;
            // Depth bound: the traversal stack holds at most depth entries.
            std::vector<std::pair<int, int>> todo{{0, 1}};
            int maxDepth = 0;
            while (!todo.empty()) {
                auto [n, depth] = todo.back();
                todo.pop_back();
                maxDepth = std::max(maxDepth, depth);
                if (nodeWord(n, 8) != 0.0f) continue;
                todo.push_back({f2i(nodeWord(n, 6)), depth + 1});
                todo.push_back({f2i(nodeWord(n, 7)), depth + 1});
            }
            if (maxDepth > kStackDepth) return RT_ERR_SCENE;
            ctx->maxTreeDepth = std::max(ctx->maxTreeDepth, maxDepth);
        }
        descs[(size_t)m] = d;
    }
    return RT_OK;
}

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

// Binary tree over one model's triangles by binned surface-area splits (16 bins per axis on the
// triangle centres), for RT_OPT_WIDE_BVH 2 and 3.  Nodes are emitted (root first) through `emit(box,
// left, right, leafRef, leaf)` with child numbers relative to the model's first node; leaves hold one
// or two triangles (three where equal centres leave no split).  `wtris` receives the model's packed
// triangles in leaf order.  Boxes are the exact bounds of the vertices; the
// margin the device arithmetic needs is added when they are quantised.
template <typename Emit>
bool sahTree(const std::vector<float> &mdl, const ModelDesc &d, const std::vector<PackedTri> &tris, std::vector<PackedTri> &wtris, Emit emit) {
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

// Binary trees of all models in one array, as the collapse into wide nodes reads them: box, children
// (indices into the array), or the leaf reference.
struct BinNode { float box[6]; int32_t left, right; int32_t leafRef; bool leaf; };
struct BinaryTrees {
    std::vector<BinNode> bin;
    std::vector<std::pair<int32_t, ModelDesc *>> roots;   // root node and model of every tree that is more than one leaf
    double reach = 0.0;                                    // largest coordinate magnitude of any root box
};

// Fills `trees` with every model's binary tree: the library's own binned-SAH tree when `rebuild` (then
// `wtris` receives the triangle records in that tree's leaf order), else the uploaded one.  A model
// whose tree is a single leaf gets that leaf's reference in wideRootRef and no entry in `roots`.
int binaryTreesOfModels(rt_ctx *ctx, std::vector<ModelDesc> &descs, const std::vector<PackedTri> &tris, std::vector<PackedTri> &wtris, bool rebuild, BinaryTrees &trees) {
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
    for (ModelDesc &d : descs) {
        d.wideRootRef = d.rootRef;   // a tree that is one leaf: same reference, wtris being a copy of tris there
        if (d.nodesCount <= 0) continue;
        if (rebuild && d.identifiersCount > 0) {
            // The library's own tree over the model's triangles: binned surface-area splits, leaves
            // of one or two triangles, a copy of the triangle records in its leaf order.
            const int32_t base = (int32_t)bin.size();
            if (!sahTree(mdl, d, tris, wtris, [&](const float *box, int32_t left, int32_t right, int32_t leafRef, bool leaf) {
                    Bin b{};
                    for (int k = 0; k < 6; ++k) b.box[k] = box[k];
                    b.left = leaf ? -1 : base + left;
                    b.right = leaf ? -1 : base + right;
                    b.leafRef = leafRef;
                    b.leaf = leaf;
                    bin.push_back(b);
                })) return RT_ERR_SCENE;
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
int buildWideNodes(rt_ctx *ctx, std::vector<ModelDesc> &descs, std::vector<WideNode> &wnodes, const std::vector<PackedTri> &tris, std::vector<PackedTri> &wtris) {
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
        while (nk < 4) {
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
void pairLeaves(std::vector<WideNode> &wnodes, std::vector<ModelDesc> &descs, const std::vector<PackedTri> &wtris, std::vector<TriPair> &wpairs) {
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
int buildWide8(rt_ctx *ctx, std::vector<ModelDesc> &descs, std::vector<Wide8Node> &w8, const std::vector<PackedTri> &tris, std::vector<PackedTri> &w8tris) {
    ctx->wideStackNeed = 0;
    BinaryTrees trees;
    std::vector<PackedTri> leafTris;   // in the binary trees' leaf order
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

// (Re)allocates every per-path buffer for `want` paths.  ctx->P is zero while the set is incomplete,
// so a failed attempt can never leave a capacity that some buffer does not have.
int allocPathBuffers(rt_ctx *ctx, uint64_t want64) {
    if (want64 > 0x7fffffffull) return RT_ERR_INVALID;
    const uint32_t want = (uint32_t)want64;
    ctx->P = 0;
    int rc;
    if (ctx->pb.cameraHits != nullptr) {   // sized with the set; ensurePathBuffers makes it again when it is wanted
        RT_HIP(hipFree(ctx->pb.cameraHits));
        ctx->pb.cameraHits = nullptr;
    }
    if ((rc = devAlloc(ctx, &ctx->pb.rays, (size_t)want * 2)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.hits, (size_t)want * 2)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.stA, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.stB, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeA, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeB, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeC, want)) != RT_OK) return rc;
    // Re-trace list of the four-wide walk: room for every ray one launch can trace.
    ctx->pb.retraceCap = 0;
    if ((rc = devAlloc(ctx, &ctx->pb.retrace, (size_t)want * 2)) != RT_OK) return rc;
    ctx->pb.retraceCap = want * 2;
    // A sub-queue receives what the blocks mapped to it produce: at most its share
    // of the paths plus one block-round of slack per block feeding it.
    const size_t maxGrid = (size_t)ctx->numCUs * 16 + kSubQueues;
    const size_t subCap = ((size_t)want + kSubQueues - 1) / kSubQueues + (maxGrid / kSubQueues + 1) * kBlock;
    if (subCap > 0xffffffffull) return RT_ERR_INVALID;
    ctx->pb.subCap = (uint32_t)subCap;
    if ((rc = devAlloc(ctx, &ctx->pb.qExt[0], subCap * kSubQueues)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.qExt[1], subCap * kSubQueues)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.qShadow, subCap * kSubQueues)) != RT_OK) return rc;
    if (ctx->pb.stages == nullptr) {
        if ((rc = devAlloc(ctx, &ctx->pb.stages, (size_t)kMaxBounces + 2)) != RT_OK) return rc;
    }
    if (ctx->pb.counters == nullptr) {
        if ((rc = devAlloc(ctx, &ctx->pb.counters, kCounters)) != RT_OK) return rc;
        RT_HIP(hipMemsetAsync(ctx->pb.counters, 0, kCounters * sizeof(unsigned long long), ctx->stream));
    }
    ctx->P = want;
    return RT_OK;
}

// Makes room for min(wantedPaths, ceiling) paths, whole sample planes, at least one.  Grows only.
int ensurePathBuffers(rt_ctx *ctx, uint64_t wantedPaths) {
    const uint32_t npt = std::max(ctx->fp.npt, 1u);
    uint64_t ceiling = ctx->maxPaths;
    size_t freeBytes = 0, totalBytes = 0;
    if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess) {
        // Path state may take up to half of what is free now (plus what it already holds).
        const uint64_t perPath = 2 * sizeof(RayRec) + 2 * sizeof(HitRec) + 5 * sizeof(float4) + 3 * sizeof(uint32_t) + 4 + 2 * sizeof(uint32_t);   // the last: re-trace list
        ceiling = std::min<uint64_t>(ceiling, (freeBytes / 2 + (uint64_t)ctx->P * perPath) / perPath);
    }
    uint64_t planes = std::max<uint64_t>(1, std::min<uint64_t>(ceiling / npt, (wantedPaths + npt - 1) / npt));
    if (!(ctx->pb.rays != nullptr && ctx->P >= planes * npt)) {
        // If the device cannot give that much after all, retry with half the planes.
        for (;; planes = (planes + 1) / 2) {
            const int rc = allocPathBuffers(ctx, planes * npt);
            if (rc == RT_OK) break;
            if (rc != RT_ERR_HIP || planes == 1) return rc;
            (void)hipGetLastError();
        }
    }
    // Hero-wavelength mode keeps the camera rays' hit records of a sample's first wavelength for the other three.
    // Without room for them every wavelength traces its own camera rays.
    if (ctx->fp.hero && ctx->pb.cameraHits == nullptr && devAlloc(ctx, &ctx->pb.cameraHits, ctx->P) != RT_OK) {
        ctx->pb.cameraHits = nullptr;
        (void)hipGetLastError();
    }
    return RT_OK;
}

void launchTrace(rt_ctx *ctx, int extParity, int stage, uint64_t maxRays) {
    hipEvent_t e0 = takeEvent(ctx), e1 = takeEvent(ctx);
    if (e0) hipEventRecord(e0, ctx->stream);
    // Camera rays (stage 0) stay in 64-ray packets; later stages refill idle lanes.
    // The library-built wide trees make rays short; they run best with a slightly later refill
    // (40 idle lanes: +1.5 %) unless the caller chose a value.
    const int refill = stage == 0 ? 64 : (!ctx->refillChosen && ctx->wideNow >= 2) ? 40 : ctx->refillLanes;
    // Kernel variant: visit counting, LDS stack depth (24 entries let 6 blocks share a
    // CU's LDS instead of 5), alpha-cut support only when some triangle needs it.
    using TraceFn = void (*)(SceneView, PathBuffers, int, int, int, int);
#define RT_TRACE_ROW(C, S, R) {{k_trace<C, S, false, false, R, false>, k_trace<C, S, false, true, R, false>}, {k_trace<C, S, true, false, R, false>, k_trace<C, S, true, true, R, false>}}
    static const TraceFn variants[2][3][2][2] = {{RT_TRACE_ROW(false, 20, false), RT_TRACE_ROW(false, 24, false), RT_TRACE_ROW(false, kStackDepth, false)},
                                                 {RT_TRACE_ROW(true, 20, false), RT_TRACE_ROW(true, 24, false), RT_TRACE_ROW(true, kStackDepth, false)}};
    static const TraceFn rootVariants[2][2][2] = {RT_TRACE_ROW(false, 20, true), RT_TRACE_ROW(false, 24, true)};
#undef RT_TRACE_ROW
    // The second walk of tied rays: reference order over the uploaded trees, reading the re-trace list.
#define RT_RETRACE_ROW(S) {{k_trace<false, S, false, false, false, false, true>, k_trace<false, S, false, true, false, false, true>}, {k_trace<false, S, true, false, false, false, true>, k_trace<false, S, true, true, false, false, true>}}
    static const TraceFn retraceVariants[3][2][2] = {RT_RETRACE_ROW(20), RT_RETRACE_ROW(24), RT_RETRACE_ROW(kStackDepth)};
#undef RT_RETRACE_ROW
#define RT_WIDE_ROW(C, R) {{k_trace<C, kWideStack, false, false, R, true>, k_trace<C, kWideStack, false, true, R, true>}, {k_trace<C, kWideStack, true, false, R, true>, k_trace<C, kWideStack, true, true, R, true>}}
    static const TraceFn wideVariants[2][2][2][2] = {{RT_WIDE_ROW(false, false), RT_WIDE_ROW(false, true)}, {RT_WIDE_ROW(true, false), RT_WIDE_ROW(true, true)}};
#undef RT_WIDE_ROW
    const int depthClass = ctx->maxTreeDepth <= 20 ? 0 : ctx->maxTreeDepth <= 24 ? 1 : 2;
    const bool prims = ctx->objectCount > 0;
    TraceFn fn;
    int wavesPerSimd;
    if (ctx->wideNow == 4) {
        static const TraceFn wide8Variants[2][2][2] = {{{k_trace8<false, false, false>, k_trace8<false, false, true>}, {k_trace8<false, true, false>, k_trace8<false, true, true>}},
                                                       {{k_trace8<true, false, false>, k_trace8<true, false, true>}, {k_trace8<true, true, false>, k_trace8<true, true, true>}}};
        fn = wide8Variants[ctx->countersOn ? 1 : 0][prims ? 1 : 0][ctx->modelsCount > 2 ? 1 : 0];
        wavesPerSimd = RT_W8;
    } else if (ctx->wideNow) {
        fn = wideVariants[ctx->countersOn ? 1 : 0][ctx->modelsCount > 2 ? 1 : 0][ctx->hasAlphaTest ? 1 : 0][prims ? 1 : 0];
        wavesPerSimd = (ctx->hasAlphaTest && prims) ? RT_WWIDE_ALPHA : RT_WWIDE;
    } else {
        const bool rootsInLds = ctx->modelsCount > 2 && depthClass < 2 && !ctx->countersOn;
        fn = rootsInLds ? rootVariants[depthClass][ctx->hasAlphaTest ? 1 : 0][prims ? 1 : 0]
                        : variants[ctx->countersOn ? 1 : 0][depthClass][ctx->hasAlphaTest ? 1 : 0][prims ? 1 : 0];
        wavesPerSimd = depthClass == 0 ? (ctx->hasAlphaTest ? RT_WALPHA : prims ? 7 : RT_W20) : depthClass == 1 ? (ctx->hasAlphaTest ? std::min(RT_WALPHA, 6) : 6) : 5;
    }
    // Blocks of four waves; the full grid is what the CUs can hold.  A launch that cannot have more
    // than a few rays per lane (one progressive frame of a small image) runs on fewer waves: each wave
    // then refills its lanes a few times instead of every wave walking one under-filled batch to its
    // slowest ray's end, which is vector work for idle lanes (+14 % frames/s at 960x540; batches of
    // more than ~2.8 M rays are not affected).
    if (ctx->countersOn && kCountingWaves > 0) wavesPerSimd = std::min(wavesPerSimd, kCountingWaves);
    const int fullGrid = ctx->numCUs * wavesPerSimd;
    constexpr int handsPerWave = 6;   // 64-ray hands a wave should get to deal with (swept 2..12: 3 -> -4 %, 12 -> -8 %)
    const uint64_t fewRaysGrid = std::max<uint64_t>(maxRays / (64 * handsPerWave * (kBlock / 64)), (uint64_t)ctx->numCUs);
    const int grid = (int)std::min<uint64_t>((uint64_t)fullGrid, fewRaysGrid);
    const bool exactTies = ctx->wideNow >= 3;
    PathBuffers pb = ctx->pb;
    if (!exactTies) pb.retrace = nullptr;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->view, pb, extParity, stage, refill, ctx->innerShare);
    if (exactTies) {
        // Rays that met a same-model tie go through the reference-order kernel once more (usually none:
        // a small grid, every wave of which finds the list empty and leaves).
        const TraceFn exact = retraceVariants[depthClass][ctx->hasAlphaTest ? 1 : 0][prims ? 1 : 0];
        hipLaunchKernelGGL(exact, dim3(ctx->numCUs), dim3(kBlock), 0, ctx->stream, ctx->view, pb, 0, stage, 32, ctx->innerShare);
    }
    if (e1) hipEventRecord(e1, ctx->stream);
    if (e0 && e1) ctx->traceEvents.push_back({e0, e1});
    ctx->traceLaunches++;
}

// Queues every kernel of one batch.
int runBatch(rt_ctx *ctx, const Batch &bt) {
    const FrameParams &fp = ctx->fp;
    const bool spectral = ctx->type == RT_TYPE_SPECTRAL;
    const int cnt = ctx->countersOn ? 1 : 0;
    const uint64_t paths = (uint64_t)bt.K * fp.npt;
    const int gGen = gridFor(ctx, paths, 16), gShade = gridFor(ctx, paths, 8);
    // Hero-wavelength mode runs the whole bounce loop once per wavelength of the camera sample.
    const int passes = (spectral && fp.hero) ? 4 : 1;

    for (int pass = 0; pass < passes; ++pass) {
        RT_HIP(hipMemsetAsync(ctx->pb.stages, 0, sizeof(Stage) * ((size_t)fp.bounces + 2), ctx->stream));
        if (spectral) hipLaunchKernelGGL(k_generate<true>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, fp.hero ? pass : -1);
        else hipLaunchKernelGGL(k_generate<false>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, -1);

        for (int b = 0; b <= fp.bounces; ++b) {
            // The four wavelengths of a camera sample start along the same camera ray (same seed, src/RayTracer.cpp:1051,
            // 1357-1364): the camera rays are traced for the first and their hit records kept for the other three.
            const bool sameCameraRays = passes > 1 && b == 0 && ctx->pb.cameraHits != nullptr;
            if (sameCameraRays && pass > 0) {
                RT_HIP(hipMemcpyAsync(ctx->pb.hits, ctx->pb.cameraHits, (size_t)paths * sizeof(HitRec), hipMemcpyDeviceToDevice, ctx->stream));
            } else {
                launchTrace(ctx, b & 1, b, b == 0 ? paths : 2 * paths);
                if (sameCameraRays) RT_HIP(hipMemcpyAsync(ctx->pb.cameraHits, ctx->pb.hits, (size_t)paths * sizeof(HitRec), hipMemcpyDeviceToDevice, ctx->stream));
            }
            if (b > 0 && fp.nee) {   // without NEE there are no shadow rays to connect
                if (spectral) hipLaunchKernelGGL(k_connect<true>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, b, cnt, fp.shadowEarlyOut);
                else hipLaunchKernelGGL(k_connect<false>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, b, cnt, fp.shadowEarlyOut);
            }
            if (spectral) hipLaunchKernelGGL(k_shade_spectral, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, fp, bt, ctx->pb, b, cnt);
            else hipLaunchKernelGGL(k_shade_rgb, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, fp, bt, ctx->pb, b, cnt);
        }
        const int last = fp.bounces + 1;
        if (fp.nee) {
            launchTrace(ctx, -1, last, paths);
            if (spectral) hipLaunchKernelGGL(k_connect<true>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, last, cnt, fp.shadowEarlyOut);
            else hipLaunchKernelGGL(k_connect<false>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, last, cnt, fp.shadowEarlyOut);
        }
    }

    const int gRes = gridFor(ctx, fp.npt, 16);
    accumWait(ctx);   // frames reach the sum buffer in call order, whichever stream traced them
    if (spectral) hipLaunchKernelGGL(k_resolve<true>, dim3(gRes), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, ctx->dAccum, ctx->dPartial);
    else hipLaunchKernelGGL(k_resolve<false>, dim3(gRes), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, ctx->dAccum, ctx->dPartial);
    accumMark(ctx);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

// RCCL is bound at run time so that the library has no link dependency on it: only a caller
// that shards over several GPUs from C/C++ (no torch.distributed around) ever loads it.
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*allReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*errorString)(ncclResult_t) = nullptr;
};

Rccl *rccl() {
    static Rccl api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (lib == nullptr) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (lib != nullptr) {
            api.getUniqueId = reinterpret_cast<decltype(api.getUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
            api.commInitRank = reinterpret_cast<decltype(api.commInitRank)>(dlsym(lib, "ncclCommInitRank"));
            api.commDestroy = reinterpret_cast<decltype(api.commDestroy)>(dlsym(lib, "ncclCommDestroy"));
            api.reduce = reinterpret_cast<decltype(api.reduce)>(dlsym(lib, "ncclReduce"));
            api.allReduce = reinterpret_cast<decltype(api.allReduce)>(dlsym(lib, "ncclAllReduce"));
            api.errorString = reinterpret_cast<decltype(api.errorString)>(dlsym(lib, "ncclGetErrorString"));
            if (api.getUniqueId && api.commInitRank && api.commDestroy && api.reduce && api.allReduce && api.errorString) api.lib = lib;
        }
    }
    return api.lib != nullptr ? &api : nullptr;
}

#define RT_NCCL(ctx, api, call)                                                               \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            (ctx)->hipError = std::string(#call) + ": " + (api)->errorString(r_);             \
            return RT_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

void closeComm(rt_ctx *ctx) {
    if (ctx->comm != nullptr) {
        if (Rccl *api = rccl()) api->commDestroy(static_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
    }
    if (ctx->dFrameCount != nullptr) {
        hipFree(ctx->dFrameCount);
        ctx->dFrameCount = nullptr;
    }
    ctx->commRank = 0;
    ctx->commWorld = 1;
}

// Host allocations of the calls that copy or re-pack whole scenes must not leave the C boundary as exceptions.
template <typename Body>
int noThrow(Body body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return RT_ERR_NOMEM;
    } catch (...) {
        return RT_ERR_INVALID;
    }
}

}  // namespace

extern "C" {

int rt_comm_unique_id(void *id, size_t bytes) {
    if (id == nullptr || bytes != sizeof(ncclUniqueId)) return RT_ERR_INVALID;
    Rccl *api = rccl();
    if (api == nullptr) return RT_ERR_STATE;
    ncclUniqueId uid;
    if (api->getUniqueId(&uid) != ncclSuccess) return RT_ERR_HIP;
    std::memcpy(id, &uid, sizeof uid);
    return RT_OK;
}

int rt_comm_init(rt_ctx *ctx, int rank, int world, const void *id, size_t bytes) {
    if (ctx == nullptr || id == nullptr || bytes != sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) return RT_ERR_INVALID;
    Rccl *api = rccl();
    if (api == nullptr) {
        ctx->hipError = "rt_comm_init: librccl.so.1 could not be loaded";
        return RT_ERR_STATE;
    }
    RT_HIP(hipSetDevice(ctx->device));
    closeComm(ctx);
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    RT_NCCL(ctx, api, api->commInitRank(&comm, world, uid, rank));
    ctx->comm = comm;
    ctx->commRank = rank;
    ctx->commWorld = world;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->dFrameCount), sizeof(unsigned long long)));
    return RT_OK;
}

int rt_reduce(rt_ctx *ctx, int root) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    if (ctx->comm == nullptr || ctx->dAccum == nullptr) return RT_ERR_STATE;
    if (root < 0 || root >= ctx->commWorld) return RT_ERR_INVALID;
    Rccl *api = rccl();
    if (api == nullptr) return RT_ERR_STATE;
    RT_HIP(hipSetDevice(ctx->device));
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    const size_t floats = (size_t)ctx->W * ctx->H * 4;
    unsigned long long frames = ctx->framesAccum;
    accumWait(ctx);
    RT_HIP(hipMemcpyAsync(ctx->dFrameCount, &frames, sizeof frames, hipMemcpyHostToDevice, ctx->stream));
    // Both collectives are queued on the context's stream, behind the render work already there.
    RT_NCCL(ctx, api, api->reduce(ctx->dAccum, ctx->dAccum, floats, ncclFloat, ncclSum, root, comm, ctx->stream));
    RT_NCCL(ctx, api, api->allReduce(ctx->dFrameCount, ctx->dFrameCount, 1, ncclUint64, ncclSum, comm, ctx->stream));
    RT_HIP(hipMemcpyAsync(&frames, ctx->dFrameCount, sizeof frames, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->commRank != root) RT_HIP(hipMemsetAsync(ctx->dAccum, 0, floats * sizeof(float), ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    drainEvents(ctx);
    // The root now holds every rank's frames; the others restart from zero, so that the
    // next merge adds only what was rendered since this one.
    ctx->framesAccum = ctx->commRank == root ? (uint32_t)frames : 0u;
    return RT_OK;
}

int rt_create(rt_ctx **out, int device) {
    if (out == nullptr) return RT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RT_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return RT_ERR_INVALID;
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (ctx == nullptr) return RT_ERR_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return RT_ERR_HIP;
    }
    // Every GPU holds its own copy of the file-scope staging array at its own address.
    void *scratchPtr = nullptr;
    if (hipGetSymbolAddress(&scratchPtr, HIP_SYMBOL(scratch)) != hipSuccess || scratchPtr == nullptr) {
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return RT_ERR_HIP;
    }
    ctx->displayScratch = static_cast<uchar4 *>(scratchPtr);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->numCUs = prop.multiProcessorCount;
    if (const char *env = std::getenv("RT_MAX_PATHS")) {
        long v = std::atol(env);
        if (v > 0) ctx->maxPaths = (uint32_t)std::min<long>(v, 1l << 30);
    }
    if (const char *env = std::getenv("RT_TILE_MAJOR")) ctx->planesInner = std::atoi(env) != 0;
    if (const char *env = std::getenv("RT_PIPELINE")) ctx->pipelineOn = std::atoi(env) != 0;
    if (const char *env = std::getenv("RT_WIDE_BVH")) {
        ctx->wideBvh = std::min(std::max(std::atoi(env), 0), 4);
        ctx->wideChosen = true;
    }
    if (const char *env = std::getenv("RT_INNER_SHARE")) {
        int v = std::atoi(env);
        if (v >= 1 && v <= 64) ctx->innerShare = v;
    }
    if (const char *env = std::getenv("RT_REFILL_LANES")) {
        int v = std::atoi(env);
        if (v >= 1 && v <= 64) { ctx->refillLanes = v; ctx->refillChosen = true; }
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx *ctx) {
    if (ctx == nullptr) return;
    hipSetDevice(ctx->device);
    syncAllStreams(ctx);
    for (rt_ctx::Pipe &pipe : ctx->pipes) {
        void *mine[] = {pipe.pb.rays, pipe.pb.hits, pipe.pb.stA, pipe.pb.stB, pipe.pb.neeA, pipe.pb.neeB, pipe.pb.neeC, pipe.pb.qExt[0], pipe.pb.qExt[1],
                        pipe.pb.qShadow, pipe.pb.stages, pipe.pb.retrace, pipe.pb.overflow, pipe.pb.cameraHits};   // the counters belong to pipe 0
        for (void *p : mine) if (p) hipFree(p);
        if (pipe.stream) hipStreamDestroy(pipe.stream);
        if (pipe.fence) hipEventDestroy(pipe.fence);
    }
    if (ctx->fence) hipEventDestroy(ctx->fence);
    for (rt_ctx::DisplaySlot &d : ctx->displaySlots) {
        if (d.staging) hipFree(d.staging);
        if (d.readDone) hipEventDestroy(d.readDone);
        if (d.copied) hipEventDestroy(d.copied);
    }
    for (float *p : ctx->dStream) if (p) hipFree(p);
    void *owned[] = {ctx->dNodes, ctx->dWNodes, ctx->dW8Nodes, ctx->dWPairs, ctx->pb.overflow, ctx->pb.retrace, ctx->dWTris, ctx->dTris, ctx->dShade, ctx->dUVs, ctx->dLightTris, ctx->dPrimAt, ctx->dModelDesc, ctx->accumExternal ? nullptr : (void *)ctx->dAccum, ctx->dPartial,
                     ctx->pb.rays, ctx->pb.hits, ctx->pb.stA, ctx->pb.stB, ctx->pb.neeA, ctx->pb.neeB, ctx->pb.neeC,
                     ctx->pb.qExt[0], ctx->pb.qExt[1], ctx->pb.qShadow, ctx->pb.stages, ctx->pb.counters, ctx->pb.cameraHits};
    for (void *p : owned) if (p) hipFree(p);
    drainEvents(ctx);
    closeComm(ctx);
    for (hipEvent_t e : ctx->eventPool) hipEventDestroy(e);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rt_upload(rt_ctx *ctx, int slot, const void *data, size_t bytes) {
    if (ctx == nullptr || slot < 1 || slot > 5 || (data == nullptr && bytes != 0) || (bytes % 4) != 0) return RT_ERR_INVALID;
    const float *w = static_cast<const float *>(data);
    const int rc = noThrow([&]() -> int { ctx->host[slot].assign(w, w + bytes / 4); return RT_OK; });
    if (rc != RT_OK) return rc;
    ctx->dirty = true;
    ctx->committed = false;
    return RT_OK;
}

static int commitScene(rt_ctx *ctx, int objectCount, int modelsCount, int lightSourcesCount);
int rt_commit_scene(rt_ctx *ctx, int objectCount, int modelsCount, int lightSourcesCount) {
    return noThrow([&]() -> int { return commitScene(ctx, objectCount, modelsCount, lightSourcesCount); });
}

static int commitScene(rt_ctx *ctx, int objectCount, int modelsCount, int lightSourcesCount) {
    if (ctx == nullptr || objectCount < 0 || modelsCount < 0 || lightSourcesCount < 0) return RT_ERR_INVALID;
    if (ctx->committed && !ctx->dirty && objectCount == ctx->objectCount && modelsCount == ctx->modelsCount &&
        lightSourcesCount == ctx->lightSourcesCount)
        return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(syncAllStreams(ctx));
    ctx->committed = false;   // a failed commit leaves the context without a scene, never with a stale one
    ctx->objectCount = objectCount;
    ctx->modelsCount = modelsCount;
    ctx->lightSourcesCount = lightSourcesCount;

    std::vector<PackedNode> nodes;
    std::vector<PackedTri> tris;
    std::vector<PackedShade> shade;
    std::vector<PackedUV> uvs;
    std::vector<LightTri> lightTris;
    std::vector<ModelDesc> descs;
    // RT_COMMIT_TIMING=1 prints where a commit spends its host time.
    const bool timing = std::getenv("RT_COMMIT_TIMING") != nullptr;
    auto clockNow = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](const char *what, std::chrono::steady_clock::time_point &since) {
        const auto now = clockNow();
        if (timing) std::fprintf(stderr, "rt_commit_scene: %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - since).count());
        since = now;
    };
    auto mark = clockNow();
    int rc = repackScene(ctx, nodes, tris, shade, uvs, lightTris, descs);
    if (rc != RT_OK) return rc;
    lap("validate + re-pack", mark);

    // Raw streams, each padded with 8 zero words so the one-past-the-end reads the
    // reference's texel fetches can make (4th channel of the last RGB texel) stay in bounds.
    for (int s = 1; s <= 5; ++s) {
        std::vector<float> padded(ctx->host[s]);
        padded.resize(padded.size() + 8, 0.0f);
        if ((rc = devAlloc(ctx, &ctx->dStream[s], padded.size())) != RT_OK) return rc;
        RT_HIP(hipMemcpy(ctx->dStream[s], padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
        ctx->dStreamWords[s] = ctx->host[s].size();
    }
    lap("stream upload", mark);
    std::vector<WideNode> wnodes;
    std::vector<Wide8Node> w8nodes;
    std::vector<PackedTri> wtris;
    ctx->wideNow = tris.empty() ? 0 : ctx->wideBvh;   // nothing to build over analytic primitives alone
    // The eight-wide kernel has no alpha-cut phase: scenes with alpha-cut textures walk the four-wide trees.
    if (ctx->wideNow == 4 && ctx->hasAlphaTest) ctx->wideNow = 3;
    if (ctx->wideNow) {
        rc = ctx->wideNow == 4 ? buildWide8(ctx, descs, w8nodes, tris, wtris) : buildWideNodes(ctx, descs, wnodes, tris, wtris);
        // A scene the wide build cannot take (coordinates beyond its box arithmetic, NaN) is
        // walked in the reference's order, as it would be with the option off, unless the caller
        // asked for the wide trees by name.
        if (rc == RT_ERR_SCENE && !ctx->wideChosen) ctx->wideNow = 0;
        else if (rc != RT_OK) return rc;
    }
    std::vector<TriPair> wpairs;
    if (ctx->wideNow >= 1 && ctx->wideNow <= 3 && !ctx->hasAlphaTest) pairLeaves(wnodes, descs, wtris, wpairs);
    lap("wide trees", mark);
    if (ctx->wideNow) {
        if ((rc = devAlloc(ctx, &ctx->dWPairs, wpairs.size())) != RT_OK) return rc;
        if (!wpairs.empty()) RT_HIP(hipMemcpy(ctx->dWPairs, wpairs.data(), wpairs.size() * sizeof(TriPair), hipMemcpyHostToDevice));
        if ((rc = devAlloc(ctx, &ctx->dWTris, wtris.size())) != RT_OK) return rc;
        if (!wtris.empty()) RT_HIP(hipMemcpy(ctx->dWTris, wtris.data(), wtris.size() * sizeof(PackedTri), hipMemcpyHostToDevice));
        if ((rc = devAlloc(ctx, &ctx->dWNodes, wnodes.size())) != RT_OK) return rc;
        if (!wnodes.empty()) RT_HIP(hipMemcpy(ctx->dWNodes, wnodes.data(), wnodes.size() * sizeof(WideNode), hipMemcpyHostToDevice));
        if ((rc = devAlloc(ctx, &ctx->dW8Nodes, w8nodes.size())) != RT_OK) return rc;
        if (!w8nodes.empty()) RT_HIP(hipMemcpy(ctx->dW8Nodes, w8nodes.data(), w8nodes.size() * sizeof(Wide8Node), hipMemcpyHostToDevice));
        // Stack entries that do not fit in LDS: one column per lane of every wave the grid can hold
        // (4-byte entries beyond kWideStack for the four-wide walk, 8-byte entries beyond kWide8Stack for the eight-wide).
        const bool eight = ctx->wideNow == 4;
        // RT_W8_LDS_STACK (1..kWide8Stack) keeps fewer entries on chip: a testing aid for the in-memory part.
        int onChip8 = kWide8Stack;
        if (const char *env = std::getenv("RT_W8_LDS_STACK")) onChip8 = std::min(std::max(std::atoi(env), 1), kWide8Stack);
        ctx->pb.overflowFrom = (uint32_t)onChip8;
        const int deeper = std::max(1, ctx->wideStackNeed - (eight ? onChip8 : kWideStack));
        const uint32_t words = (uint32_t)deeper * (eight ? 2u : 1u);
        if (words > ctx->overflowWords || ctx->pb.overflow == nullptr) {
            const size_t waves = (size_t)ctx->numCUs * std::max(std::max(RT_WWIDE, RT_WWIDE_ALPHA), RT_W8) * (kBlock / 64);
            if ((rc = devAlloc(ctx, &ctx->pb.overflow, waves * (size_t)words * 64)) != RT_OK) return rc;
            ctx->overflowWords = words;
        }
        ctx->pb.overflowDepth = ctx->overflowWords / (eight ? 2u : 1u);   // entries per lane the kernels may index
    }
    if ((rc = devAlloc(ctx, &ctx->dNodes, nodes.size())) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->dTris, tris.size())) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->dShade, shade.size())) != RT_OK) return rc;
    if (!shade.empty()) RT_HIP(hipMemcpy(ctx->dShade, shade.data(), shade.size() * sizeof(PackedShade), hipMemcpyHostToDevice));
    if ((rc = devAlloc(ctx, &ctx->dUVs, uvs.size())) != RT_OK) return rc;
    if (!uvs.empty()) RT_HIP(hipMemcpy(ctx->dUVs, uvs.data(), uvs.size() * sizeof(PackedUV), hipMemcpyHostToDevice));
    if ((rc = devAlloc(ctx, &ctx->dLightTris, lightTris.size())) != RT_OK) return rc;
    if (!lightTris.empty()) RT_HIP(hipMemcpy(ctx->dLightTris, lightTris.data(), lightTris.size() * sizeof(LightTri), hipMemcpyHostToDevice));
    if ((rc = devAlloc(ctx, &ctx->dPrimAt, ctx->hPrimAt.size())) != RT_OK) return rc;
    if (!ctx->hPrimAt.empty()) RT_HIP(hipMemcpy(ctx->dPrimAt, ctx->hPrimAt.data(), ctx->hPrimAt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if ((rc = devAlloc(ctx, &ctx->dModelDesc, descs.size())) != RT_OK) return rc;
    if (!nodes.empty()) RT_HIP(hipMemcpy(ctx->dNodes, nodes.data(), nodes.size() * sizeof(PackedNode), hipMemcpyHostToDevice));
    if (!tris.empty()) RT_HIP(hipMemcpy(ctx->dTris, tris.data(), tris.size() * sizeof(PackedTri), hipMemcpyHostToDevice));
    if (!descs.empty()) RT_HIP(hipMemcpy(ctx->dModelDesc, descs.data(), descs.size() * sizeof(ModelDesc), hipMemcpyHostToDevice));
    ctx->hTris.swap(tris);
    lap("packed upload", mark);

    SceneView &v = ctx->view;
    v.objects = ctx->dStream[RT_SLOT_OBJECTS];
    v.materials = ctx->dStream[RT_SLOT_MATERIALS];
    v.modelInfo = ctx->dStream[RT_SLOT_MODEL_INFO];
    v.models = ctx->dStream[RT_SLOT_MODELS];
    v.textures = ctx->dStream[RT_SLOT_TEXTURES];
    v.nodes = ctx->dNodes;
    v.wnodes = ctx->wideNow ? ctx->dWNodes : nullptr;
    v.w8nodes = ctx->wideNow == 4 ? ctx->dW8Nodes : nullptr;
    v.wpairs = (ctx->wideNow >= 1 && ctx->wideNow <= 3 && !ctx->hasAlphaTest) ? ctx->dWPairs : nullptr;
    v.wtris = ctx->wideNow ? ctx->dWTris : nullptr;
    v.tris = ctx->dTris;
    v.shade = ctx->dShade;
    v.uvs = ctx->dUVs;
    v.lightTris = ctx->dLightTris;
    v.primAt = ctx->dPrimAt;
    v.modelDesc = ctx->dModelDesc;
    v.objectCount = objectCount;
    v.modelsCount = modelsCount;
    v.lightSourcesCount = lightSourcesCount;
    ctx->dirty = false;
    ctx->committed = true;
    return RT_OK;
}

int rt_resize(rt_ctx *ctx, int width, int height) {
    if (ctx == nullptr || width <= 0 || height <= 0 || (uint64_t)width * height > (1u << 28)) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(syncAllStreams(ctx));
    ctx->W = width;
    ctx->H = height;
    const size_t px = (size_t)width * height;
    int rc;
    if (ctx->accumExternal) {
        ctx->dAccum = nullptr;   // an external buffer is tied to the old size
        ctx->accumExternal = false;
    }
    if ((rc = devAlloc(ctx, &ctx->dAccum, px)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->dPartial, px)) != RT_OK) return rc;
    RT_HIP(hipMemsetAsync(ctx->dAccum, 0, px * sizeof(float4), ctx->stream));
    RT_HIP(hipMemsetAsync(ctx->dPartial, 0, px * sizeof(float4), ctx->stream));
    accumMark(ctx);
    ctx->framesAccum = 0;
    ctx->frameSet = false;
    return RT_OK;
}

int rt_set_frame(rt_ctx *ctx, const rt_camera *cam, const float sky_linear[3], int type) {
    if (ctx == nullptr || cam == nullptr || sky_linear == nullptr) return RT_ERR_INVALID;
    if (type != RT_TYPE_RGB && type != RT_TYPE_SPECTRAL) return RT_ERR_INVALID;
    if (ctx->W <= 0) return RT_ERR_STATE;
    if ((int)cam->resolution[0] != ctx->W || (int)cam->resolution[1] != ctx->H) return RT_ERR_INVALID;
    if (cam->bounces < 0 || cam->bounces > kMaxBounces || cam->rayPerPixel < 1) return RT_ERR_INVALID;

    FrameParams &fp = ctx->fp;
    fp.fov = cam->fov;
    fp.resX = cam->resolution[0];
    fp.resY = cam->resolution[1];
    fp.position = v3(cam->position[0], cam->position[1], cam->position[2]);
    fp.forward = v3(cam->forward[0], cam->forward[1], cam->forward[2]);
    fp.right = v3(cam->right[0], cam->right[1], cam->right[2]);
    fp.up = v3(cam->up[0], cam->up[1], cam->up[2]);
    fp.bounces = cam->bounces;
    fp.rayPerPixel = cam->rayPerPixel;
    fp.sky = v3(sky_linear[0], sky_linear[1], sky_linear[2]);

    // Per-launch constants of fragment main() (src/RayTracer.cpp:545-557, 570-571, 1343-1346).
    fp.rImgX = 1.0f / fp.resX;
    fp.rImgY = 1.0f / fp.resY;
    const V3 lookat = fp.forward + fp.position;
    const float viewportRatio = fp.resX * fp.rImgY;
    fp.focalLength = length(lookat - fp.position);
    const float viewportHeight = 2.0f * tan_(kRad * fp.fov * 0.5f) * fp.focalLength;
    const float viewportWidth = viewportHeight * viewportRatio;
    fp.halfWidth = viewportWidth * 0.5f;
    fp.halfHeight = viewportHeight * 0.5f;
    fp.ssq = (int)sqrt_((float)fp.rayPerPixel);
    if (fp.ssq < 1) return RT_ERR_INVALID;
    fp.rssq = 1.0f / (float)fp.ssq;
    fp.wlDt = kWlRange / (float)fp.rayPerPixel;
    fp.wlPdf = 1.0f / kWlRange;
    fp.W = ctx->W;
    fp.H = ctx->H;
    fp.tilesX = (ctx->W + 7) / 8;
    fp.tilesY = (ctx->H + 7) / 8;
    fp.npt = (uint32_t)fp.tilesX * (uint32_t)fp.tilesY * 64u;
    fp.S = (uint32_t)fp.ssq * (uint32_t)fp.ssq;
    fp.stratified = (ctx->stratified || type != RT_TYPE_RGB) ? 1 : 0;
    if (!fp.stratified) fp.S = (uint32_t)fp.rayPerPixel;
    fp.shadowEarlyOut = (ctx->shadowEarlyOut && type == RT_TYPE_RGB) ? 1 : 0;
    fp.spectralPdfAsWritten = ctx->spectralPdfAsWritten ? 1 : 0;
    fp.hero = (ctx->heroWavelength && type == RT_TYPE_SPECTRAL) ? 1 : 0;
    fp.nee = ctx->nee ? 1 : 0;
    fp.tileOffset = ctx->tileOffset;
    fp.tileStride = ctx->tileStride;
    ctx->type = type;
    ctx->frameSet = true;
    return RT_OK;
}

// Stream `pipeNo` of the frame pipeline, created on first use; 0 (the context's own) when it cannot be had.
static int usablePipe(rt_ctx *ctx, int pipeNo) {
    if (pipeNo <= 0) return 0;
    rt_ctx::Pipe &pipe = ctx->pipes[pipeNo - 1];
    if (ctx->pb.counters == nullptr) return 0;   // the shared counters are created with the first pipe's buffers
    if (pipe.stream == nullptr) {
        if (hipStreamCreate(&pipe.stream) != hipSuccess) {
            pipe.stream = nullptr;
            (void)hipGetLastError();
            return 0;
        }
        pipe.pb.counters = ctx->pb.counters;
    }
    return pipeNo;
}

// Queues sample planes [g0, g1) of a render call on stream `pipeNo`.
// Device memory stream `pipeNo` needs for a call of `paths` paths: its path buffers and, beyond pipe 0,
// its own copy of the part of the traversal stack that does not fit in LDS.  Call inside a PipeScope.
static int preparePipe(rt_ctx *ctx, const PipeScope &onPipe, int pipeNo, uint64_t paths) {
    int rc = ensurePathBuffers(ctx, paths);
    if (rc != RT_OK) return rc;
    if (pipeNo > 0 && ctx->wideNow) {
        const rt_ctx::Pipe &first = *onPipe.pipe;   // holds pipe 0's state while the scope lasts
        if (ctx->pb.overflow == nullptr || ctx->overflowWords < first.overflowWords) {
            const size_t waves = (size_t)ctx->numCUs * std::max(std::max(RT_WWIDE, RT_WWIDE_ALPHA), RT_W8) * (kBlock / 64);
            if ((rc = devAlloc(ctx, &ctx->pb.overflow, waves * (size_t)first.overflowWords * 64)) != RT_OK) return rc;
            ctx->overflowWords = first.overflowWords;
        }
        ctx->pb.overflowDepth = first.pb.overflowDepth;
        ctx->pb.overflowFrom = first.pb.overflowFrom;
    }
    return RT_OK;
}

static int renderPlanes(rt_ctx *ctx, int pipeNo, uint64_t g0, uint64_t g1, uint32_t first_frame, uint32_t frame_stride) {
    const FrameParams &fp = ctx->fp;
    PipeScope onPipe(ctx, pipeNo);
    int rc = preparePipe(ctx, onPipe, pipeNo, (g1 - g0) * fp.npt);
    if (rc != RT_OK) return rc;
    boundEvents(ctx);

    hipEvent_t e0 = takeEvent(ctx), e1 = takeEvent(ctx);
    if (e0) hipEventRecord(e0, ctx->stream);
    // Batches never exceed the current ceiling even when a larger buffer is already held.
    const uint32_t planesPerBatch = std::max(1u, std::min(ctx->P, std::max(ctx->maxPaths, fp.npt)) / fp.npt);
    for (uint64_t g = g0; g < g1; g += planesPerBatch) {
        Batch bt;
        bt.g0 = (uint32_t)g;
        bt.K = (uint32_t)std::min<uint64_t>(planesPerBatch, g1 - g);
        bt.firstFrame = first_frame;
        bt.frameStride = frame_stride;
        bt.P = ctx->P;
        bt.planesInner = ctx->planesInner ? 1u : 0u;
        if ((rc = runBatch(ctx, bt)) != RT_OK) return rc;
    }
    if (e1) hipEventRecord(e1, ctx->stream);
    if (e0 && e1) ctx->renderEvents.push_back({e0, e1});
    ctx->lastRenderStream = ctx->stream;   // (the pipe's, while the scope lasts)
    return RT_OK;
}

int rt_render(rt_ctx *ctx, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride) {
    if (ctx == nullptr || frame_stride == 0) return RT_ERR_INVALID;
    if (!ctx->committed || !ctx->frameSet || ctx->dAccum == nullptr) return RT_ERR_STATE;
    if (n_frames == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const FrameParams &fp = ctx->fp;
    const uint64_t totalPlanes = (uint64_t)n_frames * fp.S;
    if (totalPlanes > 0xffffffffull) return RT_ERR_INVALID;
    int rc = RT_OK;
    if (ctx->pipelineOn && totalPlanes * fp.npt <= ctx->pipelinePaths) {
        // A call of few paths takes the next stream of the frame pipeline.
        const int pipeNo = usablePipe(ctx, ctx->nextPipe);
        ctx->nextPipe = (ctx->nextPipe + 1) % rt_ctx::kPipes;
        rc = renderPlanes(ctx, pipeNo, 0, totalPlanes, first_frame, frame_stride);
    } else {
        // Anything larger fills the chip by itself and stays on the context's own stream (its resolve
        // passes still wait their turn).  Dealing its sample planes to two or three streams, so that the
        // drain tail of one share's kernels overlaps with another share's work, changed nothing
        // (1920x1080 x 64 planes: 109.2-109.6 ms on one stream, 110.3 on two, 109.3 on three).
        rc = renderPlanes(ctx, 0, 0, totalPlanes, first_frame, frame_stride);
    }
    if (rc != RT_OK) return rc;
    ctx->framesAccum += n_frames;
    uint64_t ownedPixels = (uint64_t)ctx->W * (uint64_t)ctx->H;
    if (fp.tileStride > 1) {
        ownedPixels = 0;
        for (uint32_t tile = fp.tileOffset; tile < (uint32_t)fp.tilesX * (uint32_t)fp.tilesY; tile += fp.tileStride) {
            const int x0 = (int)(tile % (uint32_t)fp.tilesX) * 8, y0 = (int)(tile / (uint32_t)fp.tilesX) * 8;
            ownedPixels += (uint64_t)std::min(8, fp.W - x0) * (uint64_t)std::min(8, fp.H - y0);
        }
    }
    ctx->cameraSamples += totalPlanes * ownedPixels;
    return RT_OK;
}

int rt_reserve(rt_ctx *ctx, uint32_t n_frames) {
    if (ctx == nullptr || n_frames == 0) return RT_ERR_INVALID;
    if (!ctx->frameSet) return RT_ERR_STATE;
    RT_HIP(hipSetDevice(ctx->device));
    const uint64_t totalPlanes = (uint64_t)n_frames * ctx->fp.S;
    if (totalPlanes > 0xffffffffull) return RT_ERR_INVALID;
    RT_HIP(syncAllStreams(ctx));   // the buffers may be in use by queued work
    const uint64_t paths = totalPlanes * ctx->fp.npt;
    int rc = ensurePathBuffers(ctx, paths);
    // Calls of this size rotate through the streams of the frame pipeline: each gets its buffers now.
    if (rc == RT_OK && ctx->pipelineOn && paths <= ctx->pipelinePaths)
        for (int k = 1; k < rt_ctx::kPipes && rc == RT_OK; ++k) {
            const int pipeNo = usablePipe(ctx, k);
            if (pipeNo == 0) break;
            PipeScope onPipe(ctx, pipeNo);
            rc = preparePipe(ctx, onPipe, pipeNo, paths);
        }
    return rc;
}

int rt_reset_accum(rt_ctx *ctx) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    RT_HIP(hipSetDevice(ctx->device));
    accumWait(ctx);
    RT_HIP(hipMemsetAsync(ctx->dAccum, 0, (size_t)ctx->W * ctx->H * sizeof(float4), ctx->stream));
    accumMark(ctx);
    ctx->framesAccum = 0;
    return RT_OK;
}

int rt_sync(rt_ctx *ctx) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(syncAllStreams(ctx));
    drainEvents(ctx);
    return RT_OK;
}

int rt_readback(rt_ctx *ctx, float *rgba_sum, size_t floats, uint32_t *frames) {
    if (ctx == nullptr || rgba_sum == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    if (floats != (size_t)ctx->W * ctx->H * 4) return RT_ERR_INVALID;
    int rc = rt_sync(ctx);
    if (rc != RT_OK) return rc;
    RT_HIP(hipMemcpy(rgba_sum, ctx->dAccum, floats * sizeof(float), hipMemcpyDeviceToHost));
    if (frames) *frames = ctx->framesAccum;
    return RT_OK;
}

int rt_display_rgba8(rt_ctx *ctx, uint8_t *rgba8, size_t bytes) {
    if (ctx == nullptr || rgba8 == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    const size_t px = (size_t)ctx->W * ctx->H;
    if (bytes != px * 4) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    uchar4 *dOut = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&dOut), px * sizeof(uchar4)));
    accumWait(ctx);   // every frame queued so far, on whichever stream
    hipLaunchKernelGGL(k_display, dim3(gridFor(ctx, px, 8)), dim3(kBlock), 0, ctx->stream, ctx->dAccum, dOut, (uint32_t)px);
    hipError_t e = hipMemcpyAsync(rgba8, dOut, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(dOut);
    if (e != hipSuccess) {
        ctx->hipError = std::string("rt_display_rgba8: ") + hipGetErrorString(e);
        return RT_ERR_HIP;
    }
    return RT_OK;
}

int rt_display_rgba8_noalloc(rt_ctx *ctx, uint8_t *rgba8, size_t bytes) {
    if (ctx == nullptr || rgba8 == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    const size_t px = (size_t)ctx->W * ctx->H;
    if (bytes != px * 4) return RT_ERR_INVALID;
    if (bytes > kDisplayScratchBytes) {
        ctx->hipError = "rt_display_rgba8_noalloc: a frame of " + std::to_string(px) + " pixels does not fit the fixed staging area of " +
                        std::to_string(kDisplayScratchBytes / 4) + " pixels (use rt_display_rgba8)";
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(ctx->device));
    // The kernel writes every pixel of the frame, so what an earlier call left in the staging area does not matter.
    accumWait(ctx);   // every frame queued so far, on whichever stream
    hipLaunchKernelGGL(k_display, dim3(gridFor(ctx, px, 8)), dim3(kBlock), 0, ctx->stream, ctx->dAccum, ctx->displayScratch, (uint32_t)px);
    RT_HIP(hipMemcpyAsync(rgba8, ctx->displayScratch, bytes, hipMemcpyDefault, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_display_rgba8_begin(rt_ctx *ctx, int slot, uint8_t *rgba8, size_t bytes) {
    if (ctx == nullptr || rgba8 == nullptr || slot < 0 || slot >= rt_ctx::kDisplaySlots) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    const size_t px = (size_t)ctx->W * ctx->H;
    if (bytes != px * 4) return RT_ERR_INVALID;
    rt_ctx::DisplaySlot &d = ctx->displaySlots[slot];
    if (d.busy) return RT_ERR_STATE;   // begun and not ended
    RT_HIP(hipSetDevice(ctx->device));
    const hipStream_t stream = ctx->lastRenderStream != nullptr ? ctx->lastRenderStream : ctx->stream;
    if (d.readDone == nullptr) RT_HIP(hipEventCreateWithFlags(&d.readDone, hipEventDisableTiming));
    if (d.copied == nullptr) RT_HIP(hipEventCreateWithFlags(&d.copied, hipEventDisableTiming));
    if (d.bytes != bytes) {   // first use of the slot at this resolution (rt_display_rgba8_end has waited for its last frame)
        d.bytes = 0;
        int rc = devAlloc(ctx, &d.staging, px);
        if (rc != RT_OK) return rc;
        d.bytes = bytes;
    }
    // The pass takes its place behind everything queued so far that touches the sum buffer, and what is
    // queued afterwards waits until the pass has read it.
    if (ctx->accumFence != nullptr) RT_HIP(hipStreamWaitEvent(stream, ctx->accumFence, 0));
    hipLaunchKernelGGL(k_display, dim3(gridFor(ctx, px, 8)), dim3(kBlock), 0, stream, ctx->dAccum, d.staging, (uint32_t)px);
    RT_HIP(hipEventRecord(d.readDone, stream));
    ctx->accumFence = d.readDone;
    RT_HIP(hipMemcpyAsync(rgba8, d.staging, bytes, hipMemcpyDefault, stream));
    RT_HIP(hipEventRecord(d.copied, stream));
    d.busy = true;
    return RT_OK;
}

int rt_display_rgba8_end(rt_ctx *ctx, int slot) {
    if (ctx == nullptr || slot < 0 || slot >= rt_ctx::kDisplaySlots) return RT_ERR_INVALID;
    rt_ctx::DisplaySlot &d = ctx->displaySlots[slot];
    if (!d.busy) return RT_ERR_STATE;
    RT_HIP(hipSetDevice(ctx->device));
    d.busy = false;
    RT_HIP(hipEventSynchronize(d.copied));
    return RT_OK;
}

int rt_restore_accum(rt_ctx *ctx, const float *rgba_sum, size_t floats, uint32_t frames) {
    if (ctx == nullptr || rgba_sum == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    if (floats != (size_t)ctx->W * ctx->H * 4) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(syncAllStreams(ctx));
    RT_HIP(hipMemcpy(ctx->dAccum, rgba_sum, floats * sizeof(float), hipMemcpyHostToDevice));
    ctx->framesAccum = frames;
    return RT_OK;
}

int rt_accum_device_ptr(rt_ctx *ctx, void **dev_ptr, size_t *floats) {
    if (ctx == nullptr || dev_ptr == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    *dev_ptr = ctx->dAccum;
    if (floats) *floats = (size_t)ctx->W * ctx->H * 4;
    return RT_OK;
}

int rt_bind_accum(rt_ctx *ctx, void *dev_ptr, size_t floats) {
    if (ctx == nullptr || dev_ptr == nullptr) return RT_ERR_INVALID;
    if (ctx->W <= 0) return RT_ERR_STATE;
    if (floats != (size_t)ctx->W * ctx->H * 4 || (reinterpret_cast<uintptr_t>(dev_ptr) & 15u) != 0) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(syncAllStreams(ctx));
    if (!ctx->accumExternal && ctx->dAccum != nullptr) RT_HIP(hipFree(ctx->dAccum));
    ctx->dAccum = static_cast<float4 *>(dev_ptr);
    ctx->accumExternal = true;
    ctx->framesAccum = 0;
    return RT_OK;
}

int rt_add_frames(rt_ctx *ctx, uint32_t frames) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    ctx->framesAccum += frames;
    return RT_OK;
}

int rt_trace_primary(rt_ctx *ctx, uint32_t frame, int32_t *model, int32_t *ident, float *t, float *u, float *v) {
    if (ctx == nullptr || model == nullptr || ident == nullptr || t == nullptr || u == nullptr || v == nullptr) return RT_ERR_INVALID;
    if (!ctx->committed || !ctx->frameSet) return RT_ERR_STATE;
    RT_HIP(hipSetDevice(ctx->device));
    int rc = ensurePathBuffers(ctx, ctx->fp.npt);
    if (rc != RT_OK) return rc;
    const FrameParams &fp = ctx->fp;
    Batch bt;
    bt.g0 = 0;
    bt.K = 1;
    bt.firstFrame = frame;
    bt.frameStride = 1;
    bt.P = ctx->P;
    bt.planesInner = 0u;   // one plane: both orders coincide
    RT_HIP(hipMemsetAsync(ctx->pb.stages, 0, sizeof(Stage) * 2, ctx->stream));
    const int gGen = gridFor(ctx, fp.npt, 16);
    if (ctx->type == RT_TYPE_SPECTRAL) hipLaunchKernelGGL(k_generate<true>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, -1);
    else hipLaunchKernelGGL(k_generate<false>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, -1);
    launchTrace(ctx, 0, 0, fp.npt);
    RT_HIP(hipGetLastError());
    std::vector<HitRec> hits(fp.npt);
    RT_HIP(hipMemcpyAsync(hits.data(), ctx->pb.hits, (size_t)fp.npt * sizeof(HitRec), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = rt_sync(ctx)) != RT_OK) return rc;

    for (uint32_t q = 0; q < fp.npt; ++q) {
        uint32_t tile = q >> 6, l = q & 63u;
        int x = (int)(tile % (uint32_t)fp.tilesX) * 8 + (int)(l & 7u);
        int y = (int)(tile / (uint32_t)fp.tilesX) * 8 + (int)(l >> 3);
        if (x >= fp.W || y >= fp.H) continue;
        const size_t at = (size_t)y * fp.W + x;
        const HitRec &h = hits[q];
        t[at] = h.t;
        u[at] = 0.0f;
        v[at] = 0.0f;
        if (h.id == kMiss) {
            model[at] = -1;
            ident[at] = -1;
        } else if (h.id & kPrimBit) {
            model[at] = -2;
            ident[at] = (int32_t)(h.id & kIndexMask);
        } else {
            const PackedTri &pt = ctx->hTris[h.id & kIndexMask];
            model[at] = pt.model & kTriModelMask;
            ident[at] = pt.ident;
            u[at] = h.u;
            v[at] = h.v;
        }
    }
    return RT_OK;
}

int rt_eval_colour(rt_ctx *ctx, int fn, const float *in, size_t items, float *out) {
    static const int inWidth[5] = {1, 2, 4, 3, 0}, outWidth[5] = {3, 1, 1, 3, 193};
    if (ctx == nullptr || out == nullptr || fn < 0 || fn > 4 || items == 0 || items > (1u << 24) || (in == nullptr && inWidth[fn] != 0)) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    float *dIn = nullptr, *dOut = nullptr;
    const size_t inBytes = std::max<size_t>(items * inWidth[fn], 1) * sizeof(float), outBytes = items * outWidth[fn] * sizeof(float);
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&dIn), inBytes));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dOut), outBytes);
    if (e == hipSuccess && inWidth[fn] != 0) e = hipMemcpyAsync(dIn, in, items * inWidth[fn] * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_eval_colour, dim3(gridFor(ctx, items, 4)), dim3(kBlock), 0, ctx->stream, fn, dIn, (uint32_t)items, dOut);
        e = hipMemcpyAsync(out, dOut, outBytes, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(dIn);
    if (dOut != nullptr) hipFree(dOut);
    if (e != hipSuccess) {
        ctx->hipError = std::string("rt_eval_colour: ") + hipGetErrorString(e);
        return RT_ERR_HIP;
    }
    return RT_OK;
}

int rt_get_stats(rt_ctx *ctx, rt_stats *out) {
    if (ctx == nullptr || out == nullptr) return RT_ERR_INVALID;
    int rc = rt_sync(ctx);
    if (rc != RT_OK) return rc;
    unsigned long long c[kCounters] = {};
    if (ctx->pb.counters != nullptr) RT_HIP(hipMemcpy(c, ctx->pb.counters, sizeof c, hipMemcpyDeviceToHost));
    out->rays = c[0];
    out->camera_samples = ctx->cameraSamples;
    out->node_visits = c[1];
    out->tri_tests = c[2];
    out->shaded_hits = c[3];
    out->trace_ms = ctx->traceMs;
    out->total_ms = ctx->totalMs;
    out->trace_launches = ctx->traceLaunches;
    out->max_stack = c[4];
    out->retraced = c[5];
    out->wave_steps = c[6];
    out->active_lane_steps = c[7];
    out->leaf_visits = c[8];
    out->inner_phase_steps = c[9];
    out->refills = c[10];
    return RT_OK;
}

int rt_reset_stats(rt_ctx *ctx) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    int rc = rt_sync(ctx);
    if (rc != RT_OK) return rc;
    if (ctx->pb.counters != nullptr) RT_HIP(hipMemset(ctx->pb.counters, 0, kCounters * sizeof(unsigned long long)));
    ctx->traceMs = ctx->totalMs = 0.0;
    ctx->traceLaunches = 0;
    ctx->cameraSamples = 0;
    return RT_OK;
}

int rt_set_tile_shard(rt_ctx *ctx, uint32_t offset, uint32_t stride) {
    if (ctx == nullptr || stride == 0 || offset >= stride) return RT_ERR_INVALID;
    ctx->tileOffset = offset;
    ctx->tileStride = stride;
    ctx->fp.tileOffset = offset;
    ctx->fp.tileStride = stride;
    return RT_OK;
}

int rt_set_option(rt_ctx *ctx, int option, int value) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    switch (option) {
        case RT_OPT_SHADOW_EARLY_OUT:
            ctx->shadowEarlyOut = value != 0;
            ctx->fp.shadowEarlyOut = (ctx->shadowEarlyOut && ctx->type == RT_TYPE_RGB) ? 1 : 0;
            return RT_OK;
        case RT_OPT_REFILL_LANES:
            if (value < 1 || value > 64) return RT_ERR_INVALID;
            ctx->refillLanes = value;
            ctx->refillChosen = true;
            return RT_OK;
        case RT_OPT_INNER_SHARE:
            if (value < 1 || value > 64) return RT_ERR_INVALID;
            ctx->innerShare = value;
            return RT_OK;
        case RT_OPT_SPECTRAL_PDF_AS_WRITTEN:
            ctx->spectralPdfAsWritten = value != 0;
            ctx->fp.spectralPdfAsWritten = value != 0 ? 1 : 0;
            return RT_OK;
        case RT_OPT_STRATIFIED:
            // changes the number of samples per frame: takes effect at the next rt_set_frame
            ctx->stratified = value != 0;
            ctx->frameSet = false;
            return RT_OK;
        case RT_OPT_NEE:
            ctx->nee = value != 0;
            ctx->fp.nee = ctx->nee ? 1 : 0;
            return RT_OK;
        case RT_OPT_HERO_WAVELENGTH:
            ctx->heroWavelength = value != 0;
            ctx->fp.hero = (ctx->heroWavelength && ctx->type == RT_TYPE_SPECTRAL) ? 1 : 0;
            return RT_OK;
        case RT_OPT_TILE_MAJOR:
            ctx->planesInner = value != 0;
            return RT_OK;
        case RT_OPT_MAX_PATHS:
            if (value < 1) return RT_ERR_INVALID;
            ctx->maxPaths = (uint32_t)value;
            return RT_OK;
        case RT_OPT_PIPELINE:
            ctx->pipelineOn = value != 0;
            return RT_OK;
        case RT_OPT_WIDE_BVH:
            if (value < 0 || value > 4) return RT_ERR_INVALID;
            ctx->wideChosen = true;
            if (ctx->wideBvh == value) return RT_OK;
            ctx->wideBvh = value;
            if (!ctx->committed) return RT_OK;
            ctx->dirty = true;   // the derived tree belongs to the committed scene: rebuild it now
            return rt_commit_scene(ctx, ctx->objectCount, ctx->modelsCount, ctx->lightSourcesCount);
        default:
            return RT_ERR_INVALID;
    }
}

int rt_set_counters(rt_ctx *ctx, int enabled) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    ctx->countersOn = enabled != 0;
    return RT_OK;
}

const char *rt_strerror(int status) {
    switch (status) {
        case RT_OK: return "ok";
        case RT_ERR_INVALID: return "invalid argument";
        case RT_ERR_STATE: return "call out of order";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_NO_DEVICE: return "no usable GPU";
        case RT_ERR_SCENE: return "inconsistent scene streams";
        case RT_ERR_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char *rt_last_hip_error(rt_ctx *ctx) { return ctx ? ctx->hipError.c_str() : ""; }

}  // extern "C"
