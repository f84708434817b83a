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
#include "rt_commit.h"
#include "rt_comm.h"
#include "rt_shade_fast.h"
#include "rt_tree_device.h"

using namespace rtd;
using namespace rtc;

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

struct rt_ctx : rt_scene_host {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string hipError;
    uchar4 *displayScratch = nullptr;   // this device's copy of `scratch` (hipGetSymbolAddress)

    float *dStream[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t dStreamWords[6] = {0, 0, 0, 0, 0, 0};
    bool dirty = true;

    bool committed = false;
    PackedNode *dNodes = nullptr;
    WideNode *dWNodes = nullptr;    // four-wide tree (wideBvh only)
    Wide8Node *dW8Nodes = nullptr;  // eight-wide tree (wideBvh 4)
    TriPair *dWPairs = nullptr;     // leaves of the four-wide trees as triangle pairs (scenes without alpha-cut textures)
    bool wideChosen = false;        // by option or environment: then a scene the four-wide build declines is an error
    int wideNow = 0;                // in effect for the committed scene (0 when the default was declined for it)
    PackedTri *dWTris = nullptr;    // triangles in wide-tree leaf order
    uint32_t overflowWords = 0;     // 4-byte words per lane allocated in pb.overflow
    PackedTri *dTris = nullptr;
    PackedShade *dShade = nullptr;
    PackedUV *dUVs = nullptr;
    LightTri *dLightTris = nullptr;
    int32_t *dPrimAt = nullptr;     // word offset of every analytic primitive's record in the objects stream
    ModelDesc *dModelDesc = nullptr;
    rtc::PackedTris hTris;   // kept for id decoding in rt_trace_primary
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
    bool shadowEarlyOut = true;
    bool spectralPdfAsWritten = false;
    bool heroWavelength = false;
    bool nee = true;
    bool stratified = true;
    uint32_t tileOffset = 0, tileStride = 1;
    bool fastShade = false;    // RT_OPT_FAST_SHADE: shading and connection kernels built with the hardware's own arithmetic
    bool treeOnDevice = false; // RT_OPT_TREE_BUILD / env RT_TREE_BUILD=device: the models' binary trees are built by rt_tree_device.hip
    bool textureU8 = true;     // RT_OPT_TEXTURE_U8 / env RT_TEX_U8: 8-bit textures are kept as RGBA8 words on the device
    float *dUnorm8 = nullptr;  // [512] k / 255.0f and srgb_to_linear of it (rt_create)
    uint64_t texturesPacked8 = 0, texturesKept32 = 0, textureBytesUploaded = 0, textureBytesDevice = 0;
    bool planesInner = true;   // tile-major path order inside a batch
    int waveSamples = 16;      // samples per wave of the tile-major order (Batch::waveSamples; a power of two up to 64, env RT_WAVE_SAMPLES):
                               // measured 1 / 8 / 16 / 32 / 64: 5513 / 5576 / 5581 / 5574 / 5514 Mrays/s on the default workload
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

// (validation and re-pack of the uploaded streams, the library's own trees and their collapse into wide nodes: rt_commit.cpp)

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
    // A sub-queue receives what the blocks mapped to it produce: at most its share
    // of the paths plus one block-round of slack per block feeding it.
    const size_t maxGrid = (size_t)ctx->numCUs * 16 + kSubQueues;
    const size_t subCap = ((size_t)want + kSubQueues - 1) / kSubQueues + (maxGrid / kSubQueues + 1) * kBlock;
    const size_t Q = subCap * kSubQueues;
    if (3 * Q > 0xffffffffull) return RT_ERR_INVALID;   // record numbers are 32 bits
    ctx->pb.subCap = (uint32_t)subCap;
    ctx->pb.Q = (uint32_t)Q;
    // Ray and hit records at queue positions: two extension regions (even / odd bounces) and the shadow region.
    if ((rc = devAlloc(ctx, &ctx->pb.rays, 3 * Q)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.hits, 3 * Q)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.stA, 2 * Q)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.stB, 2 * Q)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.radiance, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeA, Q)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeB, Q)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeC, Q)) != RT_OK) return rc;     // (>= want: the hero-wavelength mode indexes it by path)
    // Re-trace list of the four-wide walk: room for every ray one launch can trace.
    ctx->pb.retraceCap = 0;
    if ((rc = devAlloc(ctx, &ctx->pb.retrace, (size_t)want * 2)) != RT_OK) return rc;
    ctx->pb.retraceCap = want * 2;
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
        const uint64_t perPath = 3 * sizeof(RayRec) + 3 * sizeof(HitRec) + 8 * sizeof(float4) + 3 * sizeof(uint32_t) + 4 + 2 * sizeof(uint32_t) +   // the last: re-trace list
                                 (ctx->fp.hero ? sizeof(HitRec) : 0);   // hero-wavelength mode: the camera hits kept per path
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
    // (RT_HERO_OWN_CAMERA_RAYS=1 forces that fallback: a testing aid)
    static const bool ownCameraRays = std::getenv("RT_HERO_OWN_CAMERA_RAYS") != nullptr && std::atoi(std::getenv("RT_HERO_OWN_CAMERA_RAYS")) != 0;
    if (ctx->fp.hero && !ownCameraRays && ctx->pb.cameraHits == nullptr && devAlloc(ctx, &ctx->pb.cameraHits, ctx->P) != RT_OK) {
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

    auto connect = [&](int stage) {
        if (ctx->fastShade) rtFastConnect(spectral, gShade, ctx->stream, &ctx->view, &bt, &ctx->pb, stage, cnt, fp.shadowEarlyOut);
        else if (spectral) hipLaunchKernelGGL(k_connect<true>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, stage, cnt, fp.shadowEarlyOut);
        else hipLaunchKernelGGL(k_connect<false>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, stage, cnt, fp.shadowEarlyOut);
    };
    for (int pass = 0; pass < passes; ++pass) {
        RT_HIP(hipMemsetAsync(ctx->pb.stages, 0, sizeof(Stage) * ((size_t)fp.bounces + 2), ctx->stream));
        if (spectral) hipLaunchKernelGGL(k_generate<true>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, fp.hero ? pass : -1);
        else hipLaunchKernelGGL(k_generate<false>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, -1);

        for (int b = 0; b <= fp.bounces; ++b) {
            // The four wavelengths of a camera sample start along the same camera ray (same seed, src/RayTracer.cpp:1051,
            // 1357-1364): the camera rays are traced for the first and their hit records kept for the other three.
            const bool sameCameraRays = passes > 1 && b == 0 && ctx->pb.cameraHits != nullptr;
            if (sameCameraRays && pass > 0) {
                hipLaunchKernelGGL(k_camera_hits, dim3(gGen), dim3(kBlock), 0, ctx->stream, ctx->pb, 0);
            } else {
                launchTrace(ctx, b & 1, b, b == 0 ? paths : 2 * paths);
                if (sameCameraRays) hipLaunchKernelGGL(k_camera_hits, dim3(gGen), dim3(kBlock), 0, ctx->stream, ctx->pb, 1);
            }
            if (b > 0 && fp.nee) connect(b);   // without NEE there are no shadow rays to connect
            if (ctx->fastShade) rtFastShade(spectral, gShade, ctx->stream, &ctx->view, &fp, &bt, &ctx->pb, b, cnt);
            else if (spectral) hipLaunchKernelGGL(k_shade_spectral, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, fp, bt, ctx->pb, b, cnt);
            else hipLaunchKernelGGL(k_shade_rgb, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, fp, bt, ctx->pb, b, cnt);
        }
        const int last = fp.bounces + 1;
        if (fp.nee) {
            launchTrace(ctx, -1, last, paths);
            connect(last);
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

void closeComm(rt_ctx *ctx) {
    rtcomm::destroy(ctx->comm);
    ctx->comm = nullptr;
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

int rt_comm_unique_id(void *id, size_t bytes) { return rtcomm::uniqueId(id, bytes); }

int rt_comm_init(rt_ctx *ctx, int rank, int world, const void *id, size_t bytes) {
    if (ctx == nullptr || id == nullptr || bytes != rtcomm::kUniqueIdBytes || world < 1 || rank < 0 || rank >= world) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    closeComm(ctx);
    const int rc = rtcomm::init(&ctx->comm, rank, world, id, bytes, ctx->hipError);
    if (rc != RT_OK) return rc;
    ctx->commRank = rank;
    ctx->commWorld = world;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->dFrameCount), sizeof(unsigned long long)));
    return RT_OK;
}

int rt_reduce(rt_ctx *ctx, int root) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    if (ctx->comm == nullptr || ctx->dAccum == nullptr) return RT_ERR_STATE;
    if (root < 0 || root >= ctx->commWorld) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t floats = (size_t)ctx->W * ctx->H * 4;
    unsigned long long frames = ctx->framesAccum;
    accumWait(ctx);
    RT_HIP(hipMemcpyAsync(ctx->dFrameCount, &frames, sizeof frames, hipMemcpyHostToDevice, ctx->stream));
    // Both collectives are queued on the context's stream, behind the render work already there.
    const int rc = rtcomm::reduceSum(ctx->comm, reinterpret_cast<float *>(ctx->dAccum), floats, ctx->dFrameCount, root, ctx->stream, ctx->hipError);
    if (rc != RT_OK) return rc;
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
    // The 8-bit colour tables of textures kept as RGBA8 (512 floats).
    {
        float table[512];
        colourTables8(table);
        if (hipMalloc(reinterpret_cast<void **>(&ctx->dUnorm8), sizeof table) != hipSuccess ||
            hipMemcpy(ctx->dUnorm8, table, sizeof table, hipMemcpyHostToDevice) != hipSuccess) {
            if (ctx->dUnorm8) hipFree(ctx->dUnorm8);
            hipStreamDestroy(ctx->stream);
            delete ctx;
            return RT_ERR_HIP;
        }
    }
    if (const char *env = std::getenv("RT_TEX_U8")) ctx->textureU8 = std::atoi(env) != 0;
    if (const char *env = std::getenv("RT_TREE_BUILD")) ctx->treeOnDevice = std::strcmp(env, "device") == 0;
    if (const char *env = std::getenv("RT_FAST_SHADE")) ctx->fastShade = std::atoi(env) != 0;
    if (const char *env = std::getenv("RT_TILE_MAJOR")) ctx->planesInner = std::atoi(env) != 0;
    if (const char *env = std::getenv("RT_WAVE_SAMPLES")) {
        const int v = std::atoi(env);
        if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) ctx->waveSamples = v;
    }
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
        void *mine[] = {pipe.pb.rays, pipe.pb.hits, pipe.pb.stA, pipe.pb.stB, pipe.pb.radiance, pipe.pb.neeA, pipe.pb.neeB, pipe.pb.neeC, pipe.pb.qExt[0], pipe.pb.qExt[1],
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
    void *owned[] = {ctx->dNodes, ctx->dWNodes, ctx->dW8Nodes, ctx->dWPairs, ctx->pb.overflow, ctx->pb.retrace, ctx->dWTris, ctx->dTris, ctx->dShade, ctx->dUVs, ctx->dLightTris, ctx->dPrimAt, ctx->dModelDesc, ctx->dUnorm8, ctx->accumExternal ? nullptr : (void *)ctx->dAccum, ctx->dPartial,
                     ctx->pb.rays, ctx->pb.hits, ctx->pb.stA, ctx->pb.stB, ctx->pb.radiance, ctx->pb.neeA, ctx->pb.neeB, ctx->pb.neeC,
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

// rt_scene_host::deviceTree of a context whose model stream is on its device.
static int deviceTreeOfModel(rt_scene_host *host, const ModelDesc &d, std::vector<rtc::BinNode> &nodes, std::vector<int32_t> &order) {
    rt_ctx *ctx = static_cast<rt_ctx *>(host);
    const rttree::ModelTriangles model{ctx->dStream[RT_SLOT_MODELS], d.identBase, d.vertBase, d.identifiersCount, d.triBase};
    return rttree::buildBinaryTree(ctx->stream, model, nodes, order, ctx->hipError);
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

    rtc::PackedNodes nodes;
    rtc::PackedTris tris;
    rtc::PackedShades shade;
    rtc::PackedUVs uvs;
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

    // The textures the materials use, 8-bit ones as RGBA8 words (a quarter of the upload's bytes), and the
    // materials with their references moved along.
    TexturePack texturePack;
    packTextures(ctx, ctx->textureU8, texturePack);
    ctx->texturesPacked8 = texturePack.packed8;
    ctx->texturesKept32 = texturePack.kept32;
    ctx->textureBytesUploaded = texturePack.sourceBytes;
    ctx->textureBytesDevice = texturePack.words.size() * sizeof(float);
    lap("texture re-pack", mark);
    // Raw streams, each padded with 8 zero words so the one-past-the-end reads the
    // reference's texel fetches can make (4th channel of the last RGB texel) stay in bounds.
    for (int s = 1; s <= 5; ++s) {
        std::vector<float> &src = s == RT_SLOT_TEXTURES ? texturePack.words : s == RT_SLOT_MATERIALS ? texturePack.materials : ctx->host[s];
        const size_t words = src.size();
        if ((rc = devAlloc(ctx, &ctx->dStream[s], words + 8)) != RT_OK) return rc;
        if (words != 0) RT_HIP(hipMemcpy(ctx->dStream[s], src.data(), words * sizeof(float), hipMemcpyHostToDevice));
        RT_HIP(hipMemset(ctx->dStream[s] + words, 0, 8 * sizeof(float)));
        ctx->dStreamWords[s] = words;
    }
    lap("stream upload", mark);
    std::vector<WideNode> wnodes;
    std::vector<Wide8Node> w8nodes;
    rtc::PackedTris wtris;
    ctx->wideNow = tris.empty() ? 0 : ctx->wideBvh;   // nothing to build over analytic primitives alone
    // The eight-wide kernel has no alpha-cut phase: scenes with alpha-cut textures walk the four-wide trees.
    if (ctx->wideNow == 4 && ctx->hasAlphaTest) ctx->wideNow = 3;
    ctx->deviceTree = ctx->treeOnDevice ? deviceTreeOfModel : nullptr;
    if ((rc = devAlloc(ctx, &ctx->dTris, tris.size())) != RT_OK) return rc;
    if (!tris.empty()) RT_HIP(hipMemcpy(ctx->dTris, tris.data(), tris.size() * sizeof(PackedTri), hipMemcpyHostToDevice));
    // Four-wide trees and their leaves made on the device from the uploaded streams (the eight-wide trees take the
    // device's binary trees through the host's collapse).
    // (leaves as triangles take two records per leaf: their index must stay within the 29 bits of a leaf reference)
    bool wideOnDevice = ctx->treeOnDevice && (ctx->wideNow == 2 || ctx->wideNow == 3) && (!ctx->hasAlphaTest || tris.size() < (1u << 27));
    rttree::WideTrees deviceWide;
    if (wideOnDevice) {
        std::vector<rttree::ModelTriangles> models;
        for (const ModelDesc &d : descs)
            models.push_back({ctx->dStream[RT_SLOT_MODELS], d.identBase, d.vertBase, (d.nodesCount > 0 && d.identifiersCount > 0) ? d.identifiersCount : 0, d.triBase});
        rc = rttree::buildWideTrees(ctx->stream, models, ctx->dTris, ctx->hasAlphaTest, deviceWide, ctx->hipError);
        if (rc == RT_ERR_NOMEM) {
            // No room on the device for the build's scratch (a few hundred bytes per triangle): the host builds.
            (void)hipGetLastError();
            wideOnDevice = false;
            ctx->deviceTree = nullptr;
        } else if (rc == RT_ERR_SCENE && !ctx->wideChosen) {
            ctx->wideNow = 0;
        } else if (rc != RT_OK) {
            return rc;
        } else {
            for (size_t m = 0; m < descs.size(); ++m) {
                ModelDesc &d = descs[m];
                d.wideRootRef = models[m].count > 0 ? deviceWide.rootRefs[m] : d.rootRef;   // (no triangles: as the host build leaves it)
            }
            ctx->wideStackNeed = deviceWide.stackNeed;
        }
    }
    if (!wideOnDevice && ctx->wideNow) {
        rc = ctx->wideNow == 4 ? buildWide8(ctx, descs, w8nodes, tris, wtris) : buildWideNodes(ctx, descs, wnodes, tris, wtris);
        // A scene the wide build cannot take (coordinates beyond its box arithmetic, NaN) is
        // walked in the reference's order, as it would be with the option off, unless the caller
        // asked for the wide trees by name.
        if (rc == RT_ERR_SCENE && !ctx->wideChosen) ctx->wideNow = 0;
        else if (rc != RT_OK) return rc;
    }
    std::vector<TriPair> wpairs;
    if (!wideOnDevice && ctx->wideNow >= 1 && ctx->wideNow <= 3 && !ctx->hasAlphaTest) pairLeaves(wnodes, descs, wtris, wpairs);
    lap("wide trees", mark);
    if (ctx->wideNow && wideOnDevice) {
        // (the device build's buffers become the context's; a scene is walked by pairs or, with alpha-cut textures, by triangles)
        if (ctx->dWNodes) RT_HIP(hipFree(ctx->dWNodes));
        ctx->dWNodes = deviceWide.nodes;
        deviceWide.nodes = nullptr;
        if (ctx->hasAlphaTest) {
            if (ctx->dWTris) RT_HIP(hipFree(ctx->dWTris));
            ctx->dWTris = deviceWide.leafTris;
            deviceWide.leafTris = nullptr;
            if ((rc = devAlloc(ctx, &ctx->dWPairs, 0)) != RT_OK) return rc;
        } else {
            if (ctx->dWPairs) RT_HIP(hipFree(ctx->dWPairs));
            ctx->dWPairs = deviceWide.pairs;
            deviceWide.pairs = nullptr;
            if ((rc = devAlloc(ctx, &ctx->dWTris, 0)) != RT_OK) return rc;
        }
    } else if (ctx->wideNow) {
        if ((rc = devAlloc(ctx, &ctx->dWPairs, wpairs.size())) != RT_OK) return rc;
        if (!wpairs.empty()) RT_HIP(hipMemcpy(ctx->dWPairs, wpairs.data(), wpairs.size() * sizeof(TriPair), hipMemcpyHostToDevice));
        if ((rc = devAlloc(ctx, &ctx->dWTris, wtris.size())) != RT_OK) return rc;
        if (!wtris.empty()) RT_HIP(hipMemcpy(ctx->dWTris, wtris.data(), wtris.size() * sizeof(PackedTri), hipMemcpyHostToDevice));
        if ((rc = devAlloc(ctx, &ctx->dWNodes, wnodes.size())) != RT_OK) return rc;
        if (!wnodes.empty()) RT_HIP(hipMemcpy(ctx->dWNodes, wnodes.data(), wnodes.size() * sizeof(WideNode), hipMemcpyHostToDevice));
        if ((rc = devAlloc(ctx, &ctx->dW8Nodes, w8nodes.size())) != RT_OK) return rc;
        if (!w8nodes.empty()) RT_HIP(hipMemcpy(ctx->dW8Nodes, w8nodes.data(), w8nodes.size() * sizeof(Wide8Node), hipMemcpyHostToDevice));
    }
    if (ctx->wideNow) {
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
    v.unorm8 = ctx->dUnorm8;
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
    // Display frames begun at the old size have been delivered by the wait above; their slots are free again,
    // so that a late rt_display_rgba8_end reports RT_ERR_STATE instead of passing for a frame of the new size.
    for (rt_ctx::DisplaySlot &d : ctx->displaySlots) d.busy = false;
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
    // (the fence of the sum-buffer chain, so that the first render call on this stream creates nothing)
    if (pipe.fence == nullptr && hipEventCreateWithFlags(&pipe.fence, hipEventDisableTiming) != hipSuccess) pipe.fence = nullptr;
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
        // (the largest power of two up to the setting that divides the batch's planes: 12 planes run with 4 per wave)
        bt.waveSamples = (uint32_t)ctx->waveSamples;
        while (bt.waveSamples > 1u && bt.K % bt.waveSamples != 0u) bt.waveSamples >>= 1;
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
    bt.waveSamples = 1u;
    RT_HIP(hipMemsetAsync(ctx->pb.stages, 0, sizeof(Stage) * 2, ctx->stream));
    const int gGen = gridFor(ctx, fp.npt, 16);
    if (ctx->type == RT_TYPE_SPECTRAL) hipLaunchKernelGGL(k_generate<true>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, -1);
    else hipLaunchKernelGGL(k_generate<false>, dim3(gGen), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, -1);
    launchTrace(ctx, 0, 0, fp.npt);
    RT_HIP(hipGetLastError());
    // Hit records sit at the camera rays' queue positions; the queue entries say which path each belongs to.
    std::vector<HitRec> hits(fp.npt);
    {
        std::vector<Stage> stage0(1);
        std::vector<HitRec> byPosition(ctx->pb.Q);
        std::vector<uint32_t> entries(ctx->pb.Q);
        RT_HIP(hipMemcpyAsync(stage0.data(), ctx->pb.stages, sizeof(Stage), hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(hipMemcpyAsync(byPosition.data(), ctx->pb.hits, (size_t)ctx->pb.Q * sizeof(HitRec), hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(hipMemcpyAsync(entries.data(), ctx->pb.qExt[0], (size_t)ctx->pb.Q * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = rt_sync(ctx)) != RT_OK) return rc;
        HitRec none;
        none.t = 1e20f; none.u = 0.0f; none.v = 0.0f; none.id = kMiss;
        std::fill(hits.begin(), hits.end(), none);
        for (int sub = 0; sub < kSubQueues; ++sub) {
            const uint32_t n = std::min(stage0[0].ext[sub].count, ctx->pb.subCap);
            for (uint32_t i = 0; i < n; ++i) {
                const size_t at = (size_t)sub * ctx->pb.subCap + i;
                if (entries[at] < fp.npt) hits[entries[at]] = byPosition[at];
            }
        }
    }

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
    out->shadow_rays = c[11];
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
        case RT_OPT_FAST_SHADE:
            if (rtFastLayout[0] != sizeof(SceneView) || rtFastLayout[1] != sizeof(FrameParams) || rtFastLayout[2] != sizeof(Batch) || rtFastLayout[3] != sizeof(PathBuffers))
                return RT_ERR_STATE;   // the two translation units were built from different headers
            ctx->fastShade = value != 0;
            return RT_OK;
        case RT_OPT_TEXTURE_U8:
            if (ctx->textureU8 == (value != 0)) return RT_OK;
            ctx->textureU8 = value != 0;
            if (!ctx->committed) return RT_OK;
            ctx->dirty = true;   // the device copy of the textures belongs to the committed scene
            return rt_commit_scene(ctx, ctx->objectCount, ctx->modelsCount, ctx->lightSourcesCount);
        case RT_OPT_TREE_BUILD:
            if (value != RT_TREE_BUILD_HOST && value != RT_TREE_BUILD_DEVICE) return RT_ERR_INVALID;
            if (ctx->treeOnDevice == (value == RT_TREE_BUILD_DEVICE)) return RT_OK;
            ctx->treeOnDevice = value == RT_TREE_BUILD_DEVICE;
            if (!ctx->committed) return RT_OK;
            ctx->dirty = true;
            return rt_commit_scene(ctx, ctx->objectCount, ctx->modelsCount, ctx->lightSourcesCount);
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

int rt_get_info(rt_ctx *ctx, int what, uint64_t *value) {
    if (ctx == nullptr || value == nullptr) return RT_ERR_INVALID;
    switch (what) {
        case RT_INFO_TEXTURE_BYTES_UPLOADED: *value = ctx->textureBytesUploaded; return RT_OK;
        case RT_INFO_TEXTURE_BYTES_DEVICE: *value = ctx->textureBytesDevice; return RT_OK;
        case RT_INFO_TEXTURES_RGBA8: *value = ctx->texturesPacked8; return RT_OK;
        case RT_INFO_TEXTURES_FP32: *value = ctx->texturesKept32; return RT_OK;
        case RT_INFO_TREE_MODE: *value = (uint64_t)ctx->wideNow; return RT_OK;
        default: return RT_ERR_INVALID;
    }
}

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unversioned"
#endif
const char *rt_build_id(void) { return RT_BUILD_ID; }

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
