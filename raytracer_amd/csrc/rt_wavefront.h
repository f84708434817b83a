// rt_wavefront.h — state shared by the kernels of the wavefront pipeline: per-frame parameters,
// the batch descriptor, the sub-queues of every bounce, the per-path buffers, and the wave-level
// queue append.  Included by rt_hip.hip only (one translation unit; hence the unnamed namespace).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rt_bsdf.h"

using namespace rtd;

namespace {

#ifndef RT_W20
#define RT_W20 7
#endif
#ifndef RT_WALPHA
#define RT_WALPHA 6   // alpha-cut variants: 80 VGPRs; at 7 waves/SIMD (72) they spill inside the main loop
#endif
#ifndef RT_WWIDE
#define RT_WWIDE 7     // waves/SIMD of the four-wide variants (72 VGPRs with 40 B of spills: +4 % over 6 waves without)
#endif
#ifndef RT_WWIDE_ALPHA
#define RT_WWIDE_ALPHA 6   // alpha-cut four-wide variants WITH analytic primitives: 80 VGPRs with 28 B of spills (the ones without
                           // primitives write hit records through and fit RT_WWIDE waves with 12 B: interior 4K +4 % over 6 waves)
#endif
#ifndef RT_WSTACK
#define RT_WSTACK 16
#endif
constexpr int kWideStack = RT_WSTACK;   // LDS stack entries of the four-wide variants; deeper entries go to memory
constexpr int kBlock = 256;
constexpr int kCounters = 16;
constexpr int kMaxBounces = 62;      // stages kept on the device
// Default ceiling of paths in flight per batch (~190 B of device memory each).  Every kernel of
// a batch ends with a drain tail in which most of the chip idles, so few large batches beat many
// small ones: the default bench step (1920x1080 x 64 sample planes) fits in one batch.
constexpr uint32_t kDefaultMaxPaths = 128u << 20;
constexpr float kNeverStop = -1e30f;  // RayRec::pad1 of rays that must find their closest hit

// Per-frame uniforms plus values derived from them once on the host.
struct FrameParams {
    float fov, resX, resY;
    V3 position, forward, right, up;
    int32_t bounces, rayPerPixel;
    V3 sky;
    // fragment main(), src/RayTracer.cpp:545-563
    float rImgX, rImgY, focalLength, halfWidth, halfHeight;
    int32_t ssq;
    float rssq, wlDt, wlPdf;
    int32_t W, H, tilesX, tilesY;
    uint32_t npt;       // padded pixels per sample plane: tilesX*tilesY*64
    uint32_t S;         // samples per frame = ssq*ssq
    int32_t shadowEarlyOut;   // RGB shadow rays stop at the first hit that already proves occlusion
    int32_t spectralPdfAsWritten;   // Spectral: area pdf without the light-selection probability
    int32_t hero;             // Spectral: four wavelengths per camera sample (HERO_WAVELENGTH_ENABLE, src/RayTracer.cpp:607)
    int32_t nee;              // next-event estimation on (ENABLE_NEE, src/RayTracer.cpp:24, 606)
    int32_t stratified;       // RGB: stratified samples (`#if 1`, src/RayTracer.cpp:569); 0 = its #else branch, S = rayPerPixel
    // Tile-split sharding: this context renders only the 8x8 tiles whose row-major
    // index i satisfies i % tileStride == tileOffset (all tiles when stride is 1).
    uint32_t tileOffset, tileStride;
};

struct Batch {
    uint32_t g0;         // first global sample index (frame-major, sample-minor)
    uint32_t K;          // sample planes in this batch
    uint32_t firstFrame, frameStride;
    uint32_t P;          // capacity in paths; shadow-ray slots start at P
    uint32_t planesInner;   // path order: 1 = tile-major (all K samples of a tile adjacent), 0 = plane-major
    uint32_t waveSamples;   // tile-major only: samples per wave (1, 4, 16 or 64; divides K): a wave holds that many samples of 64 / waveSamples pixels
};

// Path index of sample plane k (0..K-1) of padded pixel q.  Tile-major order keeps the rays in
// flight at any moment inside a small screen region, so they share the tree nodes they open; with several samples
// per wave a wave's camera rays cover a 2x2-pixel block instead of 8x8 pixels, and so do the surface points its
// paths go on from.
__device__ __forceinline__ uint32_t pathIndex(const FrameParams &fp, const Batch &bt, uint32_t k, uint32_t q);


// Every ray queue is split into kSubQueues sub-queues, each with its length and
// its consumer cursor on a 128-byte line of its own: a single queue word would
// take one atomic per wave from every CU and saturates at ~90 atomics/us.
// Block b of any producer/consumer kernel owns sub-queue b % kSubQueues; the
// persistent traversal kernel drains all of them.
#ifndef RT_SUBQ
#define RT_SUBQ 32
#endif
constexpr int kSubQueues = RT_SUBQ;
static_assert(2 * kSubQueues <= 64, "the traversal kernel scans all sources with one wave-wide load");
struct __attribute__((aligned(128))) SubQueue {
    uint32_t count;      // entries appended by the producer kernel
    uint32_t cursor;     // next entry the traversal kernel hands out
    uint32_t pad[30];
};
// Queues of one bounce, all device resident.
struct Stage {
    SubQueue ext[kSubQueues];      // extension rays entering this bounce
    SubQueue shadow[kSubQueues];   // shadow rays produced by the previous bounce
    SubQueue retrace;              // rays of this stage to be traced again in the reference's order (PathBuffers::retrace)
};

struct PathBuffers {
    // Ray and hit records live at the position of the ray's queue entry, not under the path's number: record
    // region 0 and 1 belong to the extension queues of even and odd bounces, region 2 to the shadow queue, and
    // entry i of sub-queue j sits at region * Q + j * subCap + i (extRecord / shadowRecord below).  The kernel
    // that appends a ray to a queue writes its record there; the traversal kernel then needs no queue entry to
    // find the ray it was handed, and the consumers read whole lines.
    RayRec *rays;        // [3Q]
    HitRec *hits;        // [3Q]
    // State of a path that goes on, at the position of its extension-queue entry like its ray (regions 0 and 1):
    float4 *stA;         // [2Q]  RGB: throughput.rgb, prevBrdfPdf      Spectral: throughput, prevBrdfPdf, lambda, -
    float4 *stB;         // [2Q]  RGB: radiance.rgb, seed bits          Spectral: radiance, -, -, seed bits
    // What a path has gathered when it ends, under the path's number (k_resolve reads it in sample order); the
    // light connection of a path's last bounce is added here, that of a path that goes on to its record in stB.
    float4 *radiance;    // [P]   RGB: radiance.rgb, -                  Spectral: radiance, lambda, -, -
    float4 *neeA, *neeB, *neeC;   // [Q] pending NEE term of a shadow ray, at its shadow-queue position j * subCap + i
                                  //     (Spectral uses A and B only; its hero-wavelength mode keeps a per-path sum in neeC[path])
    uint32_t *qExt[2];   // [kSubQueues * subCap] each, ping-pong by bounce parity
    uint32_t *qShadow;   // [kSubQueues * subCap]
    uint32_t subCap;     // capacity of one sub-queue
    uint32_t Q;          // kSubQueues * subCap: records per region
    Stage *stages;       // [kMaxBounces + 2]: one per bounce and the last shadow stage
    unsigned long long *counters;  // [kCounters]: 0 rays, 1 nodeVisits, 2 triTests, 3 shadedHits, 4 deepest traversal stack, 5 rays traced again,
                                   // 6 wave steps of k_trace, 7 active lanes summed over them, 8 leaves opened, 9 steps in the inner-node phase, 10 refills, 11 shadow rays (part of 0)
    // Four-wide traversal only: stack entries beyond the LDS part, [wave][overflowDepth][64].
    int32_t *overflow;
    uint32_t overflowDepth;
    uint32_t overflowFrom;   // eight-wide traversal: stack entries below this number live in LDS
    // Four-wide traversal with exact ties: slots of rays that met two triangles of one model at the
    // same distance; they are traced again in the reference's visiting order.  The count is
    // stage's retrace.count (it may exceed the capacity: the surplus is not stored).
    uint32_t *retrace;
    uint32_t retraceCap;
    // Hero-wavelength mode: the camera rays' hit records of the sample's first wavelength, kept for the other
    // three, which start along the same camera rays ([P], allocated when the mode is first used).
    HitRec *cameraHits;
};

__device__ __forceinline__ uint32_t laneId() { return threadIdx.x & 63u; }
// The lane number from the execution-mask counters; volatile, so every use re-reads it.
__device__ __forceinline__ uint32_t laneIdNow() {
    uint32_t x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

// Appends `value` for every lane with `want` and returns the entry's position in the queue (undefined for the
// other lanes); one atomic per wave.  The lanes that reach the call take part: it may sit inside a branch.
__device__ __forceinline__ uint32_t wavePush(bool want, uint32_t *queue, uint32_t *count, uint32_t value) {
    unsigned long long mask = __ballot(want);
    if (mask == 0ull) return 0u;
    uint32_t lane = laneId();
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    const uint32_t at = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (want) queue[at] = value;
    return at;
}
// Where the ray and hit records of entry i of sub-queue `sub` live (see PathBuffers::rays).
__device__ __forceinline__ uint32_t extRecord(const PathBuffers &pb, int stage, uint32_t sub, uint32_t i) { return (uint32_t)(stage & 1) * pb.Q + sub * pb.subCap + i; }
// Shadow-queue entries: the path's number when the path ended at the bounce that sent the shadow ray (the light's
// contribution goes to radiance[path]), or this mark plus the position of the path's entry in the next bounce's
// extension sub-queue of the same number (the contribution goes to that record's stB).
constexpr uint32_t kGoesOn = 0x80000000u;
__device__ __forceinline__ uint32_t shadowRecord(const PathBuffers &pb, uint32_t sub, uint32_t i) { return 2u * pb.Q + sub * pb.subCap + i; }

// Path index -> pixel.  64 consecutive paths form one 8x8 pixel tile so a wave
// of primary rays is spatially compact.
__device__ __forceinline__ bool pathPixel(const FrameParams &fp, uint32_t q, int &x, int &y) {
    uint32_t tile = q >> 6, l = q & 63u;
    x = (int)(tile % (uint32_t)fp.tilesX) * 8 + (int)(l & 7u);
    y = (int)(tile / (uint32_t)fp.tilesX) * 8 + (int)(l >> 3);
    return x < fp.W && y < fp.H && (tile % fp.tileStride) == fp.tileOffset;
}

// Place of a pixel of an 8x8 tile along the Z curve (x in the even bits), and back: 4 consecutive places are a
// 2x2 block, 16 a 4x4 block.
__device__ __forceinline__ uint32_t tileCurve(uint32_t local) {   // local = y * 8 + x
    const uint32_t x = local & 7u, y = local >> 3;
    return (x & 1u) | ((y & 1u) << 1) | ((x & 2u) << 1) | ((y & 2u) << 2) | ((x & 4u) << 2) | ((y & 4u) << 3);
}
__device__ __forceinline__ uint32_t tileLocal(uint32_t curve) {
    const uint32_t x = (curve & 1u) | ((curve >> 1) & 2u) | ((curve >> 2) & 4u), y = ((curve >> 1) & 1u) | ((curve >> 2) & 2u) | ((curve >> 3) & 4u);
    return y * 8u + x;
}
// Tile-major order with B = waveSamples samples per wave: the K waves of a tile are K / B sample blocks of B pixel
// groups; wave (block b, group g) holds samples b B .. b B + B - 1 of the 64 / B pixels at curve places g 64 / B ...
__device__ __forceinline__ uint32_t pathIndex(const FrameParams &fp, const Batch &bt, uint32_t k, uint32_t q) {
    if (!bt.planesInner) return k * fp.npt + q;
    if (bt.waveSamples <= 1u) return (((q >> 6) * bt.K + k) << 6) | (q & 63u);
    const uint32_t B = bt.waveSamples, per = 64u / B, place = tileCurve(q & 63u);
    const uint32_t wave = (k / B) * B + place / per, lane = (k % B) * per + place % per;
    return (((q >> 6) * bt.K + wave) << 6) | lane;
}
// The inverse: sample plane and padded pixel of path p.
__device__ __forceinline__ void pathOf(const FrameParams &fp, const Batch &bt, uint32_t p, uint32_t &k, uint32_t &q) {
    if (!bt.planesInner) {
        k = p / fp.npt;
        q = p - k * fp.npt;
        return;
    }
    const uint32_t w = p >> 6, tile = w / bt.K, wave = w % bt.K, lane = p & 63u;
    if (bt.waveSamples <= 1u) {
        k = wave;
        q = (tile << 6) | lane;
        return;
    }
    const uint32_t B = bt.waveSamples, per = 64u / B;
    k = (wave / B) * B + lane / per;
    q = (tile << 6) | tileLocal((wave % B) * per + lane % per);
}

}  // namespace
