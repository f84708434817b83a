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
// Build with -ffp-contract=off (see rt_math.h).

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_hip.h"
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
#define RT_WWIDE_ALPHA 5
#endif
constexpr int kStackDepth = 32;      // int stack[32], src/RayScene.cpp:682
#ifndef RT_WSTACK
#define RT_WSTACK 16
#endif
constexpr int kWideStack = RT_WSTACK;   // LDS stack entries of the four-wide variants; deeper entries go to memory
constexpr int kBlock = 256;
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
};

// Path index of sample plane k (0..K-1) of padded pixel q.  Tile-major order keeps the rays in
// flight at any moment inside a small screen region, so they share the tree nodes they open.
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
};

struct PathBuffers {
    RayRec *rays;        // [2P]  extension rays [0,P), shadow rays [P,2P)
    HitRec *hits;        // [2P]
    float4 *stA;         // [P]   RGB: throughput.rgb, prevBrdfPdf      Spectral: throughput, prevBrdfPdf, lambda, -
    float4 *stB;         // [P]   RGB: radiance.rgb, seed bits          Spectral: radiance, -, -, seed bits
    float4 *neeA, *neeB, *neeC;   // [P] pending NEE term of the current bounce
    uint32_t *qExt[2];   // [kSubQueues * subCap] each, ping-pong by bounce parity
    uint32_t *qShadow;   // [kSubQueues * subCap]
    uint32_t subCap;     // capacity of one sub-queue
    Stage *stages;       // [kMaxBounces + 2]
    unsigned long long *counters;  // [8]: 0 rays, 1 nodeVisits, 2 triTests, 3 shadedHits, 4 deepest traversal stack
    // Four-wide traversal only: stack entries beyond the LDS part, [wave][overflowDepth][64].
    int32_t *overflow;
    uint32_t overflowDepth;
};

__device__ __forceinline__ uint32_t laneId() { return threadIdx.x & 63u; }
// The lane number from the execution-mask counters; volatile, so every use re-reads it.
__device__ __forceinline__ uint32_t laneIdNow() {
    uint32_t x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

// Appends `value` for every lane with `want`; one atomic per wave.  Must be
// called by all lanes of the wave.
__device__ __forceinline__ void wavePush(bool want, uint32_t *queue, uint32_t *count, uint32_t value) {
    unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    uint32_t lane = laneId();
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (want) queue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// Path index -> pixel.  64 consecutive paths form one 8x8 pixel tile so a wave
// of primary rays is spatially compact.
__device__ __forceinline__ bool pathPixel(const FrameParams &fp, uint32_t q, int &x, int &y) {
    uint32_t tile = q >> 6, l = q & 63u;
    x = (int)(tile % (uint32_t)fp.tilesX) * 8 + (int)(l & 7u);
    y = (int)(tile / (uint32_t)fp.tilesX) * 8 + (int)(l >> 3);
    return x < fp.W && y < fp.H && (tile % fp.tileStride) == fp.tileOffset;
}

__device__ __forceinline__ uint32_t pathIndex(const FrameParams &fp, const Batch &bt, uint32_t k, uint32_t q) {
    return bt.planesInner ? ((((q >> 6) * bt.K + k) << 6) | (q & 63u)) : k * fp.npt + q;
}

// ---- k_generate -----------------------------------------------------------------------
// Camera ray of fragment main() (src/RayTracer.cpp:559-578 / 1333-1366): jitter
// in units of (1/W)*right and (1/H)*up, first draw horizontal.
//
// Hero-wavelength mode (src/RayTracer.cpp:1357-1364) traces four wavelengths per camera sample one
// after the other along the same camera ray, each from the same seed: the batch runs the bounce
// loop four times, and this kernel starts pass `heroPass` (0..3; -1 = mode off).  Passes after the
// first fold the finished wavelength into the sample's XYZ sum (kept in neeC, which the Spectral
// integrator does not otherwise use) and restart the path.
// get_hero_wavelengths, src/RayTracer.cpp:1259-1272, one component; mod(x, y) = x - y*floor(x/y).
__device__ __forceinline__ float heroWavelength(int k, float baseOffset) {
    const float stride = kWlRange / 4.0f;
    const float x = (kWlMin + stride * ((float)k + baseOffset)) - kWlMin;
    return (x - kWlRange * floor_(x / kWlRange)) + kWlMin;
}

template <bool SPECTRAL>
__global__ __launch_bounds__(kBlock) void k_generate(FrameParams fp, Batch bt, PathBuffers pb, int heroPass) {
    const uint32_t total = bt.K * fp.npt;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t base = blockIdx.x * kBlock + (threadIdx.x & ~63u); base < total; base += stride) {
        uint32_t p = base + laneId();
        bool valid = false;
        if (p < total) {
            uint32_t k, q;
            if (bt.planesInner) {
                const uint32_t w = p >> 6;
                k = w % bt.K;
                q = ((w / bt.K) << 6) | (p & 63u);
            } else {
                k = p / fp.npt;
                q = p - k * fp.npt;
            }
            int x, y;
            valid = pathPixel(fp, q, x, y);
            if (valid) {
                uint32_t g = bt.g0 + k;
                uint32_t frame = bt.firstFrame + (g / fp.S) * bt.frameStride;
                uint32_t s = g % fp.S;
                int i = (int)(s / (uint32_t)fp.ssq), j = (int)(s % (uint32_t)fp.ssq);
                uint32_t seed = hash_seed((uint32_t)x, (uint32_t)y, frame, s);

                float ux = (float)x * fp.rImgX * 2.0f - 1.0f;
                float uy = (float)y * fp.rImgY * 2.0f - 1.0f;
                V3 uv = fp.halfWidth * ux * fp.right + fp.halfHeight * uy * fp.up + fp.focalLength * fp.forward + fp.position;
                float jx = rand_float(seed);
                float jy = rand_float(seed);
                V3 d = uv + (((float)j + jx) * fp.rssq) * fp.rImgX * fp.right + (((float)i + jy) * fp.rssq) * fp.rImgY * fp.up;
                if (!SPECTRAL && !fp.stratified) d = uv + jx * fp.rImgX * fp.right + jy * fp.rImgY * fp.up;   // :589
                d = normalize(d - fp.position);

                RayRec r;
                r.ox = fp.position.x; r.oy = fp.position.y; r.oz = fp.position.z; r.pad0 = 0.0f;
                r.dx = d.x; r.dy = d.y; r.dz = d.z; r.pad1 = kNeverStop;
                pb.rays[p] = r;
                if (SPECTRAL && heroPass >= 0) {
                    // The seed is passed to traceColorWavelength by value (src/RayTracer.cpp:1051), so
                    // every wavelength of the sample starts from the same state: the one after this draw.
                    const float baseOffset = (rand_float(seed) + (float)(i * fp.ssq) + (float)j) * fp.wlDt;
                    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (heroPass > 0) {
                        sum = pb.neeC[p];
                        const V3 xyz = v3(sum.x, sum.y, sum.z) + cieXYZ(pb.stA[p].z) * (pb.stB[p].x / fp.wlPdf);
                        sum = make_float4(xyz.x, xyz.y, xyz.z, 0.0f);
                    }
                    pb.neeC[p] = sum;
                    pb.stA[p] = make_float4(1.0f, 1.0f, heroWavelength(heroPass, baseOffset), 0.0f);
                    pb.stB[p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(seed));
                } else if (SPECTRAL) {
                    float lambda = (rand_float(seed) + (float)(i * fp.ssq) + (float)j) * fp.wlDt + kWlMin;
                    pb.stA[p] = make_float4(1.0f, 1.0f, lambda, 0.0f);
                    pb.stB[p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(seed));
                } else {
                    pb.stA[p] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    pb.stB[p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(seed));
                }
            }
        }
        const uint32_t sub = blockIdx.x % kSubQueues;
        wavePush(valid, pb.qExt[0] + (size_t)sub * pb.subCap, &pb.stages[0].ext[sub].count, p);
    }
}

// ---- k_trace ----------------------------------------------------------------------------
// RayBoundingBoxDst, src/RayScene.cpp:417-431, with 1/direction hoisted per ray.
__device__ __forceinline__ float slabDistance(V3 bmin, V3 bmax, V3 ro, V3 inv, float t) {
    V3 tMin = (bmin - ro) * inv;
    V3 tMax = (bmax - ro) * inv;
    V3 t1 = v3(fminI_(tMin.x, tMax.x), fminI_(tMin.y, tMax.y), fminI_(tMin.z, tMax.z));
    V3 t2 = v3(fmaxI_(tMin.x, tMax.x), fmaxI_(tMin.y, tMax.y), fmaxI_(tMin.z, tMax.z));
    float nearD = fmax3_(t1.x, t1.y, t1.z);
    float farD = fmin3_(t2.x, t2.y, t2.z);
    // The reference returns early when nearD > t; as a select it is the same value without a branch.
    return (!(nearD > t) && farD >= nearD && farD > 0.0f) ? nearD : 1e20f;
}

// Byte k of a word as float (v_cvt_f32_ubyteK).
template <int K> __device__ __forceinline__ float byteAsFloat(uint32_t w) { return (float)((w >> (8 * K)) & 0xffu); }

// Entry distance of child K of a four-wide node, 1e20 when the ray misses it or enters it
// beyond `t`.  nx..fz are the words holding the planes the ray meets first / last on each axis,
// (sx, bx) map a plane number q to the ray parameter q*sx + bx.
typedef float float2v __attribute__((ext_vector_type(2)));
template <int K>
__device__ __forceinline__ float wideChildDistance(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz,
                                                   float2v sx, float2v sy, float2v sz, float2v bx, float2v by, float2v bz, float t) {
    // (near, far) of one axis as one packed multiply-add
    const float2v tx = __builtin_elementwise_fma((float2v){byteAsFloat<K>(nx), byteAsFloat<K>(fx)}, sx, bx);
    const float2v ty = __builtin_elementwise_fma((float2v){byteAsFloat<K>(ny), byteAsFloat<K>(fy)}, sy, by);
    const float2v tz = __builtin_elementwise_fma((float2v){byteAsFloat<K>(nz), byteAsFloat<K>(fz)}, sz, bz);
    const float nearD = fmax3_(tx.x, ty.x, tz.x);
    const float farD = fmin3_(tx.y, ty.y, tz.y);
    return (nearD <= t && farD >= nearD && farD > 0.0f) ? nearD : 1e20f;
}

// Orders two (distance, reference) pairs by distance.
__device__ __forceinline__ void orderPair(float &da, int32_t &ca, float &db, int32_t &cb) {
    const bool swap = db < da;
    const float dLo = swap ? db : da, dHi = swap ? da : db;
    const int32_t cLo = swap ? cb : ca, cHi = swap ? ca : cb;
    da = dLo; db = dHi; ca = cLo; cb = cHi;
}

// Alpha-cut rejection inside hitTriangle (src/RayScene.cpp:547-577).  Only
// triangles flagged kTriAlphaTest at commit time come here.  The traversal kernel
// runs it in a phase of its own, where no node or triangle temporaries are live.
__device__ __forceinline__ bool alphaCutRejects(const SceneView &sc, const PackedTri &pt, V3 ro, V3 rd, float t) {
    const ModelDesc &md = sc.modelDesc[pt.model & kTriModelMask];
    const float *idw = sc.models + md.identBase + pt.ident * kIdentWords;
    int32_t idx[3] = {f2i(idw[0]), f2i(idw[1]), f2i(idw[2])};
    TriVerts tv;
    loadTriVerts(sc, md.vertBase, idx, tv);
    const float *m = sc.materials + pt.material * kMaterialWords;
    int baseColor = f2i(m[1]);
    float alphaCut = m[20];
    V2 uv = cramerUV(tv.p[0], tv.p[1], tv.p[2], tv.uv[0], tv.uv[1], tv.uv[2], ro + t * rd);
    int at = texelOffset(sc, baseColor, uv);
    return sc.textures[at + 3] < alphaCut;
}

// Pins loaded values at this point of the program, so that the compiler cannot sink one of the
// loads below a later branch (which turns one round trip to memory into two dependent ones).
__device__ __forceinline__ void pinLoaded(const float4 &a, const float4 &b, const float4 &c) {
    asm volatile("" : : "v"(a.x), "v"(a.w), "v"(b.x), "v"(b.w), "v"(c.x), "v"(c.w));
}

// The same for whole 16-byte values: each must sit in four consecutive registers here, which keeps
// the compiler from re-cutting the record into loads that straddle the 16-byte pieces.
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pinLoaded(float4v &a, float4v &b, float4v &c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ float4 asFloat4(const float4v &v) { return make_float4(v.x, v.y, v.z, v.w); }

// Per-lane traversal state of the refilling variant.
struct LaneRay {
    V3 ro, rd, inv;
    float closest, afterPrimitives, tModel, maxT;
    float tStop;   // any accepted hit closer than this ends the ray at once (occluded shadow ray)
    float bestU, bestV, mU, mV;
    uint32_t bestId, mBest, slot;
    int32_t ref, model, sp;
    // Alpha-cut variant only: a candidate hit on an alpha-tested triangle waits here for the
    // texture lookup (pendTri != kMiss), and `resume` is where the leaf loop continues afterwards.
    // (Recomputing t,u,v in the lookup phase and folding `resume` into the leaf reference saves
    // four registers here but makes the compiler spill inside the main loop: measured slower.)
    float pendT, pendU, pendV;
    uint32_t pendTri, resume;
};

// Root references of the first models, copied to LDS by the traversal kernel: entering a model
// then costs an LDS read instead of a round trip to memory on every ray's critical path.
constexpr int kRootCache = 64;
constexpr int32_t kNoRoot = (int32_t)0x80000000;   // model without nodes; not a valid child reference

__device__ __forceinline__ bool enterModel(const SceneView &sc, const int32_t *rootLds, LaneRay &r, int m) {
    for (; m < sc.modelsCount; ++m) {
        int32_t root;
        if (rootLds != nullptr && m < kRootCache) {
            root = rootLds[m];
        } else {
            const ModelDesc &md = sc.modelDesc[m];
            root = md.nodesCount > 0 ? md.rootRef : kNoRoot;
        }
        if (root == kNoRoot) continue;
        r.model = m;
        r.ref = root;
        r.sp = 0;
        r.tModel = 1e20f;
        r.maxT = r.closest;
        r.mBest = kMiss;
        r.mU = 0.0f;
        r.mV = 0.0f;
        r.pendTri = kMiss;
        r.resume = 0u;
        return true;
    }
    return false;
}

// Analytic primitives, linear scan (hit(), src/RayScene.cpp:855-891); leaves the
// closest primitive hit in the lane state.
template <bool COUNT>
__device__ __forceinline__ void intersectPrimitives(const SceneView &sc, LaneRay &r, uint32_t &cntTris) {
    const V3 ro = r.ro, rd = r.rd;
    float closest = 1e20f;
    uint32_t bestId = kMiss;
    float bestU = 0.0f, bestV = 0.0f;
    int at = 0;
    for (int pi = 0; pi < sc.objectCount; ++pi) {
        const int type = ldi(sc.objects, at);
        at += 2;
        if (type == 0) {            // hitSphere, src/RayScene.cpp:441-466
            V3 center = ld3(sc.objects, at);
            float radius = ldw(sc.objects, at + 3);
            at += 4;
            V3 dir = center - ro;
            float a = dot(rd, rd);
            float h = dot(rd, dir);
            float c = dot(dir, dir) - radius * radius;
            float disc = h * h - a * c;
            if (!(disc < 0.0f)) {
                float sq = sqrt_(disc);
                float t = (h - sq) / a;
                bool ok = true;
                if (t <= 1e-8f || t >= closest) {
                    t = (h + sq) / a;
                    if (t <= 1e-8f || t >= closest) ok = false;
                }
                if (ok) { closest = t; bestId = kPrimBit | (uint32_t)pi; bestU = 0.0f; bestV = 0.0f; }
            }
        } else if (type == 1) {     // hitQuad, src/RayScene.cpp:478-508 (+ cull, :870)
            V3 qq = ld3(sc.objects, at), qu = ld3(sc.objects, at + 3), qv = ld3(sc.objects, at + 6);
            bool cull = ldw(sc.objects, at + 9) != 0.0f;
            at += 10;
            V3 normal = cross(qu, qv);
            if (!(cull && dot(rd, normal) > 0.0f)) {
                float denom = dot(normal, rd);
                float nn = dot(normal, normal);
                if (!(abs_(denom) < kMinDenominator)) {
                    float t = dot(normal, qq - ro) / denom;
                    if (!(t < 1e-8f || t > closest || t >= closest)) {
                        V3 rel = (ro + t * rd) - qq;
                        float alpha = dot(normal, cross(rel, qv)) / nn;
                        float beta = dot(normal, cross(qu, rel)) / nn;
                        if (!(alpha < 0.0f || alpha > 1.0f || beta < 0.0f || beta > 1.0f)) {
                            closest = t; bestId = kPrimBit | (uint32_t)pi; bestU = 0.0f; bestV = 0.0f;
                        }
                    }
                }
            }
        } else if (type == 2) {     // top-level Triangle through hitTriangle
            V3 A = ld3(sc.objects, at), B = ld3(sc.objects, at + 3), C = ld3(sc.objects, at + 6);
            at += 18;
            if (COUNT) cntTris++;
            V3 ab = B - A, ac = C - A;
            V3 nrm = cross(ab, ac);
            float det = -dot(rd, nrm);
            if (!(abs_(det) < 1e-8f)) {
                V3 ao = ro - A;
                V3 dao = cross(ao, rd);
                float invDet = 1.0f / det;
                float t = dot(ao, nrm) * invDet;
                if (!(t < 0.0f || t > closest || t >= closest)) {
                    float u = dot(ac, dao) * invDet;
                    float v = -dot(ab, dao) * invDet;
                    if (!(u < 0.0f || v < 0.0f || u + v > 1.0f)) {
                        closest = t; bestId = kPrimBit | (uint32_t)pi; bestU = u; bestV = v;
                    }
                }
            }
        }
    }
    r.closest = closest;
    r.afterPrimitives = closest;
    r.bestId = bestId;
    r.bestU = bestU;
    r.bestV = bestV;
}

// Persistent traversal kernel: a wave keeps up to 64 rays in flight
// and, whenever `refillLanes` or more lanes are idle, reserves that many queue
// entries with one atomic and hands them to the idle lanes (refillLanes = 64
// keeps whole 64-ray packets together, which is what coherent camera rays want;
// incoherent rays run best around 32).  The traversal stack of each lane lives
// in LDS ([entry][lane], conflict free).
//
// WIDE variants walk the models' four-wide trees instead: half as many dependent node loads per
// ray.  They find the same closest hit (the later model winning a tie between models, as in the
// reference), but not by the reference's visiting order, so between two triangles of one model at
// the same distance the other one may win.  (One tree over all models was tried: fewer nodes per
// ray but slower, because lanes then mix the small models' leaves with the large model's inner
// nodes and the two phases of the loop run emptier.)
template <bool COUNT, int STACK, bool ALPHA, bool PRIMS, bool ROOTS, bool WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? (ALPHA ? RT_WWIDE_ALPHA : RT_WWIDE)
                                          : (STACK <= 20 ? (ALPHA ? RT_WALPHA : PRIMS ? 7 : RT_W20) : STACK <= 24 ? (ALPHA ? (RT_WALPHA < 6 ? RT_WALPHA : 6) : 6) : 5)) void k_trace(SceneView sc, PathBuffers pb, int extParity, int stage, int refillLanes, int innerShare) {
    __shared__ int32_t stackLds[kBlock / 64][STACK][64];
    // ROOTS variants (scenes with more than two models): the models' root references sit in LDS.
    // With one or two models the descriptor loads hit the L1 and the plain variant is faster
    // (the table costs registers: -3 %); with six models the table gives +10 %.
    __shared__ int32_t rootTable[ROOTS ? kRootCache : 1];
    const uint32_t lane = laneId();
    int32_t(*stack)[64] = stackLds[threadIdx.x >> 6];
    const int32_t *rootLds = ROOTS ? rootTable : nullptr;
    if (ROOTS && (int)threadIdx.x < kRootCache) {
        int32_t root = kNoRoot;
        if ((int)threadIdx.x < sc.modelsCount) {
            const ModelDesc &md = sc.modelDesc[threadIdx.x];
            if (md.nodesCount > 0) root = md.rootRef;
        }
        rootTable[threadIdx.x] = root;
    }
    if (ROOTS) __syncthreads();

    Stage *st = &pb.stages[stage];
    const uint32_t *qExt = extParity >= 0 ? pb.qExt[extParity] : nullptr;
    const int firstSource = qExt ? 0 : kSubQueues;
    const int numSources = 2 * kSubQueues - firstSource;
    const uint32_t waveId = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    int source = firstSource + (int)(waveId % (uint32_t)numSources);
    bool drained = false;    // every source has been seen empty

    uint32_t cntNodes = 0, cntTris = 0, maxSp = 0;
    bool active = false;
    LaneRay r;
    r.slot = 0;
    // Four-wide variants: this lane's column of the stack part that does not fit in LDS.
    int32_t *overflow = WIDE ? pb.overflow + (size_t)waveId * pb.overflowDepth * 64 + lane : nullptr;
    // The four-wide variants run at the edge of their register budget: the lane's column of the LDS
    // stack is addressed from the lane number re-read on the spot (two instructions) rather than from
    // a value kept for the whole kernel, which the compiler would keep in scratch memory and reload
    // at every push.
    int32_t *const waveStack = &stackLds[0][0][0] + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (STACK * 64);
    auto stackSlot = [&](int32_t sp) -> int32_t * { return WIDE ? waveStack + sp * 64 + laneIdNow() : &stack[sp][lane]; };
    auto pushWide = [&](int32_t ref) {
        if (r.sp < STACK) *stackSlot(r.sp) = ref;
        else overflow[(size_t)(r.sp - STACK) * 64] = ref;
        r.sp++;
    };
    // Opens a four-wide node: the nearest child the ray enters becomes the lane's next reference,
    // the others go on the stack with the farthest at the bottom.  True when no child is entered.
    // Children are culled with maxT, the closest hit of this AND the earlier models (a triangle at
    // exactly that distance still replaces it, so boxes entered at maxT stay in); the reference
    // culls with the current model's hit only and walks every model's tree from the top.
    auto openWideNode = [&](const float4 &n0, const float4 &n1, const float4 &n2, const float4 &n3) -> bool {
        if (COUNT) cntNodes++;
        const float sx1 = n1.x * r.inv.x, sy1 = n1.y * r.inv.y, sz1 = n1.z * r.inv.z;
        const float bx1 = (n0.x - r.ro.x) * r.inv.x, by1 = (n0.y - r.ro.y) * r.inv.y, bz1 = (n0.z - r.ro.z) * r.inv.z;
        const float2v sx = {sx1, sx1}, sy = {sy1, sy1}, sz = {sz1, sz1}, bx = {bx1, bx1}, by = {by1, by1}, bz = {bz1, bz1};
        const uint32_t lox = __float_as_uint(n2.x), loy = __float_as_uint(n2.y), loz = __float_as_uint(n2.z);
        const uint32_t hix = __float_as_uint(n3.x), hiy = __float_as_uint(n3.y), hiz = __float_as_uint(n3.z);
        const bool ngx = r.inv.x < 0.0f, ngy = r.inv.y < 0.0f, ngz = r.inv.z < 0.0f;
        const uint32_t nx = ngx ? hix : lox, fx = ngx ? lox : hix;
        const uint32_t ny = ngy ? hiy : loy, fy = ngy ? loy : hiy;
        const uint32_t nz = ngz ? hiz : loz, fz = ngz ? loz : hiz;
        float d0 = wideChildDistance<0>(nx, ny, nz, fx, fy, fz, sx, sy, sz, bx, by, bz, r.maxT);
        float d1 = wideChildDistance<1>(nx, ny, nz, fx, fy, fz, sx, sy, sz, bx, by, bz, r.maxT);
        float d2 = wideChildDistance<2>(nx, ny, nz, fx, fy, fz, sx, sy, sz, bx, by, bz, r.maxT);
        float d3 = wideChildDistance<3>(nx, ny, nz, fx, fy, fz, sx, sy, sz, bx, by, bz, r.maxT);
        int32_t c0 = __float_as_int(n0.w), c1 = __float_as_int(n1.w), c2 = __float_as_int(n2.w), c3 = __float_as_int(n3.w);
        orderPair(d0, c0, d1, c1);
        orderPair(d2, c2, d3, c3);
        orderPair(d0, c0, d2, c2);
        orderPair(d1, c1, d3, c3);
        orderPair(d1, c1, d2, c2);
        if (!(d0 < 1e20f)) return true;
        if (d3 < 1e20f) { pushWide(c3); }
        if (d2 < 1e20f) { pushWide(c2); }
        if (d1 < 1e20f) { pushWide(c1); }
        r.ref = c0;
        return false;
    };
    // Möller–Trumbore as hitTriangle writes it (src/RayScene.cpp:510-545); returns true when the
    // hit waits for its alpha-cut lookup.
    auto testTriangle = [&](uint32_t ti, const float4 &q0, const float4 &q1, const float4 &q2) -> bool {
        if (COUNT) cntTris++;
        const V3 A = v3(q0.x, q0.y, q0.z), ab = v3(q1.x, q1.y, q1.z), ac = v3(q2.x, q2.y, q2.z);
        const V3 nrm = cross(ab, ac);
        const float det = -dot(r.rd, nrm);
        if (abs_(det) < 1e-8f) return false;
        const V3 ao = r.ro - A;
        const V3 dao = cross(ao, r.rd);
        const float invDet = 1.0f / det;
        const float t = dot(ao, nrm) * invDet;
        if (t < 0.0f || t > r.maxT) return false;
        // A later model's triangle at exactly the closest distance so far replaces the hit, a
        // triangle of the same model does not (src/RayScene.cpp:535, :405-406).  The four-wide walk
        // keeps one distance for both tests: maxT is this model's hit once mBest is set.
        if (WIDE ? (r.mBest != kMiss && t == r.maxT) : (t >= r.tModel)) return false;
        const float u = dot(ac, dao) * invDet;
        const float v = -dot(ab, dao) * invDet;
        if (u < 0.0f || v < 0.0f || u + v > 1.0f) return false;
        if (ALPHA && (__float_as_int(q1.w) & kTriAlphaTest)) {
            r.pendT = t;
            r.pendU = u;
            r.pendV = v;
            r.pendTri = ti;
            return true;
        }
        if (!WIDE) r.tModel = t;   // the four-wide walk keeps its closest hit in maxT alone
        r.maxT = t;
        r.mBest = ti;
        r.mU = u;
        r.mV = v;
        return false;
    };

    for (;;) {
        if (COUNT && active) maxSp = max(maxSp, (uint32_t)r.sp);
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!drained && nIdle >= refillLanes) {
            // `source` is the same in every lane; saying so keeps the queue header in scalar registers
            // and turns the load of its length into a scalar load.
            const int src = __builtin_amdgcn_readfirstlane(source);
            SubQueue *sq = (src < kSubQueues) ? &st->ext[src] : &st->shadow[src - kSubQueues];
            const uint32_t n = sq->count;
            const int leader = __ffsll((long long)idleMask) - 1;
            uint32_t base = n;
            if (n != 0u) {
                if ((int)lane == leader) base = atomicAdd(&sq->cursor, (uint32_t)nIdle);
                base = __shfl(base, leader);
            }
            if (base >= n) {
                bool has = false;
                if ((int)lane >= firstSource && (int)lane < 2 * kSubQueues) {
                    SubQueue *o = ((int)lane < kSubQueues) ? &st->ext[lane] : &st->shadow[lane - kSubQueues];
                    has = __hip_atomic_load(&o->cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < o->count;
                }
                const unsigned long long live = __ballot(has);
                if (live == 0ull) {
                    drained = true;
                } else {
                    int pick = (int)(waveId % (uint32_t)__popcll(live));
                    unsigned long long m = live;
                    while (pick-- > 0) m &= m - 1ull;
                    source = __ffsll((long long)m) - 1;
                }
                if (nIdle == 64) {
                    if (drained) break;
                    continue;
                }
            } else {
                const uint32_t mine = base + (uint32_t)__popcll(idleMask & ((1ull << lane) - 1ull));
                if (!active && mine < n) {
                    const uint32_t *q = (src < kSubQueues) ? qExt + (size_t)src * pb.subCap
                                                             : pb.qShadow + (size_t)(src - kSubQueues) * pb.subCap;
                    r.slot = q[mine];
                    const float4 *rp = reinterpret_cast<const float4 *>(&pb.rays[r.slot]);
                    const float4 r0 = rp[0], r1 = rp[1];
                    r.ro = v3(r0.x, r0.y, r0.z);
                    r.rd = v3(r1.x, r1.y, r1.z);
                    if (WIDE) {
                        // Only box tests use inv here, and the boxes carry a margin: the one-instruction
                        // reciprocal will do.  Plane numbers are scaled by inv, so keep it finite (a
                        // direction component of zero then behaves like a very small one of the same sign).
                        r.inv = v3(__builtin_amdgcn_rcpf(r.rd.x), __builtin_amdgcn_rcpf(r.rd.y), __builtin_amdgcn_rcpf(r.rd.z));
                        r.inv = v3(fminI_(fmaxI_(r.inv.x, -1e18f), 1e18f), fminI_(fmaxI_(r.inv.y, -1e18f), 1e18f), fminI_(fmaxI_(r.inv.z, -1e18f), 1e18f));
                    } else {
                        r.inv = v3(1.0f / r.rd.x, 1.0f / r.rd.y, 1.0f / r.rd.z);
                    }
                    r.tStop = r1.w;
                    if (PRIMS) {
                        intersectPrimitives<COUNT>(sc, r, cntTris);
                        active = !(r.closest < r.tStop) && enterModel(sc, rootLds, r, 0);
                        if (!active) {   // no models (or already occluded): the primitive result is final
                            HitRec rec;
                            rec.t = r.closest; rec.u = r.bestU; rec.v = r.bestV; rec.id = r.bestId;
                            *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                        }
                    } else {
                        // Without analytic primitives the record starts as a miss and is
                        // overwritten whenever a model ends with a closer hit, so the best
                        // hit need not be carried in registers.
                        r.closest = 1e20f;
                        HitRec rec;
                        rec.t = 1e20f; rec.u = 0.0f; rec.v = 0.0f; rec.id = kMiss;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                        active = enterModel(sc, rootLds, r, 0);
                    }
                }
            }
        } else if (nIdle == 64) {
            break;
        }

        // One phase per iteration, chosen for the whole wave: open inner nodes while
        // at least `innerShare`/64 of the active lanes sit on one, otherwise test the
        // leaves the other lanes are parked on.  Lanes of the other kind wait, so each
        // phase runs with most lanes enabled instead of both code paths every iteration.
        const bool onInner = active && r.ref >= 0;
        int nActive = 64 - __popcll(__ballot(!active));
        const int nInner = __popcll(__ballot(onInner));
        if (ALPHA) {
            // Third phase: candidate hits on alpha-tested triangles wait until eight lanes have one
            // (or nothing else is left to do) and are then resolved together; the texture lookup
            // needs many registers, and here none of the node/triangle temporaries are live.
            const bool pending = active && r.pendTri != kMiss;
            const int nPending = __popcll(__ballot(pending));
            if (nPending > 0 && (nPending >= 8 || nPending == nActive)) {
                if (pending) {
                    if (!alphaCutRejects(sc, sc.tris[r.pendTri], r.ro, r.rd, r.pendT)) {
                        if (!WIDE) r.tModel = r.pendT;
                        r.maxT = r.pendT;
                        r.mBest = r.pendTri;
                        r.mU = r.pendU;
                        r.mV = r.pendV;
                    }
                    r.resume = r.pendTri - ((~(uint32_t)r.ref) >> 2) + 1u;
                    r.pendTri = kMiss;
                }
                continue;
            }
            nActive -= nPending;
        }
        const bool innerPhase = nInner * 64 >= nActive * innerShare;
        bool pop = false, occluded = false;
        if (innerPhase) {
            if (WIDE) {
                if (onInner) {
                    const float4 *np = reinterpret_cast<const float4 *>(&sc.wnodes[r.ref]);
                    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                    pop = openWideNode(n0, n1, n2, n3);
                }
            } else if (onInner) {
                if (COUNT) cntNodes++;
                const float4 *np = reinterpret_cast<const float4 *>(&sc.nodes[r.ref]);
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                const int32_t left = __float_as_int(n0.w), right = __float_as_int(n1.w);
                const float dL = slabDistance(v3(n0.x, n0.y, n0.z), v3(n1.x, n1.y, n1.z), r.ro, r.inv, r.tModel);
                const float dR = slabDistance(v3(n2.x, n2.y, n2.z), v3(n3.x, n3.y, n3.z), r.ro, r.inv, r.tModel);
                const bool pushL = dL < r.tModel, pushR = dR < r.tModel;
                const bool leftFirst = dL < dR;
                const bool takeNear = leftFirst ? pushL : pushR;
                const bool takeFar = leftFirst ? pushR : pushL;
                const int32_t nearRef = leftFirst ? left : right;
                const int32_t farRef = leftFirst ? right : left;
                pop = true;
                if (takeNear) {
                    if (takeFar && r.sp < STACK) stack[r.sp++][lane] = farRef;
                    r.ref = nearRef;
                    pop = false;
                } else if (takeFar) {
                    r.ref = farRef;
                    pop = false;
                }
            }
        } else if (active && r.ref < 0 && !(ALPHA && r.pendTri != kMiss)) {
            const uint32_t code = ~(uint32_t)r.ref;
            const uint32_t first = code >> 2, ntri = code & 3u;
            bool parked = false;
            if (WIDE && !ALPHA) {
                // The usual leaf holds one or two triangles: fetch both in one round trip (the binary
                // variants have no registers to spare for this: -10 % there, +2 % here).
                const float4v *tp = reinterpret_cast<const float4v *>(&sc.tris[first]);
                if (ntri > 0u) {
                    const bool two = ntri > 1u;
                    float4v a0 = tp[0], a1 = tp[1], a2 = tp[2];
                    float4v b0 = a0, b1 = a1, b2 = a2;
                    if (two) { b0 = tp[3]; b1 = tp[4]; b2 = tp[5]; }
                    pinLoaded(a0, a1, a2);
                    pinLoaded(b0, b1, b2);
                    testTriangle(first, asFloat4(a0), asFloat4(a1), asFloat4(a2));
                    if (two) testTriangle(first + 1u, asFloat4(b0), asFloat4(b1), asFloat4(b2));
                    for (uint32_t k = 2u; k < ntri; ++k) {
                        float4v q0 = tp[3 * k], q1 = tp[3 * k + 1], q2 = tp[3 * k + 2];
                        pinLoaded(q0, q1, q2);
                        testTriangle(first + k, asFloat4(q0), asFloat4(q1), asFloat4(q2));
                    }
                }
            } else {
                for (uint32_t k = ALPHA ? r.resume : 0u; k < ntri; ++k) {
                    const uint32_t ti = first + k;
                    const float4 *tp = reinterpret_cast<const float4 *>(&sc.tris[ti]);
                    const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    pinLoaded(q0, q1, q2);   // all three in one round trip, before the determinant test
                    if (testTriangle(ti, q0, q1, q2)) {
                        parked = true;
                        break;
                    }
                }
            }
            if (!parked) {
                if (ALPHA) r.resume = 0u;
                pop = true;
                occluded = WIDE ? (r.mBest != kMiss && r.maxT < r.tStop) : (r.tModel < r.tStop);
            }
        }
        if (pop) {
            if (r.sp > 0 && !occluded) {
                --r.sp;
                if (WIDE && r.sp >= STACK) r.ref = overflow[(size_t)(r.sp - STACK) * 64];
                else r.ref = *stackSlot(r.sp);
            } else {
                if (r.mBest != kMiss) {
                    const float tHit = WIDE ? r.maxT : r.tModel;
                    r.closest = tHit;
                    if (PRIMS) {
                        r.bestId = r.mBest;
                        r.bestU = r.mU;
                        r.bestV = r.mV;
                    } else {
                        HitRec rec;
                        rec.t = tHit; rec.u = r.mU; rec.v = r.mV; rec.id = r.mBest;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                    }
                }
                if (occluded || !enterModel(sc, rootLds, r, r.model + 1)) {
                    if (PRIMS) {
                        uint32_t id = r.bestId;
                        if (id != kMiss && (id & kPrimBit) == 0u && !(r.afterPrimitives > r.closest)) id |= kNoTexBit;
                        HitRec rec;
                        rec.t = r.closest; rec.u = r.bestU; rec.v = r.bestV; rec.id = id;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                    }
                    active = false;
                }
            }
        }
    }

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long total = 0;
        for (int q = 0; q < kSubQueues; ++q) total += (qExt ? st->ext[q].count : 0u) + st->shadow[q].count;
        atomicAdd(&pb.counters[0], total);
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            cntNodes += __shfl_down(cntNodes, off);
            cntTris += __shfl_down(cntTris, off);
            maxSp = max(maxSp, (uint32_t)__shfl_down(maxSp, off));
        }
        if (lane == 0) {
            atomicAdd(&pb.counters[1], (unsigned long long)cntNodes);
            atomicAdd(&pb.counters[2], (unsigned long long)cntTris);
            atomicMax(&pb.counters[4], (unsigned long long)maxSp);
        }
    }
}

// ---- shading ---------------------------------------------------------------------------
// Adds the per-lane counts with one atomic per wave.  All lanes must call it.
__device__ __forceinline__ void addShadedHits(const PathBuffers &pb, uint32_t shaded) {
    for (int off = 32; off > 0; off >>= 1) shaded += __shfl_down(shaded, off);
    if (laneId() == 0 && shaded) atomicAdd(&pb.counters[3], (unsigned long long)shaded);
}

struct ShadowHit {
    V3 normal;       // s_info.normal (unflipped)
    bool frontFace;
    float t;
    V3 emissionColor;
    float emissionStrength;
};

// State of the shadow ray's closest hit.  A miss leaves the zero-initialised
// HitInfo with the default material 0 (the pinned reading of the
// uninitialised `s_info`, src/RayTracer.cpp:435-439).
__device__ __forceinline__ ShadowHit shadowHitOf(const SceneView &sc, const RayRec &sr, const HitRec &h, unsigned long long *counters) {
    ShadowHit s;
    if (h.id == kMiss) {
        MaterialD m0 = loadMaterial(sc, 0);
        s.normal = v3s(0.0f);
        s.frontFace = false;
        s.t = 1e20f;
        s.emissionColor = m0.emissionColor;
        s.emissionStrength = m0.emissionStrength;
    } else {
        HitInfoD info = resolveHit(sc, v3(sr.ox, sr.oy, sr.oz), v3(sr.dx, sr.dy, sr.dz), h);
        s.normal = info.normal;
        s.frontFace = info.front_face;
        s.t = info.t;
        s.emissionColor = info.mat.emissionColor;
        s.emissionStrength = info.mat.emissionStrength;
    }
    (void)counters;
    return s;
}

// NEE connection (src/RayTracer.cpp:439-477 RGB, :1162-1200 Spectral) for the
// shadow rays traced in `stage`; adds into the owning path's radiance.
template <bool SPECTRAL>
__global__ __launch_bounds__(kBlock) void k_connect(SceneView sc, Batch bt, PathBuffers pb, int stage, int countOn, int skipOccluded) {
    const uint32_t sub = blockIdx.x % kSubQueues;
    const uint32_t n = pb.stages[stage].shadow[sub].count;
    const uint32_t *qIn = pb.qShadow + (size_t)sub * pb.subCap;
    const uint32_t stride = (gridDim.x / kSubQueues) * kBlock;
    uint32_t shaded = 0;
    for (uint32_t i = (blockIdx.x / kSubQueues) * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t slot = qIn[i];
        const uint32_t path = slot - bt.P;
        const RayRec sr = pb.rays[slot];
        const HitRec h = pb.hits[slot];
        const float4 a = pb.neeA[path], b = pb.neeB[path], c = pb.neeC[path];
        // RGB: a hit nearer than distToLight - 1e-4 contributes nothing; skip its material.
        if (!SPECTRAL && skipOccluded && !(h.t >= c.z - 1e-4f)) continue;
        const ShadowHit s = shadowHitOf(sc, sr, h, pb.counters);
        if (h.id != kMiss) shaded++;
        const V3 sd = v3(sr.dx, sr.dy, sr.dz);
        if (!SPECTRAL) {
            const V3 brdfDirect = v3(a.x, a.y, a.z);
            const float cosTheta = a.w;
            const V3 rayColor = v3(b.x, b.y, b.z);
            const float pdfArea = b.w, dist2 = c.x, pdfBrdfLd = c.y, distToLight = c.z;
            if (s.emissionStrength > 0.0f && s.t >= distToLight - 1e-4f) {
                V3 nl = s.normal;
                if (!s.frontFace) nl = -nl;
                nl = normalize(nl);
                float cosThetaL = fmax_(dot(-sd, nl), 0.0f);
                float G = cosThetaL / dist2;
                float pdfNee = pdfArea / fmax_(G, kMinDenominator);
                float wNee = (pdfNee * pdfNee) / fmax_(pdfNee * pdfNee + pdfBrdfLd * pdfBrdfLd, kMinDenominator);
                V3 direct = brdfDirect * s.emissionColor * s.emissionStrength * cosTheta * G / pdfArea;
                float4 sb = pb.stB[path];
                V3 L = v3(sb.x, sb.y, sb.z) + rayColor * direct * wNee;
                pb.stB[path] = make_float4(L.x, L.y, L.z, sb.w);
            }
        } else {
            const float brdfDirect = a.x, cosTheta = a.y, pdf = a.z, dist2 = a.w;
            const float pdfBrdfLd = b.x, distToLight = b.y, throughput = b.z, lambda = b.w;
            if (s.emissionStrength > 0.0f && s.t <= distToLight + 0.01f) {
                float cosThetaL = abs_(dot(-sd, normalize(s.normal)));
                float G = cosThetaL / dist2;
                float pdfNee = pdf / fmax_(G, kMinDenominator);
                float wNee = (pdfNee * pdfNee) / fmax_(pdfNee * pdfNee + pdfBrdfLd * pdfBrdfLd, kMinDenominator);
                float energy = reflectanceAt(lambda, s.emissionColor);
                float direct = brdfDirect * energy * s.emissionStrength * cosTheta * G / pdf;
                float4 sb = pb.stB[path];
                sb.x = sb.x + throughput * direct * wNee;
                pb.stB[path] = sb;
            }
        }
        (void)c;
    }
    if (countOn) addShadedHits(pb, shaded);
}

// One bounce of traceColor (src/RayTracer.cpp:333-538) for every path in the stage's queue.
#ifndef RT_SHADE_BLOCKS
#define RT_SHADE_BLOCKS 4
#endif
__global__ __launch_bounds__(kBlock, RT_SHADE_BLOCKS) void k_shade_rgb(SceneView sc, FrameParams fp, Batch bt, PathBuffers pb, int stage, int countOn) {
    const uint32_t sub = blockIdx.x % kSubQueues;
    const uint32_t n = pb.stages[stage].ext[sub].count;
    const uint32_t *qIn = pb.qExt[stage & 1] + (size_t)sub * pb.subCap;
    uint32_t *qOut = pb.qExt[(stage + 1) & 1] + (size_t)sub * pb.subCap;
    uint32_t *qShadowOut = pb.qShadow + (size_t)sub * pb.subCap;
    Stage *next = &pb.stages[stage + 1];
    const bool lastBounce = stage >= fp.bounces;
    const uint32_t stride = (gridDim.x / kSubQueues) * kBlock;
    uint32_t shaded = 0;

    for (uint32_t base = (blockIdx.x / kSubQueues) * kBlock + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t i = base + laneId();
        bool wantShadow = false, wantExt = false;
        uint32_t path = 0;
        if (i < n) {
            path = qIn[i];
            const RayRec ray = pb.rays[path];
            const HitRec h = pb.hits[path];
            const float4 sA = pb.stA[path];
            float4 sB = pb.stB[path];
            V3 rayColor = v3(sA.x, sA.y, sA.z);
            float prevBrdfPdf = sA.w;
            V3 incoming = v3(sB.x, sB.y, sB.z);
            uint32_t seed = __float_as_uint(sB.w);
            const V3 ro = v3(ray.ox, ray.oy, ray.oz), rd = v3(ray.dx, ray.dy, ray.dz);

            if (h.id == kMiss) {                                    // sky, :338-344
                float t = rd.y * 0.5f + 0.5f;
                V3 env = (1.0f - t) * v3s(1.0f) + t * fp.sky;
                if (dot(fp.sky, fp.sky) > 0.0f) incoming = incoming + env * rayColor;
                pb.stB[path] = make_float4(incoming.x, incoming.y, incoming.z, sB.w);
            } else {
                shaded++;
                HitInfoD info = resolveHit(sc, ro, rd, h);
                V3 N = normalize(info.normal);
                const V3 V = normalize(-rd);
                if (!info.front_face) N = -N;
                const MaterialD &mat = info.mat;

                if (dot(mat.emissionColor, mat.emissionColor) > 0.0f && mat.emissionStrength > 0.0f) {   // :356-378
                    if (stage == 0 || !fp.nee) {   // without NEE every emitter hit counts in full (:374-376, :1089-1091)
                        incoming = incoming + rayColor * mat.emissionColor * mat.emissionStrength;
                    } else {
                        float wBrdf;
                        if (info.isPrimitive) {
                            wBrdf = 1.0f;   // analytic emitters are never light-sampled
                        } else {
                            float pdfArea = 1.0f / info.area / (float)sc.lightSourcesCount / (float)info.modelLightCounts;
                            float cosTheta = fmax_(dot(V, N), 0.0f);
                            float pdfNee = pdfArea * (info.t * info.t) / fmax_(cosTheta, kMinDenominator);
                            wBrdf = (prevBrdfPdf * prevBrdfPdf) / fmax_(prevBrdfPdf * prevBrdfPdf + pdfNee * pdfNee, kMinDenominator);
                        }
                        incoming = incoming + rayColor * wBrdf * mat.emissionColor * mat.emissionStrength;
                    }
                    pb.stB[path] = make_float4(incoming.x, incoming.y, incoming.z, sB.w);
                } else {
                    SurfaceRGB surf = surfaceOf(mat);
                    surf.roughness = fmax_(surf.roughness, 0.0005f);
                    const LobeProbs lp = lobeProbabilities(mat);

                    V3 L;
                    const float Xi = rand_float(seed);
                    float diff = 0.0f, spec = 0.0f, sss = 0.0f;
                    bool trans = false;
                    if (Xi < lp.diffuse) {
                        L = sampleCosineHemisphere(N, seed);
                        diff = 1.0f;
                    } else if (Xi < lp.diffuse + lp.specular) {
                        L = sampleSpecular<false>(N, V, surf.roughness, seed);
                        spec = 1.0f;
                    } else if (Xi < lp.diffuse + lp.specular + lp.transmission) {
                        L = sampleTransmission<false>(N, V, info.front_face, mat.ior, surf.roughness, seed);
                        trans = true;
                    } else {
                        L = sampleCosineHemisphere(N, seed);
                        sss = 1.0f;
                    }
                    L = normalize(L);

                    const V3 H = normalize(V + L);
                    const float NoV = clamp_(dot(N, V), 0.0f, 1.0f);
                    const float NoL = clamp_(dot(N, L), 0.0f, 1.0f);
                    const float NoH = clamp_(dot(N, H), 0.0f, 1.0f);
                    const float VoH = clamp_(dot(V, H), 0.0f, 1.0f);
                    const float LoV = clamp_(dot(L, V), 0.0f, 1.0f);

                    if (!trans && fp.nee) {                         // NEE set-up, :421-437 (compiled out with ENABLE_NEE 0)
                        float area, lightTriangleCount;
                        V3 p = sampleLightPoint(sc, seed, area, lightTriangleCount);
                        if (area > 0.0f) {
                            V3 so = info.point + N * 0.001f;
                            V3 toLight = p - so;
                            float distToLight = length(toLight);
                            V3 sdir = toLight / distToLight;
                            float cosTheta = dot(N, sdir);
                            if (cosTheta > 0.0f) {
                                V3 Hd = normalize(V + sdir);
                                float NoLd = clamp_(dot(N, sdir), 0.0f, 1.0f);
                                float NoHd = clamp_(dot(N, Hd), 0.0f, 1.0f);
                                float VoHd = clamp_(dot(V, Hd), 0.0f, 1.0f);
                                float LoVd = clamp_(dot(sdir, V), 0.0f, 1.0f);
                                float pdfArea = 1.0f / area / (float)sc.lightSourcesCount / lightTriangleCount;
                                float pdfBrdfLd = lp.diffuse * (NoLd * kInvPi) + lp.specular * specularPdfRGB(NoHd, NoV, surf.roughness) +
                                                  lp.subsurface * (NoLd * kInvPi);
                                V3 brdfDirect = lp.diffuse * evalDiffuseRGB(surf, NoLd, NoV, VoHd) +
                                                lp.specular * evalSpecularRGB(surf, NoV, NoLd, NoHd, VoHd) +
                                                lp.subsurface * evalSubsurfaceRGB(surf, NoLd, NoV, LoVd);
                                RayRec sr;
                                sr.ox = so.x; sr.oy = so.y; sr.oz = so.z; sr.pad0 = 0.0f;
                                // A hit nearer than distToLight - 1e-4 fails the visibility test
                                // (s_info.t >= distToLight - 1e-4) whatever lies behind it, so the
                                // traversal may stop at the first such hit without changing the result.
                                sr.dx = sdir.x; sr.dy = sdir.y; sr.dz = sdir.z; sr.pad1 = fp.shadowEarlyOut ? distToLight - 1e-4f : kNeverStop;
                                pb.rays[bt.P + path] = sr;
                                pb.neeA[path] = make_float4(brdfDirect.x, brdfDirect.y, brdfDirect.z, cosTheta);
                                pb.neeB[path] = make_float4(rayColor.x, rayColor.y, rayColor.z, pdfArea);
                                pb.neeC[path] = make_float4(dot(toLight, toLight), pdfBrdfLd, distToLight, 0.0f);
                                wantShadow = true;
                            }
                        }
                    }

                    V3 no = info.point + N * 0.001f;                // continue path, :484-537
                    bool alive = true;
                    if (trans) {
                        no = info.point + L * 0.001f;
                        if (!info.front_face) {
                            V3 al = vmax(mat.albedo, v3s(kMinDenominator));
                            V3 tr = v3(exp_(info.t * log_(al.x)), exp_(info.t * log_(al.y)), exp_(info.t * log_(al.z)));
                            rayColor = rayColor * tr;
                        }
                        prevBrdfPdf = 1.0f;
                    } else if (NoL < kMinDenominator) {
                        alive = false;
                    } else {
                        V3 brdfSss = evalSubsurfaceRGB(surf, NoL, NoV, LoV);
                        V3 brdfSpec = evalSpecularRGB(surf, NoV, NoL, NoH, VoH);
                        V3 brdfDiff = evalDiffuseRGB(surf, NoL, NoV, VoH);
                        float pSurf = 1.0f - lp.transmission;
                        pSurf = (pSurf < 1e-8f) ? 0.0f : pSurf;
                        float norm = (pSurf > 0.0f) ? 1.0f / pSurf : 1.0f;
                        float pdfDiffRaw = (NoL * kInvPi) * norm;
                        float pdfSpecRaw = specularPdfRGB(NoH, NoV, surf.roughness) * norm;
                        float pdfSssRaw = NoL * kInvPi * norm;
                        float pdfUsed = pdfDiffRaw * lp.diffuse * diff + pdfSpecRaw * lp.specular * spec + pdfSssRaw * lp.subsurface * sss;
                        prevBrdfPdf = pdfDiffRaw * lp.diffuse + pdfSpecRaw * lp.specular + pdfSssRaw * lp.subsurface;
                        V3 brdfTotal = (brdfSpec * spec + brdfDiff * diff + brdfSss * sss);
                        V3 contribution = (brdfTotal * NoL) / fmax_(pdfUsed, kMinDenominator);
                        rayColor = rayColor * contribution;
                        float rr = fmin_(fmax_(rayColor.x, fmax_(rayColor.y, rayColor.z)), 0.95f);
                        if (rand_float(seed) > rr) alive = false;
                        else rayColor = rayColor / rr;
                    }

                    if (alive && !lastBounce) {
                        RayRec nr;
                        nr.ox = no.x; nr.oy = no.y; nr.oz = no.z; nr.pad0 = 0.0f;
                        nr.dx = L.x; nr.dy = L.y; nr.dz = L.z; nr.pad1 = kNeverStop;
                        pb.rays[path] = nr;
                        pb.stA[path] = make_float4(rayColor.x, rayColor.y, rayColor.z, prevBrdfPdf);
                        pb.stB[path] = make_float4(incoming.x, incoming.y, incoming.z, __uint_as_float(seed));
                        wantExt = true;
                    }
                }
            }
        }
        wavePush(wantShadow, qShadowOut, &next->shadow[sub].count, bt.P + path);
        wavePush(wantExt, qOut, &next->ext[sub].count, path);
    }
    if (countOn) addShadedHits(pb, shaded);
}

// One bounce of traceColorWavelength (src/RayTracer.cpp:1056-1252).
__global__ __launch_bounds__(kBlock) void k_shade_spectral(SceneView sc, FrameParams fp, Batch bt, PathBuffers pb, int stage, int countOn) {
    const uint32_t sub = blockIdx.x % kSubQueues;
    const uint32_t n = pb.stages[stage].ext[sub].count;
    const uint32_t *qIn = pb.qExt[stage & 1] + (size_t)sub * pb.subCap;
    uint32_t *qOut = pb.qExt[(stage + 1) & 1] + (size_t)sub * pb.subCap;
    uint32_t *qShadowOut = pb.qShadow + (size_t)sub * pb.subCap;
    Stage *next = &pb.stages[stage + 1];
    const bool lastBounce = stage >= fp.bounces;
    const uint32_t stride = (gridDim.x / kSubQueues) * kBlock;
    uint32_t shaded = 0;

    for (uint32_t base = (blockIdx.x / kSubQueues) * kBlock + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t i = base + laneId();
        bool wantShadow = false, wantExt = false;
        uint32_t path = 0;
        if (i < n) {
            path = qIn[i];
            const RayRec ray = pb.rays[path];
            const HitRec h = pb.hits[path];
            const float4 sA = pb.stA[path];
            float4 sB = pb.stB[path];
            float throughput = sA.x, prevBrdfPdf = sA.y;
            const float lambda = sA.z;
            float radiance = sB.x;
            uint32_t seed = __float_as_uint(sB.w);
            const V3 ro = v3(ray.ox, ray.oy, ray.oz), rd = v3(ray.dx, ray.dy, ray.dz);

            if (h.id == kMiss) {                                    // :1061-1069
                float t = rd.y * 0.5f + 0.5f;
                V3 env = (1.0f - t) * v3s(1.0f) + t * fp.sky;
                if (dot(fp.sky, fp.sky) > 0.0f) radiance += reflectanceAt(lambda, env) * throughput;
                sB.x = radiance;
                pb.stB[path] = sB;
            } else {
                shaded++;
                HitInfoD info = resolveHit(sc, ro, rd, h);
                V3 N = normalize(info.normal);
                const V3 V = normalize(-rd);
                MaterialD &mat = info.mat;

                if (dot(mat.emissionColor, mat.emissionColor) > 0.0f && mat.emissionStrength > 0.0f) {   // :1075-1093
                    float energy = reflectanceAt(lambda, mat.emissionColor);
                    if (stage == 0 || !fp.nee) {   // without NEE every emitter hit counts in full (:374-376, :1089-1091)
                        radiance += energy * throughput * mat.emissionStrength;
                    } else {
                        float wBrdf;
                        if (info.isPrimitive) {
                            wBrdf = 1.0f;
                        } else {
                            float pdfArea = fp.spectralPdfAsWritten ? 1.0f / info.area : 1.0f / info.area / (float)sc.lightSourcesCount / (float)info.modelLightCounts;
                            float pdfNee = pdfArea * (info.t * info.t) / fmax_(abs_(dot(V, N)), kMinDenominator);
                            wBrdf = (prevBrdfPdf * prevBrdfPdf) / fmax_(prevBrdfPdf * prevBrdfPdf + pdfNee * pdfNee, kMinDenominator);
                        }
                        radiance += energy * throughput * mat.emissionStrength * wBrdf;
                    }
                    sB.x = radiance;
                    pb.stB[path] = sB;
                } else {
                    if (!info.front_face) N = -N;
                    const float roughness = fmax_(mat.roughness, 0.001f);
                    const LobeProbs lp = lobeProbabilities(mat);

                    V3 L;
                    const float Xi = rand_float(seed);
                    float diff = 0.0f, spec = 0.0f, sss = 0.0f;
                    bool trans = false;
                    if (Xi < lp.diffuse) {
                        L = sampleCosineHemisphere(N, seed);
                        diff = 1.0f;
                    } else if (Xi < lp.diffuse + lp.specular) {
                        L = sampleSpecular<true>(N, V, roughness, seed);
                        spec = 1.0f;
                    } else if (Xi < lp.diffuse + lp.specular + lp.transmission) {
                        // dispersion: ior + 0.03 * ((550/lambda)^2 - 1), :1122-1126
                        float ior = mat.ior + 0.03f * (pow2_(550.0f / lambda) - 1.0f);
                        L = sampleTransmission<true>(N, V, info.front_face, ior, roughness, seed);
                        trans = true;
                    } else {
                        L = sampleCosineHemisphere(N, seed);
                        sss = 1.0f;
                    }
                    L = normalize(L);

                    const V3 H = normalize(V + L);
                    const float NoV = clamp_(dot(N, V), 0.0f, 1.0f);
                    const float NoL = clamp_(dot(N, L), 0.0f, 1.0f);
                    const float NoH = clamp_(dot(N, H), 0.0f, 1.0f);
                    const float VoH = clamp_(dot(V, H), 0.0f, 1.0f);
                    const float LoV = clamp_(dot(L, V), 0.0f, 1.0f);

                    SurfaceSpectral surf;
                    surf.albedo = reflectanceAt(lambda, mat.albedo);
                    surf.roughness = roughness;
                    surf.metallic = mat.metallic;
                    surf.specular = mat.specular;
                    surf.specularTint = mat.specularTint;

                    if (!trans && fp.nee) {                         // :1148-1160 (no cosTheta gate here; compiled out with ENABLE_NEE 0)
                        float area, lightTriangleCount;
                        V3 p = sampleLightPoint(sc, seed, area, lightTriangleCount);
                        if (area > 0.0f) {
                            V3 so = info.point + N * 0.001f;
                            V3 toLight = p - so;
                            V3 sdir = normalize(toLight);
                            float distToLight = length(p - so);
                            float cosTheta = fmax_(dot(N, sdir), 0.0f);
                            float pdf = fp.spectralPdfAsWritten ? 1.0f / area : 1.0f / area / (float)sc.lightSourcesCount / lightTriangleCount;
                            V3 Hd = normalize(V + sdir);
                            float NoLd = clamp_(dot(N, sdir), 0.0f, 1.0f);
                            float NoHd = clamp_(dot(N, Hd), 0.0f, 1.0f);
                            float VoHd = clamp_(dot(V, Hd), 0.0f, 1.0f);
                            float LoVd = clamp_(dot(sdir, V), 0.0f, 1.0f);
                            float pSurf = 1.0f - lp.transmission;
                            pSurf = (pSurf < 1e-8f) ? 0.0f : pSurf;
                            float norm = (pSurf > 0.0f) ? 1.0f / pSurf : 1.0f;
                            float pdfBrdfLd = lp.diffuse * (NoLd * kInvPi) + lp.specular * specularPdfSpectral(NoHd, VoHd, roughness) +
                                              lp.subsurface * (NoLd * kInvPi);
                            pdfBrdfLd *= norm;
                            float brdfDirect = lp.diffuse * evalDiffuseSpectral(surf, NoLd, NoV, VoHd) +
                                               lp.specular * evalSpecularSpectral(surf, NoV, NoLd, NoHd, VoHd) +
                                               lp.subsurface * evalSubsurfaceSpectral(surf, NoLd, NoV, LoVd);
                            RayRec sr;
                            sr.ox = so.x; sr.oy = so.y; sr.oz = so.z; sr.pad0 = 0.0f;
                            sr.dx = sdir.x; sr.dy = sdir.y; sr.dz = sdir.z; sr.pad1 = kNeverStop;
                            pb.rays[bt.P + path] = sr;
                            pb.neeA[path] = make_float4(brdfDirect, cosTheta, pdf, dot(toLight, toLight));
                            pb.neeB[path] = make_float4(pdfBrdfLd, distToLight, throughput, lambda);
                            wantShadow = true;
                        }
                    }

                    const V3 no = info.point + L * 0.001f;          // :1205
                    bool alive = true;
                    if (trans) {
                        if (!info.front_face) {
                            float al = fmax_(surf.albedo, kMinDenominator);
                            float absorption = -log_(al);
                            throughput *= exp_(-absorption * info.t);
                        }
                        prevBrdfPdf = 1.0f;
                    } else if (NoL < kMinDenominator) {
                        alive = false;
                    } else {
                        float brdfSpec = evalSpecularSpectral(surf, NoV, NoL, NoH, VoH);
                        float brdfDiff = evalDiffuseSpectral(surf, NoL, NoV, VoH);
                        float brdfSss = evalSubsurfaceSpectral(surf, NoL, NoV, LoV);
                        float pSurf = 1.0f - lp.transmission;
                        pSurf = (pSurf < 1e-8f) ? 0.0f : pSurf;
                        float norm = (pSurf > 0.0f) ? 1.0f / pSurf : 1.0f;
                        float pdfSss = NoL * kInvPi * lp.subsurface * norm;
                        float pdfSpec = specularPdfSpectral(NoH, VoH, roughness) * lp.specular * norm;
                        float pdfDiff = (NoL * kInvPi) * lp.diffuse * norm;
                        float pdfUsed = pdfSss * sss + pdfSpec * spec + pdfDiff * diff;
                        prevBrdfPdf = pdfUsed;
                        float brdfTotal = (brdfSpec * spec + brdfDiff * diff + brdfSss * sss);
                        float contribution = (brdfTotal * NoL) / fmax_(pdfUsed, kMinDenominator);
                        throughput *= contribution;
                        float rr = fmin_(throughput, 0.95f);
                        if (rand_float(seed) > rr) alive = false;
                        else throughput /= rr;
                    }

                    if (alive && !lastBounce) {
                        RayRec nr;
                        nr.ox = no.x; nr.oy = no.y; nr.oz = no.z; nr.pad0 = 0.0f;
                        nr.dx = L.x; nr.dy = L.y; nr.dz = L.z; nr.pad1 = kNeverStop;
                        pb.rays[path] = nr;
                        pb.stA[path] = make_float4(throughput, prevBrdfPdf, lambda, 0.0f);
                        sB.x = radiance;
                        sB.w = __uint_as_float(seed);
                        pb.stB[path] = sB;
                        wantExt = true;
                    }
                }
            }
        }
        wavePush(wantShadow, qShadowOut, &next->shadow[sub].count, bt.P + path);
        wavePush(wantExt, qOut, &next->ext[sub].count, path);
    }
    if (countOn) addShadedHits(pb, shaded);
}

// ---- k_resolve ---------------------------------------------------------------------------
// Adds the batch's samples into the frame sum in sample order and, when a frame
// completes, scales by 1/ssq^2 (Spectral: XYZ -> linear sRGB) and adds it to the
// image sum (src/RayTracer.cpp:579-583, 1368-1378).  `partial` carries an
// unfinished frame's sample sum across batches.
template <bool SPECTRAL>
__global__ __launch_bounds__(kBlock) void k_resolve(FrameParams fp, Batch bt, PathBuffers pb, float4 *accum, float4 *partial) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < fp.npt; q += stride) {
        int x, y;
        if (!pathPixel(fp, q, x, y)) continue;
        const size_t px = (size_t)y * (size_t)fp.W + (size_t)x;
        V3 c = v3s(0.0f);
        float4 acc = accum[px];
        bool accDirty = false;
        for (uint32_t k = 0; k < bt.K; ++k) {
            const uint32_t s = (bt.g0 + k) % fp.S;
            if (s == 0u) {
                c = v3s(0.0f);
            } else if (k == 0u) {
                float4 pc = partial[px];
                c = v3(pc.x, pc.y, pc.z);
            }
            const uint32_t path = pathIndex(fp, bt, k, q);
            const float4 sB = pb.stB[path];
            if (SPECTRAL) {
                const float lambda = pb.stA[path].z;
                if (fp.hero) {   // hero_wavelengths_to_rgb, src/RayTracer.cpp:1274-1280: the fourth term and the mean
                    const float4 sum = pb.neeC[path];
                    c = c + (v3(sum.x, sum.y, sum.z) + cieXYZ(lambda) * (sB.x / fp.wlPdf)) / 4.0f;
                } else {
                    c = c + cieXYZ(lambda) * (sB.x / fp.wlPdf);
                }
            } else {
                c = c + v3(sB.x, sB.y, sB.z);
            }
            if (s == fp.S - 1u) {
                V3 col = (!SPECTRAL && !fp.stratified) ? c / (float)fp.rayPerPixel : c * (fp.rssq * fp.rssq);   // :592 / :582
                if (SPECTRAL) {
                    col = col / kCieYIntegral;
                    col = xyzEToLinearSRGB(col);
                }
                acc.x += col.x;
                acc.y += col.y;
                acc.z += col.z;
                acc.w += 1.0f;
                accDirty = true;
            } else if (k == bt.K - 1u) {
                partial[px] = make_float4(c.x, c.y, c.z, 0.0f);
            }
        }
        if (accDirty) accum[px] = acc;
    }
}

// ---- k_display ---------------------------------------------------------------------------
// The reference's screen pass (src/main.cpp:36-69) on the progressive mean:
// clamp below at 0, ACES filmic curve, NaN / negative / infinite debug colours,
// gamma 1/2.2; written as 8-bit RGBA (round to nearest).
__global__ __launch_bounds__(kBlock) void k_display(const float4 *accum, uchar4 *out, uint32_t pixels) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < pixels; i += gridDim.x * kBlock) {
        const float4 a = accum[i];
        const float n = a.w > 0.0f ? a.w : 1.0f;
        V3 c = vmax(v3(a.x / n, a.y / n, a.z / n), v3s(0.0f));
        c = (c * (2.51f * c + v3s(0.03f))) / (c * (2.43f * c + v3s(0.59f)) + v3s(0.14f));
        c = vclamp(c, 0.0f, 1.0f);
        V3 o;
        if (c.x != c.x || c.y != c.y || c.z != c.z) o = v3(1.0f, 0.0f, 0.0f);
        else if (c.x < 0.0f || c.y < 0.0f || c.z < 0.0f) o = v3(0.0f, 1.0f, 0.0f);
        else if (abs_(c.x) == inf() || abs_(c.y) == inf() || abs_(c.z) == inf()) o = v3(0.0f, 0.0f, 1.0f);
        else o = v3(pow_(c.x, 1.0f / 2.2f), pow_(c.y, 1.0f / 2.2f), pow_(c.z, 1.0f / 2.2f));
        out[i] = make_uchar4((unsigned char)(o.x * 255.0f + 0.5f), (unsigned char)(o.y * 255.0f + 0.5f),
                             (unsigned char)(o.z * 255.0f + 0.5f), 255);
    }
}

}  // namespace

// ============================================================================================
// Host side
// ============================================================================================

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string hipError;

    std::vector<float> host[6];     // uploaded words per slot (1..5)
    float *dStream[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t dStreamWords[6] = {0, 0, 0, 0, 0, 0};
    bool dirty = true;

    int objectCount = 0, modelsCount = 0, lightSourcesCount = 0;
    bool committed = false;
    PackedNode *dNodes = nullptr;
    WideNode *dWNodes = nullptr;    // four-wide tree (wideBvh only)
    bool wideBvh = false;
    int wideStackNeed = 0;          // most stack entries a ray can hold in the four-wide tree
    PackedTri *dTris = nullptr;
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
    int innerShare = 48;    // inner phase runs while >= innerShare/64 of the active lanes sit on inner nodes

    // RCCL communicator of rt_comm_init (library loaded on first use), one rank per context.
    void *comm = nullptr;
    int commRank = 0, commWorld = 1;
    unsigned long long *dFrameCount = nullptr;

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
    if (ctx->traceEvents.size() + ctx->renderEvents.size() >= kSyncAt && hipStreamSynchronize(ctx->stream) == hipSuccess) drainEvents(ctx);
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
int repackScene(rt_ctx *ctx, std::vector<PackedNode> &nodes, std::vector<PackedTri> &tris, std::vector<ModelDesc> &descs) {
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

    // Primitive stream: objectCount well-formed records.
    {
        size_t at = 0;
        for (int i = 0; i < ctx->objectCount; ++i) {
            if (at + 2 > obj.size()) return RT_ERR_SCENE;
            int type = f2i(obj[at]), material = f2i(obj[at + 1]);
            if (material < 0 || material >= materialCount) return RT_ERR_SCENE;
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
        d.nodeBase = d.lightBase + d.lightSourcesCount * kIdentWords;
        d.identBase = d.nodeBase + d.nodesCount * kNodeWords;
        d.vertBase = d.identBase + d.identifiersCount * kIdentWords;
        d.triBase = (int32_t)tris.size();
        if ((size_t)d.vertBase > mdl.size()) return RT_ERR_SCENE;
        const int vertexCapacity = (int)((mdl.size() - (size_t)d.vertBase) / 8);

        auto checkIdent = [&](int wordAt) -> bool {
            for (int k = 0; k < 3; ++k) {
                int vi = f2i(mdl[(size_t)wordAt + k]);
                if (vi < 0 || vi >= vertexCapacity) return false;
            }
            int mat = f2i(mdl[(size_t)wordAt + 3]);
            return mat >= 0 && mat < materialCount;
        };
        for (int l = 0; l < d.lightSourcesCount; ++l)
            for (int k = 0; k < 3; ++k) {
                int vi = f2i(mdl[(size_t)d.lightBase + (size_t)l * kIdentWords + k]);
                if (vi < 0 || vi >= vertexCapacity) return RT_ERR_SCENE;
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
        for (int n = 0; n < d.nodesCount; ++n) {
            if (nodeWord(n, 8) != 0.0f) continue;
            int l = f2i(nodeWord(n, 6)), r = f2i(nodeWord(n, 7));
            // Children must come after their parent (pre-order layout): rules out cycles.
            if (l <= n || r <= n) return RT_ERR_SCENE;
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
    if (tris.size() > (size_t)kIndexMask) return RT_ERR_SCENE;
    return RT_OK;
}

// Four-wide tree of every model, collapsed from the binary tree of the model stream: a node adopts
// the children of its child with the largest box until it has four (or only leaves are left).
// `descs` must have passed repackScene; rootRef is redirected to the new tree.
int buildWideNodes(rt_ctx *ctx, std::vector<ModelDesc> &descs, std::vector<WideNode> &wnodes) {
    const std::vector<float> &mdl = ctx->host[RT_SLOT_MODELS];
    ctx->wideStackNeed = 0;

    // All binary nodes in one array: box, children (indices into it), or the leaf reference.
    struct Bin { float box[6]; int32_t left, right; int32_t leafRef; bool leaf; };
    std::vector<Bin> bin;
    std::vector<std::pair<int32_t, ModelDesc *>> roots;
    double reach = 0.0;
    for (ModelDesc &d : descs) {
        if (d.nodesCount <= 0) continue;
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
    for (auto &rootOf : roots) {
        rootOf.second->rootRef = (int32_t)wnodes.size();
        wnodes.emplace_back();
        todo.push_back({rootOf.first, rootOf.second->rootRef, 0});
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
        for (int i = 0; i < nk; ++i)
            for (int w = 0; w < 6; ++w)
                if (!(std::fabs(bin[(size_t)kids[i]].box[w]) <= 1e15f)) return RT_ERR_SCENE;

        WideNode wn{};
        for (int a = 0; a < 3; ++a) {
            double lower = 0.0, upper = 0.0;
            for (int i = 0; i < nk; ++i) {
                const double mn = bin[(size_t)kids[i]].box[a], mx = bin[(size_t)kids[i]].box[3 + a];
                if (mn > mx) return RT_ERR_SCENE;
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
        int32_t refs[4] = {-1, -1, -1, -1};   // ~0: a leaf without triangles
        const int held = it.held + std::max(nk - 1, 0);
        ctx->wideStackNeed = std::max(ctx->wideStackNeed, held);
        for (int i = 0; i < nk; ++i) {
            if (bin[(size_t)kids[i]].leaf) {
                refs[i] = bin[(size_t)kids[i]].leafRef;
            } else {
                refs[i] = (int32_t)wnodes.size();
                wnodes.emplace_back();
                todo.push_back({kids[i], refs[i], held});
            }
        }
        wn.c0 = refs[0]; wn.c1 = refs[1]; wn.c2 = refs[2]; wn.c3 = refs[3];
        wnodes[(size_t)it.wide] = wn;
    }
    return RT_OK;
}

// (Re)allocates every per-path buffer for `want` paths.  ctx->P is zero while the set is incomplete,
// so a failed attempt can never leave a capacity that some buffer does not have.
int allocPathBuffers(rt_ctx *ctx, uint64_t want64) {
    if (want64 > 0x7fffffffull) return RT_ERR_INVALID;
    const uint32_t want = (uint32_t)want64;
    ctx->P = 0;
    int rc;
    if ((rc = devAlloc(ctx, &ctx->pb.rays, (size_t)want * 2)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.hits, (size_t)want * 2)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.stA, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.stB, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeA, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeB, want)) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->pb.neeC, want)) != RT_OK) return rc;
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
        if ((rc = devAlloc(ctx, &ctx->pb.counters, 8)) != RT_OK) return rc;
        RT_HIP(hipMemsetAsync(ctx->pb.counters, 0, 8 * sizeof(unsigned long long), ctx->stream));
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
        const uint64_t perPath = 2 * sizeof(RayRec) + 2 * sizeof(HitRec) + 5 * sizeof(float4) + 3 * sizeof(uint32_t) + 4;
        ceiling = std::min<uint64_t>(ceiling, (freeBytes / 2 + (uint64_t)ctx->P * perPath) / perPath);
    }
    uint64_t planes = std::max<uint64_t>(1, std::min<uint64_t>(ceiling / npt, (wantedPaths + npt - 1) / npt));
    if (ctx->pb.rays != nullptr && ctx->P >= planes * npt) return RT_OK;
    // If the device cannot give that much after all, retry with half the planes.
    for (;; planes = (planes + 1) / 2) {
        const int rc = allocPathBuffers(ctx, planes * npt);
        if (rc == RT_OK || rc != RT_ERR_HIP || planes == 1) return rc;
        (void)hipGetLastError();
    }
}

void launchTrace(rt_ctx *ctx, int extParity, int stage) {
    hipEvent_t e0 = takeEvent(ctx), e1 = takeEvent(ctx);
    if (e0) hipEventRecord(e0, ctx->stream);
    // Camera rays (stage 0) stay in 64-ray packets; later stages refill idle lanes.
    const int refill = stage == 0 ? 64 : ctx->refillLanes;
    // Kernel variant: visit counting, LDS stack depth (24 entries let 6 blocks share a
    // CU's LDS instead of 5), alpha-cut support only when some triangle needs it.
    using TraceFn = void (*)(SceneView, PathBuffers, int, int, int, int);
#define RT_TRACE_ROW(C, S, R) {{k_trace<C, S, false, false, R, false>, k_trace<C, S, false, true, R, false>}, {k_trace<C, S, true, false, R, false>, k_trace<C, S, true, true, R, false>}}
    static const TraceFn variants[2][3][2][2] = {{RT_TRACE_ROW(false, 20, false), RT_TRACE_ROW(false, 24, false), RT_TRACE_ROW(false, kStackDepth, false)},
                                                 {RT_TRACE_ROW(true, 20, false), RT_TRACE_ROW(true, 24, false), RT_TRACE_ROW(true, kStackDepth, false)}};
    static const TraceFn rootVariants[2][2][2] = {RT_TRACE_ROW(false, 20, true), RT_TRACE_ROW(false, 24, true)};
#undef RT_TRACE_ROW
#define RT_WIDE_ROW(C, R) {{k_trace<C, kWideStack, false, false, R, true>, k_trace<C, kWideStack, false, true, R, true>}, {k_trace<C, kWideStack, true, false, R, true>, k_trace<C, kWideStack, true, true, R, true>}}
    static const TraceFn wideVariants[2][2][2][2] = {{RT_WIDE_ROW(false, false), RT_WIDE_ROW(false, true)}, {RT_WIDE_ROW(true, false), RT_WIDE_ROW(true, true)}};
#undef RT_WIDE_ROW
    const int depthClass = ctx->maxTreeDepth <= 20 ? 0 : ctx->maxTreeDepth <= 24 ? 1 : 2;
    const bool prims = ctx->objectCount > 0;
    TraceFn fn;
    int wavesPerSimd;
    if (ctx->wideBvh) {
        fn = wideVariants[ctx->countersOn ? 1 : 0][ctx->modelsCount > 2 ? 1 : 0][ctx->hasAlphaTest ? 1 : 0][prims ? 1 : 0];
        wavesPerSimd = ctx->hasAlphaTest ? RT_WWIDE_ALPHA : RT_WWIDE;
    } else {
        const bool rootsInLds = ctx->modelsCount > 2 && depthClass < 2 && !ctx->countersOn;
        fn = rootsInLds ? rootVariants[depthClass][ctx->hasAlphaTest ? 1 : 0][prims ? 1 : 0]
                        : variants[ctx->countersOn ? 1 : 0][depthClass][ctx->hasAlphaTest ? 1 : 0][prims ? 1 : 0];
        wavesPerSimd = depthClass == 0 ? (ctx->hasAlphaTest ? RT_WALPHA : prims ? 7 : RT_W20) : depthClass == 1 ? (ctx->hasAlphaTest ? std::min(RT_WALPHA, 6) : 6) : 5;
    }
    const int grid = ctx->numCUs * wavesPerSimd;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->view, ctx->pb, extParity, stage, refill, ctx->innerShare);
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
            launchTrace(ctx, b & 1, b);
            if (b > 0 && fp.nee) {   // without NEE there are no shadow rays to connect
                if (spectral) hipLaunchKernelGGL(k_connect<true>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, b, cnt, fp.shadowEarlyOut);
                else hipLaunchKernelGGL(k_connect<false>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, b, cnt, fp.shadowEarlyOut);
            }
            if (spectral) hipLaunchKernelGGL(k_shade_spectral, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, fp, bt, ctx->pb, b, cnt);
            else hipLaunchKernelGGL(k_shade_rgb, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, fp, bt, ctx->pb, b, cnt);
        }
        const int last = fp.bounces + 1;
        if (fp.nee) {
            launchTrace(ctx, -1, last);
            if (spectral) hipLaunchKernelGGL(k_connect<true>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, last, cnt, fp.shadowEarlyOut);
            else hipLaunchKernelGGL(k_connect<false>, dim3(gShade), dim3(kBlock), 0, ctx->stream, ctx->view, bt, ctx->pb, last, cnt, fp.shadowEarlyOut);
        }
    }

    const int gRes = gridFor(ctx, fp.npt, 16);
    if (spectral) hipLaunchKernelGGL(k_resolve<true>, dim3(gRes), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, ctx->dAccum, ctx->dPartial);
    else hipLaunchKernelGGL(k_resolve<false>, dim3(gRes), dim3(kBlock), 0, ctx->stream, fp, bt, ctx->pb, ctx->dAccum, ctx->dPartial);
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
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->numCUs = prop.multiProcessorCount;
    if (const char *env = std::getenv("RT_MAX_PATHS")) {
        long v = std::atol(env);
        if (v > 0) ctx->maxPaths = (uint32_t)std::min<long>(v, 1l << 30);
    }
    if (const char *env = std::getenv("RT_TILE_MAJOR")) ctx->planesInner = std::atoi(env) != 0;
    if (const char *env = std::getenv("RT_WIDE_BVH")) ctx->wideBvh = std::atoi(env) != 0;
    if (const char *env = std::getenv("RT_INNER_SHARE")) {
        int v = std::atoi(env);
        if (v >= 1 && v <= 64) ctx->innerShare = v;
    }
    if (const char *env = std::getenv("RT_REFILL_LANES")) {
        int v = std::atoi(env);
        if (v >= 1 && v <= 64) ctx->refillLanes = v;
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx *ctx) {
    if (ctx == nullptr) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (float *p : ctx->dStream) if (p) hipFree(p);
    void *owned[] = {ctx->dNodes, ctx->dWNodes, ctx->pb.overflow, ctx->dTris, ctx->dModelDesc, ctx->accumExternal ? nullptr : (void *)ctx->dAccum, ctx->dPartial,
                     ctx->pb.rays, ctx->pb.hits, ctx->pb.stA, ctx->pb.stB, ctx->pb.neeA, ctx->pb.neeB, ctx->pb.neeC,
                     ctx->pb.qExt[0], ctx->pb.qExt[1], ctx->pb.qShadow, ctx->pb.stages, ctx->pb.counters};
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
    ctx->host[slot].assign(w, w + bytes / 4);
    ctx->dirty = true;
    ctx->committed = false;
    return RT_OK;
}

int rt_commit_scene(rt_ctx *ctx, int objectCount, int modelsCount, int lightSourcesCount) {
    if (ctx == nullptr || objectCount < 0 || modelsCount < 0 || lightSourcesCount < 0) return RT_ERR_INVALID;
    if (ctx->committed && !ctx->dirty && objectCount == ctx->objectCount && modelsCount == ctx->modelsCount &&
        lightSourcesCount == ctx->lightSourcesCount)
        return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->committed = false;   // a failed commit leaves the context without a scene, never with a stale one
    ctx->objectCount = objectCount;
    ctx->modelsCount = modelsCount;
    ctx->lightSourcesCount = lightSourcesCount;

    std::vector<PackedNode> nodes;
    std::vector<PackedTri> tris;
    std::vector<ModelDesc> descs;
    int rc = repackScene(ctx, nodes, tris, descs);
    if (rc != RT_OK) return rc;

    // Raw streams, each padded with 8 zero words so the one-past-the-end reads the
    // reference's texel fetches can make (4th channel of the last RGB texel) stay in bounds.
    for (int s = 1; s <= 5; ++s) {
        std::vector<float> padded(ctx->host[s]);
        padded.resize(padded.size() + 8, 0.0f);
        if ((rc = devAlloc(ctx, &ctx->dStream[s], padded.size())) != RT_OK) return rc;
        RT_HIP(hipMemcpy(ctx->dStream[s], padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
        ctx->dStreamWords[s] = ctx->host[s].size();
    }
    std::vector<WideNode> wnodes;
    if (ctx->wideBvh) {
        if ((rc = buildWideNodes(ctx, descs, wnodes)) != RT_OK) return rc;
        if ((rc = devAlloc(ctx, &ctx->dWNodes, wnodes.size())) != RT_OK) return rc;
        if (!wnodes.empty()) RT_HIP(hipMemcpy(ctx->dWNodes, wnodes.data(), wnodes.size() * sizeof(WideNode), hipMemcpyHostToDevice));
        // Stack entries that do not fit in LDS: one column per lane of every wave the grid can hold.
        const int deeper = std::max(0, ctx->wideStackNeed - kWideStack);
        if ((uint32_t)deeper > ctx->pb.overflowDepth || ctx->pb.overflow == nullptr) {
            const size_t waves = (size_t)ctx->numCUs * std::max(RT_WWIDE, RT_WWIDE_ALPHA) * (kBlock / 64);
            if ((rc = devAlloc(ctx, &ctx->pb.overflow, waves * (size_t)std::max(deeper, 1) * 64)) != RT_OK) return rc;
            ctx->pb.overflowDepth = (uint32_t)std::max(deeper, 1);
        }
    }
    if ((rc = devAlloc(ctx, &ctx->dNodes, nodes.size())) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->dTris, tris.size())) != RT_OK) return rc;
    if ((rc = devAlloc(ctx, &ctx->dModelDesc, descs.size())) != RT_OK) return rc;
    if (!nodes.empty()) RT_HIP(hipMemcpy(ctx->dNodes, nodes.data(), nodes.size() * sizeof(PackedNode), hipMemcpyHostToDevice));
    if (!tris.empty()) RT_HIP(hipMemcpy(ctx->dTris, tris.data(), tris.size() * sizeof(PackedTri), hipMemcpyHostToDevice));
    if (!descs.empty()) RT_HIP(hipMemcpy(ctx->dModelDesc, descs.data(), descs.size() * sizeof(ModelDesc), hipMemcpyHostToDevice));
    ctx->hTris.swap(tris);

    SceneView &v = ctx->view;
    v.objects = ctx->dStream[RT_SLOT_OBJECTS];
    v.materials = ctx->dStream[RT_SLOT_MATERIALS];
    v.modelInfo = ctx->dStream[RT_SLOT_MODEL_INFO];
    v.models = ctx->dStream[RT_SLOT_MODELS];
    v.textures = ctx->dStream[RT_SLOT_TEXTURES];
    v.nodes = ctx->dNodes;
    v.wnodes = ctx->wideBvh ? ctx->dWNodes : nullptr;
    v.tris = ctx->dTris;
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
    RT_HIP(hipStreamSynchronize(ctx->stream));
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

int rt_render(rt_ctx *ctx, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride) {
    if (ctx == nullptr || frame_stride == 0) return RT_ERR_INVALID;
    if (!ctx->committed || !ctx->frameSet || ctx->dAccum == nullptr) return RT_ERR_STATE;
    if (n_frames == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const FrameParams &fp = ctx->fp;
    const uint64_t totalPlanes = (uint64_t)n_frames * fp.S;
    if (totalPlanes > 0xffffffffull) return RT_ERR_INVALID;
    int rc = ensurePathBuffers(ctx, totalPlanes * fp.npt);
    if (rc != RT_OK) return rc;
    boundEvents(ctx);

    hipEvent_t e0 = takeEvent(ctx), e1 = takeEvent(ctx);
    if (e0) hipEventRecord(e0, ctx->stream);

    // Batches never exceed the current ceiling even when a larger buffer is already held.
    const uint32_t planesPerBatch = std::max(1u, std::min(ctx->P, std::max(ctx->maxPaths, fp.npt)) / fp.npt);
    for (uint64_t g = 0; g < totalPlanes; g += planesPerBatch) {
        Batch bt;
        bt.g0 = (uint32_t)g;
        bt.K = (uint32_t)std::min<uint64_t>(planesPerBatch, totalPlanes - g);
        bt.firstFrame = first_frame;
        bt.frameStride = frame_stride;
        bt.P = ctx->P;
        bt.planesInner = ctx->planesInner ? 1u : 0u;
        if ((rc = runBatch(ctx, bt)) != RT_OK) return rc;
    }
    if (e1) hipEventRecord(e1, ctx->stream);
    if (e0 && e1) ctx->renderEvents.push_back({e0, e1});
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

int rt_reset_accum(rt_ctx *ctx) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipMemsetAsync(ctx->dAccum, 0, (size_t)ctx->W * ctx->H * sizeof(float4), ctx->stream));
    ctx->framesAccum = 0;
    return RT_OK;
}

int rt_sync(rt_ctx *ctx) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
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

int rt_restore_accum(rt_ctx *ctx, const float *rgba_sum, size_t floats, uint32_t frames) {
    if (ctx == nullptr || rgba_sum == nullptr) return RT_ERR_INVALID;
    if (ctx->dAccum == nullptr) return RT_ERR_STATE;
    if (floats != (size_t)ctx->W * ctx->H * 4) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
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
    RT_HIP(hipStreamSynchronize(ctx->stream));
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
    launchTrace(ctx, 0, 0);
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

int rt_get_stats(rt_ctx *ctx, rt_stats *out) {
    if (ctx == nullptr || out == nullptr) return RT_ERR_INVALID;
    int rc = rt_sync(ctx);
    if (rc != RT_OK) return rc;
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
    return RT_OK;
}

int rt_reset_stats(rt_ctx *ctx) {
    if (ctx == nullptr) return RT_ERR_INVALID;
    int rc = rt_sync(ctx);
    if (rc != RT_OK) return rc;
    if (ctx->pb.counters != nullptr) RT_HIP(hipMemset(ctx->pb.counters, 0, 8 * sizeof(unsigned long long)));
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
        case RT_OPT_WIDE_BVH:
            if (ctx->wideBvh == (value != 0)) return RT_OK;
            ctx->wideBvh = value != 0;
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
