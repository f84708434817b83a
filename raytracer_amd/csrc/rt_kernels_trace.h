// rt_kernels_trace.h — the persistent traversal kernel k_trace and its variants: binary trees in
// the reference's visiting order, and the opt-in four-wide trees.
#pragma once

#include "rt_wavefront.h"

using namespace rtd;

namespace {

// Waves per SIMD the counting instances are compiled for: nobody times them, and at the plain instances'
// budget their nine extra counters cost 14-24 spilled registers.  -DRT_WCOUNT=0 gives them the plain
// instances' bounds (they are correct there too; tools/r03_hazard_experiment.sh).
#ifndef RT_WCOUNT
#define RT_WCOUNT 4
#endif
constexpr int kCountingWaves = RT_WCOUNT;

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
__device__ __forceinline__ bool alphaCutRejects(const SceneView &sc, uint32_t tri, V3 ro, V3 rd, float t) {
    const PackedTri &pt = sc.tris[tri];
    const PackedUV &ps = sc.uvs[tri];
    const float *m = sc.materials + pt.material * kMaterialWords;
    int baseColor = f2i(m[1]);
    float alphaCut = m[20];
    V2 uv = cramerUV(v3(pt.a[0], pt.a[1], pt.a[2]), v3(pt.ab[0], pt.ab[1], pt.ab[2]), v3(pt.ac[0], pt.ac[1], pt.ac[2]),
                     v2(ps.uv0[0], ps.uv0[1]), v2(ps.uv1[0], ps.uv1[1]), v2(ps.uv2[0], ps.uv2[1]), ro + t * rd);
    const TexelAt tx = texelAt(sc, baseColor, uv);
    return texelWord(sc, tx, 3) < alphaCut;
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
// Four-wide variants mark a ray that met two triangles of one model at the same closest distance in
// the sign of tStop (a distance, never negative): one register fewer than a flag of its own, and
// reading the magnitude back costs nothing.
__device__ __forceinline__ void markTie(LaneRay &r) { r.tStop = __int_as_float(__float_as_int(r.tStop) | (int)0x80000000); }
__device__ __forceinline__ bool hasTie(const LaneRay &r) { return __float_as_int(r.tStop) < 0; }

// Root references of the first models, copied to LDS by the traversal kernel: entering a model
// then costs an LDS read instead of a round trip to memory on every ray's critical path.
constexpr int kRootCache = 64;

// LEAN (the write-through instances of k_trace): the lane keeps neither the model's best hit nor a second
// copy of the closest distance; "this model has a hit" is bit 31 of r.model, cleared here.
template <bool WIDE, bool LEAN = false>
__device__ __forceinline__ bool enterModel(const SceneView &sc, const int32_t *rootLds, LaneRay &r, int m) {
    for (; m < sc.modelsCount; ++m) {
        int32_t root;
        if (rootLds != nullptr && m < kRootCache) {
            root = rootLds[m];
        } else {
            const ModelDesc &md = sc.modelDesc[m];
            root = md.nodesCount > 0 ? (WIDE ? md.wideRootRef : md.rootRef) : kNoRoot;
        }
        if (root == kNoRoot) continue;
        r.model = m;
        r.ref = root;
        r.sp = 0;
        if (!LEAN) {
            r.tModel = 1e20f;
            r.maxT = r.closest;
            r.mBest = kMiss;
            r.mU = 0.0f;
            r.mV = 0.0f;
        }
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
// the same distance the other one may win: a ray that meets such a tie is marked and, when the
// caller passes a re-trace list, appended to it for a second walk by a RETRACE instance (reference
// order over the uploaded trees), which overwrites its hit record.  (One tree over all models was tried: fewer nodes per
// ray but slower, because lanes then mix the small models' leaves with the large model's inner
// nodes and the two phases of the loop run emptier.)
template <bool COUNT, int STACK, bool ALPHA, bool PRIMS, bool ROOTS, bool WIDE, bool RETRACE = false>
__global__ __launch_bounds__(kBlock, (COUNT && kCountingWaves > 0) ? kCountingWaves : WIDE ? ((ALPHA && PRIMS) ? RT_WWIDE_ALPHA : RT_WWIDE)
                                          : (STACK <= 20 ? (ALPHA ? RT_WALPHA : PRIMS ? 7 : RT_W20) : STACK <= 24 ? (ALPHA ? (RT_WALPHA < 6 ? RT_WALPHA : 6) : 6) : 5)) void k_trace(SceneView sc, PathBuffers pb, int extParity, int stage, int refillLanes, int innerShare) {
    // Four-wide instances write a hit record the moment a hit is accepted (the primitives' best hit when the ray
    // starts, then every accepted triangle) and keep only "this model has a hit" (bit 31 of the model number),
    // not the best hit's u, v and id nor a second copy of the closest distance: three to seven registers fewer.
    constexpr bool kWriteThrough = WIDE;
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
            if (md.nodesCount > 0) root = WIDE ? md.wideRootRef : md.rootRef;
        }
        rootTable[threadIdx.x] = root;
    }
    if (ROOTS) __syncthreads();

    Stage *st = &pb.stages[stage];
    // extParity 0/1: the bounce's extension queue; -1: shadow rays only.  RETRACE instances read the
    // stage's re-trace list left by the four-wide walk instead (one queue, whose count may exceed what was stored).
    constexpr bool retraceList = RETRACE;   // its own instances, so that the others carry none of it
    const uint32_t *qExt = retraceList ? pb.retrace : extParity >= 0 ? pb.qExt[extParity] : nullptr;
    auto lengthOf = [&](const SubQueue *q) -> uint32_t { return retraceList ? min(q->count, pb.retraceCap) : q->count; };
    const int firstSource = qExt ? 0 : kSubQueues;
    const int numSources = retraceList ? 1 : 2 * kSubQueues - firstSource;   // the list is one queue
    auto queueOf = [&](int src) -> SubQueue * {
        return retraceList ? &st->retrace : (src < kSubQueues) ? &st->ext[src] : &st->shadow[src - kSubQueues];
    };
    const uint32_t waveId = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    int source = firstSource + (int)(waveId % (uint32_t)numSources);
    bool drained = false;    // every source has been seen empty
    // Camera rays go out in whole 64-ray hands (refillLanes 64), all about as long: the hands of extension
    // sub-queue s are dealt in turn to the waves numbered s, s + kSubQueues, ... instead of being taken from a
    // cursor, which costs every hand an atomic's round trip before its rays can even be loaded (the stage's
    // launch: 9.4 -> 6.6 ms on the default workload).  Neighbouring hands go to different waves, so every wave
    // gets its share of the screen's cheap and dear parts.
    const bool dealtHands = !ALPHA && !retraceList && refillLanes >= 64 && extParity >= 0;   // (the alpha-cut instances have no register to spare for it)
    const uint32_t handStride = (gridDim.x * (kBlock / 64)) / (uint32_t)kSubQueues;   // waves per sub-queue that take part
    uint32_t nextHand = waveId / (uint32_t)kSubQueues;
    if (dealtHands) {
        source = (int)(waveId % (uint32_t)kSubQueues);
        if (nextHand >= handStride) drained = true;   // (the few waves beyond a whole number of rounds)
    }

    uint32_t cntNodes = 0, cntTris = 0, maxSp = 0;
    // COUNT only, wave-uniform: loop steps that ran a phase, the active lanes summed over those steps,
    // steps that ran the inner-node phase, refills; per lane: leaves opened.
    uint32_t cntSteps = 0, cntActive = 0, cntInnerSteps = 0, cntRefills = 0, cntLeaves = 0;
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
    auto modelHasHit = [&]() -> bool { return kWriteThrough ? r.model < 0 : r.mBest != kMiss; };
    auto writeMiss = [&]() {
        HitRec rec;
        rec.t = 1e20f; rec.u = 0.0f; rec.v = 0.0f; rec.id = kMiss;
        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
    };
    // Records an accepted hit: in the lane, or (write-through instances) in the ray's hit record at once.
    auto acceptHit = [&](float t, float u, float v, uint32_t tri) {   // (tri by value: the no-texture mark is added to the copy)
        r.maxT = t;
        if (kWriteThrough) {
            r.model |= (int32_t)0x80000000;
            // a model hit that only ties the closest primitive takes no textures (`startClosest > closest`, src/RayScene.cpp:763)
            if (PRIMS && !(r.afterPrimitives > t)) tri |= kNoTexBit;
            HitRec rec;
            rec.t = t; rec.u = u; rec.v = v; rec.id = tri;
            *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
        } else {
            r.mBest = tri;
            r.mU = u;
            r.mV = v;
        }
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
        const bool sameModelTie = WIDE && modelHasHit() && t == r.maxT;
        if (!WIDE && t >= r.tModel) return false;
        const float u = dot(ac, dao) * invDet;
        const float v = -dot(ab, dao) * invDet;
        if (u < 0.0f || v < 0.0f || u + v > 1.0f) return false;
        if (sameModelTie) {
            // Which of the two the reference reports depends on its visiting order: remember it.
            markTie(r);
            return false;
        }
        if (ALPHA && (__float_as_int(q1.w) & kTriAlphaTest)) {
            r.pendT = t;
            r.pendU = u;
            r.pendV = v;
            r.pendTri = ti;
            return true;
        }
        if (!WIDE) r.tModel = t;   // the four-wide walk keeps its closest hit in maxT alone
        acceptHit(t, u, v, WIDE ? (uint32_t)__float_as_int(q2.w) : ti);   // wide records carry the index into tris
        return false;
    };

    // The same test on the two triangles of a TriPair at once; the second is taken up after the first, as
    // two consecutive triangles of a leaf are.
    auto testPair = [&](const float4v &q0, const float4v &q1, const float4v &q2, const float4v &q3, const float4v &q4) {
        const float2v rdx = {r.rd.x, r.rd.x}, rdy = {r.rd.y, r.rd.y}, rdz = {r.rd.z, r.rd.z};
        // ao = ro - A (the vertex is not needed afterwards)
        const float2v aox = (float2v){r.ro.x, r.ro.x} - (float2v){q0.x, q0.y}, aoy = (float2v){r.ro.y, r.ro.y} - (float2v){q0.z, q0.w},
                      aoz = (float2v){r.ro.z, r.ro.z} - (float2v){q1.x, q1.y};
        const float2v abx = {q1.z, q1.w}, aby = {q2.x, q2.y}, abz = {q2.z, q2.w};
        const float2v acx = {q3.x, q3.y}, acy = {q3.z, q3.w}, acz = {q4.x, q4.y};
        // nrm = cross(ab, ac), det = -dot(rd, nrm), dot(ao, nrm)
        const float2v nx = aby * acz - acy * abz, ny = abz * acx - acz * abx, nz = abx * acy - acx * aby;
        const float2v det = -(rdx * nx + rdy * ny + rdz * nz);
        const float2v tn = aox * nx + aoy * ny + aoz * nz;
        __builtin_amdgcn_sched_barrier(0);   // keeps the normal's registers from living on beside what follows
        // dao = cross(ao, rd), dot(ac, dao), -dot(ab, dao)
        const float2v dx = aoy * rdz - rdy * aoz, dy = aoz * rdx - rdz * aox, dz = aox * rdy - rdx * aoy;
        const float2v un = acx * dx + acy * dy + acz * dz;
        const float2v vn = -(abx * dx + aby * dy + abz * dz);
        auto finish = [&](float detK, float tnK, float unK, float vnK, uint32_t identK) {
            if (COUNT && identK != kMiss) cntTris++;
            if (abs_(detK) < 1e-8f) return;       // also the empty half of a pair (all zeros)
            const float invDet = 1.0f / detK;
            const float t = tnK * invDet;
            if (t < 0.0f || t > r.maxT) return;
            const bool sameModelTie = modelHasHit() && t == r.maxT;
            const float u = unK * invDet;
            const float v = vnK * invDet;
            if (u < 0.0f || v < 0.0f || u + v > 1.0f) return;
            if (sameModelTie) {
                markTie(r);
                return;
            }
            acceptHit(t, u, v, identK);
        };
        finish(det.x, tn.x, un.x, vn.x, __float_as_uint(q4.z));
        finish(det.y, tn.y, un.y, vn.y, __float_as_uint(q4.w));
    };

    for (;;) {
        if (COUNT && active) maxSp = max(maxSp, (uint32_t)r.sp);
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!drained && nIdle >= refillLanes) {
            // `source` is the same in every lane; saying so keeps the queue header in scalar registers
            // and turns the load of its length into a scalar load.
            const int src = __builtin_amdgcn_readfirstlane(source);
            SubQueue *sq = queueOf(src);
            const uint32_t n = lengthOf(sq);
            const int leader = __ffsll((long long)idleMask) - 1;
            uint32_t base = n;
            if (dealtHands) {
                base = nextHand * 64u;
                nextHand += handStride;
                if (base >= n) drained = true;
            } else if (n != 0u) {
                if ((int)lane == leader) base = atomicAdd(&sq->cursor, (uint32_t)nIdle);
                base = __shfl(base, leader);
            }
            if (dealtHands && base >= n) {
                if (nIdle == 64) break;
                continue;
            }
            if (base >= n) {
                bool has = false;
                if ((int)lane >= firstSource && (int)lane < firstSource + numSources) {
                    SubQueue *o = queueOf((int)lane);
                    has = __hip_atomic_load(&o->cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lengthOf(o);
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
                if (COUNT) cntRefills++;
                if (!active && mine < n) {
                    // The ray's record sits at the position of its queue entry (PathBuffers::rays): nothing to look up.
                    // (The second walk's list holds record numbers.)
                    if (retraceList) r.slot = qExt[mine];
                    else r.slot = ((src < kSubQueues) ? (uint32_t)extParity * pb.Q + (uint32_t)src * pb.subCap : 2u * pb.Q + (uint32_t)(src - kSubQueues) * pb.subCap) + mine;
                    const float4 *rp = reinterpret_cast<const float4 *>(&pb.rays[r.slot]);
                    const float4 r0 = rp[0], r1 = rp[1];
                    r.ro = v3(r0.x, r0.y, r0.z);
                    r.rd = v3(r1.x, r1.y, r1.z);
                    if (WIDE) {
                        // Only box tests use inv here, and the boxes carry a margin: the one-instruction
                        // reciprocal will do.  Plane numbers are scaled by inv, so keep it finite (a
                        // direction component of zero then behaves like a very small one of the same sign).
                        r.inv = v3(__builtin_amdgcn_rcpf(r.rd.x), __builtin_amdgcn_rcpf(r.rd.y), __builtin_amdgcn_rcpf(r.rd.z));
                        r.inv = v3(clampMag_(r.inv.x, 1e18f), clampMag_(r.inv.y, 1e18f), clampMag_(r.inv.z, 1e18f));   // (not fminI_/fmaxI_: the operand is a v_rcp_f32 result)
                    } else {
                        r.inv = v3(1.0f / r.rd.x, 1.0f / r.rd.y, 1.0f / r.rd.z);
                    }
                    // (four-wide: the sign is the tie mark; no hit is nearer than a stop distance <= 0 either way)
                    r.tStop = WIDE ? (r1.w > 0.0f ? r1.w : 0.0f) : r1.w;
                    if (PRIMS && kWriteThrough) {
                        intersectPrimitives<COUNT>(sc, r, cntTris);
                        r.maxT = r.closest;
                        active = !(r.closest < r.tStop) && enterModel<WIDE, true>(sc, rootLds, r, 0);
                        if (r.bestId != kMiss) {   // the primitives' answer; accepted triangles overwrite it
                            HitRec rec;
                            rec.t = r.closest; rec.u = r.bestU; rec.v = r.bestV; rec.id = r.bestId;
                            *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                        } else if (!active) {      // (a ray that goes on writes its miss record when it ends without a hit)
                            writeMiss();
                        }
                    } else if (PRIMS) {
                        intersectPrimitives<COUNT>(sc, r, cntTris);
                        active = !(r.closest < r.tStop) && enterModel<WIDE>(sc, rootLds, r, 0);
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
                        if (kWriteThrough) r.maxT = 1e20f;   // one distance for the whole ray: it only ever shrinks
                        active = enterModel<WIDE, kWriteThrough>(sc, rootLds, r, 0);
                        // Write-through instances write the miss record when the ray ENDS without a hit (most rays hit
                        // something, and their first accepted triangle would overwrite it); the others now.
                        if (!kWriteThrough || !active) writeMiss();
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
                    // pendTri counts in the leaf order being walked; the lookup and the hit record use the index into tris
                    const uint32_t hitTri = WIDE ? (uint32_t)sc.wtris[r.pendTri].ident : r.pendTri;
                    if (!alphaCutRejects(sc, hitTri, r.ro, r.rd, r.pendT)) {
                        if (!WIDE) r.tModel = r.pendT;
                        acceptHit(r.pendT, r.pendU, r.pendV, hitTri);
                    }
                    r.resume = r.pendTri - ((~(uint32_t)r.ref) >> 2) + 1u;
                    r.pendTri = kMiss;
                }
                continue;
            }
            nActive -= nPending;
        }
        const bool innerPhase = nInner * 64 >= nActive * innerShare;
        if (COUNT) {
            cntSteps++;
            cntActive += (uint32_t)nActive;
            cntInnerSteps += innerPhase ? 1u : 0u;
        }
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
            if (COUNT) cntLeaves++;
            if (WIDE && !ALPHA) {
                // Leaves of the four-wide trees hold PAIRS of triangles with their components interleaved
                // (TriPair): one round trip fetches both, and every product and sum of the two
                // Moeller-Trumbore tests up to the divisions is one packed instruction for both triangles
                // (each component is the same IEEE operation as in the one-triangle form, in the same order).
                for (uint32_t k = 0u; k < ntri; ++k) {     // ntri counts pairs here: one, rarely two
                    const float4v *tp = reinterpret_cast<const float4v *>(&sc.wpairs[first + k]);
                    float4v q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    float4v q3 = tp[3], q4 = tp[4];
                    pinLoaded(q0, q1, q2);
                    asm volatile("" : "+v"(q3), "+v"(q4));
                    testPair(q0, q1, q2, q3, q4);
                }
            } else {
                for (uint32_t k = ALPHA ? r.resume : 0u; k < ntri; ++k) {
                    const uint32_t ti = first + k;
                    const float4 *tp = reinterpret_cast<const float4 *>(WIDE ? &sc.wtris[ti] : &sc.tris[ti]);
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
                occluded = WIDE ? (modelHasHit() && r.maxT < abs_(r.tStop)) : (r.tModel < r.tStop);
            }
        }
        if (pop) {
            if (r.sp > 0 && !occluded) {
                --r.sp;
                if (WIDE && r.sp >= STACK) r.ref = overflow[(size_t)(r.sp - STACK) * 64];
                else r.ref = *stackSlot(r.sp);
            } else {
                if (!kWriteThrough && r.mBest != kMiss) {
                    const float tHit = WIDE ? r.maxT : r.tModel;
                    r.closest = tHit;
                    if (PRIMS) {
                        r.bestId = r.mBest;
                        r.bestU = r.mU;
                        r.bestV = r.mV;
                    } else if (!kWriteThrough) {
                        HitRec rec;
                        rec.t = tHit; rec.u = r.mU; rec.v = r.mV; rec.id = r.mBest;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                    }
                }
                if (occluded || !enterModel<WIDE, kWriteThrough>(sc, rootLds, r, (r.model & 0x7fffffff) + 1)) {
                    if (PRIMS && !kWriteThrough) {
                        uint32_t id = r.bestId;
                        if (id != kMiss && (id & kPrimBit) == 0u && !(r.afterPrimitives > r.closest)) id |= kNoTexBit;
                        HitRec rec;
                        rec.t = r.closest; rec.u = r.bestU; rec.v = r.bestV; rec.id = id;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                    }
                    // (no hit in any model: every distance the wide build admits is far below the initial 1e20)
                    if (kWriteThrough && r.maxT == 1e20f) writeMiss();
                    if (WIDE && hasTie(r) && pb.retrace != nullptr) {
                        // Two triangles of one model at the closest distance: the reference's winner is
                        // the one ITS walk meets first, so this ray is traced again in that order.
                        const uint32_t at = atomicAdd(&st->retrace.count, 1u);
                        if (at < pb.retraceCap) pb.retrace[at] = r.slot;
                    }
                    active = false;
                }
            }
        }
    }

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (retraceList) {   // these rays were counted by the first walk; they get a count of their own
            const uint32_t again = lengthOf(&st->retrace);
            if (again != 0u) atomicAdd(&pb.counters[5], (unsigned long long)again);
        } else {
            unsigned long long total = 0, shadowRays = 0;
            for (int q = 0; q < kSubQueues; ++q) {
                total += (qExt ? st->ext[q].count : 0u) + st->shadow[q].count;
                shadowRays += st->shadow[q].count;
            }
            atomicAdd(&pb.counters[0], total);
            atomicAdd(&pb.counters[11], shadowRays);
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) {
            cntNodes += __shfl_down(cntNodes, off);
            cntTris += __shfl_down(cntTris, off);
            cntLeaves += __shfl_down(cntLeaves, off);
            maxSp = max(maxSp, (uint32_t)__shfl_down(maxSp, off));
        }
        if (lane == 0) {
            atomicAdd(&pb.counters[1], (unsigned long long)cntNodes);
            atomicAdd(&pb.counters[2], (unsigned long long)cntTris);
            atomicMax(&pb.counters[4], (unsigned long long)maxSp);
            atomicAdd(&pb.counters[6], (unsigned long long)cntSteps);
            atomicAdd(&pb.counters[7], (unsigned long long)cntActive);
            atomicAdd(&pb.counters[8], (unsigned long long)cntLeaves);
            atomicAdd(&pb.counters[9], (unsigned long long)cntInnerSteps);
            atomicAdd(&pb.counters[10], (unsigned long long)cntRefills);
        }
    }
}

}  // namespace
