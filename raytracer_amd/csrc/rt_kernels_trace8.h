// rt_kernels_trace8.h — k_trace8: the persistent traversal kernel over the eight-wide trees
// (RT_OPT_WIDE_BVH 4).  Same contract as the four-wide instances of k_trace: every ray's closest hit,
// the later model winning a tie between models (src/RayScene.cpp:405-406, :535), and a ray that met
// two triangles of ONE model at its closest distance marked and appended to the stage's re-trace list,
// so that the reference-order kernel decides the winner as the reference's own walk would.
//
// What differs is the walk.  A node holds eight child boxes (Wide8Node, 80 bytes, five loads in one
// round trip).  Opening it yields ONE 32-bit hit word: a bit per triangle of the node's leaf children
// that the ray's box test reached (bits 0..23) and a bit per inner child (bits 24..31), the latter
// already permuted by the ray's direction octant so that the lowest set bit is the child nearest the
// ray's origin side.  Nothing is sorted and no distance is kept.  A lane therefore carries two small
// work lists: the triangle bits of the node it opened last and a group of inner children still to
// visit (first child number, inner-slot mask, hit bits).  Triangles go first; when a lane descends
// into a child while its group still has bits, the group (8 bytes) goes on the lane's stack in LDS:
// one entry per tree level at most, where the four-wide walk stacks up to three.
#pragma once

#include "rt_kernels_trace.h"

namespace {

#ifndef RT_W8
#define RT_W8 7          // waves/SIMD the eight-wide instances are compiled for
#endif
#ifndef RT_W8STACK
#define RT_W8STACK 10    // stack entries (8 bytes each) per lane in LDS; deeper entries go to memory
#endif
constexpr int kWide8Stack = RT_W8STACK;

typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

template <int K> __device__ __forceinline__ uint32_t byteOf(uint32_t w) { return (w >> (8 * K)) & 0xffu; }

// Adds child K's bits to the hit word when the ray's box test reaches it.  near*/far* hold the plane
// numbers the ray meets first / last on each axis, (s, b) map a plane number to the ray parameter.
template <int K>
__device__ __forceinline__ uint32_t wide8Child(uint32_t hits, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz,
                                               float2v sx, float2v sy, float2v sz, float2v bx, float2v by, float2v bz, float maxT,
                                               uint32_t count4, uint32_t place4) {
    const float2v tx = __builtin_elementwise_fma((float2v){byteAsFloat<K>(nx), byteAsFloat<K>(fx)}, sx, bx);
    const float2v ty = __builtin_elementwise_fma((float2v){byteAsFloat<K>(ny), byteAsFloat<K>(fy)}, sy, by);
    const float2v tz = __builtin_elementwise_fma((float2v){byteAsFloat<K>(nz), byteAsFloat<K>(fz)}, sz, bz);
    // The interval [max(near, 0), min(far, maxT)] is not empty: the box test of the four-wide walk
    // (near <= maxT, far >= near, far > 0) except that a box ending exactly at the origin passes too.
    const float nearD = fmaxI_(fmax3_(tx.x, ty.x, tz.x), 0.0f);
    const float farD = fminI_(fmin3_(tx.y, ty.y, tz.y), maxT);
    const uint32_t bits = byteOf<K>(count4) << byteOf<K>(place4);
    return nearD <= farD ? (hits | bits) : hits;
}

struct LaneRay8 {
    V3 ro, rd, inv;
    float closest, afterPrimitives, maxT;
    float tStop;             // as LaneRay: magnitude = stop distance of an occluded shadow ray, sign = tie mark
    float bestU, bestV, mU, mV;
    uint32_t bestId, mBest, slot;
    int32_t model, sp;
    uint32_t oct4;           // direction octant (bit a set: direction negative on axis a) in every byte
    uint32_t nbase, ngroup;  // inner children still to visit: first child number; (inner-slot mask << 8) | hit bits by visiting order
    uint32_t tbase, tbits;   // triangles still to test: wtris[tbase + bit]
};

template <bool COUNT, bool PRIMS, bool ROOTS>
__global__ __launch_bounds__(kBlock, (COUNT && kCountingWaves > 0) ? kCountingWaves : RT_W8) void k_trace8(SceneView sc, PathBuffers pb, int extParity, int stage, int refillLanes, int innerShare) {
    __shared__ uint2v stackLds[kBlock / 64][kWide8Stack][64];
    __shared__ int32_t rootTable[ROOTS ? kRootCache : 1];
    const uint32_t lane = laneId();
    if (ROOTS && (int)threadIdx.x < kRootCache) {
        int32_t root = kNoRoot;
        if ((int)threadIdx.x < sc.modelsCount) {
            const ModelDesc &md = sc.modelDesc[threadIdx.x];
            if (md.nodesCount > 0) root = md.wideRootRef;
        }
        rootTable[threadIdx.x] = root;
    }
    if (ROOTS) __syncthreads();

    Stage *st = &pb.stages[stage];
    const uint32_t *qExt = extParity >= 0 ? pb.qExt[extParity] : nullptr;
    const int firstSource = qExt ? 0 : kSubQueues;
    const int numSources = 2 * kSubQueues - firstSource;
    auto queueOf = [&](int src) -> SubQueue * { return (src < kSubQueues) ? &st->ext[src] : &st->shadow[src - kSubQueues]; };
    const uint32_t waveId = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    int source = firstSource + (int)(waveId % (uint32_t)numSources);
    bool drained = false;

    uint32_t cntNodes = 0, cntTris = 0, maxSp = 0;
    uint32_t cntSteps = 0, cntActive = 0, cntInnerSteps = 0, cntRefills = 0, cntLeaves = 0;
    bool active = false;
    LaneRay8 r;
    r.slot = 0;
    uint2v *overflow = reinterpret_cast<uint2v *>(pb.overflow) + (size_t)waveId * pb.overflowDepth * 64 + lane;
    uint2v *const waveStack = &stackLds[0][0][0] + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (kWide8Stack * 64);
    const int onChip = (int)pb.overflowFrom;   // stack entries kept in LDS (kWide8Stack, or fewer when a test asks for it)
    auto push = [&](uint32_t base, uint32_t group) {
        const uint2v e = {base, group};
        if (r.sp < onChip) waveStack[r.sp * 64 + laneIdNow()] = e;
        else overflow[(size_t)(r.sp - onChip) * 64] = e;
        r.sp++;
    };
    // Start of a model: its root as a group of one child, or the triangles of a tree that is one leaf.
    auto enter = [&](int m) -> bool {
        for (; m < sc.modelsCount; ++m) {
            int32_t root;
            if (ROOTS && m < kRootCache) {
                root = rootTable[m];
            } else {
                const ModelDesc &md = sc.modelDesc[m];
                root = md.nodesCount > 0 ? md.wideRootRef : kNoRoot;
            }
            if (root == kNoRoot) continue;
            r.model = m;
            r.sp = 0;
            r.maxT = r.closest;
            r.mBest = kMiss;
            r.mU = 0.0f;
            r.mV = 0.0f;
            if (root >= 0) {
                r.nbase = (uint32_t)root;
                r.ngroup = 1u;        // no inner-slot mask: whatever slot bit 0 stands for, the child is nbase itself
                r.tbits = 0u;
                r.tbase = 0u;
            } else {
                const uint32_t code = ~(uint32_t)root;
                r.nbase = 0u;
                r.ngroup = 0u;
                r.tbase = code >> 2;
                r.tbits = (1u << (code & 3u)) - 1u;
            }
            return true;
        }
        return false;
    };
    auto testTriangle = [&](const float4 &q0, const float4 &q1, const float4 &q2) {
        if (COUNT) cntTris++;
        const V3 A = v3(q0.x, q0.y, q0.z), ab = v3(q1.x, q1.y, q1.z), ac = v3(q2.x, q2.y, q2.z);
        const V3 nrm = cross(ab, ac);
        const float det = -dot(r.rd, nrm);
        if (abs_(det) < 1e-8f) return;
        const V3 ao = r.ro - A;
        const V3 dao = cross(ao, r.rd);
        const float invDet = 1.0f / det;
        const float t = dot(ao, nrm) * invDet;
        if (t < 0.0f || t > r.maxT) return;
        const bool sameModelTie = r.mBest != kMiss && t == r.maxT;
        const float u = dot(ac, dao) * invDet;
        const float v = -dot(ab, dao) * invDet;
        if (u < 0.0f || v < 0.0f || u + v > 1.0f) return;
        if (sameModelTie) {
            r.tStop = __int_as_float(__float_as_int(r.tStop) | (int)0x80000000);
            return;
        }
        r.maxT = t;
        r.mBest = (uint32_t)__float_as_int(q2.w);
        r.mU = u;
        r.mV = v;
    };

    for (;;) {
        if (COUNT && active) maxSp = max(maxSp, (uint32_t)r.sp);
        const unsigned long long idleMask = __ballot(!active);
        const int nIdle = __popcll(idleMask);
        if (!drained && nIdle >= refillLanes) {
            const int src = __builtin_amdgcn_readfirstlane(source);
            SubQueue *sq = queueOf(src);
            const uint32_t n = sq->count;
            const int leader = __ffsll((long long)idleMask) - 1;
            uint32_t base = n;
            if (n != 0u) {
                if ((int)lane == leader) base = atomicAdd(&sq->cursor, (uint32_t)nIdle);
                base = __shfl(base, leader);
            }
            if (base >= n) {
                bool has = false;
                if ((int)lane >= firstSource && (int)lane < firstSource + numSources) {
                    SubQueue *o = queueOf((int)lane);
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
                if (COUNT) cntRefills++;
                if (!active && mine < n) {
                    // The ray's record sits at the position of its queue entry (PathBuffers::rays): nothing to look up.
                    r.slot = ((src < kSubQueues) ? (uint32_t)extParity * pb.Q + (uint32_t)src * pb.subCap : 2u * pb.Q + (uint32_t)(src - kSubQueues) * pb.subCap) + mine;
                    const float4 *rp = reinterpret_cast<const float4 *>(&pb.rays[r.slot]);
                    const float4 r0 = rp[0], r1 = rp[1];
                    r.ro = v3(r0.x, r0.y, r0.z);
                    r.rd = v3(r1.x, r1.y, r1.z);
                    // Box tests only (the boxes carry a margin): the one-instruction reciprocal, kept finite.
                    r.inv = v3(__builtin_amdgcn_rcpf(r.rd.x), __builtin_amdgcn_rcpf(r.rd.y), __builtin_amdgcn_rcpf(r.rd.z));
                    r.inv = v3(clampMag_(r.inv.x, 1e18f), clampMag_(r.inv.y, 1e18f), clampMag_(r.inv.z, 1e18f));   // (not fminI_/fmaxI_: the operand is a v_rcp_f32 result)
                    r.oct4 = ((r.inv.x < 0.0f ? 1u : 0u) | (r.inv.y < 0.0f ? 2u : 0u) | (r.inv.z < 0.0f ? 4u : 0u)) * 0x01010101u;
                    r.tStop = r1.w > 0.0f ? r1.w : 0.0f;
                    if (PRIMS) {
                        LaneRay tmp;
                        tmp.ro = r.ro;
                        tmp.rd = r.rd;
                        intersectPrimitives<COUNT>(sc, tmp, cntTris);
                        r.closest = tmp.closest;
                        r.afterPrimitives = tmp.afterPrimitives;
                        r.bestId = tmp.bestId;
                        r.bestU = tmp.bestU;
                        r.bestV = tmp.bestV;
                        active = !(r.closest < r.tStop) && enter(0);
                        if (!active) {
                            HitRec rec;
                            rec.t = r.closest; rec.u = r.bestU; rec.v = r.bestV; rec.id = r.bestId;
                            *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                        }
                    } else {
                        r.closest = 1e20f;
                        HitRec rec;
                        rec.t = 1e20f; rec.u = 0.0f; rec.v = 0.0f; rec.id = kMiss;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                        active = enter(0);
                    }
                }
            }
        } else if (nIdle == 64) {
            break;
        }

        // One phase per iteration for the whole wave, as in k_trace: open nodes while at least
        // innerShare/64 of the active lanes have no triangles waiting, otherwise test triangles.
        const bool onInner = active && r.tbits == 0u;
        const int nActive = 64 - __popcll(__ballot(!active));
        const int nInner = __popcll(__ballot(onInner));
        const bool innerPhase = nInner * 64 >= nActive * innerShare;
        if (COUNT) {
            cntSteps++;
            cntActive += (uint32_t)nActive;
            cntInnerSteps += innerPhase ? 1u : 0u;
        }
        bool worked = false;
        if (innerPhase) {
            if (onInner) {
                worked = true;
                if (COUNT) cntNodes++;
                // next child of the group: lowest hit bit = first in this ray's visiting order
                const uint32_t place = (uint32_t)__builtin_ctz(r.ngroup);
                const uint32_t slotNo = place ^ (r.oct4 & 7u);
                const uint32_t innerMask = r.ngroup >> 8;
                const uint32_t node = r.nbase + (uint32_t)__popc(innerMask & ((1u << slotNo) - 1u));
                r.ngroup &= r.ngroup - 1u;
                if ((r.ngroup & 0xffu) != 0u) push(r.nbase, r.ngroup);
                const uint4v *np = reinterpret_cast<const uint4v *>(&sc.w8nodes[node]);
                const uint4v n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
                const uint32_t e = n0.w;
                const float sx1 = __uint_as_float((e & 0xffu) << 23) * r.inv.x;
                const float sy1 = __uint_as_float(((e >> 8) & 0xffu) << 23) * r.inv.y;
                const float sz1 = __uint_as_float(((e >> 16) & 0xffu) << 23) * r.inv.z;
                const float bx1 = (__uint_as_float(n0.x) - r.ro.x) * r.inv.x, by1 = (__uint_as_float(n0.y) - r.ro.y) * r.inv.y, bz1 = (__uint_as_float(n0.z) - r.ro.z) * r.inv.z;
                const float2v sx = {sx1, sx1}, sy = {sy1, sy1}, sz = {sz1, sz1}, bx = {bx1, bx1}, by = {by1, by1}, bz = {bz1, bz1};
                const bool ngx = (r.oct4 & 1u) != 0u, ngy = (r.oct4 & 2u) != 0u, ngz = (r.oct4 & 4u) != 0u;
                // lo: n2.x n2.y (x) n2.z n2.w (y) n3.x n3.y (z); hi: n3.z n3.w (x) n4.x n4.y (y) n4.z n4.w (z)
                const uint32_t nxA = ngx ? n3.z : n2.x, fxA = ngx ? n2.x : n3.z, nxB = ngx ? n3.w : n2.y, fxB = ngx ? n2.y : n3.w;
                const uint32_t nyA = ngy ? n4.x : n2.z, fyA = ngy ? n2.z : n4.x, nyB = ngy ? n4.y : n2.w, fyB = ngy ? n2.w : n4.y;
                const uint32_t nzA = ngz ? n4.z : n3.x, fzA = ngz ? n3.x : n4.z, nzB = ngz ? n4.w : n3.y, fzB = ngz ? n3.y : n4.w;
                // per slot byte: how many bits (bits 5..7 of meta) and where (bits 0..4; inner slots: 24 + slot ^ octant)
                const uint32_t mA = n1.z, mB = n1.w;
                const uint32_t innerA = ((mA & (mA << 1) & 0x10101010u) >> 4) * 7u, innerB = ((mB & (mB << 1) & 0x10101010u) >> 4) * 7u;
                const uint32_t placeA = (mA ^ (r.oct4 & innerA)) & 0x1f1f1f1fu, placeB = (mB ^ (r.oct4 & innerB)) & 0x1f1f1f1fu;
                const uint32_t countA = (mA >> 5) & 0x07070707u, countB = (mB >> 5) & 0x07070707u;
                uint32_t hits = 0u;
                hits = wide8Child<0>(hits, nxA, nyA, nzA, fxA, fyA, fzA, sx, sy, sz, bx, by, bz, r.maxT, countA, placeA);
                hits = wide8Child<1>(hits, nxA, nyA, nzA, fxA, fyA, fzA, sx, sy, sz, bx, by, bz, r.maxT, countA, placeA);
                hits = wide8Child<2>(hits, nxA, nyA, nzA, fxA, fyA, fzA, sx, sy, sz, bx, by, bz, r.maxT, countA, placeA);
                hits = wide8Child<3>(hits, nxA, nyA, nzA, fxA, fyA, fzA, sx, sy, sz, bx, by, bz, r.maxT, countA, placeA);
                hits = wide8Child<0>(hits, nxB, nyB, nzB, fxB, fyB, fzB, sx, sy, sz, bx, by, bz, r.maxT, countB, placeB);
                hits = wide8Child<1>(hits, nxB, nyB, nzB, fxB, fyB, fzB, sx, sy, sz, bx, by, bz, r.maxT, countB, placeB);
                hits = wide8Child<2>(hits, nxB, nyB, nzB, fxB, fyB, fzB, sx, sy, sz, bx, by, bz, r.maxT, countB, placeB);
                hits = wide8Child<3>(hits, nxB, nyB, nzB, fxB, fyB, fzB, sx, sy, sz, bx, by, bz, r.maxT, countB, placeB);
                r.nbase = n1.x;
                r.ngroup = ((e >> 24) << 8) | (hits >> 24);
                r.tbase = n1.y;
                r.tbits = hits & 0xffffffu;
            }
        } else if (active && r.tbits != 0u) {
            worked = true;
            if (COUNT) cntLeaves++;
            // up to two triangles per step, fetched in one round trip
            const uint32_t b0 = (uint32_t)__builtin_ctz(r.tbits);
            r.tbits &= r.tbits - 1u;
            const bool two = r.tbits != 0u;
            const uint32_t b1 = two ? (uint32_t)__builtin_ctz(r.tbits) : b0;
            r.tbits &= r.tbits - 1u;   // no bits left: stays zero
            const float4v *tp0 = reinterpret_cast<const float4v *>(&sc.wtris[r.tbase + b0]);
            const float4v *tp1 = reinterpret_cast<const float4v *>(&sc.wtris[r.tbase + b1]);
            float4v a0 = tp0[0], a1 = tp0[1], a2 = tp0[2];
            float4v c0 = a0, c1 = a1, c2 = a2;
            if (two) { c0 = tp1[0]; c1 = tp1[1]; c2 = tp1[2]; }
            pinLoaded(a0, a1, a2);
            pinLoaded(c0, c1, c2);
            testTriangle(asFloat4(a0), asFloat4(a1), asFloat4(a2));
            if (two) testTriangle(asFloat4(c0), asFloat4(c1), asFloat4(c2));
        }
        // (after triangles) an occluded shadow ray is finished whatever else waits or is stacked
        const bool occluded = worked && !innerPhase && r.mBest != kMiss && r.maxT < abs_(r.tStop);
        if (worked && (occluded || r.tbits == 0u)) {
            bool endModel = occluded;
            if (!occluded && (r.ngroup & 0xffu) == 0u) {
                if (r.sp > 0) {
                    --r.sp;
                    const uint2v e = r.sp >= onChip ? overflow[(size_t)(r.sp - onChip) * 64] : waveStack[r.sp * 64 + laneIdNow()];
                    r.nbase = e.x;
                    r.ngroup = e.y;
                } else {
                    endModel = true;
                }
            }
            if (endModel) {
                if (r.mBest != kMiss) {
                    r.closest = r.maxT;
                    if (PRIMS) {
                        r.bestId = r.mBest;
                        r.bestU = r.mU;
                        r.bestV = r.mV;
                    } else {
                        HitRec rec;
                        rec.t = r.maxT; rec.u = r.mU; rec.v = r.mV; rec.id = r.mBest;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                    }
                }
                if (occluded || !enter(r.model + 1)) {
                    if (PRIMS) {
                        uint32_t id = r.bestId;
                        if (id != kMiss && (id & kPrimBit) == 0u && !(r.afterPrimitives > r.closest)) id |= kNoTexBit;
                        HitRec rec;
                        rec.t = r.closest; rec.u = r.bestU; rec.v = r.bestV; rec.id = id;
                        *reinterpret_cast<float4 *>(&pb.hits[r.slot]) = *reinterpret_cast<const float4 *>(&rec);
                    }
                    if (__float_as_int(r.tStop) < 0 && pb.retrace != nullptr) {
                        const uint32_t at = atomicAdd(&st->retrace.count, 1u);
                        if (at < pb.retraceCap) pb.retrace[at] = r.slot;
                    }
                    active = false;
                }
            }
        }
    }

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long total = 0, shadowRays = 0;
        for (int q = 0; q < kSubQueues; ++q) {
            total += (qExt ? st->ext[q].count : 0u) + st->shadow[q].count;
            shadowRays += st->shadow[q].count;
        }
        atomicAdd(&pb.counters[0], total);
        atomicAdd(&pb.counters[11], shadowRays);
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
