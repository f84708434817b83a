// rt_kernels_shade.h — one bounce of the two integrators: NEE connection (k_connect) and shading,
// lobe sampling, light sampling, Russian roulette (k_shade_rgb, k_shade_spectral).
#pragma once

#include "rt_wavefront.h"

using namespace rtd;

namespace {

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
        const uint32_t entry = qIn[i];
        // Where the light's contribution goes: the record of the path's next bounce, or the finished path's radiance.
        float4 *const target = (entry & kGoesOn) ? &pb.stB[extRecord(pb, stage, sub, entry & ~kGoesOn)] : &pb.radiance[entry];
        const uint32_t slot = shadowRecord(pb, sub, i), nee = sub * pb.subCap + i;   // ray, hit and NEE term sit at the queue position
        const HitRec h = pb.hits[slot];
        const float4 c = SPECTRAL ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : pb.neeC[nee];
        // RGB: a hit nearer than distToLight - 1e-4 contributes nothing; skip its material and the
        // rest of its records (the test needs 32 of the 132 bytes a connection reads).
        if (!SPECTRAL && skipOccluded && !(h.t >= c.z - 1e-4f)) continue;
        const RayRec sr = pb.rays[slot];
        const float4 a = pb.neeA[nee], b = pb.neeB[nee];
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
                float4 sb = *target;
                V3 L = v3(sb.x, sb.y, sb.z) + rayColor * direct * wNee;
                *target = make_float4(L.x, L.y, L.z, sb.w);
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
                float4 sb = *target;
                sb.x = sb.x + throughput * direct * wNee;
                *target = sb;
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
        uint32_t path = 0;
        if (i < n) {
            path = qIn[i];
            const uint32_t rec = extRecord(pb, stage, sub, i);
            const RayRec ray = pb.rays[rec];
            const HitRec h = pb.hits[rec];
            // A camera path starts with throughput and previous pdf one: k_generate does not store them.
            const float4 sA = stage == 0 ? make_float4(1.0f, 1.0f, 1.0f, 1.0f) : pb.stA[rec];
            float4 sB = stage == 0 ? make_float4(0.0f, 0.0f, 0.0f, ray.pad0) : pb.stB[rec];   // (a camera ray carries the seed)
            V3 rayColor = v3(sA.x, sA.y, sA.z);
            float prevBrdfPdf = sA.w;
            V3 incoming = v3(sB.x, sB.y, sB.z);
            uint32_t seed = __float_as_uint(sB.w);
            const V3 ro = v3(ray.ox, ray.oy, ray.oz), rd = v3(ray.dx, ray.dy, ray.dz);

            if (h.id == kMiss) {                                    // sky, :338-344
                float t = rd.y * 0.5f + 0.5f;
                V3 env = (1.0f - t) * v3s(1.0f) + t * fp.sky;
                if (dot(fp.sky, fp.sky) > 0.0f) incoming = incoming + env * rayColor;
                pb.radiance[path] = make_float4(incoming.x, incoming.y, incoming.z, 0.0f);   // the path ends here
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
                    pb.radiance[path] = make_float4(incoming.x, incoming.y, incoming.z, 0.0f);   // an emitter ends the path
                } else {
                    SurfaceRGB surf = surfaceOf(mat);
                    surf.roughness = fmax_(surf.roughness, 0.0005f);
                    const LobeProbs lp = lobeProbabilities(mat);

                    V3 L;
                    const float Xi = rand_float(seed);
                    float diff = 0.0f, spec = 0.0f, sss = 0.0f;
                    bool trans = false;
                    uint32_t *shadowEntry = nullptr;   // this bounce's shadow-queue entry, if it sends a shadow ray
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
                                // (appended here, inside the branch: the lanes that send a shadow ray share one atomic, and
                                //  the records go to the entry's position without staying in registers)
                                const uint32_t at = wavePush(true, qShadowOut, &next->shadow[sub].count, path);
                                shadowEntry = qShadowOut + at;
                                const uint32_t nee = sub * pb.subCap + at;
                                pb.rays[shadowRecord(pb, sub, at)] = sr;
                                pb.neeA[nee] = make_float4(brdfDirect.x, brdfDirect.y, brdfDirect.z, cosTheta);
                                pb.neeB[nee] = make_float4(rayColor.x, rayColor.y, rayColor.z, pdfArea);
                                pb.neeC[nee] = make_float4(dot(toLight, toLight), pdfBrdfLd, distToLight, 0.0f);
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
                        const uint32_t at = wavePush(true, qOut, &next->ext[sub].count, path);
                        const uint32_t to = extRecord(pb, stage + 1, sub, at);
                        pb.rays[to] = nr;
                        pb.stA[to] = make_float4(rayColor.x, rayColor.y, rayColor.z, prevBrdfPdf);
                        pb.stB[to] = make_float4(incoming.x, incoming.y, incoming.z, __uint_as_float(seed));
                        if (shadowEntry != nullptr) *shadowEntry = kGoesOn | at;   // the light's contribution follows the path
                    } else {
                        // The path ends: its radiance goes under its number (the shadow ray it may have sent adds to it there).
                        pb.radiance[path] = make_float4(incoming.x, incoming.y, incoming.z, 0.0f);
                    }
                }
            }
        }
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
        uint32_t path = 0;
        if (i < n) {
            path = qIn[i];
            const uint32_t rec = extRecord(pb, stage, sub, i);
            const RayRec ray = pb.rays[rec];
            const HitRec h = pb.hits[rec];
            const float4 sA = pb.stA[rec];
            float4 sB = stage == 0 ? make_float4(0.0f, 0.0f, 0.0f, ray.pad0) : pb.stB[rec];   // (a camera ray carries the seed)
            float throughput = sA.x, prevBrdfPdf = sA.y;
            const float lambda = sA.z;
            float radiance = sB.x;
            uint32_t seed = __float_as_uint(sB.w);
            const V3 ro = v3(ray.ox, ray.oy, ray.oz), rd = v3(ray.dx, ray.dy, ray.dz);

            if (h.id == kMiss) {                                    // :1061-1069
                float t = rd.y * 0.5f + 0.5f;
                V3 env = (1.0f - t) * v3s(1.0f) + t * fp.sky;
                if (dot(fp.sky, fp.sky) > 0.0f) radiance += reflectanceAt(lambda, env) * throughput;
                pb.radiance[path] = make_float4(radiance, lambda, 0.0f, 0.0f);   // the path ends here
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
                    pb.radiance[path] = make_float4(radiance, lambda, 0.0f, 0.0f);   // an emitter ends the path
                } else {
                    if (!info.front_face) N = -N;
                    const float roughness = fmax_(mat.roughness, 0.001f);
                    const LobeProbs lp = lobeProbabilities(mat);

                    V3 L;
                    const float Xi = rand_float(seed);
                    float diff = 0.0f, spec = 0.0f, sss = 0.0f;
                    bool trans = false;
                    uint32_t *shadowEntry = nullptr;   // this bounce's shadow-queue entry, if it sends a shadow ray
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
                            const uint32_t at = wavePush(true, qShadowOut, &next->shadow[sub].count, path);
                            shadowEntry = qShadowOut + at;
                            const uint32_t nee = sub * pb.subCap + at;
                            pb.rays[shadowRecord(pb, sub, at)] = sr;
                            pb.neeA[nee] = make_float4(brdfDirect, cosTheta, pdf, dot(toLight, toLight));
                            pb.neeB[nee] = make_float4(pdfBrdfLd, distToLight, throughput, lambda);
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
                        const uint32_t at = wavePush(true, qOut, &next->ext[sub].count, path);
                        const uint32_t to = extRecord(pb, stage + 1, sub, at);
                        pb.rays[to] = nr;
                        pb.stA[to] = make_float4(throughput, prevBrdfPdf, lambda, 0.0f);
                        sB.x = radiance;
                        sB.w = __uint_as_float(seed);
                        pb.stB[to] = sB;
                        if (shadowEntry != nullptr) *shadowEntry = kGoesOn | at;   // the light's contribution follows the path
                    } else {
                        // The path ends: radiance and wavelength go under its number (k_connect adds the last shadow ray's
                        // share there; k_resolve and the next hero-wavelength pass read it).
                        pb.radiance[path] = make_float4(radiance, lambda, 0.0f, 0.0f);
                    }
                }
            }
        }
    }
    if (countOn) addShadedHits(pb, shaded);
}

}  // namespace
