// oracle.cpp — CPU restatement of the reference's GLSL path tracer.
//
// TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP backend is
// compared against (tests/, the smoke check and bench.py's cpu_baseline leg).
// Nothing in the product path may call it.
//
// PARITY UNPINNED: the reference ships no tests, golden images or numeric
// fixtures, cannot be built here (it needs an OpenGL 3.3 context plus the
// absent glm/tinygltf submodules) and runs its arithmetic inside a GL driver.
// The checker therefore follows the GLSL text function by function
// (citations below are paths inside the reference tree) and is pinned only by
// analytic known answers (closed-form ray/sphere, ray/quad, PCG vectors).
//
// It reads the SAME five float streams the host serialises, one 32-bit word per
// texelFetch, exactly like the samplerBuffer decode in src/RayScene.cpp:280-334.
//
// Behaviour GLSL leaves undefined is given one definition, used by the HIP
// backend too:
//   U1  a HitInfo starts as all zeros, materialIndex = 0 (the default material);
//   U2  an emissive analytic primitive hit after the first bounce gets MIS
//       weight 1 (info.area / modelLightCounts are never written for primitives);
//   U3  top-level Triangle primitives have hasTextures = false and zero UVs;
//   U4  tangent/bitangent are valid only if the winning triangle's own UVs
//       define them; otherwise the normal map is skipped;
//   U5  min/max/clamp/normalize as pinned in rt_math.h;
//   U6  the alpha/4th-channel fetch reads the word after the rgb triple as the
//       text does (it is the next texel's red for 3-channel textures);
//   U7  operand evaluation order is left to right (the two jitter draws, the
//       light-sample draws).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no fast-math).

#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>

#include "rt_math.h"

using rtm::V3;
using rtm::v3;
using rtm::v3s;
using rtm::dot;
using rtm::cross;
using rtm::length;
using rtm::normalize;

namespace {

struct V2 { float x, y; };
static inline V2 operator*(float s, V2 a) { return V2{s * a.x, s * a.y}; }
static inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
static inline V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }

// ---- constants (src/RayTracer.cpp:29-33) -----------------------------------------
constexpr float MIN_DENOMINATOR = 1e-8f;
constexpr float PI = 3.1415926f;
constexpr float INV_PI = 1.0f / PI;
constexpr float RAD = PI / 180.0f;

// ---- uniforms / streams ---------------------------------------------------------
struct Camera {  // layout of rt_camera (include/rt_hip.h)
    float fov;
    float resolution[2];
    float position[3], forward[3], right[3], up[3];
    int32_t bounces, rayPerPixel;
};

struct Scene {
    std::vector<float> objects, materials, modelInfo, models, textures;
    int objectCount = 0, modelsCount = 0, lightSourcesCount = 0;
};

struct Counters {
    uint64_t rays = 0, nodeVisits = 0, triTests = 0, shadedHits = 0, cameraSamples = 0;
    void add(const Counters &o) {
        rays += o.rays; nodeVisits += o.nodeVisits; triTests += o.triTests; shadedHits += o.shadedHits; cameraSamples += o.cameraSamples;
    }
};

struct Ctx {
    const Scene *sc;
    Camera cam;
    V3 skyColor;
    uint32_t frameCount;
    Counters cnt;
    // Spectral integrator only: leave the light-selection probability out of the area pdf, as the
    // Spectral shader text has it (type bit 8 of orc_render); default is the RGB convention.
    bool spectralPdfAsWritten = false;
    bool heroWavelength = false;   // HERO_WAVELENGTH_ENABLE 1 (src/RayTracer.cpp:607, 1357-1364)
    bool nee = true;               // ENABLE_NEE (src/RayTracer.cpp:24, 606); false = the shaders built with 0
    bool stratified = true;        // `#if 1` of the RGB main (src/RayTracer.cpp:569); false = its #else branch
};

// ---- GLSL structs (src/RayTracer.cpp:43-88, src/RayScene.cpp:215-257) -------------
struct Ray { V3 origin, direction; };

struct MaterialTexture {
    int normalTexture = -1, baseColorTexture = -1, metallicRoughnessTexture = -1, emissiveTexture = -1, transmissionTexture = -1,
        occlusionTexture = -1;
};
struct Material {
    V3 emissionColor{0, 0, 0};
    float emissionStrength = 0;
    V3 albedo{0, 0, 0};
    float subsurface = 0, roughness = 0, metallic = 0, specular = 0, specularTint = 0, transmission = 0, ior = 0, alphaCut = 0,
          normalScale = 0, occlusionStrength = 0;
    MaterialTexture texture;
};
struct HitInfo {  // U1: zero-initialised
    V3 point{0, 0, 0}, normal{0, 0, 0}, tangent{0, 0, 0}, bitangent{0, 0, 0};
    float t = 0, area = 0;
    int modelLightCounts = 0;
    V2 uv{0, 0};
    int materialIndex = 0;
    Material mat;
    bool front_face = false;
    // Not in the reference: identity of the hit, for the primary-hit comparison.
    bool tangentValid = false;  // U4
    int idModel = -1;           // -1 none, -2 analytic primitive, >= 0 model slot
    int idIdent = -1;           // identifier index in leaf order / primitive index
    float bu = 0, bv = 0;       // barycentrics as hitTriangle computes them
};
struct TextureInfo { int width, height, channels, wrapS, wrapT, index; };
struct SphereG { float radius; V3 center; int materialIndex; };
struct QuadG { V3 q, u, v; bool cullFace; int materialIndex; };
struct TriangleG {
    V3 vertices[3], normals[3];
    V2 UVs[3];
    int materialIndex = 0;
    bool hasTextures = false;
    TriangleG() { for (int i = 0; i < 3; ++i) { vertices[i] = normals[i] = v3s(0); UVs[i] = V2{0, 0}; } }
};
struct AABBG { V3 min, max; };
struct BVHNodeG { AABBG boundingBox; int leftIndex, rightIndex; bool isLeaf; };
struct IdentifierG { int index[3]; int materialIndex; bool hasTextures; };
struct ModelG { int identifiersCount, lightSourcesCount, nodesCount, location; };

// ---- buffer decode (src/RayScene.cpp:280-334) -------------------------------------
static inline float texel(const std::vector<float> &b, int i) { return (i >= 0 && (size_t)i < b.size()) ? b[(size_t)i] : 0.0f; }
static inline float ldf(const std::vector<float> &b, int &i) { return texel(b, i++); }
static inline int ldi(const std::vector<float> &b, int &i) { return rtm::f2i(ldf(b, i)); }
static inline V2 ld2(const std::vector<float> &b, int &i) { V2 r{texel(b, i), texel(b, i + 1)}; i += 2; return r; }
static inline V3 ld3(const std::vector<float> &b, int &i) { V3 r = v3(texel(b, i), texel(b, i + 1), texel(b, i + 2)); i += 3; return r; }

static inline V3 rayAt(const Ray &r, float t) { return r.origin + t * r.direction; }

// src/RayScene.cpp:268-277
static float srgb_to_linear_c(float c) {
    if (c <= 0.04045f) return c / 12.92f;
    return rtm::pow_((c + 0.055f) / 1.055f, 2.4f);
}
static V3 srgb_to_linear(V3 s) { return v3(srgb_to_linear_c(s.x), srgb_to_linear_c(s.y), srgb_to_linear_c(s.z)); }

// src/RayScene.cpp:337-380
static TextureInfo loadTextureInfo(const Scene &sc, int at) {
    TextureInfo info;
    info.width = ldi(sc.textures, at);
    info.height = ldi(sc.textures, at);
    info.channels = ldi(sc.textures, at);
    info.wrapS = ldi(sc.textures, at);
    info.wrapT = ldi(sc.textures, at);
    info.index = at;
    return info;
}
static float wrapCoord(int mode, float x) {
    if (mode == 10497) return rtm::fract_(x);
    if (mode == 33071) return rtm::clamp_(x, 0.0f, 1.0f);
    if (mode == 33648) {
        float t = rtm::fract_(x * 0.5f) * 2.0f;
        return t > 1.0f ? 2.0f - t : t;
    }
    return x;
}
static V2 getTextureUV(const TextureInfo &info, V2 uv) {
    uv.x = wrapCoord(info.wrapS, uv.x);
    uv.y = wrapCoord(info.wrapT, uv.y);
    return V2{rtm::clamp_(uv.x, 0.0f, 0.999999f), rtm::clamp_(uv.y, 0.0f, 0.999999f)};
}
static int getTextureItemIndex(const TextureInfo &info, V2 uv) {
    return info.index + ((int)(uv.x * (float)info.width) + (int)(uv.y * (float)info.height) * info.width) * info.channels;
}

// src/RayScene.cpp:383-415
static Material loadMaterial(const Scene &sc, int materialIndex) {
    Material m;
    int at = materialIndex * (92 / 4);
    m.texture.normalTexture = ldi(sc.materials, at);
    m.texture.baseColorTexture = ldi(sc.materials, at);
    m.texture.metallicRoughnessTexture = ldi(sc.materials, at);
    m.texture.emissiveTexture = ldi(sc.materials, at);
    m.texture.transmissionTexture = ldi(sc.materials, at);
    m.texture.occlusionTexture = ldi(sc.materials, at);
    m.emissionColor = ld3(sc.materials, at);
    m.emissionStrength = ldf(sc.materials, at);
    m.albedo = ld3(sc.materials, at);
    m.subsurface = ldf(sc.materials, at);
    m.roughness = ldf(sc.materials, at);
    m.metallic = ldf(sc.materials, at);
    m.specular = ldf(sc.materials, at);
    m.specularTint = ldf(sc.materials, at);
    m.transmission = ldf(sc.materials, at);
    m.ior = ldf(sc.materials, at);
    m.alphaCut = ldf(sc.materials, at);
    m.normalScale = ldf(sc.materials, at);
    m.occlusionStrength = ldf(sc.materials, at);
    return m;
}

// src/RayScene.cpp:417-431
static float RayBoundingBoxDst(const Ray &r, const AABBG &box, float t) {
    V3 invDir = v3(1.0f / r.direction.x, 1.0f / r.direction.y, 1.0f / r.direction.z);
    V3 tMin = (box.min - r.origin) * invDir;
    V3 tMax = (box.max - r.origin) * invDir;
    V3 t1 = rtm::vmin(tMin, tMax);
    V3 t2 = rtm::vmax(tMin, tMax);
    float nearD = rtm::fmax_(rtm::fmax_(t1.x, t1.y), t1.z);
    if (nearD > t) return 1e20f;
    float farD = rtm::fmin_(rtm::fmin_(t2.x, t2.y), t2.z);
    return (farD >= nearD && farD > 0.0f) ? nearD : 1e20f;
}

// src/RayScene.cpp:441-466
static bool hitSphere(const SphereG &s, const Ray &r, float maxT, HitInfo &info) {
    V3 dir = s.center - r.origin;
    float a = dot(r.direction, r.direction);
    float h = dot(r.direction, dir);
    float c = dot(dir, dir) - s.radius * s.radius;
    float disc = h * h - a * c;
    if (disc < 0.0f) return false;
    float sqrtd = rtm::sqrt_(disc);
    float t = (h - sqrtd) / a;
    if (t <= 1e-8f || t >= maxT || t >= info.t) {
        t = (h + sqrtd) / a;
        if (t <= 1e-8f || t >= maxT || t >= info.t) return false;
    }
    info.t = t;
    info.point = rayAt(r, t);
    info.normal = normalize((info.point - s.center) / s.radius);
    info.front_face = dot(r.direction, info.normal) < 0.0f;
    return true;
}

// src/RayScene.cpp:478-508
static bool hitQuad(const QuadG &q, const Ray &r, float maxT, HitInfo &info) {
    V3 normal = cross(q.u, q.v);
    float denom = dot(normal, r.direction);
    float nn = dot(normal, normal);
    if (rtm::abs_(denom) < MIN_DENOMINATOR) return false;
    float t = dot(normal, q.q - r.origin) / denom;
    if (t < 1e-8f || t > maxT || t >= info.t) return false;
    V3 hitPos = rayAt(r, t);
    V3 rel = hitPos - q.q;
    float alpha = dot(normal, cross(rel, q.v)) / nn;
    float beta = dot(normal, cross(q.u, rel)) / nn;
    if (alpha < 0.0f || alpha > 1.0f || beta < 0.0f || beta > 1.0f) return false;
    info.t = t;
    info.point = hitPos;
    info.normal = denom < 0.0f ? normalize(normal) : -normalize(normal);
    info.front_face = dot(r.direction, info.normal) < 0.0f;
    return true;
}

// src/RayScene.cpp:525-624
static bool hitTriangle(Ctx &c, const TriangleG &tri, const Ray &r, float maxT, HitInfo &info) {
    const Scene &sc = *c.sc;
    c.cnt.triTests++;
    V3 edgeAB = tri.vertices[1] - tri.vertices[0];
    V3 edgeAC = tri.vertices[2] - tri.vertices[0];
    V3 normal = cross(edgeAB, edgeAC);

    float determinant = -dot(r.direction, normal);
    if (rtm::abs_(determinant) < 1e-8f) return false;

    V3 ao = r.origin - tri.vertices[0];
    V3 dao = cross(ao, r.direction);
    float invDet = 1.0f / determinant;

    float t = dot(ao, normal) * invDet;
    if (t < 0.0f || t > maxT || t >= info.t) return false;

    float u = dot(edgeAC, dao) * invDet;
    float v = -dot(edgeAB, dao) * invDet;
    if (u < 0.0f || v < 0.0f || u + v > 1.0f) return false;

    if (tri.hasTextures) {
        info.mat = loadMaterial(sc, tri.materialIndex);
        if (info.mat.texture.baseColorTexture != -1 && info.mat.alphaCut > 0.0f) {
            V3 e0 = tri.vertices[1] - tri.vertices[0];
            V3 e1 = tri.vertices[2] - tri.vertices[0];
            V3 vp = rayAt(r, t) - tri.vertices[0];
            float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1), d20 = dot(vp, e0), d21 = dot(vp, e1);
            float denom = d00 * d11 - d01 * d01;
            float bv = (d11 * d20 - d01 * d21) / denom;
            float bw = (d00 * d21 - d01 * d20) / denom;
            float bu = 1.0f - bv - bw;
            info.uv = bu * tri.UVs[0] + bv * tri.UVs[1] + bw * tri.UVs[2];

            TextureInfo texInfo = loadTextureInfo(sc, info.mat.texture.baseColorTexture);
            V2 uv = getTextureUV(texInfo, info.uv);
            info.mat.texture.baseColorTexture = getTextureItemIndex(texInfo, uv);
            info.mat.albedo = ld3(sc.textures, info.mat.texture.baseColorTexture);
            float a = ldf(sc.textures, info.mat.texture.baseColorTexture);
            if (a < info.mat.alphaCut) return false;
        }
    }

    info.t = t;
    info.point = rayAt(r, t);
    info.bu = u;
    info.bv = v;

    if (dot(tri.normals[0], tri.normals[0]) > 0.0f) {
        float w = 1.0f - u - v;
        info.normal = normalize(tri.normals[0] * w + tri.normals[1] * u + tri.normals[2] * v);
    } else {
        info.normal = normalize(normal);
    }
    info.front_face = dot(r.direction, info.normal) < 0.0f;

    // Tangent frame from the UVs (U4: only this triangle can validate it).
    info.tangentValid = false;
    V2 dUV1 = tri.UVs[1] - tri.UVs[0];
    V2 dUV2 = tri.UVs[2] - tri.UVs[0];
    float uvDet = dUV1.x * dUV2.y - dUV2.x * dUV1.y;
    if (rtm::abs_(uvDet) > MIN_DENOMINATOR) {
        float f = 1.0f / uvDet;
        V3 tangent = v3(f * (dUV2.y * edgeAB.x - dUV1.y * edgeAC.x), f * (dUV2.y * edgeAB.y - dUV1.y * edgeAC.y),
                        f * (dUV2.y * edgeAB.z - dUV1.y * edgeAC.z));
        tangent = tangent - dot(tangent, info.normal) * info.normal;
        if (length(tangent) > MIN_DENOMINATOR) {
            info.tangent = normalize(tangent);
            info.bitangent = cross(info.normal, info.tangent);
            info.tangentValid = true;
        }
    }
    return true;
}

// src/RayScene.cpp:627-672
static ModelG loadModel(const Scene &sc, int objectIndex) {
    ModelG m;
    int at = objectIndex * 4;
    m.identifiersCount = ldi(sc.modelInfo, at);
    m.lightSourcesCount = ldi(sc.modelInfo, at);
    m.nodesCount = ldi(sc.modelInfo, at);
    m.location = ldi(sc.modelInfo, at);
    return m;
}
static BVHNodeG loadBVHNodeAt(const Scene &sc, int at) {
    BVHNodeG n;
    n.boundingBox.min = ld3(sc.models, at);
    n.boundingBox.max = ld3(sc.models, at);
    n.leftIndex = ldi(sc.models, at);
    n.rightIndex = ldi(sc.models, at);
    n.isLeaf = ldf(sc.models, at) != 0.0f;
    return n;
}
static void loadVertexByIndex(const Scene &sc, int offset, int index, V3 &position, V3 &normal, V2 &uv) {
    int at = offset + index * 8;
    position = ld3(sc.models, at);
    normal = ld3(sc.models, at);
    uv = ld2(sc.models, at);
}
static void loadTriangleByIndex(const Scene &sc, int offset, const int index[3], TriangleG &tri) {
    for (int i = 0; i < 3; ++i) loadVertexByIndex(sc, offset, index[i], tri.vertices[i], tri.normals[i], tri.UVs[i]);
}
static IdentifierG loadIdentifier(const Scene &sc, int at) {
    IdentifierG id;
    id.index[0] = ldi(sc.models, at);
    id.index[1] = ldi(sc.models, at);
    id.index[2] = ldi(sc.models, at);
    id.materialIndex = ldi(sc.models, at);
    id.hasTextures = ldf(sc.models, at) != 0.0f;
    return id;
}
constexpr int identifierFloatSize = 5;
constexpr int nodeFloatSize = 9;

static float triArea(const TriangleG &tri) {
    return 0.5f * length(cross(tri.vertices[1] - tri.vertices[0], tri.vertices[2] - tri.vertices[0]));
}

// src/RayScene.cpp:681-736
static bool hitModel(Ctx &c, int modelSlot, const ModelG &model, const Ray &r, float maxT, HitInfo &info, int location, TriangleG &triangle) {
    const Scene &sc = *c.sc;
    int stack[32];
    int stackIndex = 0;
    stack[stackIndex++] = 0;

    HitInfo hInfo;
    hInfo.t = 1e20f;

    while (stackIndex > 0) {
        int nodeIndex = stack[--stackIndex];
        BVHNodeG node = loadBVHNodeAt(sc, location + nodeIndex * nodeFloatSize);

        if (node.isLeaf) {
            for (int offset = node.leftIndex; offset < node.rightIndex; ++offset) {
                int index = location + model.nodesCount * nodeFloatSize + offset * identifierFloatSize;
                TriangleG tri;
                IdentifierG iden = loadIdentifier(sc, index);
                tri.materialIndex = iden.materialIndex;
                tri.hasTextures = iden.hasTextures;
                loadTriangleByIndex(sc, location + model.nodesCount * nodeFloatSize + model.identifiersCount * identifierFloatSize, iden.index, tri);
                if (hitTriangle(c, tri, r, maxT, hInfo)) {
                    maxT = hInfo.t;
                    hInfo.materialIndex = tri.materialIndex;
                    triangle = tri;
                    hInfo.area = triArea(tri);
                    hInfo.modelLightCounts = model.lightSourcesCount;
                    hInfo.idModel = modelSlot;
                    hInfo.idIdent = offset;
                }
            }
            continue;
        }
        c.cnt.nodeVisits++;

        BVHNodeG leftNode = loadBVHNodeAt(sc, location + node.leftIndex * nodeFloatSize);
        BVHNodeG rightNode = loadBVHNodeAt(sc, location + node.rightIndex * nodeFloatSize);
        float leftDst = RayBoundingBoxDst(r, leftNode.boundingBox, hInfo.t);
        float rightDst = RayBoundingBoxDst(r, rightNode.boundingBox, hInfo.t);

        if (leftDst < rightDst) {
            if (rightDst < hInfo.t && stackIndex < 32) stack[stackIndex++] = node.rightIndex;
            if (leftDst < hInfo.t && stackIndex < 32) stack[stackIndex++] = node.leftIndex;
        } else {
            if (leftDst < hInfo.t && stackIndex < 32) stack[stackIndex++] = node.leftIndex;
            if (rightDst < hInfo.t && stackIndex < 32) stack[stackIndex++] = node.rightIndex;
        }
    }
    info = hInfo;
    return info.t < 1e20f;
}

static int sampleTextureIndex(const Scene &sc, int &ref, V2 uvIn) {
    TextureInfo texInfo = loadTextureInfo(sc, ref);
    V2 uv = getTextureUV(texInfo, uvIn);
    ref = getTextureItemIndex(texInfo, uv);
    return ref;
}

// src/RayScene.cpp:738-845
static void hitModels(Ctx &c, const Ray &r, HitInfo &track) {
    const Scene &sc = *c.sc;
    HitInfo tmp = track;
    float closest = tmp.t;
    float startClosest = closest;

    TriangleG tri;
    tri.hasTextures = false;
    for (int i = 0; i < sc.modelsCount; ++i) {
        ModelG model = loadModel(sc, i);
        model.location += model.lightSourcesCount * identifierFloatSize;
        bool hitted = hitModel(c, i, model, r, closest, tmp, model.location, tri);
        if (hitted) {
            closest = tmp.t;
            track = tmp;
        }
    }

    track.mat = loadMaterial(sc, track.materialIndex);
    if (tri.hasTextures && startClosest > closest) {
        V3 e0 = tri.vertices[1] - tri.vertices[0];
        V3 e1 = tri.vertices[2] - tri.vertices[0];
        V3 vp = rayAt(r, track.t) - tri.vertices[0];
        float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1), d20 = dot(vp, e0), d21 = dot(vp, e1);
        float denom = d00 * d11 - d01 * d01;
        float v = (d11 * d20 - d01 * d21) / denom;
        float w = (d00 * d21 - d01 * d20) / denom;
        float u = 1.0f - v - w;
        track.uv = u * tri.UVs[0] + v * tri.UVs[1] + w * tri.UVs[2];
        Material &m = track.mat;

        if (m.texture.baseColorTexture != -1) {
            sampleTextureIndex(sc, m.texture.baseColorTexture, track.uv);
            m.albedo = ld3(sc.textures, m.texture.baseColorTexture);
            m.albedo = srgb_to_linear(m.albedo);
            float a = ldf(sc.textures, m.texture.baseColorTexture);
            m.transmission *= 1.0f - a;
        }
        if (m.texture.metallicRoughnessTexture != -1) {
            sampleTextureIndex(sc, m.texture.metallicRoughnessTexture, track.uv);
            V3 mr = ld3(sc.textures, m.texture.metallicRoughnessTexture);
            m.roughness *= mr.y;
            m.metallic *= mr.z;
        }
        if (m.texture.normalTexture != -1) {
            sampleTextureIndex(sc, m.texture.normalTexture, track.uv);
            V3 tn = ld3(sc.textures, m.texture.normalTexture);
            if (track.tangentValid) {  // U4
                tn = tn * 2.0f - v3s(1.0f);
                tn.x *= m.normalScale;
                tn.y *= m.normalScale;
                tn = normalize(tn);
                track.normal = normalize(track.tangent * tn.x + track.bitangent * tn.y + track.normal * tn.z);
                track.front_face = dot(r.direction, track.normal) < 0.0f;
            }
        }
        if (m.texture.emissiveTexture != -1) {
            sampleTextureIndex(sc, m.texture.emissiveTexture, track.uv);
            V3 tc = ld3(sc.textures, m.texture.emissiveTexture);
            m.emissionColor = m.emissionColor * tc;
        }
        if (m.texture.transmissionTexture != -1) {
            sampleTextureIndex(sc, m.texture.transmissionTexture, track.uv);
            V3 tc = ld3(sc.textures, m.texture.transmissionTexture);
            m.transmission *= tc.x;
        }
        if (m.texture.occlusionTexture != -1) {
            sampleTextureIndex(sc, m.texture.occlusionTexture, track.uv);
            V3 tc = ld3(sc.textures, m.texture.occlusionTexture);
            m.transmission *= 1.0f - (1.0f - tc.x) * (1.0f - m.occlusionStrength);
        }
    }
}

// src/RayScene.cpp:847-898
static void hit(Ctx &c, const Ray &r, HitInfo &track) {
    const Scene &sc = *c.sc;
    c.cnt.rays++;
    HitInfo tmp = track;
    float closest = tmp.t;
    float startClosest = closest;
    int objectIndex = 0;

    for (int i = 0; i < sc.objectCount; ++i) {
        int type = ldi(sc.objects, objectIndex);
        tmp.materialIndex = ldi(sc.objects, objectIndex);
        bool hitted = false;
        switch (type) {
            case 0: {
                SphereG s;
                s.center = ld3(sc.objects, objectIndex);
                s.radius = ldf(sc.objects, objectIndex);
                s.materialIndex = tmp.materialIndex;
                hitted = hitSphere(s, r, closest, tmp);
                break;
            }
            case 1: {
                QuadG q;
                q.q = ld3(sc.objects, objectIndex);
                q.u = ld3(sc.objects, objectIndex);
                q.v = ld3(sc.objects, objectIndex);
                q.cullFace = ldf(sc.objects, objectIndex) != 0.0f;
                if (q.cullFace && dot(r.direction, cross(q.u, q.v)) > 0.0f) break;
                q.materialIndex = tmp.materialIndex;
                hitted = hitQuad(q, r, closest, tmp);
                break;
            }
            case 2: {
                TriangleG tri;  // U3
                for (int k = 0; k < 3; ++k) tri.vertices[k] = ld3(sc.objects, objectIndex);
                for (int k = 0; k < 3; ++k) tri.normals[k] = ld3(sc.objects, objectIndex);
                tri.materialIndex = tmp.materialIndex;
                hitted = hitTriangle(c, tri, r, closest, tmp);
                break;
            }
            default: break;
        }
        if (hitted) {
            tmp.idModel = -2;
            tmp.idIdent = i;
            closest = tmp.t;
            track = tmp;
        }
    }
    if (startClosest > closest) track.mat = loadMaterial(sc, track.materialIndex);
    hitModels(c, r, track);
    if (track.t < 1e20f) c.cnt.shadedHits++;
}

// src/RayScene.cpp:900-935
static V3 sampleRandomPointFromLightSouces(Ctx &c, uint32_t &seed, float &area, float &lightTriangleCount) {
    const Scene &sc = *c.sc;
    if (sc.lightSourcesCount == 0) {
        area = 0.0f;
        lightTriangleCount = 0.0f;
        return v3s(0.0f);
    }
    int randLightIndex = (int)(rtm::rand_u32(seed) % (uint32_t)sc.lightSourcesCount);
    ModelG model = loadModel(sc, randLightIndex);
    lightTriangleCount = (float)model.lightSourcesCount;
    int randomIdentifierIndex = model.location + (int)(rtm::rand_u32(seed) % (uint32_t)model.lightSourcesCount) * identifierFloatSize;
    IdentifierG iden = loadIdentifier(sc, randomIdentifierIndex);
    TriangleG tri;
    loadTriangleByIndex(sc, model.location + model.lightSourcesCount * identifierFloatSize + model.nodesCount * nodeFloatSize +
                                model.identifiersCount * identifierFloatSize,
                        iden.index, tri);
    float r1 = rtm::rand_float(seed);
    float r2 = rtm::rand_float(seed);
    if (r1 + r2 > 1.0f) {
        r1 = 1.0f - r1;
        r2 = 1.0f - r2;
    }
    area = triArea(tri);
    return r1 * tri.vertices[0] + r2 * tri.vertices[1] + (1.0f - r1 - r2) * tri.vertices[2];
}

// ---- BSDF toolbox, RGB (src/RayTracer.cpp:119-325) ---------------------------------
static V3 reflect(V3 v, V3 n) { return v - dot(v, n) * n * 2.0f; }
static V3 perpendicular(V3 v) { return (rtm::abs_(v.x) > 0.9f) ? v3(0, 1, 0) : v3(1, 0, 0); }

static V3 sampleHemisphereCosine(V3 N, uint32_t &seed) {
    float r1 = rtm::rand_float(seed);
    float r2 = rtm::rand_float(seed);
    float phi = 2.0f * PI * r1;
    float cosTheta = rtm::sqrt_(1.0f - r2);
    float sinTheta = rtm::sqrt_(r2);
    V3 local = v3(rtm::cos_(phi) * sinTheta, rtm::sin_(phi) * sinTheta, cosTheta);
    V3 T = normalize(cross(N, perpendicular(N)));
    V3 B = normalize(cross(N, T));
    return T * local.x + B * local.y + N * local.z;
}

static V3 sampleGGXVNDF_H(V3 N, V3 V, float roughness, uint32_t &seed) {
    float a = roughness * roughness;
    float r1 = rtm::rand_float(seed);
    float r2 = rtm::rand_float(seed);
    V3 T = normalize(cross(N, perpendicular(N)));
    V3 B = normalize(cross(N, T));
    V3 Vh = normalize(v3(a * dot(V, T), a * dot(V, B), dot(V, N)));
    V3 T1 = (Vh.z < 0.9999f) ? normalize(cross(v3(0, 0, 1), Vh)) : v3(1, 0, 0);
    V3 T2 = cross(Vh, T1);
    float r = rtm::sqrt_(r1);
    float phi = 2.0f * PI * r2;
    float t1 = r * rtm::cos_(phi);
    float t2 = r * rtm::sin_(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * rtm::sqrt_(1.0f - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + rtm::sqrt_(rtm::fmax_(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    V3 Hh = normalize(a * Nh.x * T1 + a * Nh.y * T2 + Nh.z * Vh);
    return normalize(T * Hh.x + B * Hh.y + N * Hh.z);
}

static V3 sampleGGXVNDF(V3 N, V3 V, float roughness, uint32_t &seed) {
    V3 H = sampleGGXVNDF_H(N, V, roughness, seed);
    V3 L = reflect(-V, H);
    if (dot(N, L) <= 0.0f) return v3s(0.0f);
    return L;
}

static V3 computeF0(const HitInfo &info) {
    float specular = rtm::clamp_(info.mat.specular, 0.0f, 1.0f);
    V3 f0 = v3s(0.16f * specular * specular);
    return rtm::vmix(f0, info.mat.albedo, info.mat.metallic);
}
static V3 fresnelSchlick3(float cosTheta, V3 F0) { return F0 + (v3s(1.0f) - F0) * rtm::pow5_(1.0f - cosTheta); }
static float NDF_GGX(float NoH, float roughness) {
    float a = roughness * roughness;
    float a2 = a * a;
    float denom = NoH * NoH * (a2 - 1.0f) + 1.0f;
    return a2 / rtm::fmax_(denom * denom, 1e-16f);
}
static float geometrySchlickGGX(float NoV, float roughness) {
    float a = roughness * roughness;
    float k = a * 0.5f;
    return NoV / rtm::fmax_(NoV * (1.0f - k) + k, MIN_DENOMINATOR);
}
static float geometrySmith(float NoV, float NoL, float roughness) {
    return geometrySchlickGGX(NoV, roughness) * geometrySchlickGGX(NoL, roughness);
}
static float specularPdf(float NoH, float NoV, float /*VoH*/, float roughness) {
    float D = NDF_GGX(NoH, roughness);
    float G1 = geometrySchlickGGX(NoV, roughness);
    return D * G1 * NoH / rtm::fmax_(4.0f * NoV, MIN_DENOMINATOR);
}
static V3 shadeSpecular(const HitInfo &info, float NoV, float NoL, float NoH, float VoH) {
    V3 F0 = computeF0(info);
    V3 F = fresnelSchlick3(VoH, F0);
    float D = NDF_GGX(NoH, info.mat.roughness);
    float G = geometrySmith(NoV, NoL, info.mat.roughness);
    return (D * G * F) / rtm::fmax_(4.0f * NoV * NoL, MIN_DENOMINATOR);
}
static V3 shadeDiffuse(const HitInfo &info, float NoL, float NoV, float VoH) {
    V3 F0 = computeF0(info);
    V3 F = fresnelSchlick3(VoH, F0);
    V3 kd = (v3s(1.0f) - F) * (1.0f - info.mat.metallic);
    float FD90 = 0.5f + 2.0f * info.mat.roughness * VoH * VoH;
    float fresnelDiffuse = (1.0f + (FD90 - 1.0f) * rtm::pow5_(1.0f - NoL)) * (1.0f + (FD90 - 1.0f) * rtm::pow5_(1.0f - NoV));
    return kd * info.mat.albedo * INV_PI * fresnelDiffuse;
}
static float diffusePdf(float NoL) { return NoL * INV_PI; }
static V3 shadeSubsurface(const HitInfo &info, float NoL, float NoV, float LoV) {
    float FL = rtm::pow5_(1.0f - NoL);
    float FV = rtm::pow5_(1.0f - NoV);
    float Fd90 = 0.5f + 2.0f * LoV * info.mat.roughness;
    float Fd = rtm::mix_(1.0f, Fd90, FL) * rtm::mix_(1.0f, Fd90, FV);
    return info.mat.albedo * Fd * INV_PI * info.mat.subsurface;
}
static V3 refract(V3 uv, V3 n, float etai_over_etat) {
    float cos_theta = rtm::fmin_(dot(-uv, n), 1.0f);
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = -rtm::sqrt_(rtm::abs_(1.0f - dot(r_out_perp, r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}
static float fresnelSchlickIor(float cosine, float reflectance_index) {
    float r0 = (1.0f - reflectance_index) / (1.0f + reflectance_index);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * rtm::pow5_(1.0f - cosine);
}

typedef V3 (*HalfVectorSampler)(V3, V3, float, uint32_t &);

static V3 sampleTransmission(V3 N, V3 V, bool front_face, const Material &mat, uint32_t &seed, HalfVectorSampler sampleH) {
    float eta = front_face ? (1.0f / mat.ior) : mat.ior;
    V3 H = sampleH(N, V, mat.roughness, seed);
    float VoH = dot(V, H);
    if (VoH < 0.0f) {
        H = N;
        VoH = dot(V, N);
    }
    float cos_theta = rtm::fmin_(VoH, 1.0f);
    float sin_theta = rtm::sqrt_(rtm::fmax_(1.0f - cos_theta * cos_theta, 0.0f));
    bool cannot_refract = eta * sin_theta > 1.0f;
    if (cannot_refract) return reflect(-V, H);
    float R = fresnelSchlickIor(cos_theta, eta);
    if (rtm::rand_float(seed) < R) return reflect(-V, H);
    return refract(-V, H, eta);
}

// ---- RGB integrator (src/RayTracer.cpp:327-541) -------------------------------------
static V3 traceColor(Ctx &c, Ray r, uint32_t &seed) {
    const Scene &sc = *c.sc;
    V3 incomingLight = v3s(0.0f);
    V3 rayColor = v3s(1.0f);
    float prevBrdfPdf = 1.0f;

    for (int i = 0; i <= c.cam.bounces; ++i) {
        HitInfo info;
        info.t = 1e20f;
        hit(c, r, info);

        if (info.t >= 1e20f) {
            float t = r.direction.y * 0.5f + 0.5f;
            V3 envColor = (1.0f - t) * v3s(1.0f) + t * c.skyColor;
            if (dot(c.skyColor, c.skyColor) > 0.0f) incomingLight = incomingLight + envColor * rayColor;
            return incomingLight;
        }

        V3 N = normalize(info.normal);
        V3 V = normalize(-r.direction);
        if (!info.front_face) N = -N;

        if (dot(info.mat.emissionColor, info.mat.emissionColor) > 0.0f && info.mat.emissionStrength > 0.0f) {
            if (i == 0 || !c.nee) {   // #else branch of ENABLE_NEE, :374-376: the same expression
                incomingLight = incomingLight + rayColor * info.mat.emissionColor * info.mat.emissionStrength;
            } else {
                float w_brdf;
                if (info.idModel == -2) {
                    w_brdf = 1.0f;  // U2
                } else {
                    float pdf_area = 1.0f / info.area / (float)sc.lightSourcesCount / (float)info.modelLightCounts;
                    float cosTheta = rtm::fmax_(dot(V, N), 0.0f);
                    float pdf_nee = pdf_area * (info.t * info.t) / rtm::fmax_(cosTheta, MIN_DENOMINATOR);
                    w_brdf = (prevBrdfPdf * prevBrdfPdf) / rtm::fmax_(prevBrdfPdf * prevBrdfPdf + pdf_nee * pdf_nee, MIN_DENOMINATOR);
                }
                incomingLight = incomingLight + rayColor * w_brdf * info.mat.emissionColor * info.mat.emissionStrength;
            }
            break;
        }

        info.mat.roughness = rtm::fmax_(info.mat.roughness, 0.0005f);

        float transmissionProb = info.mat.transmission;
        float subsurfaceProb = info.mat.subsurface * (1.0f - transmissionProb);
        float diffuseProb = (1.0f - info.mat.metallic) * (1.0f - transmissionProb);
        float specularProb = (0.5f + 0.5f * info.mat.metallic) * (1.0f - transmissionProb);
        float totalProb = subsurfaceProb + diffuseProb + specularProb + transmissionProb;
        subsurfaceProb /= totalProb;
        diffuseProb /= totalProb;
        specularProb /= totalProb;
        transmissionProb /= totalProb;

        V3 L;
        float Xi = rtm::rand_float(seed);
        float diff = 0, spec = 0, subsurface = 0, trans = 0;
        if (Xi < diffuseProb) {
            L = sampleHemisphereCosine(N, seed);
            diff = 1;
        } else if (Xi < diffuseProb + specularProb) {
            L = sampleGGXVNDF(N, V, info.mat.roughness, seed);
            spec = 1;
        } else if (Xi < diffuseProb + specularProb + transmissionProb) {
            L = sampleTransmission(N, V, info.front_face, info.mat, seed, sampleGGXVNDF_H);
            trans = 1;
        } else {
            L = sampleHemisphereCosine(N, seed);
            subsurface = 1;
        }
        L = normalize(L);

        V3 H = normalize(V + L);
        float NoV = rtm::clamp_(dot(N, V), 0.0f, 1.0f);
        float NoL = rtm::clamp_(dot(N, L), 0.0f, 1.0f);
        float NoH = rtm::clamp_(dot(N, H), 0.0f, 1.0f);
        float VoH = rtm::clamp_(dot(V, H), 0.0f, 1.0f);
        float LoV = rtm::clamp_(dot(L, V), 0.0f, 1.0f);

        if (trans == 0 && c.nee) {   // the block is compiled out with ENABLE_NEE 0, :420-481
            float area, lightTriangleCount;
            V3 p = sampleRandomPointFromLightSouces(c, seed, area, lightTriangleCount);
            if (area > 0.0f) {
                Ray sr;
                sr.origin = info.point + N * 0.001f;
                V3 toLight = p - sr.origin;
                float distToLight = length(toLight);
                sr.direction = toLight / distToLight;
                float cosTheta = dot(N, sr.direction);
                if (cosTheta > 0.0f) {
                    HitInfo s_info;
                    s_info.t = 1e20f;
                    hit(c, sr, s_info);
                    if (s_info.mat.emissionStrength > 0.0f && s_info.t >= distToLight - 1e-4f) {
                        if (!s_info.front_face) s_info.normal = -s_info.normal;
                        s_info.normal = normalize(s_info.normal);
                        float cosThetaL = rtm::fmax_(dot(-sr.direction, s_info.normal), 0.0f);
                        float pdf_area = 1.0f / area / (float)sc.lightSourcesCount / lightTriangleCount;
                        float Gfactor = cosThetaL / dot(toLight, toLight);
                        float pdf_nee = pdf_area / rtm::fmax_(Gfactor, MIN_DENOMINATOR);

                        V3 Ld = sr.direction;
                        V3 Hd = normalize(V + Ld);
                        float NoLd = rtm::clamp_(dot(N, Ld), 0.0f, 1.0f);
                        float NoHd = rtm::clamp_(dot(N, Hd), 0.0f, 1.0f);
                        float VoHd = rtm::clamp_(dot(V, Hd), 0.0f, 1.0f);
                        float LoVd = rtm::clamp_(dot(Ld, V), 0.0f, 1.0f);

                        float pdf_brdf_ld = diffuseProb * diffusePdf(NoLd) + specularProb * specularPdf(NoHd, NoV, VoHd, info.mat.roughness) +
                                            subsurfaceProb * (NoLd * INV_PI);
                        float w_nee = (pdf_nee * pdf_nee) / rtm::fmax_(pdf_nee * pdf_nee + pdf_brdf_ld * pdf_brdf_ld, MIN_DENOMINATOR);

                        V3 brdf_direct = diffuseProb * shadeDiffuse(info, NoLd, NoV, VoHd) + specularProb * shadeSpecular(info, NoV, NoLd, NoHd, VoHd) +
                                         subsurfaceProb * shadeSubsurface(info, NoLd, NoV, LoVd);
                        V3 directLight = brdf_direct * s_info.mat.emissionColor * s_info.mat.emissionStrength * cosTheta * Gfactor / pdf_area;
                        incomingLight = incomingLight + rayColor * directLight * w_nee;
                    }
                }
            }
        }

        r.origin = info.point + N * 0.001f;
        r.direction = L;

        if (trans == 1) {
            r.origin = info.point + L * 0.001f;
            if (!info.front_face) {
                V3 albedo = rtm::vmax(info.mat.albedo, v3s(MIN_DENOMINATOR));
                V3 transmittance = v3(rtm::exp_(info.t * rtm::log_(albedo.x)), rtm::exp_(info.t * rtm::log_(albedo.y)), rtm::exp_(info.t * rtm::log_(albedo.z)));
                rayColor = rayColor * transmittance;
            }
            prevBrdfPdf = 1.0f;
            continue;
        }

        if (NoL < MIN_DENOMINATOR) break;

        V3 brdf_sss = shadeSubsurface(info, NoL, NoV, LoV);
        V3 brdf_spec = shadeSpecular(info, NoV, NoL, NoH, VoH);
        V3 brdf_diff = shadeDiffuse(info, NoL, NoV, VoH);

        float p_surf = 1.0f - transmissionProb;
        p_surf = (p_surf < 1e-8f) ? 0.0f : p_surf;
        float surfaceNormalization = (p_surf > 0.0f) ? 1.0f / p_surf : 1.0f;

        float pdf_diff_raw = diffusePdf(NoL) * surfaceNormalization;
        float pdf_spec_raw = specularPdf(NoH, NoV, VoH, info.mat.roughness) * surfaceNormalization;
        float pdf_sss_raw = NoL * INV_PI * surfaceNormalization;

        float pdf_used = pdf_diff_raw * diffuseProb * diff + pdf_spec_raw * specularProb * spec + pdf_sss_raw * subsurfaceProb * subsurface;
        prevBrdfPdf = pdf_diff_raw * diffuseProb + pdf_spec_raw * specularProb + pdf_sss_raw * subsurfaceProb;

        V3 brdf_total = (brdf_spec * spec + brdf_diff * diff + brdf_sss * subsurface);
        V3 contribution = (brdf_total * NoL) / rtm::fmax_(pdf_used, MIN_DENOMINATOR);
        rayColor = rayColor * contribution;

        float rrProb = rtm::fmin_(rtm::fmax_(rayColor.x, rtm::fmax_(rayColor.y, rayColor.z)), 0.95f);
        if (rtm::rand_float(seed) > rrProb) break;
        rayColor = rayColor / rrProb;
    }
    return incomingLight;
}

#include "oracle_spectral.inc"

// ---- fragment main (src/RayTracer.cpp:543-599 / 1317-1383) ----------------------------
struct PixelBasis {
    V3 uv, cameraCenter, right, up;
    float rImgX, rImgY;
};

static PixelBasis pixelBasis(const Camera &cam, int px, int py) {
    PixelBasis b;
    float imgX = cam.resolution[0], imgY = cam.resolution[1];
    b.rImgX = 1.0f / imgX;
    b.rImgY = 1.0f / imgY;
    V3 position = v3(cam.position[0], cam.position[1], cam.position[2]);
    V3 forward = v3(cam.forward[0], cam.forward[1], cam.forward[2]);
    b.right = v3(cam.right[0], cam.right[1], cam.right[2]);
    b.up = v3(cam.up[0], cam.up[1], cam.up[2]);
    V3 lookat = forward + position;
    b.cameraCenter = position;
    float viewportRatio = imgX * b.rImgY;
    float focalLength = length(lookat - b.cameraCenter);
    float viewportHeight = 2.0f * rtm::tan_(RAD * cam.fov * 0.5f) * focalLength;
    float viewportWidth = viewportHeight * viewportRatio;
    float ux = (float)px * b.rImgX * 2.0f - 1.0f;
    float uy = (float)py * b.rImgY * 2.0f - 1.0f;
    b.uv = viewportWidth * 0.5f * ux * b.right + viewportHeight * 0.5f * uy * b.up + focalLength * forward + b.cameraCenter;
    return b;
}

static Ray cameraRay(const PixelBasis &b, int i, int j, float rssq, uint32_t &seed) {
    Ray r;
    r.origin = b.cameraCenter;
    float jx = rtm::rand_float(seed);  // U7: first draw feeds the horizontal jitter
    float jy = rtm::rand_float(seed);
    V3 d = b.uv + (((float)j + jx) * rssq) * b.rImgX * b.right + (((float)i + jy) * rssq) * b.rImgY * b.up;
    r.direction = normalize(d - b.cameraCenter);
    return r;
}

// One frame's colour for one pixel (before the running mean).
static V3 framePixel(Ctx &c, int px, int py, int type) {
    PixelBasis b = pixelBasis(c.cam, px, py);
    V3 color = v3s(0.0f);
    int ssq = (int)rtm::sqrt_((float)c.cam.rayPerPixel);
    float rssq = 1.0f / (float)ssq;
    float wl_dt = WL_RANGE / (float)c.cam.rayPerPixel;
    float wl_pdf = 1.0f / WL_RANGE;
    if (type == 0 && !c.stratified) {
        // src/RayTracer.cpp:584-593: rayPerPixel unstratified samples, jitter of one pixel, plain mean.
        for (int i = 0; i < c.cam.rayPerPixel; ++i) {
            uint32_t seed = rtm::hash_seed((uint32_t)px, (uint32_t)py, c.frameCount, (uint32_t)i);
            Ray r;
            r.origin = b.cameraCenter;
            float jx = rtm::rand_float(seed);  // U7: first draw feeds the horizontal jitter
            float jy = rtm::rand_float(seed);
            V3 d = b.uv + jx * b.rImgX * b.right + jy * b.rImgY * b.up;
            r.direction = normalize(d - b.cameraCenter);
            c.cnt.cameraSamples++;
            color = color + traceColor(c, r, seed);
        }
        return color / (float)c.cam.rayPerPixel;
    }
    for (int i = 0; i < ssq; ++i) {
        for (int j = 0; j < ssq; ++j) {
            uint32_t seed = rtm::hash_seed((uint32_t)px, (uint32_t)py, c.frameCount, (uint32_t)(j + i * ssq));
            Ray r = cameraRay(b, i, j, rssq, seed);
            c.cnt.cameraSamples++;
            if (type == 0) {
                color = color + traceColor(c, r, seed);
            } else if (c.heroWavelength) {
                // src/RayTracer.cpp:1358-1364: four wavelengths per camera sample along the same camera
                // ray; the seed goes in by value (:1051), so all four start from the same state.
                float base_offset = (rtm::rand_float(seed) + (float)(i * ssq) + (float)j) * wl_dt;
                float lambdas[4], radiances[4];
                get_hero_wavelengths(base_offset, lambdas);
                for (int k = 0; k < 4; ++k) radiances[k] = traceColorWavelength(c, r, lambdas[k], seed);
                color = color + hero_wavelengths_to_rgb(lambdas, radiances, wl_pdf);
            } else {
                float lambda = (rtm::rand_float(seed) + (float)(i * ssq) + (float)j) * wl_dt + WL_MIN;
                float radiance = traceColorWavelength(c, r, lambda, seed);
                color = color + wavelength_to_xyz(lambda, radiance, wl_pdf);
            }
        }
    }
    color = color * (rssq * rssq);
    if (type != 0) {
        color = color / CIE_Y_INTEGRAL;
        color = chromatic_adapt_E_to_D65(color);
        color = xyz_to_rgb(color);
    }
    return color;
}

template <typename F> static void parallelRows(int y0, int y1, int threads, F &&fn) {
    if (threads < 1) threads = 1;
    std::atomic<int> next(y0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            for (;;) {
                int y = next.fetch_add(1);
                if (y >= y1) break;
                fn(y, t);
            }
        });
    }
    for (auto &th : pool) th.join();
}

}  // namespace

extern "C" {

struct orc_scene {
    Scene sc;
};

orc_scene *orc_scene_create(const float *objects, size_t nObjects, const float *materials, size_t nMaterials, const float *modelInfo,
                            size_t nModelInfo, const float *models, size_t nModels, const float *textures, size_t nTextures, int objectCount,
                            int modelsCount, int lightSourcesCount) {
    orc_scene *s = new orc_scene();
    auto copy = [](std::vector<float> &dst, const float *src, size_t n) {
        if (src != nullptr && n > 0) dst.assign(src, src + n);
    };
    copy(s->sc.objects, objects, nObjects);
    copy(s->sc.materials, materials, nMaterials);
    copy(s->sc.modelInfo, modelInfo, nModelInfo);
    copy(s->sc.models, models, nModels);
    copy(s->sc.textures, textures, nTextures);
    s->sc.objectCount = objectCount;
    s->sc.modelsCount = modelsCount;
    s->sc.lightSourcesCount = lightSourcesCount;
    return s;
}

void orc_scene_free(orc_scene *s) { delete s; }

// Closest hit of the sample-0 camera ray of `frame` for pixels in [x0,x1) x [y0,y1).
// Arrays are full-frame (W*H); counters = {rays, nodeVisits, triTests}.
int orc_trace_primary(const orc_scene *s, const Camera *cam, uint32_t frame, int x0, int y0, int x1, int y1, int32_t *model, int32_t *ident,
                      float *t, float *u, float *v, uint64_t counters[3], int threads) {
    if (s == nullptr || cam == nullptr) return -1;
    const int W = (int)cam->resolution[0];
    std::vector<Counters> per((size_t)std::max(threads, 1));
    parallelRows(y0, y1, threads, [&](int y, int tid) {
        Ctx c{&s->sc, *cam, v3s(0), frame, Counters()};
        int ssq = (int)rtm::sqrt_((float)cam->rayPerPixel);
        float rssq = 1.0f / (float)ssq;
        for (int x = x0; x < x1; ++x) {
            PixelBasis b = pixelBasis(*cam, x, y);
            uint32_t seed = rtm::hash_seed((uint32_t)x, (uint32_t)y, frame, 0u);
            Ray r = cameraRay(b, 0, 0, rssq, seed);
            HitInfo info;
            info.t = 1e20f;
            hit(c, r, info);
            size_t at = (size_t)y * W + x;
            model[at] = info.t < 1e20f ? info.idModel : -1;
            ident[at] = info.t < 1e20f ? info.idIdent : -1;
            t[at] = info.t;
            u[at] = info.idModel >= 0 ? info.bu : 0.0f;
            v[at] = info.idModel >= 0 ? info.bv : 0.0f;
        }
        per[(size_t)tid].add(c.cnt);
    });
    if (counters != nullptr) {
        Counters tot;
        for (auto &p : per) tot.add(p);
        counters[0] = tot.rays;
        counters[1] = tot.nodeVisits;
        counters[2] = tot.triTests;
    }
    return 0;
}

// Renders frames first_frame, first_frame+stride, ... (n_frames of them) for
// pixels in [x0,x1) x [y0,y1).  rgba_sum (W*H*4, pre-zeroed by the caller or
// carrying an earlier sum) receives the plain sum of per-frame colours, alpha =
// number of frames added.  rgba_mean, if not null, receives the reference's
// running mean (prev*(n-1)+c)/n with n = the frame number itself, which equals
// the plain mean only for first_frame = 1, stride = 1.
// stats = {rays, nodeVisits, triTests, shadedHits, cameraSamples}.
int orc_render(const orc_scene *s, const Camera *cam, const float sky[3], int typeAndFlags, uint32_t first_frame, uint32_t n_frames, uint32_t stride,
               int x0, int y0, int x1, int y1, float *rgba_sum, float *rgba_mean, uint64_t stats[5], int threads) {
    if (s == nullptr || cam == nullptr || sky == nullptr || rgba_sum == nullptr) return -1;
    const int W = (int)cam->resolution[0];
    const int type = typeAndFlags & 0xff;
    std::vector<Counters> per((size_t)std::max(threads, 1));
    parallelRows(y0, y1, threads, [&](int y, int tid) {
        Ctx c{&s->sc, *cam, v3(sky[0], sky[1], sky[2]), 0u, Counters()};
        c.spectralPdfAsWritten = (typeAndFlags & 0x100) != 0;
        c.heroWavelength = (typeAndFlags & 0x200) != 0;
        c.nee = (typeAndFlags & 0x400) == 0;
        c.stratified = (typeAndFlags & 0x800) == 0;
        for (int x = x0; x < x1; ++x) {
            size_t at = ((size_t)y * W + x) * 4;
            for (uint32_t k = 0; k < n_frames; ++k) {
                c.frameCount = first_frame + k * stride;
                V3 color = framePixel(c, x, y, type);
                rgba_sum[at + 0] += color.x;
                rgba_sum[at + 1] += color.y;
                rgba_sum[at + 2] += color.z;
                rgba_sum[at + 3] += 1.0f;
                if (rgba_mean != nullptr) {
                    float fc = (float)c.frameCount;
                    V3 prev = v3(rgba_mean[at + 0], rgba_mean[at + 1], rgba_mean[at + 2]);
                    if (c.frameCount == 1u) prev = v3s(0.0f);
                    V3 mean = (prev * (fc - 1.0f) + color) / fc;
                    rgba_mean[at + 0] = mean.x;
                    rgba_mean[at + 1] = mean.y;
                    rgba_mean[at + 2] = mean.z;
                    rgba_mean[at + 3] = 1.0f;
                }
            }
        }
        per[(size_t)tid].add(c.cnt);
    });
    if (stats != nullptr) {
        Counters tot;
        for (auto &p : per) tot.add(p);
        stats[0] = tot.rays;
        stats[1] = tot.nodeVisits;
        stats[2] = tot.triTests;
        stats[3] = tot.shadedHits;
        stats[4] = tot.cameraSamples;
    }
    return 0;
}

// Screen pass of src/main.cpp:36-69 on a mean image (rgba float, alpha ignored) -> RGBA8.
int orc_display(const float *rgba_mean, size_t pixels, uint8_t *rgba8) {
    for (size_t i = 0; i < pixels; ++i) {
        V3 c = rtm::vmax(v3(rgba_mean[i * 4], rgba_mean[i * 4 + 1], rgba_mean[i * 4 + 2]), v3s(0.0f));
        c = (c * (2.51f * c + v3s(0.03f))) / (c * (2.43f * c + v3s(0.59f)) + v3s(0.14f));
        c = rtm::vclamp(c, 0.0f, 1.0f);
        V3 o;
        if (c.x != c.x || c.y != c.y || c.z != c.z) o = v3(1.0f, 0.0f, 0.0f);
        else if (c.x < 0.0f || c.y < 0.0f || c.z < 0.0f) o = v3(0.0f, 1.0f, 0.0f);
        else if (rtm::abs_(c.x) == rtm::inf() || rtm::abs_(c.y) == rtm::inf() || rtm::abs_(c.z) == rtm::inf()) o = v3(0.0f, 0.0f, 1.0f);
        else o = v3(rtm::pow_(c.x, 1.0f / 2.2f), rtm::pow_(c.y, 1.0f / 2.2f), rtm::pow_(c.z, 1.0f / 2.2f));
        rgba8[i * 4 + 0] = (uint8_t)(o.x * 255.0f + 0.5f);
        rgba8[i * 4 + 1] = (uint8_t)(o.y * 255.0f + 0.5f);
        rgba8[i * 4 + 2] = (uint8_t)(o.z * 255.0f + 0.5f);
        rgba8[i * 4 + 3] = 255;
    }
    return 0;
}

// Small known-answer hooks.
uint32_t orc_pcg(uint32_t v) { return rtm::pcg(v); }
uint32_t orc_hash_seed(uint32_t x, uint32_t y, uint32_t frame, uint32_t sample) { return rtm::hash_seed(x, y, frame, sample); }
float orc_rand_float(uint32_t *seed) { return rtm::rand_float(*seed); }
// fn: 0 sin, 1 cos, 2 tan, 3 exp, 4 log, 5 pow(x,y), 6 min, 7 max, 8 sqrt
float orc_math(int fn, float x, float y) {
    switch (fn) {
        case 0: return rtm::sin_(x);
        case 1: return rtm::cos_(x);
        case 2: return rtm::tan_(x);
        case 3: return rtm::exp_(x);
        case 4: return rtm::log_(x);
        case 5: return rtm::pow_(x, y);
        case 6: return rtm::fmin_(x, y);
        case 7: return rtm::fmax_(x, y);
        default: return rtm::sqrt_(x);
    }
}
// Colour science of the Spectral integrator, evaluated item by item (same numbering as the
// backend's rt_eval_colour): 0 get_cie_xyz(lambda) -> xyz; 1 smits_eval(lambda, type) -> value;
// 2 get_reflectance(lambda, rgb) -> value; 3 xyz_to_rgb(chromatic_adapt_E_to_D65(xyz)) -> rgb;
// 4 the tables themselves, CIE_X, CIE_Y, CIE_Z (41 each) then SMITS_TABLE (70), per item.
int orc_eval_colour(int fn, const float *in, size_t items, float *out) {
    for (size_t i = 0; i < items; ++i) {
        switch (fn) {
            case 0: { V3 r = get_cie_xyz(in[i]); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
            case 1: out[i] = smits_eval(in[2 * i], (int)in[2 * i + 1]); break;
            case 2: out[i] = get_reflectance(in[4 * i], v3(in[4 * i + 1], in[4 * i + 2], in[4 * i + 3])); break;
            case 3: { V3 r = xyz_to_rgb(chromatic_adapt_E_to_D65(v3(in[3 * i], in[3 * i + 1], in[3 * i + 2]))); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
            case 4:
                for (int k = 0; k < 41; ++k) { out[193 * i + k] = CIE_X[k]; out[193 * i + 41 + k] = CIE_Y[k]; out[193 * i + 82 + k] = CIE_Z[k]; }
                for (int k = 0; k < 70; ++k) out[193 * i + 123 + k] = SMITS_TABLE[k];
                break;
            default: return -1;
        }
    }
    return 0;
}
// Camera ray of pixel (px,py), frame, sample 0: out[0..2] origin, out[3..5] direction.
int orc_camera_ray(const Camera *cam, int px, int py, uint32_t frame, float out[6]) {
    int ssq = (int)rtm::sqrt_((float)cam->rayPerPixel);
    float rssq = 1.0f / (float)ssq;
    PixelBasis b = pixelBasis(*cam, px, py);
    uint32_t seed = rtm::hash_seed((uint32_t)px, (uint32_t)py, frame, 0u);
    Ray r = cameraRay(b, 0, 0, rssq, seed);
    out[0] = r.origin.x; out[1] = r.origin.y; out[2] = r.origin.z;
    out[3] = r.direction.x; out[4] = r.direction.y; out[5] = r.direction.z;
    return 0;
}
// Single-ray hit against the scene: out = {t, idModel, idIdent, nx, ny, nz, front_face}.
int orc_hit(const orc_scene *s, const float origin[3], const float direction[3], float out[7]) {
    Camera cam{};
    Ctx c{&s->sc, cam, v3s(0), 1u, Counters()};
    Ray r{v3(origin[0], origin[1], origin[2]), v3(direction[0], direction[1], direction[2])};
    HitInfo info;
    info.t = 1e20f;
    hit(c, r, info);
    out[0] = info.t;
    out[1] = (float)info.idModel;
    out[2] = (float)info.idIdent;
    out[3] = info.normal.x; out[4] = info.normal.y; out[5] = info.normal.z;
    out[6] = info.front_face ? 1.0f : 0.0f;
    return 0;
}

}  // extern "C"
