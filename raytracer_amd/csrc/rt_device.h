// rt_device.h — device-side data layout and scene decode for the HIP backend.
//
// Two views of the scene live in HBM:
//   * the five raw word streams exactly as the host serialised them (used by the
//     shading kernels, which touch each hit once), and
//   * a traversal re-pack built at commit time: 64-byte inner nodes holding both
//     child boxes, 48-byte triangle records (A, B-A, C-A + ids), and 64-byte shading
//     records (vertex normals and texture coordinates) under the same triangle index.
// Every function cites the reference GLSL it reproduces (paths in the reference tree).
#pragma once

#include "rt_math.h"

namespace rtd {

using namespace rtm;

// ---- traversal re-pack ------------------------------------------------------------
// Child reference: >= 0 -> index of an inner node in `nodes`;
//                  <  0 -> leaf: ~ref = (firstTriangle << 2) | triangleCount.
struct __attribute__((aligned(64))) PackedNode {
    float lmin[3]; int32_t left;
    float lmax[3]; int32_t right;
    float rmin[3]; int32_t pad0;
    float rmax[3]; int32_t pad1;
};
static_assert(sizeof(PackedNode) == 64, "inner node must be one 64-byte line");

// Optional four-wide node derived from the same tree at commit time (one 64-byte line): the
// children's boxes are stored as 8-bit offsets from `o` in units of `s` per axis, rounded outwards,
// so plane = o + q*s never cuts into the box it stands for.  Child references as above; an unused
// slot holds the inverted box lo=255, hi=0 and the reference ~0 (a leaf without triangles).
struct __attribute__((aligned(64))) WideNode {
    float o[3]; int32_t c0;
    float s[3]; int32_t c1;
    uint32_t lo[3]; int32_t c2;    // byte k of lo[a] = lower plane of child k on axis a
    uint32_t hi[3]; int32_t c3;
};
static_assert(sizeof(WideNode) == 64, "wide node must be one 64-byte line");

// Eight-wide node (RT_OPT_WIDE_BVH 4), 80 bytes = five 16-byte loads.  Child boxes as in WideNode,
// on a grid of cell size 2^(e-127) per axis.  Slot k (0..7) is described by byte k of `meta`:
//   0                    empty slot (its box is inverted: lo 255, hi 0)
//   0x38 + k             inner child: node number childBase + (number of inner slots below k)
//   (2^n - 1) << 5 | f   leaf of n (1..3) triangles at wtris[triBase + f .. + n), f <= 23
// so that bits 5..7 of a byte shifted left by its bits 0..4 give the child's bits in a 32-bit hit
// word: one bit per triangle of the node in bits 0..23, one bit per inner slot in bits 24..31.
struct __attribute__((aligned(16))) Wide8Node {
    float o[3]; uint32_t expImask;     // ex | ey << 8 | ez << 16 | (mask of inner slots) << 24
    uint32_t childBase, triBase, meta[2];
    uint32_t lo[3][2];                 // byte k&3 of lo[a][k>>2] = lower plane of slot k on axis a
    uint32_t hi[3][2];
};
static_assert(sizeof(Wide8Node) == 80, "eight-wide node must be five 16-byte pieces");
constexpr int32_t kNoRoot = (int32_t)0x80000000;   // root reference of a model with nothing to walk; not a valid child reference

struct __attribute__((aligned(16))) PackedTri {
    float a[3];  int32_t material;   // vertex A,   scene-wide material index
    float ab[3]; int32_t model;      // B - A,      model slot | flags (kTriAlphaTest)
    float ac[3]; int32_t ident;      // C - A,      identifier index inside the model (leaf order)
};
static_assert(sizeof(PackedTri) == 48, "triangle record must be 48 bytes");
// Two triangles of one leaf of a four-wide tree with their components interleaved, so that each 16-byte
// piece holds two aligned (first, second) pairs: the traversal kernel tests both with packed arithmetic.
// A leaf with an odd number of triangles leaves the second half of its last pair zero (ident = kMiss): a
// degenerate triangle that no ray hits.
struct __attribute__((aligned(16))) TriPair {
    float ax[2], ay[2];
    float az[2], abx[2];
    float aby[2], abz[2];
    float acx[2], acy[2];
    float acz[2]; int32_t ident[2];   // index into tris (what hit records carry)
};
static_assert(sizeof(TriPair) == 80, "triangle pair must be five 16-byte pieces");
// What shading needs of a triangle whose material has no textures, in one 64-byte line under the
// triangle's index: the vertex normals gathered from the vertex stream at commit time, and what
// resolveHit would otherwise derive from PackedTri on every hit -- cross(B - A, C - A), the area
// (0.5 * length of that, src/RayScene.cpp:677-679), material and model -- computed at commit with the
// same correctly rounded operations.  A textured triangle also reads its PackedTri and PackedUV.
struct __attribute__((aligned(64))) PackedShade {
    float n0[3], n1[3], n2[3];
    float geometric[3];          // cross(ab, ac)
    float area;
    int32_t material;            // scene-wide material index
    int32_t model;               // model slot
    int32_t hasTextures;         // the identifier's texture flag (src/RayScene.cpp:659)
};
static_assert(sizeof(PackedShade) == 64, "shading record must be one 64-byte line");
// Texture coordinates of the three vertices (textured triangles and the alpha-cut test only).
struct __attribute__((aligned(32))) PackedUV {
    float uv0[2], uv1[2], uv2[2];
    float pad[2];
};
static_assert(sizeof(PackedUV) == 32, "texture coordinates must be half a 64-byte line");

constexpr int32_t kTriAlphaTest = 1 << 30;  // triangle needs the alpha-cut texture test
constexpr int32_t kTriModelMask = (1 << 30) - 1;

struct ModelDesc {
    int32_t rootRef;            // child reference of node 0 of the uploaded binary tree
    int32_t nodesCount, identifiersCount, lightSourcesCount;
    int32_t lightBase;          // word offset of the light identifiers (the info stream's location)
    int32_t nodeBase;           // word offset of node 0
    int32_t identBase;          // word offset of identifier 0
    int32_t vertBase;           // word offset of vertex 0
    int32_t triBase;            // index of identifier 0 in the packed triangle array
    int32_t wideRootRef;        // child reference of the model's root in the four-wide tree (when that mode is on)
    int32_t lightTriBase;       // index of the model's first light triangle in SceneView::lightTris
    int32_t pad;
};
// A light triangle as sampleLightPoint needs it (the three positions and the area), gathered at commit from the
// model's light identifiers and vertices with the same single-precision operations the sampling would run.
struct __attribute__((aligned(16))) LightTri {
    float p0[3], p1[3], p2[3];
    float area;                 // 0.5 * length(cross(p1 - p0, p2 - p0))
    float pad[2];
};
static_assert(sizeof(LightTri) == 48, "light triangle record must be three 16-byte pieces");

struct SceneView {
    const float *objects, *materials, *modelInfo, *models, *textures;
    const PackedNode *nodes;
    const WideNode *wnodes;     // four-wide tree, null unless that mode is on
    const Wide8Node *w8nodes;   // eight-wide tree, null unless that mode is on
    const PackedTri *tris;
    const PackedShade *shade;   // [triangle], same index as tris
    const PackedUV *uvs;        // [triangle], same index as tris
    const LightTri *lightTris;  // light triangles of all models, a model's at ModelDesc::lightTriBase
    const int32_t *primAt;      // [objectCount]: word offset of each analytic primitive's record in `objects`
    const PackedTri *wtris;     // four-wide trees: the triangles in THEIR leaf order; `ident` holds the index into tris
    const TriPair *wpairs;      // four-wide trees of scenes without alpha-cut textures: the leaves as triangle pairs (leaf references then count pairs)
    const ModelDesc *modelDesc;
    const float *unorm8;        // [512]: k / 255.0f for k = 0..255, then srgb_to_linear of each (textures stored as RGBA8)
    int32_t objectCount, modelsCount, lightSourcesCount;
};
constexpr int32_t kTexPacked8 = 1 << 16;   // in a texture header's channel word: texels are RGBA8 words

// Hit record written by the traversal kernel: 16 bytes.
//   id == kMiss            : nothing hit (t = 1e20)
//   id & kPrimBit          : analytic primitive, index = id & kIndexMask
//   otherwise              : packed triangle index = id & kIndexMask;
//                            kNoTexBit set when the model hit only TIED the
//                            closest primitive, so textures are not applied
//                            (the `startClosest > closest` test of src/RayScene.cpp:763).
struct __attribute__((aligned(16))) HitRec {
    float t, u, v;
    uint32_t id;
};
constexpr uint32_t kMiss = 0xffffffffu;
constexpr uint32_t kPrimBit = 0x80000000u;
constexpr uint32_t kNoTexBit = 0x40000000u;
constexpr uint32_t kIndexMask = 0x3fffffffu;

struct __attribute__((aligned(16))) RayRec {
    float ox, oy, oz, pad0;
    float dx, dy, dz, pad1;
};

struct V2 { float x, y; };
RT_FN V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
RT_FN V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
RT_FN V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
RT_FN V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }

constexpr float kMinDenominator = 1e-8f;   // MIN_DENOMINATOR, src/RayTracer.cpp:29
constexpr float kPi = 3.1415926f;          // src/RayTracer.cpp:31-33
constexpr float kInvPi = 1.0f / kPi;
constexpr float kRad = kPi / 180.0f;
constexpr int kStackDepth = 32;            // int stack[32], src/RayScene.cpp:682: deepest tree the walk can hold
constexpr int kIdentWords = 5;             // src/RayScene.cpp:674-675
constexpr int kNodeWords = 9;
constexpr int kMaterialWords = 23;         // 92 / 4, src/RayScene.cpp:387

// ---- raw stream access --------------------------------------------------------------
RT_FN float ldw(const float *s, int i) { return s[i]; }
RT_FN int32_t ldi(const float *s, int i) { return f2i(s[i]); }
RT_FN V3 ld3(const float *s, int i) { return v3(s[i], s[i + 1], s[i + 2]); }
RT_FN V2 ld2(const float *s, int i) { return v2(s[i], s[i + 1]); }

struct MaterialD {
    int32_t normalTexture, baseColorTexture, metallicRoughnessTexture, emissiveTexture, transmissionTexture, occlusionTexture;
    V3 emissionColor;
    float emissionStrength;
    V3 albedo;
    float subsurface, roughness, metallic, specular, specularTint, transmission, ior, alphaCut, normalScale, occlusionStrength;
};

// loadMaterial, src/RayScene.cpp:383-415
RT_FN MaterialD loadMaterial(const SceneView &sc, int materialIndex) {
    const float *m = sc.materials + materialIndex * kMaterialWords;
    MaterialD r;
    r.normalTexture = f2i(m[0]);
    r.baseColorTexture = f2i(m[1]);
    r.metallicRoughnessTexture = f2i(m[2]);
    r.emissiveTexture = f2i(m[3]);
    r.transmissionTexture = f2i(m[4]);
    r.occlusionTexture = f2i(m[5]);
    r.emissionColor = v3(m[6], m[7], m[8]);
    r.emissionStrength = m[9];
    r.albedo = v3(m[10], m[11], m[12]);
    r.subsurface = m[13];
    r.roughness = m[14];
    r.metallic = m[15];
    r.specular = m[16];
    r.specularTint = m[17];
    r.transmission = m[18];
    r.ior = m[19];
    r.alphaCut = m[20];
    r.normalScale = m[21];
    r.occlusionStrength = m[22];
    return r;
}

// srgb_to_linear, src/RayScene.cpp:268-277
RT_FN float srgbToLinear(float c) {
    if (c <= 0.04045f) return c / 12.92f;
    return pow_((c + 0.055f) / 1.055f, 2.4f);
}

// getTextureUV + getTextureItemIndex, src/RayScene.cpp:337-380.  `header` is the
// word offset of the 5-word texture header; returns the word offset of the texel.
RT_FN float wrapCoord(int mode, float x) {
    if (mode == 10497) return fract_(x);
    if (mode == 33071) return clamp_(x, 0.0f, 1.0f);
    if (mode == 33648) {
        float t = fract_(x * 0.5f) * 2.0f;
        return t > 1.0f ? 2.0f - t : t;
    }
    return x;
}
// A texel of the device's texture buffer.  `at` is the word offset of its first channel, or, for a texture the
// commit stored with 8-bit channels (kTexPacked8 in the header's channel word), of the one word that holds
// them.  Channel k of the texel, and the words after it that the reference's fetches reach
// (src/RayScene.cpp:571-572 reads a fourth word whatever the channel count), come back from texelWord.
struct TexelAt {
    int at;
    uint32_t packed;      // 8-bit texels: channel c in byte c; a three-channel texel holds the next texel's red in byte 3
    int tail;             // 8-bit three-channel textures, last texel only: word offset of the raw word behind the texture; else 0
    bool is8;
};
RT_FN TexelAt texelAt(const SceneView &sc, int header, V2 uv) {
    const float *h = sc.textures + header;
    int width = f2i(h[0]), height = f2i(h[1]), channels = f2i(h[2]), wrapS = f2i(h[3]), wrapT = f2i(h[4]);
    float x = clamp_(wrapCoord(wrapS, uv.x), 0.0f, 0.999999f);
    float y = clamp_(wrapCoord(wrapT, uv.y), 0.0f, 0.999999f);
    const int texel = (int)(x * (float)width) + (int)(y * (float)height) * width;
    TexelAt t;
    t.is8 = (channels & kTexPacked8) != 0;
    t.at = header + 5 + (t.is8 ? texel : texel * channels);
    t.packed = t.is8 ? f2u(sc.textures[t.at]) : 0u;
    t.tail = (channels == (3 | kTexPacked8) && texel == width * height - 1) ? header + 5 + width * height : 0;
    return t;
}
// Channel k (0..3) as the float the upload held: (float)byte / 255.0f through the table for 8-bit texels.
RT_FN float texelWord(const SceneView &sc, const TexelAt &t, int k) {
    if (t.is8) {
        if (k == 3 && t.tail != 0) return sc.textures[t.tail];
        return sc.unorm8[(t.packed >> (8 * k)) & 0xffu];
    }
    return sc.textures[t.at + k];
}
// srgb_to_linear of channel k; RGBA8 texels take it from the table made with the same arithmetic at commit.
RT_FN float texelLinear(const SceneView &sc, const TexelAt &t, int k) {
    if (t.is8) return sc.unorm8[256u + ((t.packed >> (8 * k)) & 0xffu)];
    return srgbToLinear(sc.textures[t.at + k]);
}

// Barycentric UV interpolation used for texturing (Cramer form, src/RayScene.cpp:764-780).
// e0 = B - A and e1 = C - A are the differences the re-pack stored (the same fp32 subtractions).
RT_FN V2 cramerUV(V3 a, V3 e0, V3 e1, V2 ta, V2 tb, V2 tc, V3 p) {
    V3 vp = p - a;
    float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1), d20 = dot(vp, e0), d21 = dot(vp, e1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    return u * ta + v * tb + w * tc;
}

// ---- resolved hit (what `hit()` leaves in HitInfo, src/RayScene.cpp:847-898) -------
struct HitInfoD {
    V3 point, normal;
    float t, area;
    int32_t modelLightCounts;
    bool front_face;
    bool isPrimitive;
    MaterialD mat;
};

// Shading normal of a triangle hit (hitTriangle, src/RayScene.cpp:582-595).
RT_FN V3 triangleNormal(V3 n0, V3 n1, V3 n2, V3 geometric, float u, float v) {
    if (dot(n0, n0) > 0.0f) {
        float w = 1.0f - u - v;
        return normalize(n0 * w + n1 * u + n2 * v);
    }
    return normalize(geometric);
}

// Texture part of hitModels (src/RayScene.cpp:763-844): base colour, metallic-roughness, normal map,
// emissive, transmission and occlusion textures modify the material and the shading normal.
// (Inline: as a call it saves 6-12 registers, not enough for another wave, and costs 144 B of scratch.)
RT_FN void applyTextures(const SceneView &sc, HitInfoD &info, V3 rd, V3 vertexA, V3 edgeAB, V3 edgeAC, V2 uv0, V2 uv1, V2 uv2) {
    V2 uv = cramerUV(vertexA, edgeAB, edgeAC, uv0, uv1, uv2, info.point);
    MaterialD &m = info.mat;
    if (m.baseColorTexture != -1) {
        const TexelAt t = texelAt(sc, m.baseColorTexture, uv);
        m.albedo = v3(texelLinear(sc, t, 0), texelLinear(sc, t, 1), texelLinear(sc, t, 2));
        m.transmission *= 1.0f - texelWord(sc, t, 3);
    }
    if (m.metallicRoughnessTexture != -1) {
        const TexelAt t = texelAt(sc, m.metallicRoughnessTexture, uv);
        m.roughness *= texelWord(sc, t, 1);
        m.metallic *= texelWord(sc, t, 2);
    }
    if (m.normalTexture != -1) {
        // Tangent frame from the triangle's UVs (src/RayScene.cpp:597-621); the
        // normal map is skipped when they do not define one.
        V2 d1 = uv1 - uv0, d2 = uv2 - uv0;
        float uvDet = d1.x * d2.y - d2.x * d1.y;
        if (abs_(uvDet) > kMinDenominator) {
            float f = 1.0f / uvDet;
            V3 tangent = v3(f * (d2.y * edgeAB.x - d1.y * edgeAC.x), f * (d2.y * edgeAB.y - d1.y * edgeAC.y),
                            f * (d2.y * edgeAB.z - d1.y * edgeAC.z));
            tangent = tangent - dot(tangent, info.normal) * info.normal;
            if (length(tangent) > kMinDenominator) {
                tangent = normalize(tangent);
                V3 bitangent = cross(info.normal, tangent);
                const TexelAt t = texelAt(sc, m.normalTexture, uv);
                V3 tn = v3(texelWord(sc, t, 0), texelWord(sc, t, 1), texelWord(sc, t, 2)) * 2.0f - v3s(1.0f);
                tn.x *= m.normalScale;
                tn.y *= m.normalScale;
                tn = normalize(tn);
                info.normal = normalize(tangent * tn.x + bitangent * tn.y + info.normal * tn.z);
                info.front_face = dot(rd, info.normal) < 0.0f;
            }
        }
    }
    if (m.emissiveTexture != -1) {
        const TexelAt t = texelAt(sc, m.emissiveTexture, uv);
        m.emissionColor = m.emissionColor * v3(texelWord(sc, t, 0), texelWord(sc, t, 1), texelWord(sc, t, 2));
    }
    if (m.transmissionTexture != -1) {
        const TexelAt t = texelAt(sc, m.transmissionTexture, uv);
        m.transmission *= texelWord(sc, t, 0);
    }
    if (m.occlusionTexture != -1) {
        const TexelAt t = texelAt(sc, m.occlusionTexture, uv);
        m.transmission *= 1.0f - (1.0f - texelWord(sc, t, 0)) * (1.0f - m.occlusionStrength);
    }
}

// Rebuilds the full HitInfo of a recorded hit.  The record must not be a miss.
RT_FN HitInfoD resolveHit(const SceneView &sc, V3 ro, V3 rd, const HitRec &h) {
    HitInfoD info;
    info.t = h.t;
    info.point = ro + h.t * rd;   // rayAt
    info.area = 0.0f;
    info.modelLightCounts = 0;
    info.isPrimitive = (h.id & kPrimBit) != 0u;

    if (info.isPrimitive) {
        // Record `index` of the variable-length primitive stream (src/RayScene.cpp:855-891), found at commit.
        int at = sc.primAt[h.id & kIndexMask];
        const int type = ldi(sc.objects, at), material = ldi(sc.objects, at + 1);
        at += 2;
        if (type == 0) {            // hitSphere, src/RayScene.cpp:461-464
            V3 center = ld3(sc.objects, at);
            float radius = ldw(sc.objects, at + 3);
            info.normal = normalize((info.point - center) / radius);
        } else if (type == 1) {     // hitQuad, src/RayScene.cpp:480-506
            V3 qu = ld3(sc.objects, at + 3), qv = ld3(sc.objects, at + 6);
            V3 normal = cross(qu, qv);
            float denom = dot(normal, rd);
            info.normal = denom < 0.0f ? normalize(normal) : -normalize(normal);
        } else {                    // top-level Triangle, src/RayScene.cpp:511-523
            V3 a = ld3(sc.objects, at), b = ld3(sc.objects, at + 3), c = ld3(sc.objects, at + 6);
            V3 n0 = ld3(sc.objects, at + 9), n1 = ld3(sc.objects, at + 12), n2 = ld3(sc.objects, at + 15);
            info.normal = triangleNormal(n0, n1, n2, cross(b - a, c - a), h.u, h.v);
        }
        info.front_face = dot(rd, info.normal) < 0.0f;
        info.mat = loadMaterial(sc, material);
        return info;
    }

    const uint32_t tri = h.id & kIndexMask;
    const PackedShade &ps = sc.shade[tri];
    const ModelDesc &md = sc.modelDesc[ps.model];
    const bool hasTextures = ps.hasTextures != 0;
    const V3 n0 = v3(ps.n0[0], ps.n0[1], ps.n0[2]), n1 = v3(ps.n1[0], ps.n1[1], ps.n1[2]), n2 = v3(ps.n2[0], ps.n2[1], ps.n2[2]);

    info.normal = triangleNormal(n0, n1, n2, v3(ps.geometric[0], ps.geometric[1], ps.geometric[2]), h.u, h.v);
    info.front_face = dot(rd, info.normal) < 0.0f;
    info.area = ps.area;                                        // triArea, src/RayScene.cpp:677-679
    info.modelLightCounts = md.lightSourcesCount;
    info.mat = loadMaterial(sc, ps.material);

    if (hasTextures && (h.id & kNoTexBit) == 0u) {              // src/RayScene.cpp:763
        const PackedTri &pt = sc.tris[tri];
        const PackedUV &pu = sc.uvs[tri];
        applyTextures(sc, info, rd, v3(pt.a[0], pt.a[1], pt.a[2]), v3(pt.ab[0], pt.ab[1], pt.ab[2]), v3(pt.ac[0], pt.ac[1], pt.ac[2]),
                      v2(pu.uv0[0], pu.uv0[1]), v2(pu.uv1[0], pu.uv1[1]), v2(pu.uv2[0], pu.uv2[1]));
    }
    return info;
}

// sampleRandomPointFromLightSouces, src/RayScene.cpp:900-935.  Four draws when
// any light model exists, none otherwise.
RT_FN V3 sampleLightPoint(const SceneView &sc, uint32_t &seed, float &area, float &lightTriangleCount) {
    if (sc.lightSourcesCount == 0) {
        area = 0.0f;
        lightTriangleCount = 0.0f;
        return v3s(0.0f);
    }
    int pick = (int)(rand_u32(seed) % (uint32_t)sc.lightSourcesCount);
    const ModelDesc &md = sc.modelDesc[pick];
    lightTriangleCount = (float)md.lightSourcesCount;
    const LightTri &lt = sc.lightTris[md.lightTriBase + (int)(rand_u32(seed) % (uint32_t)md.lightSourcesCount)];
    const V3 p0 = v3(lt.p0[0], lt.p0[1], lt.p0[2]), p1 = v3(lt.p1[0], lt.p1[1], lt.p1[2]), p2 = v3(lt.p2[0], lt.p2[1], lt.p2[2]);
    float r1 = rand_float(seed);
    float r2 = rand_float(seed);
    if (r1 + r2 > 1.0f) {
        r1 = 1.0f - r1;
        r2 = 1.0f - r2;
    }
    area = lt.area;
    return r1 * p0 + r2 * p1 + (1.0f - r1 - r2) * p2;
}

}  // namespace rtd
