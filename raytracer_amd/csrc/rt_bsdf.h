// rt_bsdf.h — sampling and evaluation of the reference's layered material model
// for the HIP shading kernels.  Operation order follows the GLSL exactly (no
// re-association, no FMA) so results match the CPU checker bit for bit.
// RGB functions: reference src/RayTracer.cpp:119-325; Spectral: :701-1049.
#pragma once

#include "rt_device.h"

namespace rtd {

RT_FN V3 reflectAbout(V3 v, V3 n) { return v - dot(v, n) * n * 2.0f; }
RT_FN V3 perpendicular(V3 v) { return (abs_(v.x) > 0.9f) ? v3(0, 1, 0) : v3(1, 0, 0); }

struct Frame {
    V3 t, b, n;
};
RT_FN Frame frameAround(V3 n) {
    Frame f;
    f.n = n;
    f.t = normalize(cross(n, perpendicular(n)));
    f.b = normalize(cross(n, f.t));
    return f;
}
RT_FN V3 toWorld(const Frame &f, V3 local) { return f.t * local.x + f.b * local.y + f.n * local.z; }

// sampleHemisphereCosine, src/RayTracer.cpp:127-141 (two draws)
RT_FN V3 sampleCosineHemisphere(V3 n, uint32_t &seed) {
    float r1 = rand_float(seed);
    float r2 = rand_float(seed);
    float phi = 2.0f * kPi * r1;
    float cosTheta = sqrt_(1.0f - r2);
    float sinTheta = sqrt_(r2);
    float s, c;
    sincos_(phi, &s, &c);
    return toWorld(frameAround(n), v3(c * sinTheta, s * sinTheta, cosTheta));
}

// sampleGGXVNDF_H (Heitz visible-normal sampling), src/RayTracer.cpp:143-174 (two draws)
RT_FN V3 sampleHalfVectorVNDF(V3 n, V3 v, float roughness, uint32_t &seed) {
    float a = roughness * roughness;
    float r1 = rand_float(seed);
    float r2 = rand_float(seed);
    Frame f = frameAround(n);
    V3 vh = normalize(v3(a * dot(v, f.t), a * dot(v, f.b), dot(v, n)));
    V3 t1 = (vh.z < 0.9999f) ? normalize(cross(v3(0, 0, 1), vh)) : v3(1, 0, 0);
    V3 t2 = cross(vh, t1);
    float r = sqrt_(r1);
    float phi = 2.0f * kPi * r2;
    float s, c;
    sincos_(phi, &s, &c);
    float p1 = r * c;
    float p2 = r * s;
    float w = 0.5f * (1.0f + vh.z);
    p2 = (1.0f - w) * sqrt_(1.0f - p1 * p1) + w * p2;
    V3 nh = p1 * t1 + p2 * t2 + sqrt_(fmax_(0.0f, 1.0f - p1 * p1 - p2 * p2)) * vh;
    V3 hh = normalize(a * nh.x * t1 + a * nh.y * t2 + nh.z * vh);
    return normalize(toWorld(f, hh));
}

// Spectral half-vector: classic GGX NDF sampling, src/RayTracer.cpp:725-745 (two draws)
RT_FN V3 sampleHalfVectorNDF(V3 n, float roughness, uint32_t &seed) {
    float a = roughness * roughness;
    float r1 = rand_float(seed);
    float r2 = rand_float(seed);
    float phi = 2.0f * kPi * r1;
    float d = fmax_(1.0f + (a * a - 1.0f) * r2, kMinDenominator);
    float cosTheta = sqrt_((1.0f - r2) / d);
    float sinTheta = sqrt_(1.0f - cosTheta * cosTheta);
    float s, c;
    sincos_(phi, &s, &c);
    return toWorld(frameAround(n), v3(c * sinTheta, s * sinTheta, cosTheta));
}

template <bool SPECTRAL> RT_FN V3 sampleHalfVector(V3 n, V3 v, float roughness, uint32_t &seed) {
    if (SPECTRAL) return sampleHalfVectorNDF(n, roughness, seed);
    return sampleHalfVectorVNDF(n, v, roughness, seed);
}

// sampleGGXVNDF, src/RayTracer.cpp:176-183 / :747-752: zero vector below the horizon.
template <bool SPECTRAL> RT_FN V3 sampleSpecular(V3 n, V3 v, float roughness, uint32_t &seed) {
    V3 h = sampleHalfVector<SPECTRAL>(n, v, roughness, seed);
    V3 l = reflectAbout(-v, h);
    return (dot(n, l) > 0.0f) ? l : v3s(0.0f);
}

RT_FN float pow5(float x) { return pow5_(x); }

// refract, src/RayTracer.cpp:280-285
RT_FN V3 refractThrough(V3 uv, V3 n, float etaRatio) {
    float cosTheta = fmin_(dot(-uv, n), 1.0f);
    V3 perp = etaRatio * (uv + cosTheta * n);
    V3 parallel = -sqrt_(abs_(1.0f - dot(perp, perp))) * n;
    return perp + parallel;
}

// sampleTransmission, src/RayTracer.cpp:294-325 (two draws + one more unless total internal reflection)
template <bool SPECTRAL> RT_FN V3 sampleTransmission(V3 n, V3 v, bool frontFace, float ior, float roughness, uint32_t &seed) {
    float eta = frontFace ? (1.0f / ior) : ior;
    V3 h = sampleHalfVector<SPECTRAL>(n, v, roughness, seed);
    float voh = dot(v, h);
    if (voh < 0.0f) {
        h = n;
        voh = dot(v, n);
    }
    float cosTheta = fmin_(voh, 1.0f);
    float sinTheta = sqrt_(fmax_(1.0f - cosTheta * cosTheta, 0.0f));
    if (eta * sinTheta > 1.0f) return reflectAbout(-v, h);
    float r0 = (1.0f - eta) / (1.0f + eta);
    r0 = r0 * r0;
    float reflectance = r0 + (1.0f - r0) * pow5(1.0f - cosTheta);
    if (rand_float(seed) < reflectance) return reflectAbout(-v, h);
    return refractThrough(-v, h, eta);
}

// NDF_GGX / geometrySchlickGGX / geometrySmith, src/RayTracer.cpp:208-225
RT_FN float ndfGGX(float noh, float roughness) {
    float a = roughness * roughness;
    float a2 = a * a;
    float denom = noh * noh * (a2 - 1.0f) + 1.0f;
    return a2 / fmax_(denom * denom, 1e-16f);
}
RT_FN float g1SchlickGGX(float nov, float roughness) {
    float a = roughness * roughness;
    float k = a * 0.5f;
    return nov / fmax_(nov * (1.0f - k) + k, kMinDenominator);
}
RT_FN float gSmith(float nov, float nol, float roughness) { return g1SchlickGGX(nov, roughness) * g1SchlickGGX(nol, roughness); }

// Lobe selection probabilities, src/RayTracer.cpp:382-391
struct LobeProbs {
    float diffuse, specular, transmission, subsurface;
};
RT_FN LobeProbs lobeProbabilities(const MaterialD &m) {
    LobeProbs p;
    p.transmission = m.transmission;
    p.subsurface = m.subsurface * (1.0f - p.transmission);
    p.diffuse = (1.0f - m.metallic) * (1.0f - p.transmission);
    p.specular = (0.5f + 0.5f * m.metallic) * (1.0f - p.transmission);
    float total = p.subsurface + p.diffuse + p.specular + p.transmission;
    p.subsurface /= total;
    p.diffuse /= total;
    p.specular /= total;
    p.transmission /= total;
    return p;
}

// ---- RGB evaluation (src/RayTracer.cpp:185-269) ----------------------------------------
struct SurfaceRGB {
    V3 albedo;
    float roughness, metallic, specular, subsurface;
};
RT_FN SurfaceRGB surfaceOf(const MaterialD &m) {
    SurfaceRGB s;
    s.albedo = m.albedo;
    s.roughness = m.roughness;
    s.metallic = m.metallic;
    s.specular = m.specular;
    s.subsurface = m.subsurface;
    return s;
}
RT_FN V3 f0RGB(const SurfaceRGB &s) {
    float specular = clamp_(s.specular, 0.0f, 1.0f);
    return vmix(v3s(0.16f * specular * specular), s.albedo, s.metallic);
}
RT_FN V3 schlickRGB(float cosTheta, V3 f0) { return f0 + (v3s(1.0f) - f0) * pow5(1.0f - cosTheta); }
RT_FN float specularPdfRGB(float noh, float nov, float roughness) {
    return ndfGGX(noh, roughness) * g1SchlickGGX(nov, roughness) * noh / fmax_(4.0f * nov, kMinDenominator);
}
RT_FN V3 evalSpecularRGB(const SurfaceRGB &s, float nov, float nol, float noh, float voh) {
    V3 f = schlickRGB(voh, f0RGB(s));
    float d = ndfGGX(noh, s.roughness);
    float g = gSmith(nov, nol, s.roughness);
    return (d * g * f) / fmax_(4.0f * nov * nol, kMinDenominator);
}
RT_FN V3 evalDiffuseRGB(const SurfaceRGB &s, float nol, float nov, float voh) {
    V3 f = schlickRGB(voh, f0RGB(s));
    V3 kd = (v3s(1.0f) - f) * (1.0f - s.metallic);
    float fd90 = 0.5f + 2.0f * s.roughness * voh * voh;
    float retro = (1.0f + (fd90 - 1.0f) * pow5(1.0f - nol)) * (1.0f + (fd90 - 1.0f) * pow5(1.0f - nov));
    return kd * s.albedo * kInvPi * retro;
}
RT_FN V3 evalSubsurfaceRGB(const SurfaceRGB &s, float nol, float nov, float lov) {
    float fl = pow5(1.0f - nol);
    float fv = pow5(1.0f - nov);
    float fd90 = 0.5f + 2.0f * lov * s.roughness;
    float fd = mix_(1.0f, fd90, fl) * mix_(1.0f, fd90, fv);
    return s.albedo * fd * kInvPi * s.subsurface;
}

// ---- Spectral evaluation (src/RayTracer.cpp:754-1049) ---------------------------------
struct SurfaceSpectral {
    float albedo;  // reflectance of the base colour at the path's wavelength
    float roughness, metallic, specular, specularTint;
};
RT_FN float schlickScalar(float cosTheta, float f0) { return f0 + (1.0f - f0) * pow5(clamp_(1.0f - cosTheta, 0.0f, 1.0f)); }
RT_FN float f0Spectral(const SurfaceSpectral &s) {
    float specular = clamp_(s.specular, 0.0f, 1.0f);
    float tint = clamp_(s.specularTint, 0.0f, 1.0f);
    float dielectric = 0.16f * specular * specular;
    float tinted = mix_(dielectric, s.albedo, tint);
    return mix_(tinted, s.albedo, s.metallic);
}
RT_FN float specularPdfSpectral(float noh, float voh, float roughness) {
    return ndfGGX(noh, roughness) * noh / fmax_(4.0f * voh, kMinDenominator);
}
RT_FN float evalSpecularSpectral(const SurfaceSpectral &s, float nov, float nol, float noh, float voh) {
    float f = schlickScalar(voh, f0Spectral(s));
    float d = ndfGGX(noh, s.roughness);
    float g = gSmith(nov, nol, s.roughness);
    return (d * g * f) / fmax_(4.0f * nov * nol, kMinDenominator);
}
RT_FN float evalDiffuseSpectral(const SurfaceSpectral &s, float nol, float nov, float voh) {
    float f = schlickScalar(voh, f0Spectral(s));
    float kd = (1.0f - f) * (1.0f - s.metallic);
    float fd90 = 0.5f + 2.0f * s.roughness * voh * voh;
    float fl = pow5(1.0f - nol);
    float fv = pow5(1.0f - nov);
    float retro = (1.0f + (fd90 - 1.0f) * fl) * (1.0f + (fd90 - 1.0f) * fv);
    return kd * s.albedo * retro * kInvPi;
}
RT_FN float evalSubsurfaceSpectral(const SurfaceSpectral &s, float nol, float nov, float lov) {
    float fl = pow5(1.0f - nol);
    float fv = pow5(1.0f - nov);
    float fss90 = lov * s.roughness;
    float fss = mix_(1.0f, fss90, fl) * mix_(1.0f, fss90, fv);
    float ss = 1.25f * (fss * (1.0f / (nol + nov) - 0.5f) + 0.5f);
    return s.albedo * ss * kInvPi;
}

// ---- colour science for the Spectral integrator (src/RayTracer.cpp:833-978, 1282-1315) ----
constexpr float kWlMin = 380.0f, kWlMax = 780.0f, kWlRange = kWlMax - kWlMin, kCieYIntegral = 106.856895f;

#if defined(__HIPCC__)
#define RT_TABLE __device__ __constant__ static const
#else
#define RT_TABLE static const
#endif

RT_TABLE float kCieX[41] = {0.0014f, 0.0042f, 0.0143f, 0.0435f, 0.1344f, 0.2839f, 0.3483f, 0.3362f, 0.2908f, 0.1954f, 0.0956f,
                            0.0320f, 0.0049f, 0.0093f, 0.0633f, 0.1655f, 0.2904f, 0.4334f, 0.5945f, 0.7621f, 0.9163f, 1.0263f,
                            1.0622f, 1.0026f, 0.8544f, 0.6424f, 0.4479f, 0.2835f, 0.1649f, 0.0874f, 0.0468f, 0.0227f, 0.0114f,
                            0.0058f, 0.0029f, 0.0014f, 0.0007f, 0.0003f, 0.0002f, 0.0001f, 0.0000f};
RT_TABLE float kCieY[41] = {0.0000f, 0.0001f, 0.0004f, 0.0012f, 0.0040f, 0.0116f, 0.0230f, 0.0380f, 0.0600f, 0.0910f, 0.1390f,
                            0.2080f, 0.3230f, 0.5030f, 0.7100f, 0.8620f, 0.9540f, 0.9950f, 0.9950f, 0.9520f, 0.8700f, 0.7570f,
                            0.6310f, 0.5030f, 0.3810f, 0.2650f, 0.1750f, 0.1070f, 0.0610f, 0.0320f, 0.0170f, 0.0082f, 0.0041f,
                            0.0021f, 0.0010f, 0.0005f, 0.0003f, 0.0001f, 0.0001f, 0.0000f, 0.0000f};
RT_TABLE float kCieZ[41] = {0.0065f, 0.0201f, 0.0679f, 0.2074f, 0.6456f, 1.3856f, 1.7471f, 1.7721f, 1.6692f, 1.2876f, 0.8130f,
                            0.4652f, 0.2720f, 0.1582f, 0.0782f, 0.0422f, 0.0203f, 0.0087f, 0.0039f, 0.0021f, 0.0017f, 0.0011f,
                            0.0008f, 0.0003f, 0.0002f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f,
                            0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.0000f};
// Seven 10-sample basis spectra: white, cyan, magenta, yellow, red, green, blue.
RT_TABLE float kSmits[70] = {
    1.0000f, 1.0000f, 0.9999f, 0.9993f, 0.9992f, 0.9998f, 1.0000f, 1.0000f, 1.0000f, 1.0000f,
    0.9710f, 0.9426f, 1.0007f, 1.0007f, 1.0007f, 1.0007f, 0.1564f, 0.0000f, 0.0000f, 0.0000f,
    1.0000f, 1.0000f, 0.9685f, 0.2229f, 0.0000f, 0.0458f, 0.8369f, 1.0000f, 1.0000f, 0.9959f,
    0.0001f, 0.0000f, 0.1088f, 0.6651f, 1.0000f, 1.0000f, 0.9996f, 0.9586f, 0.9685f, 0.9840f,
    0.1012f, 0.0515f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.8325f, 1.0149f, 1.0149f, 1.0149f,
    0.0000f, 0.0000f, 0.0273f, 0.7937f, 1.0000f, 0.9418f, 0.1719f, 0.0000f, 0.0000f, 0.0025f,
    1.0000f, 1.0000f, 0.8916f, 0.3323f, 0.0000f, 0.0000f, 0.0003f, 0.0369f, 0.0483f, 0.0496f};

// get_cie_xyz: the five-step search tops out at table index 18 / 38, so the last
// two intervals of each half are extrapolated from the one before them.
RT_FN V3 cieXYZ(float lambda) {
    if (lambda < 380.0f || lambda > 780.0f) return v3s(0.0f);
    int idx = 0;
    if (lambda >= 580.0f) idx += 20;
    if (lambda >= 380.0f + 10.0f * (float)(idx + 10)) idx += 10;
    if (lambda >= 380.0f + 10.0f * (float)(idx + 5)) idx += 5;
    if (lambda >= 380.0f + 10.0f * (float)(idx + 2)) idx += 2;
    if (lambda >= 380.0f + 10.0f * (float)(idx + 1)) idx += 1;
    float t = (lambda - (380.0f + 10.0f * (float)idx)) / 10.0f;
    return v3(mix_(kCieX[idx], kCieX[idx + 1], t), mix_(kCieY[idx], kCieY[idx + 1], t), mix_(kCieZ[idx], kCieZ[idx + 1], t));
}

RT_FN float smitsBasis(float lambda, int which) {
    lambda = clamp_(lambda, 380.0f, 740.0f);
    float t = (lambda - 380.0f) / 40.0f;
    int idx = (int)floor_(t);
    idx = idx < 0 ? 0 : (idx > 8 ? 8 : idx);
    float frac = t - (float)idx;
    return mix_(kSmits[which * 10 + idx], kSmits[which * 10 + idx + 1], frac);
}

// get_reflectance: Smits RGB -> spectrum decomposition evaluated at one wavelength.
RT_FN float reflectanceAt(float lambda, V3 rgb) {
    rgb = vclamp(rgb, 0.0f, 1.0f);
    float result = 0.0f;
    float lowest = fmin_(rgb.x, fmin_(rgb.y, rgb.z));
    result += lowest * smitsBasis(lambda, 0);
    V3 ex = rgb - v3s(lowest);
    if (rgb.x == lowest) {
        result += fmin_(ex.y, ex.z) * smitsBasis(lambda, 1);
        if (ex.y < ex.z) result += (ex.z - ex.y) * smitsBasis(lambda, 6);
        else result += (ex.y - ex.z) * smitsBasis(lambda, 5);
    } else if (rgb.y == lowest) {
        result += fmin_(ex.x, ex.z) * smitsBasis(lambda, 2);
        if (ex.x < ex.z) result += (ex.z - ex.x) * smitsBasis(lambda, 6);
        else result += (ex.x - ex.z) * smitsBasis(lambda, 4);
    } else {
        result += fmin_(ex.x, ex.y) * smitsBasis(lambda, 3);
        if (ex.x < ex.y) result += (ex.y - ex.x) * smitsBasis(lambda, 5);
        else result += (ex.x - ex.y) * smitsBasis(lambda, 4);
    }
    return clamp_(result, 0.0f, 1.0f);
}

RT_FN V3 mulColumns(V3 c0, V3 c1, V3 c2, V3 v) { return c0 * v.x + c1 * v.y + c2 * v.z; }

// chromatic_adapt_E_to_D65 (Bradford) then xyz_to_rgb.
RT_FN V3 xyzEToLinearSRGB(V3 xyz) {
    const V3 a0 = v3(0.8951f, -0.7502f, 0.0389f), a1 = v3(0.2664f, 1.7135f, -0.0685f), a2 = v3(-0.1614f, 0.0367f, 1.0296f);
    const V3 i0 = v3(0.9870f, 0.4323f, -0.0085f), i1 = v3(-0.1471f, 0.5184f, 0.0400f), i2 = v3(0.1600f, 0.0493f, 0.9685f);
    V3 src = mulColumns(a0, a1, a2, v3(0.997065f, 1.002169f, 0.988182f));
    V3 dst = mulColumns(a0, a1, a2, v3(0.95047f, 1.00000f, 1.08883f));
    V3 gain = dst / src;
    V3 adapted = mulColumns(a0, a1, a2, xyz) * gain;
    adapted = mulColumns(i0, i1, i2, adapted);
    return mulColumns(v3(3.2406f, -0.9689f, 0.0557f), v3(-1.5372f, 1.8758f, -0.2040f), v3(-0.4986f, 0.0415f, 1.0570f), adapted);
}

}  // namespace rtd
