// rt_math.h — pinned fp32 math shared by the HIP kernels and the CPU checker.
//
// GLSL leaves the precision of normalize/sqrt/pow/exp/log/sin/cos/tan and the
// NaN behaviour of min/max/clamp to the driver.  This header fixes one
// definition of each, written only with IEEE-754 +,-,*,/ and sqrt (all correctly
// rounded on x86-64 SSE and on gfx950 when built with -ffp-contract=off and
// -fhip-fp32-correctly-rounded-divide-sqrt), integer bit manipulation and
// exact float<->int conversions.  The same source therefore produces the same
// bits on the CPU and on the GPU, which is what lets radiance be compared
// sample for sample.  Build every user of this header with -ffp-contract=off
// and without fast-math.
//
// Pinned choices:
//   min/max/clamp : IEEE minNum/maxNum (a NaN operand yields the other operand);
//                   clamp(x,lo,hi) = min(max(x,lo),hi).  The sign of a zero
//                   result is unspecified (+0 and -0 compare equal everywhere
//                   they are consumed).
//   normalize(v)  : v * (1 / sqrt(dot(v,v)))  (one division, three products)
//   dot           : ((a.x*b.x + a.y*b.y) + a.z*b.z), no fused multiply-add
//   mix(a,b,t)    : a*(1-t) + b*t
//   pow(x,y)      : exp(y*log(x)) for x > 0; 0 for x == 0; NaN for x < 0;
//                   the integer powers the shaders use are products instead:
//                   pow(x,5) = ((x*x)*(x*x))*x and pow(x,2) = x*x
//   sin/cos       : 3-term Cody-Waite reduction by pi/2 + degree-7/8 polynomials
//   exp/log       : Cephes single-precision kernels (about 1 ulp without FMA)
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_FN __host__ __device__ __forceinline__
#else
#define RT_FN static inline
#endif

namespace rtm {

RT_FN uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
RT_FN float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
RT_FN int32_t f2i(float f) { return (int32_t)f2u(f); }

RT_FN float inf() { return u2f(0x7f800000u); }
RT_FN float nan() { return u2f(0x7fc00000u); }

#if defined(__HIP_DEVICE_COMPILE__)
RT_FN float fmin_(float a, float b) { return __builtin_fminf(a, b); }
RT_FN float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
// The same operations written as instructions, for the box test of the traversal kernel:
// v_min_f32 / v_max_f32 are minNum / maxNum for quiet NaNs, and arithmetic never produces a
// signalling one; the compiler cannot know that and puts a canonicalising v_max_f32 x, x in
// front of operands of the builtins (14 extra instructions per opened tree node).
// Hand-written instructions are invisible to the compiler's hazard recogniser: on gfx950 a vector
// instruction that reads the result of a transcendental one (v_rcp_f32, v_rsq_f32, v_sqrt_f32 ...)
// needs one instruction in between, and the hardware does not wait by itself.  Operands of these four
// must therefore never come straight from such an instruction; clampMag_ below is for that case
// (tools/isa_audit.py checks the generated code for the pair).
RT_FN float fminI_(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_FN float fmaxI_(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_FN float fmin3_(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RT_FN float fmax3_(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// min(max(x, -m), m) as one instruction the compiler knows (v_med3_f32), so that it can keep the
// required distance from a transcendental instruction that produced x.
RT_FN float clampMag_(float x, float m) { return __builtin_amdgcn_fmed3f(x, -m, m); }
#else
RT_FN float fmin_(float a, float b) { return (a != a) ? b : ((b < a) ? b : a); }
RT_FN float fmax_(float a, float b) { return (a != a) ? b : ((b > a) ? b : a); }
RT_FN float fminI_(float a, float b) { return fmin_(a, b); }
RT_FN float fmaxI_(float a, float b) { return fmax_(a, b); }
RT_FN float fmin3_(float a, float b, float c) { return fmin_(fmin_(a, b), c); }
RT_FN float fmax3_(float a, float b, float c) { return fmax_(fmax_(a, b), c); }
RT_FN float clampMag_(float x, float m) { return fmin_(fmax_(x, -m), m); }
#endif
RT_FN float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
RT_FN float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
// RT_FAST_SHADE (the opt-in shading kernels of rt_shade_fast.hip, device code only): the hardware's own
// square root, reciprocal square root, exponential, logarithm, sine and cosine in place of the pinned
// sequences.  Results then differ from the checker's in the last bits; the traversal never uses these.
#if defined(RT_FAST_SHADE) && defined(__HIP_DEVICE_COMPILE__)
#define RT_FAST_DEVICE_MATH 1
RT_FN float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
RT_FN float sqrt_(float x) { return __builtin_sqrtf(x); }
#endif
RT_FN float floor_(float x) { return __builtin_floorf(x); }
RT_FN float fract_(float x) { return x - floor_(x); }
RT_FN float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }

// 2^n * z for integer n, built from exponent bits; flushes below the normal
// range to 0 and saturates above it to +inf (z is in [0.5, 2]).
RT_FN float scale2_(float z, int n) {
    if (n < -125) return 0.0f;
    if (n > 127) return inf();
    return z * u2f((uint32_t)(n + 127) << 23);
}

#ifdef RT_FAST_DEVICE_MATH
RT_FN float exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
RT_FN float log_(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
#else
RT_FN float exp_(float x) {
    if (x != x) return x;
    if (x > 88.7228394f) return inf();
    if (x < -87.3365479f) return 0.0f;
    float z = floor_(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    float xx = x * x;
    float p = 1.9875691500e-4f;
    p = p * x + 1.3981999507e-3f;
    p = p * x + 8.3334519073e-3f;
    p = p * x + 4.1665795894e-2f;
    p = p * x + 1.6666665459e-1f;
    p = p * x + 5.0000001201e-1f;
    p = p * xx + x + 1.0f;
    return scale2_(p, n);
}

RT_FN float log_(float x) {
    if (x != x || x < 0.0f) return nan();
    if (x == 0.0f) return -inf();
    if (x == inf()) return x;
    int e = 0;
    uint32_t u = f2u(x);
    if (u < 0x00800000u) {  // subnormal: renormalise by 2^23
        x = x * 8388608.0f;
        u = f2u(x);
        e = -23;
    }
    e += (int)(u >> 23) - 126;                         // x = m * 2^e, m in [0.5, 1)
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = p * m - 1.1514610310e-1f;
    p = p * m + 1.1676998740e-1f;
    p = p * m - 1.2420140846e-1f;
    p = p * m + 1.4249322787e-1f;
    p = p * m - 1.6668057665e-1f;
    p = p * m + 2.0000714765e-1f;
    p = p * m - 2.4999993993e-1f;
    p = p * m + 3.3333331174e-1f;
    float y = p * m * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    r = r + 0.693359375f * fe;
    return r;
}
#endif

RT_FN float pow_(float x, float y) {
    if (x == 0.0f) return 0.0f;
    return exp_(y * log_(x));
}
RT_FN float pow5_(float x) {
    float x2 = x * x;
    return (x2 * x2) * x;
}
RT_FN float pow2_(float x) { return x * x; }

// sin and cos of x (radians) for |x| up to a few thousand.
#ifdef RT_FAST_DEVICE_MATH
RT_FN void sincos_(float x, float *s, float *c) {
    const float turns = x * 0.159154943091895336f;   // v_sin_f32 / v_cos_f32 take revolutions
    *s = __builtin_amdgcn_sinf(turns);
    *c = __builtin_amdgcn_cosf(turns);
}
#else
RT_FN void sincos_(float x, float *s, float *c) {
    float k = floor_(x * 0.636619772367581343f + 0.5f);
    int q = (int)k;
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.549789948768648e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    float sn = r + r * z * ps;
    float pc = 2.443315711809948e-5f;
    pc = pc * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    float cs = 1.0f - 0.5f * z + z * z * pc;
    switch (q & 3) {
        case 0: *s = sn; *c = cs; break;
        case 1: *s = cs; *c = -sn; break;
        case 2: *s = -sn; *c = -cs; break;
        default: *s = -cs; *c = sn; break;
    }
}
#endif
RT_FN float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
RT_FN float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }
RT_FN float tan_(float x) { float s, c; sincos_(x, &s, &c); return s / c; }

// ---- vec3 ----------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
RT_FN V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_FN V3 v3s(float s) { return v3(s, s, s); }
RT_FN V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_FN V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_FN V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_FN V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
RT_FN V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_FN V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RT_FN V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_FN V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_FN V3 cross(V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
RT_FN float length(V3 a) { return sqrt_(dot(a, a)); }
#ifdef RT_FAST_DEVICE_MATH
RT_FN V3 normalize(V3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }
#else
RT_FN V3 normalize(V3 a) { return a * (1.0f / sqrt_(dot(a, a))); }
#endif
RT_FN V3 vmin(V3 a, V3 b) { return v3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
RT_FN V3 vmax(V3 a, V3 b) { return v3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
RT_FN V3 vclamp(V3 a, float lo, float hi) { return v3(clamp_(a.x, lo, hi), clamp_(a.y, lo, hi), clamp_(a.z, lo, hi)); }
RT_FN V3 vmix(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }

// ---- RNG (reference src/RayTracer.cpp:99-117) -----------------------------------
RT_FN uint32_t pcg(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
RT_FN uint32_t hash_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t sample) {
    uint32_t h = (px * 73856093u) ^ (py * 19349663u) ^ (frame * 83492791u) ^ (sample * 2654435761u);
    return pcg(h);
}
RT_FN uint32_t rand_u32(uint32_t &seed) {
    seed = pcg(seed);
    return seed;
}
// float(u32) rounds to nearest, so the result lies in [0, 1] inclusive.
RT_FN float rand_float(uint32_t &seed) { return (float)rand_u32(seed) / 4294967296.0f; }

}  // namespace rtm
