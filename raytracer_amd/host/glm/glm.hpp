// Minimal glm-compatible math shim.
//
// The reference includes the icaven/glm submodule (absent offline) from almost
// every header (reference src/RayCamera.h:3, src/Material.h:3, src/AABB.h:2).
// This header provides only the surface those host classes use, with glm's
// conventions: column-major matrices, `normalize(v) = v * inversesqrt(dot(v,v))`,
// `radians(d) = d * (pi/180)`, `vec * mat` = row-vector product.
#pragma once

#include <cmath>
#include <cstddef>
#include <algorithm>

namespace glm {

enum qualifier { packed_highp = 0, highp = packed_highp, packed = packed_highp, defaultp = highp };

template <int N, typename T, qualifier Q = defaultp> struct vec;

template <typename T, qualifier Q> struct vec<2, T, Q> {
    T x, y;
    vec() : x(T(0)), y(T(0)) {}
    explicit vec(T s) : x(s), y(s) {}
    template <typename A, typename B> vec(A a, B b) : x(static_cast<T>(a)), y(static_cast<T>(b)) {}
    template <typename U, qualifier P> vec(const vec<2, U, P> &o) : x(static_cast<T>(o.x)), y(static_cast<T>(o.y)) {}
    T &operator[](int i) { return (&x)[i]; }
    const T &operator[](int i) const { return (&x)[i]; }
};

template <typename T, qualifier Q> struct vec<3, T, Q> {
    T x, y, z;
    vec() : x(T(0)), y(T(0)), z(T(0)) {}
    explicit vec(T s) : x(s), y(s), z(s) {}
    template <typename A, typename B, typename C>
    vec(A a, B b, C c) : x(static_cast<T>(a)), y(static_cast<T>(b)), z(static_cast<T>(c)) {}
    template <typename U, qualifier P>
    vec(const vec<3, U, P> &o) : x(static_cast<T>(o.x)), y(static_cast<T>(o.y)), z(static_cast<T>(o.z)) {}
    template <typename U, qualifier P>
    explicit vec(const vec<4, U, P> &o) : x(static_cast<T>(o.x)), y(static_cast<T>(o.y)), z(static_cast<T>(o.z)) {}
    T &operator[](int i) { return (&x)[i]; }
    const T &operator[](int i) const { return (&x)[i]; }
};

template <typename T, qualifier Q> struct vec<4, T, Q> {
    T x, y, z, w;
    vec() : x(T(0)), y(T(0)), z(T(0)), w(T(0)) {}
    explicit vec(T s) : x(s), y(s), z(s), w(s) {}
    template <typename A, typename B, typename C, typename D>
    vec(A a, B b, C c, D d)
        : x(static_cast<T>(a)), y(static_cast<T>(b)), z(static_cast<T>(c)), w(static_cast<T>(d)) {}
    template <typename U, qualifier P, typename D>
    vec(const vec<3, U, P> &v, D d)
        : x(static_cast<T>(v.x)), y(static_cast<T>(v.y)), z(static_cast<T>(v.z)), w(static_cast<T>(d)) {}
    T &operator[](int i) { return (&x)[i]; }
    const T &operator[](int i) const { return (&x)[i]; }
};

using vec2 = vec<2, float>;
using vec3 = vec<3, float>;
using vec4 = vec<4, float>;
using ivec2 = vec<2, int>;
using ivec3 = vec<3, int>;

// ---- component-wise arithmetic (2, 3 and 4 wide) ----------------------------
#define GLM_SHIM_BINOP(OP)                                                                                   \
    template <typename T, qualifier Q> inline vec<2, T, Q> operator OP(const vec<2, T, Q> &a, const vec<2, T, Q> &b) { \
        return vec<2, T, Q>(a.x OP b.x, a.y OP b.y);                                                         \
    }                                                                                                        \
    template <typename T, qualifier Q> inline vec<3, T, Q> operator OP(const vec<3, T, Q> &a, const vec<3, T, Q> &b) { \
        return vec<3, T, Q>(a.x OP b.x, a.y OP b.y, a.z OP b.z);                                             \
    }                                                                                                        \
    template <typename T, qualifier Q> inline vec<4, T, Q> operator OP(const vec<4, T, Q> &a, const vec<4, T, Q> &b) { \
        return vec<4, T, Q>(a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w);                                 \
    }                                                                                                        \
    template <typename T, qualifier Q> inline vec<2, T, Q> operator OP(const vec<2, T, Q> &a, T s) {         \
        return vec<2, T, Q>(a.x OP s, a.y OP s);                                                             \
    }                                                                                                        \
    template <typename T, qualifier Q> inline vec<3, T, Q> operator OP(const vec<3, T, Q> &a, T s) {         \
        return vec<3, T, Q>(a.x OP s, a.y OP s, a.z OP s);                                                   \
    }                                                                                                        \
    template <typename T, qualifier Q> inline vec<4, T, Q> operator OP(const vec<4, T, Q> &a, T s) {         \
        return vec<4, T, Q>(a.x OP s, a.y OP s, a.z OP s, a.w OP s);                                         \
    }                                                                                                        \
    template <typename T, qualifier Q> inline vec<3, T, Q> operator OP(T s, const vec<3, T, Q> &a) {         \
        return vec<3, T, Q>(s OP a.x, s OP a.y, s OP a.z);                                                   \
    }
GLM_SHIM_BINOP(+)
GLM_SHIM_BINOP(-)
GLM_SHIM_BINOP(*)
GLM_SHIM_BINOP(/)
#undef GLM_SHIM_BINOP

template <typename T, qualifier Q> inline vec<3, T, Q> operator-(const vec<3, T, Q> &a) { return vec<3, T, Q>(-a.x, -a.y, -a.z); }
template <typename T, qualifier Q> inline vec<3, T, Q> &operator+=(vec<3, T, Q> &a, const vec<3, T, Q> &b) { a = a + b; return a; }
template <typename T, qualifier Q> inline vec<3, T, Q> &operator-=(vec<3, T, Q> &a, const vec<3, T, Q> &b) { a = a - b; return a; }
template <typename T, qualifier Q> inline vec<3, T, Q> &operator*=(vec<3, T, Q> &a, T s) { a = a * s; return a; }
template <typename T, qualifier Q> inline vec<2, T, Q> &operator+=(vec<2, T, Q> &a, const vec<2, T, Q> &b) { a = a + b; return a; }
template <typename T, qualifier Q> inline bool operator==(const vec<3, T, Q> &a, const vec<3, T, Q> &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
template <typename T, qualifier Q> inline bool operator!=(const vec<3, T, Q> &a, const vec<3, T, Q> &b) { return !(a == b); }
template <typename T, qualifier Q> inline bool operator==(const vec<2, T, Q> &a, const vec<2, T, Q> &b) { return a.x == b.x && a.y == b.y; }

// ---- common functions --------------------------------------------------------
template <typename T> inline T min(T a, T b) { return (b < a) ? b : a; }
template <typename T> inline T max(T a, T b) { return (a < b) ? b : a; }
template <typename T> inline T clamp(T x, T lo, T hi) { return min(max(x, lo), hi); }
template <typename T, qualifier Q> inline vec<3, T, Q> min(const vec<3, T, Q> &a, const vec<3, T, Q> &b) {
    return vec<3, T, Q>(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z));
}
template <typename T, qualifier Q> inline vec<3, T, Q> max(const vec<3, T, Q> &a, const vec<3, T, Q> &b) {
    return vec<3, T, Q>(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z));
}
template <typename T, qualifier Q> inline vec<3, T, Q> clamp(const vec<3, T, Q> &v, T lo, T hi) {
    return vec<3, T, Q>(clamp(v.x, lo, hi), clamp(v.y, lo, hi), clamp(v.z, lo, hi));
}
template <qualifier Q> inline vec<3, float, Q> pow(const vec<3, float, Q> &b, const vec<3, float, Q> &e) {
    return vec<3, float, Q>(std::pow(b.x, e.x), std::pow(b.y, e.y), std::pow(b.z, e.z));
}
inline float radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }
template <typename T, qualifier Q> inline T dot(const vec<3, T, Q> &a, const vec<3, T, Q> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T, qualifier Q> inline T dot(const vec<4, T, Q> &a, const vec<4, T, Q> &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
template <typename T, qualifier Q> inline vec<3, T, Q> cross(const vec<3, T, Q> &a, const vec<3, T, Q> &b) {
    return vec<3, T, Q>(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
template <qualifier Q> inline float length(const vec<3, float, Q> &v) { return std::sqrt(dot(v, v)); }
template <qualifier Q> inline vec<3, float, Q> normalize(const vec<3, float, Q> &v) { return v * (1.0f / std::sqrt(dot(v, v))); }

// ---- matrices (column-major, m[col][row]) -------------------------------------
struct mat4;
struct mat3 {
    vec3 c[3];
    mat3() : mat3(1.0f) {}
    explicit mat3(float d) { c[0] = vec3(d, 0, 0); c[1] = vec3(0, d, 0); c[2] = vec3(0, 0, d); }
    mat3(const vec3 &a, const vec3 &b, const vec3 &d) { c[0] = a; c[1] = b; c[2] = d; }
    mat3(const mat4 &m);
    vec3 &operator[](int i) { return c[i]; }
    const vec3 &operator[](int i) const { return c[i]; }
};
struct mat4 {
    vec4 c[4];
    mat4() : mat4(1.0f) {}
    explicit mat4(float d) { c[0] = vec4(d, 0, 0, 0); c[1] = vec4(0, d, 0, 0); c[2] = vec4(0, 0, d, 0); c[3] = vec4(0, 0, 0, d); }
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
};
inline mat3::mat3(const mat4 &m) { c[0] = vec3(m[0]); c[1] = vec3(m[1]); c[2] = vec3(m[2]); }

inline vec3 operator*(const mat3 &m, const vec3 &v) { return m[0] * v.x + m[1] * v.y + m[2] * v.z; }
inline vec3 operator*(const vec3 &v, const mat3 &m) { return vec3(dot(v, m[0]), dot(v, m[1]), dot(v, m[2])); }
// glm pairs the four products as (c0*x + c1*y) + (c2*z + c3*w).
inline vec4 operator*(const mat4 &m, const vec4 &v) { return (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w); }
inline mat4 operator*(const mat4 &a, const mat4 &b) {
    mat4 r(0.0f);
    for (int j = 0; j < 4; ++j) r[j] = a[0] * b[j].x + a[1] * b[j].y + a[2] * b[j].z + a[3] * b[j].w;
    return r;
}
inline mat4 &operator*=(mat4 &a, const mat4 &b) { a = a * b; return a; }
inline mat3 transpose(const mat3 &m) {
    return mat3(vec3(m[0].x, m[1].x, m[2].x), vec3(m[0].y, m[1].y, m[2].y), vec3(m[0].z, m[1].z, m[2].z));
}
inline mat3 inverse(const mat3 &m) {
    float det = m[0].x * (m[1].y * m[2].z - m[2].y * m[1].z) - m[1].x * (m[0].y * m[2].z - m[2].y * m[0].z) +
                m[2].x * (m[0].y * m[1].z - m[1].y * m[0].z);
    float id = 1.0f / det;
    mat3 r;
    r[0].x = +(m[1].y * m[2].z - m[2].y * m[1].z) * id;
    r[1].x = -(m[1].x * m[2].z - m[2].x * m[1].z) * id;
    r[2].x = +(m[1].x * m[2].y - m[2].x * m[1].y) * id;
    r[0].y = -(m[0].y * m[2].z - m[2].y * m[0].z) * id;
    r[1].y = +(m[0].x * m[2].z - m[2].x * m[0].z) * id;
    r[2].y = -(m[0].x * m[2].y - m[2].x * m[0].y) * id;
    r[0].z = +(m[0].y * m[1].z - m[1].y * m[0].z) * id;
    r[1].z = -(m[0].x * m[1].z - m[1].x * m[0].z) * id;
    r[2].z = +(m[0].x * m[1].y - m[1].x * m[0].y) * id;
    return r;
}

// ---- quaternion (w, x, y, z constructor order, as glm) -----------------------------
struct quat {
    float x, y, z, w;
    quat() : x(0), y(0), z(0), w(1) {}
    quat(float w_, float x_, float y_, float z_) : x(x_), y(y_), z(z_), w(w_) {}
};

}  // namespace glm
