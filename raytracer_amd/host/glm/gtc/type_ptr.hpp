// glm/gtc/type_ptr.hpp subset: value_ptr / make_mat4 (column-major source).
#pragma once
#include "../glm.hpp"

namespace glm {

template <int N, typename T, qualifier Q> inline const T *value_ptr(const vec<N, T, Q> &v) { return &v.x; }
inline const float *value_ptr(const mat4 &m) { return &m[0].x; }
inline const float *value_ptr(const mat3 &m) { return &m[0].x; }

template <typename T> inline mat4 make_mat4(const T *p) {
    mat4 m(0.0f);
    for (int j = 0; j < 4; ++j) m[j] = vec4(p[j * 4 + 0], p[j * 4 + 1], p[j * 4 + 2], p[j * 4 + 3]);
    return m;
}

}  // namespace glm
