// glm/gtc/matrix_transform.hpp subset: translate / scale (post-multiplying, as glm).
#pragma once
#include "../glm.hpp"
#include "quaternion.hpp"

namespace glm {

inline mat4 translate(const mat4 &m, const vec3 &v) {
    mat4 r = m;
    r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}

inline mat4 scale(const mat4 &m, const vec3 &v) {
    mat4 r = m;
    r[0] = m[0] * v.x;
    r[1] = m[1] * v.y;
    r[2] = m[2] * v.z;
    return r;
}

}  // namespace glm
