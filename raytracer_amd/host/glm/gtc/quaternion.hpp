// glm/gtc/quaternion.hpp subset: quaternion -> rotation matrix.
#pragma once
#include "../glm.hpp"

namespace glm {

inline mat3 mat3_cast(const quat &q) {
    mat3 r(1.0f);
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
    float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    r[0].x = 1.0f - 2.0f * (qyy + qzz);
    r[0].y = 2.0f * (qxy + qwz);
    r[0].z = 2.0f * (qxz - qwy);
    r[1].x = 2.0f * (qxy - qwz);
    r[1].y = 1.0f - 2.0f * (qxx + qzz);
    r[1].z = 2.0f * (qyz + qwx);
    r[2].x = 2.0f * (qxz + qwy);
    r[2].y = 2.0f * (qyz - qwx);
    r[2].z = 1.0f - 2.0f * (qxx + qyy);
    return r;
}

inline mat4 mat4_cast(const quat &q) {
    mat3 r = mat3_cast(q);
    mat4 m(1.0f);
    for (int j = 0; j < 3; ++j) m[j] = vec4(r[j], 0.0f);
    return m;
}

}  // namespace glm
