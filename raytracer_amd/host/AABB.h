// Axis-aligned box with the reference's padding rule (reference src/AABB.h:4-10).
#pragma once
#include "glm/glm.hpp"

struct AABB {
    AABB() = default;
    // Box around [lo, hi], grown by 1e-6 per side in f32 arithmetic.
    AABB(glm::vec3 lo, glm::vec3 hi);
    // Union of two boxes, grown AGAIN by 1e-6 per side; the pad is added in
    // double and narrowed to f32, so repeated unions grow the box cumulatively.
    AABB(const AABB &a, const AABB &b);

    glm::vec3 min;   // lower corner (padded)
    glm::vec3 max;   // upper corner (padded)
};
