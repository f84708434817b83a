// Padding arithmetic follows reference src/AABB.cpp:4-16 exactly: the tree's
// boxes (and therefore which nodes a ray opens) depend on it.
#include "AABB.h"
#include <algorithm>

AABB::AABB(glm::vec3 lo, glm::vec3 hi) {
    const glm::vec3 pad(static_cast<float>(1e-6));
    min = lo - pad;
    max = hi + pad;
}

AABB::AABB(const AABB &a, const AABB &b) {
    for (int k = 0; k < 3; ++k) {
        min[k] = static_cast<float>(static_cast<double>(std::min(a.min[k], b.min[k])) - 1e-6);
        max[k] = static_cast<float>(static_cast<double>(std::max(a.max[k], b.max[k])) + 1e-6);
    }
}
