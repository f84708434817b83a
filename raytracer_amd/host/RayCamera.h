// Pinhole camera parameters fed to the ray generator (reference src/RayCamera.h:9-29).
#pragma once

#include "glm/glm.hpp"
#include "util.h"
#include <cmath>

struct RayCamera {
    glm::vec3 position = {0, 0, 0};
    f32 yaw = 90, pitch = 0, fov = 45;  // degrees

    glm::vec3 up = glm::vec3(0, 1, 0), right = glm::vec3(1, 0, 0), forward = glm::vec3(0, 0, 1);

    glm::ivec2 resolution = {1024, 1024};
    i32 bounces = 5, rayPerPixel = 1;

    // Rebuilds forward/right/up from yaw and pitch; yaw = 90 looks down +z
    // with +x to the right (the x term uses 180 + yaw).
    void updateDirection() {
        const f32 cp = std::cos(glm::radians(pitch));
        glm::vec3 dir;
        dir.x = std::cos(glm::radians(180 + yaw)) * cp;
        dir.y = std::sin(glm::radians(pitch));
        dir.z = std::sin(glm::radians(yaw)) * cp;
        forward = glm::normalize(dir);
        right = glm::normalize(-glm::cross(forward, glm::vec3(0, 1, 0)));
        up = glm::cross(-right, forward);
    }
};
