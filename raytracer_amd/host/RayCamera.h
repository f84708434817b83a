// Pinhole camera parameters fed to the ray generator (reference src/RayCamera.h:9-29).
#pragma once

#include <cmath>

#include "glm/glm.hpp"
#include "util.h"

struct RayCamera {
    // Where the camera stands, and where it looks: angles in degrees.  yaw = 90 looks down +z with
    // +x to the right.
    glm::vec3 position = {0, 0, 0};
    f32 yaw = 90;
    f32 pitch = 0;
    f32 fov = 45;   // vertical field of view

    // Orthonormal frame derived from the angles by updateDirection().
    glm::vec3 up = glm::vec3(0, 1, 0);
    glm::vec3 right = glm::vec3(1, 0, 0);
    glm::vec3 forward = glm::vec3(0, 0, 1);

    // Image size in pixels; path length and samples per pixel per frame (the integrators take
    // floor(sqrt(rayPerPixel))^2 stratified samples).
    glm::ivec2 resolution = {1024, 1024};
    i32 bounces = 5;
    i32 rayPerPixel = 1;

    // Rebuilds forward/right/up from yaw and pitch (the x term uses 180 + yaw).
    void updateDirection() {
        const f32 cosPitch = std::cos(glm::radians(pitch));
        const glm::vec3 look(std::cos(glm::radians(180 + yaw)) * cosPitch,
                             std::sin(glm::radians(pitch)),
                             std::sin(glm::radians(yaw)) * cosPitch);
        forward = glm::normalize(look);
        right = glm::normalize(-glm::cross(forward, glm::vec3(0, 1, 0)));
        up = glm::cross(-right, forward);
    }
};
