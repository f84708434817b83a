// Headless counterpart of the reference's program (reference src/main.cpp:120-215): the same
// camera, engine and scene calls, with the window loop replaced by a fixed number of
// progressive frames and the screen quad by a PNG of the display pass.
//
//   rt_render [--rgb] [--reference-order|--wide|--wide2] [--frames N] [--size WxH] [--bounces B] [--scene cornell|roughness|three]
//             [--out file.png] [model.glb ...]
//   scenes: cornell (mesh Cornell box + glass and metal spheres), roughness and three
//   (RaySceneBuilder::RoughnessMetallicDemo / ThreeColorDemo)
//
// With .glb arguments the scene is built like the reference's: black sky + one addModel per file.
#include "ImageIO.h"
#include "ProceduralMesh.h"
#include "RayEngine.h"
#include "RaySceneBuilder.h"
#include "rt_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    int width = 1280, height = 720, frames = 64, bounces = 5;
    RayTracer::Type type = RayTracer::Type::Spectral;   // the reference's default integrator
    // Trees the backend walks (RT_OPT_WIDE_BVH): -1 the library's default, 0 the uploaded trees in the
    // reference's order, 1 four-wide trees from the uploaded ones, 2 from trees the library builds itself
    int wide = -1;
    std::string sceneName = "cornell", out = "render.png";
    std::vector<std::string> models;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--rgb") type = RayTracer::Type::RGB;
        else if (a == "--reference-order") wide = 0;
        else if (a == "--wide") wide = 1;
        else if (a == "--wide2") wide = 2;
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--bounces") bounces = std::atoi(next());
        else if (a == "--size") std::sscanf(next(), "%dx%d", &width, &height);
        else if (a == "--scene") sceneName = next();
        else if (a == "--out") out = next();
        else models.push_back(a);
    }

    RayCamera camera;
    camera.rayPerPixel = 1;
    camera.bounces = bounces;
    camera.fov = 50;
    camera.resolution = {width, height};
    camera.pitch = 0;
    camera.position = {0, 1, -2.2f};
    camera.updateDirection();

    RayEngine rayEngine;
    if (!rayEngine.initialize(camera, type)) {
        std::fprintf(stderr, "no usable GPU: the HIP backend has no CPU fallback\n");
        return 1;
    }

    {
        auto &scene = rayEngine.getScene();
        if (!models.empty()) {
            scene.setSkyColor({});
            for (const std::string &path : models) scene.addModel(path);
            if (scene.getModelsCount() != models.size()) {
                std::fprintf(stderr, "a model failed to load\n");
                return 1;
            }
        } else if (sceneName == "roughness") {
            RaySceneBuilder::RoughnessMetallicDemo(scene);
        } else if (sceneName == "three") {
            RaySceneBuilder::ThreeColorDemo(scene);
        } else {
            // Triangle-mesh Cornell box (its lamp is a model light, so next-event estimation finds it;
            // RaySceneBuilder::BuildCornellBox makes analytic quads, which are only ever hit by chance).
            scene.setSkyColor({});
            scene.addModel(procedural::CornellBox({-1, 0, -1}, 2.0f, 0.6f, 15.0f));
            Material glass;
            glass.transmission = 1.0f;
            glass.roughness = 0.0f;
            scene.addObject(Sphere{glm::vec3{0.35f, 0.4f, -0.1f}, 0.4f}, glass);
            Material metal;
            metal.metallic = 1.0f;
            metal.roughness = 0.15f;
            metal.albedo = glm::vec3(0.9f, 0.8f, 0.5f);
            scene.addObject(Sphere{glm::vec3{-0.45f, 0.3f, 0.35f}, 0.3f}, metal);
        }
        scene.submit();   // submit scene to the GPU
    }

    // Backend switch with no counterpart in the reference: which trees the rays walk.
    if (wide >= 0 && rt_set_option(rayEngine.getRayTracer().backend(), RT_OPT_WIDE_BVH, wide) != RT_OK) return 1;
    rayEngine.render(static_cast<u32>(frames));   // accumulate
    if (rayEngine.getRayTracer().status() != RT_OK) {
        std::fprintf(stderr, "render failed: %s\n", rt_strerror(rayEngine.getRayTracer().status()));
        return 1;
    }

    rt_ctx *ctx = rayEngine.getRayTracer().backend();
    std::vector<uint8_t> rgba8(static_cast<size_t>(width) * height * 4);
    if (rt_display_rgba8(ctx, rgba8.data(), rgba8.size()) != RT_OK || !imageio::WritePNG(out, rgba8.data(), width, height)) {
        std::fprintf(stderr, "could not write %s\n", out.c_str());
        return 1;
    }
    rt_stats st;
    rt_get_stats(ctx, &st);
    std::printf("%d frames, %.1f Mrays, %.1f ms of device time -> %s\n", frames, st.rays / 1e6, st.total_ms, out.c_str());
    return 0;
}
