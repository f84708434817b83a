// C entry points over the C++ host classes plus the named-scene table shared
// by tests, bench and the smoke check.
#include "rt_host.h"

#include "ImageIO.h"
#include "ProceduralMesh.h"
#include "RayEngine.h"
#include "RaySceneBuilder.h"

#include <cstring>
#include <memory>
#include <new>
#include <string>

namespace {

struct SceneParams {
    int nu = 96, nv = 24, texSize = 64;
    u32 seed = 0xC0FFEEu;
    BVHTree::Mode bvh = BVHTree::defaultMode();
};

// Models built while one of these is alive use the requested tree builder.
struct BvhModeScope {
    BVHTree::Mode saved = BVHTree::defaultMode();
    explicit BvhModeScope(BVHTree::Mode m) { BVHTree::setDefaultMode(m); }
    ~BvhModeScope() { BVHTree::setDefaultMode(saved); }
};

SceneParams parseParams(const int32_t *ip, int n) {
    SceneParams p;
    if (ip != nullptr) {
        if (n > 0 && ip[0] > 0) p.nu = ip[0];
        if (n > 1 && ip[1] > 0) p.nv = ip[1];
        if (n > 2 && ip[2] > 0) p.texSize = ip[2];
        if (n > 3 && ip[3] != 0) p.seed = static_cast<u32>(ip[3]);
        if (n > 4 && ip[4] == 1) p.bvh = BVHTree::Mode::SAH;
        if (n > 4 && ip[4] == 2) p.bvh = BVHTree::Mode::Reference;
    }
    return p;
}

Material knotDiffuse() {
    Material m;
    m.albedo = glm::vec3(.9f, .75f, .7f);
    m.roughness = 0.3f;
    m.metallic = 0.0f;
    return m;
}

Material knotGlass() {
    Material m;
    m.albedo = glm::vec3(.95f, .97f, .95f);
    m.transmission = 1.0f;
    m.ior = 1.5f;
    m.roughness = 0.05f;
    return m;
}

// The 2x2x2 box sits in front of the reference's default camera (0,1,-2.2) looking down +z.
const glm::vec3 kBoxPos{-1.0f, 0.0f, -1.0f};

bool buildNamedScene(RayScene &scene, const std::string &name, const SceneParams &p) {
    if (name == "roughness_demo") {
        RaySceneBuilder::RoughnessMetallicDemo(scene);
        return true;
    }
    if (name == "three_color") {
        RaySceneBuilder::ThreeColorDemo(scene);
        return true;
    }
    if (name == "furnace") {
        // White-furnace check: a uniformly white environment around one sphere whose material is
        // chosen by nu (0 matte white, 1 rough white metal, 2 clear glass, 3 white subsurface).
        scene.setSkyColor(glm::vec3(1.0f));
        Material m;
        m.albedo = glm::vec3(1.0f);
        m.specular = 0.0f;
        if (p.nu == 1) { m.metallic = 1.0f; m.roughness = 0.6f; }
        if (p.nu == 2) { m.transmission = 1.0f; m.roughness = 0.0f; }
        if (p.nu == 3) { m.subsurface = 1.0f; }
        scene.addObject(Sphere{glm::vec3{0.0f, 1.0f, 0.0f}, 0.8f}, m);
        return true;
    }
    if (name == "empty") return true;   // no objects, no models: every ray ends in the default sky
    if (name == "unlit_knot") {         // a model but no light source anywhere: NEE finds nothing, the sky lights it
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{0.0f, 0.95f, 0.1f}, 0.25f, p.seed, knotDiffuse()));
        return true;
    }
    scene.setSkyColor(glm::vec3(0.0f));
    if (name.rfind("glb:", 0) == 0) {
        // "glb:a.glb;b.glb": one addModel(path) per file, in order, under a black sky
        // (the reference's model scene is addModel("cornellBox.glb") + addModel("test1.glb"),
        // src/main.cpp:145-146).  A file that fails to load makes the whole build fail.
        size_t at = 4;
        int loaded = 0;
        while (at <= name.size()) {
            size_t stop = name.find(';', at);
            if (stop == std::string::npos) stop = name.size();
            if (stop > at) {
                const size_t before = scene.getModelsCount();
                scene.addModel(name.substr(at, stop - at));
                if (scene.getModelsCount() == before) return false;
                ++loaded;
            }
            at = stop + 1;
        }
        return loaded > 0;
    }
    if (name == "cornell_quads") {
        RaySceneBuilder::BuildCornellBox(scene, kBoxPos, 2.0f, 0.6f, 15.0f, true);
        Material metal;
        metal.albedo = glm::vec3(.9f, .8f, .4f);
        metal.metallic = 1.0f;
        metal.roughness = 0.2f;
        scene.addObject(Sphere{glm::vec3{-0.45f, 0.35f, 0.3f}, 0.35f}, metal);
        Material glass;
        glass.transmission = 1.0f;
        glass.roughness = 0.02f;
        glass.albedo = glm::vec3(.9f, .95f, 1.0f);
        scene.addObject(Sphere{glm::vec3{0.45f, 0.3f, -0.2f}, 0.3f}, glass);
        Material sss;
        sss.albedo = glm::vec3(.8f, .5f, .4f);
        sss.subsurface = 0.7f;
        scene.addObject(Triangle{glm::vec3{-0.6f, 0.9f, 0.6f}, glm::vec3{0.6f, 0.9f, 0.6f}, glm::vec3{0.0f, 1.7f, 0.8f}}, sss);
        Material tilted;
        tilted.albedo = glm::vec3(.4f, .4f, .9f);
        tilted.roughness = 0.5f;
        RaySceneBuilder::BuildBox(scene, tilted, glm::vec3{0.4f, 0.5f, 0.4f}, glm::vec3{0.1f, 0.25f, 0.45f}, glm::quat(0.9238795f, 0.0f, 0.3826834f, 0.0f));
        return true;
    }
    if (name == "cornell_knot" || name == "knot_glass") {
        scene.addModel(procedural::CornellBox(kBoxPos, 2.0f, 0.7f, 17.0f));
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{0.0f, 0.95f, 0.1f}, 0.25f, p.seed, name == "knot_glass" ? knotGlass() : knotDiffuse()));
        return true;
    }
    if (name == "prism") {
        scene.addModel(procedural::CornellBox(kBoxPos, 2.0f, 0.25f, 4.0f));
        scene.addModel(procedural::LightPanel(glm::vec3{-0.99f, 0.95f, -0.3f}, glm::vec3{0, 0.1f, 0}, glm::vec3{0, 0, 0.6f}, glm::vec3(1.0f), 40.0f));
        scene.addModel(procedural::LightPanel(glm::vec3{0.99f, 0.55f, -0.3f}, glm::vec3{0, 0, 0.6f}, glm::vec3{0, 0.1f, 0}, glm::vec3(1.0f, 0.9f, 0.8f), 40.0f));
        Material glass = knotGlass();
        glass.roughness = 0.0f;
        scene.addModel(procedural::Prism(glm::vec3{0.0f, 0.8f, 0.0f}, 0.8f, 1.0f, glass));
        return true;
    }
    if (name == "textured_room") {
        scene.addModel(procedural::TexturedRoom(kBoxPos, 2.0f, p.texSize, p.seed));
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{0.3f, 0.7f, -0.1f}, 0.12f, p.seed + 1, knotDiffuse()));
        Material mirror;
        mirror.metallic = 1.0f;
        mirror.roughness = 0.05f;
        scene.addObject(Sphere{glm::vec3{-0.5f, 0.3f, -0.3f}, 0.3f}, mirror);
        return true;
    }
    if (name == "interior") {
        // Interior-class stand-in: textured room (base colour, normal, metallic-roughness, emissive,
        // occlusion, alpha-cut and transmission maps of texSize^2 texels), a second emissive panel and
        // four displaced knots with different materials; 8*nu*nv + 16 triangles in six models.
        scene.addModel(procedural::TexturedRoom(kBoxPos, 2.0f, p.texSize, p.seed));
        scene.addModel(procedural::LightPanel(glm::vec3{-0.99f, 0.6f, -0.5f}, glm::vec3{0, 0.6f, 0}, glm::vec3{0, 0, 0.9f}, glm::vec3(1.0f, 0.8f, 0.6f), 5.0f));
        Material metal;
        metal.albedo = glm::vec3(0.95f, 0.8f, 0.5f);
        metal.metallic = 1.0f;
        metal.roughness = 0.2f;
        Material wax;
        wax.albedo = glm::vec3(0.8f, 0.45f, 0.35f);
        wax.subsurface = 0.6f;
        wax.roughness = 0.6f;
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{-0.45f, 0.45f, -0.25f}, 0.12f, p.seed + 11, knotDiffuse()));
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{0.45f, 0.5f, -0.1f}, 0.13f, p.seed + 12, metal));
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{-0.35f, 1.35f, 0.35f}, 0.12f, p.seed + 13, knotGlass()));
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{0.4f, 1.4f, 0.3f}, 0.11f, p.seed + 14, wax));
        return true;
    }
    if (name == "fuzz") {
        // Differential-testing scene: two random soups (nu triangles each, nv materials; the second
        // without lights), random analytic primitives, sky colour from the seed.
        u32 st = p.seed * 747796405u + 1u;
        auto u01 = [&st]() { st = st * 1664525u + 1013904223u; return static_cast<float>(st >> 8) * (1.0f / 16777216.0f); };
        scene.setSkyColor((p.seed & 1u) ? glm::vec3(u01(), u01(), u01()) : glm::vec3(0.0f));
        scene.addModel(procedural::RandomSoup(kBoxPos, 2.0f, p.nu, p.nv, 3, p.seed));
        scene.addModel(procedural::RandomSoup(kBoxPos + glm::vec3(0.3f, 0.2f, 0.4f), 1.4f, p.nu / 2 + 3, p.nv, 0, p.seed + 99u));
        for (int k = 0; k < 5; ++k) {
            Material m;
            m.albedo = glm::vec3(u01(), u01(), u01());
            m.roughness = u01();
            m.metallic = u01() < 0.5f ? 0.0f : 1.0f;
            m.transmission = u01() < 0.3f ? 1.0f : 0.0f;
            if (k == 0) { m.emissionColor = glm::vec3(1.0f); m.emissionStrength = 5.0f; }
            const glm::vec3 c = kBoxPos + glm::vec3(u01(), u01(), u01()) * 2.0f;
            if (k % 3 == 0) scene.addObject(Sphere{c, 0.1f + 0.3f * u01()}, m);
            else if (k % 3 == 1) scene.addObject(Quad{c, glm::vec3{0.5f * u01(), 0, 0.2f}, glm::vec3{0, 0.5f * u01(), 0.1f}}, m);
            else scene.addObject(Triangle{c, c + glm::vec3{0.4f, 0.1f, 0}, c + glm::vec3{0, 0.5f, 0.3f}}, m);
        }
        return true;
    }
    if (name == "panel_floor") {
        // One emissive rectangle (a model, so next-event estimation samples it) facing a large floor
        // quad: the direct light a camera ray sees on the floor has a quadrature answer.  nu = 1..10: tenths
        // of roughness (else the default material's 1.0); nv = 100..200: hundredths of metallic plus 100.
        scene.addModel(procedural::LightPanel(glm::vec3{-0.3f, 1.5f, 0.8f}, glm::vec3{0.6f, 0, 0}, glm::vec3{0, 0, 0.4f}, glm::vec3(1.0f, 0.8f, 0.6f), 5.0f));
        Material floor;
        floor.albedo = glm::vec3(0.8f, 0.7f, 0.6f);
        if (p.nu > 0 && p.nu <= 10) floor.roughness = static_cast<float>(p.nu) / 10.0f;
        if (p.nv >= 100 && p.nv <= 200) floor.metallic = static_cast<float>(p.nv - 100) / 100.0f;
        scene.addObject(Quad{glm::vec3{-5.0f, 0.0f, -5.0f}, glm::vec3{10.0f, 0, 0}, glm::vec3{0, 0, 10.0f}}, floor);
        return true;
    }
    if (name == "ties") {
        // A lit room whose floor is covered by nv coplanar copies of an nu x nu grid of squares:
        // every hit on it is a distance tie between triangles of one model.
        scene.addModel(procedural::CornellBox(kBoxPos, 2.0f, 0.5f, 12.0f, false, glm::vec3(0.85f)));
        scene.addModel(procedural::CoplanarStack(kBoxPos + glm::vec3(0.2f, 0.35f, 0.2f), 1.6f, p.nu > 0 ? p.nu : 6, p.nv > 0 ? p.nv : 4, p.seed));
        return true;
    }
    if (name == "mixed") {
        // Non-light model first, then two light models: exercises the light-first
        // ordering of submit() and multi-model light selection.
        scene.addModel(procedural::KnotTube(p.nu, p.nv, glm::vec3{-0.35f, 0.6f, 0.2f}, 0.11f, p.seed, knotDiffuse()));
        scene.addModel(procedural::CornellBox(kBoxPos, 2.0f, 0.5f, 12.0f, false, glm::vec3(0.85f)));
        scene.addModel(procedural::LightPanel(glm::vec3{-0.99f, 0.4f, -0.4f}, glm::vec3{0, 0.5f, 0}, glm::vec3{0, 0, 0.8f}, glm::vec3(0.4f, 0.6f, 1.0f), 6.0f));
        scene.addModel(procedural::KnotTube(p.nu / 2 > 3 ? p.nu / 2 : 3, p.nv, glm::vec3{0.45f, 1.2f, 0.3f}, 0.09f, p.seed + 7, knotGlass()));
        Material gold;
        gold.albedo = glm::vec3(1.0f, 0.77f, 0.34f);
        gold.metallic = 1.0f;
        gold.roughness = 0.3f;
        scene.addObject(Sphere{glm::vec3{0.4f, 0.3f, -0.4f}, 0.3f}, gold);
        Material shell;
        shell.albedo = glm::vec3(.7f, .3f, .3f);
        shell.subsurface = 0.5f;
        shell.specularTint = 0.5f;
        scene.addObject(Quad{glm::vec3{-0.9f, 0.02f, -0.9f}, glm::vec3{0.8f, 0, 0}, glm::vec3{0, 0, 0.8f}}, shell);
        return true;
    }
    return false;
}

// No exception leaves the C boundary: allocation failures become RT_ERR_NOMEM (or a null handle),
// anything else RT_ERR_INVALID.
template <typename Body>
int guarded(Body body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return RT_ERR_NOMEM;
    } catch (...) {
        return RT_ERR_INVALID;
    }
}
template <typename T, typename Body>
T *guardedHandle(Body body) {
    try {
        return body();
    } catch (...) {
        return nullptr;
    }
}

}  // namespace

struct rth_scene {
    RayScene scene;
};

struct rth_engine {
    RayEngine engine;
    int frames = 0;
};

extern "C" {

rth_scene *rth_scene_build(const char *name, const int32_t *iparams, int n) {
    if (name == nullptr) return nullptr;
    return guardedHandle<rth_scene>([&]() -> rth_scene * {
        auto holder = std::make_unique<rth_scene>();
        holder->scene.initialize(nullptr);
        const SceneParams params = parseParams(iparams, n);
        BvhModeScope scope(params.bvh);
        if (!buildNamedScene(holder->scene, name, params)) return nullptr;
        holder->scene.submit();
        return holder.release();
    });
}

void rth_scene_free(rth_scene *scene) { delete scene; }

int rth_scene_counts(const rth_scene *s, int32_t *objectCount, int32_t *modelsCount, int32_t *lightSourcesCount) {
    if (s == nullptr) return RT_ERR_INVALID;
    if (objectCount) *objectCount = static_cast<int32_t>(s->scene.getObjectsCount());
    if (modelsCount) *modelsCount = static_cast<int32_t>(s->scene.getModelsCount());
    if (lightSourcesCount) *lightSourcesCount = s->scene.getLightSourcesCount();
    return RT_OK;
}

int rth_scene_sky(const rth_scene *s, float sky[3]) {
    if (s == nullptr || sky == nullptr) return RT_ERR_INVALID;
    const glm::vec3 c = s->scene.getLinearSkyColor();
    sky[0] = c.x; sky[1] = c.y; sky[2] = c.z;
    return RT_OK;
}

int rth_scene_stream(const rth_scene *s, int slot, const float **words, size_t *bytes) {
    if (s == nullptr || words == nullptr || bytes == nullptr || slot < 1 || slot > 5) return RT_ERR_INVALID;
    const std::vector<f32> &w = s->scene.streamWords(slot);
    *words = w.data();
    *bytes = w.size() * sizeof(f32);
    return RT_OK;
}

int rth_scene_model_stats(const rth_scene *s, int model, int32_t out[10]) {
    if (s == nullptr || out == nullptr || model < 0 || model >= static_cast<int>(s->scene.models().size())) return RT_ERR_INVALID;
    const Model &m = *s->scene.models()[model];
    Model::TreeStats t;
    const int rc = guarded([&]() -> int { t = m.treeStats(); return RT_OK; });
    if (rc != RT_OK) return rc;
    out[0] = static_cast<int32_t>(m.bvh.getNodes().size());
    out[1] = static_cast<int32_t>(m.meshData.identifiers.size());
    out[2] = static_cast<int32_t>(m.meshData.vertices.size());
    out[3] = m.meshData.lightSourcesCount;
    out[4] = t.minHeight;
    out[5] = t.maxHeight;
    out[6] = t.leafCount;
    out[7] = t.minLeafTris;
    out[8] = t.maxLeafTris;
    out[9] = t.valid ? 1 : 0;
    return RT_OK;
}

static RayCamera makeCamera(const float position[3], float yaw, float pitch, float fov, int w, int h, int bounces, int rpp) {
    RayCamera cam;
    cam.position = glm::vec3(position[0], position[1], position[2]);
    cam.yaw = yaw;
    cam.pitch = pitch;
    cam.fov = fov;
    cam.resolution = glm::ivec2(w, h);
    cam.bounces = bounces;
    cam.rayPerPixel = rpp;
    cam.updateDirection();
    return cam;
}

int rth_camera_make(rt_camera *out, const float position[3], float yaw, float pitch, float fov, int w, int h, int bounces, int rpp) {
    if (out == nullptr || position == nullptr) return RT_ERR_INVALID;
    const RayCamera cam = makeCamera(position, yaw, pitch, fov, w, h, bounces, rpp);
    out->fov = cam.fov;
    out->resolution[0] = static_cast<float>(w);
    out->resolution[1] = static_cast<float>(h);
    for (int k = 0; k < 3; ++k) {
        out->position[k] = cam.position[k];
        out->forward[k] = cam.forward[k];
        out->right[k] = cam.right[k];
        out->up[k] = cam.up[k];
    }
    out->bounces = bounces;
    out->rayPerPixel = rpp;
    return RT_OK;
}

rth_engine *rth_engine_create(const char *scene_name, const int32_t *iparams, int n, const float position[3], float yaw, float pitch,
                              float fov, int w, int h, int bounces, int rpp, int type, int device) {
    if (scene_name == nullptr || position == nullptr || w <= 0 || h <= 0) return nullptr;
    return guardedHandle<rth_engine>([&]() -> rth_engine * {
        auto holder = std::make_unique<rth_engine>();
        const RayCamera cam = makeCamera(position, yaw, pitch, fov, w, h, bounces, rpp);
        if (!holder->engine.initialize(cam, type == RT_TYPE_SPECTRAL ? RayTracer::Type::Spectral : RayTracer::Type::RGB, device)) return nullptr;
        const SceneParams params = parseParams(iparams, n);
        BvhModeScope scope(params.bvh);
        if (!buildNamedScene(holder->engine.getScene(), scene_name, params)) return nullptr;
        holder->engine.getScene().submit();
        return holder.release();
    });
}

int rth_engine_render(rth_engine *e, int frames) {
    if (e == nullptr || frames < 0) return RT_ERR_INVALID;
    return guarded([&]() -> int {
        e->engine.render(static_cast<u32>(frames));
        const int rc = e->engine.getRayTracer().status();   // device out of memory, refused scene ...: handed on, not trapped
        if (rc == RT_OK) e->frames += frames;
        return rc;
    });
}

int rth_engine_frame(rth_engine *e, float *rgba, size_t floats) {
    if (e == nullptr || rgba == nullptr) return RT_ERR_INVALID;
    return guarded([&]() -> int {
        RayFrame &f = e->engine.getRayTracer().getCurrentFrame();
        if (e->engine.getRayTracer().status() != RT_OK) return e->engine.getRayTracer().status();
        if (floats != f.rgba.size()) return RT_ERR_INVALID;
        std::memcpy(rgba, f.rgba.data(), floats * sizeof(float));
        return e->frames;
    });
}

void rth_engine_free(rth_engine *e) { delete e; }

int rth_write_png(const char *path, const uint8_t *rgba8, int width, int height) {
    if (path == nullptr || rgba8 == nullptr) return RT_ERR_INVALID;
    return guarded([&]() -> int { return imageio::WritePNG(path, rgba8, width, height) ? RT_OK : RT_ERR_INVALID; });
}

int rth_write_pfm(const char *path, const float *rgba, int width, int height, float scale) {
    if (path == nullptr || rgba == nullptr) return RT_ERR_INVALID;
    return guarded([&]() -> int { return imageio::WritePFM(path, rgba, width, height, scale) ? RT_OK : RT_ERR_INVALID; });
}

}  // extern "C"
