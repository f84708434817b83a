/* rt_host.h — C entry points of librt_host.so, the C++ host layer that keeps the
 * reference's scene API (RayScene / RaySceneBuilder / BVHTree / RayEngine).
 * They exist so that non-C++ callers (the Python tests and bench.py) can build
 * the reference's scenes, read the five serialised streams and drive RayEngine.
 * They are conveniences above the backend ABI in rt_hip.h, not part of it. */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;
typedef struct rth_engine rth_engine;

/* Builds one of the named scenes into a host-only RayScene (streams kept on the
 * host) and submits it.  Names: "cornell_quads", "roughness_demo", "three_color",
 * "cornell_knot", "knot_glass", "prism", "textured_room", "interior", "mixed", "empty", "unlit_knot", "fuzz" (random soup: nu triangles, nv materials), and
 * "glb:<a.glb>[;<b.glb>...]" = one RayScene::addModel(path) per binary glTF file under a black sky
 * (NULL if a file cannot be loaded).
 * iparams: [0]=knot nu, [1]=knot nv, [2]=texture size, [3]=seed (0 = defaults),
 * [4]=tree builder: 0 = process default (reference median split unless RT_BVH_MODE=sah),
 * 1 = binned SAH (opt-in, not the reference's tree), 2 = reference median split. */
rth_scene *rth_scene_build(const char *name, const int32_t *iparams, int n_iparams);
void rth_scene_free(rth_scene *scene);

int rth_scene_counts(const rth_scene *scene, int32_t *objectCount, int32_t *modelsCount, int32_t *lightSourcesCount);
int rth_scene_sky(const rth_scene *scene, float sky_linear[3]);
/* Borrowed pointer to the words submitted for `slot` (RT_SLOT_*); valid until rth_scene_free. */
int rth_scene_stream(const rth_scene *scene, int slot, const float **words, size_t *bytes);
/* out[10] = nodes, triangles, vertices, lightTriangles, minHeight, maxHeight, leaves, minLeafTris, maxLeafTris, valid */
int rth_scene_model_stats(const rth_scene *scene, int model, int32_t out[10]);

/* RayCamera{position,yaw,pitch,fov,resolution,bounces,rayPerPixel}.updateDirection() -> uniform block. */
int rth_camera_make(rt_camera *out, const float position[3], float yaw, float pitch, float fov,
                    int width, int height, int bounces, int rayPerPixel);

/* RayEngine driven end to end on HIP device `device`: initialize + named scene + submit. */
rth_engine *rth_engine_create(const char *scene_name, const int32_t *iparams, int n_iparams,
                              const float position[3], float yaw, float pitch, float fov,
                              int width, int height, int bounces, int rayPerPixel, int type, int device);
/* RayEngine::render(frames).  0, or the rt_status the backend returned (device out of memory, a scene or
 * camera it refuses ...): the host layer hands the code on instead of trapping.  No entry point
 * here lets a C++ exception out; allocation failures read RT_ERR_NOMEM or a NULL handle. */
int rth_engine_render(rth_engine *engine, int frames);
/* Progressive mean, width*height*4 floats; returns frames rendered so far or < 0. */
int rth_engine_frame(rth_engine *engine, float *rgba, size_t floats);
void rth_engine_free(rth_engine *engine);

/* File output of a frame (no reference counterpart: the reference only draws to its window).
 * rgba8 as rt_display_rgba8 returns it; rgba as rt_readback returns it, scaled by `scale`
 * (1/frames gives the progressive mean).  Row 0 = bottom row in both.  0 or RT_ERR_INVALID. */
int rth_write_png(const char *path, const uint8_t *rgba8, int width, int height);
int rth_write_pfm(const char *path, const float *rgba, int width, int height, float scale);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
