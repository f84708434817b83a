/* rt_hip.h — C ABI of the MI355X path-tracing backend (librt_hip.so).
 *
 * This is the boundary a maintainer of the reference renderer binds instead of
 * its OpenGL path: flat float streams in, accumulated RGBA32F frames out.
 * Plain pointers and sizes only; every call returns 0 (RT_OK) or a negative
 * rt_status; no exceptions cross the boundary.  A context is owned by one
 * caller thread at a time and uses one HIP stream on one device.
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the reference tree).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument (null pointer, size mismatch, bad slot) */
    RT_ERR_STATE = -2,     /* call out of order (render before commit/resize/set_frame) */
    RT_ERR_HIP = -3,       /* a HIP runtime call failed; rt_last_hip_error() has the text */
    RT_ERR_NO_DEVICE = -4, /* no usable GPU */
    RT_ERR_SCENE = -5,     /* stream contents inconsistent (offsets out of range, depth > stack) */
    RT_ERR_NOMEM = -6
} rt_status;

/* Stream slots: the texture-unit numbers of src/RayScene.cpp:30-52. */
enum {
    RT_SLOT_OBJECTS = 1,    /* objectsBuffer          src/RayScene.cpp:32 */
    RT_SLOT_MATERIALS = 2,  /* materialsBuffer        src/RayScene.cpp:36 */
    RT_SLOT_MODEL_INFO = 3, /* modelInfoObjectsBuffer src/RayScene.cpp:40 */
    RT_SLOT_MODELS = 4,     /* modelObjectsBuffer     src/RayScene.cpp:44 */
    RT_SLOT_TEXTURES = 5    /* texturesBuffer         src/RayScene.cpp:48 */
};

/* Integrator selector: RayTracer::Type, src/RayTracer.h:17-20. */
enum { RT_TYPE_RGB = 0, RT_TYPE_SPECTRAL = 1 };

/* The `camera.*` uniform block of src/RayTracer.cpp:1441-1448. */
typedef struct rt_camera {
    float fov;           /* degrees */
    float resolution[2]; /* width, height (as floats, like the vec2 uniform) */
    float position[3];
    float forward[3];
    float right[3];
    float up[3];
    int32_t bounces;     /* loop runs i = 0..bounces inclusive */
    int32_t rayPerPixel; /* ssq = int(sqrt(rayPerPixel)); ssq*ssq samples per frame */
} rt_camera;

/* Work counters of the last rt_render/rt_trace_primary calls since rt_reset_stats. */
typedef struct rt_stats {
    uint64_t rays;          /* hit() calls: path segments + NEE shadow rays */
    uint64_t camera_samples;/* camera samples started */
    uint64_t node_visits;   /* inner BVH nodes opened (only if counters enabled) */
    uint64_t tri_tests;     /* ray/triangle tests (only if counters enabled) */
    uint64_t shaded_hits;   /* hits whose material was resolved */
    double trace_ms;        /* device time in the traversal kernel (HIP events) */
    double total_ms;        /* device time of rt_render as a whole, summed over calls (calls that overlap on the device count in full each) */
    uint64_t trace_launches;
    uint64_t max_stack;     /* most traversal-stack entries any ray held (only if counters enabled) */
    uint64_t retraced;      /* RT_OPT_WIDE_BVH 3 and 4: rays traced a second time, in the reference's order, after a distance tie */
    /* Occupancy of the traversal kernel's lanes (only if counters enabled).  A wave of 64 lanes runs
     * one phase per step, inner nodes or leaves; a lane does work in a step if it opens a node or a
     * leaf in it: lane utilisation = (node_visits + leaf_visits) / (64 * wave_steps). */
    uint64_t wave_steps;        /* loop steps of all waves that ran a phase */
    uint64_t active_lane_steps; /* lanes holding a ray, summed over those steps */
    uint64_t leaf_visits;       /* leaves opened */
    uint64_t inner_phase_steps; /* steps that ran the inner-node phase */
    uint64_t refills;           /* times a wave fetched new rays for its idle lanes */
    uint64_t shadow_rays;       /* the NEE shadow rays among `rays` (the rest are path segments) */
} rt_stats;

/* Creates a context on HIP device `device`.  Replaces the GL context +
 * gl::TextureBuffer objects owned by gpu::TBOStorage (src/gpu/TBOStorage.cpp:5-7). */
int rt_create(rt_ctx **out, int device);
void rt_destroy(rt_ctx *ctx);

/* gpu::Storage::upload (src/gpu/Storage.h:7; glBufferData in
 * src/glUtilities/TextureBuffer.cpp:23-38): copies `bytes` of 32-bit words;
 * the caller may free `data` on return.  bytes may be 0. */
int rt_upload(rt_ctx *ctx, int slot, const void *data, size_t bytes);

/* RayScene::bindShader + the scene uniforms (src/RayScene.cpp:30-52,
 * src/RayTracer.cpp:1436-1437): fixes the counts and re-packs the uploaded
 * streams into the traversal layout (64-byte inner nodes, 48-byte triangles).  Every offset the
 * kernels follow is validated first (RT_ERR_SCENE otherwise): sections inside their streams, vertex /
 * material / texture references in range, trees at most 32 levels deep with children after their
 * parent and one parent per node, leaves of at most 3 triangles, fewer than 2^29 triangles in all. */
int rt_commit_scene(rt_ctx *ctx, int objectCount, int modelsCount, int lightSourcesCount);

/* RayTracer::initialize / changeResolution frame textures
 * (src/RayTracer.cpp:1415-1424, 1454-1465).  Zeroes the accumulation buffer. */
int rt_resize(rt_ctx *ctx, int width, int height);

/* Per-frame uniforms (src/RayTracer.cpp:1438-1448).  `type` = RT_TYPE_*. */
int rt_set_frame(rt_ctx *ctx, const rt_camera *cam, const float sky_linear[3], int type);

/* RayTracer::renderToTexture + the full-screen draw (src/RayTracer.cpp:1428-1452,
 * src/RayEngine.cpp:28-42) for frameCount = first_frame .. first_frame+n_frames-1
 * in steps of frame_stride (sample-split sharding uses stride = number of GPUs).
 * Adds each frame's colour into the sum buffer.  Asynchronous: the call queues its kernels and
 * returns.  Calls of few paths (up to 1 Mi: one progressive frame of a viewer, as the reference's main
 * loop issues them, src/main.cpp:163-202) are queued on three streams in turn, each with path
 * buffers of its own, so that consecutive calls overlap on the device; the passes that add to the sum
 * buffer are chained in call order, so the buffer holds the same bits as without the overlap
 * (RT_OPT_PIPELINE).  Every other entry point that reads or writes the sum buffer takes its place in
 * that order too. */
int rt_render(rt_ctx *ctx, uint32_t first_frame, uint32_t n_frames, uint32_t frame_stride);

/* Allocates now the per-path device memory that rt_render(first, n_frames, stride) would need
 * (no reference counterpart: the reference's per-fragment state lives in shader registers).  rt_render
 * grows the buffers on demand, which puts a hipMalloc into the first call of a new size; a caller
 * that wants its render calls free of allocation, or wants an out-of-memory answer before it starts,
 * calls this after rt_set_frame.  Same ceiling as rt_render (RT_OPT_MAX_PATHS, half of the free device
 * memory): a request above it is served in several batches, as it would be by rt_render. */
int rt_reserve(rt_ctx *ctx, uint32_t n_frames);

/* RayTracer::reset (src/RayTracer.cpp:1467-1470): zero the sum buffer and frame count. */
int rt_reset_accum(rt_ctx *ctx);

/* Blocks until all work the context has queued, on any of its streams, is complete. */
int rt_sync(rt_ctx *ctx);

/* Reads the sum buffer (W*H*4 floats, row 0 = bottom row as gl_FragCoord) and the
 * number of frames added.  mean = sum / frames is what getCurrentFrame()
 * (src/RayTracer.h:39) holds in the reference.  Synchronises. */
int rt_readback(rt_ctx *ctx, float *rgba_sum, size_t floats, uint32_t *frames);

/* The screen pass of src/main.cpp:36-69 applied to the progressive mean: clamp at 0,
 * ACES filmic curve, red/green/blue debug colours for NaN/negative/infinite, gamma 1/2.2.
 * Writes W*H RGBA8 pixels (row 0 = bottom row).  Synchronises. */
int rt_display_rgba8(rt_ctx *ctx, uint8_t *rgba8, size_t bytes);

/* rt_display_rgba8 for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc, code that runs between the captures of a HIP graph; the call itself launches on the
 * context's own stream and waits for it, so it cannot be part of a capture).  No counterpart in the
 * reference.  Same result, but the
 * RGBA8 frame is staged in a fixed 32 MiB array that every GPU the library runs on holds until the
 * process ends, instead of a buffer allocated per call: no hipMalloc/hipFree happens here.  Frames
 * of more than 8,388,608 pixels (3840x2160 fits) are refused with RT_ERR_INVALID and a message in
 * rt_last_hip_error.  `rgba8` may be host memory (pinned memory keeps the runtime's own copy path
 * free of staging) or device memory.  All contexts on one GPU share the array: calls that overlap
 * on one GPU race.  Synchronises. */
int rt_display_rgba8_noalloc(rt_ctx *ctx, uint8_t *rgba8, size_t bytes);

/* The screen pass without the wait, for a viewer that shows every frame as the reference's main loop does
 * (src/main.cpp:163-202: render one frame, draw the tonemapped quad, swap; in GL both are queued commands
 * and the loop runs ahead of the device).  rt_display_rgba8_begin queues the pass and the copy into `rgba8`
 * behind every frame queued so far and returns; frames queued afterwards overlap with them, except that
 * their additions to the sum buffer wait until the pass has read it.  rt_display_rgba8_end(slot) blocks
 * until `rgba8` of that slot's last begin holds the frame.  Four slots (0..3), each with a staging frame
 * on the device (allocated at a slot's first use at a resolution); a slot that was begun must be ended
 * before it is begun again (RT_ERR_STATE otherwise).  `rgba8` must stay valid until the end call and
 * should be pinned host memory or device memory: a copy into pageable memory makes begin wait for it.
 * rt_resize completes the frames in flight (their buffers hold the old size's frames) and frees every slot: an
 * end call for a slot begun before the resize then returns RT_ERR_STATE.
 * A viewer keeps two or three frames in flight:
 *     rt_render(ctx, f, 1, 1); rt_display_rgba8_begin(ctx, f % 3, buf[f % 3], bytes);
 *     if (f >= 2) { rt_display_rgba8_end(ctx, (f - 2) % 3); present(buf[(f - 2) % 3]); }                */
int rt_display_rgba8_begin(rt_ctx *ctx, int slot, uint8_t *rgba8, size_t bytes);
int rt_display_rgba8_end(rt_ctx *ctx, int slot);

/* Checkpoint/resume of the progressive state (no reference counterpart: the reference keeps
 * it only in two GPU textures and m_frameCount, src/RayTracer.h:24-27): uploads a sum buffer
 * previously obtained from rt_readback together with its frame count. */
int rt_restore_accum(rt_ctx *ctx, const float *rgba_sum, size_t floats, uint32_t frames);

/* Device pointer of the sum buffer, for an in-place RCCL reduce by the caller. */
int rt_accum_device_ptr(rt_ctx *ctx, void **dev_ptr, size_t *floats);
/* Makes the context accumulate into caller-owned device memory (W*H*4 floats). */
int rt_bind_accum(rt_ctx *ctx, void *dev_ptr, size_t floats);
/* Adds `frames` to the frame count after an external reduce merged other ranks' sums. */
int rt_add_frames(rt_ctx *ctx, uint32_t frames);

/* Multi-GPU merge for callers without torch.distributed (no reference counterpart: the reference
 * is single-GPU).  One context per process per GPU.  Rank 0 obtains a 128-byte id with
 * rt_comm_unique_id and hands it to the other ranks by its own means; every rank calls
 * rt_comm_init(ctx, rank, world, id, 128) (collective).  rt_reduce (collective) sums the ranks'
 * sum buffers onto `root` with one RCCL reduce on the context's stream, adds up the frame counts,
 * and clears the other ranks' buffers so that a later merge adds only new frames.  It blocks until
 * done.  RCCL (librccl.so.1) is loaded on first use; RT_ERR_STATE if it cannot be. */
int rt_comm_unique_id(void *id, size_t bytes);
int rt_comm_init(rt_ctx *ctx, int rank, int world, const void *id, size_t bytes);
int rt_reduce(rt_ctx *ctx, int root);

/* Tile-split sharding (no reference counterpart): the context renders only the 8x8
 * pixel tiles whose row-major index i has i % stride == offset; other pixels of its
 * sum buffer stay untouched (zero after rt_resize / rt_reset_accum), so the shards'
 * buffers add up to the full image.  stride = 1 renders everything (default). */
int rt_set_tile_shard(rt_ctx *ctx, uint32_t offset, uint32_t stride);

/* Parity hook (no reference counterpart: the reference never outputs IDs): closest
 * hit of the first camera sample's ray of frame `frame`, sample index 0, per pixel.
 * model = index in the model-info stream or -1; ident = identifier index inside
 * that model (leaf order) or, when model == -2, the primitive index; t,u,v as
 * hitTriangle computes them (src/RayScene.cpp:540-544).  Host arrays of W*H. */
int rt_trace_primary(rt_ctx *ctx, uint32_t frame, int32_t *model, int32_t *ident,
                     float *t, float *u, float *v);

/* Parity hook (no reference counterpart): evaluates the Spectral integrator's colour functions on the
 * device exactly as the kernels use them, `items` inputs at a time (host arrays).  fn 0: get_cie_xyz
 * (src/RayTracer.cpp:871-891), in = lambda, out = xyz; fn 1: smits_eval (:920-934), in = lambda,
 * spectrum type, out = value; fn 2: get_reflectance (:936-978), in = lambda, r, g, b, out = value;
 * fn 3: xyz_to_rgb(chromatic_adapt_E_to_D65(xyz)) (:1297-1315), in = xyz, out = rgb; fn 4: the tables
 * the kernels read, CIE x, y, z (41 each) then the 70 Smits values, 193 floats per item, `in` unused. */
int rt_eval_colour(rt_ctx *ctx, int fn, const float *in, size_t items, float *out);

int rt_get_stats(rt_ctx *ctx, rt_stats *out);
int rt_reset_stats(rt_ctx *ctx);
/* Enables per-ray node/triangle visit counting (slower; off by default). */
int rt_set_counters(rt_ctx *ctx, int enabled);

/* Switches.  Only RT_OPT_SPECTRAL_PDF_AS_WRITTEN, RT_OPT_HERO_WAVELENGTH, RT_OPT_NEE,
 * RT_OPT_STRATIFIED and RT_OPT_WIDE_BVH (values 1 and 2) can change output values; RT_OPT_HERO_WAVELENGTH,
 * RT_OPT_NEE and RT_OPT_STRATIFIED are the reference's shader-compile-time switches.
 *   RT_OPT_SHADOW_EARLY_OUT  RGB shadow rays stop at the first hit nearer than
 *                            distToLight - 1e-4: such a hit already fails the reference's
 *                            visibility test (src/RayTracer.cpp:439), so finishing the
 *                            closest-hit search cannot change the result.  Default 1.
 *                            With 0 every ray does the reference's full closest-hit walk
 *                            (the visit counters then equal the reference algorithm's).
 *   RT_OPT_REFILL_LANES      idle lanes that trigger a refill in the traversal kernel (1..64)
 *   RT_OPT_INNER_SHARE       share (x/64) of active lanes on inner nodes that selects the
 *                            inner-node phase (1..64)
 *   RT_OPT_MAX_PATHS         ceiling of paths in flight per batch (whole sample planes, at least
 *                            one; about 190 bytes of device memory per path).  Default 128 Mi,
 *                            further limited to half of the free device memory.  A render call
 *                            that needs fewer paths allocates only what it needs.
 *   RT_OPT_SPECTRAL_PDF_AS_WRITTEN   The Spectral shader text does not link against the shared scene
 *                            code (it calls a 2-argument light sampler, src/RayTracer.cpp:783,1150,
 *                            while src/RayScene.cpp:900 defines three).  0 (default): the sampler's
 *                            triangle count is used and the area pdf is 1/area/lightSources/
 *                            lightTriangles as in the RGB integrator (src/RayTracer.cpp:363-365,
 *                            427).  1: pdf = 1/area, literally as src/RayTracer.cpp:1082,1165 write it.
 *   RT_OPT_TILE_MAJOR        order of the paths inside a batch: 1 (default) keeps all sample
 *                            planes of an 8x8 pixel tile adjacent, 0 orders plane by plane
 *   RT_OPT_WIDE_BVH          Which trees the rays walk.  Default 3.  Changing it rebuilds the
 *                            derived trees of a committed scene at once.
 *                            0: the uploaded binary trees, in the order of hitModel
 *                            (src/RayScene.cpp:681-736).
 *                            1: a four-wide tree with 8-bit child boxes that the library derives
 *                            from each uploaded binary tree (same leaves, same triangles).  Every
 *                            ray still gets its closest hit, but the trees are not visited in
 *                            hitModel's order, so where two triangles of one model are hit at the
 *                            same distance (shared edges, coplanar duplicates) the other one may
 *                            win.
 *                            2: as 1, but the four-wide trees are collapsed from binary trees the
 *                            library builds itself over each model's triangles (binned
 *                            surface-area splits) instead of the uploaded median-split trees.  Same
 *                            closest hits, same tie caveat, about a third of the node visits and
 *                            half of the triangle tests of the reference order.
 *                            3: as 2, without the tie caveat: a ray that met two triangles of one
 *                            model at its closest distance is traced a second time by the
 *                            reference-order kernel over the uploaded trees, whose answer replaces
 *                            the first (rt_stats.retraced counts such rays; 4 bytes of device
 *                            memory per ray slot hold the list).
 *                            4: as 3 on eight-wide trees (80-byte nodes, children visited by the
 *                            ray's direction octant, no sorting): two thirds of the node visits of 3
 *                            but about twice the arithmetic per node, 7-19 % slower than 3 on every
 *                            workload measured; kept as an option.  Scenes with alpha-cut
 *                            textures walk the four-wide trees under this value.
 *                            The uploaded node section is validated for every value.  Values 2
 *                            to 4 need finite coordinates within 1e15: left at the default such a
 *                            scene is walked as with 0; chosen by this option or by the
 *                            environment variable RT_WIDE_BVH the commit fails with RT_ERR_SCENE.
 *                            Scenes without triangles are not affected.
 *   RT_OPT_HERO_WAVELENGTH   1: the Spectral integrator built with HERO_WAVELENGTH_ENABLE 1
 *                            (src/RayTracer.cpp:607, 1257-1280, 1357-1364): every camera sample traces
 *                            four wavelengths, spread over the range from one random offset, along
 *                            the same camera ray and each from the same seed (it is passed by value,
 *                            src/RayTracer.cpp:1051), and adds the mean of their XYZ contributions.  Four
 *                            times the bounce loop per sample; the camera ray the four share is traced
 *                            once and its hit record kept for the other three (16 bytes of device
 *                            memory per path; rt_stats.rays counts the rays traced, three per camera
 *                            sample fewer than the shader's).  Default 0, the reference's default.
 *                            Ignored by RT_TYPE_RGB.
 *   RT_OPT_NEE               0: both integrators as built with ENABLE_NEE 0 (src/RayTracer.cpp:24,
 *                            606): no light sampling and no shadow rays, every emitter hit counts
 *                            in full (:357-376, :420-481, :1077-1091, :1147-1203).  Default 1.
 *   RT_OPT_STRATIFIED        0: the RGB shader's `#else` sampling branch (src/RayTracer.cpp:584-593):
 *                            rayPerPixel unstratified samples per frame (not floor(sqrt)^2), jitter
 *                            over one pixel, plain mean.  Default 1 (`#if 1`, :569-583).  The
 *                            Spectral shader has no such branch and ignores it.  Call rt_set_frame
 *                            again after changing it.
 *   RT_OPT_PIPELINE          1 (default): small render calls overlap on the device, see rt_render.
 *                            0: every call runs on the context's one stream, after the call before
 *                            it (also: environment variable RT_PIPELINE=0).  Same output either way.
 *   RT_OPT_TEXTURE_U8        1 (default): a four-channel texture whose uploaded words all equal (float)k / 255.0f
 *                            for k = 0..255 -- what the reference's loader makes of an 8-bit image,
 *                            src/MeshData.cpp:556-587 -- is kept on the device as one RGBA8 word per texel, a
 *                            quarter of the bytes; fetches (src/RayScene.cpp:337-380) return the same floats, bit for
 *                            bit, through a 256-entry table.  Textures no material refers to are not copied at
 *                            all.  0 keeps the uploaded words.  Re-commits a committed scene.  Env RT_TEX_U8.
 *   RT_OPT_FAST_SHADE        1: the shading and light-connection kernels (traceColor's arithmetic,
 *                            src/RayTracer.cpp:119-541 and :700-1255) run a build that uses the hardware's one-instruction
 *                            reciprocal, square root, exponential, logarithm, sine and cosine and fused multiply-adds
 *                            instead of the pinned, correctly rounded sequences.  Camera rays and traversal are
 *                            untouched: hit ids, t, u, v stay bit-exact; radiance then agrees with the default
 *                            (and the checker) only within a per-pixel relative RMSE (2 % at 64 spp is the bar the
 *                            tests hold it to).  Default 0.  Env RT_FAST_SHADE.
 *   RT_OPT_TREE_BUILD        RT_TREE_BUILD_DEVICE: the trees of RT_OPT_WIDE_BVH 2 to 4 are built on the device
 *                            from the uploaded model stream -- triangles sorted along a Morton curve, neighbouring
 *                            clusters of that order merged bottom-up by least surface, and for values 2 and 3 the
 *                            four-wide nodes and their leaves made there as well, level by level, so that only counts
 *                            travel to the host (value 4 takes the device's binary trees through the host's collapse).
 *                            RT_TREE_BUILD_HOST (default): the host threads' binned surface-area trees.  The
 *                            reference builds its trees on the CPU when a model is loaded (src/BVHTree.cpp:6-85);
 *                            that uploaded tree is never replaced, both settings only choose who builds the
 *                            library's own.  Same images bit for bit (closest hits do not depend on the tree;
 *                            ties are re-traced under value 3 and 4 as before).  Measured: commit of 872 k triangles
 *                            0.31 s -> 0.05 s, of 3.5 M triangles 0.73 s -> 0.15 s; tracing 0.4 to 1.5 % slower
 *                            (6 % more node visits).  Re-commits a committed scene.  Env RT_TREE_BUILD=device.
 */
enum { RT_OPT_SHADOW_EARLY_OUT = 1, RT_OPT_REFILL_LANES = 2, RT_OPT_INNER_SHARE = 3, RT_OPT_MAX_PATHS = 4, RT_OPT_TILE_MAJOR = 5,
       RT_OPT_SPECTRAL_PDF_AS_WRITTEN = 6, RT_OPT_WIDE_BVH = 7, RT_OPT_HERO_WAVELENGTH = 8, RT_OPT_NEE = 9,
       RT_OPT_STRATIFIED = 10, RT_OPT_PIPELINE = 11, RT_OPT_TEXTURE_U8 = 12, RT_OPT_FAST_SHADE = 13, RT_OPT_TREE_BUILD = 14 };
enum { RT_TREE_BUILD_HOST = 0, RT_TREE_BUILD_DEVICE = 1 };
int rt_set_option(rt_ctx *ctx, int option, int value);

/* Facts about the committed scene as the device holds it (no reference counterpart: the reference's
 * TBO upload, src/glUtilities/TextureBuffer.cpp:23-38, reports nothing).
 *   RT_INFO_TEXTURE_BYTES_UPLOADED  bytes of the texture stream given to rt_upload
 *   RT_INFO_TEXTURE_BYTES_DEVICE    bytes of the texture buffer the kernels read (RT_OPT_TEXTURE_U8)
 *   RT_INFO_TEXTURES_RGBA8 / _FP32  textures in use kept as RGBA8 words / as fp32 words
 *   RT_INFO_TREE_MODE               RT_OPT_WIDE_BVH value in effect for this scene (0 when the default was declined for it) */
enum { RT_INFO_TEXTURE_BYTES_UPLOADED = 1, RT_INFO_TEXTURE_BYTES_DEVICE = 2, RT_INFO_TEXTURES_RGBA8 = 3, RT_INFO_TEXTURES_FP32 = 4,
       RT_INFO_TREE_MODE = 5 };
int rt_get_info(rt_ctx *ctx, int what, uint64_t *value);

/* Identifies the build of the library: a hash over its kernel and host sources and compiler flags,
 * made by the Makefile.  Profiles record it, so that a measurement can be told from one taken on
 * other code (no reference counterpart). */
const char *rt_build_id(void);

const char *rt_strerror(int status);
const char *rt_last_hip_error(rt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
