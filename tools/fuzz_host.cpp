// fuzz_host.cpp — sanitizer driver for everything of the backend and the host layer that runs on the CPU:
// the PNG / JPEG / binary-glTF readers and the commit-time stream validator with the tree builders behind it
// (raytracer_amd/csrc/rt_commit.cpp).  Built by `make asan` (AddressSanitizer + UndefinedBehaviorSanitizer)
// and `make tsan` (ThreadSanitizer) with g++ alone; no GPU, no HIP.
//
// The driver links the host layer's sources directly and stands in for the eleven backend entry points
// the host layer calls (rt_upload captures the streams, rt_commit_scene runs the real commit code on
// them, the render calls report "no device"), so RayScene / RayEngine code paths run unchanged.
//
//   fuzz_host selftest                 every named scene through submit + commit in all five tree modes,
//                                      the threaded tree build with 1, 3 and 8 threads (same trees required)
//   fuzz_host streams  N SEED          N mutations of valid scene streams through the commit
//   fuzz_host png|jpeg N SEED FILE...  N mutations of the given files through the decoder
//   fuzz_host glb      N SEED FILE...  N mutations of the given files through MeshData::LoadMeshData
//
// A run passes when it ends with "ok": every mutated input must come back as a status code or a
// refusal, never as a crash, a sanitizer report or a hang.

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "rt_commit.h"
#include "rt_host.h"

#include "ImageIO.h"
#include "MeshData.h"

// ---- backend stand-in (C ABI of include/rt_hip.h as far as the host layer links it) ----------------
struct rt_ctx : rt_scene_host {
    bool committed = false;
};

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint32_t next() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return (uint32_t)(s >> 16);
    }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

// The commit as rt_commit_scene runs it, for one tree mode.  Returns its status.
int commitOnce(rt_ctx *ctx, int mode) {
    using namespace rtd;
    rtc::PackedNodes nodes;
    rtc::PackedTris tris, wtris;
    rtc::PackedShades shade;
    rtc::PackedUVs uvs;
    std::vector<LightTri> lightTris;
    std::vector<ModelDesc> descs;
    int rc = rtc::repackScene(ctx, nodes, tris, shade, uvs, lightTris, descs);
    if (rc != RT_OK) return rc;
    ctx->wideBvh = mode;
    std::vector<WideNode> wnodes;
    std::vector<Wide8Node> w8;
    std::vector<TriPair> pairs;
    if (mode == 0 || tris.empty()) return RT_OK;
    if (mode == 4 && !ctx->hasAlphaTest) return rtc::buildWide8(ctx, descs, w8, tris, wtris);
    rc = rtc::buildWideNodes(ctx, descs, wnodes, tris, wtris);
    if (rc == RT_OK && !ctx->hasAlphaTest) rtc::pairLeaves(wnodes, descs, wtris, pairs);
    return rc;
}

uint64_t hashBytes(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// Hash of the four-wide trees the library builds for the context's streams.
uint64_t wideTreeHash(rt_ctx *ctx) {
    using namespace rtd;
    rtc::PackedNodes nodes;
    rtc::PackedTris tris, wtris;
    rtc::PackedShades shade;
    rtc::PackedUVs uvs;
    std::vector<LightTri> lightTris;
    std::vector<ModelDesc> descs;
    if (rtc::repackScene(ctx, nodes, tris, shade, uvs, lightTris, descs) != RT_OK) return 0;
    ctx->wideBvh = 3;
    std::vector<WideNode> wnodes;
    if (rtc::buildWideNodes(ctx, descs, wnodes, tris, wtris) != RT_OK) return 0;
    uint64_t h = hashBytes(wnodes.data(), wnodes.size() * sizeof(WideNode));
    return hashBytes(wtris.data(), wtris.size() * sizeof(PackedTri), h);
}

const char *kScenes[] = {"cornell_quads", "roughness_demo", "three_color", "cornell_knot", "knot_glass", "prism",
                         "textured_room", "interior", "mixed", "empty", "unlit_knot", "fuzz", "ties", "panel_floor"};

bool loadScene(rt_ctx *ctx, const char *name, int nu, int nv, int seed) {
    const int32_t ip[5] = {nu, nv, 16, seed, 0};
    rth_scene *s = rth_scene_build(name, ip, 5);
    if (s == nullptr) return false;
    for (int slot = 1; slot <= 5; ++slot) {
        const float *w = nullptr;
        size_t bytes = 0;
        rth_scene_stream(s, slot, &w, &bytes);
        ctx->host[slot].assign(w, w + bytes / 4);
    }
    int32_t o = 0, m = 0, l = 0;
    rth_scene_counts(s, &o, &m, &l);
    ctx->objectCount = o; ctx->modelsCount = m; ctx->lightSourcesCount = l;
    rth_scene_free(s);
    return true;
}

int selftest() {
    int scenes = 0;
    for (const char *name : kScenes) {
        rt_ctx ctx;
        if (!loadScene(&ctx, name, 48, 10, 7)) { std::printf("selftest: scene %s unknown, skipped\n", name); continue; }
        for (int mode = 0; mode <= 4; ++mode) {
            const int rc = commitOnce(&ctx, mode);
            if (rc != RT_OK) { std::printf("selftest: %s mode %d -> %d\n", name, mode, rc); return 1; }
        }
        ++scenes;
    }
    // The threaded build: same trees whatever the thread count (and a clean ThreadSanitizer run).
    rt_ctx big;
    if (!loadScene(&big, "cornell_knot", 512, 64, 3)) return 1;
    uint64_t first = 0;
    for (const char *threads : {"1", "3", "8"}) {
        setenv("RT_BUILD_THREADS", threads, 1);
        const uint64_t h = wideTreeHash(&big);
        if (h == 0 || (first != 0 && h != first)) { std::printf("selftest: trees differ with %s threads\n", threads); return 1; }
        first = h;
    }
    // The host layer end to end on the stand-in backend: RayEngine reports the missing device as a status.
    const float pos[3] = {0.0f, 1.0f, -2.2f};
    rth_engine *e = rth_engine_create("mixed", nullptr, 0, pos, 90.0f, 0.0f, 50.0f, 32, 24, 2, 1, RT_TYPE_RGB, 0);
    if (e != nullptr) {
        const int rc = rth_engine_render(e, 1);
        if (rc != RT_ERR_NO_DEVICE) { std::printf("selftest: render on the stand-in returned %d\n", rc); return 1; }
        rth_engine_free(e);
    }
    std::printf("selftest: %d scenes x 5 tree modes, threaded build x 3 thread counts: ok\n", scenes);
    return 0;
}

void mutateWords(std::vector<float> &w, Rng &rng) {
    if (w.empty()) return;
    static const uint32_t extremes[] = {0u, 1u, 0xffffffffu, 0x7fffffffu, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u,
                                        0x3f800000u, 0x00000001u, 0x7f7fffffu, 1000000u, 0x20000000u, 65536u};
    const int edits = 1 + (int)rng.below(4);
    for (int e = 0; e < edits; ++e) {
        const size_t at = rng.below((uint32_t)w.size());
        uint32_t bits;
        std::memcpy(&bits, &w[at], 4);
        switch (rng.below(5)) {
            case 0: bits ^= 1u << rng.below(32); break;
            case 1: bits = extremes[rng.below(sizeof extremes / sizeof extremes[0])]; break;
            case 2: bits += rng.below(9) - 4; break;
            case 3: bits = rng.next(); break;
            default: if (w.size() > 1) w.resize(1 + rng.below((uint32_t)w.size() - 1)); continue;   // truncate
        }
        if (at < w.size()) std::memcpy(&w[at], &bits, 4);
    }
}

int fuzzStreams(int n, uint64_t seed) {
    Rng rng(seed);
    std::vector<rt_ctx> seeds(4);
    const char *names[4] = {"mixed", "textured_room", "interior", "fuzz"};
    for (int i = 0; i < 4; ++i)
        if (!loadScene(&seeds[(size_t)i], names[i], 24, 6, 5 + i)) return 1;
    int accepted = 0;
    for (int it = 0; it < n; ++it) {
        rt_ctx ctx = seeds[rng.below(4)];
        // Node, identifier and material words are the interesting ones; counts go wrong now and then too.
        const int slot = (rng.below(4) == 0) ? 1 + (int)rng.below(5) : (rng.below(2) ? RT_SLOT_MODELS : RT_SLOT_MODEL_INFO);
        mutateWords(ctx.host[slot], rng);
        if (rng.below(16) == 0) ctx.modelsCount += (int)rng.below(3) - 1;
        if (rng.below(16) == 0) ctx.objectCount += (int)rng.below(3) - 1;
        if (rng.below(16) == 0) ctx.lightSourcesCount += (int)rng.below(3) - 1;
        if (ctx.modelsCount < 0 || ctx.objectCount < 0 || ctx.lightSourcesCount < 0) continue;   // refused by rt_commit_scene itself
        const int rc = commitOnce(&ctx, (int)rng.below(5));
        if (rc == RT_OK) ++accepted;
        else if (rc != RT_ERR_SCENE && rc != RT_ERR_INVALID) { std::printf("streams: iteration %d -> status %d\n", it, rc); return 1; }
    }
    std::printf("streams: %d mutated scenes (seed %llu), %d still valid, the rest refused: ok\n", n, (unsigned long long)seed, accepted);
    return 0;
}

std::vector<unsigned char> readFile(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

void mutateBytes(std::vector<unsigned char> &b, Rng &rng) {
    if (b.empty()) return;
    const int edits = 1 + (int)rng.below(6);
    for (int e = 0; e < edits; ++e) {
        const size_t at = rng.below((uint32_t)b.size());
        switch (rng.below(6)) {
            case 0: b[at] ^= (unsigned char)(1u << rng.below(8)); break;
            case 1: b[at] = (unsigned char)rng.next(); break;
            case 2: b[at] = rng.below(2) ? 0xff : 0x00; break;
            case 3: b.resize(1 + rng.below((uint32_t)b.size())); break;                                   // truncate
            case 4: b.insert(b.begin() + (ptrdiff_t)at, (size_t)(1 + rng.below(8)), (unsigned char)rng.next()); break;
            default: {                                                                                      // overwrite a 4-byte field
                const uint32_t v = rng.below(2) ? 0xffffffffu : rng.next();
                for (size_t k = 0; k < 4 && at + k < b.size(); ++k) b[at + k] = (unsigned char)(v >> (8 * k));
            }
        }
    }
}

int fuzzFiles(const std::string &kind, int n, uint64_t seed, int argc, char **argv) {
    std::vector<std::vector<unsigned char>> seeds;
    for (int i = 0; i < argc; ++i) {
        seeds.push_back(readFile(argv[i]));
        if (seeds.back().empty()) { std::printf("%s: cannot read %s\n", kind.c_str(), argv[i]); return 1; }
    }
    if (seeds.empty()) { std::printf("%s: no seed files\n", kind.c_str()); return 1; }
    Rng rng(seed);
    int decoded = 0;
    const std::string tmp = "/tmp/fuzz_host_" + std::to_string((unsigned long long)seed) + ".glb";
    for (int it = 0; it < n; ++it) {
        std::vector<unsigned char> b = seeds[rng.below((uint32_t)seeds.size())];
        if (it >= (int)seeds.size()) mutateBytes(b, rng);   // the first pass takes the seeds as they are
        if (kind == "glb") {
            std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char *>(b.data()), (std::streamsize)b.size());
            MeshData md = MeshData::LoadMeshData(tmp);
            if (!md.vertices.empty()) ++decoded;
        } else {
            int w = 0, h = 0;
            std::vector<unsigned char> rgba;
            std::string why;
            const bool ok = kind == "png" ? imageio::DecodePngRGBA(b.data(), b.size(), w, h, rgba, why) : imageio::DecodeJpegRGBA(b.data(), b.size(), w, h, rgba, why);
            if (ok) {
                if (w <= 0 || h <= 0 || rgba.size() != (size_t)w * h * 4) { std::printf("%s: iteration %d decoded to an inconsistent image\n", kind.c_str(), it); return 1; }
                ++decoded;
            }
        }
    }
    std::remove(tmp.c_str());
    std::printf("%s: %d inputs from %zu seed files (seed %llu), %d decoded, the rest refused: ok\n", kind.c_str(), n, seeds.size(), (unsigned long long)seed, decoded);
    return 0;
}

}  // namespace

extern "C" {

int rt_create(rt_ctx **out, int) {
    if (out == nullptr) return RT_ERR_INVALID;
    *out = new rt_ctx();
    return RT_OK;
}
void rt_destroy(rt_ctx *ctx) { delete ctx; }
int rt_upload(rt_ctx *ctx, int slot, const void *data, size_t bytes) {
    if (ctx == nullptr || slot < 1 || slot > 5 || (bytes % 4) != 0) return RT_ERR_INVALID;
    const float *w = static_cast<const float *>(data);
    ctx->host[slot].assign(w, w + bytes / 4);
    return RT_OK;
}
int rt_commit_scene(rt_ctx *ctx, int objectCount, int modelsCount, int lightSourcesCount) {
    if (ctx == nullptr || objectCount < 0 || modelsCount < 0 || lightSourcesCount < 0) return RT_ERR_INVALID;
    ctx->objectCount = objectCount; ctx->modelsCount = modelsCount; ctx->lightSourcesCount = lightSourcesCount;
    const int rc = commitOnce(ctx, 3);
    ctx->committed = rc == RT_OK;
    return rc;
}
int rt_resize(rt_ctx *, int, int) { return RT_OK; }
int rt_set_frame(rt_ctx *, const rt_camera *, const float *, int) { return RT_OK; }
int rt_render(rt_ctx *, uint32_t, uint32_t, uint32_t) { return RT_ERR_NO_DEVICE; }
int rt_reset_accum(rt_ctx *) { return RT_OK; }
int rt_readback(rt_ctx *, float *, size_t, uint32_t *) { return RT_ERR_NO_DEVICE; }
const char *rt_strerror(int) { return "stand-in backend"; }
const char *rt_last_hip_error(rt_ctx *) { return ""; }

}  // extern "C"

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "selftest") return selftest();
    if (argc < 4) {
        std::printf("usage: fuzz_host selftest | streams N SEED | png|jpeg|glb N SEED FILE...\n");
        return 2;
    }
    const int n = std::atoi(argv[2]);
    const uint64_t seed = std::strtoull(argv[3], nullptr, 10);
    if (mode == "streams") return fuzzStreams(n, seed);
    if (mode == "png" || mode == "jpeg" || mode == "glb") return fuzzFiles(mode, n, seed, argc - 4, argv + 4);
    return 2;
}
