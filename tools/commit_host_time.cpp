// Times the commit-time host code (stream validation, re-pack, tree builds of rt_commit.cpp) on a scene whose five
// streams were written to files, without a device:  commit_host_time DIR  (DIR/slot1.bin .. slot5.bin, DIR/counts.txt
// holding "objects models lightSources").  RT_COMMIT_TIMING=1 prints the laps; RT_BUILD_THREADS sets the threads.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "rt_commit.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    rt_scene_host ctx;
    for (int s = 1; s <= 5; ++s) {
        std::ifstream in(dir + "/slot" + std::to_string(s) + ".bin", std::ios::binary | std::ios::ate);
        if (!in) continue;
        const size_t bytes = (size_t)in.tellg();
        in.seekg(0);
        ctx.host[s].resize(bytes / 4);
        in.read(reinterpret_cast<char *>(ctx.host[s].data()), (std::streamsize)bytes);
    }
    std::ifstream counts(dir + "/counts.txt");
    counts >> ctx.objectCount >> ctx.modelsCount >> ctx.lightSourcesCount;
    for (int round = 0; round < 3; ++round) {
        rtc::PackedNodes nodes;
        rtc::PackedTris tris, wtris;
        rtc::PackedShades shade;
        rtc::PackedUVs uvs;
        std::vector<rtd::LightTri> lightTris;
        std::vector<rtd::ModelDesc> descs;
        std::vector<rtd::WideNode> wnodes;
        std::vector<rtd::TriPair> wpairs;
        auto t0 = std::chrono::steady_clock::now();
        const int rc = rtc::repackScene(&ctx, nodes, tris, shade, uvs, lightTris, descs);
        auto t1 = std::chrono::steady_clock::now();
        int rc2 = 0;
        if (argc > 2) {
            rc2 = rtc::buildWideNodes(&ctx, descs, wnodes, tris, wtris);
            if (!ctx.hasAlphaTest) rtc::pairLeaves(wnodes, descs, wtris, wpairs);
        }
        auto t2 = std::chrono::steady_clock::now();
        std::printf("re-pack %d: %.1f ms (%zu triangles, %zu nodes); wide trees %d: %.1f ms\n", rc, std::chrono::duration<double, std::milli>(t1 - t0).count(), tris.size(),
                    nodes.size(), rc2, std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    return 0;
}
