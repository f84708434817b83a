// Latency of a chain of dependent 64-byte record loads by ONE wave (all lanes follow the same
// random permutation, or each lane its own), on an otherwise idle GPU, per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void k_chain(const uint4 *table, int steps, int perLane, unsigned *out) {
    unsigned at = perLane ? (threadIdx.x * 7919u + blockIdx.x * 104729u) & ((1u << 20) - 1u) : 0u;
    unsigned acc = 0;
    for (int i = 0; i < steps; ++i) {
        const uint4 a = table[(size_t)at * 4], b = table[(size_t)at * 4 + 3];
        at = a.x;
        acc += b.w;
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = at + acc;
}

int main() {
    const size_t recs = 1u << 20;   // 64 MB
    std::vector<unsigned> perm(recs);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<uint4> host(recs * 4);
    for (size_t i = 0; i < recs; ++i) host[(size_t)perm[i] * 4].x = perm[(i + 1) % recs];
    uint4 *table;
    unsigned *out;
    (void)hipMalloc(&table, recs * 64);
    (void)hipMalloc(&out, (size_t)1792 * 256 * sizeof(unsigned));
    (void)hipMemcpy(table, host.data(), recs * 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int blocks : {1, 256, 1792})
        for (int perLane = 0; perLane < 2; ++perLane)
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, table, 40, perLane, out);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) printf("%4d blocks, %s chains of 40 dependent 64-B loads: %.1f us per launch (%.2f us per step)\n", blocks, perLane ? "per-lane" : "uniform ", ms * 1e3, ms * 1e3 / 40);
            }
    return 0;
}
