// Throughput of per-lane 64-byte record gathers (the traversal kernel's node fetch) in two shapes:
//   A  every lane loads its own record with four 16-byte loads (64 different lines per instruction)
//   B  the four lanes of a quad load one record together, 16 bytes each (16 lines per instruction),
//      four instructions cover the quad's four records
// Records are picked pseudo-randomly from a table of `tableKB` kilobytes (first argument).  Independent iterations:
// this measures issue/tag throughput, not dependent latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

template <int SHAPE>
__global__ __launch_bounds__(256) void k_gather(const float4 *table, uint32_t mask, int iters, float *out) {
    uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 9781u + 1u;
    float acc = 0.0f;
    const uint32_t lane = threadIdx.x & 63u;
    for (int it = 0; it < iters; ++it) {
        seed = pcg(seed);
        const uint32_t rec = seed & mask;
        if (SHAPE == 0) {
            const float4 *p = table + (size_t)rec * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else {
            float4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t other = __shfl(rec, (int)((lane & ~3u) + j));
                r[j] = table[(size_t)other * 4 + (lane & 3u)];
            }
            acc += r[0].x + r[1].y + r[2].z + r[3].w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const int tableKB = argc > 1 ? atoi(argv[1]) : 32768;
    const int iters = 2000;
    size_t recs = (size_t)tableKB * 1024 / 64;
    uint32_t mask = 1;
    while ((size_t)mask * 2 <= recs) mask *= 2;
    mask -= 1;
    float4 *table;
    float *out;
    hipMalloc(&table, ((size_t)mask + 1) * 64);
    hipMemset(table, 0, ((size_t)mask + 1) * 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int blocksPerCu : {4, 7, 8}) {
        const int grid = cus * blocksPerCu;
        hipMalloc(&out, (size_t)grid * 256 * 4);
        for (int shape = 0; shape < 2; ++shape) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, table, mask, iters, out);
                else hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, table, mask, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double records = (double)grid * 256 * iters;
            printf("table %d KB, %d blocks/CU, shape %c: %.3f ms, %.2f G records/s, %.2f TB/s, %.2f lane-loads/ns/CU\n", tableKB, blocksPerCu,
                   shape ? 'B' : 'A', ms, records / ms / 1e6, records * 64 / ms / 1e9, records * 4 / ms / 1e6 / cus);
        }
        hipFree(out);
    }
    return 0;
}
