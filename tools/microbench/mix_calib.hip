// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the traversal kernel's own MIX of accesses, with
// known byte counts.  One "ray" of k_mix<V, P> does what a ray of k_trace does to memory:
//   reads a 4-byte queue entry and a 32-byte ray record   (streams: consecutive lanes, consecutive records)
//   V gathers of a 64-byte node, four 16-byte loads        (per-lane scattered records of a 2 GiB table)
//   P gathers of an 80-byte triangle pair, five 16-byte loads (per-lane scattered, records straddle sectors)
//   writes a 16-byte hit record                            (stream)
// Every record of the two tables is read at most once (the record number is a bijection of the access
// number), and the tables are far larger than the L2s and the Infinity Cache, so every byte must cross the
// L2 -> fabric boundary: the program prints, per kernel, the bytes requested and the bytes of the 32-byte
// sectors those requests touch.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- tools/microbench/mix_calib      (and once more with WRITE_SIZE)
// tools/roofline_report.py compares the counters with these figures.  Three mixes: the kernel's access mix
// on the default workload (11 nodes and 2.5 pairs per ray), a leaf-heavy one as the misses are (the top of the
// tree stays in the caches) and a stream-heavy one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr unsigned kLogNodes = 25;   // 32 Mi 64-byte records = 2 GiB
constexpr unsigned kLogPairs = 24;   // 16 Mi 80-byte records = 1.25 GiB

template <int V, int P>
__global__ __launch_bounds__(256) void k_mix(const float4 *nodes, const float4 *pairs, const unsigned *queue, const float4 *rays, float4 *hits, unsigned n) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned slot = queue[i];
    const float4 r0 = rays[2 * (size_t)slot], r1 = rays[2 * (size_t)slot + 1];
    float acc = r0.x + r1.w;
#pragma unroll 1
    for (int v = 0; v < V; ++v) {
        const unsigned rec = ((i * (unsigned)V + (unsigned)v) * 2654435761u) & ((1u << kLogNodes) - 1u);
        const float4 *p = nodes + (size_t)rec * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
#pragma unroll 1
    for (int q = 0; q < P; ++q) {
        const unsigned rec = ((i * (unsigned)P + (unsigned)q) * 2246822519u) & ((1u << kLogPairs) - 1u);
        const float4 *p = pairs + (size_t)rec * 5;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
        acc += a.x + b.y + c.z + d.w + e.x;
    }
    hits[slot] = make_float4(acc, 0.0f, 0.0f, 0.0f);
}

template <int V, int P>
void run(const float4 *nodes, const float4 *pairs, const unsigned *queue, const float4 *rays, float4 *hits) {
    // as many rays as use every node record at most once
    const unsigned n = (unsigned)(((size_t)1 << kLogNodes) / (V > 0 ? V : 1)) & ~255u;
    hipLaunchKernelGGL((k_mix<V, P>), dim3(n / 256), dim3(256), 0, 0, nodes, pairs, queue, rays, hits, n);
    // 80-byte records start at multiples of 80 bytes: 32-byte sectors touched = those between the first and last byte
    unsigned long long pairSectors = 0;
    for (unsigned k = 0; k < 8; ++k) {   // the pattern repeats every 8 records (80 * 8 = 640 = 20 sectors)
        const unsigned off = k * 80u;
        pairSectors += (off + 79u) / 32u - off / 32u + 1u;
    }
    const double sectorBytesPerPair = 32.0 * (double)pairSectors / 8.0;
    const double gatherRequested = (double)n * (V * 64.0 + P * 80.0), gatherSectors = (double)n * (V * 64.0 + P * sectorBytesPerPair);
    const double streamRead = (double)n * 36.0, streamWritten = (double)n * 16.0;
    printf("k_mix<%d, %d> rays %u gather_requested %.0f gather_sectors %.0f stream_read %.0f stream_written %.0f\n", V, P, n, gatherRequested, gatherSectors,
           streamRead, streamWritten);
}

int main() {
    float4 *nodes, *pairs, *rays, *hits;
    unsigned *queue;
    const size_t maxRays = (size_t)1 << kLogNodes;   // upper bound of n over the mixes below (V >= 1)
    if (hipMalloc(&nodes, ((size_t)64 << kLogNodes)) != hipSuccess || hipMalloc(&pairs, ((size_t)80 << kLogPairs)) != hipSuccess ||
        hipMalloc(&rays, maxRays * 32) != hipSuccess || hipMalloc(&hits, maxRays * 16) != hipSuccess || hipMalloc(&queue, maxRays * 4) != hipSuccess)
        return 1;
    hipMemset(nodes, 0, (size_t)64 << kLogNodes);
    hipMemset(pairs, 0, (size_t)80 << kLogPairs);
    hipMemset(rays, 0, maxRays * 32);
    // queue[i] = i: the streams are read and written in order, as the first bounce's are
    unsigned *h = (unsigned *)malloc(maxRays * 4);
    for (size_t i = 0; i < maxRays; ++i) h[i] = (unsigned)i;
    hipMemcpy(queue, h, maxRays * 4, hipMemcpyHostToDevice);
    free(h);
    run<11, 2>(nodes, pairs, queue, rays, hits);   // access mix of the default workload (11.2 nodes, 4.9 triangles = 2.5 pairs per ray)
    run<4, 2>(nodes, pairs, queue, rays, hits);    // leaf-heavy: what is left after the caches have served the top of the tree
    run<2, 1>(nodes, pairs, queue, rays, hits);    // stream-heavy
    run<1, 0>(nodes, pairs, queue, rays, hits);    // one node per ray: streams are a third of the bytes
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return 0;
}
