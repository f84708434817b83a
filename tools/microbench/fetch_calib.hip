// Calibration of rocprofv3's FETCH_SIZE for the access shapes of the traversal kernel.  Every record of
// a table much larger than the caches is read exactly once, in a scattered order, so the bytes that
// must cross the L2 -> fabric boundary are known: records x record size.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/microbench/fetch_calib
// and compare FETCH_SIZE of each kernel with the byte count this program prints for it.
//   k_calib<64>  per-lane 64-byte records, four 16-byte loads (a tree node)
//   k_calib<48>  per-lane 48-byte records, three 16-byte loads (a triangle; records straddle lines)
//   k_calib<16>  16 bytes per lane, consecutive lanes consecutive addresses (a streamed array)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int BYTES>
__global__ __launch_bounds__(256) void k_calib(const float4 *table, unsigned logRecords, float *out) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    // multiplication by an odd number is a bijection of [0, 2^logRecords): every record exactly once
    const unsigned rec = BYTES == 16 ? i : (i * 2654435761u) & ((1u << logRecords) - 1u);
    const float4 *p = table + (size_t)rec * (BYTES / 16);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < BYTES / 16; ++k) {
        const float4 v = p[k];
        acc += v.x + v.w;
    }
    if (acc == 123.456f) out[0] = acc;   // never true for a zeroed table: keeps the loads alive
}

int main() {
    const unsigned logRecords = 25;   // 32 Mi records: 2 GiB of 64-byte records, far beyond L2 and the Infinity Cache
    const size_t records = (size_t)1 << logRecords;
    float4 *table;
    float *out;
    if (hipMalloc(&table, records * 64) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(table, 0, records * 64);
    const unsigned grid = (unsigned)(records / 256);
    hipLaunchKernelGGL(k_calib<64>, dim3(grid), dim3(256), 0, 0, table, logRecords, out);
    hipLaunchKernelGGL(k_calib<48>, dim3(grid), dim3(256), 0, 0, table, logRecords, out);
    hipLaunchKernelGGL(k_calib<16>, dim3(grid), dim3(256), 0, 0, table, logRecords, out);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("k_calib<64> bytes %zu\nk_calib<48> bytes %zu\nk_calib<16> bytes %zu\n", records * 64, records * 48, records * 16);
    return 0;
}
