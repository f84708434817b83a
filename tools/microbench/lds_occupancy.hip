// Reports how many blocks with a given static LDS footprint the runtime places on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out) {
    __shared__ float buf[BYTES / 4];
    buf[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    out[blockIdx.x * THREADS + threadIdx.x] = buf[(threadIdx.x * 7) % (BYTES / 4)];
}

template <int BYTES, int THREADS>
void report() {
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES, THREADS>, THREADS, 0);
    printf("LDS %6d B/block, %4d threads: %d blocks/CU (%s) -> %d B, %d waves\n", BYTES, THREADS, n, hipGetErrorString(e), n * BYTES, n * THREADS / 64);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerMultiprocessor %zu, regsPerMultiprocessor %d\n", p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerMultiprocessor, p.regsPerMultiprocessor);
    report<20480, 256>();
    report<24576, 256>();
    report<16384, 256>();
    report<53248, 512>();
    report<49152, 512>();
    report<40960, 512>();
    report<32768, 512>();
    report<65536, 1024>();
    report<65536, 512>();
    return 0;
}
