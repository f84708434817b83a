// rt_comm.h — the RCCL side of rt_comm_unique_id / rt_comm_init / rt_reduce (rt_comm.cpp, plain C++).
#pragma once

#include <cstddef>
#include <string>

namespace rtcomm {

constexpr size_t kUniqueIdBytes = 128;   // sizeof(ncclUniqueId) == RT_COMM_ID_BYTES

// Status codes are those of rt_hip.h.  `error` receives the failing call and RCCL's message.
int uniqueId(void *id, size_t bytes);
int init(void **comm, int rank, int world, const void *id, size_t bytes, std::string &error);
void destroy(void *comm);
// Sum of `floats` fp32 values into `root` plus an all-reduce of one uint64 counter, both queued on `stream`
// (a hipStream_t).  Buffers are device memory.
int reduceSum(void *comm, float *accum, size_t floats, unsigned long long *counter, int root, void *stream, std::string &error);

}  // namespace rtcomm
