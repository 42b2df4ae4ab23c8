// comm.hpp -- RCCL communicator for the data-parallel training path (one process per GPU).
//
// RCCL is bound at run time with dlopen("/opt/rocm/lib/librccl.so.1"): the library must use the
// RCCL that is linked against the SAME HIP runtime as libvambhip (PyTorch-ROCm wheels bundle private
// copies of libamdhip64 / librccl whose streams and allocations are not interchangeable with ours),
// and libvambhip must stay loadable on machines without a GPU.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

struct vh_comm {
    void* nccl_comm = nullptr;
    int rank = 0;
    int world = 1;
};

namespace vh {

// throws HipError / InvalidArg on failure
void rccl_unique_id(unsigned char out[128]);
vh_comm* rccl_comm_create(int rank, int world, const unsigned char id[128]);
void rccl_comm_destroy(vh_comm* c);
// in-place sum all-reduce on `stream` (float32 or float64 elements)
void rccl_allreduce_sum_f32(vh_comm* c, float* buf, size_t count, hipStream_t stream);
void rccl_allreduce_sum_f64(vh_comm* c, double* buf, size_t count, hipStream_t stream);
// exact integer accumulators of the row-sharded cluster scan
void rccl_allreduce_sum_u64(vh_comm* c, unsigned long long* buf, size_t count, hipStream_t stream);
// every rank contributes `count` 32-bit words; recv holds world * count words in rank order
void rccl_allgather_u32(vh_comm* c, const uint32_t* send, uint32_t* recv, size_t count, hipStream_t stream);

}  // namespace vh
