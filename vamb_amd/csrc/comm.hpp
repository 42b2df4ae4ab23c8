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

// Host data plane (vh_comm_create_host): the same collectives through two caller-supplied functions operating on HOST memory --
// what a torch.distributed gloo group offers.  Every rccl_* wrapper below then synchronises the stream, stages the buffer
// through the host and calls them.  It exists so that the multi-rank paths (data-parallel training, the row-sharded cluster
// sweep and its native state machine) can be RUN and checked on one GPU with several processes (RCCL refuses two ranks on one
// device) and on machines whose RCCL cannot be loaded; it is slow by construction and never chosen when RCCL is available.
//   allreduce(ctx, buf, count, dtype): in-place sum over the ranks; dtype 0 = float32, 1 = float64, 2 = uint64
//   allgather(ctx, send, recv, bytes): recv = the ranks' `bytes`-byte blocks in rank order
typedef int (*vh_comm_allreduce_fn)(void* ctx, void* buf, int64_t count, int dtype);
typedef int (*vh_comm_allgather_fn)(void* ctx, const void* send, void* recv, int64_t bytes);

struct vh_comm {
    void* nccl_comm = nullptr;
    int rank = 0;
    int world = 1;
    vh_comm_allreduce_fn cb_allreduce = nullptr;
    vh_comm_allgather_fn cb_allgather = nullptr;
    void* cb_ctx = nullptr;
    void* stage = nullptr;       // pinned staging buffer of the host plane
    size_t stage_bytes = 0;
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
// every rank contributes `bytes` bytes (a multiple of 4); recv holds world * bytes in rank order
void rccl_allgather_bytes(vh_comm* c, const void* send, void* recv, size_t bytes, hipStream_t stream);
// ranks of the communicator as the collective library itself reports them (ncclCommCount), or `world` on the host plane
int comm_reported_ranks(vh_comm* c);

}  // namespace vh
