// comm.hip -- dlopen-bound RCCL wrappers + C ABI of the communicator (see comm.hpp).
#include "comm.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <string>

#include "common.hpp"

using namespace vh;

namespace {

// the few RCCL entry points we need, with their rccl.h prototypes spelled out so that the header
// (and a link-time dependency) is not required
typedef struct { char internal[128]; } UniqueId;
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void**, int, UniqueId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);
typedef int (*fn_comm_count)(const void*, int*);

constexpr int kNcclUint32 = 3;   // ncclUint32
constexpr int kNcclUint64 = 5;   // ncclUint64
constexpr int kNcclFloat32 = 7;  // ncclFloat32 / ncclFloat
constexpr int kNcclFloat64 = 8;  // ncclFloat64 / ncclDouble
constexpr int kNcclSum = 0;      // ncclSum

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    fn_comm_count comm_count = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle) return r;
    std::string tried;
    // library location: the string option comm.rccl_library (vh_set_option_string), then the ROCm installation
    const char* opt = option_string("comm.rccl_library");
    const std::string explicit_path = opt ? opt : "";
    const char* rocm_opt = option_string("comm.rocm_path");
    const std::string rocm = rocm_opt ? rocm_opt : "/opt/rocm";
    const std::string candidates[] = {explicit_path, rocm + "/lib/librccl.so.1", rocm + "/lib/librccl.so",
                                      "/opt/rocm/lib/librccl.so.1"};
    for (const auto& c : candidates) {
        if (c.empty()) continue;
        r.handle = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
        tried += c + " ";
    }
    if (!r.handle) throw InvalidArg{"cannot load RCCL (tried: " + tried + ")"};
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
    r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
    r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
    r.comm_count = (fn_comm_count)dlsym(r.handle, "ncclCommCount");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.all_gather)
        throw InvalidArg{"RCCL library lacks a required symbol"};
    return r;
}

void check_nccl(int rc, const char* what) {
    if (rc == 0) return;
    Rccl& r = rccl();
    const char* msg = r.error_string ? r.error_string(rc) : "?";
    char buf[256];
    snprintf(buf, sizeof(buf), "RCCL error %d (%s) in %s", rc, msg, what);
    throw InvalidArg{std::string(buf)};
}

}  // namespace

namespace vh {

void rccl_unique_id(unsigned char out[128]) {
    UniqueId id;
    check_nccl(rccl().get_unique_id(&id), "ncclGetUniqueId");
    memcpy(out, id.internal, 128);
}

vh_comm* rccl_comm_create(int rank, int world, const unsigned char idbytes[128]) {
    UniqueId id;
    memcpy(id.internal, idbytes, 128);
    vh_comm* c = new vh_comm();
    c->rank = rank;
    c->world = world;
    int rc = rccl().comm_init_rank(&c->nccl_comm, world, id, rank);
    if (rc != 0) {
        delete c;
        check_nccl(rc, "ncclCommInitRank");
    }
    return c;
}

void rccl_comm_destroy(vh_comm* c) {
    if (!c) return;
    if (c->nccl_comm) (void)rccl().comm_destroy(c->nccl_comm);
    if (c->stage) (void)hipHostFree(c->stage);
    delete c;
}

namespace {
// host plane: device buffer -> pinned staging -> callback -> device buffer, all of it synchronous
void* host_stage(vh_comm* c, size_t bytes) {
    if (c->stage_bytes < bytes) {
        if (c->stage) VH_HIP(hipHostFree(c->stage));
        c->stage = nullptr;
        c->stage_bytes = 0;
        VH_HIP(hipHostMalloc(&c->stage, bytes, hipHostMallocDefault));
        c->stage_bytes = bytes;
    }
    return c->stage;
}
void host_allreduce(vh_comm* c, void* buf, size_t count, int dtype, size_t elem, hipStream_t stream) {
    void* st = host_stage(c, count * elem);
    VH_HIP(hipMemcpyAsync(st, buf, count * elem, hipMemcpyDeviceToHost, stream));
    VH_HIP(hipStreamSynchronize(stream));
    const int rc = c->cb_allreduce(c->cb_ctx, st, (int64_t)count, dtype);
    if (rc != 0) throw InvalidArg{"host data plane: the all-reduce callback failed"};
    VH_HIP(hipMemcpyAsync(buf, st, count * elem, hipMemcpyHostToDevice, stream));
    VH_HIP(hipStreamSynchronize(stream));   // the staging buffer is reused by the next collective
}
}  // namespace

void rccl_allreduce_sum_f32(vh_comm* c, float* buf, size_t count, hipStream_t stream) {
    if (c->cb_allreduce) return host_allreduce(c, buf, count, 0, 4, stream);
    check_nccl(rccl().all_reduce(buf, buf, count, kNcclFloat32, kNcclSum, c->nccl_comm, stream), "ncclAllReduce(f32)");
}

void rccl_allreduce_sum_f64(vh_comm* c, double* buf, size_t count, hipStream_t stream) {
    if (c->cb_allreduce) return host_allreduce(c, buf, count, 1, 8, stream);
    check_nccl(rccl().all_reduce(buf, buf, count, kNcclFloat64, kNcclSum, c->nccl_comm, stream), "ncclAllReduce(f64)");
}

void rccl_allreduce_sum_u64(vh_comm* c, unsigned long long* buf, size_t count, hipStream_t stream) {
    if (c->cb_allreduce) return host_allreduce(c, buf, count, 2, 8, stream);
    check_nccl(rccl().all_reduce(buf, buf, count, kNcclUint64, kNcclSum, c->nccl_comm, stream), "ncclAllReduce(u64)");
}

void rccl_allgather_bytes(vh_comm* c, const void* send, void* recv, size_t bytes, hipStream_t stream) {
    if (bytes % 4 != 0) throw InvalidArg{"all-gather blocks must be a multiple of 4 bytes"};
    if (c->cb_allgather) {
        char* st = static_cast<char*>(host_stage(c, bytes * (size_t)(c->world + 1)));
        VH_HIP(hipMemcpyAsync(st, send, bytes, hipMemcpyDeviceToHost, stream));
        VH_HIP(hipStreamSynchronize(stream));
        const int rc = c->cb_allgather(c->cb_ctx, st, st + bytes, (int64_t)bytes);
        if (rc != 0) throw InvalidArg{"host data plane: the all-gather callback failed"};
        VH_HIP(hipMemcpyAsync(recv, st + bytes, bytes * (size_t)c->world, hipMemcpyHostToDevice, stream));
        VH_HIP(hipStreamSynchronize(stream));
        return;
    }
    check_nccl(rccl().all_gather(send, recv, bytes / 4, kNcclUint32, c->nccl_comm, stream), "ncclAllGather");
}

void rccl_allgather_u32(vh_comm* c, const uint32_t* send, uint32_t* recv, size_t count, hipStream_t stream) {
    rccl_allgather_bytes(c, send, recv, count * 4, stream);
}

int comm_reported_ranks(vh_comm* c) {
    if (c->nccl_comm && rccl().comm_count) {
        int n = 0;
        check_nccl(rccl().comm_count(c->nccl_comm, &n), "ncclCommCount");
        return n;
    }
    return c->world;
}

}  // namespace vh

extern "C" {

int vh_comm_unique_id(unsigned char* out128) {
    return guarded([&] {
        VH_REQUIRE(out128 != nullptr, "NULL argument");
        rccl_unique_id(out128);
    });
}

int vh_comm_create(int rank, int world, const unsigned char* id128, vh_comm** out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr && id128 != nullptr, "NULL argument");
        VH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
        *out = rccl_comm_create(rank, world, id128);
    });
}

int vh_comm_destroy(vh_comm* c) {
    return guarded([&] { rccl_comm_destroy(c); });
}

int vh_comm_create_host(int rank, int world, vh_comm_allreduce_fn allreduce, vh_comm_allgather_fn allgather, void* ctx,
                        vh_comm** out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr && allreduce != nullptr && allgather != nullptr, "NULL argument");
        VH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
        vh_comm* c = new vh_comm();
        c->rank = rank;
        c->world = world;
        c->cb_allreduce = allreduce;
        c->cb_allgather = allgather;
        c->cb_ctx = ctx;
        *out = c;
    });
}

int vh_comm_info(vh_comm* c, int* rank, int* world, int* reported_ranks, int* is_rccl) {
    return guarded([&] {
        VH_REQUIRE(c != nullptr, "NULL argument");
        if (rank) *rank = c->rank;
        if (world) *world = c->world;
        if (reported_ranks) *reported_ranks = comm_reported_ranks(c);
        if (is_rccl) *is_rccl = c->nccl_comm != nullptr ? 1 : 0;
    });
}

int vh_device_synchronize(void) {
    return guarded([&] { VH_HIP(hipDeviceSynchronize()); });
}

}  // extern "C"
