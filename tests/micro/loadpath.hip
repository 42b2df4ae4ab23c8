// loadpath.hip -- developer micro-benchmark (not product code): how fast can one CU pull bf16 operand tiles of the
// VAE GEMMs out of L2, by path?  Mirrors gemm_bf16.hpp's staging pattern (8 waves, 128 x 128 tile, K-tile 64: a wave
// instruction moves 8 rows x 128 B) without any MFMA, so the number is the ceiling of the K loop's operand feed.
//   mode 0  LDS-DMA (global_load_lds_dwordx4) for both operands            [what the GEMM does today]
//   mode 1  global_load_dwordx4 into registers, no LDS write               [L2 -> VGPR ceiling]
//   mode 2  global_load_dwordx4 + ds_write_b128 for both operands          [register staging]
//   mode 3  A by LDS-DMA, B through registers + ds_write_b128              [mixed staging]
//   mode 4  LDS-DMA, contiguous 1 KiB pieces (no row gather)               [is the 8-row pattern the cost?]
//   mode 5  LDS-DMA, two K-tiles in flight (4 buffers, no drain per tile)  [latency or throughput?]
//   mode 6  mode 1 with 16 loads in flight per lane
// usage: loadpath [K] [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned short bf16_t;

__device__ __forceinline__ void glds16(const void* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz16(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }

template <int MODE>
__global__ __launch_bounds__(512) void loadpath_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int lda, int ldb,
                                                       int ktiles_total, int ktiles_wrap, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order as in the GEMM: grid (4, 64)
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
    const int bid = blockIdx.x + gx * blockIdx.y;
    const int t = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int bx = t % gx, by = t / gx;
    const int m0 = by * 128, n0 = bx * 128;
    const bf16_t* a_src[2];
    const bf16_t* b_src[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * (wave + 8 * r) + (lane >> 3);
        const int ks = 8 * ((lane & 7) ^ swz16(row));
        if (MODE == 4) {   // contiguous KiB: lane-linear bytes of a 1 KiB block
            a_src[r] = A + (size_t)(m0 + 8 * (wave + 8 * r)) * lda + lane * 8;
            b_src[r] = B + (size_t)(n0 + 8 * (wave + 8 * r)) * ldb + lane * 8;
        } else {
            a_src[r] = A + (size_t)(m0 + row) * lda + ks;
            b_src[r] = B + (size_t)(n0 + row) * ldb + ks;
        }
    }
    unsigned acc = 0;
    constexpr int A_BYTES = 128 * 128;
    unsigned char* As = smem;
    unsigned char* Bs = smem + 4 * A_BYTES;
    if (MODE == 0 || MODE == 4) {
        for (int kt = 0; kt < ktiles_total; ++kt) {
            const int k0 = (kt % ktiles_wrap) * 64;
            const int buf = kt & 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                glds16(a_src[r] + k0, As + buf * A_BYTES + (wave + 8 * r) * 1024);
                glds16(b_src[r] + k0, Bs + buf * A_BYTES + (wave + 8 * r) * 1024);
            }
            __syncthreads();   // drains the DMA queue (vmcnt(0)) like the GEMM's K loop
        }
        acc = *reinterpret_cast<unsigned*>(As + tid * 4);
    } else if (MODE == 5) {
        // two K-tiles in flight: wait for the older one only (4 DMA instructions per tile and wave)
        for (int kt = 0; kt < ktiles_total; ++kt) {
            const int k0 = (kt % ktiles_wrap) * 64;
            const int buf = kt & 3;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                glds16(a_src[r] + k0, As + buf * A_BYTES + (wave + 8 * r) * 1024);
                glds16(b_src[r] + k0, Bs + buf * A_BYTES + (wave + 8 * r) * 1024);
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = *reinterpret_cast<unsigned*>(As + tid * 4);
    } else if (MODE == 1 || MODE == 6) {
        constexpr int DEPTH = MODE == 6 ? 4 : 1;
        for (int kt = 0; kt < ktiles_total; kt += DEPTH) {
            uint4 v[DEPTH][4];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int k0 = ((kt + d) % ktiles_wrap) * 64;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    v[d][2 * r] = *reinterpret_cast<const uint4*>(a_src[r] + k0);
                    v[d][2 * r + 1] = *reinterpret_cast<const uint4*>(b_src[r] + k0);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc ^= v[d][q].x ^ v[d][q].y ^ v[d][q].z ^ v[d][q].w;
            if (MODE == 1) __syncthreads();
        }
    } else if (MODE == 2) {
        uint4 v[4];
        for (int kt = 0; kt < ktiles_total; ++kt) {
            const int k0 = (kt % ktiles_wrap) * 64;
            const int buf = kt & 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                v[2 * r] = *reinterpret_cast<const uint4*>(a_src[r] + k0);
                v[2 * r + 1] = *reinterpret_cast<const uint4*>(b_src[r] + k0);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                *reinterpret_cast<uint4*>(As + buf * A_BYTES + (wave + 8 * r) * 1024 + lane * 16) = v[2 * r];
                *reinterpret_cast<uint4*>(Bs + buf * A_BYTES + (wave + 8 * r) * 1024 + lane * 16) = v[2 * r + 1];
            }
            __syncthreads();
        }
        acc = *reinterpret_cast<unsigned*>(As + tid * 4);
    } else if (MODE == 3) {
        uint4 v[2];
        for (int kt = 0; kt < ktiles_total; ++kt) {
            const int k0 = (kt % ktiles_wrap) * 64;
            const int buf = kt & 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) glds16(a_src[r] + k0, As + buf * A_BYTES + (wave + 8 * r) * 1024);
#pragma unroll
            for (int r = 0; r < 2; ++r) v[r] = *reinterpret_cast<const uint4*>(b_src[r] + k0);
#pragma unroll
            for (int r = 0; r < 2; ++r) *reinterpret_cast<uint4*>(Bs + buf * A_BYTES + (wave + 8 * r) * 1024 + lane * 16) = v[r];
            __syncthreads();
        }
        acc = *reinterpret_cast<unsigned*>(As + tid * 4);
    }
    if (acc == 0x12345677u) sink[0] = acc;   // keeps the loads alive
}

template <int MODE>
double run(const bf16_t* A, const bf16_t* B, int K, int ktiles, unsigned* sink, int reps) {
    auto kern = loadpath_kernel<MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 128));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wrap = K / 64;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(4, 64), dim3(512), 8 * 128 * 128, 0, A, B, K, K, ktiles, wrap, sink);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(4, 64), dim3(512), 8 * 128 * 128, 0, A, B, K, K, ktiles, wrap, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms * 1e3 / reps;   // us per launch
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int ktiles = 256;
    bf16_t *A, *B;
    unsigned* sink;
    CK(hipMalloc(&A, (size_t)8192 * K * 2));
    CK(hipMalloc(&B, (size_t)512 * K * 2));
    CK(hipMalloc(&sink, 64));
    std::vector<bf16_t> h((size_t)8192 * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 22));
    CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)512 * K * 2, hipMemcpyHostToDevice));
    const char* names[] = {"lds-dma both", "global_load -> vgpr only", "global_load + ds_write both", "A lds-dma + B registers",
                           "lds-dma contiguous KiB", "lds-dma 2 tiles in flight", "global_load x16 in flight, no barrier"};
    double us[7];
    us[0] = run<0>(A, B, K, ktiles, sink, reps);
    us[1] = run<1>(A, B, K, ktiles, sink, reps);
    us[2] = run<2>(A, B, K, ktiles, sink, reps);
    us[3] = run<3>(A, B, K, ktiles, sink, reps);
    us[4] = run<4>(A, B, K, ktiles, sink, reps);
    us[5] = run<5>(A, B, K, ktiles, sink, reps);
    us[6] = run<6>(A, B, K, ktiles, sink, reps);
    printf("{\"K\": %d, \"ktiles\": %d, \"bytes_per_cu_per_ktile\": 32768, \"modes\": [", K, ktiles);
    for (int m = 0; m < 7; ++m) {
        const double per_tile_us = us[m] / ktiles;
        const double gbps_cu = 32768.0 / per_tile_us * 1e-3;   // GB/s per CU
        printf("%s{\"mode\": %d, \"name\": \"%s\", \"us_per_launch\": %.2f, \"us_per_ktile\": %.4f, \"GBps_per_cu\": %.1f, \"B_per_clk_at_2p4GHz\": %.1f, \"chip_TBps\": %.2f}",
               m ? ", " : "", m, names[m], us[m], per_tile_us, gbps_cu, gbps_cu / 2.4, gbps_cu * 256e-3);
    }
    printf("]}\n");
    return 0;
}
