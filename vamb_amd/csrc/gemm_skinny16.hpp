// gemm_skinny16.hpp -- latent-wide contractions of the bf16 step with their elementwise consumer fused, gfx950.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      A: [M][K] bf16, B: [N][K] bf16 (both K-contiguous), N = nlatent padded (32 or 64)
//
// Two products of a training step have an output that is only nlatent wide: mu = h W_mu^T (encode.py:268) and the input
// gradient of the first decoder layer, dz_lat = dZ W_dec1.  Until round 4 each was a split-K launch of gemm_bf16_kernel
// (128 x 32 tiles, 4-8 slabs of fp32 partial sums in HBM) followed by a one-pass kernel that added the slabs and did the
// elementwise work (reparameterisation, encode.py:276-286; latent backward): two dependent launches of 5-6 us + 4-5 us for
// 0.27 GFLOP.  Here ONE workgroup owns 32 rows of the output and ALL of K:
//   * wave w contracts the k range of slab w (the same ranges, K-tiles and MFMA order as the split-K launch, so every
//     partial sum has the same bits); its operands -- A [32][k range], B [N][k range] -- arrive by LDS-DMA in ONE burst
//     (no K loop, no double buffering: 64-96 KB of LDS per workgroup, one workgroup per CU, 256 workgroups at batch 8192);
//     the wave waits for its OWN DMAs only (vmcnt), so there is no workgroup barrier in front of the MFMAs;
//   * the partial accumulators meet in LDS (each wave overwrites the head of its own operand region), one barrier, then every
//     thread adds the slabs of its elements in ascending order -- the order of the slab-summing kernels -- and applies the
//     consumer: mu + bias, z = mu + eps (bf16) / dmu = dKLD + dz_lat (bf16).
// LDS image, swizzle and fragment reads are those of gemm_bf16.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gemm_bf16.hpp"

namespace vh {

enum Skinny16Epi : int { SK16_REPARAM = 0, SK16_LATENT_BWD = 1 };

struct Skinny16Args {
    const bf16_t* A;
    int64_t lda;
    const bf16_t* B;
    int64_t ldb;
    int M, K;             // M multiple of 32
    int k_per_wave;       // contraction elements per wave (slab), multiple of 64
    int nslab;            // waves that contract (the workgroup may hold more: they only help in the epilogue)
    const bf16_t* zeros;  // >= 16 bytes of zeros
    int bs;               // rows that belong to the batch
    // SK16_REPARAM: MU = bias + sum of slabs; Z16 = bf16(MU + eps) on real rows / columns, 0 elsewhere
    const float* bias;
    const float* E;       // injected noise [M][N] or nullptr
    uint64_t key;
    const unsigned long long* step_ptr;
    int noise;
    int L;                // real latent columns
    float* MU;            // [M][N]
    bf16_t* Z16;          // [M][N]
    // SK16_LATENT_BWD: dMU16 = bf16(dMUk + sum of slabs) on real rows, 0 on the padding rows
    const float* dMUk;    // [M][N]
    bf16_t* dMU16;        // [M][N]
};

constexpr int kSkinnyMinWaves = 4;    // the epilogue wants >= 256 threads (elements per thread bounded at compile time)

template <int NB>
constexpr size_t skinny16_tile_bytes() { return (size_t)(32 + 32 * NB) * 128; }

template <int NB, int EPI>
__global__ __launch_bounds__(512) void gemm_skinny16_kernel(const Skinny16Args g) {
    constexpr int N = 32 * NB;
    constexpr int TILE_BYTES = (32 + N) * 128;   // one K-tile of a wave: A image [32][64 bf16], then B image [N][64 bf16]
    constexpr int PB = N / 8;                    // 1 KiB DMA pieces of the B image (the A image has 4)
    constexpr int EPT = 4 * NB;                  // output elements per thread with >= 256 threads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sk[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ktw = g.k_per_wave >> 6;
    const size_t region_bytes = (size_t)ktw * TILE_BYTES;
    unsigned char* const region = smem_sk + (size_t)wave * region_bytes;
    const int m0 = blockIdx.x * 32;
    const bool contracts = wave < g.nslab;       // wave-uniform
    const int kbeg = wave * g.k_per_wave;
    const int kend = min(g.K, kbeg + g.k_per_wave);
    const int nk = contracts ? (kend - kbeg + 63) >> 6 : 0;

    // ---- the whole operand set of this wave, one burst.  Lane (row = 8 piece + lane / 8, LDS slot lane % 8) fetches
    // k-slot (lane % 8) ^ f(row) of its row (gemm_bf16.hpp); slots past the end of K come from the zero block
    for (int t = 0; t < nk; ++t) {
        const int k0 = kbeg + 64 * t;
        const int room = kend - k0;
        unsigned char* const ab = region + (size_t)t * TILE_BYTES;
        unsigned char* const bb = ab + 32 * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = 8 * p + (lane >> 3);
            const int ks = 8 * ((lane & 7) ^ swz16(row));
            glds16(ks < room ? g.A + (int64_t)(m0 + row) * g.lda + k0 + ks : g.zeros, ab + p * 1024);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int row = 8 * p + (lane >> 3);
            const int ks = 8 * ((lane & 7) ^ swz16(row));
            glds16(ks < room ? g.B + (int64_t)row * g.ldb + k0 + ks : g.zeros, bb + p * 1024);
        }
    }

    // ---- what the epilogue needs from HBM travels under the DMA burst: thread t owns the output elements t, t + nthreads, ...
    // of the row-major [32][N] block
    float pre[EPT];
    uint64_t nkey = 0;
    if constexpr (EPI == SK16_REPARAM) {
        if (g.E == nullptr && g.noise) nkey = step_key(g.key, g.step_ptr);
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * nthreads;
        pre[j] = 0.f;
        if (e < 32 * N) {
            const int r = m0 + e / N, c = e % N;
            const int64_t i = (int64_t)m0 * N + e;
            if constexpr (EPI == SK16_REPARAM) {
                if (r < g.bs && c < g.L) {
                    if (g.E) pre[j] = g.E[i];
                    else if (g.noise) pre[j] = hash_randn(nkey, (uint64_t)i);
                }
            } else {
                if (r < g.bs) pre[j] = g.dMUk[i];
            }
        }
    }

    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs have landed (nobody else reads its region yet)
    __builtin_amdgcn_sched_barrier(0);

    // ---- fragments: lane (r = lane & 31, h = lane >> 5) reads k = 16 s + 8 h .. + 7 of row r
    const int fr = lane & 31, fh = lane >> 5;
    const int a_swz = swz16(fr);
    int b_swz[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) b_swz[j] = swz16(32 * j + fr);
    for (int t = 0; t < nk; ++t) {
        const unsigned char* const ab = region + (size_t)t * TILE_BYTES;
        const unsigned char* const bb = ab + 32 * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(ab + fr * 128 + 16 * ((2 * s + fh) ^ a_swz));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(bb + (32 * j + fr) * 128 + 16 * ((2 * s + fh) ^ b_swz[j]));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[j], 0, 0, 0);
            }
        }
    }

    // ---- the slabs meet in LDS: wave w's [32][N] fp32 block replaces the head of its own operand region (4 / 8 KB of >= 8 /
    // 12 KB; its own fragment reads are complete -- the MFMAs consumed them)
    if (contracts) {
        float* const part = reinterpret_cast<float*>(region);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * fh;
                part[row * N + 32 * j + fr] = acc[j][reg];
            }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * nthreads;
        if (e >= 32 * N) continue;
        const int r = m0 + e / N, c = e % N;
        const int64_t i = (int64_t)m0 * N + e;
        if constexpr (EPI == SK16_REPARAM) {
            float m = g.bias[c];
            for (int w = 0; w < g.nslab; ++w) m += reinterpret_cast<const float*>(smem_sk + (size_t)w * region_bytes)[e];
            g.MU[i] = m;
            float z = 0.f;
            if (r < g.bs && c < g.L) z = m + pre[j];
            g.Z16[i] = f2bf(z);
        } else {
            float t = 0.f;
            if (r < g.bs) {
                t = pre[j];
                for (int w = 0; w < g.nslab; ++w) t += reinterpret_cast<const float*>(smem_sk + (size_t)w * region_bytes)[e];
            }
            g.dMU16[i] = f2bf(t);
        }
    }
}

// LDS a launch needs; 0 when the shape does not fit one workgroup (the caller keeps the split-K launch + slab kernel)
inline size_t skinny16_smem_bytes(int N, int k_per_wave, int nslab) {
    if (N != 32 && N != 64) return 0;
    if (nslab < 1 || nslab > 8 || (k_per_wave & 63) != 0) return 0;
    const size_t tile = (size_t)(32 + N) * 128;
    const size_t need = (size_t)nslab * (size_t)(k_per_wave >> 6) * tile;
    return need <= (size_t)150 * 1024 ? need : 0;
}

}  // namespace vh
