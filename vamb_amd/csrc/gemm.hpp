// gemm.hpp -- fp32-input MFMA GEMM for the VAE's dense contractions (gfx950).
//
//   C[m][n] = sum_k A'(m,k) * B'(n,k)        m < M, n < N, k < K (K multiple of 32)
//
// built on v_mfma_f32_32x32x2_f32: exact fp32 products accumulated as a k-ordered fmaf chain (the
// guide's "SGEMM class": 64 FLOP/clk/SIMD, 157 TF/s chip peak), which is what BASELINE config C1
// ("fp32") asks for.  Each operand can be stored either "K-contiguous" (row-major [rows][K]: the
// forward activations and weights) or "row-contiguous" ([K][rows]: what the backward passes need,
// dX = dY*W reads W as [contraction n][k], dW = dY^T*X reads both operands as [contraction m][..]),
// so no transposed copies of weights or activations are ever materialised.
//
// Operand transform (the primes above), applied in registers between the global load and the LDS store:
//   XF_BN : a = h * scale[col] + shift[col]      BatchNorm1d (training statistics) of the producing layer
// so the normalised activations are never written to HBM.  The per-column coefficients are derived once
// per workgroup from the batch sums that the producing GEMM's epilogue accumulated with fp64 atomics, so
// BatchNorm needs no finalize / apply kernels.  (Forming dZ on load as well was measured 2-3x slower per
// GEMM -- every A tile is transformed once per column tile -- so the elementwise backward keeps its own
// bandwidth-bound kernel, vae_dz_kernel.)
//
// Tile: BM x BN x BK per workgroup of WM x WN wavefronts, each wave TM x TN MFMA tiles of 32x32.
// Global -> registers (float4, coalesced along the contiguous dimension) -> LDS [k][row] (stride
// BM+1 for transposing ds_write_b32 of K-contiguous operands, BM+4 for ds_write_b128 of
// row-contiguous ones; both conflict-free) -> ds_read_b32 fragments (lane l: row l&31, k l>>5).
// Double-buffered LDS, next tile's global loads in flight during the current tile's MFMAs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace vh {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kBnEpsF = 1e-5f;     // torch.nn.BatchNorm1d eps (encode.py:238,246)
constexpr float kLeakySlopeF = 0.01f;  // torch.nn.LeakyReLU default (encode.py:252)

enum EpiKind : int {
    EPI_STORE = 0,         // C = acc
    EPI_BIAS = 1,          // C = acc + bias[n]
    EPI_HIDDEN_TRAIN = 2,  // h = dropout(leaky_relu(acc + bias)); C = h; batch sums of h, h^2 (encode.py:264,292)
    EPI_HIDDEN_EVAL = 3,   // C = leaky_relu(acc + bias) * scale[n] + shift[n]   (eval-mode BatchNorm folded)
    EPI_SPLITK = 4,        // C[blockIdx.z] = acc   (partial slabs, summed by the consumer)
    EPI_LATENT_MASK = 5,   // C = bits(acc + bias) & ~0xFFF   (encode.py:483, vambtools.py:324-330)
    EPI_STORE_BNRED = 6    // C = acc (= dA of the layer below) + batch sums of dA and dA*xhat for its BatchNorm backward
};

enum XfKind : int { XF_NONE = 0, XF_BN = 1 };

// Batch statistics of one BatchNorm1d layer as accumulated by EPI_HIDDEN_TRAIN: fstat[0][c] = sum h,
// fstat[1][c] = sum h^2 over the bs real rows; plus its affine parameters.
struct BnSrc {
    const double* fstat;   // [2][n_p]
    const float* gamma;
    const float* beta;
    int n_p;
    int bs;
};

struct GemmArgs {
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    float* C;
    int64_t ldc;
    int M, N, K;
    int k_per_split;      // contraction elements per blockIdx.z (multiple of 32)
    int64_t slab_stride;  // elements between split-K slabs
    const float* bias;
    const float* scale;
    const float* shift;
    int m_real;           // rows that belong to the batch (statistics / dropout rows)
    double* fstat_out;    // EPI_HIDDEN_TRAIN: [2][N] batch sums (fp64 atomics)
    float drop_scale;     // 1/(1-p); p == 0 disables dropout
    uint32_t drop_thresh; // keep iff hash32 >= drop_thresh  (p * 2^32)
    uint64_t drop_key;    // seed ^ layer (^ step, read from *step_ptr so that a captured graph stays valid)
    const unsigned long long* step_ptr;  // device-resident global step counter (may be nullptr)
    const uint8_t* drop_mask;  // injected keep-mask [m_real][ld_mask] (parity mode) or nullptr
    int64_t ld_mask;
    // operand transforms
    BnSrc bnA, bnB;       // XF_BN on A / B: statistics of the layer that produced the operand
    // EPI_STORE_BNRED
    const float* Hbelow;  // activations of the layer below (same shape / ld as C)
    BnSrc bnC;            // its forward statistics
    double* bstat_out;    // [2][N]
    int wide_k;           // 1: 64-wide K-tiles (host-side dispatch only; forward GEMMs, which run alone on the GPU)
    int bf16;             // 1: launch the bf16-operand instantiation (host-side dispatch only)
    int xcd_remap;        // 1: give every XCD a contiguous chunk of the tile grid (L2 reuse of operand panels)
    int group_floats;     // K-group instantiations (KS > 1): LDS floats per group (set by the launcher)
};

// counter-based uniform 32-bit hash (splitmix64 finaliser); also used by the backward transforms so the
// dropout mask is regenerated instead of stored
__host__ __device__ __forceinline__ uint32_t hash32(uint64_t key, uint64_t idx) {
    uint64_t z = key + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

// dropout keep bits: one 32-bit mix (3 integer multiplies instead of the 4 64-bit ones of hash32 + index)
// yields two 16-bit uniforms, i.e. the decisions of two vertically adjacent elements.  The backward pass never
// re-evaluates it (a dropped unit is stored as exactly 0), and parity tests inject their masks.
__device__ __forceinline__ uint32_t hash_drop(uint64_t key, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u + (uint32_t)key;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x ^= (uint32_t)(key >> 32);
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

// standard-normal noise: Box-Muller over the counter-based hash (free-running mode)
__device__ __forceinline__ float hash_randn(uint64_t key, uint64_t id) {
    const float u1 = ((float)hash32(key, 2 * id) + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)hash32(key, 2 * id + 1) * 2.3283064365386963e-10f;
    // hardware log2 / cos (v_log_f32, v_cos_f32): the libm-accurate versions made the noise the most expensive part of
    // the kernels that draw it; 1-2 ulp of a standard-normal deviate is far below its own sampling noise
    return __fsqrt_rn(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

__device__ __forceinline__ uint64_t step_key(uint64_t base, const unsigned long long* step_ptr) {
    return step_ptr ? (base ^ ((uint64_t)(*step_ptr) << 8)) : base;
}

// mean / 1/std / scale / shift of one BatchNorm column from the batch sums (biased variance)
__device__ __forceinline__ void bn_column(const BnSrc& s, int col, float& mean, float& istd, float& scale,
                                          float& shift) {
    const double inv_bs = 1.0 / (double)s.bs;
    const double m = s.fstat[col] * inv_bs;
    double var = s.fstat[s.n_p + col] * inv_bs - m * m;   // fp64: E[h^2] - mean^2 cancels
    if (var < 0.0) var = 0.0;
    istd = 1.0f / sqrtf((float)var + kBnEpsF);
    mean = (float)m;
    scale = s.gamma[col] * istd;
    shift = s.beta[col] - mean * scale;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// a 16-byte global load the compiler does not track (its completion is awaited by a hand-written s_waitcnt), and the empty asm
// a register passes through so that no use of it is scheduled above that wait
using f32x4 = __attribute__((ext_vector_type(4))) float;   // (a register tuple for the asm constraints; HIP's float4 is a struct)
__device__ __forceinline__ void asm_load_x4(f32x4& dst, const float* src) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
__device__ __forceinline__ void asm_pin_x4(f32x4& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ float4 keep4(float4 v, bool keep) {   // (component selects: a ?: between two structs goes through memory)
    v.x = keep ? v.x : 0.f; v.y = keep ? v.y : 0.f; v.z = keep ? v.z : 0.f; v.w = keep ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ float4 as_float4(const float4& v) { return v; }
__device__ __forceinline__ float4 as_float4(const f32x4& v) { return make_float4(v.x, v.y, v.z, v.w); }

// PF (fp32 operands only): K-tiles kept in flight in registers.  1 = the loop as it always was (tile t + 1 requested while tile t
// is contracted): right when several workgroups share a CU and hide each other's latency.  4 = the small-batch variant (the joint
// TaxVamb step at batch 256: 32 workgroups on 256 CUs, each alone on its CU with a 16-deep K loop whose every iteration waited a
// full L2 / HBM round trip, ~1.3 us, for 0.43 us of MFMA work): tile t + 4 is requested while tile t is contracted, every load
// unconditional (clamped to the matrix; out-of-range rows are zeroed on the way into LDS) so that the compiler's counted vmcnt
// waits survive.  The host picks it when the K loop is a multiple of PF tiles deep (launch_gemm).
// KS (fp32 operands, one wavefront per group: WM = WN = 1): KS wavefront GROUPS per workgroup, group i contracting the i-th
// slab of the workgroup's K range into its own accumulators over its own LDS tiles; the slabs meet in LDS in ascending order
// (deterministic) and group 0 runs the epilogue.  For launches that would otherwise be a few dozen workgroups: 256 x 512 x 512
// is 32 workgroups of 64 x 64 -- 128 of the chip's 1024 matrix pipes, 6.8 us of fp32 MFMA time each -- or 128 workgroups of
// 32 x 32 x (4 x 128) with four times the pipes busy and a K loop a quarter as deep.  The sum of the slabs is formed in another
// order than the plain K loop's: same value up to float32 rounding, as with every split-K launch of this file.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int EPI, int XFA = XF_NONE, int XFB = XF_NONE,
          int BK = 32, int DT = 0, int PF = 1, int KS = 1>
__global__ __launch_bounds__(WM * WN * KS * 64) void gemm_f32_kernel(const GemmArgs g) {
    // DT = 0: fp32 operands on v_mfma_f32_32x32x2_f32.  DT = 1 (BASELINE configs C2+): the operands are
    // rounded to bf16 (RNE) while they are staged into LDS and contracted on v_mfma_f32_32x32x16_bf16 with
    // fp32 accumulation; global tensors, transforms and epilogues stay fp32.
    constexpr bool BF = DT == 1;
    constexpr int NT = WM * WN * 64;       // threads per workgroup (1 wavefront per 32x32-tile group)
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int KQ = BK / 4;             // float4 units along K of an operand row
    // float4 registers per thread per K-tile (bf16, row-contiguous operand: pairs of K rows)
    constexpr int UA = (BF && !A_KC) ? 2 * ((4 * BM + NT - 1) / NT) : (BM * KQ + NT - 1) / NT;
    constexpr int UB = (BF && !B_KC) ? 2 * ((4 * BN + NT - 1) / NT) : (BN * KQ + NT - 1) / NT;
    static_assert(BK == 32 || (BK == 64 && DT == 0), "the swizzled LDS image is 8 or 16 quads wide (bf16: 32 elements)");
    static_assert(NT % KQ == 0, "every unit of a thread shares its k-quad");
    static_assert(WM * WN == 4 || WM * WN == 1, "4 wavefronts per workgroup, or a single free-running one");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    static_assert(PF == 1 || DT == 0, "the deep prefetch is an fp32-operand variant");
    static_assert(KS == 1 || (WM * WN == 1 && DT == 0), "K groups are single wavefronts, fp32 operands");

    // LDS image of an operand tile: [rows][32 floats], i.e. K-contiguous rows of 8 quads (16 B each), the
    // quad of logical index kq stored in slot kq ^ swz(row).  Every fragment read is one ds_read_b128 (256
    // B/clk/CU, full rate from one wave per SIMD; ds_read_b32 needs ~4 waves per SIMD for half of that) and
    // conflict-free: the 16 lanes a b128 access services together hold 16 distinct (row & 1, swz(row)).
    // A row-contiguous operand (element (k, row) at k*ld + row) keeps the image [32 k][rows + 4]: its float4
    // loads are stored as they are (one ds_write_b128) and its fragments are single ds_read_b32 -- transposing
    // it into the K-contiguous image costs more in conflicting ds_write_b64 than the wide reads give back
    // (measured: dW GEMM 57 -> 52 TF/s).
    constexpr int SA = BM + 4, SB = BN + 4;                 // row-contiguous image: floats per k row
    // bf16: every operand uses the K-contiguous image [rows][32 bf16] (64 B per row, 4 quads of 8 elements, quad q
    // in slot q ^ ((row >> 2) & 3)): one ds_read_b128 is the 8-element operand of one 32x32x16 MFMA.
    constexpr int TILE_A = BF ? BM * BK / 2 : (A_KC ? BM * BK : BK * SA);        // floats per buffer
    constexpr int TILE_B = BF ? BN * BK / 2 : (B_KC ? BN * BK : BK * SB);
    extern __shared__ __attribute__((aligned(16))) float gemm_smem_all[];
    // K groups: every group owns one copy of the layout below (g.group_floats apart) and the slab [kbeg, kend) of the K range
    const int grp = KS > 1 ? (int)threadIdx.x / NT : 0;
    float* gemm_smem = gemm_smem_all + (KS > 1 ? (size_t)grp * g.group_floats : 0);
    float* As = gemm_smem;                 // [2][TILE_A]
    float* Bs = gemm_smem + 2 * TILE_A;    // [2][TILE_B]
    float* coef = Bs + 2 * TILE_B;         // per-column coefficients of the operand transforms (see below)

    const int tid = KS > 1 ? (int)threadIdx.x % NT : (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroup b is observed to run on XCD b % 8, each XCD with a private 4 MiB L2.  With the plain mapping
    // every XCD walks one column of tiles over ALL row panels of A; the bijective remap below hands XCD x the
    // x-th contiguous eighth of the (z, m, n)-ordered tile list, so the A / B panels of neighbouring tiles are
    // fetched once per XCD instead of once per workgroup.  Speed only, never correctness.  (Measured neutral
    // at the C1 shapes, where both operands fit the L2s / Infinity Cache either way -- DESIGN.md section 5.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd_remap) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int nwg = gx * gy * (int)gridDim.z;
        const int bid = bx + gx * (by + gy * bz);
        const int xcd = bid & 7, local = bid >> 3;
        const int t = xcd * (nwg >> 3) + min(xcd, nwg & 7) + local;
        bx = t % gx;
        by = (t / gx) % gy;
        bz = t / (gx * gy);
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int kslab = g.k_per_split / KS;                          // (KS > 1: the host guarantees K = splits * KS * kslab)
    const int kbeg = bz * g.k_per_split + grp * kslab;
    const int kend = min(g.K, kbeg + kslab);
    const int nchunks = (kend - kbeg) / BK;

    // ---- coefficient tables in LDS ---------------------------------------------------------------
    // The columns a transform indexes are the contraction range [kbeg, kend) for a K-contiguous operand
    // and the workgroup's own BM (BN) rows for a row-contiguous one.
    const int ncolA = A_KC ? (kend - kbeg) : BM;
    const int colA0 = A_KC ? kbeg : m0;
    const int ncolB = B_KC ? (kend - kbeg) : BN;
    const int colB0 = B_KC ? kbeg : n0;
    // layout: [A: XF_BN 2 arrays][B: XF_BN 2 arrays][C: BNRED 2 arrays of BN]
    float* coefA = coef;
    constexpr int NARR_A = XFA == XF_BN ? 2 : 0;
    float* coefB = coefA + NARR_A * ncolA;
    constexpr int NARR_B = XFB == XF_BN ? 2 : 0;
    float* coefC = coefB + NARR_B * ncolB;
    if constexpr (XFA == XF_BN) {
        const int limit = A_KC ? g.K : g.M;
        for (int c = tid; c < ncolA; c += NT) {
            float mean, istd, sc = 0.f, sh = 0.f;
            if (colA0 + c < limit) bn_column(g.bnA, colA0 + c, mean, istd, sc, sh);
            coefA[c] = sc;
            coefA[ncolA + c] = sh;
        }
    }
    if constexpr (XFB == XF_BN) {
        const int limit = B_KC ? g.K : g.N;
        for (int c = tid; c < ncolB; c += NT) {
            float mean, istd, sc = 0.f, sh = 0.f;
            if (colB0 + c < limit) bn_column(g.bnB, colB0 + c, mean, istd, sc, sh);
            coefB[c] = sc;
            coefB[ncolB + c] = sh;
        }
    }
    if constexpr (EPI == EPI_STORE_BNRED) {
        for (int c = tid; c < BN; c += NT) {
            float mean = 0.f, istd = 0.f, sc, sh;
            if (n0 + c < g.N) bn_column(g.bnC, n0 + c, mean, istd, sc, sh);
            coefC[c] = mean;
            coefC[BN + c] = istd;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[UA], rb[UB];
    f32x4 rra[PF][UA], rrb[PF][UB];   // PF > 1 only
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // PF > 1: stage st <- K-tile at k0, unconditionally (rows clamped to the matrix).  The loads and the waits for them are
    // written out: left to the compiler, a register that rotates through the stages of the unrolled loop is "maybe pending" at
    // the loop's back edge and every wait becomes vmcnt(0) -- the round trip per iteration the variant exists to remove.
    // stage_wait(st) returns when stage st's loads have landed, given that exactly PF - 1 younger stages are in flight.
    auto gload_pf = [&](int st, int k0) {   // (st is a constant after unrolling)
#pragma unroll
        for (int r = 0; r < UA; ++r) {
            const int u = min(tid + NT * r, BM * KQ - 1);
            const float* src;
            if constexpr (A_KC) {
                const int row = u / KQ, kq = u % KQ, gm = min(m0 + row, g.M - 1);
                src = g.A + (int64_t)gm * g.lda + k0 + 4 * kq;
            } else {
                const int k = u / (BM / 4), mq = u % (BM / 4), gm = min(m0 + 4 * mq, g.M - 4);
                src = g.A + (int64_t)(k0 + k) * g.lda + gm;
            }
            asm_load_x4(rra[st][r], src);
        }
#pragma unroll
        for (int r = 0; r < UB; ++r) {
            const int u = min(tid + NT * r, BN * KQ - 1);
            const float* src;
            if constexpr (B_KC) {
                const int row = u / KQ, kq = u % KQ, gn = min(n0 + row, g.N - 1);
                src = g.B + (int64_t)gn * g.ldb + k0 + 4 * kq;
            } else {
                const int k = u / (BN / 4), nq = u % (BN / 4), gn = min(n0 + 4 * nq, g.N - 4);
                src = g.B + (int64_t)(k0 + k) * g.ldb + gn;
            }
            asm_load_x4(rrb[st][r], src);
        }
    };
    auto stage_wait = [&](int st) {
        constexpr int behind = (PF - 1) * (UA + UB);
        static_assert(behind < 64, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(behind) : "memory");
        // (the stage's registers pass through an empty asm so that nothing reading them is scheduled above the wait)
#pragma unroll
        for (int r = 0; r < UA; ++r) asm_pin_x4(rra[st][r]);
#pragma unroll
        for (int r = 0; r < UB; ++r) asm_pin_x4(rrb[st][r]);
    };

    // Global -> registers for the K-tile starting at k0.
    //   K-contiguous operand: unit u = (row u/8, quad u%8): one float4 = 4 consecutive k of one row.
    //   row-contiguous operand (element (k, row) at k*ld + row): unit u = (k u / (rows/4), row-quad
    //   u % (rows/4)): one float4 = rows 4q..4q+3 of one k (coalesced along the rows).
    auto gload = [&](int k0) {
        if constexpr (A_KC) {
#pragma unroll
            for (int r = 0; r < UA; ++r) {
                const int u = tid + NT * r;
                const int row = u / KQ, kq = u % KQ, gm = m0 + row;
                ra[r] = zero4;
                if (u < BM * KQ && gm < g.M) ra[r] = *reinterpret_cast<const float4*>(g.A + (int64_t)gm * g.lda + k0 + 4 * kq);
            }
        } else if constexpr (BF) {
            // pair unit u = (k-pair u / (rows/4), row-quad u % (rows/4)): rows 4q..4q+3 at k = 2p and 2p + 1
#pragma unroll
            for (int r = 0; r < UA / 2; ++r) {
                const int u = tid + NT * r;
                const int kp = u / (BM / 4), mq = u % (BM / 4), gm = m0 + 4 * mq;
                ra[2 * r] = zero4;
                ra[2 * r + 1] = zero4;
                if (u < 4 * BM && gm < g.M) {
                    const float* src = g.A + (int64_t)(k0 + 2 * kp) * g.lda + gm;
                    ra[2 * r] = *reinterpret_cast<const float4*>(src);
                    ra[2 * r + 1] = *reinterpret_cast<const float4*>(src + g.lda);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < UA; ++r) {
                const int u = tid + NT * r;
                const int k = u / (BM / 4), mq = u % (BM / 4), gm = m0 + 4 * mq;
                ra[r] = zero4;
                if (u < BM * KQ && gm < g.M) ra[r] = *reinterpret_cast<const float4*>(g.A + (int64_t)(k0 + k) * g.lda + gm);
            }
        }
        if constexpr (B_KC) {
#pragma unroll
            for (int r = 0; r < UB; ++r) {
                const int u = tid + NT * r;
                const int row = u / KQ, kq = u % KQ, gn = n0 + row;
                rb[r] = zero4;
                if (u < BN * KQ && gn < g.N) rb[r] = *reinterpret_cast<const float4*>(g.B + (int64_t)gn * g.ldb + k0 + 4 * kq);
            }
        } else if constexpr (BF) {
#pragma unroll
            for (int r = 0; r < UB / 2; ++r) {
                const int u = tid + NT * r;
                const int kp = u / (BN / 4), nq = u % (BN / 4), gn = n0 + 4 * nq;
                rb[2 * r] = zero4;
                rb[2 * r + 1] = zero4;
                if (u < 4 * BN && gn < g.N) {
                    const float* src = g.B + (int64_t)(k0 + 2 * kp) * g.ldb + gn;
                    rb[2 * r] = *reinterpret_cast<const float4*>(src);
                    rb[2 * r + 1] = *reinterpret_cast<const float4*>(src + g.ldb);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < UB; ++r) {
                const int u = tid + NT * r;
                const int k = u / (BN / 4), nq = u % (BN / 4), gn = n0 + 4 * nq;
                rb[r] = zero4;
                if (u < BN * KQ && gn < g.N) rb[r] = *reinterpret_cast<const float4*>(g.B + (int64_t)(k0 + k) * g.ldb + gn);
            }
        }
    };

    // BatchNorm-on-load (XF_BN): v * scale[col] + shift[col]; the column is the k index of a K-contiguous
    // operand (4 consecutive coefficients per float4) and the row index of a row-contiguous one.
    auto bn4 = [](float4 v, const float* sc, const float* sh) -> float4 {
        const float4 s4 = *reinterpret_cast<const float4*>(sc);
        const float4 t4 = *reinterpret_cast<const float4*>(sh);
        v.x = v.x * s4.x + t4.x; v.y = v.y * s4.y + t4.y; v.z = v.z * s4.z + t4.z; v.w = v.w * s4.w + t4.w;
        return v;
    };
    // 8 quads per row (128 B): rows r, r + 1 share a bank half, so the slot mixes (row >> 1) and (row >> 4);
    // 16 quads per row (256 B = all 64 banks): every row starts at bank 0, slot = quad ^ (row & 15)
    auto swz = [](int row) -> int { return KQ == 8 ? (((row >> 1) ^ (row >> 4)) & 7) : (row & 15); };

    // bf16 image helpers: element (row, k) at row * 32 + 8 * ((k >> 3) ^ swzb(row)) + (k & 7)   [bf16 units]
    auto swzb = [](int row) -> int { return (row >> 2) & 3; };
    auto pack2 = [](float lo, float hi) -> unsigned int {
        const __bf16 a = (__bf16)lo, b = (__bf16)hi;   // round to nearest even (v_cvt_pk_bf16_f32)
        return (unsigned int)__builtin_bit_cast(unsigned short, a) | ((unsigned int)__builtin_bit_cast(unsigned short, b) << 16);
    };
    auto sstore_bf16 = [&](int buf, int k0) {
        unsigned short* as = reinterpret_cast<unsigned short*>(As + buf * TILE_A);
        unsigned short* bs = reinterpret_cast<unsigned short*>(Bs + buf * TILE_B);
        if constexpr (A_KC) {
#pragma unroll
            for (int r = 0; r < UA; ++r) {
                const int u = tid + NT * r;
                const int row = u / KQ, kq = u % KQ;
                float4 v = ra[r];
                if constexpr (XFA == XF_BN) v = bn4(v, coefA + (k0 - kbeg) + 4 * kq, coefA + ncolA + (k0 - kbeg) + 4 * kq);
                if (u < BM * KQ)
                    *reinterpret_cast<uint2*>(as + row * 32 + 8 * ((kq >> 1) ^ swzb(row)) + 4 * (kq & 1)) =
                        make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
            }
        } else {
#pragma unroll
            for (int r = 0; r < UA / 2; ++r) {
                const int u = tid + NT * r;
                const int kp = u / (BM / 4), mq = u % (BM / 4);
                float4 v0 = ra[2 * r], v1 = ra[2 * r + 1];
                if constexpr (XFA == XF_BN) {
                    v0 = bn4(v0, coefA + 4 * mq, coefA + ncolA + 4 * mq);
                    v1 = bn4(v1, coefA + 4 * mq, coefA + ncolA + 4 * mq);
                }
                if (u < 4 * BM) {
                    const float e0[4] = {v0.x, v0.y, v0.z, v0.w}, e1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 4 * mq + i;
                        *reinterpret_cast<unsigned int*>(as + row * 32 + 8 * ((kp >> 2) ^ swzb(row)) + 2 * (kp & 3)) =
                            pack2(e0[i], e1[i]);
                    }
                }
            }
        }
        if constexpr (B_KC) {
#pragma unroll
            for (int r = 0; r < UB; ++r) {
                const int u = tid + NT * r;
                const int row = u / KQ, kq = u % KQ;
                float4 v = rb[r];
                if constexpr (XFB == XF_BN) v = bn4(v, coefB + (k0 - kbeg) + 4 * kq, coefB + ncolB + (k0 - kbeg) + 4 * kq);
                if (u < BN * KQ)
                    *reinterpret_cast<uint2*>(bs + row * 32 + 8 * ((kq >> 1) ^ swzb(row)) + 4 * (kq & 1)) =
                        make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
            }
        } else {
#pragma unroll
            for (int r = 0; r < UB / 2; ++r) {
                const int u = tid + NT * r;
                const int kp = u / (BN / 4), nq = u % (BN / 4);
                float4 v0 = rb[2 * r], v1 = rb[2 * r + 1];
                if constexpr (XFB == XF_BN) {
                    v0 = bn4(v0, coefB + 4 * nq, coefB + ncolB + 4 * nq);
                    v1 = bn4(v1, coefB + 4 * nq, coefB + ncolB + 4 * nq);
                }
                if (u < 4 * BN) {
                    const float e0[4] = {v0.x, v0.y, v0.z, v0.w}, e1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 4 * nq + i;
                        *reinterpret_cast<unsigned int*>(bs + row * 32 + 8 * ((kp >> 2) ^ swzb(row)) + 2 * (kp & 3)) =
                            pack2(e0[i], e1[i]);
                    }
                }
            }
        }
    };

    // registers -> LDS (applied when the prefetched data has arrived)
    auto sstore_from = [&](const auto& xa, const auto& xb, int buf, int k0) {
        constexpr bool kFullA = (BM * KQ) % NT == 0, kFullB = (BN * KQ) % NT == 0;   // every thread's every unit is inside the tile
        float* as = As + buf * TILE_A;
        float* bs = Bs + buf * TILE_B;
        // every unit of a thread has the same k-quad (NT is a multiple of 8): one coefficient pair per K-tile
        float4 sA = zero4, tA = zero4, sB = zero4, tB = zero4;
        if constexpr (A_KC && XFA == XF_BN) {
            sA = *reinterpret_cast<const float4*>(coefA + (k0 - kbeg) + 4 * (tid % KQ));
            tA = *reinterpret_cast<const float4*>(coefA + ncolA + (k0 - kbeg) + 4 * (tid % KQ));
        }
        if constexpr (B_KC && XFB == XF_BN) {
            sB = *reinterpret_cast<const float4*>(coefB + (k0 - kbeg) + 4 * (tid % KQ));
            tB = *reinterpret_cast<const float4*>(coefB + ncolB + (k0 - kbeg) + 4 * (tid % KQ));
        }
        // row-contiguous operands: the coefficient index is the unit's row quad, the same for every unit of a thread
        // when NT is a multiple of rows / 4 (64- and 128-row tiles with 256 threads)
        constexpr bool kRowQuadA = !A_KC && XFA == XF_BN && NT % (BM / 4) == 0;
        constexpr bool kRowQuadB = !B_KC && XFB == XF_BN && NT % (BN / 4) == 0;
        if constexpr (kRowQuadA) {
            sA = *reinterpret_cast<const float4*>(coefA + 4 * (tid % (BM / 4)));
            tA = *reinterpret_cast<const float4*>(coefA + ncolA + 4 * (tid % (BM / 4)));
        }
        if constexpr (kRowQuadB) {
            sB = *reinterpret_cast<const float4*>(coefB + 4 * (tid % (BN / 4)));
            tB = *reinterpret_cast<const float4*>(coefB + ncolB + 4 * (tid % (BN / 4)));
        }
        auto fma4 = [](float4 v, const float4& s4, const float4& t4) -> float4 {
            v.x = v.x * s4.x + t4.x; v.y = v.y * s4.y + t4.y; v.z = v.z * s4.z + t4.z; v.w = v.w * s4.w + t4.w;
            return v;
        };
#pragma unroll
        for (int r = 0; r < UA; ++r) {
            const int u = tid + NT * r;
            float4 v = as_float4(xa[r]);
            if constexpr (A_KC) {
                const int row = u / KQ, kq = u % KQ;
                if constexpr (PF > 1) v = keep4(v, m0 + row < g.M);   // (the deep prefetch loads clamped rows)
                if constexpr (XFA == XF_BN) v = fma4(v, sA, tA);
                if (kFullA || u < BM * KQ) *reinterpret_cast<float4*>(as + row * BK + 4 * (kq ^ swz(row))) = v;
            } else {
                const int k = u / (BM / 4), mq = u % (BM / 4);
                if constexpr (PF > 1) v = keep4(v, m0 + 4 * mq < g.M);
                if constexpr (kRowQuadA) v = fma4(v, sA, tA);
                else if constexpr (XFA == XF_BN) v = bn4(v, coefA + 4 * mq, coefA + ncolA + 4 * mq);
                if (kFullA || u < BM * KQ) *reinterpret_cast<float4*>(as + k * SA + 4 * mq) = v;
            }
        }
#pragma unroll
        for (int r = 0; r < UB; ++r) {
            const int u = tid + NT * r;
            float4 v = as_float4(xb[r]);
            if constexpr (B_KC) {
                const int row = u / KQ, kq = u % KQ;
                if constexpr (PF > 1) v = keep4(v, n0 + row < g.N);
                if constexpr (XFB == XF_BN) v = fma4(v, sB, tB);
                if (kFullB || u < BN * KQ) *reinterpret_cast<float4*>(bs + row * BK + 4 * (kq ^ swz(row))) = v;
            } else {
                const int k = u / (BN / 4), nq = u % (BN / 4);
                if constexpr (PF > 1) v = keep4(v, n0 + 4 * nq < g.N);
                if constexpr (kRowQuadB) v = fma4(v, sB, tB);
                else if constexpr (XFB == XF_BN) v = bn4(v, coefB + 4 * nq, coefB + ncolB + 4 * nq);
                if (kFullB || u < BN * KQ) *reinterpret_cast<float4*>(bs + k * SB + 4 * nq) = v;
            }
        }
    };
    auto sstore = [&](int buf, int k0) {
        if constexpr (BF) sstore_bf16(buf, k0);
        else sstore_from(ra, rb, buf, k0);
    };

    if constexpr (PF > 1) {
        // the host guarantees nchunks % PF == 0 (launch_gemm): the first PF tiles go out together
#pragma unroll
        for (int st = 0; st < PF; ++st) gload_pf(st, kbeg + st * BK);
        __syncthreads();   // coefficient tables are complete
        stage_wait(0);
        sstore_from(rra[0], rrb[0], 0, kbeg);
        __syncthreads();
    } else {
        if (nchunks > 0) gload(kbeg);
        __syncthreads();   // coefficient tables are complete
        if (nchunks > 0) sstore(0, kbeg);
        __syncthreads();
    }

    // MFMA 32x32x2: lanes 0-31 feed k-slot 0, lanes 32-63 k-slot 1 of each step.  Lane (r, h) reads the
    // quads 2q + h of its row, so step 4q + j contracts k = 8q + j (h = 0) with k = 8q + 4 + j (h = 1):
    // every k of the tile exactly once, A and B alike.
    const int frag_h = lane >> 5;
    const int frag_r = lane & 31;
    //   K-contiguous image: offset of quad 2q + h of the lane's row; row-contiguous image: offset of
    //   (k = 4h, lane's row), the step (q, e) then adds (8q + e) * S.
    constexpr int NQ = KQ / 2;             // quad pairs (8 k) per K-tile
    int a_off[TM][NQ], b_off[TN][NQ];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + frag_r;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            a_off[i][q] = A_KC ? row * BK + 4 * ((2 * q + frag_h) ^ swz(row)) : (8 * q + 4 * frag_h) * SA + row;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + frag_r;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            b_off[j][q] = B_KC ? row * BK + 4 * ((2 * q + frag_h) ^ swz(row)) : (8 * q + 4 * frag_h) * SB + row;
    }
    // contraction of the K-tile in LDS buffer `buf` (fp32 operands)
    auto contract = [&](int buf) {
        const float* as = As + buf * TILE_A;
        const float* bs = Bs + buf * TILE_B;
        float af[TM][NQ][4], bf[TN][NQ][4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (A_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(as + a_off[i][q]);
                    af[i][q][0] = v.x; af[i][q][1] = v.y; af[i][q][2] = v.z; af[i][q][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[i][q][e] = as[a_off[i][q] + e * SA];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (B_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(bs + b_off[j][q]);
                    bf[j][q][0] = v.x; bf[j][q][1] = v.y; bf[j][q][2] = v.z; bf[j][q][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bf[j][q][e] = bs[b_off[j][q] + e * SB];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q][e], bf[j][q][e], acc[i][j], 0, 0, 0);
    };
    if constexpr (PF > 1) {
        // stage (c % PF) holds tile c; it was copied into LDS one iteration ago, so tile c + PF may overwrite it now
        for (int c0 = 0; c0 < nchunks; c0 += PF) {
#pragma unroll
            for (int st = 0; st < PF; ++st) {
                const int c = c0 + st;
                const int buf = c & 1;
                gload_pf(st, kbeg + min(c + PF, nchunks - 1) * BK);   // (past the end: a harmless re-read of the last tile)
                contract(buf);
                stage_wait((st + 1) % PF);
                if (c + 1 < nchunks) sstore_from(rra[(st + 1) % PF], rrb[(st + 1) % PF], buf ^ 1, kbeg + (c + 1) * BK);
                __syncthreads();
            }
        }
        // The clamped re-reads of the last tile are still in flight here, into registers whose VALUES nobody will read: to the
        // compiler they are free from the loop's exit on.  A bare wait does not say otherwise -- in the split-K instantiations
        // (EPI_SPLITK) the scheduler moved the epilogue's first address computation above it, into a register a late load then
        // overwrote: a wild output row, a memory fault in ~2 % of 500-step fp32 runs at the C2 shape, whenever the side stream's
        // traffic delayed that load (round 6; vamb_amd/csrc/isa_pending_loads.py finds the pattern in the ISA and is part of the build).
        // Every stage register therefore stays live THROUGH the wait: the empty asm statements below use them after it, and
        // volatile asm statements keep their order.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int st = 0; st < PF; ++st) {
#pragma unroll
            for (int r = 0; r < UA; ++r) asm volatile("" ::"v"(rra[st][r]));
#pragma unroll
            for (int r = 0; r < UB; ++r) asm volatile("" ::"v"(rrb[st][r]));
        }
    } else
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) gload(kbeg + (c + 1) * BK);
        if constexpr (BF) {
            // two 32x32x16 steps per K-tile: lane (r, h) supplies k = 16 t + 8 h .. + 7 of its row (A and B alike)
            const unsigned short* as16 = reinterpret_cast<const unsigned short*>(As + buf * TILE_A);
            const unsigned short* bs16 = reinterpret_cast<const unsigned short*>(Bs + buf * TILE_B);
            bf16x8 a8[TM][2], b8[TN][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 32 + frag_r;
                    a8[i][t] = *reinterpret_cast<const bf16x8*>(as16 + row * 32 + 8 * ((2 * t + frag_h) ^ swzb(row)));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = (wn * TN + j) * 32 + frag_r;
                    b8[j][t] = *reinterpret_cast<const bf16x8*>(bs16 + row * 32 + 8 * ((2 * t + frag_h) ^ swzb(row)));
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i][t], b8[j][t], acc[i][j], 0, 0, 0);
            if (c + 1 < nchunks) sstore(buf ^ 1, kbeg + (c + 1) * BK);
            __syncthreads();
            continue;
        }
        contract(buf);
        if (c + 1 < nchunks) sstore(buf ^ 1, kbeg + (c + 1) * BK);
        __syncthreads();
    }

    // ---------------------------------------------------------------------------------------
    // epilogue.  acc[i][j][reg] is C[row][col] with
    //   row = m0 + (wm*TM + i)*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5),  col = n0 + (wn*TN + j)*32 + (lane&31)
    // ---------------------------------------------------------------------------------------
    if constexpr (KS > 1) {
        // every group is past the K loop's last barrier: the operand tiles are free.  part[group - 1][register][lane]
        float* part = gemm_smem_all;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[(((grp - 1) * TM * TN + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (grp > 0) return;   // (a later barrier of the epilogue counts the live wavefronts only)
#pragma unroll
        for (int q = 1; q < KS; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += part[(((q - 1) * TM * TN + i * TN + j) * 16 + r) * 64 + lane];
        // (one wavefront: its reads of `part` precede, in program order, the epilogue's writes to the same LDS)
    }
    float* Cout = g.C;
    if constexpr (EPI == EPI_SPLITK) Cout += (int64_t)bz * g.slab_stride;

    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    uint64_t drop_key = 0;
    if constexpr (EPI == EPI_HIDDEN_TRAIN) drop_key = step_key(g.drop_key, g.step_ptr);

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cb = (wn * TN + j) * 32 + frag_r;
        const int col = n0 + cb;
        const bool col_ok = col < g.N;
        float bias = 0.f, sc = 1.f, sh = 0.f, cmean = 0.f, cistd = 0.f;
        if constexpr (EPI == EPI_BIAS || EPI == EPI_HIDDEN_TRAIN || EPI == EPI_HIDDEN_EVAL || EPI == EPI_LATENT_MASK)
            bias = col_ok ? g.bias[col] : 0.f;
        if constexpr (EPI == EPI_HIDDEN_EVAL) {
            sc = col_ok ? g.scale[col] : 0.f;
            sh = col_ok ? g.shift[col] : 0.f;
        }
        if constexpr (EPI == EPI_STORE_BNRED) {
            cmean = coefC[cb];
            cistd = coefC[BN + cb];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            uint32_t pair_bits = 0;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * frag_h;
                if (!col_ok || row >= g.M) continue;
                float v = acc[i][j][reg];
                if constexpr (EPI == EPI_BIAS) {
                    v += bias;
                } else if constexpr (EPI == EPI_HIDDEN_TRAIN) {
                    v += bias;
                    v = v > 0.f ? v : kLeakySlopeF * v;
                    if (g.drop_scale != 1.0f || g.drop_mask) {
                        bool keep = row < g.m_real;
                        if (keep) {
                            if (g.drop_mask) {
                                keep = g.drop_mask[(int64_t)row * g.ld_mask + col] != 0;
                            } else {
                                // rows 2p and 2p + 1 of a column share one hash (reg and reg ^ 1 of this lane)
                                if ((reg & 1) == 0) pair_bits = hash_drop(drop_key, (uint32_t)(row >> 1) * (uint32_t)g.N + (uint32_t)col);
                                const uint32_t u16 = (reg & 1) ? (pair_bits >> 16) : (pair_bits & 0xFFFFu);
                                keep = u16 >= (g.drop_thresh >> 16);
                            }
                        }
                        v = keep ? v * g.drop_scale : 0.f;
                    }
                    if (row < g.m_real) { s1[j] += v; s2[j] += v * v; }
                } else if constexpr (EPI == EPI_HIDDEN_EVAL) {
                    v += bias;
                    v = v > 0.f ? v : kLeakySlopeF * v;
                    v = v * sc + sh;
                } else if constexpr (EPI == EPI_LATENT_MASK) {
                    v += bias;
                    v = __uint_as_float(__float_as_uint(v) & 0xFFFFF000u);
                } else if constexpr (EPI == EPI_STORE_BNRED) {
                    if (row < g.m_real) {
                        const float xh = (g.Hbelow[(int64_t)row * g.ldc + col] - cmean) * cistd;
                        s1[j] += v;
                        s2[j] += v * xh;
                    }
                }
                Cout[(int64_t)row * g.ldc + col] = v;
            }
        }
    }

    if constexpr (EPI == EPI_HIDDEN_TRAIN || EPI == EPI_STORE_BNRED) {
        // per-column sums of this workgroup's BM rows, combined in a fixed order, then one fp64 atomic per
        // column and statistic into the layer's [2][N] accumulator
        float* red = gemm_smem;  // [2][WM][BN], reuses the operand tiles (all waves are past the last barrier)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (lane < 32) {
                const int cb = (wn * TN + j) * 32 + lane;
                red[(0 * WM + wm) * BN + cb] = s1[j];
                red[(1 * WM + wm) * BN + cb] = s2[j];
            }
        }
        __syncthreads();
        double* out = EPI == EPI_HIDDEN_TRAIN ? g.fstat_out : g.bstat_out;
        for (int t = tid; t < 2 * BN; t += NT) {
            const int stat = t / BN, cb = t % BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[(stat * WM + w) * BN + cb];
            const int col = n0 + cb;
            if (col < g.N) atomicAdd(&out[(int64_t)stat * g.N + col], (double)s);
        }
    }

}

// dynamic LDS bytes: operand tiles + the coefficient tables of the requested transforms
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int XFA, int XFB, int BK = 32, int DT = 0>
inline size_t gemm_smem_bytes(int k_per_split) {
    size_t fl = DT == 1 ? (size_t)BK * (BM + BN) : 2 * BK * ((A_KC ? BM : BM + 4) + (B_KC ? BN : BN + 4));
    const int narr_a = XFA == XF_BN ? 2 : 0;
    const int narr_b = XFB == XF_BN ? 2 : 0;
    fl += (size_t)narr_a * (A_KC ? k_per_split : BM);
    fl += (size_t)narr_b * (B_KC ? k_per_split : BN);
    if (EPI == EPI_STORE_BNRED) fl += 2 * BN;
    return sizeof(float) * fl;
}

}  // namespace vh
