// gemm_bf16.hpp -- bf16-storage GEMM for the VAE's dense contractions at BASELINE configs C2-C4
// ("bf16 MFMA with fp32 accumulate"), gfx950.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        A: [M][K] bf16, B: [N][K] bf16, both K-contiguous ("NT")
//
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulators.  Unlike gemm.hpp (fp32 tensors in HBM, operands
// rounded while they are staged through registers) every operand here already IS bf16 in memory, so a
// K-tile travels HBM/L2 -> LDS without touching a VGPR:
//   * staging: global_load_lds_dwordx4 (LDS-DMA).  One wave instruction moves 8 tile rows x 128 B (64 bf16)
//     = 1 KiB into a lane-linear LDS span; the 16-byte slots of a row are XOR-permuted on the SOURCE side
//     (lane (row, s) fetches k-slot s ^ f(row)) and read back with the same involution, so the LDS image
//     is the swizzled [rows][64 bf16] tile whose ds_read_b128 fragment reads are bank-conflict free
//     (f(row) = ((row >> 1) ^ (row >> 4)) & 7: the 16 lanes a b128 access services together hit 16 distinct
//     (row & 1, slot) pairs = all 64 banks).
//   * two LDS buffers, BK = 64 (four 32x32x16 steps), next tile's DMA in flight under the MFMAs of the
//     current one, one barrier per K-tile (the structure the CDNA4 guide lists for a one-workgroup-per-CU
//     GEMM whose LDS image can be lane-linear).
//   * pieces outside the matrix (rows >= M / N, k >= the split's end) are fetched from a 16-byte block of
//     zeros instead -- no divergent control flow around the DMA, any K that is a multiple of 8 works.
//   * all four operand layouts the training step needs (forward, dX, dW) are expressed as NT products by
//     keeping a transposed bf16 copy of whatever is contracted along its slow dimension: the producing
//     kernels write it for free from the MFMA C layout (a lane holds 4 consecutive rows of one column).
//
// Epilogues (fp32 math on the accumulators):
//   E16_SPLITK      C32[slab z] = acc                                   (dW, latent-wide outputs)
//   E16_BIAS        C32 = acc + bias[n]                                 (reconstruction)
//   E16_LATENT_MASK C32 = bits(acc + bias) & ~0xFFF, compact [M][N]     (encode, vambtools.py:324-330)
//   E16_HIDDEN_TRAIN h = dropout(leaky_relu(acc + bias)) rounded to bf16; C16 = h, C16T = h^T (if wanted); fp64 batch
//                   sums of h and h^2 taken from the fp32 values before rounding
//   E16_HIDDEN_EVAL C16 = bf16(leaky_relu(acc + bias) * scale[n] + shift[n])
//   E16_STORE_BNRED C16 = bf16(acc) (= dA of the layer below) + fp64 batch sums of dA and dA * xhat(Hbelow)
// bf16 outputs leave through LDS images of the tile (row-major and, for the hidden-train epilogue, transposed) and
// 16-byte stores (the transposed copy as direct 8-byte stores from the accumulator layout cost 4 us per GEMM more).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "gemm.hpp"

namespace vh {

typedef unsigned short bf16_t;   // raw bfloat16 bits

enum Epi16 : int {
    E16_SPLITK = 0,
    E16_BIAS = 1,
    E16_LATENT_MASK = 2,
    E16_HIDDEN_TRAIN = 3,
    E16_HIDDEN_EVAL = 4,
    E16_STORE_BNRED = 5
};

// ---- elementwise backward of one hidden layer (BatchNorm over the batch, dropout, LeakyReLU; vamb/encode.py:259-266 backward)
//   dZ = keep * slope(h) * (ca dA + ch h + c0),   ca = drop_scale istd gamma,  ch = -ca istd S2/B,  c0 = -ca S1/B - ch mean
// with S1 = sum dA, S2 = sum dA xhat over the batch.  ONE definition of the coefficients and of the element, with explicit
// fused multiply-adds (vae_dz16_kernel).  (Round 6 also formed dZ inside the consuming input-gradient GEMM, on its A-operand
// path through registers -- bit-identical, and slower: profiles/r06c_*, DESIGN.md section 4.3 -- and removed that kernel again.)
struct DzCoefSrc {
    const float* mean;     // [n_p] mean / 1/std of the layer's BatchNorm as the forward fold left them (vae_fold_bn_kernel)
    const float* istd;
    const float* gamma;
    const double* bstat;   // [2][n_p] fp64 batch sums of dA and dA * xhat (E16_STORE_BNRED of the GEMM that produced dA)
    int n_p;
    int stat_bs;           // the statistics' batch (all ranks under SyncBN)
    float drop_scale;
};
__device__ __forceinline__ void dz16_coeffs(const DzCoefSrc& s, int col, float& ca, float& ch, float& c0) {
    const float mean = s.mean[col], istd = s.istd[col];
    const double inv_bs = 1.0 / (double)s.stat_bs;
    const float c1 = (float)(s.bstat[col] * inv_bs);
    const float c2 = (float)(s.bstat[s.n_p + col] * inv_bs);
    ca = s.drop_scale * istd * s.gamma[col];
    ch = -ca * istd * c2;
    c0 = __builtin_fmaf(-ch, mean, -(ca * c1));
}
// hashed_drop: the forward pass dropped by the counter hash and stored exact zeros for dropped elements
__device__ __forceinline__ float dz16_elem(float d, float h, float ca, float ch, float c0, bool hashed_drop) {
    const float l = __builtin_fmaf(ca, d, __builtin_fmaf(ch, h, c0));
    const float dz = l * (h > 0.f ? 1.0f : kLeakySlopeF);
    return (hashed_drop && h == 0.f) ? 0.f : dz;
}
struct Gemm16Args {
    const bf16_t* A;
    int64_t lda;
    const bf16_t* B;
    int64_t ldb;
    int M, N, K;
    int k_per_split;       // contraction elements per blockIdx.z (multiple of 64 unless there is one split)
    int64_t slab_stride;   // elements between split-K slabs of C32
    const bf16_t* zeros;   // >= 16 bytes of zeros
    float* C32;
    int64_t ldc32;
    bf16_t* C16;
    int64_t ldc16;
    bf16_t* C16T;          // [N][M] (may be nullptr)
    int64_t ldc16t;
    const float* bias;
    const float* scale;
    const float* shift;
    int m_real;            // rows that belong to the batch (statistics / dropout rows)
    double* fstat_out;     // E16_HIDDEN_TRAIN: [2][N]
    float drop_scale;
    uint32_t drop_thresh;
    uint64_t drop_key;
    const unsigned long long* step_ptr;
    const uint8_t* drop_mask;
    int64_t ld_mask;
    // E16_STORE_BNRED
    const bf16_t* Hbelow;  // raw activations of the layer below, [M][N] bf16 with leading dimension ldh
    int64_t ldh;
    BnSrc bnC;
    const float* bn_mean;  // mean / 1/std of the layer below, as the kernel that folded its BatchNorm left them ([N] floats each)
    const float* bn_istd;
    double* bstat_out;     // [2][N]
    int xcd_remap;
    int dbg;               // timing experiments (vh_debug_gemm16): 1 no fp64 atomics, 2 no transposed copy, 4 no row-major copy
    unsigned long long* tstamps;   // diagnostic (vh_debug_gemm16, variant flag 8): [workgroup][8] s_memtime stamps -- entry, first tile
                                   // landed, K loop done, epilogue phase 1 done, stores issued, exit
};

__device__ __forceinline__ bf16_t f2bf(float x) {
    const __bf16 b = (__bf16)x;   // round to nearest even
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float((uint32_t)b << 16); }
// two floats -> one dword of two bf16 (lo in bits 0-15), round to nearest even: ONE v_cvt_pk_bf16_f32.  (Written as two scalar
// conversions + shift + or, the compiler pairs the conversions of elements 0 / 2 and 1 / 3 of a quad and re-pairs the halves with
// four more instructions per quad: found in the ISA of the training epilogue, round 6.)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(f, bf16x2_t);
    return __builtin_bit_cast(uint32_t, b);
}

// slot permutation of the [rows][64 bf16] LDS image (8 slots of 16 B per 128-byte row)
__device__ __forceinline__ int swz16(int row) { return ((row >> 1) ^ (row >> 4)) & 7; }

__device__ __forceinline__ void glds16(const bf16_t* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ds_read_b64_tr_b16: sixteen lanes address a [4][16] block of 16-bit elements (lane s: row s / 4, elements 4 (s % 4) .. + 3 of
// it, 8 contiguous bytes; the row stride is free); lane c of the group receives element c of the four rows (measured mapping:
// profiles/r03a_tr16_probe.txt).
typedef short tr_v4s __attribute__((ext_vector_type(4)));
typedef short tr_v8s __attribute__((ext_vector_type(8)));
__device__ __forceinline__ tr_v4s lds_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(p));
}

__device__ __forceinline__ void gemm16_stamp(const Gemm16Args& g, int slot) {
    if (g.tstamps != nullptr && threadIdx.x == 0) {
        const unsigned nb = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        g.tstamps[(size_t)nb * 8 + slot] = __builtin_amdgcn_s_memtime();
    }
}

// STG: how a K-tile reaches the LDS.
//   0 = LDS-DMA (global_load_lds_dwordx4), two buffers, the whole next tile requested in front of the current tile's MFMAs
//   1 = through registers (global_load_dwordx4 issued before the MFMAs of the current tile, ds_write_b128 after them)
//   2 = LDS-DMA, THREE buffers, the pieces of tile t + 2 issued one by one BETWEEN the MFMA groups of tile t, counted
//       vmcnt + raw s_barrier.  Why (profiles/r03a_loadpath_*.json, tests/micro/loadpath.hip): the DMA path alone moves a
//       32 KB K-tile in 0.39 us with a drain per tile and in 0.25 us with two tiles in flight (54 B/clk/CU = the L2's rate),
//       yet the STG = 0 loop needs 0.84 us per tile: a global_load_lds occupies its wave until the texture addresser takes
//       it (~17-27 clk per 1 KiB piece, 32 pieces per tile and CU), all waves run in lockstep, so "issue 4 pieces, then
//       12 ds_reads + 8 MFMAs, then drain" is a DMA phase FOLLOWED by a matrix phase.  Spreading the pieces over the MFMA
//       groups puts the issue stalls under matrix-pipe time, and the third buffer takes the landing latency off the barrier.
// TAG: no effect on the code -- a launch site that wants its own line in a kernel trace instantiates its own copy (the K = D
// encoder GEMM of the training step, the roofline kernel of bench.py: TAG 1; until round 6 it shared one name with the three
// other hidden-layer launches of a step and rocprofv3 --stats could only average over the four).
template <int BM, int BN, int WM, int WN, int EPI, int STG = 0, int TAG = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(const Gemm16Args g) {
    constexpr int NWAVE = WM * WN;
    constexpr int NT = NWAVE * 64;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int BK = 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;   // one buffer of each operand
    constexpr int PA = BM / 8, PB = BN / 8;                 // 1 KiB DMA pieces per K-tile
    constexpr int RA = PA / NWAVE, RB = PB / NWAVE;         // pieces per wave
    constexpr int NBUF = STG == 2 ? 3 : 2;
    static_assert(TM >= 1 && TN >= 1 && PA % NWAVE == 0 && PB % NWAVE == 0, "tile / wave layout");
    // ALL LDS of the kernel is this one array (a second __shared__ object makes hipcc drain the DMA queue
    // in front of every fragment read)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned char* const As = smem16;                     // [NBUF][A_BYTES]
    unsigned char* const Bs = smem16 + NBUF * A_BYTES;    // [NBUF][B_BYTES]

    gemm16_stamp(g, 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd_remap) {   // XCD x gets the x-th contiguous eighth of the (z, m, n)-ordered tile list (speed only)
        const int gx = gridDim.x, gy = gridDim.y;
        const int nwg = gx * gy * (int)gridDim.z;
        const int bid = bx + gx * (by + gy * bz);
        const int xcd = bid & 7, local = bid >> 3;
        const int t = xcd * (nwg >> 3) + min(xcd, nwg & 7) + local;
        bx = t % gx;
        by = (t / gx) % gy;
        bz = t / (gx * gy);
    }
    // (the remap's divisions run on the VALU: without this the compiler treats every tile coordinate -- and with them the K
    // loop's trip count and the DMA guards -- as divergent and wraps them in exec-mask branches)
    bx = __builtin_amdgcn_readfirstlane(bx);
    by = __builtin_amdgcn_readfirstlane(by);
    bz = __builtin_amdgcn_readfirstlane(bz);
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = bz * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    // ---- DMA source addresses: lane (row = 8 piece + lane / 8, LDS slot s = lane % 8) fetches k-slot s ^ f(row)
    const bf16_t* a_src[RA];
    const bf16_t* b_src[RB];
    int a_k[RA], b_k[RB];      // k offset of the lane's slot inside a K-tile; >= BK marks a row outside the matrix
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int row = 8 * (wave + NWAVE * r) + (lane >> 3);
        const int ks = 8 * ((lane & 7) ^ swz16(row));
        const bool ok = m0 + row < g.M;
        a_k[r] = ok ? ks : (1 << 30);
        a_src[r] = g.A + (int64_t)(ok ? m0 + row : 0) * g.lda + ks;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = 8 * (wave + NWAVE * r) + (lane >> 3);
        const int ks = 8 * ((lane & 7) ^ swz16(row));
        const bool ok = n0 + row < g.N;
        b_k[r] = ok ? ks : (1 << 30);
        b_src[r] = g.B + (int64_t)(ok ? n0 + row : 0) * g.ldb + ks;
    }
    auto stage = [&](unsigned char* abuf, unsigned char* bbuf, int k0) {
        const int room = kend - k0;   // slots with k offset >= room are past the end of this split
#pragma unroll
        for (int r = 0; r < RA; ++r)
            glds16(a_k[r] < room ? a_src[r] + k0 : g.zeros, abuf + (wave + NWAVE * r) * 1024);
#pragma unroll
        for (int r = 0; r < RB; ++r)
            glds16(b_k[r] < room ? b_src[r] + k0 : g.zeros, bbuf + (wave + NWAVE * r) * 1024);
    };

    // ---- fragment addresses: lane (r = lane & 31, h = lane >> 5) reads k = 16 t + 8 h .. + 7 of its row
    const int frag_r = lane & 31, frag_h = lane >> 5;
    int a_off[TM], a_swz[TM], b_off[TN], b_swz[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + frag_r;
        a_off[i] = row * 128;
        a_swz[i] = swz16(row);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + frag_r;
        b_off[j] = row * 128;
        b_swz[j] = swz16(row);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- epilogue operands that do not depend on the product: per-column bias, the dropout stream's seed (a load of the
    // device-resident step counter + a hash), the BatchNorm coefficients of the layer below (fp64 batch sums) and the tile of its
    // activations.  Left at the head of the epilogue they formed a chain of dependent round trips -- kernel argument, step
    // counter, hash; kernel argument, bias -- each behind its own s_waitcnt, ~2 k clk per GEMM with every matrix pipe idle
    // (found in the ISA, round 5).  Now: epilogue_loads() issues the raw loads in FRONT of the first K-tile's DMAs (older than
    // them, so the counted vmcnt waits of the K loop are unchanged and the values have landed when tile 0 has) and
    // pin_epilogue_operands() after the loop keeps them live in registers across it (the compiler does not move loads over the
    // loop's asm statements by itself).  mean / 1/std of the layer below come as floats from the kernel that folded its
    // BatchNorm (vae_fold_bn_kernel), so no fp64 arithmetic is left in this kernel.
    [[maybe_unused]] float bias_pre[TN];
    [[maybe_unused]] float sc_pre[TN], sh_pre[TN];
    [[maybe_unused]] unsigned long long step_raw = 0ull;
    // E16_STORE_BNRED: a wave instruction of the store pass covers 16 rows x 32 columns (lane = 16 g4 + r16: row r16 of the block,
    // columns 8 g4 .. + 7); this thread's 8 columns and its NU row blocks are fixed by (wave, lane)
    constexpr int UC_PRE = BN / 32, UR_PRE = BM / 16, WG_PRE = (NWAVE / UC_PRE) > 0 ? NWAVE / UC_PRE : 1;
    constexpr int NU_PRE = (UR_PRE / WG_PRE) > 0 ? UR_PRE / WG_PRE : 1;
    [[maybe_unused]] float4 mean_pre[2], istd_pre[2];
    [[maybe_unused]] uint4 hv_pre[NU_PRE];
    // Every load below is UNCONDITIONAL with a clamped address (a load inside a branch makes the compiler wait for it at the
    // join, in front of the DMAs); edge tiles, which take the general epilogue, simply do not use the values.
    auto epilogue_loads = [&]() {
        if constexpr (EPI == E16_HIDDEN_TRAIN || EPI == E16_HIDDEN_EVAL || EPI == E16_BIAS || EPI == E16_LATENT_MASK) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = min(n0 + (wn * TN + j) * 32 + (lane & 31), g.N - 1);
                bias_pre[j] = g.bias[col];
                sc_pre[j] = 1.f; sh_pre[j] = 0.f;
                if constexpr (EPI == E16_HIDDEN_EVAL) { sc_pre[j] = g.scale[col]; sh_pre[j] = g.shift[col]; }
            }
            if constexpr (EPI == E16_HIDDEN_TRAIN) {
                const unsigned long long* sp = g.step_ptr ? g.step_ptr : reinterpret_cast<const unsigned long long*>(g.zeros);
                step_raw = *sp;
            }
        }
        if constexpr (EPI == E16_STORE_BNRED) {
            const int c0p = min(n0 + (wave % UC_PRE) * 32 + 8 * (lane >> 4), g.N - 8);
            mean_pre[0] = *reinterpret_cast<const float4*>(g.bn_mean + c0p);
            mean_pre[1] = *reinterpret_cast<const float4*>(g.bn_mean + c0p + 4);
            istd_pre[0] = *reinterpret_cast<const float4*>(g.bn_istd + c0p);
            istd_pre[1] = *reinterpret_cast<const float4*>(g.bn_istd + c0p + 4);
#pragma unroll
            for (int k = 0; k < NU_PRE; ++k) {
                const int row = min(m0 + (wave / UC_PRE + WG_PRE * k) * 16 + (lane & 15), g.M - 1);
                hv_pre[k] = *reinterpret_cast<const uint4*>(g.Hbelow + (int64_t)row * g.ldh + c0p);
            }
        }
    };
    // (an empty asm that "reads" the values: the compiler must have them in registers at this point of the program)
    auto pin_epilogue_operands = [&]() {
        if constexpr (EPI == E16_BIAS || EPI == E16_LATENT_MASK) {
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bias_pre[j]));
        }
        if constexpr (EPI == E16_HIDDEN_TRAIN || EPI == E16_HIDDEN_EVAL) {
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bias_pre[j]), "v"(sc_pre[j]), "v"(sh_pre[j]));
            asm volatile("" ::"v"((uint32_t)step_raw), "v"((uint32_t)(step_raw >> 32)));
        }
        if constexpr (EPI == E16_STORE_BNRED) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                asm volatile("" ::"v"(mean_pre[q].x), "v"(mean_pre[q].y), "v"(mean_pre[q].z), "v"(mean_pre[q].w), "v"(istd_pre[q].x),
                             "v"(istd_pre[q].y), "v"(istd_pre[q].z), "v"(istd_pre[q].w));
#pragma unroll
            for (int k = 0; k < NU_PRE; ++k) asm volatile("" ::"v"(hv_pre[k].x), "v"(hv_pre[k].y), "v"(hv_pre[k].z), "v"(hv_pre[k].w));
        }
    };

    auto compute = [&](const unsigned char* abuf, const unsigned char* bbuf) {
        // Software pipeline inside the K-tile: the fragments of step t + 2 are requested while step t multiplies
        // (hipcc on its own sinks every ds_read_b128 to just in front of its MFMA and waits lgkmcnt(0) four times per
        // tile).  sched_barrier(0) pins the order; the compiler still places the counted lgkmcnt waits.
        bf16x8 a8[BK / 16][TM], b8[BK / 16][TN];
        auto frags = [&](int t) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a8[t][i] = *reinterpret_cast<const bf16x8*>(abuf + a_off[i] + 16 * ((2 * t + frag_h) ^ a_swz[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b8[t][j] = *reinterpret_cast<const bf16x8*>(bbuf + b_off[j] + 16 * ((2 * t + frag_h) ^ b_swz[j]));
        };
        auto mfmas = [&](int t) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[t][i], b8[t][j], acc[i][j], 0, 0, 0);
        };
        frags(0);
        frags(1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        frags(2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        frags(3);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(2);
        mfmas(3);
        __builtin_amdgcn_sched_barrier(0);   // the barrier (and its DMA drain) stays BEHIND the MFMAs
    };

    // K loop, unrolled by two so that every LDS address is a compile-time offset of the one array.  (A third buffer with
    // the DMA of tile t + 2 in flight across the barrier measured 0-8 % slower: the loop is bound by the per-CU LDS-DMA
    // rate, not by its latency.)
    epilogue_loads();
    if constexpr (STG == 0) {
        if (nk > 0) stage(As, Bs, kbeg);
        __syncthreads();   // (hipcc drains the DMA queue in front of this barrier)
        gemm16_stamp(g, 1);
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            stage(As + A_BYTES, Bs + B_BYTES, kbeg + (kt + 1) * BK);
            compute(As, Bs);
            __syncthreads();
            if (kt + 2 < nk) stage(As, Bs, kbeg + (kt + 2) * BK);
            compute(As + A_BYTES, Bs + B_BYTES);
            __syncthreads();
        }
        if (kt < nk) {   // odd number of K-tiles: the last one sits in buffer 0
            compute(As, Bs);
            __syncthreads();
        }
    } else if constexpr (STG == 2) {
        // one DMA piece of the tile starting at k0 (q < RA: A piece q, else B piece q - RA)
        auto piece = [&](unsigned char* abuf, unsigned char* bbuf, int k0, int q) {
            const int room = kend - k0;
            if (q < RA) glds16(a_k[q] < room ? a_src[q] + k0 : g.zeros, abuf + (wave + NWAVE * q) * 1024);
            else glds16(b_k[q - RA] < room ? b_src[q - RA] + k0 : g.zeros, bbuf + (wave + NWAVE * (q - RA)) * 1024);
        };
        constexpr int NP = RA + RB;   // pieces per wave and tile
        // MFMA groups of tile `cur`, the pieces of the tile two ahead in between; fragments two k-steps ahead as in compute()
        auto compute_stage = [&](const unsigned char* abuf, const unsigned char* bbuf, unsigned char* anext, unsigned char* bnext,
                                 int knext, bool prefetch) {
            bf16x8 a8[BK / 16][TM], b8[BK / 16][TN];
            auto frags = [&](int t) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a8[t][i] = *reinterpret_cast<const bf16x8*>(abuf + a_off[i] + 16 * ((2 * t + frag_h) ^ a_swz[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b8[t][j] = *reinterpret_cast<const bf16x8*>(bbuf + b_off[j] + 16 * ((2 * t + frag_h) ^ b_swz[j]));
            };
            frags(0);
            frags(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < BK / 16; ++t) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[t][i], b8[t][j], acc[i][j], 0, 0, 0);
                if (prefetch) {   // workgroup-uniform
#pragma unroll
                    for (int q = (NP * t) / 4; q < (NP * (t + 1)) / 4; ++q) piece(anext, bnext, knext, q);
                }
                if (t + 2 < BK / 16) frags(t + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto abuf = [&](int b) { return As + b * A_BYTES; };
        auto bbuf = [&](int b) { return Bs + b * B_BYTES; };
        // prologue: tiles 0 and 1 requested; tile 0 awaited (tile 1's NP pieces may stay in flight)
        if (nk > 0) stage(abuf(0), bbuf(0), kbeg);
        if (nk > 1) stage(abuf(1), bbuf(1), kbeg + BK);
        if (nk > 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        gemm16_stamp(g, 1);
        // iteration kt computes buffer kt % 3 and requests tile kt + 2 into buffer (kt + 2) % 3 -- the buffer every wave
        // finished reading before the barrier that ended iteration kt - 1.  At its end the pieces of tile kt + 1 (requested
        // during iteration kt - 1) must have landed: all but the NP newest DMAs of this wave, then the workgroup barrier.
        auto iteration = [&](int kt, int cur, auto steady) {
            // steady: tile kt + 2 exists for sure (no branch in the loop body)
            const bool pre = decltype(steady)::value ? true : kt + 2 < nk;
            const int nxt = cur == 0 ? 2 : cur - 1;   // (cur + 2) % 3
            compute_stage(abuf(cur), bbuf(cur), abuf(nxt), bbuf(nxt), kbeg + (kt + 2) * BK, pre);
            if (pre) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 4 < nk; kt += 3) {   // all three iterations prefetch: kt + 2 + 2 < nk
            iteration(kt, 0, std::true_type{});
            iteration(kt + 1, 1, std::true_type{});
            iteration(kt + 2, 2, std::true_type{});
        }
        // at most four tiles left, starting in buffer 0
        if (kt < nk) iteration(kt, 0, std::false_type{});
        if (kt + 1 < nk) iteration(kt + 1, 1, std::false_type{});
        if (kt + 2 < nk) iteration(kt + 2, 2, std::false_type{});
        if (kt + 3 < nk) iteration(kt + 3, 0, std::false_type{});
    } else {
        uint4 ra[RA], rb[RB];
        auto fetch = [&](int k0) {
            const int room = kend - k0;
#pragma unroll
            for (int r = 0; r < RA; ++r) ra[r] = *reinterpret_cast<const uint4*>(a_k[r] < room ? a_src[r] + k0 : g.zeros);
#pragma unroll
            for (int r = 0; r < RB; ++r) rb[r] = *reinterpret_cast<const uint4*>(b_k[r] < room ? b_src[r] + k0 : g.zeros);
        };
        auto put = [&](unsigned char* abuf, unsigned char* bbuf) {
#pragma unroll
            for (int r = 0; r < RA; ++r) *reinterpret_cast<uint4*>(abuf + (wave + NWAVE * r) * 1024 + lane * 16) = ra[r];
#pragma unroll
            for (int r = 0; r < RB; ++r) *reinterpret_cast<uint4*>(bbuf + (wave + NWAVE * r) * 1024 + lane * 16) = rb[r];
        };
        if (nk > 0) {
            fetch(kbeg);
            put(As, Bs);
        }
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            fetch(kbeg + (kt + 1) * BK);
            compute(As, Bs);
            put(As + A_BYTES, Bs + B_BYTES);
            __syncthreads();
            if (kt + 2 < nk) fetch(kbeg + (kt + 2) * BK);
            compute(As + A_BYTES, Bs + B_BYTES);
            if (kt + 2 < nk) put(As, Bs);
            __syncthreads();
        }
        if (kt < nk) {
            compute(As, Bs);
            __syncthreads();
        }
    }

    pin_epilogue_operands();
    gemm16_stamp(g, 2);
    // ---------------------------------------------------------------------------------------------------
    // epilogue.  acc[i][j][reg] is C[row][col] with
    //   row = m0 + (wm*TM + i)*32 + (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5),   col = n0 + (wn*TN + j)*32 + (lane & 31)
    // ---------------------------------------------------------------------------------------------------
    if constexpr (EPI == E16_SPLITK || EPI == E16_BIAS || EPI == E16_LATENT_MASK) {
        float* Cout = g.C32;
        if constexpr (EPI == E16_SPLITK) Cout += (int64_t)bz * g.slab_stride;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + frag_r;
            const bool col_ok = col < g.N;
            float bias = 0.f;
            if constexpr (EPI != E16_SPLITK) bias = col_ok ? bias_pre[j] : 0.f;   // (loaded in the prologue)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * frag_h;
                    if (!col_ok || row >= g.M) continue;
                    float v = acc[i][j][reg] + bias;
                    if constexpr (EPI == E16_LATENT_MASK) v = __uint_as_float(__float_as_uint(v) & 0xFFFFF000u);
                    Cout[(int64_t)row * g.ldc32 + col] = v;
                }
        }
        gemm16_stamp(g, 4);
        return;
    } else {
        // ---- lean epilogue: tiles that lie completely inside the batch and the matrix (all of them at the BASELINE shapes), no
        // injected dropout masks.  The round-2 epilogue below spent 9 500 clk per 128 x 128 tile in its accumulator-layout pass
        // (profiles/r03c_gemm16_timeline.txt: twice the K loop of the C2 encoder GEMM) -- ~31 instructions per element: a
        // predicate and a branch per element for the matrix edges, two integer-multiply hashes per four elements, one
        // ds_write_b16 per element.  Here: no edge tests; dropout bits from a per-lane xorshift32 stream seeded once per launch
        // by the counter hash (6 full-rate VALU per 32 bits); the image is written TRANSPOSED, [BN][BM + 8], four consecutive
        // rows of the lane's column per ds_write_b64 (8 stores per lane instead of 32), and the row-major 16-byte chunks of
        // the global stores are assembled by the transposing LDS read.
        bool lean = m0 + BM <= g.m_real && m0 + BM <= g.M && n0 + BN <= g.N && !(g.dbg & 16);
        if constexpr (EPI == E16_HIDDEN_TRAIN) lean = lean && g.drop_mask == nullptr && (g.C16T == nullptr || (g.dbg & 2));
        if constexpr (NWAVE % (BN / 32) != 0 || BM % 16 != 0) lean = false;
        if (lean) {
            constexpr int IP = BM + 8;                         // elements per image row (one output COLUMN)
            constexpr int IMG_BYTES = BN * IP * 2;
            bf16_t* const img = reinterpret_cast<bf16_t*>(smem16);
            float* const lred = reinterpret_cast<float*>(smem16 + IMG_BYTES);
            float s1[TN], s2[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
            uint32_t rng = 1u;
            float dsa = 1.0f, dsb = kLeakySlopeF;
            bool hashed = false;
            uint32_t thresh16 = 0;
            if constexpr (EPI == E16_HIDDEN_TRAIN) {
                hashed = g.drop_scale != 1.0f;
                dsa = g.drop_scale;
                dsb = g.drop_scale * kLeakySlopeF;
                thresh16 = g.drop_thresh >> 16;
                if (hashed) {   // (the step counter was loaded in the prologue)
                    const uint64_t key = g.step_ptr ? (g.drop_key ^ ((uint64_t)step_raw << 8)) : g.drop_key;
                    rng = hash_drop(key, (uint32_t)(by * (int)gridDim.x + bx) * NT + tid) | 1u;
                }
            }
            // the dropout stream of a lane: seeded once per launch by the counter hash, then x += x << 13; x ^= x >> 17; x += x << 5
            // per 32 bits (two odd multiplications around a xor-shift: 4 full-rate instructions; the xorshift32 of rounds 3-5 took 6).
            // A lane draws 16 words per launch from a hashed seed -- checked on 4 M seeds: rate 0.2 +- 4e-4 at every position,
            // no correlation above 4 sigma between successive draws, adjacent lanes or distant positions (same as xorshift32).
            // (the empty asm keeps the compiler from folding the last shift-add of one draw and the first of the next into one
            // quarter-rate v_mul_lo_u32 by 0x42021)
            auto draw = [&]() { rng += rng << 13; rng ^= rng >> 17; rng += rng << 5; asm volatile("" : "+v"(rng)); return rng; };
            // HASHED chosen once per workgroup (tested inside the loops it cost a scalar branch per quad)
            auto phase_lean = [&](auto hashed_c) {
                constexpr bool HASHED = decltype(hashed_c)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = (wn * TN + j) * 32 + frag_r;
                const float bias = bias_pre[j], sc = sc_pre[j], sh = sh_pre[j];   // loaded in the prologue
                // drop_scale * leaky_relu(acc + bias) = max of two fused multiply-adds on the accumulator (round 6: one instruction
                // per element fewer than add + two multiplies)
                [[maybe_unused]] const float ba = bias * dsa, bb = bias * dsb;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                        if constexpr (EPI == E16_HIDDEN_TRAIN) {
                            uint32_t r0 = 0xFFFFFFFFu, r1 = 0xFFFFFFFFu;
                            if constexpr (HASHED) {   // two 32-bit draws = four 16-bit uniforms
                                r0 = draw();
                                r1 = draw();
                            }
                            const uint32_t u[4] = {r0 & 0xFFFFu, r0 >> 16, r1 & 0xFFFFu, r1 >> 16};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float y;
                                if constexpr (HASHED) {
                                    y = fmaxf(__builtin_fmaf(v[e], dsa, ba), __builtin_fmaf(v[e], dsb, bb));
                                } else {   // no dropout (drop_scale 1): the arithmetic of the general epilogue, bit for bit
                                    const float x = v[e] + bias;
                                    y = fmaxf(x * dsa, x * dsb);
                                }
                                y = u[e] >= thresh16 ? y : 0.f;
                                s1[j] += y;
                                s2[j] = fmaf(y, y, s2[j]);
                                v[e] = y;
                            }
                        } else if constexpr (EPI == E16_HIDDEN_EVAL) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x = v[e] + bias;
                                v[e] = fmaxf(x, x * kLeakySlopeF) * sc + sh;
                            }
                        }
                        const int rl4 = (wm * TM + i) * 32 + 8 * q + 4 * frag_h;
                        uint2 w;
                        w.x = pack2bf(v[0], v[1]);
                        w.y = pack2bf(v[2], v[3]);
                        *reinterpret_cast<uint2*>(img + cl * IP + rl4) = w;
                    }
            }
            };
            if (hashed) phase_lean(std::true_type{});
            else phase_lean(std::false_type{});
            if constexpr (EPI == E16_HIDDEN_TRAIN) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    s1[j] += __shfl_xor(s1[j], 32);
                    s2[j] += __shfl_xor(s2[j], 32);
                    if (lane < 32) {
                        const int cl = (wn * TN + j) * 32 + lane;
                        lred[(0 * WM + wm) * BN + cl] = s1[j];
                        lred[(1 * WM + wm) * BN + cl] = s2[j];
                    }
                }
            }
            __syncthreads();
            gemm16_stamp(g, 3);
            // row-major stores: a wave instruction covers 16 rows x 32 columns (lane = 16 g4 + r16: row r16 of the block, columns
            // 8 g4 .. + 7: two transposing reads of [4 columns][16 rows] blocks of the image)
            constexpr int UC = BN / 32, UR = BM / 16, WG = NWAVE / UC;   // column passes, row blocks, waves per column pass
            const int g4 = lane >> 4, r16 = lane & 15;
            const int cp = wave % UC;
            const int c0 = cp * 32 + 8 * g4;
            const bf16_t* const isrc = img + (c0 + (r16 >> 2)) * IP + 4 * (r16 & 3);
            float mean8[8], istd8[8], t1[8], t2[8];
            if constexpr (EPI == E16_STORE_BNRED) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {   // (coefficients of the layer below: loaded in the prologue)
                    mean8[e] = (&mean_pre[e >> 2].x)[e & 3]; istd8[e] = (&istd_pre[e >> 2].x)[e & 3];
                    t1[e] = 0.f; t2[e] = 0.f;
                }
            }
            constexpr int NU = UR / WG;    // row blocks per wave
            static_assert(UR % WG == 0, "row blocks per wave");
            static_assert(NU == NU_PRE && UC == UC_PRE && WG == WG_PRE, "prologue prefetch layout");
            uint4 hv[NU];
            if constexpr (EPI == E16_STORE_BNRED) {
#pragma unroll
                for (int k = 0; k < NU; ++k) hv[k] = hv_pre[k];   // (Hbelow tile: requested in the prologue)
            }
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                const int rb = wave / UC + WG * k;
                const tr_v4s lo = lds_tr16(isrc + rb * 16);
                const tr_v4s hi = lds_tr16(isrc + 4 * IP + rb * 16);
                uint4 o;
                o.x = (uint32_t)(unsigned short)lo[0] | ((uint32_t)(unsigned short)lo[1] << 16);
                o.y = (uint32_t)(unsigned short)lo[2] | ((uint32_t)(unsigned short)lo[3] << 16);
                o.z = (uint32_t)(unsigned short)hi[0] | ((uint32_t)(unsigned short)hi[1] << 16);
                o.w = (uint32_t)(unsigned short)hi[2] | ((uint32_t)(unsigned short)hi[3] << 16);
                const int row = m0 + rb * 16 + r16;
                if (!(g.dbg & 4)) *reinterpret_cast<uint4*>(g.C16 + (int64_t)row * g.ldc16 + n0 + c0) = o;
                if constexpr (EPI == E16_STORE_BNRED) {
                    const uint32_t dw[4] = {o.x, o.y, o.z, o.w};
                    const uint32_t hw[4] = {hv[k].x, hv[k].y, hv[k].z, hv[k].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = __uint_as_float((e & 1) ? (dw[e >> 1] & 0xFFFF0000u) : (dw[e >> 1] << 16));
                        const float hh = __uint_as_float((e & 1) ? (hw[e >> 1] & 0xFFFF0000u) : (hw[e >> 1] << 16));
                        t1[e] += d;
                        t2[e] = fmaf(d, (hh - mean8[e]) * istd8[e], t2[e]);
                    }
                }
            }
            if constexpr (EPI == E16_STORE_BNRED) {
                // the 16 row-lanes of a group and the WG waves of a column pass hold partial sums of the same 8 columns
                float* const bred = reinterpret_cast<float*>(smem16 + IMG_BYTES);   // [2][WG * 16][BN]
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    bred[(0 * WG * 16 + (wave / UC) * 16 + r16) * BN + c0 + e] = t1[e];
                    bred[(1 * WG * 16 + (wave / UC) * 16 + r16) * BN + c0 + e] = t2[e];
                }
                __syncthreads();
                for (int t = tid; t < 2 * BN; t += NT) {
                    const int stat = t / BN, cb = t % BN;
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < WG * 16; ++w) s += bred[(stat * WG * 16 + w) * BN + cb];
                    atomicAdd(&g.bstat_out[(int64_t)stat * g.N + n0 + cb], (double)s);
                }
            }
            if constexpr (EPI == E16_HIDDEN_TRAIN) {
                for (int t = tid; t < 2 * BN; t += NT) {
                    const int stat = t / BN, cb = t % BN;
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) s += lred[(stat * WM + w) * BN + cb];
                    if (!(g.dbg & 1)) atomicAdd(&g.fstat_out[(int64_t)stat * g.N + n0 + cb], (double)s);
                }
            }
            gemm16_stamp(g, 4);
            return;
        }
        // ---- general epilogue (edge tiles, injected dropout masks, the transposed copy of the round-2 dataflow)
        // bf16 image of the output tile in LDS: [BM][CP] (operand buffers are free: every wave is past the last barrier)
        constexpr int CP = BN + 8;                         // elements per image row (keeps 16-byte alignment)
        constexpr int CT_BYTES = BM * CP * 2;
        bf16_t* const ct = reinterpret_cast<bf16_t*>(smem16);
        // second image, transposed [BN][CPT], for the transposed copy (hidden-train epilogue only)
        constexpr int CPT = BM + 8;
        constexpr int CTT_BYTES = EPI == E16_HIDDEN_TRAIN ? BN * CPT * 2 : 0;
        bf16_t* const ctT = reinterpret_cast<bf16_t*>(smem16 + CT_BYTES);
        float* const red = reinterpret_cast<float*>(smem16 + CT_BYTES + CTT_BYTES);   // reduction scratch behind the images
        float s1[TN], s2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        uint64_t drop_key = 0;
        if constexpr (EPI == E16_HIDDEN_TRAIN) drop_key = step_key(g.drop_key, g.step_ptr);
        // phase 1 (accumulator layout): transform, round, write the image (+ the transposed copy, + the sums).
        // DROP = 0: no dropout, 1: counter-based hash, 2: injected keep-masks -- chosen once per workgroup
        const bool want_t = g.C16T != nullptr && !(g.dbg & 2);   // the transposed copy exists only in the round-2 dataflow
        auto phase1 = [&](auto drop_mode) {
            constexpr int DROP = decltype(drop_mode)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = (wn * TN + j) * 32 + frag_r;
                const int col = n0 + cl;
                const bool col_ok = col < g.N;
                float bias = 0.f, sc = 1.f, sh = 0.f;
                if constexpr (EPI == E16_HIDDEN_TRAIN || EPI == E16_HIDDEN_EVAL) bias = col_ok ? g.bias[col] : 0.f;
                if constexpr (EPI == E16_HIDDEN_EVAL) {
                    sc = col_ok ? g.scale[col] : 0.f;
                    sh = col_ok ? g.shift[col] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16_t hb[4];
                        uint32_t pair_bits = 0;
                        const int rl4 = (wm * TM + i) * 32 + 8 * q + 4 * frag_h;
                        if constexpr (DROP == 1) {
                            // rows 2p and 2p + 1 of a column share one hash (same rule as gemm.hpp); rl4 is a multiple of 4
                            pair_bits = hash_drop(drop_key, (uint32_t)((m0 + rl4) >> 1) * (uint32_t)g.N + (uint32_t)col);
                        }
                        uint32_t pair_bits2 = 0;
                        if constexpr (DROP == 1)
                            pair_bits2 = hash_drop(drop_key, (uint32_t)(((m0 + rl4) >> 1) + 1) * (uint32_t)g.N + (uint32_t)col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rl = rl4 + e;
                            const int row = m0 + rl;
                            float v = acc[i][j][4 * q + e];
                            if constexpr (EPI == E16_HIDDEN_TRAIN) {
                                v += bias;
                                v = v > 0.f ? v : kLeakySlopeF * v;
                                if constexpr (DROP != 0) {
                                    bool keep = row < g.m_real && col_ok;
                                    if constexpr (DROP == 2) {
                                        if (keep) keep = g.drop_mask[(int64_t)row * g.ld_mask + col] != 0;
                                    } else {
                                        const uint32_t pb = e < 2 ? pair_bits : pair_bits2;
                                        const uint32_t u16 = (e & 1) ? (pb >> 16) : (pb & 0xFFFFu);
                                        keep = keep && (u16 >= (g.drop_thresh >> 16));
                                    }
                                    v = keep ? v * g.drop_scale : 0.f;
                                }
                            } else if constexpr (EPI == E16_HIDDEN_EVAL) {
                                v += bias;
                                v = v > 0.f ? v : kLeakySlopeF * v;
                                v = v * sc + sh;
                            }
                            const bf16_t b = col_ok ? f2bf(v) : (bf16_t)0;
                            if constexpr (EPI == E16_HIDDEN_TRAIN) {
                                // BatchNorm batch sums from the fp32 value, before it is rounded for storage (the rounding
                                // error of an activation is unbiased and 2^-9 relative: the statistics are those of the
                                // exact activations to first order, and one conversion per element is saved)
                                if (row < g.m_real && col_ok) {
                                    s1[j] += v;
                                    s2[j] += v * v;
                                }
                            }
                            hb[e] = b;
                            ct[rl * CP + cl] = b;
                        }
                        if constexpr (EPI == E16_HIDDEN_TRAIN) {
                            if (want_t) {   // uniform: transposed image, 4 consecutive rows of this lane's column = 8 bytes
                                uint2 w;
                                w.x = (uint32_t)hb[0] | ((uint32_t)hb[1] << 16);
                                w.y = (uint32_t)hb[2] | ((uint32_t)hb[3] << 16);
                                *reinterpret_cast<uint2*>(ctT + cl * CPT + rl4) = w;
                            }
                        }
                    }
                }
            }
        };
        if constexpr (EPI == E16_HIDDEN_TRAIN) {
            if (g.drop_mask) phase1(std::integral_constant<int, 2>{});
            else if (g.drop_scale != 1.0f) phase1(std::integral_constant<int, 1>{});
            else phase1(std::integral_constant<int, 0>{});
        } else {
            phase1(std::integral_constant<int, 0>{});
        }
        if constexpr (EPI == E16_HIDDEN_TRAIN) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                s1[j] += __shfl_xor(s1[j], 32);
                s2[j] += __shfl_xor(s2[j], 32);
                if (lane < 32) {
                    const int cl = (wn * TN + j) * 32 + lane;
                    red[(0 * WM + wm) * BN + cl] = s1[j];
                    red[(1 * WM + wm) * BN + cl] = s2[j];
                }
            }
        }
        __syncthreads();
        gemm16_stamp(g, 3);
        // row-major pass over the image: 16-byte chunks (8 columns), CPR chunks per row
        constexpr int CPR = BN / 8;
        constexpr int RPP = NT / CPR;         // rows covered by one pass of the workgroup
        static_assert(NT % CPR == 0 && BM % RPP == 0, "row-major pass layout");
        const int cc = tid % CPR, r0 = tid / CPR;
        const int col8 = n0 + 8 * cc;
        if constexpr (EPI == E16_STORE_BNRED) {
            // per-thread BatchNorm coefficients of its 8 columns (layer below), then the two batch sums
            float mean8[8], istd8[8], t1[8], t2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sc, sh;
                mean8[e] = 0.f; istd8[e] = 0.f; t1[e] = 0.f; t2[e] = 0.f;
                if (col8 + e < g.N) bn_column(g.bnC, col8 + e, mean8[e], istd8[e], sc, sh);
            }
            // all Hbelow loads of the thread in flight before the first use
            uint4 hv[BM / RPP], dv[BM / RPP];
#pragma unroll
            for (int p = 0; p < BM / RPP; ++p) {
                const int rl = r0 + RPP * p, row = m0 + rl;
                hv[p] = make_uint4(0, 0, 0, 0);
                if (row < g.M && col8 < g.N) hv[p] = *reinterpret_cast<const uint4*>(g.Hbelow + (int64_t)row * g.ldh + col8);
                dv[p] = *reinterpret_cast<const uint4*>(ct + rl * CP + 8 * cc);
            }
#pragma unroll
            for (int p = 0; p < BM / RPP; ++p) {
                const int rl = r0 + RPP * p, row = m0 + rl;
                if (row < g.M && col8 < g.N) {
                    *reinterpret_cast<uint4*>(g.C16 + (int64_t)row * g.ldc16 + col8) = dv[p];
                    if (row < g.m_real) {
                        const uint32_t dw[4] = {dv[p].x, dv[p].y, dv[p].z, dv[p].w};
                        const uint32_t hw[4] = {hv[p].x, hv[p].y, hv[p].z, hv[p].w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float d = __uint_as_float((e & 1) ? (dw[e >> 1] & 0xFFFF0000u) : (dw[e >> 1] << 16));
                            const float hh = __uint_as_float((e & 1) ? (hw[e >> 1] & 0xFFFF0000u) : (hw[e >> 1] << 16));
                            t1[e] += d;
                            t2[e] += d * ((hh - mean8[e]) * istd8[e]);
                        }
                    }
                }
            }
            // combine the RPP row groups that share a chunk: red[2][RPP][BN]
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(0 * RPP + r0) * BN + 8 * cc + e] = t1[e];
                red[(1 * RPP + r0) * BN + 8 * cc + e] = t2[e];
            }
            __syncthreads();
            for (int t = tid; t < 2 * BN; t += NT) {
                const int stat = t / BN, cb = t % BN;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < RPP; ++w) s += red[(stat * RPP + w) * BN + cb];
                if (n0 + cb < g.N) atomicAdd(&g.bstat_out[(int64_t)stat * g.N + n0 + cb], (double)s);
            }
        } else {
#pragma unroll
            for (int p = 0; p < BM / RPP; ++p) {
                const int rl = r0 + RPP * p, row = m0 + rl;
                if (row < g.M && col8 < g.N && !(g.dbg & 4))
                    *reinterpret_cast<uint4*>(g.C16 + (int64_t)row * g.ldc16 + col8) =
                        *reinterpret_cast<const uint4*>(ct + rl * CP + 8 * cc);
            }
            if constexpr (EPI == E16_HIDDEN_TRAIN) {
                // transposed copy: 16-byte chunks of 8 rows, consecutive threads along the rows of one column
                if (want_t) {
                    constexpr int CPC = BM / 8;   // chunks per column
                    for (int idx = tid; idx < BN * CPC; idx += NT) {
                        const int c = idx / CPC, mc = idx % CPC;
                        if (n0 + c < g.N && m0 + 8 * mc < g.M)
                            *reinterpret_cast<uint4*>(g.C16T + (int64_t)(n0 + c) * g.ldc16t + m0 + 8 * mc) =
                                *reinterpret_cast<const uint4*>(ctT + c * CPT + 8 * mc);
                    }
                }
                for (int t = tid; t < 2 * BN; t += NT) {
                    const int stat = t / BN, cb = t % BN;
                    float s = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) s += red[(stat * WM + w) * BN + cb];
                    if (n0 + cb < g.N && !(g.dbg & 1)) atomicAdd(&g.fstat_out[(int64_t)stat * g.N + n0 + cb], (double)s);
                }
            }
        }
        gemm16_stamp(g, 4);
    }
}

// dynamic LDS bytes of an instantiation: the operand buffers, or the output image + reduction scratch if larger
template <int BM, int BN, int WM, int WN, int EPI, int STG = 0>
constexpr size_t gemm16_smem_bytes() {
    size_t ops = (STG == 2 ? 3 : 2) * (size_t)(BM + BN) * 128;
    if (EPI == E16_SPLITK || EPI == E16_BIAS || EPI == E16_LATENT_MASK) return ops;
    const size_t img = (size_t)BM * (BN + 8) * 2 + (EPI == E16_HIDDEN_TRAIN ? (size_t)BN * (BM + 8) * 2 : 0);
    const size_t nt = (size_t)WM * WN * 64, rpp = nt / (BN / 8);
    const size_t red = EPI == E16_STORE_BNRED ? 2 * rpp * BN * 4 : 2 * (size_t)WM * BN * 4;
    // lean epilogue: transposed image [BN][BM + 8] + reduction scratch [2][16 * waves per column pass][BN]
    const size_t wg = nt / 64 / (BN / 32 > 0 ? BN / 32 : 1);
    const size_t lean = (size_t)BN * (BM + 8) * 2 + (EPI == E16_STORE_BNRED ? 2 * (wg > 0 ? wg : 1) * 16 * BN * 4 : 2 * (size_t)WM * BN * 4);
    size_t need = ops > img + red ? ops : img + red;
    return need > lean ? need : lean;
}

}  // namespace vh
