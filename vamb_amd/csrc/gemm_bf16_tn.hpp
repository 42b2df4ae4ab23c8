// gemm_bf16_tn.hpp -- weight-gradient GEMM of the bf16-storage step on ROW-major operands, gfx950.
//
//   C[z][m][n] = sum_{k in split z} A[k][m] * B[k][n]        A: [K][lda] bf16, B: [K][ldb] bf16  ("TN": both operands
//                                                            are contracted along their SLOW dimension, the batch)
//
// dW = dZ^T In of every Linear layer (vamb/encode.py:226-249 backward) has exactly this shape: dZ [batch][out] and
// In [batch][in] are the row-major tensors the forward / backward chain already holds.  gemm_bf16.hpp can only contract
// K-contiguous operands, so until round 3 every producer also wrote a transposed bf16 copy ([out][batch], [in][batch]:
// 8.4 MB per tensor and step at C2, four extra transpose launches).  Here the tile stays row-major in LDS and the MFMA
// operand -- 8 consecutive k per lane -- is assembled by the transposing LDS read of CDNA4:
//   * staging: global_load_lds_dwordx4, one wave instruction = 1 KiB = (64 / SPR) tile rows of SPR 16-byte slots
//     (SPR = tile width / 8).  The LDS image is the row-major [64 k][BM] tile with the 64-byte chunks of a row XOR-permuted
//     on the SOURCE side (chunk ^= f(k)), read back with the same involution.
//   * fragments: ds_read_b64_tr_b16.  Sixteen lanes address a [4 k][16 m] block (lane s: row s / 4, columns 4 (s % 4) .. + 3,
//     8 contiguous bytes); the instruction hands lane c the column c of that block: 4 consecutive k.  Two reads (k + 0..3,
//     k + 4..7) give the 8 k of a v_mfma_f32_32x32x16_bf16 operand.  The 32 lanes an LDS cycle services touch 4 rows x 64
//     contiguous bytes; f(k) puts those four chunks into the four quarters of the 64 banks.
//   * everything else as gemm_bf16.hpp: two LDS buffers, BK = 64, fp32 accumulators, split-K slabs, XCD-aware tile order,
//     zero-filled pieces outside the matrix or the split.
//   * COLSUM: the workgroups of the first tile column also accumulate sum_k A[k][m] (the bias gradient of the layer) from
//     the A fragments they hold anyway and add it into an fp64 vector.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "gemm_bf16.hpp"

namespace vh {

struct Gemm16TnArgs {
    const bf16_t* A;       // [K][lda], columns m
    int64_t lda;
    const bf16_t* B;       // [K][ldb], columns n
    int64_t ldb;
    int M, N, K;           // M, N multiples of 8
    int k_per_split;       // multiple of 64 unless there is one split
    int k_real;            // rows k >= k_real do not enter the column sums (padding rows of the batch)
    int64_t slab_stride;
    const bf16_t* zeros;   // >= 16 bytes of zeros
    float* C32;            // [splits][M][ldc]
    int64_t ldc;
    double* colsum;        // [M] += sum_k A[k][m] (nullptr: not wanted)
    int xcd_remap;
};


// 64-byte-chunk permutation of a tile row (in 16-byte slot units), by row width
template <int SPR>
__device__ __forceinline__ int tn_swz(int k) {
    if constexpr (SPR >= 16) return 4 * (k & 3);          // 256-byte rows: rows k .. k + 3 would share all banks
    else if constexpr (SPR == 8) return 4 * ((k >> 1) & 1);   // 128-byte rows: rows k and k + 2 would
    else return 0;                                        // 64-byte rows: four rows are 256 contiguous bytes
}

// STG: 0 = two LDS buffers, the next tile requested in front of the current tile's MFMAs; 2 = three buffers, the pieces of
// tile t + 2 issued between the MFMA groups of tile t, counted vmcnt + raw barrier (see gemm_bf16.hpp)
// The workgroup program.  (bid, gx, gy, gz): this workgroup's linear index in ITS problem's (n-tile, m-tile, split) grid and that
// grid's extents -- the launch's own for gemm_bf16_tn_kernel, a slice of a one-dimensional launch for the grouped kernel below.
template <int BM, int BN, int WM, int WN, int COLSUM, int STG>
__device__ __forceinline__ void gemm_bf16_tn_body(const Gemm16TnArgs& g, int bid, int gx, int gy, int gz) {
    constexpr int NWAVE = WM * WN;
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int BK = 64;
    constexpr int SPA = BM / 8, SPB = BN / 8;               // 16-byte slots per tile row
    constexpr int RBA = BM * 2, RBB = BN * 2;               // bytes per tile row
    constexpr int A_BYTES = BK * RBA, B_BYTES = BK * RBB;   // one buffer of each operand
    constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024; // DMA pieces per K-tile
    constexpr int RA = PA / NWAVE, RB = PB / NWAVE;
    constexpr int NBUF = STG == 2 ? 3 : 2;
    static_assert(TM >= 1 && TN >= 1 && PA % NWAVE == 0 && PB % NWAVE == 0 && RA >= 1 && RB >= 1, "tile / wave layout");
    static_assert(SPA == 4 || SPA == 8 || SPA == 16, "tile height 32, 64 or 128");
    static_assert(SPB == 4 || SPB == 8 || SPB == 16, "tile width 32, 64 or 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];   // the kernel's only LDS object
    unsigned char* const As = smem16;                     // [NBUF][A_BYTES]
    unsigned char* const Bs = smem16 + NBUF * A_BYTES;    // [NBUF][B_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx, by, bz;
    {
        int t = bid;
        if (g.xcd_remap) {   // XCD x gets the x-th contiguous eighth of the (z, m, n)-ordered tile list (speed only)
            const int nwg = gx * gy * gz;
            const int xcd = bid & 7, local = bid >> 3;
            t = xcd * (nwg >> 3) + min(xcd, nwg & 7) + local;
        }
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

    // ---- DMA sources.  Piece p of a tile holds its rows p * (64 / SPR) ...; lane l: row l / SPR, LDS slot l % SPR, which
    // receives the global slot (l % SPR) ^ f(row).  a_row >= BK marks a slot outside the matrix.
    const bf16_t* a_src[RA];
    const bf16_t* b_src[RB];
    int a_row[RA], b_row[RB];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int row = (wave + NWAVE * r) * (64 / SPA) + lane / SPA;
        const int col8 = 8 * ((lane % SPA) ^ tn_swz<SPA>(row));
        const bool ok = m0 + col8 < g.M;
        a_row[r] = ok ? row : (1 << 30);
        a_src[r] = g.A + (int64_t)row * g.lda + (ok ? m0 + col8 : 0);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = (wave + NWAVE * r) * (64 / SPB) + lane / SPB;
        const int col8 = 8 * ((lane % SPB) ^ tn_swz<SPB>(row));
        const bool ok = n0 + col8 < g.N;
        b_row[r] = ok ? row : (1 << 30);
        b_src[r] = g.B + (int64_t)row * g.ldb + (ok ? n0 + col8 : 0);
    }
    auto stage = [&](unsigned char* abuf, unsigned char* bbuf, int k0) {
        const int room = kend - k0;   // tile rows >= room are past the end of this split
#pragma unroll
        for (int r = 0; r < RA; ++r)
            glds16(a_row[r] < room ? a_src[r] + (int64_t)k0 * g.lda : g.zeros, abuf + (wave + NWAVE * r) * 1024);
#pragma unroll
        for (int r = 0; r < RB; ++r)
            glds16(b_row[r] < room ? b_src[r] + (int64_t)k0 * g.ldb : g.zeros, bbuf + (wave + NWAVE * r) * 1024);
    };

    // ---- fragment addresses (see the header): lane = 16 grp + s; block column half grp & 1, k half grp >> 1
    const int grp = lane >> 4, s16 = lane & 15;
    const int krow = 8 * (grp >> 1) + (s16 >> 2);            // row inside a 16-deep k-step, first read (second: + 4)
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int col = (wm * TM + i) * 32 + 16 * (grp & 1) + 4 * (s16 & 3);
        a_off[i] = krow * RBA + 16 * ((col >> 3) ^ tn_swz<SPA>(krow)) + 2 * (col & 7);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + 16 * (grp & 1) + 4 * (s16 & 3);
        b_off[j] = krow * RBB + 16 * ((col >> 3) ^ tn_swz<SPB>(krow)) + 2 * (col & 7);
    }
    // f(k) depends on k & 3 only and the rows of one lane are krow + 4 r + 16 t: the permutation is the same for all of them
    // when SPR != 8; with 128-byte rows it flips with bit 1 of the row, i.e. never within (+4, +16) steps either.

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float csum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) csum[i] = 0.f;
    const bool do_colsum = COLSUM && g.colsum != nullptr && bx == 0 && wn == 0;

    auto piece = [&](unsigned char* abuf, unsigned char* bbuf, int k0, int q) {
        const int room = kend - k0;
        if (q < RA) glds16(a_row[q] < room ? a_src[q] + (int64_t)k0 * g.lda : g.zeros, abuf + (wave + NWAVE * q) * 1024);
        else glds16(b_row[q - RA] < room ? b_src[q - RA] + (int64_t)k0 * g.ldb : g.zeros, bbuf + (wave + NWAVE * (q - RA)) * 1024);
    };
    constexpr int NP = RA + RB;
    // MFMA groups of the tile at k0; with `prefetch` the pieces of the tile at knext are issued in between (STG == 2)
    auto compute = [&](const unsigned char* abuf, const unsigned char* bbuf, int k0, unsigned char* anext, unsigned char* bnext,
                       int knext, bool prefetch) {
        bf16x8 a8[BK / 16][TM], b8[BK / 16][TN];
        auto frags = [&](int t) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const tr_v4s lo = lds_tr16(abuf + a_off[i] + (16 * t) * RBA);
                const tr_v4s hi = lds_tr16(abuf + a_off[i] + (16 * t + 4) * RBA);
                const tr_v8s v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                a8[t][i] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const tr_v4s lo = lds_tr16(bbuf + b_off[j] + (16 * t) * RBB);
                const tr_v4s hi = lds_tr16(bbuf + b_off[j] + (16 * t + 4) * RBB);
                const tr_v8s v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                b8[t][j] = __builtin_bit_cast(bf16x8, v);
            }
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
            if constexpr (STG == 2) {
                if (prefetch) {   // workgroup-uniform
#pragma unroll
                    for (int q = (NP * t) / 4; q < (NP * (t + 1)) / 4; ++q) piece(anext, bnext, knext, q);
                }
            }
            if (t + 2 < BK / 16) frags(t + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (COLSUM != 0) {
            if (do_colsum) {   // wave-uniform
                // lane (column lane & 31 of block i, k half lane >> 5) holds k = k0 + 16 t + 8 (lane >> 5) + e
#pragma unroll
                for (int t = 0; t < BK / 16; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const tr_v8s v = __builtin_bit_cast(tr_v8s, a8[t][i]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int k = k0 + 16 * t + 8 * (lane >> 5) + e;
                            const float x = __uint_as_float((uint32_t)(unsigned short)v[e] << 16);
                            csum[i] += k < g.k_real ? x : 0.f;
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (STG == 0) {
        if (nk > 0) stage(As, Bs, kbeg);
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            stage(As + A_BYTES, Bs + B_BYTES, kbeg + (kt + 1) * BK);
            compute(As, Bs, kbeg + kt * BK, nullptr, nullptr, 0, false);
            __syncthreads();
            if (kt + 2 < nk) stage(As, Bs, kbeg + (kt + 2) * BK);
            compute(As + A_BYTES, Bs + B_BYTES, kbeg + (kt + 1) * BK, nullptr, nullptr, 0, false);
            __syncthreads();
        }
        if (kt < nk) {
            compute(As, Bs, kbeg + kt * BK, nullptr, nullptr, 0, false);
            __syncthreads();
        }
    } else {
        auto abuf = [&](int b) { return As + b * A_BYTES; };
        auto bbuf = [&](int b) { return Bs + b * B_BYTES; };
        if (nk > 0) stage(abuf(0), bbuf(0), kbeg);
        if (nk > 1) {
            stage(abuf(1), bbuf(1), kbeg + BK);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        auto iteration = [&](int kt, int cur, auto steady) {
            const bool pre = decltype(steady)::value ? true : kt + 2 < nk;
            const int nxt = cur == 0 ? 2 : cur - 1;   // (cur + 2) % 3
            compute(abuf(cur), bbuf(cur), kbeg + kt * BK, abuf(nxt), bbuf(nxt), kbeg + (kt + 2) * BK, pre);
            if (pre) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 4 < nk; kt += 3) {
            iteration(kt, 0, std::true_type{});
            iteration(kt + 1, 1, std::true_type{});
            iteration(kt + 2, 2, std::true_type{});
        }
        if (kt < nk) iteration(kt, 0, std::false_type{});
        if (kt + 1 < nk) iteration(kt + 1, 1, std::false_type{});
        if (kt + 2 < nk) iteration(kt + 2, 2, std::false_type{});
        if (kt + 3 < nk) iteration(kt + 3, 0, std::false_type{});
    }

    // ---- epilogue: acc[i][j][reg] is C[m][n], m = m0 + (wm TM + i) 32 + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5),
    //      n = n0 + (wn TN + j) 32 + (lane & 31)
    float* Cout = g.C32 + (int64_t)bz * g.slab_stride;
    const int frag_r = lane & 31, frag_h = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + frag_r;
        if (col >= g.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * frag_h;
                if (row < g.M) Cout[(int64_t)row * g.ldc + col] = acc[i][j][reg];
            }
    }
    if constexpr (COLSUM != 0) {
        if (do_colsum) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float tot = csum[i] + __shfl_xor(csum[i], 32);
                const int m = m0 + (wm * TM + i) * 32 + frag_r;
                if (frag_h == 0 && m < g.M) atomicAdd(&g.colsum[m], (double)tot);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int COLSUM, int STG = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_tn_kernel(const Gemm16TnArgs g) {
    const int gx = gridDim.x, gy = gridDim.y;
    gemm_bf16_tn_body<BM, BN, WM, WN, COLSUM, STG>(g, (int)(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z)), gx, gy, (int)gridDim.z);
}

// Two weight-gradient products in ONE launch (round 6): the last two of a training step -- encoder layer 0's and layer 1's -- wait
// for the same kernel and end the backward pass one after the other, 14 + 15 us of a ~260 us step at C2, each alone on a chip it
// fills only once (192 / 256 workgroups of 48 KB of LDS: three fit a CU).  As one launch of nwg0 + nwg1 workgroups they share the
// CUs and one dispatch ramp / release.  Every workgroup runs the unchanged program on its own problem: same tiles, same slabs,
// same bits.
struct Gemm16TnPair {
    Gemm16TnArgs p[2];
    int gx[2], gy[2], gz[2];
    int nwg0;              // workgroups of problem 0 (a multiple of 8 keeps the XCD remap of problem 1 aligned; speed only)
};
template <int BM, int BN, int WM, int WN, int STG = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_tn_pair_kernel(const Gemm16TnPair a) {
    const int b = (int)blockIdx.x;
    if (b < a.nwg0) gemm_bf16_tn_body<BM, BN, WM, WN, 0, STG>(a.p[0], b, a.gx[0], a.gy[0], a.gz[0]);
    else gemm_bf16_tn_body<BM, BN, WM, WN, 0, STG>(a.p[1], b - a.nwg0, a.gx[1], a.gy[1], a.gz[1]);
}

template <int BM, int BN, int STG = 0>
constexpr size_t gemm16_tn_smem_bytes() {
    return (STG == 2 ? 3 : 2) * (size_t)64 * (BM + BN) * 2;
}

}  // namespace vh
