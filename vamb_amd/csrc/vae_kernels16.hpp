// vae_kernels16.hpp -- the non-GEMM kernels of the bf16-storage training step (BASELINE configs C2-C4), gfx950.
//
// In this mode every activation / gradient tensor that feeds a contraction lives in HBM as bfloat16, row-major
// [bs_p][n_p], plus -- where the weight-gradient GEMM contracts over the batch -- a transposed copy [n_p][bs_p],
// so that all GEMMs are K-contiguous "NT" products staged by LDS-DMA (gemm_bf16.hpp).  Master parameters, the
// optimiser moments, BatchNorm statistics (fp64 sums), the loss and the reconstruction stay fp32.
//
// BatchNorm between two Linear layers is folded into the consumer's weights once per step:
//     BN(h) W^T + b = h (W diag(s))^T + (b + W t),   s = gamma / sqrt(var + eps),  t = beta - mean s
// (vae_fold_bn_kernel), so the raw activations stream into the next GEMM untouched; the weight gradient of such
// a layer is recovered from the raw product G = dZ^T h as dW = G diag(s) + dbias t^T inside the optimiser kernel.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gemm_bf16.hpp"
#include "vae_kernels.hpp"

namespace vh {

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

// ---- batch assembly: Xb (fp32, the loss targets), Xb16 (the first GEMM's operand), Wb ----------------------------
// blockDim (64, 4): one wavefront per row, float4 per lane.
// LABELS = false is the plain VAE's kernel as it was before the label block existed (the label-aware instantiation measured
// 21 us against 8 us at C2 although the extra work is a handful of selects per lane: kept apart).
template <bool LABELS>
__global__ void vae_gather16_kernel(const float* __restrict__ X, int64_t ld_src, int64_t ldx, const float* __restrict__ w_all,
                                    const int64_t* __restrict__ idx, const ShuffleSpec shuffle,
                                    const long long* __restrict__ batch_ptr, int64_t base, int bs, int bs_p,
                                    float* __restrict__ Xb, bf16_t* __restrict__ Xb16, float* __restrict__ Wb,
                                    const LabelSrc lab, int32_t* __restrict__ Lb, long long* __restrict__ Rb) {
    // Rb (plain VAE, vae.loss_from_dataset): the batch's dataset rows, for the loss kernel to read its targets from -- the fp32
    // copy of the batch (Xb == nullptr then) is a third of this kernel's traffic and has no other reader in the bf16 step
    const int r = blockIdx.x * 4 + threadIdx.y;
    if (r >= bs_p) return;
    const bool real = r < bs;
    const int64_t first = base + (batch_ptr ? (int64_t)(*batch_ptr) * bs : 0);
    int64_t src = 0;
    if (real) src = idx ? idx[first + r] : (int64_t)shuffle_index(shuffle, (unsigned long long)(first + r));
    if constexpr (!LABELS) {
        const float4* s = reinterpret_cast<const float4*>(X + src * ldx);
        float4* d = reinterpret_cast<float4*>(Xb + (int64_t)r * ldx);
        uint2* d16 = reinterpret_cast<uint2*>(Xb16 + (int64_t)r * ldx);
        const int dq = (int)(ldx / 4);
        for (int c = threadIdx.x; c < dq; c += 64) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (real) v = s[c];
            if (Xb) d[c] = v;
            d16[c] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
        }
        if (threadIdx.x == 0) {
            Wb[r] = real ? w_all[src] : 0.f;
            if (Rb) Rb[r] = (long long)src;
        }
    } else {
        const float4* s = reinterpret_cast<const float4*>(X + src * ld_src);
        float4* d = Xb ? reinterpret_cast<float4*>(Xb + (int64_t)r * ldx) : nullptr;   // (encode pass: only the bf16 operand)
        uint2* d16 = reinterpret_cast<uint2*>(Xb16 + (int64_t)r * ldx);
        const int dq = (int)(ldx / 4), sq = (int)(ld_src / 4);
        int hot = -1;   // one-hot label column (see vae_gather_kernel)
        if (lab.labels && real) hot = lab.col0 + lab.labels[src];
        for (int c = threadIdx.x; c < dq; c += 64) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (real && c < sq) v = s[c];
            if ((hot >> 2) == c && hot >= 0) {   // (selects, not an indexed write: the compiler moves an indexed float4 into LDS)
                const int e = hot & 3;
                v.x = e == 0 ? 1.0f : v.x; v.y = e == 1 ? 1.0f : v.y; v.z = e == 2 ? 1.0f : v.z; v.w = e == 3 ? 1.0f : v.w;
            }
            if (d) d[c] = v;
            d16[c] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
        }
        if (threadIdx.x == 0) {
            if (Wb) Wb[r] = real ? w_all[src] : 0.f;
            if (Lb) Lb[r] = hot >= 0 ? hot - lab.col0 : 0;
        }
    }
}

// rows of an fp32 matrix -> bf16 (encode pass: the resident feature matrix is fp32)
__global__ void vae_cast16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
}

// ---- BatchNorm folded into the consumer's weights ------------------------------------------------------------------
// W [n_rows][ldw] fp32 (master), stats of the producing layer over its K columns -> W16 = bf16(W diag(s)) [n_rows][ldw],
// bias_out[n] = bias[n] + sum_k W[n][k] t_k.  One wavefront per weight row; s, t are rebuilt per workgroup.
// from_running != 0: eval mode, s / t from the running statistics (scale_in / shift_in precomputed by
// vae_bn_eval_coeff_kernel).
constexpr int kFoldRowsPerWave = 1;
constexpr int kFoldRegQ = 5;   // float4 registers per lane holding the weight row (covers K <= 1280; longer rows loop)
__global__ __launch_bounds__(256) void vae_fold_bn_kernel(const float* __restrict__ W, int64_t ldw, int n_rows, int K,
                                                          const float* __restrict__ bias, const BnSrc bn,
                                                          const float* __restrict__ scale_in,
                                                          const float* __restrict__ shift_in,
                                                          bf16_t* __restrict__ W16, float* __restrict__ bias_out,
                                                          float* __restrict__ scale_out, float* __restrict__ shift_out,
                                                          float* __restrict__ mean_out, float* __restrict__ istd_out) {
    extern __shared__ __attribute__((aligned(16))) float fold_st[];   // [2][K]
    float* s_s = fold_st;
    float* t_s = fold_st + K;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    // this wave's weight row goes into registers first: the loads fly while s, t are rebuilt from the fp64 batch sums
    const float* w = W + (int64_t)(n < n_rows ? n : 0) * ldw;
    float4 wr[kFoldRegQ];
#pragma unroll
    for (int q = 0; q < kFoldRegQ; ++q) {
        const int k = 4 * lane + 256 * q;
        wr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) wr[q] = *reinterpret_cast<const float4*>(w + k);
    }
    for (int k = threadIdx.x; k < K; k += 256) {
        float sc, sh;
        if (scale_in) {
            sc = scale_in[k];
            sh = shift_in[k];
        } else {
            float mean, istd;
            bn_column(bn, k, mean, istd, sc, sh);
            // the backward kernels of this step (input-gradient GEMM epilogue, elementwise BatchNorm backward) read these
            // instead of re-deriving them from the fp64 sums
            if (blockIdx.x == 0 && mean_out != nullptr) { mean_out[k] = mean; istd_out[k] = istd; }
        }
        s_s[k] = sc;
        t_s[k] = sh;
        // the coefficients this step's forward pass uses, kept for the optimiser (dW = G diag(s) + dbias t^T must be completed
        // with THESE s, t: re-deriving them from gamma / beta inside the kernel that updates gamma / beta is a race)
        if (blockIdx.x == 0 && scale_out != nullptr) { scale_out[k] = sc; shift_out[k] = sh; }
    }
    __syncthreads();
    if (n >= n_rows) return;
    bf16_t* o = W16 + (int64_t)n * ldw;
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < kFoldRegQ; ++q) {
        const int k = 4 * lane + 256 * q;
        if (k < K) {
            const float4 v = wr[q];
            const float4 s4 = *reinterpret_cast<const float4*>(s_s + k);
            const float4 t4 = *reinterpret_cast<const float4*>(t_s + k);
            dot += v.x * t4.x + v.y * t4.y + v.z * t4.z + v.w * t4.w;
            *reinterpret_cast<uint2*>(o + k) = make_uint2(pack_bf2(v.x * s4.x, v.y * s4.y), pack_bf2(v.z * s4.z, v.w * s4.w));
        }
    }
    for (int k = 4 * lane + 256 * kFoldRegQ; k < K; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(w + k);
        const float4 s4 = *reinterpret_cast<const float4*>(s_s + k);
        const float4 t4 = *reinterpret_cast<const float4*>(t_s + k);
        dot += v.x * t4.x + v.y * t4.y + v.z * t4.z + v.w * t4.w;
        *reinterpret_cast<uint2*>(o + k) = make_uint2(pack_bf2(v.x * s4.x, v.y * s4.y), pack_bf2(v.z * s4.z, v.w * s4.w));
    }
    dot = wave_sum(dot);
    if (lane == 0) bias_out[n] = bias[n] + dot;
}

// ---- bf16 shadows of the weights (init / set_param; during training the optimiser writes them itself) --------------
// W16[r][c] = bf16(P[r][c]);  W16T[c][r] = the same, for r < rows_p, c < cols_p.
__global__ void vae_shadow_kernel(const float* __restrict__ P, int rows_p, int cols_p, bf16_t* __restrict__ W16,
                                  bf16_t* __restrict__ W16T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows_p * cols_p) return;
    const int r = (int)(i / cols_p), c = (int)(i % cols_p);
    const bf16_t b = f2bf(P[i]);
    W16[i] = b;
    if (W16T) W16T[(int64_t)c * rows_p + r] = b;
}

// ---- reparameterisation: MU (fp32) = slabs + bias; Z16 = bf16(MU + eps) on real rows / columns ---------------------
__global__ void vae_reparam16_kernel(const float* __restrict__ slabs, int nslab, int64_t stride,
                                     const float* __restrict__ bias, const float* __restrict__ E, uint64_t key,
                                     const unsigned long long* __restrict__ step_ptr, int noise,
                                     float* __restrict__ MU, bf16_t* __restrict__ Z16, int bs, int L, int L_p, int bs_p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)bs_p * L_p) return;
    const int r = (int)(i / L_p), c = (int)(i % L_p);
    float m = bias[c];
    {
        constexpr int kMaxSlabs = 8;
        float v[kMaxSlabs];
#pragma unroll
        for (int s = 0; s < kMaxSlabs; ++s) v[s] = s < nslab ? slabs[(int64_t)s * stride + i] : 0.f;
#pragma unroll
        for (int s = 0; s < kMaxSlabs; ++s)
            if (s < nslab) m += v[s];
        for (int s = kMaxSlabs; s < nslab; ++s) m += slabs[(int64_t)s * stride + i];
    }
    MU[i] = m;
    float z = 0.f;
    if (r < bs && c < L) {
        float e = 0.f;
        if (E) e = E[i];
        else if (noise) e = hash_randn(step_key(key, step_ptr), (uint64_t)i);
        z = m + e;
    }
    Z16[i] = f2bf(z);
}

// ---- loss + backward seed, bf16 gradient of the reconstruction ------------------------------------------------------
// Same arithmetic as vae_loss_kernel (encode.py:316-357); dR leaves as bf16 (the operand of the two GEMMs that
// consume it), the KLD part of dL/dmu stays fp32.
struct Loss16Args {
    const float* R;
    const float* X;            // targets: the fp32 batch [bs_p][ld], or (rows != nullptr) the dataset, row rows[r] of it for batch row r
    const long long* rows;
    int64_t ld;
    const float* MU;
    int64_t ldl;
    float inv_b2;
    int bs, bs_p, S, L;
    float ce_w, ab_w, sse_w, kld_w;
    bf16_t* dR16;
    float* dMUk;
    float* part;
    int NL, lab0, ntnf, nab;   // label block, see LossArgs
    const int32_t* Lb;
    float* lab_part;
};

// 8 waves per SIMD: the compiler's own choice was 95 VGPRs (5 waves per SIMD, 1280 of the 2048 workgroups of a batch of 8192
// resident at once); with 50 VGPRs every row's wavefront is resident from the start: 14.4 -> 12.8 us (profiles/r03zf_*)
// DPP: the ten per-row reductions on the VALU's data-parallel primitives (wave_reduce_dpp) instead of ds_bpermute butterflies
// (option vae.loss_dpp; the sums are then formed in another order: same value up to float32 rounding)
template <bool DPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void vae_loss16_kernel(const Loss16Args a) {
    auto rsum = [](float v) { return DPP ? wave_sum_dpp(v) : wave_sum(v); };
    auto rmax = [](float v) { return DPP ? wave_max_dpp(v) : wave_max(v); };
    // dynamic LDS: per wave the reconstruction row and the target row, read from HBM once with 16-byte loads (the
    // softmax / CE / SSE passes below re-read them four times)
    extern __shared__ __attribute__((aligned(16))) float loss_rows[];   // [4 waves][2][ld]
    __shared__ float red[4][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    float ab_t = 0.f, ce_t = 0.f, sse_t = 0.f, kld_t = 0.f, cel_t = 0.f, hit_t = 0.f;
    if (row < a.bs_p) {
        bf16_t* dr = a.dR16 + (int64_t)row * a.ld;
        float* dm = a.dMUk + (int64_t)row * a.ldl;
        if (row >= a.bs) {
            for (int c = lane; c < a.ld; c += 64) dr[c] = 0;
            for (int c = lane; c < a.ldl; c += 64) dm[c] = 0.f;
        } else {
            float* r = loss_rows + (size_t)wave * 2 * a.ld;
            float* x = r + a.ld;
            {
                const float4* rg = reinterpret_cast<const float4*>(a.R + (int64_t)row * a.ld);
                const float4* xg = reinterpret_cast<const float4*>(a.X + (a.rows ? (int64_t)a.rows[row] : (int64_t)row) * a.ld);
                for (int c = lane; c < (int)(a.ld / 4); c += 64) {
                    reinterpret_cast<float4*>(r)[c] = rg[c];
                    reinterpret_cast<float4*>(x)[c] = xg[c];
                }
            }
            const float g = a.inv_b2;
            const int S = a.S;
            // softmax / cross-entropy over the S abundance columns: one exponential and one logarithm per element (the
            // staged rows are overwritten with exp(r - max) and x / (p + 1e-9)); hardware exp2 / log2 based intrinsics --
            // the kernel was bound by libm-accurate transcendentals (3 expf + logf per element: 17 us at C2)
            float mx = -3.0e38f;
            for (int c = lane; c < S; c += 64) mx = fmaxf(mx, r[c]);
            mx = rmax(mx);
            float se = 0.f;
            for (int c = lane; c < S; c += 64) {
                const float e = __expf(r[c] - mx);
                r[c] = e;
                se += e;
            }
            se = rsum(se);
            const float inv = S > 0 ? 1.0f / se : 0.0f;
            float ce = 0.f, pdp = 0.f;
            for (int c = lane; c < S; c += 64) {
                const float p = r[c] * inv;
                const float q = p + 1e-9f;
                const float xv = x[c];
                const float t = __fdividef(xv, q);
                ce -= __logf(q) * xv;
                pdp -= p * t;
                r[c] = p;
                x[c] = t;
            }
            ce = rsum(ce);
            pdp = rsum(pdp);
            const float gce = g * a.ce_w;
            for (int c = lane; c < S; c += 64) dr[c] = f2bf(gce * r[c] * (-x[c] - pdp));
            float sse = 0.f;
            const float gsse = g * a.sse_w * 2.0f;
            for (int c = S + lane; c < S + a.ntnf; c += 64) {
                const float diff = r[c] - x[c];
                sse += diff * diff;
                dr[c] = f2bf(gsse * diff);
            }
            sse = rsum(sse);
            float ab = 0.f;
            if (lane == 0 && a.nab) {
                const int c = S + a.ntnf;
                const float diff = r[c] - x[c];
                ab = diff * diff;
                dr[c] = f2bf(g * a.ab_w * 2.0f * diff);
            }
            ab = rsum(ab);
            if (a.NL > 0)
                label_block(r + a.lab0, a.NL, a.Lb[row], g, [&](int c, float v) { dr[a.lab0 + c] = f2bf(v); }, cel_t, hit_t);
            for (int c = S + a.ntnf + a.nab + a.NL + lane; c < a.ld; c += 64) dr[c] = 0;
            const float* mu = a.MU + (int64_t)row * a.ldl;
            float kld = 0.f;
            const float gk = g * a.kld_w;
            for (int c = lane; c < a.ldl; c += 64) {
                const float m = c < a.L ? mu[c] : 0.f;
                kld += m * m;
                dm[c] = gk * m;
            }
            kld = 0.5f * rsum(kld);
            ab_t = ab * a.ab_w;
            ce_t = ce * a.ce_w;
            sse_t = sse * a.sse_w;
            kld_t = kld * a.kld_w;
        }
    }
    if (lane == 0) {
        red[wave][0] = ab_t; red[wave][1] = ce_t; red[wave][2] = sse_t; red[wave][3] = kld_t;
        red[wave][4] = cel_t; red[wave][5] = hit_t;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int t = threadIdx.x;
        a.part[(int64_t)blockIdx.x * 4 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    } else if (threadIdx.x < 6 && a.lab_part) {
        const int t = threadIdx.x;
        a.lab_part[(int64_t)blockIdx.x * 2 + (t - 4)] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    }
}

// The same loss with the two rows of a wavefront held in REGISTERS (round 6; plain VAE, no label block): lane l owns the columns
// l + 64 i, i < NV, of the reconstruction row and of the target row -- NV values each instead of a staging area in LDS that the
// softmax / cross-entropy / SSE passes re-read through five rolled loops (one ds_read + lgkmcnt(0) + exec-mask bookkeeping per
// iteration).  Every per-lane sum runs over ascending columns l, l + 64, ... like the loops of vae_loss16_kernel and every
// reduction is the same DPP tree, so abundance cross-entropy, its gradient, the abundance-total term, KLD and every element of dR
// carry the same bits; only the TNF sum of squares is grouped by column mod 64 instead of (column - S) mod 64 (the reported
// `sse` differs in the last float digit, its gradient does not).
template <int NV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NV <= 8 ? 8 : 4, 8))) void vae_loss16_reg_kernel(const Loss16Args a) {
    __shared__ float red[4][4];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;   // wave-uniform
    float ab_t = 0.f, ce_t = 0.f, sse_t = 0.f, kld_t = 0.f;
    if (row < a.bs_p) {
        bf16_t* dr = a.dR16 + (int64_t)row * a.ld;
        float* dm = a.dMUk + (int64_t)row * a.ldl;
        const int ld = (int)a.ld;
        if (row >= a.bs) {
            for (int c = lane; c < ld; c += 64) dr[c] = 0;
            for (int c = lane; c < a.ldl; c += 64) dm[c] = 0.f;
        } else {
            const float* rg = a.R + (int64_t)row * a.ld;
            const float* xg = a.X + (a.rows ? (int64_t)a.rows[row] : (int64_t)row) * a.ld;
            float r[NV], x[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                r[i] = 0.f; x[i] = 0.f;
                if (c < ld) { r[i] = rg[c]; x[i] = xg[c]; }
            }
            const float g = a.inv_b2;
            const int S = a.S, T0 = a.S + a.ntnf;   // [0, S) abundances, [S, T0) TNF, T0 the abundance total (if nab)
            float mx = -3.0e38f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lane + 64 * i < S) mx = fmaxf(mx, r[i]);
            mx = wave_max_dpp(mx);
            float se = 0.f;
            float e_[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                e_[i] = 0.f;
                if (lane + 64 * i < S) {
                    e_[i] = __expf(r[i] - mx);
                    se += e_[i];
                }
            }
            se = wave_sum_dpp(se);
            const float inv = S > 0 ? 1.0f / se : 0.0f;
            float ce = 0.f, pdp = 0.f;
            float t_[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                t_[i] = 0.f;
                if (lane + 64 * i < S) {
                    const float p = e_[i] * inv;
                    const float q = p + 1e-9f;
                    const float xv = x[i];
                    const float t = __fdividef(xv, q);
                    ce -= __logf(q) * xv;
                    pdp -= p * t;
                    e_[i] = p;
                    t_[i] = t;
                }
            }
            ce = wave_sum_dpp(ce);
            pdp = wave_sum_dpp(pdp);
            const float gce = g * a.ce_w;
            const float gsse = g * a.sse_w * 2.0f;
            const float gab = g * a.ab_w * 2.0f;
            float sse = 0.f, ab = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < ld) {
                    float out = 0.f;
                    if (c < S) {
                        out = gce * e_[i] * (-t_[i] - pdp);
                    } else if (c < T0) {
                        const float diff = r[i] - x[i];
                        sse += diff * diff;
                        out = gsse * diff;
                    } else if (c == T0 && a.nab) {
                        const float diff = r[i] - x[i];
                        ab = diff * diff;
                        out = gab * diff;
                    }
                    dr[c] = f2bf(out);
                }
            }
            sse = wave_sum_dpp(sse);
            ab = wave_sum_dpp(ab);
            const float* mu = a.MU + (int64_t)row * a.ldl;
            float kld = 0.f;
            const float gk = g * a.kld_w;
            for (int c = lane; c < a.ldl; c += 64) {
                const float m = c < a.L ? mu[c] : 0.f;
                kld += m * m;
                dm[c] = gk * m;
            }
            kld = 0.5f * wave_sum_dpp(kld);
            ab_t = ab * a.ab_w;
            ce_t = ce * a.ce_w;
            sse_t = sse * a.sse_w;
            kld_t = kld * a.kld_w;
        }
    }
    if (lane == 0) { red[wave][0] = ab_t; red[wave][1] = ce_t; red[wave][2] = sse_t; red[wave][3] = kld_t; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int t = threadIdx.x;
        a.part[(int64_t)blockIdx.x * 4 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    }
}

// ---- elementwise backward of one hidden layer, bf16 in / out ---------------------------------------------------------
//   dZ = keep * slope(h) * drop_scale * istd*gamma * (dA - S1/B - xhat * S2/B)      (see vae_dz_kernel)
// reads dA16, H16 [bs_p][n_p]; writes dZ16 [bs_p][n_p] and accumulates the fp64 column sums of dZ (bias gradient).  A
// workgroup owns kDz16Rows rows x kDz16Cols columns.
struct Dz16Args {
    const bf16_t* DA;
    const bf16_t* H;
    bf16_t* DZ;
    int n_p, bs, bs_p;
    BnSrc bn;
    const float* mean;    // mean / 1/std of this layer's BatchNorm as the forward fold left them
    const float* istd;
    const double* bstat;
    float drop_scale;
    const uint8_t* drop_mask;
    int64_t ld_mask;
    double* dbias;
};
// (Two register-transposing variants without LDS -- a thread owning an 8 x 8 block, 128 x 128 tiles with 4 waves or
// 64 x 64 tiles with one wave -- measured 18 and 37 us against 13.7 us for this kernel at 8192 x 512: profiles/README.md.)
// Tile of a workgroup, measured as step time at C2 / the C3 shape on one box (round 6, profiles/r06o_step_dz_tile_*.txt,
// r06p_step_dz_tile_*.txt; us per step):   128 x 64 (rounds 3-5) 245.8 / 343.1    64 x 128  241.6 / 339.8    64 x 64  245.6
//   128 x 32  254.6 / 342.8    256 x 32  256.7 / 346.8    256 x 64  259.6 / 343.0    128 x 128  257.4    512 x 32  271.0 / 354.7
//   512 x 16  283.9    32 x 128  249.5 / 345.6    32 x 256  249.3 / 346.7    64 x 256  256.8 / 345.2    16 x 256  270.3 / 367.0
// One 128-byte line of each tensor per row and workgroup, and the taller the tile the fewer fp64 atomics meet on a column of
// the bias gradient (8192 rows: 64 per column instead of 128) -- until the grid falls under two workgroups per CU.
constexpr int kDz16Cols = 64;
constexpr int kDz16Rows = 128;

// COLS x ROWS elements per workgroup of 256 threads: a thread owns 8 consecutive columns, COLS / 8 threads share a row
// MASKED: injected keep-masks (parity tests); the production instantiation has no per-element control flow (round 6: with the mask
// test inside, every ELEMENT sat in its own exec-mask region -- s_and_saveexec / s_cbranch_execz / waits -- 32 of them per thread)
template <int COLS, int ROWS, bool MASKED>
__global__ __launch_bounds__(256) void vae_dz16_kernel(const Dz16Args a) {
    constexpr int TPR = COLS / 8;        // threads per row
    constexpr int RP = 256 / TPR;        // rows per pass of the workgroup
    constexpr int PASS = ROWS / RP;
    static_assert(COLS % 8 == 0 && 256 % TPR == 0 && ROWS % RP == 0 && PASS >= 1, "tile layout");
    __shared__ float red[RP][COLS];
    __shared__ float cf[3][COLS];
    const int tid = threadIdx.x;
    const int col0 = blockIdx.x * COLS, row0 = blockIdx.y * ROWS;
    const int c8 = (tid % TPR) * 8;         // this thread's 8 columns inside the tile
    const int rt = tid / TPR;               // row lane
    const int col = col0 + c8;
    // the thread's 16-byte loads go out first; the per-column coefficients (fp64 statistics) are formed underneath them
    uint4 da[PASS], hh[PASS];
#pragma unroll
    for (int p = 0; p < PASS; ++p) {
        const int r = row0 + rt + RP * p;
        da[p] = make_uint4(0, 0, 0, 0);
        hh[p] = da[p];
        if (r < a.bs && col < a.n_p) {
            const int64_t i = (int64_t)r * a.n_p + col;
            da[p] = *reinterpret_cast<const uint4*>(a.DA + i);
            hh[p] = *reinterpret_cast<const uint4*>(a.H + i);
        }
    }
    for (int t = tid; t < COLS; t += 256) {
        const int colc = col0 + t;
        float ca = 0.f, ch = 0.f, c0 = 0.f;
        if (colc < a.n_p) {   // (a.mean / a.istd: the floats the forward fold left -- the same bn_column forms)
            const DzCoefSrc src{a.mean, a.istd, a.bn.gamma, a.bstat, a.n_p, a.bn.bs, a.drop_scale};
            dz16_coeffs(src, colc, ca, ch, c0);
        }
        cf[0][t] = ca; cf[1][t] = ch; cf[2][t] = c0;
    }
    __syncthreads();
    const bool hashed_drop = (a.drop_scale != 1.0f) && !MASKED;
    float ca[8], ch[8], c0[8], s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = cf[0][c8 + e]; ch[e] = cf[1][c8 + e]; c0[e] = cf[2][c8 + e]; s[e] = 0.f; }
#pragma unroll
    for (int p = 0; p < PASS; ++p) {
        const int rl = rt + RP * p, r = row0 + rl;
        const uint32_t dw[4] = {da[p].x, da[p].y, da[p].z, da[p].w};
        const uint32_t hw[4] = {hh[p].x, hh[p].y, hh[p].z, hh[p].w};
        uint32_t ow[4] = {0, 0, 0, 0};
        if (r < a.bs && col < a.n_p) {   // (rows / columns outside the batch were loaded as zeros and stay zeros)
            [[maybe_unused]] uint64_t m8 = ~0ull;
            if constexpr (MASKED) m8 = *reinterpret_cast<const uint64_t*>(a.drop_mask + (int64_t)r * a.ld_mask + col);   // 8 keep bytes
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dz[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = 2 * q + u;
                    const float d = u ? bf_hi(dw[q]) : bf_lo(dw[q]);
                    const float h = u ? bf_hi(hw[q]) : bf_lo(hw[q]);
                    dz[u] = dz16_elem(d, h, ca[e], ch[e], c0[e], hashed_drop);
                    if constexpr (MASKED) dz[u] = ((m8 >> (8 * e)) & 0xFFull) != 0 ? dz[u] : 0.f;
                }
                const uint32_t w = pack2bf(dz[0], dz[1]);   // both roundings in one instruction
                s[2 * q] += bf_lo(w);
                s[2 * q + 1] += bf_hi(w);
                ow[q] = w;
            }
        }
        const uint4 o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        if (r < a.bs_p && col < a.n_p) *reinterpret_cast<uint4*>(a.DZ + (int64_t)r * a.n_p + col) = o;
    }
    if (a.dbias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rt][c8 + e] = s[e];
    }
    if (a.dbias) __syncthreads();   // (kernel-argument uniform)
    if (a.dbias) {
        for (int t = tid; t < COLS; t += 256) {
            const int c = col0 + t;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < RP; ++i) acc += red[i][t];
            if (c < a.n_p) atomicAdd(&a.dbias[c], (double)acc);
        }
    }
}

// latent: dMU16 = bf16((sum of the split-K slabs of dZlat) + KLD part) on the real rows, 0 on the padding
__global__ void vae_latent_bwd16_kernel(const float* __restrict__ slabs, int nslab, int64_t stride,
                                        const float* __restrict__ dMUk, bf16_t* __restrict__ dMU16, int L_p, int bs,
                                        int bs_p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)bs_p * L_p) return;
    const int r = (int)(i / L_p);
    float t = 0.f;
    if (r < bs) {
        constexpr int kMaxSlabs = 8;
        float v[kMaxSlabs];
#pragma unroll
        for (int s = 0; s < kMaxSlabs; ++s) v[s] = s < nslab ? slabs[(int64_t)s * stride + i] : 0.f;
        t = dMUk[i];
#pragma unroll
        for (int s = 0; s < kMaxSlabs; ++s)
            if (s < nslab) t += v[s];
        for (int s = kMaxSlabs; s < nslab; ++s) t += slabs[(int64_t)s * stride + i];
    }
    dMU16[i] = f2bf(t);
}

// ---- D-Adapt-Adam for the bf16 step ------------------------------------------------------------------------------------
// Same update as vae_dadapt_kernel.  Differences: a workgroup that belongs to a weight MATRIX covers a 32 x 32 tile of
// it (instead of 1024 consecutive elements), so that besides P / M1 / M2 / S it can write the bf16 shadow W16 and --
// through one LDS transpose -- the transposed shadow W16T that the input-gradient GEMMs contract against; and the
// gradient of a weight that consumes BatchNorm-ed activations is completed here:  dW = G diag(s) + dbias t^T.
struct Opt16Tensor {
    const double* dsrc;   // fp64 accumulator gradient (vectors), or nullptr
    float dscale;         // 1 / world for accumulators that SyncBN turns into all-rank sums (see TensorDesc::dscale)
    const float* slab;    // split-K slabs (matrices)
    int nslab;
    int rows_p, cols_p;   // padded shape (vectors: rows_p == 1)
    int64_t stride;
    int64_t p_off;        // offset in the flat parameter / moment buffers
    bf16_t* w16;          // [rows_p][cols_p] shadow (matrices) or nullptr
    bf16_t* w16t;         // [cols_p][rows_p] shadow or nullptr
    // BatchNorm of the layer that produced this weight's input (nullptr: the input is not normalised): the scale / shift
    // vectors s, t the forward pass of THIS step folded into the weights (written by vae_fold_bn_kernel)
    const float* bn_scale;
    const float* bn_shift;
    const double* dbias;  // [rows_p] fp64 column sums of this layer's dZ (needed with bn_scale)
    int blk_start;        // first workgroup of the tensor
};
constexpr int kMaxOpt16 = 4 * 2 * 8 + 4;

constexpr unsigned int kOptTicketGroups = 32, kOptTicketStride = 32;   // arrival words: [0] = top, [32 (1 + g)] = group g (128-byte lines)
// The scalar tail of the optimiser step inside the update kernel (ticket == nullptr: a separate vae_dadapt_finalize_kernel follows)
struct Opt16Tail {
    unsigned int* ticket;   // arrival counter, 0 between steps
    StepState* st;
    double* statbuf;        // the step's fp64 accumulators, cleared for the next step
    int nstat;
    int nblocks;            // partial-sum pairs of the whole step
    int adam;
};

// gradient of 4 consecutive elements (tensor-local index `local` = row * cols_p + col): slab sum or fp64 accumulator,
// completed for weights that consume BatchNorm-ed activations:  dW = G diag(s) + dbias t^T
__device__ __forceinline__ float4 opt16_grad(const Opt16Tensor& td, int64_t local, int row, int col, int bs, int allrank = 0) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (td.dsrc) {
        const float sc = allrank ? td.dscale : 1.0f;
        g.x = (float)td.dsrc[local + 0] * sc; g.y = (float)td.dsrc[local + 1] * sc;
        g.z = (float)td.dsrc[local + 2] * sc; g.w = (float)td.dsrc[local + 3] * sc;
        return g;
    }
    int s = 0;
    for (; s + 8 <= td.nslab; s += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4*>(td.slab + (int64_t)(s + q) * td.stride + local);
#pragma unroll
        for (int q = 0; q < 8; ++q) { g.x += v[q].x; g.y += v[q].y; g.z += v[q].z; g.w += v[q].w; }
    }
    for (; s < td.nslab; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(td.slab + (int64_t)s * td.stride + local);
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
    }
    if (td.bn_scale) {
        const float db = (float)td.dbias[row];
        const float4 s4 = *reinterpret_cast<const float4*>(td.bn_scale + col);
        const float4 t4 = *reinterpret_cast<const float4*>(td.bn_shift + col);
        g.x = g.x * s4.x + db * t4.x; g.y = g.y * s4.y + db * t4.y;
        g.z = g.z * s4.z + db * t4.z; g.w = g.w * s4.w + db * t4.w;
    }
    return g;
}

// workgroup `lb` of a tensor -> (row, col, local index, live) of this thread's 4 elements
__device__ __forceinline__ bool opt16_locate(const Opt16Tensor& td, int lb, int& row, int& col, int64_t& local) {
    if (td.rows_p > 1) {
        const int tiles_c = td.cols_p / 32;
        row = (lb / tiles_c) * 32 + (threadIdx.x >> 3);
        col = (lb % tiles_c) * 32 + (threadIdx.x & 7) * 4;
        local = (int64_t)row * td.cols_p + col;
        return row < td.rows_p;
    }
    row = 0;
    local = (int64_t)lb * 1024 + threadIdx.x * 4;
    col = (int)local;
    return local < td.cols_p;
}

// data-parallel path: G[flat] = this rank's complete gradient (then all-reduced over the ranks)
__global__ __launch_bounds__(256) void vae_grad16_kernel(const Opt16Tensor* __restrict__ tab, const uint8_t* __restrict__ blk2t,
                                                         int bs, float* __restrict__ G, int blk0, int allrank) {
    const int blk = (int)blockIdx.x + blk0;   // the launch covers the table's workgroups [blk0, blk0 + gridDim.x)
    const Opt16Tensor td = tab[blk2t[blk]];
    int row, col;
    int64_t local;
    if (!opt16_locate(td, blk - td.blk_start, row, col, local)) return;
    *reinterpret_cast<float4*>(G + td.p_off + local) = opt16_grad(td, local, row, col, bs, allrank);
}

// blk2t[workgroup] = its tensor.  (Until round 5 every workgroup walked the table -- `while (blk >= tab[t + 1].blk_start) ++t` --
// one dependent load per tensor in front of everything else: up to 27 round trips for the workgroups of the last tensors.)
__global__ __launch_bounds__(256) void vae_dadapt16_kernel(const Opt16Tensor* __restrict__ tab, const uint8_t* __restrict__ blk2t, int bs,
                                                           float* __restrict__ P, float* __restrict__ M1,
                                                           float* __restrict__ M2, float* __restrict__ Sv,
                                                           const StepState* st, double* partials, float adam_lr, int blk0,
                                                           const Opt16Tail tail) {   // (st / partials: also written by the tail)
    __shared__ double red[2][4];
    __shared__ bf16_t wt[32][32 + 2];
    const int blk = (int)blockIdx.x + blk0;   // the launch covers the table's workgroups [blk0, blk0 + gridDim.x)
    const Opt16Tensor td = tab[blk2t[blk]];
    const int lb = blk - td.blk_start;
    const bool matrix = td.rows_p > 1;
    int row, col;
    int64_t local;
    const bool live = opt16_locate(td, lb, row, col, local);
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const double sqrt_b2d = sqrt(0.999);
    const double dlr = st->d;
    const float gscale = (float)st->wsum;
    const float a_m = (float)(dlr * (1.0 - 0.9));
    const float a_s = (float)(dlr * (1.0 - sqrt_b2d));
    const float sqrt_b2 = (float)sqrt_b2d;
    const float one_m_b2 = (float)(1.0 - 0.999);
    float num = 0.f, sk = 0.f;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && adam_lr > 0.f) {   // torch.optim.Adam, see vae_dadapt_kernel
        float4 g = opt16_grad(td, local, row, col, bs);
        g.x *= gscale; g.y *= gscale; g.z *= gscale; g.w *= gscale;
        const int64_t o = td.p_off + local;
        p = *reinterpret_cast<float4*>(P + o);
        float4 m = *reinterpret_cast<float4*>(M1 + o), v = *reinterpret_cast<float4*>(M2 + o);
        const double tt = (double)(st->k + 1);
        const float step_size = (float)((double)adam_lr / (1.0 - pow(0.9, tt)));
        const float bc2_sqrt = (float)sqrt(1.0 - pow(0.999, tt));
        float* pg = &g.x; float* pp = &p.x; float* pm = &m.x; float* pv = &v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = pg[e];
            pm[e] = pm[e] + (gi - pm[e]) * (1.0f - b1);
            pv[e] = pv[e] * b2 + one_m_b2 * gi * gi;
            pp[e] -= step_size * (pm[e] / (sqrtf(pv[e]) / bc2_sqrt + eps));
        }
        *reinterpret_cast<float4*>(P + o) = p;
        *reinterpret_cast<float4*>(M1 + o) = m;
        *reinterpret_cast<float4*>(M2 + o) = v;
    } else if (live) {
        float4 g = opt16_grad(td, local, row, col, bs);
        g.x *= gscale; g.y *= gscale; g.z *= gscale; g.w *= gscale;
        const int64_t o = td.p_off + local;
        p = *reinterpret_cast<float4*>(P + o);
        float4 m = *reinterpret_cast<float4*>(M1 + o), v = *reinterpret_cast<float4*>(M2 + o),
               s = *reinterpret_cast<float4*>(Sv + o);
        float* pg = &g.x; float* pp = &p.x; float* pm = &m.x; float* pv = &v.x; float* ps = &s.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = pg[e];
            num += gi * (ps[e] / (sqrtf(pv[e]) + eps));
            pm[e] = pm[e] * b1 + a_m * gi;
            pv[e] = pv[e] * b2 + one_m_b2 * gi * gi;
            ps[e] = ps[e] * sqrt_b2 + a_s * gi;
            sk += fabsf(ps[e]);
            pp[e] -= pm[e] / (sqrtf(pv[e]) + eps);
        }
        *reinterpret_cast<float4*>(P + o) = p;
        *reinterpret_cast<float4*>(M1 + o) = m;
        *reinterpret_cast<float4*>(M2 + o) = v;
        *reinterpret_cast<float4*>(Sv + o) = s;
    }
    if (matrix && td.w16) {   // uniform per workgroup
        const uint2 w = make_uint2(pack_bf2(p.x, p.y), pack_bf2(p.z, p.w));
        if (live) *reinterpret_cast<uint2*>(td.w16 + local) = w;
        if (td.w16t) {
            const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 4;
            wt[tr][tc + 0] = (bf16_t)(w.x & 0xFFFFu); wt[tr][tc + 1] = (bf16_t)(w.x >> 16);
            wt[tr][tc + 2] = (bf16_t)(w.y & 0xFFFFu); wt[tr][tc + 3] = (bf16_t)(w.y >> 16);
            __syncthreads();
            // thread (c = tid >> 3, r4 = (tid & 7) * 4): 4 consecutive rows of column c -> 8 contiguous bytes of W16T
            const int c = threadIdx.x >> 3, r4 = (threadIdx.x & 7) * 4;
            const int tiles_c = td.cols_p / 32;
            const int gr = (lb / tiles_c) * 32 + r4, gc = (lb % tiles_c) * 32 + c;
            uint2 o;
            o.x = (uint32_t)wt[r4 + 0][c] | ((uint32_t)wt[r4 + 1][c] << 16);
            o.y = (uint32_t)wt[r4 + 2][c] | ((uint32_t)wt[r4 + 3][c] << 16);
            if (gr < td.rows_p && gc < td.cols_p) *reinterpret_cast<uint2*>(td.w16t + (int64_t)gc * td.rows_p + gr) = o;
        }
    }
    const double wn = wave_sum_f64((double)num), ws = wave_sum_f64((double)sk);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = wn; red[1][wave] = ws; }
    __syncthreads();
    const double part0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double part1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    if (tail.ticket == nullptr) {
        if (threadIdx.x == 0) {
            partials[(int64_t)blk * 2 + 0] = part0;
            partials[(int64_t)blk * 2 + 1] = part1;
        }
        return;
    }
    // ---- the scalar part of the step (vae_dadapt_finalize_kernel) by the workgroup that arrives LAST, instead of one more
    // dependent launch.  Hand-off as the CDNA4 guide prescribes for data another CU will read: write-through (device-scope)
    // stores of the partial sums, drained (s_waitcnt vmcnt(0)), THEN the arrival ticket; the last arriver reads every partial
    // with device-scope loads.  No fence (a release fence writes the whole L2 back).  Every other workgroup has read the step
    // state and its accumulators before it took its ticket, so the last one may update / clear them.
    // Two levels of tickets (round 6): one word takes ~88 atomics per microsecond, so the ~900 arrivals of a launch on ONE word cost
    // ~10 us (round 5 measured the one-level tail 5 us SLOWER than the launch it replaced).  Workgroup b arrives at group word
    // b % 32 (each word on its own 128-byte line); the last arrival of a group re-arms its word and arrives at the top word; the
    // last arrival there runs the tail: at most ~28 + 32 arrivals on any word.
    __shared__ int last_s;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partials[(int64_t)blk * 2 + 0], part0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&partials[(int64_t)blk * 2 + 1], part1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int nb = gridDim.x, grp = (unsigned int)blockIdx.x % kOptTicketGroups;
        const unsigned int ngroups = nb < kOptTicketGroups ? nb : kOptTicketGroups;
        const unsigned int gsize = (nb - grp + kOptTicketGroups - 1) / kOptTicketGroups;
        unsigned int* const gword = tail.ticket + kOptTicketStride * (1 + grp);
        int last = 0;
        if (__hip_atomic_fetch_add(gword, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
            __hip_atomic_store(gword, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // armed for the next step
            last = __hip_atomic_fetch_add(tail.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u;
        }
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    StepState* const stw = tail.st;
    for (int i = threadIdx.x; i < tail.nstat; i += 256) tail.statbuf[i] = 0.0;
    __shared__ double fin[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < tail.nblocks; i += 256) {   // (same order as the finalize kernel: same bits)
        a += __hip_atomic_load(&partials[(int64_t)i * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b += __hip_atomic_load(&partials[(int64_t)i * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    fin[0][threadIdx.x] = a;
    fin[1][threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            fin[0][threadIdx.x] += fin[0][threadIdx.x + off];
            fin[1][threadIdx.x] += fin[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sqrt_b2s = sqrt(0.999);
        const double d = stw->d;
        const double numerator_acum = d * fin[0][0];
        const double sk_l1 = fin[1][0];
        const double nw = sqrt_b2s * stw->numerator_weighted + (1.0 - sqrt_b2s) * numerator_acum;
        if (tail.adam) {
            stw->k += 1;
        } else if (sk_l1 != 0.0) {
            const double d_hat = nw / ((1.0 - sqrt_b2s) * sk_l1);
            stw->d = d_hat > d ? d_hat : d;
            stw->numerator_weighted = nw;
            stw->k += 1;
        }
        stw->step += 1;
        stw->batch += 1;
        __hip_atomic_store(tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // armed for the next step
    }
}

}  // namespace vh
