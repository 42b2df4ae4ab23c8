// vae_kernels.hpp -- the non-GEMM kernels of the VAE step (gfx950): batch gather, BatchNorm
// statistics / apply / backward, reparameterisation, fused softmax-CE/SSE/KLD loss with its
// backward seed, column reductions, and the fused D-Adapt-Adam update.
//
// Conventions: activations are row-major [bs_p][n_p] (bs_p multiple of 128, n_p multiple of 32);
// rows >= bs and columns >= n are padding and carry zeros wherever a zero is required for
// correctness (inputs, dropout-ed activations' gradients).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gemm.hpp"

namespace vh {

constexpr int kRB = 32;            // rows per workgroup in the column-parallel kernels
constexpr float kBnEps = 1e-5f;    // torch.nn.BatchNorm1d defaults (encode.py:238,246)
constexpr float kBnMomentum = 0.1f;
constexpr float kLeakySlope = 0.01f;  // torch.nn.LeakyReLU default (encode.py:252)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// The same reductions on the VALU's data-parallel primitives (DPP) instead of six ds_bpermute round trips through the LDS
// crossbar: quad swaps, the two row mirrors (every lane of a 16-lane row then holds its row's result), row_bcast:15 / :31 (lane 63
// holds the wave's) and one v_readlane.  ~8 VALU-rate steps instead of 6 x (LDS issue + lgkmcnt wait) -- in a kernel whose duration
// IS one wavefront's dependent chain (the bf16 loss kernel: one row per wavefront, ten reductions per row, 60 ds_bpermute in its
// ISA) that is microseconds.  The order of the additions differs from the butterfly's: same value up to float32 rounding.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float x) {   // rows outside ROW_MASK keep x
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, false));
}
template <class Op>
__device__ __forceinline__ float wave_reduce_dpp(float v, Op op) {
    v = op(v, dpp_move<0xB1, 0xF>(v));    // quad_perm [1,0,3,2]
    v = op(v, dpp_move<0x4E, 0xF>(v));    // quad_perm [2,3,0,1]
    v = op(v, dpp_move<0x141, 0xF>(v));   // row_half_mirror
    v = op(v, dpp_move<0x140, 0xF>(v));   // row_mirror: every lane of a row now holds the row's result
    // row_bcast:15 brings lane 15 of rows 0/2 into rows 1/3, row_bcast:31 lane 31 into rows 2/3.  Only lane 63 is read afterwards and
    // both steps write its row, so what the other rows hold after these two steps does not matter.
    v = op(v, dpp_move<0x142, 0xA>(v));
    v = op(v, dpp_move<0x143, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float v) { return wave_reduce_dpp(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_reduce_dpp(v, [](float a, float b) { return fmaxf(a, b); }); }

// ---- batch assembly ----------------------------------------------------------------------------
// Epoch shuffle without a permutation array: a keyed bijection of [0, 2^bits) (odd multiplies and
// xor-shifts, each invertible modulo 2^bits) restricted to [0, n) by cycle walking.  Every epoch uses a
// fresh key, so the gather kernel draws its rows directly from position (batch * bs + r) -- no host
// randperm, no upload, no sort.
struct ShuffleSpec {
    unsigned long long key;   // 0 = identity (no shuffle)
    unsigned long long n;     // rows in the (local) dataset
    int bits;                 // ceil(log2(n)), >= 1
};

__host__ __device__ __forceinline__ unsigned long long shuffle_round(unsigned long long x, unsigned long long key,
                                                                      unsigned long long mask, int bits) {
    const int s1 = (bits + 1) / 2, s2 = (bits + 2) / 3 > 0 ? (bits + 2) / 3 : 1;
    x = (x * 0x9E3779B97F4A7C15ull + key) & mask;
    x ^= x >> s1;
    x = (x * 0xBF58476D1CE4E5B9ull + (key >> 17)) & mask;
    x ^= x >> s2;
    x = (x * 0x94D049BB133111EBull + (key >> 31)) & mask;
    x ^= x >> s1;
    return x;
}

__host__ __device__ __forceinline__ unsigned long long shuffle_index(const ShuffleSpec& sp, unsigned long long i) {
    if (sp.key == 0ull) return i;
    const unsigned long long mask = (sp.bits >= 64) ? ~0ull : ((1ull << sp.bits) - 1ull);
    unsigned long long x = i;
    do { x = shuffle_round(x, sp.key, mask, sp.bits); } while (x >= sp.n);
    return x;
}

// Label block of the semi-supervised models (semisupervised_encode.py:25-47: the collate functions one-hot the integer label
// of every row).  The dataset keeps the int32 label per row; the batch row gets a 1.0 at column col0 + label.
struct LabelSrc {
    const int32_t* labels;   // [n] or nullptr (no label block)
    int col0;                // first label column of the batch row
};

// Xb[r][:] = X[src(r)][:], Wb[r] = w[src(r)] for r < bs; zero rows for the padding.  blockDim (64,4).
// src(r) = idx[first + r] when an explicit row list is given, else shuffle(first + r); first = base + batch * bs
// with the batch index read from device memory (advanced by the optimiser's finalize kernel).
// The dataset row is ld_src wide (0: the dataset has no feature columns), the batch row ldx; Lb[r] = label of the row.
// LABELS = false: the plain VAE's kernel as it was before the label block existed (see vae_gather16_kernel).
template <bool LABELS>
__global__ void vae_gather_kernel(const float* __restrict__ X, int64_t ld_src, int64_t ldx, const float* __restrict__ w_all,
                                  const int64_t* __restrict__ idx, const ShuffleSpec shuffle,
                                  const long long* __restrict__ batch_ptr, int64_t base, int bs, int bs_p,
                                  float* __restrict__ Xb, float* __restrict__ Wb, const LabelSrc lab,
                                  int32_t* __restrict__ Lb) {
    const int r = blockIdx.x * 4 + threadIdx.y;
    if (r >= bs_p) return;
    const bool real = r < bs;
    const int64_t first = base + (batch_ptr ? (int64_t)(*batch_ptr) * bs : 0);
    int64_t src = 0;
    if (real) src = idx ? idx[first + r] : (int64_t)shuffle_index(shuffle, (unsigned long long)(first + r));
    float4* d = reinterpret_cast<float4*>(Xb + (int64_t)r * ldx);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const int dq = (int)(ldx / 4);
    if constexpr (!LABELS) {
        const float4* s = reinterpret_cast<const float4*>(X + src * ldx);
        for (int c = threadIdx.x; c < dq; c += 64) {
            float4 v = z;
            if (real) v = s[c];
            d[c] = v;
        }
        if (threadIdx.x == 0) Wb[r] = real ? w_all[src] : 0.f;
    } else {
        const float4* s = reinterpret_cast<const float4*>(X + src * ld_src);
        const int sq = (int)(ld_src / 4);
        int hot = -1;
        if (lab.labels && real) hot = lab.col0 + lab.labels[src];
        for (int c = threadIdx.x; c < dq; c += 64) {
            float4 v = z;
            if (real && c < sq) v = s[c];
            if ((hot >> 2) == c && hot >= 0) {   // (selects, not an indexed write: the compiler moves an indexed float4 into LDS)
                const int e = hot & 3;
                v.x = e == 0 ? 1.0f : v.x; v.y = e == 1 ? 1.0f : v.y; v.z = e == 2 ? 1.0f : v.z; v.w = e == 3 ? 1.0f : v.w;
            }
            d[c] = v;
        }
        if (threadIdx.x == 0) {
            if (Wb) Wb[r] = real ? w_all[src] : 0.f;
            if (Lb) Lb[r] = hot >= 0 ? hot - lab.col0 : 0;
        }
    }
}

// data-parallel planning on the device: out[b] = sum of the weights of this rank's rows of batch b
__global__ __launch_bounds__(256) void vae_batch_wsum_kernel(const float* __restrict__ w_all,
                                                             const int64_t* __restrict__ idx,
                                                             const ShuffleSpec shuffle, int bs,
                                                             float* __restrict__ out) {
    __shared__ double red[256];
    const int64_t first = (int64_t)blockIdx.x * bs;
    double s = 0.0;
    for (int r = threadIdx.x; r < bs; r += 256) {
        const int64_t src = idx ? idx[first + r] : (int64_t)shuffle_index(shuffle, (unsigned long long)(first + r));
        s += (double)w_all[src];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)red[0];
}

// out[0] = sum of v[0..n)  (single workgroup, fixed tree => deterministic)
__global__ __launch_bounds__(256) void vae_sum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

// ---- BatchNorm1d running statistics (training): momentum 0.1, unbiased variance for running_var -------
// fstat = a layer's fp64 batch sums [2][n_p] accumulated by the forward GEMM's epilogue.  One launch
// (blockIdx.y = layer) updates every hidden layer.
struct RunningTable {
    int n;
    const double* fstat[16];
    float* rm[16];
    float* rv[16];
    int n_p[16];
};
__global__ void vae_bn_running_kernel(const RunningTable tab, int bs) {
    const int l = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= tab.n || col >= tab.n_p[l]) return;
    const double* fstat = tab.fstat[l];
    const int n_p = tab.n_p[l];
    const double m = fstat[col] / bs;
    double var = fstat[n_p + col] / bs - m * m;
    if (var < 0.0) var = 0.0;
    const double unbiased = bs > 1 ? var * ((double)bs / (double)(bs - 1)) : var;
    tab.rm[l][col] = (1.0f - kBnMomentum) * tab.rm[l][col] + kBnMomentum * (float)m;
    tab.rv[l][col] = (1.0f - kBnMomentum) * tab.rv[l][col] + kBnMomentum * (float)unbiased;
}

// eval mode: scale/shift from the running statistics
__global__ void vae_bn_eval_coeff_kernel(int n_p, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ rm, const float* __restrict__ rv,
                                         float* __restrict__ scale, float* __restrict__ shift) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_p) return;
    const float istd = 1.0f / sqrtf(rv[col] + kBnEps);
    const float sc = gamma[col] * istd;
    scale[col] = sc;
    shift[col] = beta[col] - rm[col] * sc;
}

// ---- reparameterisation (encode.py:276-286) ------------------------------------------------------
// MU = sum of the split-K slabs + bias;  Z = MU + eps on the real rows / columns, 0 on the padding.
// eps comes from E (injected, parity mode) or is generated in place (E == nullptr); noise == 0 disables it.
__global__ void vae_reparam_kernel(const float* __restrict__ slabs, int nslab, int64_t stride,
                                   const float* __restrict__ bias, const float* __restrict__ E, uint64_t key,
                                   const unsigned long long* __restrict__ step_ptr, int noise,
                                   float* __restrict__ MU, float* __restrict__ Z, int bs, int L, int L_p, int bs_p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)bs_p * L_p) return;
    const int r = (int)(i / L_p), c = (int)(i % L_p);
    float m = bias[c];
    {   // all slab loads in flight before the first add (ascending order kept)
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
    Z[i] = z;
}

// ---- loss (encode.py:316-357) + backward seed ------------------------------------------------------
// One wavefront per row.  Emits dL/d(recon) and the KLD part of dL/d(mu), and per-workgroup partial
// sums of the four per-row loss terms.  NOTE the reference's broadcasting quirk (encode.py:347):
// the [B] row losses times the [B,1] weights form a [B,B] outer product, so
//   loss.mean() = mean_i(row_i) * mean_j(w_j)   and   dL/d(row_i) = mean(w)/B  for every row.
struct LossArgs {
    const float* R;      // reconstruction [bs_p][ld]
    const float* X;      // batch inputs (targets) [bs_p][ld]
    int64_t ld;
    const float* MU;     // [bs_p][ldl]
    int64_t ldl;
    float inv_b2;        // 1 / B_global^2 ; the factor sum(w) is applied by the optimiser (gradients are
                         // linear in it), so the backward pass does not wait for a reduction over the weights
    int bs, bs_p, S, L;
    float ce_w, ab_w, sse_w, kld_w;
    float* dR;           // [bs_p][ld]
    float* dMUk;         // [bs_p][ldl]
    float* part;         // [gridDim.x][4] : ab, ce, sse, kld (already multiplied by their weights)
    // label block (semisupervised_encode.py:248-257, 515-569): NL logits at columns lab0.. of R, the row's class in Lb;
    // cross-entropy of the row (the reference's CrossEntropyLoss is the batch mean of these) with weight 1 and
    // argmax(logits) == class.  ntnf / nab: 103 / 1, or 0 / 0 for the labels-only model.
    int NL, lab0, ntnf, nab;
    const int32_t* Lb;
    float* lab_part;     // [gridDim.x][2] : cross-entropy sum, correct predictions
    // hierarchical label loss (taxvamb_encode.py:277-538 with hloss_misc.FlatSoftmaxNLL, the default 'flat_softmax'): the row's
    // label is a NODE of the taxonomy, the loss sees only the first n_leaves logits and is -log of the softmax mass on the
    // leaves at or below the node; leaf_masks[node][leaf] says which.  n_leaves == 0: one-hot cross-entropy over all NL logits.
    const uint8_t* leaf_masks;
    int n_leaves;
};

// Cross-entropy of one row's label logits and its gradient (softmax - onehot) * g; returns (ce, correct) on every lane.
// torch.max(dim=1) returns the FIRST maximal index: ties go to the lowest column.
template <class Store>
__device__ __forceinline__ void label_block(const float* __restrict__ r, int NL, int y, float g, Store&& store, float& ce_out,
                                            float& correct_out) {
    const int lane = threadIdx.x & 63;
    float mx = -3.0e38f;
    int arg = 0x7fffffff;
    for (int c = lane; c < NL; c += 64) {
        const float v = r[c];
        if (v > mx) { mx = v; arg = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(mx, off);
        const int oa = __shfl_xor(arg, off);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    float se = 0.f;
    for (int c = lane; c < NL; c += 64) se += expf(r[c] - mx);
    se = wave_sum(se);
    const float inv = 1.0f / se;
    for (int c = lane; c < NL; c += 64) {
        const float p = expf(r[c] - mx) * inv;
        store(c, g * (p - (c == y ? 1.0f : 0.0f)));
    }
    ce_out = (logf(se) + mx) - r[y];
    correct_out = arg == y ? 1.0f : 0.0f;
}

// hloss_misc.py:1121-1133 for one row: logp = log_softmax(first n_leaves logits); loss = -logsumexp(logp over the leaves of the
// row's node); gradient g * (p - [leaf in mask] * p / sum_mask p) on those logits, 0 on the other NL - n_leaves columns (the
// reference's narrow(1, .., n_leaves) never lets a gradient reach them).  "correct" does not exist here (the HLoss classes
// return a constant 0, taxvamb_encode.py:355).  Both exponent sums are taken about their own maxima, so a node whose leaves hold
// a vanishing share of the softmax mass still has a finite loss, as torch.logsumexp has.
template <class Store>
__device__ __forceinline__ void label_block_hier(const float* __restrict__ r, int NL, int n_leaves, const uint8_t* __restrict__ mask,
                                                 float g, Store&& store, float& ce_out) {
    const int lane = threadIdx.x & 63;
    float mx = -3.0e38f, mxm = -3.0e38f;
    for (int c = lane; c < n_leaves; c += 64) {
        const float v = r[c];
        mx = fmaxf(mx, v);
        if (mask[c]) mxm = fmaxf(mxm, v);
    }
    mx = wave_max(mx);
    mxm = wave_max(mxm);
    float se = 0.f, sm = 0.f;
    for (int c = lane; c < n_leaves; c += 64) {
        const float v = r[c];
        se += expf(v - mx);
        if (mask[c]) sm += expf(v - mxm);
    }
    se = wave_sum(se);
    sm = wave_sum(sm);
    const float inv = 1.0f / se, invm = 1.0f / sm;
    for (int c = lane; c < NL; c += 64) {
        float d = 0.f;
        if (c < n_leaves) {
            const float v = r[c];
            d = expf(v - mx) * inv;
            if (mask[c]) d -= expf(v - mxm) * invm;
        }
        store(c, g * d);
    }
    ce_out = (logf(se) + mx) - (logf(sm) + mxm);
}

__global__ __launch_bounds__(256) void vae_loss_kernel(const LossArgs a) {
    __shared__ float red[4][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    float ab_t = 0.f, ce_t = 0.f, sse_t = 0.f, kld_t = 0.f, cel_t = 0.f, hit_t = 0.f;
    if (row < a.bs_p) {
        float* dr = a.dR + (int64_t)row * a.ld;
        float* dm = a.dMUk + (int64_t)row * a.ldl;
        if (row >= a.bs) {
            for (int c = lane; c < a.ld; c += 64) dr[c] = 0.f;
            for (int c = lane; c < a.ldl; c += 64) dm[c] = 0.f;
        } else {
            const float* r = a.R + (int64_t)row * a.ld;
            const float* x = a.X + (int64_t)row * a.ld;
            const float g = a.inv_b2;
            const int S = a.S;
            // softmax over the S abundance logits (S = 0: the labels-only model has none; every loop below is empty)
            float mx = -3.0e38f;
            for (int c = lane; c < S; c += 64) mx = fmaxf(mx, r[c]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int c = lane; c < S; c += 64) se += expf(r[c] - mx);
            se = wave_sum(se);
            const float inv = S > 0 ? 1.0f / se : 0.0f;
            float ce = 0.f, pdp = 0.f;
            for (int c = lane; c < S; c += 64) {
                const float p = expf(r[c] - mx) * inv;
                const float q = p + 1e-9f;
                ce -= logf(q) * x[c];
                pdp += p * (-x[c] / q);
            }
            ce = wave_sum(ce);
            pdp = wave_sum(pdp);
            const float gce = g * a.ce_w;
            for (int c = lane; c < S; c += 64) {
                const float p = expf(r[c] - mx) * inv;
                const float dp = -x[c] / (p + 1e-9f);
                dr[c] = gce * p * (dp - pdp);
            }
            // TNF sum of squared errors
            float sse = 0.f;
            const float gsse = g * a.sse_w * 2.0f;
            for (int c = S + lane; c < S + a.ntnf; c += 64) {
                const float diff = r[c] - x[c];
                sse += diff * diff;
                dr[c] = gsse * diff;
            }
            sse = wave_sum(sse);
            // total-abundance squared error (one column)
            float ab = 0.f;
            if (lane == 0 && a.nab) {
                const int c = S + a.ntnf;
                const float diff = r[c] - x[c];
                ab = diff * diff;
                dr[c] = g * a.ab_w * 2.0f * diff;
            }
            ab = wave_sum(ab);
            // label logits
            if (a.NL > 0 && a.n_leaves > 0)
                label_block_hier(r + a.lab0, a.NL, a.n_leaves, a.leaf_masks + (int64_t)a.Lb[row] * a.n_leaves, g,
                                 [&](int c, float v) { dr[a.lab0 + c] = v; }, cel_t);
            else if (a.NL > 0) label_block(r + a.lab0, a.NL, a.Lb[row], g, [&](int c, float v) { dr[a.lab0 + c] = v; }, cel_t, hit_t);
            // zero the padding columns
            for (int c = S + a.ntnf + a.nab + a.NL + lane; c < a.ld; c += 64) dr[c] = 0.f;
            // KLD = 0.5 * sum(mu^2)
            const float* mu = a.MU + (int64_t)row * a.ldl;
            float kld = 0.f;
            const float gk = g * a.kld_w;
            for (int c = lane; c < a.ldl; c += 64) {
                const float m = c < a.L ? mu[c] : 0.f;
                kld += m * m;
                dm[c] = gk * m;
            }
            kld = 0.5f * wave_sum(kld);
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

// device-resident scalars shared by the step kernels
struct StepState {
    double d;                  // D-Adapt estimate
    double numerator_weighted;
    long long k;
    unsigned long long step;   // global step counter (seeds dropout / noise)
    long long batch;           // index of the current batch inside the epoch's row list
    double wsum;               // sum of the (all-rank) batch weights of the current step
    double step_loss[5];       // loss, ab, ce, sse, kld of the last step (calc_loss order)
    double epoch_loss[5];      // running sums over the epoch's batches
    long long epoch_batches;
    double epoch_label[2];     // label block: sums over the epoch of the batch-mean cross-entropy / of the correct predictions
    double step_label[2];      // ... of the last step
};

// reduce the loss partials, produce the five means (encode.py:350-356), add them to the epoch sums and
// publish the batch weight sum: sum(Wb) on one GPU, gwsum[batch] (planned on the host) under data
// parallelism.  bs_global: rows of the whole (all-rank) batch; every rank adds its own share
// local_sum / bs_global and the epoch sums are all-reduced once per epoch.
// One workgroup of kLossFinThreads threads.  The loads are 16 bytes wide and issued in batches of four per thread before
// the first add (the 256-thread scalar version walked 8 + 32 DEPENDENT load round trips per thread: 17 us for 40 KB);
// fp64 wavefront reductions, one LDS exchange.
constexpr int kLossFinThreads = 512;   // (1024 threads left 128 VGPRs per thread: the seven fp64 sums and the batched loads spilled to scratch)
__global__ __launch_bounds__(kLossFinThreads) void vae_loss_finalize_kernel(const float* __restrict__ part, int nblocks,
                                                                            const float* __restrict__ Wb, int bs,
                                                                            const float* __restrict__ gwsum, int bs_global,
                                                                            StepState* __restrict__ st,
                                                                            const float* __restrict__ lab_part) {
    __shared__ double red[7][kLossFinThreads / 64];
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    if (lab_part)   // [nblocks][2]: cross-entropy sums, correct predictions
        for (int b = threadIdx.x; b < nblocks; b += kLossFinThreads) {
            const float2 v = reinterpret_cast<const float2*>(lab_part)[b];
            s[5] += (double)v.x; s[6] += (double)v.y;
        }
    const float4* p4 = reinterpret_cast<const float4*>(part);   // one float4 (ab, ce, sse, kld) per loss workgroup
    for (int b0 = threadIdx.x; b0 < nblocks; b0 += 4 * kLossFinThreads) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * kLossFinThreads;
            v[u] = b < nblocks ? p4[b] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[0] += (double)v[u].x; s[1] += (double)v[u].y; s[2] += (double)v[u].z; s[3] += (double)v[u].w; }
    }
    if (!gwsum) {
        const int n4 = bs >> 2;   // Wb is a 16-byte aligned device buffer
        const float4* w4 = reinterpret_cast<const float4*>(Wb);
        for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * kLossFinThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kLossFinThreads;
                v[u] = i < n4 ? w4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s[4] += ((double)v[u].x + (double)v[u].y) + ((double)v[u].z + (double)v[u].w);
        }
        for (int i = 4 * n4 + threadIdx.x; i < bs; i += kLossFinThreads) s[4] += (double)Wb[i];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        double v = s[t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[t][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[7];
        for (int t = 0; t < 7; ++t) {
            double v = 0.0;
            for (int w = 0; w < kLossFinThreads / 64; ++w) v += red[t][w];
            tot[t] = v;
        }
        const double bsg = (double)bs_global;
        const double wsum = gwsum ? (double)gwsum[st->batch] : (double)(float)tot[4];
        const double ab = tot[0] / bsg, ce = tot[1] / bsg, sse = tot[2] / bsg, kld = tot[3] / bsg;
        const double wmean = wsum / bsg;
        // the label cross-entropy (weight 1) joins the reconstruction terms (semisupervised_encode.py:553-556; a model without
        // a label block adds exactly 0.0)
        const double cel = tot[5] / bsg;
        const double loss = lab_part ? (((ce + ab + sse) + cel) + kld) * wmean : ((ce + ab + sse) + kld) * wmean;
        const double v[5] = {loss, ab, ce, sse, kld};
        for (int t = 0; t < 5; ++t) { st->step_loss[t] = v[t]; st->epoch_loss[t] += v[t]; }
        st->step_label[0] = cel; st->step_label[1] = tot[6];
        st->epoch_label[0] += cel; st->epoch_label[1] += tot[6];
        st->epoch_batches += 1;
        st->wsum = wsum;
    }
}

// softmax of the first S columns of R (the module's depths_out, encode.py:302); one wave per row
__global__ __launch_bounds__(256) void vae_softmax_out_kernel(const float* __restrict__ R, int64_t ld, int bs, int S,
                                                              float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= bs) return;
    const float* r = R + (int64_t)row * ld;
    float mx = -3.0e38f;
    for (int c = lane; c < S; c += 64) mx = fmaxf(mx, r[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < S; c += 64) se += expf(r[c] - mx);
    se = wave_sum(se);
    for (int c = lane; c < S; c += 64) out[(int64_t)row * S + c] = expf(r[c] - mx) / se;
}

// ---- column-parallel backward helpers -----------------------------------------------------------------
// blockDim (32, 8): a workgroup owns 32 columns x kRB rows; thread (tx, ty) walks rows r0+ty, r0+ty+8, ...
// and the 8 row-lanes are combined through LDS in a fixed order.
constexpr int kCT = 32;
constexpr int kRL = 8;

__device__ __forceinline__ float column_block_sum(float v, float (*red)[kCT + 1]) {
    red[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.y == 0)
#pragma unroll
        for (int i = 0; i < kRL; ++i) s += red[i][threadIdx.x];
    __syncthreads();
    return s;  // valid on threadIdx.y == 0
}

// part[rb][col] = sum over the workgroup's rows of G[row][col]
__global__ __launch_bounds__(256) void vae_colsum_partial_kernel(const float* __restrict__ G, int64_t ld, int n_p,
                                                                 int rows, float* __restrict__ part) {
    __shared__ float red[kRL][kCT + 1];
    const int col = blockIdx.x * kCT + threadIdx.x;
    const int r0 = blockIdx.y * kRB, r1 = min(rows, r0 + kRB);
    float s = 0.f;
    if (col < n_p)
        for (int r = r0 + threadIdx.y; r < r1; r += kRL) s += G[(int64_t)r * ld + col];
    s = column_block_sum(s, red);
    if (threadIdx.y == 0 && col < n_p) part[(int64_t)blockIdx.y * n_p + col] = s;
}

// Elementwise backward of one hidden layer (encode.py:264: BN(dropout(leaky_relu(z)))), one pass:
//   dZ = keep * slope(h) * drop_scale * istd*gamma * (dA - S1/B - xhat * S2/B),   xhat = (h - mean) * istd
// with mean/istd from the forward batch sums (fstat) and S1 = sum dA, S2 = sum dA*xhat (bstat) that the
// producing GEMM epilogues accumulated; also the column sums of dZ (bias gradient, fp64 atomics).
// "kept" is read off H itself when the mask was generated (a dropped unit was stored as exactly 0; a kept
// unit with z == 0.0f exactly is the only mis-classified case and carries the 0.01-slope gradient of a
// measure-zero event); injected masks (parity tests) are looked up.
struct DzArgs {
    const float* DA;
    const float* H;
    float* DZ;
    int n_p, bs, bs_p;
    BnSrc bn;
    const double* bstat;
    float drop_scale;
    const uint8_t* drop_mask;
    int64_t ld_mask;
    double* dbias;   // [n_p]
};

// blockDim (32, 8): a workgroup owns kDzCols = 128 columns (one float4 quad per threadIdx.x) x kDzRows = 64
// rows (8 per thread): 512-byte row segments, 16 independent 16-byte loads in flight per thread, one
// workgroup per CU at the C1 shape (4096 x 512).
constexpr int kDzCols = 128;
constexpr int kDzRows = 64;

__global__ __launch_bounds__(256) void vae_dz_kernel(const DzArgs a) {
    __shared__ float red[kRL][kDzCols + 4];
    __shared__ float cf[3][kDzCols];
    const int tid = threadIdx.y * 32 + threadIdx.x;
    if (tid < kDzCols) {
        const int col = blockIdx.x * kDzCols + tid;
        float ca = 0.f, ch = 0.f, c0 = 0.f;
        if (col < a.n_p) {
            float mean, istd, sc, sh;
            bn_column(a.bn, col, mean, istd, sc, sh);
            const double inv_bs = 1.0 / (double)a.bn.bs;   // the statistics' batch (all ranks under SyncBN)
            const float c1 = (float)(a.bstat[col] * inv_bs);
            const float c2 = (float)(a.bstat[a.n_p + col] * inv_bs);
            ca = a.drop_scale * istd * a.bn.gamma[col];
            ch = -ca * istd * c2;
            c0 = -ca * c1 - ch * mean;
        }
        cf[0][tid] = ca; cf[1][tid] = ch; cf[2][tid] = c0;
    }
    __syncthreads();
    const int cq = 4 * threadIdx.x;                       // first of this thread's 4 columns inside the tile
    const int col = blockIdx.x * kDzCols + cq;
    const float4 ca = *reinterpret_cast<const float4*>(&cf[0][cq]);
    const float4 ch = *reinterpret_cast<const float4*>(&cf[1][cq]);
    const float4 c0 = *reinterpret_cast<const float4*>(&cf[2][cq]);
    const int r0 = blockIdx.y * kDzRows;
    const bool hashed_drop = (a.drop_scale != 1.0f) && (a.drop_mask == nullptr);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < a.n_p) {
        constexpr int RPT = kDzRows / kRL;
        float4 da[RPT], hh[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + threadIdx.y + kRL * k;
            da[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            hh[k] = da[k];
            if (r < a.bs) {
                const int64_t i = (int64_t)r * a.n_p + col;
                da[k] = *reinterpret_cast<const float4*>(a.DA + i);
                hh[k] = *reinterpret_cast<const float4*>(a.H + i);
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + threadIdx.y + kRL * k;
            if (r >= a.bs_p) continue;
            float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < a.bs) {
                bool k0 = true, k1 = true, k2 = true, k3 = true;
                if (hashed_drop) {
                    k0 = hh[k].x != 0.f; k1 = hh[k].y != 0.f; k2 = hh[k].z != 0.f; k3 = hh[k].w != 0.f;
                } else if (a.drop_mask) {
                    const uint8_t* m = a.drop_mask + (int64_t)r * a.ld_mask + col;
                    k0 = m[0] != 0; k1 = m[1] != 0; k2 = m[2] != 0; k3 = m[3] != 0;
                }
                const float l0 = ca.x * da[k].x + ch.x * hh[k].x + c0.x;
                const float l1 = ca.y * da[k].y + ch.y * hh[k].y + c0.y;
                const float l2 = ca.z * da[k].z + ch.z * hh[k].z + c0.z;
                const float l3 = ca.w * da[k].w + ch.w * hh[k].w + c0.w;
                dz.x = k0 ? l0 * (hh[k].x > 0.f ? 1.0f : kLeakySlope) : 0.f;
                dz.y = k1 ? l1 * (hh[k].y > 0.f ? 1.0f : kLeakySlope) : 0.f;
                dz.z = k2 ? l2 * (hh[k].z > 0.f ? 1.0f : kLeakySlope) : 0.f;
                dz.w = k3 ? l3 * (hh[k].w > 0.f ? 1.0f : kLeakySlope) : 0.f;
            }
            *reinterpret_cast<float4*>(a.DZ + (int64_t)r * a.n_p + col) = dz;
            s.x += dz.x; s.y += dz.y; s.z += dz.z; s.w += dz.w;
        }
    }
    // bias gradient: the 8 row lanes combined through LDS in a fixed order, one fp64 atomic per column
    *reinterpret_cast<float4*>(&red[threadIdx.y][cq]) = s;
    __syncthreads();
    if (tid < kDzCols) {
        const int c = blockIdx.x * kDzCols + tid;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < kRL; ++i) t += red[i][tid];
        if (c < a.n_p) atomicAdd(&a.dbias[c], (double)t);
    }
}

// latent: dMU = (sum of the split-K slabs of dZlat) + KLD part (zero on padding rows); column partial
// sums for the mu bias
__global__ __launch_bounds__(256) void vae_latent_bwd_kernel(const float* __restrict__ slabs, int nslab,
                                                             int64_t stride, const float* __restrict__ dMUk,
                                                             float* __restrict__ DZ, int L_p, int bs, int bs_p,
                                                             float* __restrict__ part) {
    __shared__ float red[kRL][kCT + 1];
    const int col = blockIdx.x * kCT + threadIdx.x;
    const int r0 = blockIdx.y * kRB, r1 = min(bs_p, r0 + kRB);
    float s = 0.f;
    if (col < L_p) {
        // every load of the thread's kRB / kRL rows is issued before the first add (the kernel is a chain of
        // dependent L2 latencies otherwise: 10.6 us for 4.7 MB)
        constexpr int RPT = kRB / kRL;
        constexpr int kMaxSlabs = 8;
        float v[RPT], sl[RPT][kMaxSlabs];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + threadIdx.y + kRL * k;
            const int64_t i = (int64_t)r * L_p + col;
            const bool live = r < r1 && r < bs;
            v[k] = live ? dMUk[i] : 0.f;
#pragma unroll
            for (int q = 0; q < kMaxSlabs; ++q) sl[k][q] = (live && q < nslab) ? slabs[(int64_t)q * stride + i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + threadIdx.y + kRL * k;
            if (r >= r1) continue;
            float t = v[k];
#pragma unroll
            for (int q = 0; q < kMaxSlabs; ++q)
                if (q < nslab) t += sl[k][q];
            if (r < bs)
                for (int q = kMaxSlabs; q < nslab; ++q) t += slabs[(int64_t)q * stride + (int64_t)r * L_p + col];
            DZ[(int64_t)r * L_p + col] = t;
            s += t;
        }
    }
    s = column_block_sum(s, red);
    if (threadIdx.y == 0 && col < L_p) part[(int64_t)blockIdx.y * L_p + col] = s;
}

// ---- D-Adapt-Adam (dadaptation==3.2 DAdaptAdam.step as Vamb configures it, encode.py:578) ----------
struct TensorDesc {
    const double* dsrc;   // if non-null the gradient is this fp64 accumulator (bias / gamma / beta of hidden layers)
    float dscale;         // 1 / world for accumulators that SyncBN turns into all-rank sums (applied only in a step that
                          // really all-reduced the statistics: the kernels' `allrank` argument), 1 otherwise
    const float* slab;    // gradient slabs; g[i] = sum_s slab[s*stride + i]
    int nslab;
    int64_t stride;
    int64_t p_off;        // offset of the tensor in the flat parameter / moment buffers
    int64_t size;         // padded element count actually used
};

// All parameter tensors of the model, passed BY VALUE in the kernel arguments (scalar loads, no
// dependent global lookups): tensor t owns workgroups [blk_start[t], blk_start[t+1]), 1024 elements each.
constexpr int kMaxOptTensors = 4 * 2 * 8 + 4;   // (W, b, gamma, beta) x (encoder + decoder) x 8 layers + mu + out
struct OptTable {
    int n;
    int blk_start[kMaxOptTensors + 1];
    TensorDesc d[kMaxOptTensors];
};

__device__ __forceinline__ int opt_find_tensor(const OptTable& tab, int blk) {
    int t = 0;
    while (t + 1 < tab.n && blk >= tab.blk_start[t + 1]) ++t;
    return t;
}

__device__ __forceinline__ float4 fetch_grad(const TensorDesc& td, int64_t local, int allrank = 0) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (td.dsrc) {
        const float sc = allrank ? td.dscale : 1.0f;
        g.x = (float)td.dsrc[local + 0] * sc; g.y = (float)td.dsrc[local + 1] * sc;
        g.z = (float)td.dsrc[local + 2] * sc; g.w = (float)td.dsrc[local + 3] * sc;
        return g;
    }
    // eight slab loads in flight per round (same ascending summation order as before)
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
    return g;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// data-parallel path: G[flat] = sum of this rank's gradient slabs (then all-reduced over the ranks)
__global__ __launch_bounds__(256) void vae_reduce_slabs_kernel(const OptTable tab, float* __restrict__ G, int allrank) {
    const int t = opt_find_tensor(tab, blockIdx.x);
    const TensorDesc& td = tab.d[t];
    const int64_t local = (int64_t)(blockIdx.x - tab.blk_start[t]) * 1024 + threadIdx.x * 4;
    if (local >= td.size) return;
    *reinterpret_cast<float4*>(G + td.p_off + local) = fetch_grad(td, local, allrank);
}

__global__ void vae_scale_kernel(float* __restrict__ v, int64_t n, float f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        v[i] *= f;
}

// One pass over every parameter: moments, s, parameter update, and the two global reductions
// (numerator dot and |s|_1) as per-workgroup partials.  Each workgroup covers 1024 elements of ONE tensor.
__global__ __launch_bounds__(256) void vae_dadapt_kernel(const OptTable tab, float* __restrict__ P,
                                                         float* __restrict__ M1, float* __restrict__ M2,
                                                         float* __restrict__ Sv, const StepState* __restrict__ st,
                                                         double* __restrict__ partials /*[all blocks][2]*/, int blk0,
                                                         float adam_lr, float gscale_fixed = 0.0f) {
    __shared__ double red[2][4];
    const int blk = blockIdx.x + blk0;   // the launch covers the table's workgroups [blk0, blk0 + gridDim.x)
    const int t = opt_find_tensor(tab, blk);
    const TensorDesc& td = tab.d[t];
    const int64_t local = (int64_t)(blk - tab.blk_start[t]) * 1024 + threadIdx.x * 4;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const double sqrt_b2d = sqrt(0.999);
    const double dlr = st->d;  // lr == 1
    // the loss' factor sum(w), see LossArgs::inv_b2 (gscale_fixed > 0: the gradient in the table is already complete)
    const float gscale = gscale_fixed > 0.0f ? gscale_fixed : (float)st->wsum;
    const float a_m = (float)(dlr * (1.0 - 0.9));
    const float a_s = (float)(dlr * (1.0 - sqrt_b2d));
    const float sqrt_b2 = (float)sqrt_b2d;
    const float one_m_b2 = (float)(1.0 - 0.999);
    float num = 0.f, sk = 0.f;
    if (local < td.size && adam_lr > 0.f) {
        // torch.optim.Adam(lr) with its defaults (semisupervised_encode.py:405: betas 0.9 / 0.999, eps 1e-8, no weight decay):
        //   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
        float4 g = fetch_grad(td, local);
        g.x *= gscale; g.y *= gscale; g.z *= gscale; g.w *= gscale;
        const int64_t o = td.p_off + local;
        float4 p = *reinterpret_cast<float4*>(P + o), m = *reinterpret_cast<float4*>(M1 + o),
               v = *reinterpret_cast<float4*>(M2 + o);
        const double t = (double)(st->k + 1);
        const float step_size = (float)((double)adam_lr / (1.0 - pow(0.9, t)));
        const float bc2_sqrt = (float)sqrt(1.0 - pow(0.999, t));
        float* pg = &g.x; float* pp = &p.x; float* pm = &m.x; float* pv = &v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = pg[e];
            pm[e] = pm[e] + (gi - pm[e]) * (1.0f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
            pv[e] = pv[e] * b2 + one_m_b2 * gi * gi;
            pp[e] -= step_size * (pm[e] / (sqrtf(pv[e]) / bc2_sqrt + eps));
        }
        *reinterpret_cast<float4*>(P + o) = p;
        *reinterpret_cast<float4*>(M1 + o) = m;
        *reinterpret_cast<float4*>(M2 + o) = v;
    } else if (local < td.size) {
        float4 g = fetch_grad(td, local);
        g.x *= gscale; g.y *= gscale; g.z *= gscale; g.w *= gscale;
        const int64_t o = td.p_off + local;
        float4 p = *reinterpret_cast<float4*>(P + o), m = *reinterpret_cast<float4*>(M1 + o),
               v = *reinterpret_cast<float4*>(M2 + o), s = *reinterpret_cast<float4*>(Sv + o);
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
    const double wn = wave_sum_f64((double)num), ws = wave_sum_f64((double)sk);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = wn; red[1][wave] = ws; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[(int64_t)blk * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[(int64_t)blk * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// scalar part of DAdaptAdam.step: numerator_weighted, d_hat, d, k
// ... and clears the hidden layers' fp64 batch accumulators for the next step (every reader has finished).
__global__ __launch_bounds__(256) void vae_dadapt_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                                  StepState* __restrict__ st,
                                                                  double* __restrict__ statbuf, int nstat, int adam) {
    for (int i = threadIdx.x; i < nstat; i += 256) statbuf[i] = 0.0;
    __shared__ double red[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += partials[(int64_t)i * 2 + 0];
        b += partials[(int64_t)i * 2 + 1];
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sqrt_b2 = sqrt(0.999);
        const double d = st->d;
        const double numerator_acum = d * red[0][0];  // dlr * sum of the per-tensor dots
        const double sk_l1 = red[1][0];
        const double nw = sqrt_b2 * st->numerator_weighted + (1.0 - sqrt_b2) * numerator_acum;
        if (adam) {
            st->k += 1;   // Adam's step count
        } else if (sk_l1 != 0.0) {
            const double d_hat = nw / ((1.0 - sqrt_b2) * sk_l1);
            st->d = d_hat > d ? d_hat : d;  // growth_rate = inf
            st->numerator_weighted = nw;
            st->k += 1;
        }
        st->step += 1;   // next step: fresh dropout / noise streams, next batch of the epoch's row list
        st->batch += 1;
    }
}

// forward-only calls advance the random streams too
__global__ void vae_advance_step_kernel(StepState* __restrict__ st) { st->step += 1; }

}  // namespace vh
