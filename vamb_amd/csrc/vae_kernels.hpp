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

// ---- batch assembly ----------------------------------------------------------------------------
// Xb[r][:] = X[idx[r]][:], Wb[r] = w[idx[r]] for r < bs; zero rows for the padding.  blockDim (64,4).
// idx (if given) is the epoch's row list; the batch to use is *batch_ptr (device resident, advanced by
// the optimiser's finalize kernel, so that one captured graph serves every step of the epoch).
__global__ void vae_gather_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ w_all,
                                  const int64_t* __restrict__ idx, const long long* __restrict__ batch_ptr, int bs,
                                  int bs_p, float* __restrict__ Xb, float* __restrict__ Wb) {
    const int r = blockIdx.x * 4 + threadIdx.y;
    if (r >= bs_p) return;
    const bool real = r < bs;
    const int64_t first = batch_ptr ? (int64_t)(*batch_ptr) * bs : 0;
    const int64_t src = real ? (idx ? idx[first + r] : (int64_t)r) : 0;
    const float4* s = reinterpret_cast<const float4*>(X + src * ldx);
    float4* d = reinterpret_cast<float4*>(Xb + (int64_t)r * ldx);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const int dq = (int)(ldx / 4);
    for (int c = threadIdx.x; c < dq; c += 64) {
        float4 v = z;
        if (real) v = s[c];
        d[c] = v;
    }
    if (threadIdx.x == 0) Wb[r] = real ? w_all[src] : 0.f;
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

// ---- BatchNorm1d forward (training): finalise statistics from the GEMM epilogue's partial sums --
// blockDim (16, 16): 16 columns per workgroup, 16 lanes share the nb partials of each column
// (fixed assignment and fixed combination order => deterministic).
__global__ __launch_bounds__(256) void vae_bn_finalize_kernel(
    const float* __restrict__ part, int nb, int ld, int n_p, int bs, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rm, float* __restrict__ rv, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double r1[16][17], r2[16][17];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = blockIdx.x * 16 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (col < n_p)
        for (int b = ty; b < nb; b += 16) {
            s1 += (double)part[((int64_t)b * 2 + 0) * ld + col];
            s2 += (double)part[((int64_t)b * 2 + 1) * ld + col];
        }
    r1[ty][tx] = s1;
    r2[ty][tx] = s2;
    __syncthreads();
    if (ty != 0 || col >= n_p) return;
    s1 = 0.0; s2 = 0.0;
    for (int i = 0; i < 16; ++i) { s1 += r1[i][tx]; s2 += r2[i][tx]; }
    const double m = s1 / bs;
    double var = s2 / bs - m * m;  // biased variance (normalisation)
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)kBnEps));
    const float sc = gamma[col] * istd;
    mean[col] = (float)m;
    invstd[col] = istd;
    scale[col] = sc;
    shift[col] = beta[col] - (float)m * sc;
    const double unbiased = bs > 1 ? var * ((double)bs / (double)(bs - 1)) : var;
    rm[col] = (1.0f - kBnMomentum) * rm[col] + kBnMomentum * (float)m;
    rv[col] = (1.0f - kBnMomentum) * rv[col] + kBnMomentum * (float)unbiased;
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

// A = H * scale[col] + shift[col]
__global__ void vae_bn_apply_kernel(const float* __restrict__ H, float* __restrict__ A, int64_t total4, int n_p,
                                    const float* __restrict__ scale, const float* __restrict__ shift) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)((i * 4) % n_p);
        const float4 h = reinterpret_cast<const float4*>(H)[i];
        const float4 s = *reinterpret_cast<const float4*>(scale + col);
        const float4 t = *reinterpret_cast<const float4*>(shift + col);
        float4 a;
        a.x = h.x * s.x + t.x;
        a.y = h.y * s.y + t.y;
        a.z = h.z * s.z + t.z;
        a.w = h.w * s.w + t.w;
        reinterpret_cast<float4*>(A)[i] = a;
    }
}

// ---- reparameterisation (encode.py:276-286) ------------------------------------------------------
// standard-normal noise: Box-Muller over the counter-based hash (free-running mode)
__device__ __forceinline__ float hash_randn(uint64_t key, uint64_t id) {
    const float u1 = ((float)hash32(key, 2 * id) + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)hash32(key, 2 * id + 1) * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

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
    for (int s = 0; s < nslab; ++s) m += slabs[(int64_t)s * stride + i];
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
};

__global__ __launch_bounds__(256) void vae_loss_kernel(const LossArgs a) {
    __shared__ float red[4][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    float ab_t = 0.f, ce_t = 0.f, sse_t = 0.f, kld_t = 0.f;
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
            // softmax over the S abundance logits
            float mx = -3.0e38f;
            for (int c = lane; c < S; c += 64) mx = fmaxf(mx, r[c]);
            mx = wave_max(mx);
            float se = 0.f;
            for (int c = lane; c < S; c += 64) se += expf(r[c] - mx);
            se = wave_sum(se);
            const float inv = 1.0f / se;
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
            for (int c = S + lane; c < S + 103; c += 64) {
                const float diff = r[c] - x[c];
                sse += diff * diff;
                dr[c] = gsse * diff;
            }
            sse = wave_sum(sse);
            // total-abundance squared error (one column) + zero the padding columns
            float ab = 0.f;
            if (lane == 0) {
                const int c = S + 103;
                const float diff = r[c] - x[c];
                ab = diff * diff;
                dr[c] = g * a.ab_w * 2.0f * diff;
            }
            ab = wave_sum(ab);
            for (int c = S + 104 + lane; c < a.ld; c += 64) dr[c] = 0.f;
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
    if (lane == 0) { red[wave][0] = ab_t; red[wave][1] = ce_t; red[wave][2] = sse_t; red[wave][3] = kld_t; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int t = threadIdx.x;
        a.part[(int64_t)blockIdx.x * 4 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
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
};

// reduce the loss partials, produce the five means (encode.py:350-356), add them to the epoch sums and
// publish the batch weight sum: sum(Wb) on one GPU, gwsum[batch] (planned on the host) under data
// parallelism.  bs_global: rows of the whole (all-rank) batch; every rank adds its own share
// local_sum / bs_global and the epoch sums are all-reduced once per epoch.
__global__ __launch_bounds__(256) void vae_loss_finalize_kernel(const float* __restrict__ part, int nblocks,
                                                                const float* __restrict__ Wb, int bs,
                                                                const float* __restrict__ gwsum, int bs_global,
                                                                StepState* __restrict__ st) {
    __shared__ double red[5][256];
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int t = 0; t < 4; ++t) s[t] += (double)part[(int64_t)b * 4 + t];
    if (!gwsum)
        for (int i = threadIdx.x; i < bs; i += 256) s[4] += (double)Wb[i];
    for (int t = 0; t < 5; ++t) red[t][threadIdx.x] = s[t];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off)
            for (int t = 0; t < 5; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double bsg = (double)bs_global;
        const double wsum = gwsum ? (double)gwsum[st->batch] : (double)(float)red[4][0];
        const double ab = red[0][0] / bsg, ce = red[1][0] / bsg, sse = red[2][0] / bsg, kld = red[3][0] / bsg;
        const double wmean = wsum / bsg;
        const double loss = ((ce + ab + sse) + kld) * wmean;
        const double v[5] = {loss, ab, ce, sse, kld};
        for (int t = 0; t < 5; ++t) { st->step_loss[t] = v[t]; st->epoch_loss[t] += v[t]; }
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

// BatchNorm backward, pass 1: partial sums of dA and dA * xhat
__global__ __launch_bounds__(256) void vae_bn_bwd_reduce_kernel(const float* __restrict__ DA,
                                                                const float* __restrict__ H, int n_p, int bs,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                float* __restrict__ part /*[nrb][2][n_p]*/) {
    __shared__ float red[kRL][kCT + 1];
    const int col = blockIdx.x * kCT + threadIdx.x;
    const int r0 = blockIdx.y * kRB, r1 = min(bs, r0 + kRB);
    float s1 = 0.f, s2 = 0.f;
    if (col < n_p) {
        const float m = mean[col], is = invstd[col];
        for (int r = r0 + threadIdx.y; r < r1; r += kRL) {
            const float da = DA[(int64_t)r * n_p + col];
            const float xh = (H[(int64_t)r * n_p + col] - m) * is;
            s1 += da;
            s2 += da * xh;
        }
    }
    s1 = column_block_sum(s1, red);
    s2 = column_block_sum(s2, red);
    if (threadIdx.y == 0 && col < n_p) {
        part[((int64_t)blockIdx.y * 2 + 0) * n_p + col] = s1;
        part[((int64_t)blockIdx.y * 2 + 1) * n_p + col] = s2;
    }
}

// pass 2: totals; these are also the gradients of beta (S1) and gamma (S2).  blockDim (16, 16).
__global__ __launch_bounds__(256) void vae_bn_bwd_finalize_kernel(const float* __restrict__ part, int nrb, int n_p,
                                                                  float* __restrict__ S12,
                                                                  float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta) {
    __shared__ double r1[16][17], r2[16][17];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = blockIdx.x * 16 + tx;
    double s1 = 0.0, s2 = 0.0;
    if (col < n_p)
        for (int b = ty; b < nrb; b += 16) {
            s1 += (double)part[((int64_t)b * 2 + 0) * n_p + col];
            s2 += (double)part[((int64_t)b * 2 + 1) * n_p + col];
        }
    r1[ty][tx] = s1;
    r2[ty][tx] = s2;
    __syncthreads();
    if (ty != 0 || col >= n_p) return;
    s1 = 0.0; s2 = 0.0;
    for (int i = 0; i < 16; ++i) { s1 += r1[i][tx]; s2 += r2[i][tx]; }
    S12[col] = (float)s1;
    S12[n_p + col] = (float)s2;
    dbeta[col] = (float)s1;
    dgamma[col] = (float)s2;
}

struct BnBwdArgs {
    const float* DA;
    const float* H;
    float* DZ;
    int n_p, bs, bs_p;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* S12;
    float drop_scale;
    uint32_t drop_thresh;
    uint64_t drop_key;
    const unsigned long long* step_ptr;
    const uint8_t* drop_mask;
    int64_t ld_mask;
    float* dbias_part;  // [nrb][n_p]
};

// pass 3: dZ = BN'(dA) * dropout' * leaky_relu'  and the per-workgroup column sums of dZ (bias grads)
__global__ __launch_bounds__(256) void vae_bn_bwd_apply_kernel(const BnBwdArgs a) {
    __shared__ float red[kRL][kCT + 1];
    const int col = blockIdx.x * kCT + threadIdx.x;
    const int r0 = blockIdx.y * kRB, r1 = min(a.bs_p, r0 + kRB);
    float s = 0.f;
    if (col < a.n_p) {
        const float m = a.mean[col], is = a.invstd[col], gm = a.gamma[col];
        const float c1 = a.S12[col] / (float)a.bs, c2 = a.S12[a.n_p + col] / (float)a.bs;
        const bool use_drop = (a.drop_scale != 1.0f) || (a.drop_mask != nullptr);
        const uint64_t key = step_key(a.drop_key, a.step_ptr);
        for (int r = r0 + threadIdx.y; r < r1; r += kRL) {
            const int64_t i = (int64_t)r * a.n_p + col;
            float dz = 0.f;
            if (r < a.bs) {
                const float h = a.H[i];
                const float xh = (h - m) * is;
                float dh = is * gm * (a.DA[i] - c1 - xh * c2);
                bool keep = true;
                if (use_drop) {
                    keep = a.drop_mask
                               ? (a.drop_mask[(int64_t)r * a.ld_mask + col] != 0)
                               : (hash32(key, (uint64_t)r * (uint64_t)a.n_p + (uint64_t)col) >= a.drop_thresh);
                    dh *= a.drop_scale;
                }
                dz = keep ? dh * (h > 0.f ? 1.0f : kLeakySlope) : 0.f;
            }
            a.DZ[i] = dz;
            s += dz;
        }
    }
    s = column_block_sum(s, red);
    if (threadIdx.y == 0 && col < a.n_p) a.dbias_part[(int64_t)blockIdx.y * a.n_p + col] = s;
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
    if (col < L_p)
        for (int r = r0 + threadIdx.y; r < r1; r += kRL) {
            const int64_t i = (int64_t)r * L_p + col;
            float v = 0.f;
            if (r < bs) {
                v = dMUk[i];
                for (int k = 0; k < nslab; ++k) v += slabs[(int64_t)k * stride + i];
            }
            DZ[i] = v;
            s += v;
        }
    s = column_block_sum(s, red);
    if (threadIdx.y == 0 && col < L_p) part[(int64_t)blockIdx.y * L_p + col] = s;
}

// ---- D-Adapt-Adam (dadaptation==3.2 DAdaptAdam.step as Vamb configures it, encode.py:578) ----------
struct TensorDesc {
    const float* slab;    // gradient slabs; g[i] = sum_s slab[s*stride + i]
    int nslab;
    int64_t stride;
    int64_t p_off;        // offset of the tensor in the flat parameter / moment buffers
    int64_t size;         // padded element count actually used
};

// data-parallel path: G[flat] = sum of this rank's gradient slabs (then all-reduced over the ranks)
__global__ __launch_bounds__(256) void vae_reduce_slabs_kernel(const TensorDesc* __restrict__ descs,
                                                               const int* __restrict__ blk_tensor,
                                                               const int* __restrict__ blk_local,
                                                               float* __restrict__ G) {
    const TensorDesc td = descs[blk_tensor[blockIdx.x]];
    const int64_t local = (int64_t)blk_local[blockIdx.x] * 1024 + threadIdx.x * 4;
    if (local >= td.size) return;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < td.nslab; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(td.slab + (int64_t)s * td.stride + local);
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
    }
    *reinterpret_cast<float4*>(G + td.p_off + local) = g;
}

__global__ void vae_scale_kernel(float* __restrict__ v, int64_t n, float f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        v[i] *= f;
}

// One pass over every parameter: moments, s, parameter update, and the two global reductions
// (numerator dot and |s|_1) as per-workgroup partials.  Each workgroup covers 1024 elements of ONE tensor.
__global__ __launch_bounds__(256) void vae_dadapt_kernel(const TensorDesc* __restrict__ descs,
                                                         const int* __restrict__ blk_tensor,
                                                         const int* __restrict__ blk_local,
                                                         float* __restrict__ P, float* __restrict__ M1,
                                                         float* __restrict__ M2, float* __restrict__ Sv,
                                                         const StepState* __restrict__ st,
                                                         double* __restrict__ partials /*[gridDim.x][2]*/) {
    __shared__ double red[2][256];
    const TensorDesc td = descs[blk_tensor[blockIdx.x]];
    const int64_t local = (int64_t)blk_local[blockIdx.x] * 1024 + threadIdx.x * 4;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const double sqrt_b2d = sqrt(0.999);
    const double dlr = st->d;  // lr == 1
    const float gscale = (float)st->wsum;  // the loss' factor sum(w), see LossArgs::inv_b2
    const float a_m = (float)(dlr * (1.0 - 0.9));
    const float a_s = (float)(dlr * (1.0 - sqrt_b2d));
    const float sqrt_b2 = (float)sqrt_b2d;
    const float one_m_b2 = (float)(1.0 - 0.999);
    float num = 0.f, sk = 0.f;
    if (local < td.size) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < td.nslab; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(td.slab + (int64_t)s * td.stride + local);
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
        }
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
    red[0][threadIdx.x] = (double)num;
    red[1][threadIdx.x] = (double)sk;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[(int64_t)blockIdx.x * 2 + 0] = red[0][0];
        partials[(int64_t)blockIdx.x * 2 + 1] = red[1][0];
    }
}

// scalar part of DAdaptAdam.step: numerator_weighted, d_hat, d, k
__global__ __launch_bounds__(256) void vae_dadapt_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                                  StepState* __restrict__ st) {
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
        if (sk_l1 != 0.0) {
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
