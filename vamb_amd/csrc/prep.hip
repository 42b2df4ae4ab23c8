// prep.hip -- the matrix passes of make_dataloader (vamb/encode.py:98-119, vamb/vambtools.py:250-288) on the device,
// writing straight into the resident feature matrix X[n][D_p] the VAE trains on.
//
// Bit-exactness against the reference's numpy code is by construction, not by tolerance: every reduction below
// reproduces numpy's summation ORDER in float32 (this file is compiled with -ffp-contract=off and correctly rounded
// division), and everything that is O(n) or O(columns) -- 1e6 / column sums, log + z-score of the total abundance,
// mean / std of a column from its sums, the contig weights -- stays in numpy on the host (vamb_amd/encode.py), fed
// with the vectors these kernels produce.
//   * a.sum(axis=0) of a C-contiguous [n][c] float32 array is out[j] += a[i][j] for i ascending (one sequential
//     chain per column): prep_column_sums_kernel.
//   * a.sum(axis=1) is numpy's pairwise summation of each row (8 interleaved accumulators per block of <= 128
//     elements, halves split at a multiple of 8, numpy/_core/src/umath/loops_utils.h.src): prep_rows_kernel runs the
//     split tree as a small postfix program built by the host (vamb_amd/encode.py:_pairwise_program; the same
//     program executed in numpy float32 is compared with numpy.sum on the CPU, tests/test_prep_host.py).
#include "dataset.hpp"

#include <algorithm>
#include <memory>

using namespace vh;

struct vh_prep {
    std::unique_ptr<vh_dataset> d;
    hipStream_t stream = nullptr;
    DevBuf<float> vec_a, vec_b;      // column vectors (scale / centre / mean / std / results)
    DevBuf<float> rows_a, rows_b;    // per-row vectors (totals, total abundance, weights)
    DevBuf<int> program;
    bool uploaded = false;
    ~vh_prep() {
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

constexpr int kColsPerBlock = 16;
constexpr int kTileRows = 512;
constexpr int kSumThreads = 256;

// Sequential (row-order) float32 sums of columns [c0, c0 + ncols) of M[n][ld]; with `centre` the summand is
// (x - centre[j])^2 rounded after the subtraction and after the product (numpy's _var: x = arr - mean; x *= x).
// One workgroup per 16 columns: all 256 threads stage 512-row tiles through LDS (coalesced 64-byte row segments,
// 32 loads in flight per thread), lanes 0..15 of the first wavefront add them in row order.
__global__ __launch_bounds__(kSumThreads) void prep_column_sums_kernel(const float* __restrict__ M, int64_t ld, int64_t n,
                                                                       int c0, int ncols,
                                                                       const float* __restrict__ centre,
                                                                       float* __restrict__ out) {
    __shared__ float tile[2][kTileRows][kColsPerBlock];
    const int tid = threadIdx.x;
    const int col = tid % kColsPerBlock, row0 = tid / kColsPerBlock;     // 16 rows per pass of the workgroup
    constexpr int kPasses = kTileRows / (kSumThreads / kColsPerBlock);   // 32
    const int c = blockIdx.x * kColsPerBlock + col;
    const bool col_ok = c < ncols;
    const float ctr = (centre != nullptr && col_ok) ? centre[c] : 0.0f;
    const float* src = M + c0 + (col_ok ? c : 0);
    const int64_t ntiles = (n + kTileRows - 1) / kTileRows;

    float v[kPasses];
    auto fetch = [&](int64_t t) {
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            const int64_t r = t * kTileRows + row0 + p * (kSumThreads / kColsPerBlock);
            v[p] = (col_ok && r < n) ? src[r * ld] : 0.0f;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            float x = v[p];
            if (centre != nullptr) {
                x = x - ctr;
                x = x * x;
            }
            tile[buf][row0 + p * (kSumThreads / kColsPerBlock)][col] = x;
        }
    };
    float acc = 0.0f;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int64_t t = 0; t < ntiles; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntiles) fetch(t + 1);
        if (tid < kColsPerBlock) {
            const int64_t rows = min((int64_t)kTileRows, n - t * kTileRows);
            int r = 0;
            if (t == 0) {   // numpy's add.reduce starts from its identity +0.0: a column of -0.0 sums to +0.0 (0.0f + -0.0f)
                acc = 0.0f + tile[buf][0][tid];
                r = 1;
            }
            // batches of 16 LDS reads in flight, then the 16 dependent adds (the order of the adds is the row order)
            for (; r + 16 <= (int)rows; r += 16) {
                float x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = tile[buf][r + i][tid];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = acc + x[i];
            }
            for (; r < (int)rows; ++r) acc = acc + tile[buf][r][tid];
        }
        if (t + 1 < ntiles) stage(buf ^ 1);
        __syncthreads();
    }
    if (tid < kColsPerBlock && blockIdx.x * kColsPerBlock + tid < ncols) out[blockIdx.x * kColsPerBlock + tid] = acc;
}

// Postfix program of numpy's pairwise row sum: {0, start, len} = leaf (push the sum of len elements from start),
// {1, 0, 0} = add (pop two, push the sum).
constexpr int kMaxStack = 24;
constexpr int kRowThreads = 256;

// Rows of the depth block: x = a * scale[j] (encode.py:105), total = pairwise sum of x (106), then
// x / total, or 1 / n_samples for rows whose total is zero (109-113).  Eight lanes per row hold numpy's eight
// interleaved accumulators; a wavefront works on eight rows.
__global__ __launch_bounds__(kRowThreads) void prep_rows_kernel(float* __restrict__ X, int64_t ld, int64_t n, int S,
                                                                const float* __restrict__ scale,
                                                                const int* __restrict__ program, int n_ops,
                                                                float uniform, float* __restrict__ totals) {
    __shared__ float stack[kMaxStack][kRowThreads];
    const int tid = threadIdx.x;
    const int j = tid & 7;
    const int64_t row = (int64_t)blockIdx.x * (kRowThreads / 8) + (tid >> 3);
    const bool ok = row < n;
    float* a = X + (ok ? row : 0) * ld;
    int sp = 0;
    for (int op = 0; op < n_ops; ++op) {
        const int kind = program[3 * op], start = program[3 * op + 1], len = program[3 * op + 2];
        if (kind == 1) {
            const float right = stack[sp - 1][tid], left = stack[sp - 2][tid];
            stack[sp - 2][tid] = left + right;
            --sp;
            continue;
        }
        float res;
        if (len < 8) {
            res = 0.0f;
            for (int i = 0; i < len; ++i) res = res + a[start + i] * scale[start + i];
        } else {
            const int body = len - (len & 7);
            float r = a[start + j] * scale[start + j];
#pragma unroll 4
            for (int i = 8; i < body; i += 8) r = r + a[start + i + j] * scale[start + i + j];
            // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); float addition commutes, so the butterfly leaves the
            // same bits in all eight lanes
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            r = r + __shfl_xor(r, 4);
            res = r;
            for (int i = body; i < len; ++i) res = res + a[start + i] * scale[start + i];
        }
        stack[sp][tid] = res;
        ++sp;
    }
    // numpy's reduction adds the pairwise sum to the identity: 0 + x (exact, but it turns -0.0 into +0.0)
    const float total = 0.0f + stack[0][tid];
    if (ok && j == 0) totals[row] = total;
    if (!ok) return;
    const bool zero = total == 0.0f;
    const float div = zero ? 1.0f : total;
    for (int k = j; k < S; k += 8) {
        const float x = zero ? uniform : a[k] * scale[k];
        a[k] = x / div;
    }
}

// z-score of the TNF block with the column means / standard deviations the host derived from the column sums
// (vambtools.py:283-284: array -= mean; array /= std)
__global__ void prep_zscore_kernel(float* __restrict__ X, int64_t ld, int64_t n, int c0, int ncols,
                                   const float* __restrict__ mean, const float* __restrict__ stdev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ncols) return;
    const int64_t r = i / ncols;
    const int c = (int)(i - r * ncols);
    float* p = X + r * ld + c0 + c;
    const float x = *p - mean[c];
    *p = x / stdev[c];
}

__global__ void prep_set_rows_kernel(float* __restrict__ X, int64_t ld, int64_t n, int col,
                                     const float* __restrict__ total_abundance, const float* __restrict__ weights,
                                     float* __restrict__ w) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    X[r * ld + col] = total_abundance[r];
    w[r] = weights[r];
}

}  // namespace

extern "C" {

int vh_prep_create(int64_t n, int nsamples, vh_prep** out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr, "NULL argument");
        VH_REQUIRE(n >= 1, "empty dataset");
        VH_REQUIRE(nsamples >= 1 && nsamples <= (1 << 20), "nsamples must be in [1, 2^20]");
        std::unique_ptr<vh_prep> p(new vh_prep());
        p->d.reset(new vh_dataset());
        vh_dataset* d = p->d.get();
        d->n = n;
        d->S = nsamples;
        d->D_p = (int)round_up(nsamples + VH_NTNF + 1, kDatasetColPad);
        d->X.alloc((size_t)n * d->D_p);
        d->w.alloc((size_t)n);
        VH_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        VH_HIP(hipMemsetAsync(d->X.p, 0, sizeof(float) * (size_t)n * d->D_p, p->stream));   // the padding columns stay zero
        p->vec_a.alloc((size_t)std::max(nsamples, VH_NTNF));
        p->vec_b.alloc((size_t)std::max(nsamples, VH_NTNF));
        p->rows_a.alloc((size_t)n);
        p->rows_b.alloc((size_t)n);
        VH_HIP(hipStreamSynchronize(p->stream));
        *out = p.release();
    });
}

int vh_prep_destroy(vh_prep* p) {
    delete p;
    return VH_OK;
}

int vh_prep_upload(vh_prep* p, const float* abundance, const float* tnf) {
    return guarded([&] {
        VH_REQUIRE(p != nullptr && p->d && abundance != nullptr && tnf != nullptr, "NULL argument");
        vh_dataset* d = p->d.get();
        const size_t dpitch = sizeof(float) * (size_t)d->D_p;
        VH_HIP(hipMemcpy2DAsync(d->X.p, dpitch, abundance, sizeof(float) * (size_t)d->S, sizeof(float) * (size_t)d->S,
                                (size_t)d->n, hipMemcpyHostToDevice, p->stream));
        VH_HIP(hipMemcpy2DAsync(d->X.p + d->S, dpitch, tnf, sizeof(float) * VH_NTNF, sizeof(float) * VH_NTNF,
                                (size_t)d->n, hipMemcpyHostToDevice, p->stream));
        VH_HIP(hipStreamSynchronize(p->stream));
        p->uploaded = true;
    });
}

int vh_prep_column_sums(vh_prep* p, int block, const float* centre, float* out) {
    return guarded([&] {
        VH_REQUIRE(p != nullptr && p->d && out != nullptr, "NULL argument");
        VH_REQUIRE(p->uploaded, "vh_prep_upload has not been called");
        VH_REQUIRE(block == 0 || block == 1, "block must be 0 (depths) or 1 (tnf)");
        vh_dataset* d = p->d.get();
        const int c0 = block == 0 ? 0 : d->S, ncols = block == 0 ? d->S : VH_NTNF;
        if (centre)
            VH_HIP(hipMemcpyAsync(p->vec_a.p, centre, sizeof(float) * ncols, hipMemcpyHostToDevice, p->stream));
        hipLaunchKernelGGL(prep_column_sums_kernel, dim3((unsigned)ceil_div(ncols, kColsPerBlock)), dim3(kSumThreads), 0,
                           p->stream, d->X.p, (int64_t)d->D_p, d->n, c0, ncols, centre ? p->vec_a.p : nullptr, p->vec_b.p);
        VH_HIP(hipGetLastError());
        VH_HIP(hipMemcpyAsync(out, p->vec_b.p, sizeof(float) * ncols, hipMemcpyDeviceToHost, p->stream));
        VH_HIP(hipStreamSynchronize(p->stream));
    });
}

int vh_prep_normalise_rows(vh_prep* p, const float* scale, const int32_t* program, int n_ops, float uniform,
                           float* totals) {
    return guarded([&] {
        VH_REQUIRE(p != nullptr && p->d && scale != nullptr && program != nullptr && totals != nullptr, "NULL argument");
        VH_REQUIRE(p->uploaded, "vh_prep_upload has not been called");
        vh_dataset* d = p->d.get();
        // validate the program: every leaf inside the row, the stack never deeper than kMaxStack, one value left
        int sp = 0;
        VH_REQUIRE(n_ops >= 1 && n_ops <= (1 << 20), "bad pairwise program");
        for (int i = 0; i < n_ops; ++i) {
            const int kind = program[3 * i], start = program[3 * i + 1], len = program[3 * i + 2];
            if (kind == 0) {
                VH_REQUIRE(start >= 0 && len >= 0 && start + len <= d->S, "pairwise program: leaf outside the row");
                ++sp;
                VH_REQUIRE(sp <= kMaxStack, "pairwise program: stack deeper than %d", kMaxStack);
            } else {
                VH_REQUIRE(kind == 1 && sp >= 2, "pairwise program: bad operation");
                --sp;
            }
        }
        VH_REQUIRE(sp == 1, "pairwise program leaves %d values", sp);
        p->program.ensure((size_t)3 * n_ops);
        VH_HIP(hipMemcpyAsync(p->program.p, program, sizeof(int) * 3 * (size_t)n_ops, hipMemcpyHostToDevice, p->stream));
        VH_HIP(hipMemcpyAsync(p->vec_a.p, scale, sizeof(float) * d->S, hipMemcpyHostToDevice, p->stream));
        hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)ceil_div(d->n, kRowThreads / 8)), dim3(kRowThreads), 0, p->stream,
                           d->X.p, (int64_t)d->D_p, d->n, d->S, p->vec_a.p, p->program.p, n_ops, uniform, p->rows_a.p);
        VH_HIP(hipGetLastError());
        VH_HIP(hipMemcpyAsync(totals, p->rows_a.p, sizeof(float) * (size_t)d->n, hipMemcpyDeviceToHost, p->stream));
        VH_HIP(hipStreamSynchronize(p->stream));
    });
}

int vh_prep_zscore_tnf(vh_prep* p, const float* mean, const float* stdev) {
    return guarded([&] {
        VH_REQUIRE(p != nullptr && p->d && mean != nullptr && stdev != nullptr, "NULL argument");
        VH_REQUIRE(p->uploaded, "vh_prep_upload has not been called");
        vh_dataset* d = p->d.get();
        VH_HIP(hipMemcpyAsync(p->vec_a.p, mean, sizeof(float) * VH_NTNF, hipMemcpyHostToDevice, p->stream));
        VH_HIP(hipMemcpyAsync(p->vec_b.p, stdev, sizeof(float) * VH_NTNF, hipMemcpyHostToDevice, p->stream));
        const int64_t total = d->n * VH_NTNF;
        hipLaunchKernelGGL(prep_zscore_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, p->stream, d->X.p,
                           (int64_t)d->D_p, d->n, d->S, VH_NTNF, p->vec_a.p, p->vec_b.p);
        VH_HIP(hipGetLastError());
        VH_HIP(hipStreamSynchronize(p->stream));
    });
}

int vh_prep_finish(vh_prep* p, const float* total_abundance, const float* weights, vh_dataset** out) {
    return guarded([&] {
        VH_REQUIRE(p != nullptr && p->d && total_abundance != nullptr && weights != nullptr && out != nullptr, "NULL argument");
        VH_REQUIRE(p->uploaded, "vh_prep_upload has not been called");
        vh_dataset* d = p->d.get();
        VH_HIP(hipMemcpyAsync(p->rows_a.p, total_abundance, sizeof(float) * (size_t)d->n, hipMemcpyHostToDevice, p->stream));
        VH_HIP(hipMemcpyAsync(p->rows_b.p, weights, sizeof(float) * (size_t)d->n, hipMemcpyHostToDevice, p->stream));
        hipLaunchKernelGGL(prep_set_rows_kernel, dim3((unsigned)ceil_div(d->n, 256)), dim3(256), 0, p->stream, d->X.p,
                           (int64_t)d->D_p, d->n, d->S + VH_NTNF, p->rows_a.p, p->rows_b.p, d->w.p);
        VH_HIP(hipGetLastError());
        VH_HIP(hipStreamSynchronize(p->stream));
        *out = p->d.release();
    });
}

int vh_dataset_shape(vh_dataset* d, int64_t* n, int* nsamples) {
    return guarded([&] {
        VH_REQUIRE(d != nullptr, "NULL argument");
        if (n) *n = d->n;
        if (nsamples) *nsamples = d->S;
    });
}

int vh_dataset_download(vh_dataset* d, float* depths, float* tnf, float* total_abundance, float* weights) {
    return guarded([&] {
        VH_REQUIRE(d != nullptr, "NULL argument");
        const size_t spitch = sizeof(float) * (size_t)d->D_p;
        if (depths)
            VH_HIP(hipMemcpy2D(depths, sizeof(float) * (size_t)d->S, d->X.p, spitch, sizeof(float) * (size_t)d->S, (size_t)d->n,
                               hipMemcpyDeviceToHost));
        if (tnf)
            VH_HIP(hipMemcpy2D(tnf, sizeof(float) * VH_NTNF, d->X.p + d->S, spitch, sizeof(float) * VH_NTNF, (size_t)d->n,
                               hipMemcpyDeviceToHost));
        if (total_abundance)
            VH_HIP(hipMemcpy2D(total_abundance, sizeof(float), d->X.p + d->S + VH_NTNF, spitch, sizeof(float), (size_t)d->n,
                               hipMemcpyDeviceToHost));
        if (weights) VH_HIP(hipMemcpy(weights, d->w.p, sizeof(float) * (size_t)d->n, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
