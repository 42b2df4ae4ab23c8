// tnf.hip -- tetranucleotide frequencies on the device: the step immediately upstream of
// make_dataloader.
//   * vh_tnf_kmercounts: vambcore.kmercounts (vamb/vambtools.py:444-447; Rust, not in the reference tree) for a batch of
//     sequences.  Semantics pinned by the reference's own test (test/test_vambtools.py:137-151): a 4-mer counts iff all
//     four bytes are A, C, G or T in either case, index = base-4 number with A, C, G, T = 0..3, first base most
//     significant.  Every other byte -- including U, which the error text at vamb/parsecontigs.py:194-199 mentions but the
//     pinned definition does not count (the Rust source is absent: UNPINNED either way) -- voids the 4-mers it is part of.
//     Integer work: bit-exact.
//   * vh_tnf_project: Composition._project (vamb/parsecontigs.py:140-150) -- row sums (numpy's pairwise order), 1 / s,
//     scale, - 1/256, then the [n x 256] . [256 x 103] projection -- and mask_lower_bits(., 12) (parsecontigs.py:211).
//     Everything before the projection is bit-identical to numpy (this file is compiled with -ffp-contract=off); the
//     projection itself is numpy.dot = BLAS sgemm in the reference, whose summation order is unknowable, so it is held to
//     a float32 tolerance: here v_mfma_f32_32x32x2_f32, exact products accumulated in ascending k.
#include "common.hpp"

#include <algorithm>
#include <memory>

using namespace vh;

struct vh_tnf {
    hipStream_t stream = nullptr;
    DevBuf<float> kernel_p;       // [256][128] zero-padded projection kernel
    DevBuf<uint8_t> bases;
    DevBuf<int64_t> offsets;
    DevBuf<uint32_t> counts;      // [n][256] of the last kmercounts call
    DevBuf<float> fourmers, tnf;
    int64_t n_counts = 0;
    ~vh_tnf() {
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

constexpr int kKmers = 256;
constexpr int kTnf = VH_NTNF;      // 103
constexpr int kTnfPad = 128;

__device__ __forceinline__ unsigned int base_code(unsigned int c) {
    c |= 0x20u;   // lower case
    return c == 'a' ? 0u : (c == 'c' ? 1u : (c == 'g' ? 2u : (c == 't' ? 3u : 4u)));
}

// one workgroup per sequence (grid-stride), a 256-bin LDS histogram
__global__ __launch_bounds__(256) void tnf_kmercounts_kernel(const uint8_t* __restrict__ bases,
                                                             const int64_t* __restrict__ offsets, int64_t n,
                                                             uint32_t* __restrict__ counts) {
    __shared__ unsigned int hist[kKmers];
    for (int64_t s = blockIdx.x; s < n; s += gridDim.x) {
        hist[threadIdx.x] = 0u;
        __syncthreads();
        const int64_t lo = offsets[s], len = offsets[s + 1] - lo;
        const uint8_t* seq = bases + lo;
        for (int64_t i = threadIdx.x; i + 3 < len; i += 256) {
            const unsigned int b0 = base_code(seq[i]), b1 = base_code(seq[i + 1]), b2 = base_code(seq[i + 2]),
                               b3 = base_code(seq[i + 3]);
            if ((b0 | b1 | b2 | b3) < 4u) atomicAdd(&hist[(b0 << 6) | (b1 << 4) | (b2 << 2) | b3], 1u);
        }
        __syncthreads();
        counts[s * kKmers + threadIdx.x] = hist[threadIdx.x];
        __syncthreads();
    }
}

// fourmers (uint32 counts or float32) -> float32 row-normalised: s = row sum in numpy's pairwise order (two blocks of 128,
// eight interleaved accumulators each), s == 0 -> 1, x * (1 / s) + -(1 / 256).  Eight lanes per row.
template <bool FROM_COUNTS>
__global__ __launch_bounds__(256) void tnf_normalise_kernel(const void* src, int64_t n, float* out) {   // src may alias out (in place)
    const int tid = threadIdx.x, j = tid & 7;
    const int64_t row = (int64_t)blockIdx.x * 32 + (tid >> 3);
    const bool ok = row < n;
    const int64_t r = ok ? row : 0;
    auto at = [&](int k) -> float {
        return FROM_COUNTS ? (float)reinterpret_cast<const uint32_t*>(src)[r * kKmers + k]
                           : reinterpret_cast<const float*>(src)[r * kKmers + k];
    };
    float half[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float acc = at(128 * b + j);
#pragma unroll
        for (int i = 8; i < 128; i += 8) acc = acc + at(128 * b + i + j);
        acc = acc + __shfl_xor(acc, 1);
        acc = acc + __shfl_xor(acc, 2);
        acc = acc + __shfl_xor(acc, 4);
        half[b] = acc;
    }
    float s = 0.0f + (half[0] + half[1]);
    if (s == 0.0f) s = 1.0f;
    const float inv = 1.0f / s;
    if (!ok) return;
    for (int k = j; k < kKmers; k += 8) out[row * kKmers + k] = at(k) * inv + (-0.00390625f);
}

typedef __attribute__((ext_vector_type(16))) float tnf_f32x16;

// C[n][103] = F[n][256] . K[256][103] on v_mfma_f32_32x32x2_f32, low `mask_bits` mantissa bits cleared.  A workgroup owns
// 64 rows x 128 (padded) columns; wavefront w: rows 32 (w & 1), column tiles 2 (w >> 1) and 2 (w >> 1) + 1.
__global__ __launch_bounds__(256) void tnf_project_kernel(const float* __restrict__ F, int64_t n,
                                                          const float* __restrict__ Kp, int mask_bits,
                                                          float* __restrict__ C) {
    __shared__ float As[64][33];
    __shared__ float Bs[32][kTnfPad + 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int wr = (w & 1) * 32, wc = (w >> 1) * 64;
    const int fr = lane & 31, fh = lane >> 5;
    tnf_f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    for (int k0 = 0; k0 < kKmers; k0 += 32) {
        {   // A chunk: 64 rows x 32 k, 8 consecutive k per thread
            const int r = tid >> 2, kq = (tid & 3) * 8;
            const int64_t row = row0 + r;
            float4 v0 = make_float4(0, 0, 0, 0), v1 = v0;
            if (row < n) {
                v0 = *reinterpret_cast<const float4*>(F + row * kKmers + k0 + kq);
                v1 = *reinterpret_cast<const float4*>(F + row * kKmers + k0 + kq + 4);
            }
            As[r][kq + 0] = v0.x; As[r][kq + 1] = v0.y; As[r][kq + 2] = v0.z; As[r][kq + 3] = v0.w;
            As[r][kq + 4] = v1.x; As[r][kq + 5] = v1.y; As[r][kq + 6] = v1.z; As[r][kq + 7] = v1.w;
        }
        {   // B chunk: 32 k x 128 columns, 16 consecutive columns per thread
            const int k = tid >> 3, c = (tid & 7) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(Kp + (int64_t)(k0 + k) * kTnfPad + c + 4 * q);
                Bs[k][c + 4 * q + 0] = v.x; Bs[k][c + 4 * q + 1] = v.y; Bs[k][c + 4 * q + 2] = v.z; Bs[k][c + 4 * q + 3] = v.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) {   // ascending k: the defined accumulation order
            const float a = As[wr + fr][2 * s + fh];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[2 * s + fh][wc + fr], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[2 * s + fh][wc + 32 + fr], acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    const uint32_t keep = mask_bits > 0 ? ~((1u << mask_bits) - 1u) : 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = wc + 32 * t + fr;
        if (col >= kTnf) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t row = row0 + wr + (reg & 3) + 8 * (reg >> 2) + 4 * fh;
            if (row < n) {
                const float v = t == 0 ? acc0[reg] : acc1[reg];
                C[row * kTnf + col] = __uint_as_float(__float_as_uint(v) & keep);
            }
        }
    }
}

}  // namespace

extern "C" {

int vh_tnf_create(const float* kernel, vh_tnf** out) {
    return guarded([&] {
        VH_REQUIRE(kernel != nullptr && out != nullptr, "NULL argument");
        std::unique_ptr<vh_tnf> t(new vh_tnf());
        VH_HIP(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
        std::vector<float> padded((size_t)kKmers * kTnfPad, 0.0f);
        for (int k = 0; k < kKmers; ++k) memcpy(padded.data() + (size_t)k * kTnfPad, kernel + (size_t)k * kTnf, sizeof(float) * kTnf);
        t->kernel_p.alloc(padded.size());
        VH_HIP(hipMemcpy(t->kernel_p.p, padded.data(), sizeof(float) * padded.size(), hipMemcpyHostToDevice));
        *out = t.release();
    });
}

int vh_tnf_destroy(vh_tnf* t) {
    delete t;
    return VH_OK;
}

int vh_tnf_kmercounts(vh_tnf* t, const uint8_t* bases, const int64_t* offsets, int64_t n, uint32_t* counts_out) {
    return guarded([&] {
        VH_REQUIRE(t != nullptr && offsets != nullptr, "NULL argument");
        VH_REQUIRE(n >= 0, "negative sequence count");
        for (int64_t i = 0; i < n; ++i) VH_REQUIRE(offsets[i + 1] >= offsets[i], "offsets must be non-decreasing");
        t->n_counts = n;
        if (n == 0) return;
        const int64_t total = offsets[n] - offsets[0];
        VH_REQUIRE(total == 0 || bases != nullptr, "NULL argument");
        t->bases.ensure((size_t)std::max<int64_t>(1, total));
        t->offsets.ensure((size_t)n + 1);
        t->counts.ensure((size_t)n * kKmers);
        std::vector<int64_t> rel((size_t)n + 1);
        for (int64_t i = 0; i <= n; ++i) rel[(size_t)i] = offsets[i] - offsets[0];
        if (total) VH_HIP(hipMemcpyAsync(t->bases.p, bases + offsets[0], (size_t)total, hipMemcpyHostToDevice, t->stream));
        VH_HIP(hipMemcpyAsync(t->offsets.p, rel.data(), sizeof(int64_t) * rel.size(), hipMemcpyHostToDevice, t->stream));
        const int grid = (int)std::min<int64_t>(n, 256 * 16);
        hipLaunchKernelGGL(tnf_kmercounts_kernel, dim3(grid), dim3(256), 0, t->stream, t->bases.p, t->offsets.p, n, t->counts.p);
        VH_HIP(hipGetLastError());
        if (counts_out)
            VH_HIP(hipMemcpyAsync(counts_out, t->counts.p, sizeof(uint32_t) * (size_t)n * kKmers, hipMemcpyDeviceToHost, t->stream));
        VH_HIP(hipStreamSynchronize(t->stream));
    });
}

int vh_tnf_project(vh_tnf* t, const float* fourmers, int64_t n, int mask_bits, float* tnf_out) {
    return guarded([&] {
        VH_REQUIRE(t != nullptr && tnf_out != nullptr, "NULL argument");
        VH_REQUIRE(mask_bits >= 0 && mask_bits <= 23, "Must mask between 0 and 23 bits");
        VH_REQUIRE(n >= 0, "negative row count");
        VH_REQUIRE(fourmers != nullptr || n == t->n_counts, "no resident counts for %lld rows (call vh_tnf_kmercounts first)",
                   (long long)n);
        if (n == 0) return;
        t->fourmers.ensure((size_t)n * kKmers);
        t->tnf.ensure((size_t)n * kTnf);
        const unsigned nb = (unsigned)ceil_div(n, 32);
        if (fourmers) {
            // staged in the output buffer of the normalisation (normalised in place on the device)
            VH_HIP(hipMemcpyAsync(t->fourmers.p, fourmers, sizeof(float) * (size_t)n * kKmers, hipMemcpyHostToDevice, t->stream));
            hipLaunchKernelGGL(tnf_normalise_kernel<false>, dim3(nb), dim3(256), 0, t->stream, (const void*)t->fourmers.p, n,
                               t->fourmers.p);
        } else {
            hipLaunchKernelGGL(tnf_normalise_kernel<true>, dim3(nb), dim3(256), 0, t->stream, (const void*)t->counts.p, n,
                               t->fourmers.p);
        }
        VH_HIP(hipGetLastError());
        hipLaunchKernelGGL(tnf_project_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, t->stream, t->fourmers.p, n,
                           t->kernel_p.p, mask_bits, t->tnf.p);
        VH_HIP(hipGetLastError());
        VH_HIP(hipMemcpyAsync(tnf_out, t->tnf.p, sizeof(float) * (size_t)n * kTnf, hipMemcpyDeviceToHost, t->stream));
        VH_HIP(hipStreamSynchronize(t->stream));
    });
}

}  // extern "C"
