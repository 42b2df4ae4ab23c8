// cluster.hip -- MI355X (gfx950) kernels + C ABI for the medoid-scan side of vamb/cluster.py.
//
// Data layout in HBM (per handle):
//   Mt      float [L4][ld]   the L2-normalised latent matrix, COLUMN-major (SoA): column c of every
//                            row is contiguous, so a wavefront reading 4 rows per lane issues fully
//                            coalesced 1 KiB global_load_dwordx4 per column.  ld = round_up(n, 1024),
//                            L4 = round_up(L, 4) (zero columns: fmaf(0,0,acc) == acc, exact).
//   lengths float [ld]       contig lengths as float32 (torch.Tensor(lengths), cluster.py:277)
//   kept    u8    [ld]       live mask (kept_mask, cluster.py:227); padding rows are 0
// One scan pass reads n*(4*L4 + 4 + 1) bytes and is HBM-bound (0.5 flop/B per medoid); up to 32
// medoids share one pass (the <=25 candidates of a wander_medoid round, cluster.py:415-450).
//
// Arithmetic contract (bit-exact with oracle/cluster_scan.c):
//   dot = fmaf chain over columns ascending from +0.0f; d = 0.5f - dot; d(medoid) = 0
//   density / histogram accumulate exactly in int64 fixed point (order-free => atomics are legal)
// This file is compiled with -ffp-contract=off so that nothing but the explicit fmaf is fused.
#include "comm.hpp"
#include "common.hpp"

#include <algorithm>
#include <dlfcn.h>
#include <execinfo.h>
#include <map>
#include <mutex>
#include <chrono>
#include <cmath>
#include <deque>
#include <unordered_map>
#include <atomic>
#include <memory>

namespace vh {
thread_local std::string g_last_error;

namespace {
std::mutex g_option_mutex;
std::map<std::string, int64_t>& option_table() {
    static std::map<std::string, int64_t> t;
    return t;
}
std::map<std::string, std::string>& option_strings() {
    static std::map<std::string, std::string> t;
    return t;
}
}  // namespace

int64_t option(const char* name, int64_t dflt) {
    std::lock_guard<std::mutex> lock(g_option_mutex);
    const auto it = option_table().find(name);
    return it == option_table().end() ? dflt : it->second;
}

const char* option_string(const char* name) {
    std::lock_guard<std::mutex> lock(g_option_mutex);
    const auto it = option_strings().find(name);
    return it == option_strings().end() ? nullptr : it->second.c_str();
}

// ---- guard mode of the device allocator (common.hpp) ----
namespace {
struct GuardRec {
    size_t bytes, guard;
    uint64_t id;
    void* frames[10];
    int n_frames;
};
std::mutex g_guard_mutex;
std::map<void*, GuardRec>& guard_table() {
    static std::map<void*, GuardRec> t;
    return t;
}
uint64_t g_guard_next_id = 0;
constexpr unsigned char kGuardByte = 0xA5;
}  // namespace
size_t guard_bytes() { return (size_t)option("debug.guard_bytes", 0); }
void guard_track(void* p, size_t bytes, size_t guard) {
    (void)hipMemset(static_cast<char*>(p) + bytes, kGuardByte, guard);
    GuardRec r{bytes, guard, 0, {}, 0};
    r.n_frames = backtrace(r.frames, 10);
    std::lock_guard<std::mutex> lock(g_guard_mutex);
    r.id = g_guard_next_id++;
    guard_table()[p] = r;
}
void guard_forget(void* p) {
    std::lock_guard<std::mutex> lock(g_guard_mutex);
    guard_table().erase(p);
}
}

using namespace vh;

namespace {

constexpr int kRowsPerThread = 4;
constexpr int kBlock = 256;
constexpr int kRowsPerBlock = kBlock * kRowsPerThread;  // 1024
constexpr int kResultWords = VH_NBINS + 4;              // density, hist[60], n_within, n_lt, list cursor
constexpr int kMaxMedoids = 32;
constexpr int kListCap = 2048;    // rows within the medoid radius kept per medoid by the scan itself
constexpr int64_t kMinScanBlocks = 768;   // workgroups wanted before lanes are given more than one row
constexpr int kListRing = 64;     // scans whose lists / histograms stay readable in host-mapped memory (17 MB)
// The pass accumulators exist in kResultReplicas copies: a workgroup flushes its non-zero LDS accumulators into copy
// blockIdx.x % kResultReplicas, the publish kernel adds the copies up.  With ONE copy every workgroup of a pass added
// into the same ~25 addresses (density, two counts, the populated histogram bins) and the serialised same-address
// device atomics were 40 % of the kernel time of a small pass (100 k rows, 1 medoid: 13.1 us with the flush, 7.0 us without;
// profiles/r02q_scan_ablation.txt).  The list cursor (word 63) lives in copy 0 only.
constexpr int kResultReplicas = 8;    // (the publish kernel reads and zeroes every copy: more copies cost it more than they save)
constexpr int kLocalCap = 128;    // per-block, per-medoid staging of list entries in LDS
constexpr int kSpecWindow = 16;  // upcoming seeds the native state machine looks at when it fills the free medoid slots of a pass
                                 // (8 until the fill's seed-row memo: 283 k -> 258 k passes per C2 sweep, profiles/r04w_*)
constexpr int kKeepList = 32;    // within-radius lists up to this length are copied out of the ring when a row is scanned ahead
constexpr int64_t kMaxEntryAgeDefault = 32;   // measured at C2 (profiles/r03r_*): 8 -> 15.9 s, 32 -> 14.9 s, 128 -> 15.3 s, 512 -> 22.4 s   // emissions a cached entry may lag behind before it is dropped unseen (its lazy check walks the log)
constexpr size_t kMaxCached = 16384; // cached medoid statistics (hard cap; the age rule keeps it far below)

// medoid rows travel in the kernel arguments (no upload, no gather launch)
struct MedoidRows {
    long long row[kMaxMedoids];
};

// Rows whose live flag the NEXT pass clears before it reads any flag (round 6).  The state machine removes the members of
// a cluster it emitted from a cached within-radius list -- 174 k of the 218 k emissions of a C2 sweep -- by clearing <= 32
// flags per one-thread-block launch in front of the next pass: one more dependent launch (~3 us of stream time and ~2.5 us
// of host time) per emission.  Now the rows wait on the handle and travel in the kernel arguments of the next scan: EVERY
// workgroup clears all of them (the same zeros to the same bytes) in front of the barrier that ends its prologue, so its own
// later loads see them whatever the other workgroups have done so far -- a workgroup only reads the flags of rows it scans.
// Longer lists, and every other reader of the flags (select, compaction, downloads), go through the remove kernel first.
constexpr int kRmCap = 56;
struct RmRows {
    int n;
    int32_t row[kRmCap];
};
__device__ __forceinline__ void apply_removals(uint8_t* kept_w, const RmRows& rm) {
    if ((int)threadIdx.x < rm.n) kept_w[rm.row[threadIdx.x]] = 0;   // (a __syncthreads() of the caller's prologue follows)
}

// Timing build only (-DVAMBHIP_TIMING_EXPERIMENTS): constant-clock (100 MHz) stamps of the phases of a pass, one row of 8 per
// workgroup (row kStampRows - 1: the publish kernel), read back by vh_debug_scan_timeline.  Absent from the product build.
#ifdef VAMBHIP_TIMING_EXPERIMENTS
constexpr int kStampRows = 4096;
__device__ unsigned long long* g_scan_stamps = nullptr;
#define SCAN_STAMP(slot)                                                                                                   \
    do {                                                                                                                   \
        if (g_scan_stamps != nullptr && threadIdx.x == 0 && blockIdx.x < kStampRows - 1)                                   \
            g_scan_stamps[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64();                                               \
    } while (0)
#define PUBLISH_STAMP(slot)                                                                                                \
    do {                                                                                                                   \
        if (g_scan_stamps != nullptr && threadIdx.x == 0) g_scan_stamps[(size_t)(kStampRows - 1) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define SCAN_STAMP(slot) do { } while (0)
#define PUBLISH_STAMP(slot) do { } while (0)
#endif

// torch.linspace(0.0, 0.3, 61) float32 bit patterns (== edges torch.histogram writes, cluster.py:288,
// 475-481).  tests/test_lib_abi.py asserts the table equals torch.linspace.
__constant__ uint32_t c_edge_bits[VH_NBINS + 1] = {
    0x00000000u, 0x3ba3d70bu, 0x3c23d70bu, 0x3c75c290u, 0x3ca3d70bu, 0x3cccccceu,
    0x3cf5c290u, 0x3d0f5c2au, 0x3d23d70bu, 0x3d3851ecu, 0x3d4cccceu, 0x3d6147afu,
    0x3d75c290u, 0x3d851eb9u, 0x3d8f5c2au, 0x3d99999au, 0x3da3d70bu, 0x3dae147cu,
    0x3db851ecu, 0x3dc28f5du, 0x3dccccceu, 0x3dd70a3eu, 0x3de147afu, 0x3deb8520u,
    0x3df5c290u, 0x3e000001u, 0x3e051eb9u, 0x3e0a3d71u, 0x3e0f5c2au, 0x3e147ae2u,
    0x3e19999au, 0x3e1eb852u, 0x3e23d70au, 0x3e28f5c3u, 0x3e2e147bu, 0x3e333333u,
    0x3e3851ecu, 0x3e3d70a4u, 0x3e428f5cu, 0x3e47ae15u, 0x3e4ccccdu, 0x3e51eb85u,
    0x3e570a3eu, 0x3e5c28f6u, 0x3e6147aeu, 0x3e666667u, 0x3e6b851fu, 0x3e70a3d8u,
    0x3e75c290u, 0x3e7ae148u, 0x3e800000u, 0x3e828f5cu, 0x3e851eb9u, 0x3e87ae15u,
    0x3e8a3d71u, 0x3e8ccccdu, 0x3e8f5c29u, 0x3e91eb85u, 0x3e947ae2u, 0x3e970a3eu,
    0x3e99999au};

// ---------------------------------------------------------------------------------------------
// The REFERENCE's evaluation order (option scan.reference_order).  `matrix.matmul(matrix[index])` (cluster.py:674) and
// `matrix.norm(dim=1)` (cluster.py:668) are float32 reductions whose order the reference does not define; on the torch 2.10 /
// oneMKL 2024.2 / AVX-512 CPU build they were measured (oracle/probe_reference_order.py, profiles/r03_reference_order_probe.txt)
// and restated in oracle/cluster_scan.c (dot_ref / norm_ref), which then equals torch bit for bit for every latent width.
// With these two functions the distances and the normalised matrix ARE the reference's on that build, and the near-tie of
// the 100 k sigma = 0.5 fixture (0.04999998 vs 0.05000007 at the medoid radius) falls on the reference's side.
//   dot : s = a0 x0;  16 lanes {s, 0, ...}; every full block of 16 columns from column 1 on accumulated lane-wise with fma;
//         halving tree (p + 8, p + 4, p + 2, p + 1); the (L - 1) % 16 columns left form one more block whose lane 0 starts
//         from the running sum
//   norm: 8 lanes of fma over the full blocks of 8 columns, lanes added 0..7 in order; of the L % 8 columns left the first
//         four (if there are four) add their ROUNDED squares one by one, the last <= 3 are fused; sqrtf
// (this file is compiled with -ffp-contract=off: a * b + c below is a rounded product and a rounded sum)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ref_tree16(const float (&v)[16]) {
    float a[8], b[4];
#pragma unroll
    for (int p = 0; p < 8; ++p) a[p] = v[p] + v[p + 8];
#pragma unroll
    for (int p = 0; p < 4; ++p) b[p] = a[p] + a[p + 4];
    const float c0 = b[0] + b[2], c1 = b[1] + b[3];
    return c0 + c1;
}

template <class FX, class FQ>
__device__ __forceinline__ float ref_dot(int L, FX x, FQ q) {
    float s = x(0) * q(0);
    const int nfull = (L - 1) / 16, rem = (L - 1) % 16;
    int k = 1;
    float acc[16];
    if (nfull > 0) {
        acc[0] = s;
#pragma unroll
        for (int p = 1; p < 16; ++p) acc[p] = 0.0f;
        for (int b = 0; b < nfull; ++b, k += 16) {
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[p] = __builtin_fmaf(x(k + p), q(k + p), acc[p]);
        }
        s = ref_tree16(acc);
    }
    if (rem > 0) {
        acc[0] = s;
#pragma unroll
        for (int p = 1; p < 16; ++p) acc[p] = 0.0f;
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (p < rem) acc[p] = __builtin_fmaf(x(k + p), q(k + p), acc[p]);
        s = ref_tree16(acc);
    }
    return s;
}

// d = 0.5f - <x, q> in the reference's order, FOUR LANES PER PAIR.  The reference's order is a 16-lane SIMD program (AVX-512);
// lane p = lane & 3 of a group of four plays its lanes p, p + 4, p + 8, p + 12, so the halving tree's first two levels are
// in-lane adds and the last two are xor-2 / xor-1 exchanges.  A wavefront re-evaluates 16 pairs per call with ~12 registers and
// one round trip to memory.  (Round 3 evaluated one pair per lane with ref_dot inlined at every drain site: 32 accumulators and 64
// loads in flight raised the register count of EVERY scan kernel -- 56 -> 115 VGPRs for the one-medoid kernel, scratch spills in
// the matrix-pipe kernel -- which was the 1.6x of that round's filter mode.)  x, q: strided vectors in global memory; every lane
// of the wavefront must be active (cross-lane exchanges); all four lanes of a group return the distance.
__device__ __forceinline__ float ref_g4_tree(float a0, float a1, float a2, float a3) {
    const float t0 = a0 + a2;   // v[p] + v[p + 8]
    const float t1 = a1 + a3;   // v[p + 4] + v[p + 12]
    const float b = t0 + t1;    // a[p] + a[p + 4]
    const float c = b + __shfl_xor(b, 2);
    return c + __shfl_xor(c, 1);
}
__device__ __forceinline__ float ref_distance_g4(const float* __restrict__ x, int64_t xs, const float* __restrict__ q, int64_t qs,
                                                 int L, int p) {
    // Per 16-lane block the lane's four (x, q) pairs are REQUESTED TOGETHER and then used: one trip to memory per block (written
    // as load-use-load-use the compiler waits for every pair: nine dependent trips per round at L = 32 instead of three, which
    // was ~7 us per pass of a C2 sweep, profiles/r04c_*).  Columns of the remainder block beyond the width re-read column L - 1
    // and are zeroed: 0 * q added to a +0 lane is +0, what the reference's masked lanes hold.
    const int last = L - 1;
    const float xc = x[0], qc = q[0];   // (column 0 travels with the first block: its product is formed behind that block's loads)
    float s;
    const int nfull = (L - 1) >> 4, rem = (L - 1) & 15;
    int k = 1;
    if (nfull > 0) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int b = 0; b < nfull; ++b, k += 16) {
            const int64_t c = k + p;
            const float x0 = x[c * xs], x1 = x[(c + 4) * xs], x2 = x[(c + 8) * xs], x3 = x[(c + 12) * xs];
            const float q0 = q[c * qs], q1 = q[(c + 4) * qs], q2 = q[(c + 8) * qs], q3 = q[(c + 12) * qs];
            if (b == 0 && p == 0) a0 = xc * qc;
            a0 = __builtin_fmaf(x0, q0, a0);
            a1 = __builtin_fmaf(x1, q1, a1);
            a2 = __builtin_fmaf(x2, q2, a2);
            a3 = __builtin_fmaf(x3, q3, a3);
        }
        s = ref_g4_tree(a0, a1, a2, a3);
    } else {
        s = xc * qc;
    }
    if (rem > 0) {
        const int c = k + p;
        const int c0 = c <= last ? c : last, c1 = c + 4 <= last ? c + 4 : last, c2 = c + 8 <= last ? c + 8 : last,
                  c3 = c + 12 <= last ? c + 12 : last;
        float x0 = x[(int64_t)c0 * xs], x1 = x[(int64_t)c1 * xs], x2 = x[(int64_t)c2 * xs], x3 = x[(int64_t)c3 * xs];
        const float q0 = q[(int64_t)c0 * qs], q1 = q[(int64_t)c1 * qs], q2 = q[(int64_t)c2 * qs], q3 = q[(int64_t)c3 * qs];
        if (c > last) x0 = 0.0f;
        if (c + 4 > last) x1 = 0.0f;
        if (c + 8 > last) x2 = 0.0f;
        if (c + 12 > last) x3 = 0.0f;
        // (lane p = 0 of the block always holds a real column -- the block exists -- so the running sum is never paired with a zero)
        const float a0 = __builtin_fmaf(x0, q0, p == 0 ? s : 0.0f);
        const float a1 = __builtin_fmaf(x1, q1, 0.0f), a2 = __builtin_fmaf(x2, q2, 0.0f), a3 = __builtin_fmaf(x3, q3, 0.0f);
        s = ref_g4_tree(a0, a1, a2, a3);
    }
    return 0.5f - s;
}

// Reference-order distances for the lanes of a wavefront that `need` one (all 64 lanes must be active): sixteen pairs per round,
// the g-th needy lane of a round served by lane group g.  xrow(r): the row vector of matrix row r (stride xs); qptr(j, &qs): the
// query vector of medoid slot j.  Returns the lane's own new distance (d unchanged where !need).
// Query vector of slot j: q_rows + j * q_stride_j (unit stride) when q_rows != nullptr, else column `med[j]` of Mt (stride ld).
// (Inlined at every drain site -- up to 32 in one kernel; cluster.hip is compiled with a raised -pragma-unroll-threshold so that
// the evaluation loops around those sites still unroll.  As a real call it would cost every kernel ~60 VGPRs: arguments of device
// functions travel in vector registers.)
__device__ __forceinline__ float ref_recheck_wave(bool need, float d, int32_t row, int j, const float* __restrict__ Mt,
                                                            int64_t ld, int L, const float* __restrict__ q_rows, int q_stride_j,
                                                            const int32_t* __restrict__ med) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned long long m = __builtin_amdgcn_ballot_w64(need);
    const int g = lane >> 2;
    const unsigned long long below = (1ull << lane) - 1ull;
    while (m != 0ull) {   // (uniform)
        const unsigned long long m0 = m;
        int src = -1;
        for (int i = 0; i < 16 && m != 0ull; ++i) {   // (uniform) the i-th needy lane goes to group i
            const int b = __builtin_ctzll(m);
            m &= m - 1ull;
            if (g == i) src = b;
        }
        const int from = src < 0 ? lane : src;
        // (both exchanges unconditional: a bpermute executed by a subset of the lanes reads 0 from every lane outside it)
        const int32_t srow = __shfl(row, from);
        const int sj_any = __shfl(j, from);
        const int sj = src < 0 ? 0 : sj_any;
        const float* q = q_rows ? q_rows + (size_t)sj * q_stride_j : Mt + med[sj];
        const int64_t qs = q_rows ? 1 : ld;
        const float dd = ref_distance_g4(Mt + (src < 0 ? 0 : srow), ld, q, qs, L, lane & 3);   // idle groups re-read row 0: harmless
        const int rank = __popcll(m0 & below);
        const float got = __shfl(dd, 4 * (rank & 15));
        if (((m0 >> lane) & 1ull) != 0ull && rank < 16) d = got;
    }
    return d;
}

template <class FX>
__device__ __forceinline__ float ref_norm(int L, FX x) {
    float acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = 0.0f;
    int d = 0;
    for (; d + 8 <= L; d += 8) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float v = x(d + p);
            acc[p] = __builtin_fmaf(v, v, acc[p]);
        }
    }
    float s = acc[0];
#pragma unroll
    for (int p = 1; p < 8; ++p) s = s + acc[p];
    if (L - d >= 4) {
        for (int p = 0; p < 4; ++p) {
            const float v = x(d + p);
            const float sq = v * v;
            s = s + sq;
        }
        d += 4;
    }
    for (; d < L; ++d) {
        const float v = x(d);
        s = __builtin_fmaf(v, v, s);
    }
    return __builtin_sqrtf(s);
}

// ---------------------------------------------------------------------------------------------
// K9: normalise rows (cluster.py:653-669) and transpose to the SoA layout.  One thread per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void clu_normalize_transpose_kernel(float* __restrict__ rowmajor, int64_t n, int L,
                                                                     int do_normalize, float* __restrict__ Mt,
                                                                     int64_t ld, int ref_order) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    float* r = rowmajor + row * (int64_t)L;
    if (!do_normalize) {
        for (int k = 0; k < L; ++k) Mt[(int64_t)k * ld + row] = r[k];
        return;
    }
    bool allzero = true;
    float ss = 0.0f;
    for (int k = 0; k < L; ++k) {
        const float x = r[k];
        allzero = allzero && (x == 0.0f);
        ss = __builtin_fmaf(x, x, ss);
    }
    const float inv_l = (float)(1.0 / (double)L);
    if (allzero) {
        ss = 0.0f;
        for (int k = 0; k < L; ++k) ss = __builtin_fmaf(inv_l, inv_l, ss);
    }
    const float sqrt2 = (float)1.4142135623730951;
    float nrm = __builtin_sqrtf(ss);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    if (ref_order) nrm = ref_norm(L, [&](int k) { return allzero ? inv_l : r[k]; });
    const float denom = nrm * sqrt2;
    for (int k = 0; k < L; ++k) {
        const float x = allzero ? inv_l : r[k];
        const float y = x / denom;
        r[k] = y;
        Mt[(int64_t)k * ld + row] = y;
    }
}

// row-major gather of arbitrary rows (get_rows): out[i][c] = Mt[c][rows ? rows[i] : i]
__global__ void clu_rows_to_rowmajor_kernel(const float* __restrict__ Mt, int64_t ld, int L,
                                            const int64_t* __restrict__ rows, int64_t k, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= k * L) return;
    const int64_t i = idx / L;
    const int c = (int)(idx - i * L);
    const int64_t r = rows ? rows[i] : i;
    out[idx] = Mt[(int64_t)c * ld + r];
}

// torch.histogram bin rule: edge[b] <= d < edge[b+1], last bin closed; -1 outside [edge0, edge60]
__device__ __forceinline__ int bin_of(float d, const float* edges) {
    if (!(d >= edges[0]) || !(d <= edges[VH_NBINS])) return -1;
    int b = (int)(d * 200.0f);
    b = b < 0 ? 0 : (b > VH_NBINS - 1 ? VH_NBINS - 1 : b);
    while (b > 0 && d < edges[b]) --b;
    while (b < VH_NBINS - 1 && d >= edges[b + 1]) ++b;
    return b;
}

// the same rule without loops: the estimate (int)(d * 200) is off by at most one bin (edges are linspace(0, 0.3, 61)
// in float32), so one downward or one upward correction is exact; callers guarantee edges[0] <= d <= edges[60]
__device__ __forceinline__ int bin_of_in_range(float d, const float* edges) {
    int b = (int)(d * 200.0f);
    b = b > VH_NBINS - 1 ? VH_NBINS - 1 : b;
    if (b > 0 && d < edges[b]) --b;
    else if (b < VH_NBINS - 1 && d >= edges[b + 1]) ++b;
    return b;
}

// round-to-nearest-even of a non-negative float product to an unsigned 64-bit integer; the float path is exact
// for x < 2^32 (rintf of an exactly representable product), larger values take the double path
__device__ __forceinline__ unsigned long long rn_u64(float x) {
    return x < 4.0e9f ? (unsigned long long)(unsigned int)__builtin_rintf(x)
                      : (unsigned long long)__double2ll_rn((double)x);
}

// ---------------------------------------------------------------------------------------------
// K6: multi-medoid scan.  Each thread owns RPT consecutive rows (4 for few medoids: 1 KiB coalesced
// loads per wave-instruction; 2 when KM >= 12 so that KM*RPT accumulators stay within ~128 VGPRs).
// ---------------------------------------------------------------------------------------------
template <int RPT>
__device__ __forceinline__ void load_rows(const float* __restrict__ p, float (&x)[RPT]) {
    if constexpr (RPT == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if constexpr (RPT == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        x[0] = v.x; x[1] = v.y;
    } else {
        x[0] = *p;
    }
}

template <int RPT>
__device__ __forceinline__ void load_live(const uint8_t* __restrict__ p, unsigned char (&x)[RPT]) {
    if constexpr (RPT == 4) {
        const uchar4 v = *reinterpret_cast<const uchar4*>(p);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if constexpr (RPT == 2) {
        const uchar2 v = *reinterpret_cast<const uchar2*>(p);
        x[0] = v.x; x[1] = v.y;
    } else {
        x[0] = *p;
    }
}

// Rows within the histogram range of a medoid (d <= 0.3) are rare (~1 % of the pairs) but nearly every wavefront has one
// per (medoid, row slot): handled in place they cost a divergent ~50-instruction body each.  Instead a lane that
// has such a pair appends it to its wavefront's queue in LDS (ballot + mbcnt, no atomics) and the queue is drained
// with all lanes busy: one hit per lane.  All accumulators are integers, so the order of accumulation is free.
constexpr int kHitCap = 256;      // queued hits per wavefront (drained when fewer than 64 slots remain)

// One (row, medoid) pair inside the histogram range: what sample_medoid / find_threshold record for it (all integer, order-free)
__device__ __forceinline__ void record_pair(float d, float len, int32_t row, int j, unsigned long long* __restrict__ acc_s,
                                            unsigned int* __restrict__ lcnt_s, int32_t* __restrict__ llist_s,
                                            const float* __restrict__ edges_s, int dbg) {
    const float radius = 0.05f;
    // rows inside the medoid radius: exact integer accumulation and the medoid's candidate list
    // (sample_medoid's `cluster`, cluster.py:621-626)
    if (d <= radius) {
        // len * (radius - d) in float32 as the reference computes it, then * 2^16 (exact) and RNE
        const float p = len * (radius - d);
        atomicAdd(&acc_s[j * kResultWords + 0], rn_u64(p * (float)VH_DENSITY_SCALE));
        atomicAdd(&acc_s[j * kResultWords + 1 + VH_NBINS], 1ull);
        if (d < radius) atomicAdd(&acc_s[j * kResultWords + 2 + VH_NBINS], 1ull);
        // appended block-locally in LDS, flushed once per block -- per-row global atomics on one cursor
        // serialise in L2 for dense medoids
        const unsigned int lp = atomicAdd(&lcnt_s[j], 1u);
        if (lp < (unsigned int)kLocalCap) llist_s[j * kLocalCap + lp] = row;
    }
    if (dbg & 2) return;       // timing experiment: no histogram
    // fixed-point histogram weight: len * 2^8 is exact in float32
    if (d >= edges_s[0])
        atomicAdd(&acc_s[j * kResultWords + 1 + bin_of_in_range(d, edges_s)], rn_u64(len * (float)VH_HIST_SCALE));
}

// scan.reference_order = 2 (the default): the tuned kernels act as a FILTER -- a pair is queued when its ascending-chain
// distance is within the slack of the histogram range -- and the drain re-evaluates a queued pair in the reference's order
// (ref_dot, from the resident matrix) only where the two orders can decide differently: inside the medoid radius, where the
// distance's value is recorded, and next to a bin edge.  Two float32 evaluations of the same dot product of rows of norm
// 1 / sqrt(2) differ by less than ref_slack(L), so no pair the reference order would record is missed and every recorded
// quantity is the reference order's, bit for bit (tests: test_scan_accumulators_bit_exact against the oracle in that order);
// the hot loops are untouched.
struct RefSrc {
    const float* Mt;       // resident matrix [L4][ld]
    int64_t ld;
    int L, L4;
    const float* q_rows;   // explicit query vectors [km][L4] (row-sharded execution / vh_clu_scan with queries) or nullptr
    float slack;           // bound on |d(ascending chain) - d(reference order)| for this latent width (ref_slack below)
};
// Two float32 evaluations of the same dot product differ by at most 2 gamma_L sum|x_i q_i| <= 2 L 2^-24 * 0.5 (rows have norm
// 1 / sqrt(2)) plus the two roundings of 0.5f - dot: 6e-8 (L + 1).  Five times that, never below 1e-5.
inline float ref_slack(int L) { return std::max(1.0e-5f, 3.0e-7f * (float)L); }

// Can a distance within `slack` of the ascending-chain distance d fall into another histogram bin, or on the other side of the
// last edge?  The edges are float32 linspace(0, 0.3, 61): within 1.5e-8 of i * 0.005, and d * 200 rounds by < 4e-6.
__device__ __forceinline__ bool near_bin_edge(float d, float slack) {
    const float t = d * 200.0f;
    return __builtin_fabsf(t - __builtin_rintf(t)) <= slack * 200.0f + 2.0e-5f;
}

template <bool REF = false>
__device__ __forceinline__ void drain_hits(const float4* __restrict__ hq, int qn, int lane,
                                           unsigned long long* __restrict__ acc_s, unsigned int* __restrict__ lcnt_s,
                                           int32_t* __restrict__ llist_s, const float* __restrict__ edges_s,
                                           const int32_t* __restrict__ med_s, int dbg, const RefSrc& ro,
                                           const float* __restrict__ lengths = nullptr) {
    if (dbg & 8) return;   // timing experiment: queued pairs are dropped
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int e0 = 0; e0 < qn; e0 += 64) {   // (uniform trip count: the reference-order re-evaluation exchanges data across lanes)
        const int e = e0 + lane;
        const bool valid = e < qn;
        const float4 v = hq[valid ? e : 0];
        const int32_t row = __float_as_int(v.z);
        const int j = __float_as_int(v.w);
        // the distance of a medoid to itself is 0 by definition (cluster.py:619), not 0.5 - <q, q>: decided here, once
        // per queued pair, instead of once per (row, medoid) pair in the scan loop
        const bool self = row == med_s[j];
        float d = self ? 0.0f : v.x;
        if constexpr (REF) {
            // The reference-order distance is within ro.slack of the chain distance.  Inside the medoid radius (+ slack) its VALUE
            // is recorded (density), elsewhere only its histogram bin: the pair is re-evaluated in the reference's order only
            // if it is that close to the radius or to a bin edge (0.4 % of the queued pairs at L = 32) -- everywhere else the
            // chain distance decides exactly what the reference-order distance would.
            const bool need = valid && !self && (d <= 0.05f + ro.slack || near_bin_edge(d, ro.slack));
            if (__builtin_amdgcn_ballot_w64(need) != 0ull)
                d = ref_recheck_wave(need, d, row, j, ro.Mt, ro.ld, ro.L, ro.q_rows, ro.L4, med_s);
        }
        if (!valid || !(d <= edges_s[VH_NBINS])) continue;   // (beyond the last edge: passed the filter's slack only)
        // the matrix-pipe kernel queues (d, row, medoid) only and leaves the length to this loop: a global load in its
        // tile loop would make every tile with a pair of interest wait for the prefetched tiles behind it
        const float len = lengths ? lengths[row] : v.y;
        record_pair(d, len, row, j, acc_s, lcnt_s, llist_s, edges_s, dbg);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// End of a scan workgroup: non-zero LDS accumulators to the pass accumulators (exact integer atomics), then the
// block-local candidate lists: one cursor atomic per (block, medoid with hits).  A block that ran out of local room
// poisons the cursor; the host then sees cursor != n_within and runs a select pass.
template <int KM>
__device__ __forceinline__ void scan_flush(int tid, const unsigned long long* __restrict__ acc_s,
                                           const unsigned int* __restrict__ lcnt_s, const int32_t* __restrict__ llist_s,
                                           unsigned long long* __restrict__ results, int32_t* __restrict__ lists, int dbg = 0) {
    __shared__ unsigned int start_s[kMaxMedoids];
    __syncthreads();
    SCAN_STAMP(5);
    if (dbg & 4) return;   // timing experiment: no flush
    unsigned long long* copy = results + (size_t)(blockIdx.x % kResultReplicas) * kMaxMedoids * kResultWords;
    for (int i = tid; i < KM * kResultWords; i += kBlock) {
        const unsigned long long v = acc_s[i];
        if (v != 0ull) atomicAdd(&copy[i], v);
    }
    // The cursor atomics of ALL medoids with local entries go out together (thread j owns medoid j): one round trip.
    // (They used to be issued one medoid at a time between two barriers -- a returning device atomic each, ~1 us -- which made
    // the flush of a workgroup holding members of many medoids the longest part of a dense pass.)
    if (tid < KM) {
        const unsigned int cnt = lcnt_s[tid];
        if (cnt != 0u) {
            unsigned int* cursor = reinterpret_cast<unsigned int*>(&results[tid * kResultWords + 3 + VH_NBINS]);
            start_s[tid] = cnt > (unsigned int)kLocalCap ? atomicOr(cursor, 0x80000000u) | 0x80000000u : atomicAdd(cursor, cnt);
        }
    }
    __syncthreads();
    for (int j = 0; j < KM; ++j) {
        const unsigned int cnt = lcnt_s[j];
        if (cnt == 0u) continue;
        const unsigned int start = start_s[j];
        if (start + cnt <= (unsigned int)kListCap && tid < (int)cnt) lists[j * kListCap + start + tid] = llist_s[j * kLocalCap + tid];
    }
#ifdef VAMBHIP_TIMING_EXPERIMENTS
    __builtin_amdgcn_s_waitcnt(0);   // (every counter at 0: the stamp marks the flush RETIRED, not issued)
    SCAN_STAMP(6);
#endif
}

// (Requesting the first row block before the prologue -- LDS initialisation, query gather, barrier -- to overlap the two
// memory round trips of a small pass measured neutral at 10^5 rows and 25 % slower at 2 M x 32, k = 8: +25 VGPRs.)
// LC: latent width known at compile time (0 = runtime L4).  With LC every column load of a lane's rows is issued
// before the first fmaf, so a wavefront pays one memory round trip per row block instead of L4/4 dependent ones:
// the matrices the generator scans late in a sweep (10^5 rows, ~1 workgroup per CU) are latency-bound.
// PIPE: (runtime-width loop, many medoids) software pipeline, see the kernel body.
template <int KM, int RPT, int LC, int PIPE = 0, bool REF = false>
__global__ __launch_bounds__(kBlock) void clu_scan_kernel(const float* __restrict__ Mt, int64_t ld, int L4,
                                                          const float* __restrict__ lengths,
                                                          const uint8_t* kept, int64_t n,
                                                          const float* __restrict__ q_ext,
                                                          const MedoidRows medoid,
                                                          unsigned long long* __restrict__ results,
                                                          int32_t* __restrict__ lists, int dbg, const RefSrc ro, uint8_t* kept_w,
                                                          const RmRows rm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* hq_s = reinterpret_cast<float4*>(smem_raw);                                  // [kBlock/64][kHitCap]
    unsigned long long* acc_s = reinterpret_cast<unsigned long long*>(hq_s + (kBlock / 64) * kHitCap);   // [KM][kResultWords]
    float* edges_s = reinterpret_cast<float*>(acc_s + KM * kResultWords);               // [64]
    float* q_s = edges_s + 64;                                                          // [KM][L4]
    unsigned int* lcnt_s = reinterpret_cast<unsigned int*>(q_s + KM * L4);              // [KM]
    int32_t* llist_s = reinterpret_cast<int32_t*>(lcnt_s + KM);                         // [KM][kLocalCap]
    int32_t* med_s = llist_s + KM * kLocalCap;                                          // [KM] medoid rows (-1: none)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    SCAN_STAMP(0);
    float4* hq = hq_s + (tid >> 6) * kHitCap;
    for (int i = tid; i < KM; i += kBlock) med_s[i] = (int32_t)medoid.row[i];
    for (int i = tid; i < KM * kResultWords; i += kBlock) acc_s[i] = 0ull;
    for (int i = tid; i <= VH_NBINS; i += kBlock) edges_s[i] = __uint_as_float(c_edge_bits[i]);
    for (int i = tid; i < KM; i += kBlock) lcnt_s[i] = 0u;
    apply_removals(kept_w, rm);
    // query vectors: explicit (row-sharded execution) or row medoid.row[j] of the resident matrix
    if constexpr (PIPE == 0) {
        for (int i = tid; i < KM * L4; i += kBlock) {
            const int j = i / L4, c = i - j * L4;
            q_s[i] = q_ext ? q_ext[i] : Mt[(int64_t)c * ld + medoid.row[j]];
        }
    }
    __syncthreads();
    SCAN_STAMP(1);

    const float edge_hi = edges_s[VH_NBINS];
    int qn = 0;   // hits queued by this wavefront (uniform)

    for (int64_t base = ((int64_t)blockIdx.x * kBlock + tid) * RPT; base < n;
         base += (int64_t)gridDim.x * kBlock * RPT) {
        unsigned char live[RPT];
        load_live<RPT>(kept + base, live);
        bool any = false;
#pragma unroll
        for (int r = 0; r < RPT; ++r) any = any || (live[r] != 0);
        // whole wavefront dead (already emitted rows): skip the column loads
        if (__ballot(any) == 0ull) continue;

        const float* col = Mt + base;
        (void)col;
        float len[RPT];
        // Liveness (and the timing switch) folded into a per-row threshold: a pair is of interest iff d <= thr[r]
        // (beyond the last histogram edge, 0.3 > radius, there is nothing to record)
        float thr[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) thr[r] = (live[r] != 0 && !(dbg & 1)) ? (REF ? edge_hi + ro.slack : edge_hi) : -__builtin_inff();
        // Evaluation of the finished dot products of medoids j0 .. j0 + NJ - 1.  Pairs of interest are rare (a medoid's
        // neighbourhood is a few hundred of 10^6 rows), so the common path is kept to two VALU instructions per pair --
        // d = 0.5 - dot and one compare whose lane mask is OR-ed on the scalar unit -- and ONE branch per group of four
        // medoids; the previous form (64-bit self-row compare, three conditions and a branch per pair) cost more than
        // the fmaf chains themselves (measured, no pair of interest at all: 3.4 us per medoid at 2 M x 32 against 0.8 us
        // of fmaf, profiles/r02d_scan_bench_dbg1.json).
        auto evaluate = [&](auto& acc, int j0) {
            constexpr int NJ = (int)(sizeof(acc) / sizeof(acc[0]));
            constexpr int GE = NJ % 4 == 0 ? 4 : (NJ % 2 == 0 ? 2 : 1);
            // every dot product is finished here: without the pin the compiler sinks each medoid's fmaf chain into
            // the branchy evaluation below and keeps the query registers of all medoids alive across it
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < RPT; ++r) asm volatile("" : "+v"(acc[j][r]));
#pragma unroll
            for (int g0 = 0; g0 < NJ; g0 += GE) {
                unsigned long long any = 0ull;
#pragma unroll
                for (int jj = g0; jj < g0 + GE; ++jj)
#pragma unroll
                    for (int r = 0; r < RPT; ++r) any |= __builtin_amdgcn_ballot_w64((0.5f - acc[jj][r]) <= thr[r]);
                if (any == 0ull) continue;
#pragma unroll
                for (int jj = g0; jj < g0 + GE; ++jj) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        const float d = 0.5f - acc[jj][r];
                        const bool hit = d <= thr[r];
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                        if (m != 0ull) {
                            if (hit) {
                                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32),
                                                                                    __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                                hq[pos] = make_float4(d, len[r], __int_as_float((int32_t)(base + r)), __int_as_float(j0 + jj));
                            }
                            qn += __popcll(m);
                            if (qn > kHitCap - 64) {
                                drain_hits<REF>(hq, qn, lane, acc_s, lcnt_s, llist_s, edges_s, med_s, dbg, ro);
                                qn = 0;
                            }
                        }
                    }
                }
            }
        };
        if constexpr (LC > 0) {
            // scalar column base + one 32-bit byte offset per lane (n < 2^30 rows, checked at creation): the
            // 32 x RPT loads in flight do not need 32 address pairs
            const uint32_t boff = (uint32_t)base * 4u;
            float x[LC][RPT];
#pragma unroll
            for (int c = 0; c < LC; ++c)
                load_rows<RPT>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(Mt + (int64_t)c * ld) + boff), x[c]);
            load_rows<RPT>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(lengths) + boff), len);
            __builtin_amdgcn_sched_barrier(0);   // every load is issued before the first fmaf ...
            float acc[KM][RPT];
#pragma unroll
            for (int j = 0; j < KM; ++j)
#pragma unroll
                for (int r = 0; r < RPT; ++r) acc[j][r] = 0.0f;
#pragma unroll
            for (int c = 0; c < LC; c += 4) {
#pragma unroll
                for (int j = 0; j < KM; ++j) {
                    const float4 qq = *reinterpret_cast<const float4*>(q_s + j * LC + c);
                    const float qv[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)   // columns ascending: the defined fmaf order
#pragma unroll
                        for (int r = 0; r < RPT; ++r) acc[j][r] = __builtin_fmaf(x[c + i][r], qv[i], acc[j][r]);
                    // ... and the query reads are not hoisted across the whole unrolled body (register pressure)
                    if ((j & 3) == 3 || j == KM - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            evaluate(acc, 0);
            SCAN_STAMP(2);
        } else if constexpr (PIPE != 0) {
            // Many medoids per pass.  Measured (profiles/r02b_scan_bench_lc*.json): 45 us + 4 us per medoid at 2 M x 32
            // whether the column loads are issued up front or four at a time -- neither HBM nor the loads bound it.  The
            // ISA shows why: each medoid's query quad was a broadcast ds_read_b128 issued immediately before the 8 fmaf
            // that consume it, so a wavefront (2 per SIMD at ~190 VGPRs) sat out one LDS round trip per 2 medoids.  The
            // queries are wave-uniform: here they come from global memory through the scalar cache (s_load_dwordx4 into
            // SGPRs, batched by the compiler) and enter the fmaf as scalar operands -- no LDS traffic, no VGPRs -- and
            // the rows of the column quad after the next are requested before the chains of the current one start.
            float acc[KM][RPT];
#pragma unroll
            for (int j = 0; j < KM; ++j)
#pragma unroll
                for (int r = 0; r < RPT; ++r) acc[j][r] = 0.0f;
            float xc[4][RPT], xn[4][RPT], xnn[4][RPT];
            auto load_x = [&](float (&x)[4][RPT], int c) {
#pragma unroll
                for (int i = 0; i < 4; ++i) load_rows<RPT>(col + (int64_t)(c + i) * ld, x[i]);
            };
            load_x(xc, 0);
            if (4 < L4) load_x(xn, 4);
            load_rows<RPT>(lengths + base, len);
            for (int c = 0; c < L4; c += 4) {
                if (c + 8 < L4) load_x(xnn, c + 8);
                // q_ext is QUAD-MAJOR here (clu_gather_quads_kernel): [L4 / 4][KM][4], so the KM quads of this column
                // quad are 16 KM contiguous bytes behind one scalar base
                const float4* qquad = reinterpret_cast<const float4*>(q_ext) + (c >> 2) * KM;
                // Batches of G medoids, column-major inside a batch: consecutive fmaf belong to DIFFERENT chains (a
                // chain's next link is G instructions away).  Medoid-major order put the 4 links of a quad back to
                // back and the VALU spent ~4 issue slots per fmaf waiting for its own result (measured: 17 cycles per
                // v_pk_fma_f32 at 2 wavefronts per SIMD).  Every chain still sees its columns in ascending order.
                constexpr int G = (KM % 8 == 0) ? 8 : 4;
#pragma unroll
                for (int g = 0; g < KM / G; ++g) {
                    float4 qb[G];
#pragma unroll
                    for (int j = 0; j < G; ++j) qb[j] = qquad[g * G + j];   // uniform address: s_load_dwordx16
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const float qv = i == 0 ? qb[j].x : (i == 1 ? qb[j].y : (i == 2 ? qb[j].z : qb[j].w));
#pragma unroll
                            for (int r = 0; r < RPT; ++r)
                                acc[g * G + j][r] = __builtin_fmaf(xc[i][r], qv, acc[g * G + j][r]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        xc[i][r] = xn[i][r];
                        xn[i][r] = xnn[i][r];
                    }
            }
            evaluate(acc, 0);
        } else {
            float acc[KM][RPT];
#pragma unroll
            for (int j = 0; j < KM; ++j)
#pragma unroll
                for (int r = 0; r < RPT; ++r) acc[j][r] = 0.0f;
            for (int c = 0; c < L4; c += 4) {
                float x[4][RPT];
#pragma unroll
                for (int i = 0; i < 4; ++i) load_rows<RPT>(col + (int64_t)(c + i) * ld, x[i]);
#pragma unroll
                for (int j = 0; j < KM; ++j) {
                    const float4 qq = *reinterpret_cast<const float4*>(q_s + j * L4 + c);
                    const float qv[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)   // columns ascending: the defined fmaf order
#pragma unroll
                        for (int r = 0; r < RPT; ++r) acc[j][r] = __builtin_fmaf(x[i][r], qv[i], acc[j][r]);
                }
            }
            load_rows<RPT>(lengths + base, len);
            evaluate(acc, 0);
        }
    }
    SCAN_STAMP(3);
    drain_hits<REF>(hq, qn, lane, acc_s, lcnt_s, llist_s, edges_s, med_s, dbg, ro);
    SCAN_STAMP(4);

    scan_flush<KM>(tid, acc_s, lcnt_s, llist_s, results, lists, dbg);
}

// ---------------------------------------------------------------------------------------------
// K6r: the scan in the REFERENCE's evaluation order (option scan.reference_order).  One row per lane, the distance of every
// (row, medoid) pair evaluated by ref_dot; 32 medoid slots like the matrix-pipe kernel (slots >= k_real are empty).  A plain
// kernel: this mode exists to be bit-identical with the reference's own distances, the tuned kernels above keep the
// ascending chain.  Same LDS layout, accumulators, candidate lists and flush as clu_scan_kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void clu_scan_ref_kernel(const float* __restrict__ Mt, int64_t ld, int L, int L4,
                                                              const float* __restrict__ lengths,
                                                              const uint8_t* kept, int64_t n,
                                                              const float* __restrict__ q_ext, const MedoidRows medoid,
                                                              int k_real, unsigned long long* __restrict__ results,
                                                              int32_t* __restrict__ lists, int dbg, uint8_t* kept_w, const RmRows rm) {
    constexpr int KM = kMaxMedoids;
    apply_removals(kept_w, rm);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* hq_s = reinterpret_cast<float4*>(smem_raw);                                  // (unused here; keeps the layout)
    unsigned long long* acc_s = reinterpret_cast<unsigned long long*>(hq_s + (kBlock / 64) * kHitCap);   // [KM][kResultWords]
    float* edges_s = reinterpret_cast<float*>(acc_s + KM * kResultWords);               // [64]
    float* q_s = edges_s + 64;                                                          // [KM][L4]
    unsigned int* lcnt_s = reinterpret_cast<unsigned int*>(q_s + KM * L4);              // [KM]
    int32_t* llist_s = reinterpret_cast<int32_t*>(lcnt_s + KM);                         // [KM][kLocalCap]
    int32_t* med_s = llist_s + KM * kLocalCap;                                          // [KM]
    const int tid = threadIdx.x;
    for (int i = tid; i < KM; i += kBlock) med_s[i] = (int32_t)medoid.row[i];
    for (int i = tid; i < KM * kResultWords; i += kBlock) acc_s[i] = 0ull;
    for (int i = tid; i <= VH_NBINS; i += kBlock) edges_s[i] = __uint_as_float(c_edge_bits[i]);
    for (int i = tid; i < KM; i += kBlock) lcnt_s[i] = 0u;
    for (int i = tid; i < KM * L4; i += kBlock) {
        const int j = i / L4, c = i - j * L4;
        float v = 0.0f;
        if (j < k_real) v = q_ext ? q_ext[i] : Mt[(int64_t)c * ld + medoid.row[j]];
        q_s[i] = v;
    }
    __syncthreads();
    const float edge_hi = edges_s[VH_NBINS];
    for (int64_t row = (int64_t)blockIdx.x * kBlock + tid; row < n; row += (int64_t)gridDim.x * kBlock) {
        if (kept[row] == 0) continue;
        const float len = lengths[row];
        const float* col = Mt + row;
        for (int j = 0; j < k_real; ++j) {
            float d = 0.0f;
            if ((int32_t)row != med_s[j]) {
                const float* qj = q_s + j * L4;
                d = 0.5f - ref_dot(L, [&](int c) { return col[(int64_t)c * ld]; }, [&](int c) { return qj[c]; });
            }
            if (d <= edge_hi && !(dbg & 1)) record_pair(d, len, (int32_t)row, j, acc_s, lcnt_s, llist_s, edges_s, dbg);
        }
    }
    scan_flush<KM>(tid, acc_s, lcnt_s, llist_s, results, lists, dbg);
}

// ---------------------------------------------------------------------------------------------
// K6m: passes with more than 8 medoids on the matrix pipe.  The VALU issues one wavefront fmaf per 4 cycles (PMC of the
// many-medoid VALU kernel, profiles/r02f_pmc_scan_*.txt: SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU quad-cycles, VALU ~70 % busy at
// 2 wavefronts per SIMD, 114 us for 32 medoids over 2 M x 32 with no pair of interest at all), i.e. 16 fmaf lanes per clock
// and SIMD; v_mfma_f32_32x32x2_f32 retires 32 exact fp32 multiply-adds per clock and SIMD on a pipe of its own.  One chain of
// L4 / 2 such MFMAs is the [32 rows] x [32 medoids] block of dot products, each accumulated as the k-ordered fmaf chain from
// +0 the VALU kernel runs (CDNA4 guide: bitwise equal), so accumulators, lists and cluster streams stay bit-identical.
//   A operand: lane l supplies x[row base + (l & 31)][column 2 s + (l >> 5)]   (two coalesced 128-byte segments of Mt)
//   B operand: lane l supplies q[medoid l & 31][column 2 s + (l >> 5)]         (NK registers, loaded once per kernel)
//   D: lane l holds medoid j = l & 31 against rows base + (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), reg = 0..15
// A wavefront owns 32-row tiles (grid-stride) and keeps the loads of the next two tiles in flight.  The common path after
// the chain is ONE compare per pair: d = 0.5 - dot <= 0.3 (the last histogram edge) iff dot >= a threshold found once per
// kernel (float subtraction is monotonic), OR-ed over the 16 registers on the scalar unit; liveness, lengths and the
// self-distance fix-up are looked at only for the rare pairs that pass, which go through the same per-wavefront hit queue
// and drain as in the VALU kernel.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float scan_f32x16;

template <int NK>
struct ScanTile {
    float xa[NK];
    uint4 k0, k1;       // live flags of the 32 rows (every lane loads the same 32 bytes)
};

template <int NK>
__device__ __forceinline__ void scan_tile_load(ScanTile<NK>& t, const float* __restrict__ Mt, int64_t ld, int nk,
                                               const uint8_t* __restrict__ kept, int64_t base, int j, int h) {
    t.k0 = *reinterpret_cast<const uint4*>(kept + base);
    t.k1 = *reinterpret_cast<const uint4*>(kept + base + 16);
    // scalar column-pair base + one 32-bit byte offset per lane (h ld + j < 2^30, checked by the dispatcher; the tile base is wave-uniform): the
    // NK loads of a tile share one address register
    const uint32_t off = (uint32_t)((int64_t)h * ld + j) * 4u;
    // column pairs beyond the latent width re-read the last pair: their query operand is zero, so they add exactly nothing
    // (finite x * 0 = +-0), and the loop stays free of branches
#pragma unroll
    for (int s = 0; s < NK; ++s)
        t.xa[s] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Mt + (int64_t)(2 * min(s, nk - 1)) * ld + base) + off);
}

template <int NK, bool REF = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(REF && NK <= 16 ? 3 : 1, REF && NK <= 16 ? 3 : 8))) void clu_scan_mfma_kernel(const float* __restrict__ Mt, int64_t ld, int L4,
                                                               const float* __restrict__ lengths,
                                                               const uint8_t* kept,
                                                               const float* __restrict__ q_ext, const MedoidRows medoid,
                                                               int k_real, unsigned long long* __restrict__ results,
                                                               int32_t* __restrict__ lists, int dbg, const RefSrc ro, uint8_t* kept_w,
                                                               const RmRows rm) {
    constexpr int KM = kMaxMedoids;
    apply_removals(kept_w, rm);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* hq_s = reinterpret_cast<float4*>(smem_raw);                                  // [kBlock/64][kHitCap]
    unsigned long long* acc_s = reinterpret_cast<unsigned long long*>(hq_s + (kBlock / 64) * kHitCap);   // [KM][kResultWords]
    float* edges_s = reinterpret_cast<float*>(acc_s + KM * kResultWords);               // [64]
    unsigned int* lcnt_s = reinterpret_cast<unsigned int*>(edges_s + 64);               // [KM]
    int32_t* llist_s = reinterpret_cast<int32_t*>(lcnt_s + KM);                         // [KM][kLocalCap]
    int32_t* med_s = llist_s + KM * kLocalCap;                                          // [KM]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: tile bases and live flags go scalar
    float4* hq = hq_s + wave * kHitCap;
    for (int i = tid; i < KM; i += kBlock) med_s[i] = (int32_t)medoid.row[i];
    for (int i = tid; i < KM * kResultWords; i += kBlock) acc_s[i] = 0ull;
    for (int i = tid; i <= VH_NBINS; i += kBlock) edges_s[i] = __uint_as_float(c_edge_bits[i]);
    for (int i = tid; i < KM; i += kBlock) lcnt_s[i] = 0u;
    const int j = lane & 31, h = lane >> 5;
    const int nk = L4 >> 1;
    const long long my_med = medoid.row[j];
    float qb[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int k = 2 * s + h;
        qb[s] = 0.0f;   // unused medoid slots (j >= k_real, row -1) keep a zero query: dot = 0, never of interest
        if (s < nk && j < k_real) qb[s] = q_ext ? q_ext[j * L4 + k] : Mt[(int64_t)k * ld + my_med];
    }
    __syncthreads();
    const float edge_hi = edges_s[VH_NBINS];
    // smallest dot product whose distance 0.5f - dot (float32, round to nearest) is <= the last histogram edge
    float dot_min = 0.5f - edge_hi;
    while (0.5f - __uint_as_float(__float_as_uint(dot_min) - 1u) <= edge_hi) dot_min = __uint_as_float(__float_as_uint(dot_min) - 1u);
    while (!(0.5f - dot_min <= edge_hi)) dot_min = __uint_as_float(__float_as_uint(dot_min) + 1u);
    if constexpr (REF) dot_min -= 2.0f * ro.slack;   // filter only (slack + the rounding of 0.5f - dot): the drain decides in the reference's order
    if (dbg & 1) dot_min = __builtin_inff();   // timing experiment: no pair of interest
    int qn = 0;   // hits queued by this wavefront (uniform)

    const int64_t ntiles = ld >> 5;
    const int64_t stride = (int64_t)gridDim.x * (kBlock / 64);
    int64_t tile = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    // DEPTH tile buffers: the tile being multiplied plus DEPTH - 1 in flight; a buffer is reloaded (tile + DEPTH strides)
    // as soon as its MFMA chain has been issued, before the chain's result is looked at
    constexpr int DEPTH = NK <= 16 ? 3 : 2;
    ScanTile<NK> tl[DEPTH];
    // Every prefetch is UNCONDITIONAL: past the end of the matrix the last tile is requested again and never looked at.  With the
    // request inside `if (tile + DEPTH * stride < ntiles)` the compiler cannot know how many newer loads are in flight when a
    // tile is consumed -- possibly none -- and waits for ALL of them (s_waitcnt vmcnt(0) in front of every MFMA chain, found in
    // the ISA in round 5): the three tile buffers then hide nothing and every tile pays a full memory round trip.  With a fixed
    // number of loads per iteration it waits for this tile's loads only (vmcnt = the loads of the DEPTH - 1 tiles behind it).
    auto tile_base = [&](int64_t t) { return (t < ntiles ? t : ntiles - 1) << 5; };   // (wave-uniform)
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) scan_tile_load<NK>(tl[u], Mt, ld, nk, kept, tile_base(tile + u * stride), j, h);
    auto step = [&](ScanTile<NK>& t) {
        const int64_t base = tile << 5;
        const uint32_t kw[8] = {t.k0.x, t.k0.y, t.k0.z, t.k0.w, t.k1.x, t.k1.y, t.k1.z, t.k1.w};
        const uint32_t any_live = (kw[0] | kw[1]) | (kw[2] | kw[3]) | (kw[4] | kw[5]) | (kw[6] | kw[7]);
        // a tile of already emitted rows (uniform: every lane holds the same flags) costs no matrix-pipe time
        const bool work = __builtin_amdgcn_readfirstlane(any_live) != 0u;
        scan_f32x16 acc;
        if (work) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            // column pairs beyond the latent width hold zeros on both sides: fmaf(0, 0, acc) == acc, bit for bit
            if (dbg & 16) {   // timing experiment: no matrix pipe (the operands are still consumed: one VALU op each)
#pragma unroll
                for (int q = 0; q < NK; ++q) acc[q & 15] += t.xa[q] * qb[q];
            } else {
#pragma unroll
                for (int q = 0; q < NK; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t.xa[q], qb[q], acc, 0, 0, 0);
            }
        }
        if (!(dbg & 32))   // timing experiment: 32 = no loads behind the first DEPTH tiles (matrix pipe on stale operands)
            scan_tile_load<NK>(t, Mt, ld, nk, kept, tile_base(tile + DEPTH * stride), j, h);
        if (!work) return;
        unsigned long long any = 0ull;
#pragma unroll
        for (int r = 0; r < 16; ++r) any |= __builtin_amdgcn_ballot_w64(acc[r] >= dot_min);
        if (any == 0ull) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // live flags of this half-wave's rows 8 q + 4 h .. + 3 = dword 2 q + h of the 8 flag dwords
            const unsigned long long pair = ((unsigned long long)kw[2 * q + 1] << 32) | kw[2 * q];
            const uint32_t live4 = (uint32_t)(pair >> (32 * h));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool hit = acc[4 * q + e] >= dot_min && ((live4 >> (8 * e)) & 0xFFu) != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (m != 0ull) {
                    if (hit) {
                        const int64_t row = base + 8 * q + 4 * h + e;
                        const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                        hq[pos] = make_float4(0.5f - acc[4 * q + e], 0.0f, __int_as_float((int32_t)row), __int_as_float(j));
                    }
                    qn += __popcll(m);
                    if (qn > kHitCap - 64) {
                        drain_hits<REF>(hq, qn, lane, acc_s, lcnt_s, llist_s, edges_s, med_s, dbg, ro, lengths);
                        qn = 0;
                    }
                }
            }
        }
    };
    while (tile < ntiles) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (tile < ntiles) {
                step(tl[u]);
                tile += stride;
            }
        }
    }
    drain_hits<REF>(hq, qn, lane, acc_s, lcnt_s, llist_s, edges_s, med_s, dbg, ro, lengths);
    scan_flush<KM>(tid, acc_s, lcnt_s, llist_s, results, lists, dbg);
}

// ---------------------------------------------------------------------------------------------
// K6r: the matrix-pipe pass over a ROW-major copy of the matrix (round 5; scan.reference_order = 2 only).
// What the counters said about K6m at 32 medoids x 620 k rows (profiles/r05d_pmc_scan_mfma_*, r05e_scan_core.txt): the pass with
// neither its loads nor its MFMAs still takes 19.5 of 33.9 us -- ~320 issued instructions per 32-row tile beside the 16 MFMAs:
// sixteen 4-byte loads, each with its own 64-bit address built from a column base that had been spilled to a VGPR lane, the
// live flags through vector registers, sixteen compares + ballots.  The matrix pipe is busy 22 % of the kernel.
// The column-major layout forces one load per (column pair) because a lane needs ONE float of each.  But in this mode the kernel
// is only a FILTER (the drain re-evaluates every pair that can decide differently in the reference's order), so the order in
// which the products are accumulated is free: step s of the chain multiplies column  h * NK + s  (h = lane >> 5) instead of
// 2 s + h.  Then a lane needs NK CONSECUTIVE floats of its row -- NK / 4 sixteen-byte loads from a row-major matrix
// [ld][LR] (LR = 32 or 64 floats per row, zero padded), addressed as one scalar tile base + one per-lane byte offset that
// never changes.  Live flags are fetched (scalar) only by tiles that have a candidate pair; the candidate test is a max tree +
// one compare.  Everything downstream (hit queue, drain, flush) is K6m's.
// ---------------------------------------------------------------------------------------------
typedef float scan_f32x4 __attribute__((ext_vector_type(4)));

// The tile loads and their waits are written out by hand.  Left to the compiler, every MFMA chain was preceded by
// s_waitcnt vmcnt(0) -- the hit path between two chains contains loads of its own inside branches, and the waitcnt pass then gives
// up counting across them -- so the tile buffers hid nothing (same finding as in K6m).  An asm load is invisible to that pass:
// the wait below is ours alone.  Loads return in order, so "at most (DEPTH - 1) * NK / 4 loads still in flight" means the
// tile about to be multiplied has landed whatever else was issued in between (waits the compiler adds for its own loads can
// only wait for more).  The "+v" operands tie the registers to the wait, so nothing reads them before it.
template <int NK>
__device__ __forceinline__ void scan_tile_load_rm(scan_f32x4 (&xa)[NK / 4], const float* __restrict__ Mr, int64_t tile, uint32_t lane_off) {
    const float* base = Mr + tile * (int64_t)(32 * 2 * NK);   // (wave-uniform: a scalar register pair)
    static_assert(NK == 16 || NK == 32, "row width 32 or 64");
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xa[0]) : "v"(lane_off), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(xa[1]) : "v"(lane_off), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(xa[2]) : "v"(lane_off), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(xa[3]) : "v"(lane_off), "s"(base) : "memory");
    if constexpr (NK == 32) {
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(xa[4]) : "v"(lane_off), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:80" : "=v"(xa[5]) : "v"(lane_off), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:96" : "=v"(xa[6]) : "v"(lane_off), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:112" : "=v"(xa[7]) : "v"(lane_off), "s"(base) : "memory");
    }
}
// wait until at most `behind` of OUR tile loads are still in flight, i.e. until the tile in `xa` has landed
template <int NK, int BEHIND>
__device__ __forceinline__ void scan_tile_wait_rm(scan_f32x4 (&xa)[NK / 4]) {
    if constexpr (NK == 16)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]) : "n"(BEHIND) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]),
                     "+v"(xa[7]) : "n"(BEHIND) : "memory");
}

template <int NK>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(NK <= 16 ? 3 : 2, NK <= 16 ? 3 : 8)))
void clu_scan_mfma_rm_kernel(const float* __restrict__ Mr, int64_t ld, const float* __restrict__ lengths,
                             const uint8_t* kept, const float* __restrict__ q_ext, int q_ld,
                             const MedoidRows medoid, int k_real, unsigned long long* __restrict__ results,
                             int32_t* __restrict__ lists, int dbg, const RefSrc ro, uint8_t* kept_w, const RmRows rm) {
    constexpr int KM = kMaxMedoids;
    constexpr int LR = 2 * NK;
    SCAN_STAMP(0);
    apply_removals(kept_w, rm);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float4* hq_s = reinterpret_cast<float4*>(smem_raw);                                  // [kBlock/64][kHitCap]
    unsigned long long* acc_s = reinterpret_cast<unsigned long long*>(hq_s + (kBlock / 64) * kHitCap);   // [KM][kResultWords]
    float* edges_s = reinterpret_cast<float*>(acc_s + KM * kResultWords);               // [64]
    unsigned int* lcnt_s = reinterpret_cast<unsigned int*>(edges_s + 64);               // [KM]
    int32_t* llist_s = reinterpret_cast<int32_t*>(lcnt_s + KM);                         // [KM][kLocalCap]
    int32_t* med_s = llist_s + KM * kLocalCap;                                          // [KM]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float4* hq = hq_s + wave * kHitCap;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane_off = (uint32_t)(j * LR + h * NK) * 4u;
    const int64_t ntiles = ld >> 5;
    const int64_t stride = (int64_t)gridDim.x * (kBlock / 64);
    int64_t tile = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    auto clamp_tile = [&](int64_t t) { return t < ntiles ? t : ntiles - 1; };   // (wave-uniform; see K6m on unconditional prefetches)
    constexpr int DEPTH = NK <= 16 ? 3 : 2;
    constexpr int TILE_LOADS = NK / 4;
    scan_f32x4 xa[DEPTH][NK / 4];
    // B operand: lane (medoid j, half h) holds q[j][h NK .. + NK - 1]; unused medoid slots keep zeros (dot = 0, never a candidate)
    float qb[NK];
#ifndef VAMBHIP_K6R_OLD_PROLOGUE
    // Round 6 (profiles/r06v_scan_timeline.txt: the prologue of this kernel ended 5.3 - 7.9 us after its entry, the VALU kernels' after
    // 1.1 - 1.7 us).  Loads return in order, and everything the prologue needs used to be requested BEHIND the first DEPTH tiles
    // (37 MB over the chip at 768 workgroups): the medoid rows as a per-lane read of the kernel arguments, then -- a dependent round
    // trip later -- the query vectors, then the edge table, each awaited by a compiler-made vmcnt(0) that drained the tiles as well;
    // no matrix-pipe work started before all of that had landed.  Now the medoid rows come through the scalar cache (wave-uniform
    // reads of the arguments, one select per slot), and the edge table and the query block are requested by asm loads IN FRONT of
    // the tiles: one counted wait (the DEPTH tiles issued behind them stay in flight) ends the prologue.
    int med32 = 0;   // lane (j, h): physical row of medoid slot j, -1 = empty slot (rows < 2^31, checked at creation)
#pragma unroll
    for (int i = 0; i < KM; ++i) {
        const int r = (int)medoid.row[i];   // (uniform index: s_load; one compare + select per slot)
        med32 = j == i ? r : med32;
    }
    const bool q_rows = q_ext == nullptr;        // (uniform) the queries are rows of the matrix
    const bool have_q = j < k_real && (!q_rows || med32 >= 0);
    uint32_t edge_bits;
    scan_f32x4 qv[NK / 4];
    {
        const uint32_t eoff = (uint32_t)(tid <= VH_NBINS ? tid : VH_NBINS) * 4u;
        const uint32_t* etab = c_edge_bits;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(edge_bits) : "v"(eoff), "s"(etab) : "memory");
        // (explicit queries -- row-sharded passes -- are fetched by the compiler's own loads below; the request here then reads row 0)
        const float* qsrc = Mr + (int64_t)(have_q && q_rows ? med32 : 0) * LR + h * NK;   // (a per-lane 64-bit address)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qv[0]) : "v"(qsrc) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(qv[1]) : "v"(qsrc) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(qv[2]) : "v"(qsrc) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(qv[3]) : "v"(qsrc) : "memory");
        if constexpr (NK == 32) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(qv[4]) : "v"(qsrc) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:80" : "=v"(qv[5]) : "v"(qsrc) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:96" : "=v"(qv[6]) : "v"(qsrc) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:112" : "=v"(qv[7]) : "v"(qsrc) : "memory");
        }
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) scan_tile_load_rm<NK>(xa[u], Mr, clamp_tile(tile + u * stride), lane_off);
    for (int i = tid; i < KM * kResultWords; i += kBlock) acc_s[i] = 0ull;
    for (int i = tid; i < KM; i += kBlock) lcnt_s[i] = 0u;
    // the edge word and the query block have landed once at most the DEPTH tiles requested behind them are in flight
    if constexpr (NK == 16)
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(edge_bits), "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]) : "n"(DEPTH * TILE_LOADS) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%9)" : "+v"(edge_bits), "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]), "+v"(qv[4]), "+v"(qv[5]),
                     "+v"(qv[6]), "+v"(qv[7]) : "n"(DEPTH * TILE_LOADS) : "memory");
    if (tid <= VH_NBINS) edges_s[tid] = __uint_as_float(edge_bits);
    if (tid < KM) med_s[tid] = med32;
#pragma unroll
    for (int v = 0; v < NK / 4; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) qb[4 * v + e] = have_q ? qv[v][e] : 0.0f;
    if (!q_rows) {   // (row-sharded passes; the compiler's wait in front of the first use drains the tiles too, as it always did)
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            const int k = h * NK + s;
            qb[s] = (j < k_real && k < q_ld) ? q_ext[j * q_ld + k] : 0.0f;
        }
    }
    __syncthreads();
    SCAN_STAMP(1);
    const float edge_hi = edges_s[VH_NBINS];
    // smallest dot product whose distance 0.5f - dot (float32, round to nearest) is <= the last histogram edge, minus the filter's slack
#else
    // the first tiles are requested before anything else: they travel under the LDS initialisation and the query fetch
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) scan_tile_load_rm<NK>(xa[u], Mr, clamp_tile(tile + u * stride), lane_off);
    {
        const long long my_med = medoid.row[j];
#pragma unroll
        for (int s = 0; s < NK; ++s) qb[s] = 0.0f;
        if (j < k_real) {
            if (q_ext != nullptr) {
#pragma unroll
                for (int s = 0; s < NK; ++s) {
                    const int k = h * NK + s;
                    if (k < q_ld) qb[s] = q_ext[j * q_ld + k];
                }
            } else {
                const float4* src = reinterpret_cast<const float4*>(Mr + (int64_t)my_med * LR + h * NK);
#pragma unroll
                for (int v = 0; v < NK / 4; ++v) {
                    const float4 x = src[v];
                    qb[4 * v + 0] = x.x; qb[4 * v + 1] = x.y; qb[4 * v + 2] = x.z; qb[4 * v + 3] = x.w;
                }
            }
        }
    }
    for (int i = tid; i < KM; i += kBlock) med_s[i] = (int32_t)medoid.row[i];
    for (int i = tid; i < KM * kResultWords; i += kBlock) acc_s[i] = 0ull;
    for (int i = tid; i <= VH_NBINS; i += kBlock) edges_s[i] = __uint_as_float(c_edge_bits[i]);
    for (int i = tid; i < KM; i += kBlock) lcnt_s[i] = 0u;
    __syncthreads();
    SCAN_STAMP(1);
    const float edge_hi = edges_s[VH_NBINS];
    // smallest dot product whose distance 0.5f - dot (float32, round to nearest) is <= the last histogram edge, minus the filter's slack
#endif
    float dot_min = 0.5f - edge_hi;
    while (0.5f - __uint_as_float(__float_as_uint(dot_min) - 1u) <= edge_hi) dot_min = __uint_as_float(__float_as_uint(dot_min) - 1u);
    while (!(0.5f - dot_min <= edge_hi)) dot_min = __uint_as_float(__float_as_uint(dot_min) + 1u);
    dot_min -= 2.0f * ro.slack;
    if (dbg & 1) dot_min = __builtin_inff();   // timing experiment: no pair of interest
    int qn = 0;   // hits queued by this wavefront (uniform)
    // The query registers must be KNOWN complete before the loop: otherwise the compiler keeps them "maybe pending" across the
    // back edge and puts its own s_waitcnt vmcnt(0) in front of every chain -- which also drains our tile loads.
#pragma unroll
    for (int s = 0; s < NK; ++s) asm volatile("" ::"v"(qb[s]));

    // Before any code the compiler is free to allocate registers for (the drain: up to +60 VGPRs) runs, NOTHING of ours is in flight:
    // an asm load the compiler does not know about must never be pending while it shuffles registers.  (The drain starts with a
    // load + wait of its own, which happens to drain the queue as well; this makes it deliberate.  Rare path: a drain per ~190 pairs.)
    auto settle = [&]() {
        if constexpr (NK == 16)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[0][2]), "+v"(xa[0][3]), "+v"(xa[1][0]), "+v"(xa[1][1]),
                         "+v"(xa[1][2]), "+v"(xa[1][3]), "+v"(xa[DEPTH - 1][0]), "+v"(xa[DEPTH - 1][1]), "+v"(xa[DEPTH - 1][2]),
                         "+v"(xa[DEPTH - 1][3]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[0][2]), "+v"(xa[0][3]), "+v"(xa[0][4]), "+v"(xa[0][5]),
                         "+v"(xa[0][6]), "+v"(xa[0][7]), "+v"(xa[DEPTH - 1][0]), "+v"(xa[DEPTH - 1][1]), "+v"(xa[DEPTH - 1][2]),
                         "+v"(xa[DEPTH - 1][3]), "+v"(xa[DEPTH - 1][4]), "+v"(xa[DEPTH - 1][5]), "+v"(xa[DEPTH - 1][6]),
                         "+v"(xa[DEPTH - 1][7]) : : "memory");
    };
    auto step = [&](scan_f32x4 (&t)[NK / 4]) {
        const int64_t base = tile << 5;
        scan_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        scan_tile_wait_rm<NK, (DEPTH - 1) * TILE_LOADS>(t);   // this tile has landed; the DEPTH - 1 behind it may still be in flight
#pragma unroll
        for (int q = 0; q < NK; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(t[q >> 2][q & 3], qb[q], acc, 0, 0, 0);
        scan_tile_load_rm<NK>(t, Mr, clamp_tile(tile + DEPTH * stride), lane_off);
        // candidate test: the largest of the lane's 16 dot products against the threshold, one ballot
        float m0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), m1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]);
        float m2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]), m3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]);
        float m4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]);
        m0 = fmaxf(fmaxf(m0, m1), m2);
        m3 = fmaxf(fmaxf(m3, m4), acc[15]);
        if (__builtin_amdgcn_ballot_w64(fmaxf(m0, m3) >= dot_min) == 0ull) return;
        // live flags of the tile's 32 rows: the same 32 bytes for every lane (scalar loads), looked at by candidate tiles only
        const uint32_t* kp = reinterpret_cast<const uint32_t*>(kept + base);
        uint32_t kw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kw[i] = __builtin_amdgcn_readfirstlane(kp[i]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // live flags of this half-wave's rows 8 q + 4 h .. + 3 = dword 2 q + h of the 8 flag dwords
            const uint32_t live4 = h ? kw[2 * q + 1] : kw[2 * q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool hit = acc[4 * q + e] >= dot_min && ((live4 >> (8 * e)) & 0xFFu) != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (m != 0ull) {
                    if (hit) {
                        const int64_t row = base + 8 * q + 4 * h + e;
                        const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                        // (the drain fetches the row's length: taking it from a per-tile prefetch instead measured no faster)
                        hq[pos] = make_float4(0.5f - acc[4 * q + e], 0.0f, __int_as_float((int32_t)row), __int_as_float(j));
                    }
                    qn += __popcll(m);
                    if (qn > kHitCap - 64) {
                        settle();
                        drain_hits<true>(hq, qn, lane, acc_s, lcnt_s, llist_s, edges_s, med_s, dbg, ro, lengths);
                        qn = 0;
                    }
                }
            }
        }
    };
    while (tile < ntiles) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (tile < ntiles) {
                step(xa[u]);
                tile += stride;
            }
        }
    }
    settle();
    SCAN_STAMP(3);
    drain_hits<true>(hq, qn, lane, acc_s, lcnt_s, llist_s, edges_s, med_s, dbg, ro, lengths);
    SCAN_STAMP(4);
    scan_flush<KM>(tid, acc_s, lcnt_s, llist_s, results, lists, dbg);
}

// Mr[i][0 .. LR) = the normalised row i (L floats) + zero padding, rows >= n all zero
__global__ __launch_bounds__(kBlock) void clu_rows_pad_kernel(const float* __restrict__ rows, int64_t n, int L,
                                                              float* __restrict__ Mr, int64_t ld, int LR) {
    const int64_t total = ld * (int64_t)LR;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t i = idx / LR;
        const int c = (int)(idx - i * LR);
        Mr[idx] = (i < n && c < L) ? rows[i * (int64_t)L + c] : 0.0f;
    }
}

// compaction of the row-major copy: out[i][:] = in[src[i]][:] for i < n_new, zero rows behind (16 bytes per thread)
__global__ __launch_bounds__(kBlock) void clu_gather_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 const int32_t* __restrict__ src, int64_t n_new,
                                                                 int64_t ld_out, int LR) {
    const int q4 = LR / 4;
    const int64_t total = ld_out * (int64_t)q4;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t i = idx / q4;
        const int c = (int)(idx - i * q4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n_new) v = reinterpret_cast<const float4*>(in + (int64_t)src[i] * LR)[c];
        reinterpret_cast<float4*>(out + i * (int64_t)LR)[c] = v;
    }
}

// Query vectors of a many-medoid pass in quad-major order [L4 / 4][km][4] for the scalar loads of the pipelined scan
// kernel: from the resident rows medoid.row[j], or from explicit vectors q_src[km][L4] (row-sharded execution).
__global__ __launch_bounds__(kBlock) void clu_gather_quads_kernel(const float* __restrict__ Mt, int64_t ld, int L4,
                                                                  const float* __restrict__ q_src, const MedoidRows medoid,
                                                                  int km, float* __restrict__ qp) {
    for (int i = threadIdx.x; i < km * L4; i += kBlock) {
        const int j = i / L4, c = i - j * L4;
        const float v = q_src ? q_src[i] : Mt[(int64_t)c * ld + medoid.row[j]];
        qp[((c >> 2) * km + j) * 4 + (c & 3)] = v;
    }
}

// Row-sharded execution: the accumulator copies of a pass are folded into copy 0 before the all-reduce over the ranks
__global__ __launch_bounds__(kBlock) void clu_fold_replicas_kernel(int km, unsigned long long* __restrict__ results) {
    for (int i = threadIdx.x; i < km * kResultWords; i += kBlock) {
        unsigned long long v = results[i];
        for (int r = 1; r < kResultReplicas; ++r) {
            v += results[(size_t)r * kMaxMedoids * kResultWords + i];
            results[(size_t)r * kMaxMedoids * kResultWords + i] = 0ull;
        }
        results[i] = v;
    }
}

// Row-sharded execution: query vectors of the medoids THIS rank owns (medoid.row[j] >= 0), zeros for the others; the sum
// all-reduce over the ranks distributes every vector exactly (x + 0 + ... + 0).
__global__ __launch_bounds__(kBlock) void clu_gather_owned_kernel(const float* __restrict__ Mt, int64_t ld, int L4,
                                                                  const MedoidRows medoid, int km, float* __restrict__ q) {
    for (int i = threadIdx.x; i < km * L4; i += kBlock) {
        const int j = i / L4, c = i - j * L4;
        const long long row = medoid.row[j];
        q[i] = row >= 0 ? Mt[(int64_t)c * ld + row] : 0.0f;
    }
}

constexpr int kPublishThreads = 256;    // ONE workgroup (the flag must follow every write).  Measured, C1 sweep under rocprofv3:
                                        // 256 threads + list copy 8.4 us average / 2.9 us minimum per pass; 1024 threads 9.1 /
                                        // 5.4 us (a 16-wavefront workgroup starts later and 1024 threads sit in the system fence).
                                        // Round 6, behind passes with 32 medoids (in-kernel stamps, profiles/r06y3_publish_width.txt):
                                        // 256 / 512 / 1024 threads form the sums in 5.6 / 4.4 / 4.2 us and then need 2.9 / 3.8 / 8.0 us
                                        // to get them into host memory behind the system fence; a C2 sweep 12.0 / 12.3 / 12.6 s.
// ---- row-sharded pass of the NATIVE state machine: ONE collective per pass --------------------------------------------------
// Every rank scans its shard with explicit query vectors (each rank holds a host copy of the whole normalised matrix for
// the generator's validity checks, so no query exchange is needed) and then contributes ONE block to an all-gather:
//   [k][63] u64   the exact integer accumulators of its shard (density, 60 bins, n_within, n_lt), replica copies folded
//   [k][1 + kXListCap] u32   its part of every medoid's within-radius list as GLOBAL physical rows (count, rows), or the
//                            count 0xFFFFFFFF when the part is incomplete / longer than kXListCap (the caller then selects)
// The publishing kernel adds the accumulators of all ranks (integer sums: order-free, so the result does not depend on the
// sharding) and hands accumulators and list parts to the host through mapped memory.
constexpr int kXListCap = 64;
constexpr int kXAccWords = 2 * (kResultWords - 1);              // u32 words of one medoid's 63 accumulators
constexpr int kXMedWords = kXAccWords + 1 + kXListCap;         // u32 words of one medoid in a block
// k: medoids of the pass (the block holds these); k_slots >= k: medoid slots the scan kernel accumulated into (a VALU pass
// pads its medoid-count bucket with copies of medoid 0): all of them are re-zeroed for the next pass.
__global__ __launch_bounds__(kBlock) void clu_xblock_kernel(int k, int k_slots, unsigned long long* __restrict__ results,
                                                            const int32_t* __restrict__ lists, uint32_t row_offset,
                                                            uint32_t* __restrict__ block) {
    __shared__ unsigned long long nwithin_s[kMaxMedoids];
    __shared__ unsigned int cursor_s[kMaxMedoids];
    const int tid = threadIdx.x;
    for (int i = tid; i < k_slots * kResultWords; i += kBlock) {
        const int j = i / kResultWords, w = i - j * kResultWords;
        unsigned long long v = 0ull;
        for (int r = 0; r < kResultReplicas; ++r) {
            const size_t at = (size_t)r * kMaxMedoids * kResultWords + i;
            const unsigned long long x = results[at];
            v += x;
            if (x != 0ull) results[at] = 0ull;
        }
        if (j >= k) continue;
        if (w == kResultWords - 1) { cursor_s[j] = (unsigned int)v; continue; }   // (the list cursor lives in copy 0 only)
        if (w == 1 + VH_NBINS) nwithin_s[j] = v;
        block[(size_t)j * kXMedWords + 2 * w] = (uint32_t)v;
        block[(size_t)j * kXMedWords + 2 * w + 1] = (uint32_t)(v >> 32);
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        uint32_t* out = block + (size_t)j * kXMedWords + kXAccWords;
        const unsigned long long nw = nwithin_s[j];
        const bool ok = (unsigned long long)cursor_s[j] == nw && nw <= (unsigned long long)kXListCap;
        if (tid == 0) out[0] = ok ? (uint32_t)nw : 0xFFFFFFFFu;
        if (ok && tid < (int)nw) out[1 + tid] = (uint32_t)lists[j * kListCap + tid] + row_offset;
    }
}

__global__ __launch_bounds__(kPublishThreads) void clu_publish_sharded_kernel(int k, int world, const uint32_t* __restrict__ gathered,
                                                                              size_t block_words,
                                                                              unsigned long long* __restrict__ host_summary,
                                                                              unsigned long long* __restrict__ host_hist,
                                                                              uint32_t* __restrict__ host_lists,
                                                                              unsigned long long* __restrict__ host_flag,
                                                                              unsigned long long seq) {
    const int tid = threadIdx.x;
    for (int i = tid; i < k * (kResultWords - 1); i += kPublishThreads) {
        const int j = i / (kResultWords - 1), w = i - j * (kResultWords - 1);
        unsigned long long v = 0ull;
        for (int r = 0; r < world; ++r) {
            const uint32_t* b = gathered + (size_t)r * block_words + (size_t)j * kXMedWords + 2 * w;
            v += (unsigned long long)b[0] | ((unsigned long long)b[1] << 32);
        }
        if (w == 0) host_summary[4 * j + 0] = v;
        else if (w <= VH_NBINS) host_hist[(size_t)j * VH_NBINS + (w - 1)] = v;
        else host_summary[4 * j + (w - VH_NBINS)] = v;   // n_within -> 1, n_lt -> 2
    }
    // list parts: [rank][medoid][1 + kXListCap]
    for (int i = tid; i < world * k * (1 + kXListCap); i += kPublishThreads) {
        const int e = i % (1 + kXListCap), rj = i / (1 + kXListCap);
        const int j = rj % k, r = rj / k;
        host_lists[i] = gathered[(size_t)r * block_words + (size_t)j * kXMedWords + kXAccWords + e];
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {   // (host_flag + 1: the histogram flag of vh_clu::hist_flag(), raised with the pass flag here)
        __hip_atomic_store(host_flag + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// K6b: publication without a copy-engine round trip.  One block moves the accumulators and the candidate
// lists into host-mapped memory, zeroes the accumulators for the next scan and then raises the sequence flag
// the host spins on.  (A separate launch, not a last-block-done tail of the scan: the kernel boundary is the
// cheap way to make the other XCDs' L2 contents visible -- a per-block agent-scope fence writes L2 back and
// made the scan 5x slower.)
__global__ __launch_bounds__(kPublishThreads) void clu_publish_kernel(int km, unsigned long long* __restrict__ results,
                                                                      unsigned long long* __restrict__ host_summary,
                                                                      unsigned long long* __restrict__ host_hist,
                                                                      unsigned long long* __restrict__ host_flag,
                                                                      unsigned long long* __restrict__ host_hist_flag,
                                                                      unsigned long long seq, int dbg) {
    const int tid = threadIdx.x;
    PUBLISH_STAMP(0);
    // the accumulator copies of the pass are added up first (and zeroed for the next pass)
    __shared__ unsigned long long red_s[kMaxMedoids * kResultWords];
#pragma unroll 4
    for (int i = tid; i < km * kResultWords; i += kPublishThreads) {   // up to 4 x 8 independent loads in flight per thread
        unsigned long long part[kResultReplicas];
#pragma unroll
        for (int r = 0; r < kResultReplicas; ++r) part[r] = results[(size_t)r * kMaxMedoids * kResultWords + i];
        unsigned long long v = 0ull;
#pragma unroll
        for (int r = 0; r < kResultReplicas; ++r) v += part[r];
        red_s[i] = v;
#pragma unroll
        for (int r = 0; r < kResultReplicas; ++r)
            if (part[r] != 0ull) results[(size_t)r * kMaxMedoids * kResultWords + i] = 0ull;
    }
    __syncthreads();
    PUBLISH_STAMP(1);
    // (the candidate lists are already in the host-mapped ring: the scan kernels append them there)
    // the four words every candidate needs (density, n_within, n_lt, list cursor) go to a compact block of
    // their own: the host reads 32 bytes per medoid instead of 512 (host reads of this memory are expensive)
    for (int i = tid; i < km * 4; i += kPublishThreads) {
        const int j = i >> 2, w = i & 3;
        host_summary[i] = red_s[j * kResultWords + (w == 0 ? 0 : VH_NBINS + w)];
    }
    if (!(dbg & 32))
        for (int i = tid; i < km * VH_NBINS; i += kPublishThreads) {
            const int j = i / VH_NBINS, b = i - j * VH_NBINS;
            host_hist[i] = red_s[j * kResultWords + 1 + b];
        }
    __threadfence_system();
    __syncthreads();
    PUBLISH_STAMP(2);
    if (tid == 0) {   // (the histogram flag first: whoever has seen the pass flag may read the histograms)
        __hip_atomic_store(host_hist_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same in TWO steps (round 6; the default).  In-kernel stamps (profiles/r06v_scan_timeline.txt) put the publish kernel of a
// 32-medoid pass at 9.2 us: 5.6 us until the eight copies of 32 x 64 words are added up, 2.9 us until they are in host memory.  What
// the state machine needs at once are the FOUR summary words of a medoid (density, two counts, list cursor); the 60 histogram words
// are read for the one medoid a cluster is built around, later, if at all.  So: the summary words first (up to 128 threads, one round
// of eight loads each), fence, sequence flag -- then the histograms and the re-arming of the copies, fence, a second flag (hist_flag)
// that the readers of the histogram ring wait for (wait_for_hist).  The next scan is ordered behind this kernel by the stream, whatever
// the host does in between.  Measured per C2 sweep on one box: -0.75 s behind the passes with more than 8 medoids, another -0.24 s
// behind the others (profiles/r06y4_sweep_publish_split.txt, r06y5_sweep_publish_split_all.txt).
__global__ __launch_bounds__(kPublishThreads) void clu_publish2_kernel(int km, unsigned long long* __restrict__ results,
                                                                       unsigned long long* __restrict__ host_summary,
                                                                       unsigned long long* __restrict__ host_hist,
                                                                       unsigned long long* __restrict__ host_flag,
                                                                       unsigned long long* __restrict__ host_hist_flag,
                                                                       unsigned long long seq, int dbg) {
    const int tid = threadIdx.x;
    PUBLISH_STAMP(0);
    auto fold = [&](int word) {   // sum of the copies of one accumulator word, the copies re-armed
        unsigned long long part[kResultReplicas];
#pragma unroll
        for (int r = 0; r < kResultReplicas; ++r) part[r] = results[(size_t)r * kMaxMedoids * kResultWords + word];
        unsigned long long v = 0ull;
#pragma unroll
        for (int r = 0; r < kResultReplicas; ++r) v += part[r];
#pragma unroll
        for (int r = 0; r < kResultReplicas; ++r)
            if (part[r] != 0ull) results[(size_t)r * kMaxMedoids * kResultWords + word] = 0ull;
        return v;
    };
    for (int i = tid; i < km * 4; i += kPublishThreads) {
        const int j = i >> 2, w = i & 3;
        host_summary[i] = fold(j * kResultWords + (w == 0 ? 0 : VH_NBINS + w));
    }
    PUBLISH_STAMP(1);
    __threadfence_system();
    __syncthreads();
    PUBLISH_STAMP(2);
    if (tid == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll 4
    for (int i = tid; i < km * VH_NBINS; i += kPublishThreads) {
        const int j = i / VH_NBINS, b = i - j * VH_NBINS;
        const unsigned long long v = fold(j * kResultWords + 1 + b);
        if (!(dbg & 32)) host_hist[i] = v;
    }
    __threadfence_system();
    __syncthreads();
    PUBLISH_STAMP(3);
    if (tid == 0) __hip_atomic_store(host_hist_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// K7a: select -- unordered append of live rows with d <= threshold (host sorts; lists are short),
// optionally clearing their live flag.  Same fmaf chain as the scan => identical distances.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void clu_select_kernel(const float* __restrict__ Mt, int64_t ld, int L4,
                                                            uint8_t* __restrict__ kept, int64_t n,
                                                            const float* __restrict__ q_ext, int64_t medoid,
                                                            float threshold, int remove,
                                                            int32_t* __restrict__ out_rows,
                                                            unsigned int* __restrict__ out_count, int ref_L, float ref_slack,
                                                            int ref_all, const RmRows rm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* q_s = reinterpret_cast<float*>(smem_raw);
    __shared__ int32_t med_one[1];   // (the reference-order re-evaluation takes medoid rows from an array)
    const int tid = threadIdx.x;
    apply_removals(kept, rm);   // rows the state machine removed since the last pass (as in the scans: in front of the prologue's barrier)
    for (int i = tid; i < L4; i += kBlock) q_s[i] = q_ext ? q_ext[i] : Mt[(int64_t)i * ld + medoid];
    if (tid == 0) med_one[0] = (int32_t)medoid;
    __syncthreads();
    const int lane = tid & 63;

    for (int64_t base = ((int64_t)blockIdx.x * kBlock + tid) * kRowsPerThread; base < n;
         base += (int64_t)gridDim.x * kRowsPerBlock) {
        uchar4 kp = *reinterpret_cast<const uchar4*>(kept + base);
        if (__ballot((kp.x | kp.y | kp.z | kp.w) != 0) == 0ull) continue;
        float acc[kRowsPerThread] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* col = Mt + base;
        for (int c = 0; c < L4; c += 4) {
            const float4 x0 = *reinterpret_cast<const float4*>(col + (int64_t)(c + 0) * ld);
            const float4 x1 = *reinterpret_cast<const float4*>(col + (int64_t)(c + 1) * ld);
            const float4 x2 = *reinterpret_cast<const float4*>(col + (int64_t)(c + 2) * ld);
            const float4 x3 = *reinterpret_cast<const float4*>(col + (int64_t)(c + 3) * ld);
            const float4 qq = *reinterpret_cast<const float4*>(q_s + c);
            acc[0] = __builtin_fmaf(x0.x, qq.x, acc[0]);
            acc[1] = __builtin_fmaf(x0.y, qq.x, acc[1]);
            acc[2] = __builtin_fmaf(x0.z, qq.x, acc[2]);
            acc[3] = __builtin_fmaf(x0.w, qq.x, acc[3]);
            acc[0] = __builtin_fmaf(x1.x, qq.y, acc[0]);
            acc[1] = __builtin_fmaf(x1.y, qq.y, acc[1]);
            acc[2] = __builtin_fmaf(x1.z, qq.y, acc[2]);
            acc[3] = __builtin_fmaf(x1.w, qq.y, acc[3]);
            acc[0] = __builtin_fmaf(x2.x, qq.z, acc[0]);
            acc[1] = __builtin_fmaf(x2.y, qq.z, acc[1]);
            acc[2] = __builtin_fmaf(x2.z, qq.z, acc[2]);
            acc[3] = __builtin_fmaf(x2.w, qq.z, acc[3]);
            acc[0] = __builtin_fmaf(x3.x, qq.w, acc[0]);
            acc[1] = __builtin_fmaf(x3.y, qq.w, acc[1]);
            acc[2] = __builtin_fmaf(x3.z, qq.w, acc[2]);
            acc[3] = __builtin_fmaf(x3.w, qq.w, acc[3]);
        }
        float dist[kRowsPerThread];
#pragma unroll
        for (int r = 0; r < kRowsPerThread; ++r) dist[r] = 0.5f - acc[r];
        if (ref_L > 0) {
            // scan.reference_order: the decision d <= threshold in the reference build's order (ref_L = latent width).  The two
            // orders differ by less than ref_slack: only a row that close to the threshold is evaluated again.
#pragma unroll
            for (int r = 0; r < kRowsPerThread; ++r) {
                const bool need = ref_all || __builtin_fabsf(dist[r] - threshold) <= ref_slack;
                if (__builtin_amdgcn_ballot_w64(need) != 0ull)   // (uniform: every lane of the wavefront is in this loop)
                    dist[r] = ref_recheck_wave(need, dist[r], (int32_t)(base + r), 0, Mt, ld, ref_L, q_ext, 0, med_one);
            }
        }
        const unsigned char live[4] = {kp.x, kp.y, kp.z, kp.w};
        unsigned char newlive[4] = {kp.x, kp.y, kp.z, kp.w};
        bool any_removed = false;
#pragma unroll
        for (int r = 0; r < kRowsPerThread; ++r) {
            float d = dist[r];
            if (base + r == medoid) d = 0.0f;
            const bool hit = live[r] && (d <= threshold);
            const unsigned long long ball = __ballot(hit);
            if (ball != 0ull) {
                const int leader = __ffsll((long long)ball) - 1;
                unsigned int start = 0;
                if (lane == leader) start = atomicAdd(out_count, (unsigned int)__popcll(ball));
                start = __shfl(start, leader);
                if (hit) {
                    const unsigned long long below = ball & ((1ull << lane) - 1ull);
                    out_rows[start + (unsigned int)__popcll(below)] = (int32_t)(base + r);
                    if (remove) { newlive[r] = 0; any_removed = true; }
                }
            }
        }
        if (any_removed) {
            uchar4 w;
            w.x = newlive[0]; w.y = newlive[1]; w.z = newlive[2]; w.w = newlive[3];
            *reinterpret_cast<uchar4*>(kept + base) = w;
        }
    }
}

// kept[row] = 0 for up to 32 rows passed in the kernel arguments (no upload, no host synchronisation)
__global__ void clu_remove_args_kernel(uint8_t* __restrict__ kept, const MedoidRows rows, int n) {
    if ((int)threadIdx.x < n) kept[rows.row[threadIdx.x]] = 0;
}

// publication of a select pass: count + the first kSelHostCap rows into host-mapped memory, counter re-armed, then the
// sequence flag the host spins on (same scheme as clu_publish_kernel: no copy-engine round trip, no event wait)
constexpr int kSelHostCap = 4096;
__global__ __launch_bounds__(kBlock) void clu_publish_select_kernel(unsigned int* __restrict__ count,
                                                                    const int32_t* __restrict__ rows,
                                                                    int32_t* __restrict__ host_rows,
                                                                    unsigned long long* __restrict__ host_meta,
                                                                    unsigned long long seq) {
    const unsigned int cnt = *count;
    const unsigned int ncopy = cnt < (unsigned int)kSelHostCap ? cnt : (unsigned int)kSelHostCap;
    for (unsigned int i = threadIdx.x; i < ncopy; i += kBlock) host_rows[i] = rows[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        host_meta[0] = cnt;
        *count = 0u;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_meta + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void clu_remove_kernel(uint8_t* __restrict__ kept, const int64_t* __restrict__ rows, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) kept[rows[i]] = 0;
}

// ---------------------------------------------------------------------------------------------
// K7b: order-preserving compaction (pack).  count -> single-block exclusive scan -> source list
// -> column gather.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void clu_count_kept_kernel(const uint8_t* __restrict__ kept, int64_t n_pad,
                                                                unsigned int* __restrict__ block_counts) {
    __shared__ unsigned int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kRowsPerThread;
    unsigned int c = 0;
    if (base < n_pad) {
        const uchar4 kp = *reinterpret_cast<const uchar4*>(kept + base);
        c = (kp.x != 0) + (kp.y != 0) + (kp.z != 0) + (kp.w != 0);
    }
    // wave reduce then one LDS atomic per wave
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s, c);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s;
}

// single block: exclusive scan of nb counts in place; total written to *total
__global__ __launch_bounds__(1024) void clu_exclusive_scan_kernel(unsigned int* __restrict__ counts, int nb,
                                                                  unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = t * per, hi = min(nb, lo + per);
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = part[t];
    for (int i = lo; i < hi; ++i) { const unsigned int v = counts[i]; counts[i] = (unsigned int)run; run += v; }
}

// src[new] = old physical row, ascending.  Block b owns rows [b*1024, b*1024+1024).
__global__ __launch_bounds__(kBlock) void clu_build_src_kernel(const uint8_t* __restrict__ kept, int64_t n_pad,
                                                               const unsigned int* __restrict__ block_offsets,
                                                               int32_t* __restrict__ src) {
    __shared__ unsigned int wave_tot[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = ((int64_t)blockIdx.x * kBlock + tid) * kRowsPerThread;
    uchar4 kp = make_uchar4(0, 0, 0, 0);
    if (base < n_pad) kp = *reinterpret_cast<const uchar4*>(kept + base);
    const unsigned int c = (kp.x != 0) + (kp.y != 0) + (kp.z != 0) + (kp.w != 0);
    // inclusive wave prefix
    unsigned int inc = c;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int v = __shfl_up(inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned int wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += wave_tot[w];
    unsigned int pos = block_offsets[blockIdx.x] + wave_base + inc - c;
    if (kp.x) src[pos++] = (int32_t)(base + 0);
    if (kp.y) src[pos++] = (int32_t)(base + 1);
    if (kp.z) src[pos++] = (int32_t)(base + 2);
    if (kp.w) src[pos++] = (int32_t)(base + 3);
}

// out[c][i] = in[c][src[i]] for i < n_new, 0 for the padding; blockIdx.y = column (L4 = matrix, +1 = lengths)
__global__ __launch_bounds__(kBlock) void clu_gather_columns_kernel(const float* __restrict__ in, int64_t ld_in,
                                                                    float* __restrict__ out, int64_t ld_out,
                                                                    const int32_t* __restrict__ src, int64_t n_new,
                                                                    const float* __restrict__ len_in,
                                                                    float* __restrict__ len_out, int L4) {
    const int c = blockIdx.y;
    const float* pin = (c < L4) ? in + (int64_t)c * ld_in : len_in;
    float* pout = (c < L4) ? out + (int64_t)c * ld_out : len_out;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ld_out; i += (int64_t)gridDim.x * kBlock)
        pout[i] = (i < n_new) ? pin[src[i]] : 0.0f;
}

__global__ void clu_fill_kept_kernel(uint8_t* __restrict__ kept, int64_t n_live, int64_t n_pad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x)
        kept[i] = (i < n_live) ? 1 : 0;
}

int pick_km(int k) {
    static const int sizes[] = {1, 2, 4, 8, 12, 16, 24, 32};
    for (int s : sizes)
        if (k <= s) return s;
    return -1;
}

int scan_grid(int64_t n) {
    const int64_t blocks = ceil_div(n, kRowsPerBlock);
    return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, 256 * 8));
}

}  // namespace

struct vh_clu {
    int L = 0, L4 = 0;
    int64_t n_rows = 0;   // physical rows
    int64_t n_live = 0;
    int64_t ld = 0;       // leading dimension (>= n_rows, multiple of 1024)
    hipStream_t stream = nullptr;
    DevBuf<float> Mt, Mt_alt, lengths, lengths_alt, q, qp;   // qp: quad-major queries of a many-medoid pass
    DevBuf<float> Mr, Mr_alt;     // row-major copy [ld][LR] of the matrix for the row-major matrix-pipe pass (K6r), or empty
    int LR = 0;                   // its row width: 32 or 64 floats (zero padded); 0 = no copy (K6m serves the many-medoid passes)
    DevBuf<uint8_t> kept;
    std::vector<int32_t> pend_rm;   // rows removed on the host side whose flags are still set on the device (see RmRows)
    RmRows rm_pass{};               // ... the part of them the pass being launched applies
    DevBuf<unsigned long long> results;
    // host-mapped (pinned, coherent) publication buffers written by the scan kernel itself
    int32_t* lists = nullptr;     // [kListRing][kMaxMedoids][kListCap] rows within the medoid radius, per scan
    int32_t* lists_pass = nullptr;   // ring slot of the running pass: the scan kernels append the candidate lists straight into
                                     // host-mapped memory (posted PCIe writes spread over the scan; the publish kernel's
                                     // copy of them was 4.5 of its 10 us, profiles/run_publish_ablation.sh)
    unsigned long long* host_results = nullptr;   // [kListRing][kMaxMedoids][4] summaries, [kListRing][kMaxMedoids]
                                                  // [VH_NBINS] histograms, 1 flag word (offsets below)
    unsigned long long* summary(int slot) { return host_results + (size_t)slot * kMaxMedoids * 4; }
    unsigned long long* hist(int slot) {
        return host_results + (size_t)kListRing * kMaxMedoids * 4 + (size_t)slot * kMaxMedoids * VH_NBINS;
    }
    unsigned long long* flag() { return host_results + (size_t)kListRing * kMaxMedoids * (4 + VH_NBINS); }
    unsigned long long* hist_flag() { return flag() + 1; }   // sequence number of the last pass whose HISTOGRAMS are in the ring (clu_publish2_kernel)
    int scan_dbg = 0;             // VAMBHIP_SCAN_DBG: timing experiments only (wrong results)
    bool publish_split = true;    // option scan.publish_split: a pass publishes its summary words first, its histograms behind a second flag
                                  // (clu_publish2_kernel); 0 = everything in one step (clu_publish_kernel)
    int max_k = kMaxMedoids;      // medoids per pass the LDS can hold for this latent width (query vectors are staged there)
    vh_comm* comm = nullptr;      // row-sharded execution (vh_clu_attach_comm): the ranks holding the other shards
    int64_t max_shard_ld = 0;     // largest padded shard over the ranks (size of the select exchange buffers)
    DevBuf<uint32_t> xch_counts, xch_rows;
    // one-collective sharded pass (native state machine): scan lists stay on the device, blocks are exchanged
    DevBuf<int32_t> lists_dev;        // [kMaxMedoids][kListCap] local rows within the radius, appended by the scan kernels
    DevBuf<uint32_t> xsend, xrecv;    // this rank's block / all ranks' blocks
    uint32_t* xlists_host = nullptr;  // host-mapped [world][k][1 + kXListCap] list parts of the last sharded pass
    bool ref_filter = false;      // scan.reference_order = 2: the same arithmetic, the tuned kernels as a filter + ref_dot in their drain
    const float* q_rows_pass = nullptr;   // explicit row-major query vectors of the running pass (or nullptr)
    bool ref_order = false;       // scan.reference_order = 1: distances and normalisation in the reference build's evaluation order
                                  // (ref_dot / ref_norm; the scan runs on clu_scan_ref_kernel)
    bool use_mfma = true;         // scan.mfma = 0: passes with more than 8 medoids stay on the VALU kernels (A/B)
    bool mfma_pass = false;       // set by scan_core for the pass being launched
    int mfma_k = 0;               // its medoid count
    int scan_lc = 1;              // column-loop variant (VAMBHIP_SCAN_LC, A/B measurements): 0 runtime-width loop everywhere,
                                  // 1 unrolled loads up to 8 medoids + pipelined query / row fetches from 12 medoids
    int64_t min_blocks = kMinScanBlocks;   // option scan.min_blocks: workgroups wanted before lanes take more than one row
    bool small_rpt = true;        // VAMBHIP_SCAN_WIDE=1 disables the narrow variant (A/B measurements)
    uint64_t scan_seq = 0;        // number of scans issued; scan s wrote ring slot s % kListRing
    int last_k = 0;
    std::vector<unsigned int> last_counts[kListRing];   // list lengths of the scans still in the ring
    std::vector<unsigned long long> last_summary[kListRing];   // (density, n_within, n_lt, cursor) per medoid
    hipEvent_t ev_done = nullptr; // recorded after the result copy: the host waits on it, not on the stream
    DevBuf<int32_t> sel_rows;   // select output / pack source list
    DevBuf<unsigned int> counts;  // [0]: select counter; [1..]: pack block counts
    DevBuf<unsigned long long> total;
    DevBuf<int64_t> row_idx;
    PinnedBuf<unsigned long long> h_results;
    PinnedBuf<float> h_q;
    std::vector<int32_t> h_sel;
    EventTimer timer;
    // select publication (host-mapped): rows [kSelHostCap], meta {count, sequence flag}
    int32_t* sel_host = nullptr;
    unsigned long long* sel_meta = nullptr;
    uint64_t sel_seq = 0;
    // host copy of the normalised matrix, row-major, by ORIGINAL row (the generator's validity checks of
    // speculatively scanned seeds; never used for a decision that reaches the output)
    std::vector<float> host_rows;
#ifdef VAMBHIP_TIMING_EXPERIMENTS
    DevBuf<unsigned long long> stamps;   // [kStampRows][8] phase stamps of the last pass (vh_debug_scan_timeline)
    double host_us[4] = {0, 0, 0, 0};    // scan_core of the last pass: entry -> scan launched -> publish launched -> flag seen
    int last_grid = 0;
#endif

    ~vh_clu() {
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (lists) (void)hipHostFree(lists);
        if (host_results) (void)hipHostFree(host_results);
        if (sel_host) (void)hipHostFree(sel_host);
        if (sel_meta) (void)hipHostFree(sel_meta);
        if (xlists_host) (void)hipHostFree(xlists_host);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

// Spin on the sequence flag the scan kernel's last block stores into host-mapped memory (a few hundred ns
// after the kernel retires, against ~10 us for a copy + event wait).  The stream is polled now and then so
// that a faulted kernel surfaces as an error instead of a hang.
// every row waiting in h->pend_rm through the remove kernel (<= 32 rows per launch, in the kernel arguments)
void flush_pending_rm(vh_clu* h) {
    const size_t n = h->pend_rm.size();
    for (size_t lo = 0; lo < n; lo += kMaxMedoids) {
        const int k = (int)std::min<size_t>(kMaxMedoids, n - lo);
        MedoidRows mr;
        for (int j = 0; j < kMaxMedoids; ++j) mr.row[j] = h->pend_rm[lo + (size_t)(j < k ? j : 0)];
        hipLaunchKernelGGL(clu_remove_args_kernel, dim3(1), dim3(64), 0, h->stream, h->kept.p, mr, k);
        VH_HIP(hipGetLastError());
    }
    h->pend_rm.clear();
}
// ... or, up to kRmCap of them, in the arguments of the scan about to be launched (h->rm_pass)
void take_pending_rm(vh_clu* h) {
    if (h->pend_rm.size() > (size_t)kRmCap) flush_pending_rm(h);
    h->rm_pass.n = (int)h->pend_rm.size();
    for (int i = 0; i < kRmCap; ++i) h->rm_pass.row[i] = i < h->rm_pass.n ? h->pend_rm[(size_t)i] : 0;
    h->pend_rm.clear();
}

void wait_for_scan(vh_clu* h, unsigned long long seq) {
    volatile unsigned long long* flag = h->flag();
    for (unsigned long long spins = 1;; ++spins) {
        if (*flag == seq) break;
        if ((spins & 0xFFFFull) == 0) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {
                if (*flag == seq) break;
                throw ::vh::HipError{hipErrorUnknown, "scan kernel retired without publishing its results", __FILE__, __LINE__};
            }
            if (q != hipErrorNotReady) VH_HIP(q);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
}

// the histograms of scan number `seq` (1-based, = the flag value its publish kernel stores) are in the host-mapped ring
void wait_for_hist(vh_clu* h, unsigned long long seq) {
    volatile unsigned long long* flag = h->hist_flag();
    for (unsigned long long spins = 1; *flag < seq; ++spins) {
        if ((spins & 0xFFFFull) == 0) {
            const hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {
                if (*flag >= seq) break;
                throw ::vh::HipError{hipErrorUnknown, "publish kernel retired without its histograms", __FILE__, __LINE__};
            }
            if (q != hipErrorNotReady) VH_HIP(q);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
}

constexpr size_t kScanLdsBudget = 160 * 1024 - 512;   // one workgroup may use (almost) the whole LDS of a CU

size_t scan_smem_bytes(int km, int L4) {
    return (size_t)(kBlock / 64) * kHitCap * 16 + (size_t)km * kResultWords * 8 + 64 * 4 + (size_t)km * L4 * 4 +
           (size_t)km * 4 * (2 + kLocalCap);
}

RefSrc ref_src(const vh_clu* h) { return RefSrc{h->Mt.p, h->ld, h->L, h->L4, h->q_rows_pass, ref_slack(h->L)}; }

template <int KM, int RPT, int LC, int PIPE, bool REF>
void launch_scan_lc_impl(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    const size_t smem = scan_smem_bytes(KM, h->L4);
    static bool attr_set = false;
    if (!attr_set) {   // wide latent spaces need more than the default 64 KiB of dynamic LDS (query vectors live there)
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(clu_scan_kernel<KM, RPT, LC, PIPE, REF>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScanLdsBudget));
        attr_set = true;
    }
    VH_REQUIRE(smem <= kScanLdsBudget, "internal: %d medoids x %d latent columns do not fit the LDS", KM, h->L4);
    const int64_t blocks = ceil_div(h->ld, (int64_t)kBlock * RPT);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(blocks, 256 * 8));
    hipLaunchKernelGGL((clu_scan_kernel<KM, RPT, LC, PIPE, REF>), dim3(grid), dim3(kBlock), smem, h->stream, h->Mt.p, h->ld, h->L4,
                       h->lengths.p, h->kept.p, h->ld, q_ext, med, h->results.p, h->lists_pass, h->scan_dbg, ref_src(h), h->kept.p, h->rm_pass);
}
template <int KM, int RPT, int LC, int PIPE = 0>
void launch_scan_lc(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    if (h->ref_filter) launch_scan_lc_impl<KM, RPT, LC, PIPE, true>(h, med, q_ext);
    else launch_scan_lc_impl<KM, RPT, LC, PIPE, false>(h, med, q_ext);
}

template <int KM, int RPT>
void launch_scan_rpt(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    // the default latent width (32) has its column loads fully unrolled where the registers allow it
    if constexpr (KM <= 8) {
        if constexpr ((KM + 32) * RPT <= 192) {
            if (h->L4 == 32 && h->scan_lc >= 1 && h->ld < ((int64_t)1 << 30)) {   // 32-bit byte offsets in the kernel
                launch_scan_lc<KM, RPT, 32>(h, med, q_ext);
                return;
            }
        }
    } else {
        if (h->scan_lc >= 1) {
            hipLaunchKernelGGL(clu_gather_quads_kernel, dim3(1), dim3(kBlock), 0, h->stream, h->Mt.p, h->ld, h->L4, q_ext, med,
                               KM, h->qp.p);
            VH_HIP(hipGetLastError());
            launch_scan_lc<KM, RPT, 0, 1>(h, med, h->qp.p);
            return;
        }
    }
    launch_scan_lc<KM, RPT, 0>(h, med, q_ext);
}

// Rows per lane: 4 (few medoids) or 2 keep the loads wide for matrices that stream from HBM.  A matrix that
// fits the Infinity Cache is latency-bound instead: it gets the widest variant that still yields ~3 workgroups
// per CU, down to one row per lane (C1 sweep: scan kernels 262 -> 206 ms with this rule).
template <int KM>
void launch_scan(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    constexpr int RPT = (KM >= 12) ? 2 : 4;
    int rpt = RPT;
    if (h->small_rpt)
        while (rpt > 1 && ceil_div(h->n_rows, (int64_t)kBlock * rpt) < h->min_blocks) rpt >>= 1;
    if (rpt == 1) launch_scan_rpt<KM, 1>(h, med, q_ext);
    else if (rpt == 2) launch_scan_rpt<KM, 2>(h, med, q_ext);
    else launch_scan_rpt<KM, RPT>(h, med, q_ext);
}

template <int NK, bool REF>
void launch_scan_mfma_impl(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    const size_t smem = scan_smem_bytes(kMaxMedoids, 0);
    static bool attr_set = false;
    if (!attr_set) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(clu_scan_mfma_kernel<NK, REF>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScanLdsBudget));
        attr_set = true;
    }
    const int64_t tiles = h->ld >> 5;
    // three workgroups per CU are resident (LDS; two for latent widths above 32, whose operand registers allow two wavefronts
    // per SIMD): one resident set, every wavefront strides over its tiles
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(tiles, kBlock / 64), 256 * (NK <= 16 ? 3 : 2)));
    hipLaunchKernelGGL((clu_scan_mfma_kernel<NK, REF>), dim3(grid), dim3(kBlock), smem, h->stream, h->Mt.p, h->ld, h->L4,
                       h->lengths.p, h->kept.p, q_ext, med, h->mfma_k, h->results.p, h->lists_pass, h->scan_dbg, ref_src(h), h->kept.p, h->rm_pass);
}
template <int NK>
void launch_scan_mfma(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    if (h->ref_filter) launch_scan_mfma_impl<NK, true>(h, med, q_ext);
    else launch_scan_mfma_impl<NK, false>(h, med, q_ext);
}

// K6r: the same pass over the row-major copy (NK = LR / 2 floats per lane and tile)
template <int NK>
void launch_scan_mfma_rm(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    const size_t smem = scan_smem_bytes(kMaxMedoids, 0);
    static bool attr_set = false;
    if (!attr_set) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(clu_scan_mfma_rm_kernel<NK>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kScanLdsBudget));
        attr_set = true;
    }
    const int64_t tiles = h->ld >> 5;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(tiles, kBlock / 64), 256 * (NK <= 16 ? 3 : 2)));
    hipLaunchKernelGGL((clu_scan_mfma_rm_kernel<NK>), dim3(grid), dim3(kBlock), smem, h->stream, h->Mr.p, h->ld, h->lengths.p,
                       h->kept.p, q_ext, h->L4, med, h->mfma_k, h->results.p, h->lists_pass, h->scan_dbg, ref_src(h), h->kept.p, h->rm_pass);
}

// more than 8 medoids and a latent width the B operand registers hold: the matrix-pipe kernel (always 32 medoid slots)
bool scan_uses_mfma(const vh_clu* h, int k) {
    return h->use_mfma && k > 8 && h->L4 <= 64 && h->max_k >= kMaxMedoids && h->ld < ((int64_t)1 << 28);   // 32-bit byte offsets
}

void launch_scan_ref(vh_clu* h, const MedoidRows& med, const float* q_ext) {
    const size_t smem = scan_smem_bytes(kMaxMedoids, h->L4);
    static bool attr_set = false;
    if (!attr_set) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(clu_scan_ref_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kScanLdsBudget));
        attr_set = true;
    }
    VH_REQUIRE(smem <= kScanLdsBudget, "internal: %d latent columns do not fit the LDS", h->L4);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(h->ld, (int64_t)kBlock), 256 * 8));
    hipLaunchKernelGGL(clu_scan_ref_kernel, dim3(grid), dim3(kBlock), smem, h->stream, h->Mt.p, h->ld, h->L, h->L4, h->lengths.p,
                       h->kept.p, h->ld, q_ext, med, h->mfma_k, h->results.p, h->lists_pass, h->scan_dbg, h->kept.p, h->rm_pass);
}

void dispatch_scan(vh_clu* h, int km, const MedoidRows& med, const float* q_ext) {
    if (h->ref_order && !h->ref_filter) {
        launch_scan_ref(h, med, q_ext);
        VH_HIP(hipGetLastError());
        return;
    }
    if (h->mfma_pass) {
        if (h->LR == 32 && h->ref_filter) launch_scan_mfma_rm<16>(h, med, q_ext);
        else if (h->LR == 64 && h->ref_filter) launch_scan_mfma_rm<32>(h, med, q_ext);
        else if (h->L4 <= 32) launch_scan_mfma<16>(h, med, q_ext);
        else launch_scan_mfma<32>(h, med, q_ext);
        VH_HIP(hipGetLastError());
        return;
    }
    switch (km) {
        case 1: launch_scan<1>(h, med, q_ext); break;
        case 2: launch_scan<2>(h, med, q_ext); break;
        case 4: launch_scan<4>(h, med, q_ext); break;
        case 8: launch_scan<8>(h, med, q_ext); break;
        case 12: launch_scan<12>(h, med, q_ext); break;
        case 16: launch_scan<16>(h, med, q_ext); break;
        case 24: launch_scan<24>(h, med, q_ext); break;
        default: launch_scan<32>(h, med, q_ext); break;
    }
    VH_HIP(hipGetLastError());
}

}  // namespace

extern "C" {

const char* vh_last_error(void) { return g_last_error.c_str(); }

int vh_set_option(const char* name, int64_t value) {
    return guarded([&] {
        VH_REQUIRE(name != nullptr && name[0] != 0, "NULL argument");
        std::lock_guard<std::mutex> lock(g_option_mutex);
        option_table()[name] = value;
    });
}

int vh_unset_option(const char* name) {
    return guarded([&] {
        VH_REQUIRE(name != nullptr, "NULL argument");
        std::lock_guard<std::mutex> lock(g_option_mutex);
        option_table().erase(name);
        option_strings().erase(name);
    });
}

int vh_get_option(const char* name, int64_t* value) {
    return guarded([&] {
        VH_REQUIRE(name != nullptr && value != nullptr, "NULL argument");
        *value = option(name, *value);
    });
}

int vh_set_option_string(const char* name, const char* value) {
    return guarded([&] {
        VH_REQUIRE(name != nullptr && value != nullptr, "NULL argument");
        std::lock_guard<std::mutex> lock(g_option_mutex);
        option_strings()[name] = value;
    });
}
const char* vh_version(void) { return "vambhip 0.1 (gfx950)"; }

int vh_device_count(int* n) {
    return guarded([&] {
        VH_REQUIRE(n != nullptr, "n is NULL");
        *n = 0;
        int c = 0;
        hipError_t e = hipGetDeviceCount(&c);
        if (e != hipSuccess) throw HipError{e, "hipGetDeviceCount", __FILE__, __LINE__};
        *n = c;
    });
}

int vh_set_device(int device) {
    return guarded([&] { VH_HIP(hipSetDevice(device)); });
}

int vh_clu_create(const float* matrix, const float* lengths, int64_t n, int L, int normalized, float* normalized_out,
                  vh_clu** out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr, "out is NULL");
        *out = nullptr;
        VH_REQUIRE(matrix != nullptr && lengths != nullptr, "matrix/lengths is NULL");
        VH_REQUIRE(n >= 1, "Matrix must have at least 1 observation.");
        VH_REQUIRE(L >= 1 && L <= 4096, "latent width %d outside [1, 4096]", L);
        VH_REQUIRE(n < (int64_t)2147483647 - 2048, "more than 2^31 rows per shard are not supported");
        std::unique_ptr<vh_clu> h(new vh_clu());
        h->L = L;
        h->L4 = (int)round_up(L, 4);
        h->n_rows = h->n_live = n;
        h->ld = round_up(n, kRowsPerBlock);
        VH_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->Mt.alloc((size_t)h->L4 * h->ld);
        h->lengths.alloc((size_t)h->ld);
        h->kept.alloc((size_t)h->ld);
        h->q.alloc((size_t)kMaxMedoids * h->L4);
        h->qp.alloc((size_t)kMaxMedoids * h->L4);
        {   // the reference has no limit on the latent width; here the scan stages up to 32 query vectors in LDS, so wide
            // latent spaces take fewer medoids per pass (L = 4096: 4) -- never a launch failure in the middle of a sweep
            static const int buckets[] = {32, 24, 16, 12, 8, 4, 2, 1};
            h->max_k = 0;
            for (int b : buckets)
                if (scan_smem_bytes(b, h->L4) <= kScanLdsBudget) { h->max_k = b; break; }
            VH_REQUIRE(h->max_k >= 1, "latent width %d does not fit the scan kernel's LDS", L);
        }
        h->small_rpt = true;
        h->min_blocks = kMinScanBlocks;   // (measured neutral between 384 and 1536)
        h->scan_lc = (int)option("scan.column_loop", 1);
        h->use_mfma = option("scan.mfma", 1) != 0;
        h->publish_split = option("scan.publish_split", 1) != 0;
        {
            const int64_t mode = option("scan.reference_order", 2);
            VH_REQUIRE(mode >= 0 && mode <= 2, "scan.reference_order: 0, 1 or 2");
            h->ref_order = mode != 0;          // normalisation + select in the reference's order
            h->ref_filter = mode == 2;         // scans: the tuned kernels (filter) instead of the plain kernel
        }
        VH_REQUIRE(!h->ref_order || h->ref_filter || h->max_k >= kMaxMedoids, "scan.reference_order = 1: latent width %d is too wide", L);
        // scan.debug switches parts of the scan kernels OFF for timing experiments (wrong results): only honoured by a build made for
        // them (-DVAMBHIP_TIMING_EXPERIMENTS); a stray VAMBHIP_SCAN_DBG cannot corrupt a product build's clustering
#ifdef VAMBHIP_TIMING_EXPERIMENTS
        h->scan_dbg = (int)option("scan.debug", 0);
#else
        h->scan_dbg = 0;
        VH_REQUIRE(option("scan.debug", 0) == 0, "scan.debug needs a library built with -DVAMBHIP_TIMING_EXPERIMENTS (it produces wrong results)");
#endif
        h->results.alloc((size_t)kResultReplicas * kMaxMedoids * kResultWords);
#ifdef VAMBHIP_TIMING_EXPERIMENTS
        h->stamps.alloc((size_t)kStampRows * 8);
        VH_HIP(hipMemset(h->stamps.p, 0, h->stamps.bytes()));
        {
            unsigned long long* sp = h->stamps.p;
            VH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scan_stamps), &sp, sizeof(sp)));
        }
#endif
        VH_HIP(hipHostMalloc((void**)&h->lists, (size_t)kListRing * kMaxMedoids * kListCap * sizeof(int32_t),
                             hipHostMallocMapped | hipHostMallocCoherent));
        const size_t host_words = (size_t)kListRing * kMaxMedoids * (4 + VH_NBINS) + 2;
        VH_HIP(hipHostMalloc((void**)&h->host_results, host_words * 8, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->host_results, 0, host_words * 8);
        VH_HIP(hipHostMalloc((void**)&h->sel_host, (size_t)kSelHostCap * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
        VH_HIP(hipHostMalloc((void**)&h->sel_meta, 2 * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
        h->sel_meta[0] = h->sel_meta[1] = 0ull;
        VH_HIP(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
        VH_HIP(hipMemsetAsync(h->results.p, 0, h->results.bytes(), h->stream));
        h->counts.alloc((size_t)(1 + h->ld / kRowsPerBlock));
        VH_HIP(hipMemsetAsync(h->counts.p, 0, sizeof(unsigned int), h->stream));
        h->total.alloc(1);
        h->h_results.ensure((size_t)kMaxMedoids * kResultWords);
        h->h_q.ensure((size_t)kMaxMedoids * h->L4);

        DevBuf<float> staging;
        staging.alloc((size_t)n * L);
        VH_HIP(hipMemsetAsync(h->Mt.p, 0, h->Mt.bytes(), h->stream));
        VH_HIP(hipMemsetAsync(h->lengths.p, 0, h->lengths.bytes(), h->stream));
        VH_HIP(hipMemcpyAsync(staging.p, matrix, (size_t)n * L * sizeof(float), hipMemcpyHostToDevice, h->stream));
        VH_HIP(hipMemcpyAsync(h->lengths.p, lengths, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(clu_normalize_transpose_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, h->stream,
                           staging.p, n, L, normalized ? 0 : 1, h->Mt.p, h->ld, h->ref_order ? 1 : 0);
        VH_HIP(hipGetLastError());
        hipLaunchKernelGGL(clu_fill_kept_kernel, dim3(1024), dim3(256), 0, h->stream, h->kept.p, n, h->ld);
        VH_HIP(hipGetLastError());
        // the row-major copy of K6r: only in the default mode (the tuned kernels as a filter) and for latent widths its operand
        // registers hold; scan.mfma_rowmajor = 0 keeps K6m (A/B)
        if (h->use_mfma && h->ref_filter && L <= 64 && h->max_k >= kMaxMedoids && option("scan.mfma_rowmajor", 1) != 0) {
            h->LR = L <= 32 ? 32 : 64;
            h->Mr.alloc((size_t)h->ld * h->LR);
            hipLaunchKernelGGL(clu_rows_pad_kernel, dim3(2048), dim3(kBlock), 0, h->stream, staging.p, n, L, h->Mr.p, h->ld, h->LR);
            VH_HIP(hipGetLastError());
        }
        if (normalized_out)
            VH_HIP(hipMemcpyAsync(normalized_out, staging.p, (size_t)n * L * sizeof(float), hipMemcpyDeviceToHost,
                                  h->stream));
        h->host_rows.resize((size_t)n * L);
        VH_HIP(hipMemcpyAsync(h->host_rows.data(), staging.p, (size_t)n * L * sizeof(float), hipMemcpyDeviceToHost,
                              h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
        *out = h.release();
    });
}

int vh_clu_destroy(vh_clu* h) {
    return guarded([&] { delete h; });
}

int vh_clu_max_medoids(vh_clu* h, int* k) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && k != nullptr, "NULL argument");
        *k = h->max_k;
    });
}

int vh_clu_rows(vh_clu* h, int64_t* n_rows, int64_t* n_live) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "handle is NULL");
        if (n_rows) *n_rows = h->n_rows;
        if (n_live) *n_live = h->n_live;
    });
}

int vh_clu_set_timing(vh_clu* h, int enable) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "handle is NULL");
        h->timer.enable(enable != 0);
    });
}

int vh_clu_last_kernel_ms(vh_clu* h, float* ms) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && ms != nullptr, "NULL argument");
        *ms = h->timer.last_ms;
    });
}

}  // extern "C"

namespace {

// Launch one pass for k medoids and wait for its publication; returns the ring slot of the results.
// The accumulators were zeroed by the publish kernel of the previous pass, the medoid rows travel in the
// kernel arguments and the query vectors are gathered by the scan kernel itself.
// sharded: 0 = this handle alone; 1 = row-sharded, the owners' query vectors and the accumulators all-reduced (vh_clu_scan_sharded,
// driven by the Python state machine); 2 = row-sharded with explicit queries and ONE all-gather of accumulators + list parts
// (the native state machine; row_offset = first global physical row of this shard)
int scan_core(vh_clu* h, int k, const int64_t* medoid_rows, const float* queries, int sharded = 0,
              const std::function<void()>* while_waiting = nullptr, int64_t row_offset = 0) {
    VH_REQUIRE(h != nullptr && medoid_rows != nullptr, "NULL argument");
    VH_REQUIRE(sharded == 0 || h->comm != nullptr, "sharded scan needs vh_clu_attach_comm");
    VH_REQUIRE(sharded != 1 || queries == nullptr, "vh_clu_scan_sharded takes no explicit queries");
    VH_REQUIRE(sharded != 2 || queries != nullptr, "internal: the one-collective sharded pass needs explicit queries");
    VH_REQUIRE(k >= 1 && k <= h->max_k, "k=%d outside [1, %d] (vh_clu_max_medoids)", k, h->max_k);
    h->mfma_pass = (h->ref_order && !h->ref_filter) || scan_uses_mfma(h, k);   // (both kernels take 32 medoid slots, the unused ones empty)
    const int km = h->mfma_pass ? kMaxMedoids : pick_km(k);
    MedoidRows med;
    h->mfma_k = k;
    for (int j = 0; j < kMaxMedoids; ++j) {
        const int64_t m = medoid_rows[j < k ? j : 0];
        VH_REQUIRE(m >= -1 && m < h->n_rows, "medoid row %lld out of range", (long long)m);
        VH_REQUIRE(queries != nullptr || sharded || m >= 0, "medoid row -1 needs an explicit query vector");
        med.row[j] = (h->mfma_pass && j >= k) ? -1 : m;   // the matrix-pipe kernel leaves unused slots empty
    }
    const float* q_ext = nullptr;
    if (queries) {
        for (int j = 0; j < km; ++j) {
            const float* src = queries + (size_t)(j < k ? j : 0) * h->L;
            float* dst = h->h_q.p + (size_t)j * h->L4;
            for (int c = 0; c < h->L4; ++c) dst[c] = c < h->L ? src[c] : 0.0f;
        }
        VH_HIP(hipMemcpyAsync(h->q.p, h->h_q.p, (size_t)km * h->L4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
        q_ext = h->q.p;
    }
    if (sharded == 1) {
        // owners contribute their medoids' vectors, RCCL sums them on this stream: no host round trip
        hipLaunchKernelGGL(clu_gather_owned_kernel, dim3(1), dim3(kBlock), 0, h->stream, h->Mt.p, h->ld, h->L4, med, km, h->q.p);
        VH_HIP(hipGetLastError());
        rccl_allreduce_sum_f32(h->comm, h->q.p, (size_t)km * h->L4, h->stream);
        q_ext = h->q.p;
    }
    const int slot = (int)(h->scan_seq % kListRing);
    int32_t* lists = h->lists + (size_t)slot * kMaxMedoids * kListCap;
    h->lists_pass = lists;
    if (sharded == 2) {   // the list parts travel in the exchange block: the kernels append them on the device
        h->lists_dev.ensure((size_t)kMaxMedoids * kListCap);
        h->lists_pass = h->lists_dev.p;
    }
    h->q_rows_pass = q_ext;   // row-major [km][L4] or nullptr (the quad-major copy some kernels take is made from it)
    take_pending_rm(h);       // rows the state machine removed since the last pass: cleared by this pass's own prologue
#ifdef VAMBHIP_TIMING_EXPERIMENTS
    const auto ht0 = std::chrono::steady_clock::now();
    auto host_stamp = [&](int i) { h->host_us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count(); };
    host_stamp(0);
#endif
    h->timer.start(h->stream);
    dispatch_scan(h, km, med, q_ext);
    h->timer.stop(h->stream);
#ifdef VAMBHIP_TIMING_EXPERIMENTS
    host_stamp(1);
#endif
    // the exact integer accumulators of all shards: order-free sums, so the result does not depend on the sharding
    if (sharded == 1) {
        hipLaunchKernelGGL(clu_fold_replicas_kernel, dim3(1), dim3(kBlock), 0, h->stream, km, h->results.p);
        VH_HIP(hipGetLastError());
        rccl_allreduce_sum_u64(h->comm, h->results.p, (size_t)km * kResultWords, h->stream);
    }
    const int world = h->comm ? h->comm->world : 1;
    if (sharded == 2) {
        const size_t block_words = (size_t)k * kXMedWords;
        h->xsend.ensure(block_words);
        h->xrecv.ensure(block_words * (size_t)world);
        if (!h->xlists_host)
            VH_HIP(hipHostMalloc((void**)&h->xlists_host, (size_t)world * kMaxMedoids * (1 + kXListCap) * sizeof(uint32_t),
                                 hipHostMallocMapped | hipHostMallocCoherent));
        hipLaunchKernelGGL(clu_xblock_kernel, dim3(1), dim3(kBlock), 0, h->stream, k, h->mfma_pass ? k : km, h->results.p,
                           h->lists_dev.p, (uint32_t)row_offset, h->xsend.p);
        VH_HIP(hipGetLastError());
        rccl_allgather_bytes(h->comm, h->xsend.p, h->xrecv.p, block_words * 4, h->stream);
        hipLaunchKernelGGL(clu_publish_sharded_kernel, dim3(1), dim3(kPublishThreads), 0, h->stream, k, world, h->xrecv.p,
                           block_words, h->summary(slot), h->hist(slot), h->xlists_host, h->flag(),
                           (unsigned long long)(h->scan_seq + 1));
        VH_HIP(hipGetLastError());
    } else {
        const int pk = h->mfma_pass ? k : km;
        if (h->publish_split)
            hipLaunchKernelGGL(clu_publish2_kernel, dim3(1), dim3(kPublishThreads), 0, h->stream, pk, h->results.p, h->summary(slot),
                               h->hist(slot), h->flag(), h->hist_flag(), (unsigned long long)(h->scan_seq + 1), h->scan_dbg);
        else
            hipLaunchKernelGGL(clu_publish_kernel, dim3(1), dim3(kPublishThreads), 0, h->stream, pk, h->results.p, h->summary(slot),
                               h->hist(slot), h->flag(), h->hist_flag(), (unsigned long long)(h->scan_seq + 1), h->scan_dbg);
        VH_HIP(hipGetLastError());
    }
#ifdef VAMBHIP_TIMING_EXPERIMENTS
    host_stamp(2);
#endif
    if (while_waiting) (*while_waiting)();   // host work that does not depend on this pass, under the pass
    wait_for_scan(h, h->scan_seq + 1);
#ifdef VAMBHIP_TIMING_EXPERIMENTS
    host_stamp(3);
#endif
    if (h->timer.enabled) {
        VH_HIP(hipStreamSynchronize(h->stream));
        h->timer.collect();
    }
    // one bulk copy of the compact summaries into ordinary memory
    unsigned long long sm[kMaxMedoids * 4];
    memcpy(sm, h->summary(slot), (size_t)k * 4 * 8);
    h->last_counts[slot].assign(kMaxMedoids, 0u);
    if (sharded == 2) {
        // merge the ranks' list parts (global rows; rank order = global row order, every part sorted here) into the ring
        // slot the rest of the code reads lists from; a medoid with an incomplete part has no list (the caller selects)
        for (int j = 0; j < k; ++j) {
            int32_t* dst = lists + (size_t)j * kListCap;
            size_t total = 0;
            bool complete = true;
            for (int r = 0; r < world && complete; ++r) {
                const uint32_t* part = h->xlists_host + ((size_t)r * k + j) * (1 + kXListCap);
                const uint32_t cnt = part[0];
                if (cnt == 0xFFFFFFFFu || total + cnt > (size_t)kListCap) { complete = false; break; }
                for (uint32_t i = 0; i < cnt; ++i) dst[total + i] = (int32_t)part[1 + i];
                std::sort(dst + total, dst + total + cnt);
                total += cnt;
            }
            sm[4 * j + 3] = complete ? (unsigned long long)total : ~0ull;
            VH_REQUIRE(!complete || total == sm[4 * j + 1], "internal: sharded list parts do not add up to n_within");
            h->last_counts[slot][j] = complete ? (unsigned int)total : (unsigned int)kListCap + 1u;
        }
        h->last_summary[slot].assign(sm, sm + (size_t)k * 4);
    } else {
        h->last_summary[slot].assign(sm, sm + (size_t)k * 4);
        for (int j = 0; j < k; ++j) {
            const unsigned long long n_within = sm[4 * j + 1], cursor = sm[4 * j + 3];
            // the list is complete iff every within-radius row was appended: cursor == n_within <= capacity
            h->last_counts[slot][j] = (cursor == n_within && n_within <= (unsigned long long)kListCap)
                                          ? (unsigned int)n_within : (unsigned int)kListCap + 1u;
        }
    }
    h->last_k = k;
    h->scan_seq++;
    return slot;
}

}  // namespace

extern "C" {

int vh_clu_scan(vh_clu* h, int k, const int64_t* medoid_rows, const float* queries, vh_scan_result* out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr, "NULL argument");
        const int slot = scan_core(h, k, medoid_rows, queries);
        const std::vector<unsigned long long>& sm = h->last_summary[slot];
        std::vector<unsigned long long> hist((size_t)k * VH_NBINS);
        wait_for_hist(h, h->scan_seq);   // (scan_core has counted the pass: its flag value is scan_seq)
        memcpy(hist.data(), h->hist(slot), hist.size() * 8);
        for (int j = 0; j < k; ++j) {
            out[j].density_fx = (int64_t)sm[4 * j + 0];
            for (int b = 0; b < VH_NBINS; ++b) out[j].hist_fx[b] = (int64_t)hist[(size_t)j * VH_NBINS + b];
            out[j].n_within = (int64_t)sm[4 * j + 1];
            out[j].n_lt = (int64_t)sm[4 * j + 2];
        }
    });
}

int vh_clu_attach_comm(vh_clu* h, vh_comm* comm) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        h->comm = comm;
        if (!comm) return;
        // the select exchange gathers fixed-size chunks: learn the largest shard
        h->xch_counts.ensure((size_t)comm->world + 1);
        const uint32_t mine = (uint32_t)h->ld;
        VH_HIP(hipMemcpyAsync(h->xch_counts.p + comm->world, &mine, 4, hipMemcpyHostToDevice, h->stream));
        rccl_allgather_u32(comm, h->xch_counts.p + comm->world, h->xch_counts.p, 1, h->stream);
        std::vector<uint32_t> all((size_t)comm->world);
        VH_HIP(hipMemcpyAsync(all.data(), h->xch_counts.p, 4 * all.size(), hipMemcpyDeviceToHost, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
        h->max_shard_ld = *std::max_element(all.begin(), all.end());
        h->sel_rows.ensure((size_t)h->max_shard_ld);
    });
}

int vh_clu_scan_sharded(vh_clu* h, int k, const int64_t* local_rows, vh_scan_result* out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr, "NULL argument");
        const int slot = scan_core(h, k, local_rows, nullptr, 1);
        const std::vector<unsigned long long>& sm = h->last_summary[slot];
        std::vector<unsigned long long> hist((size_t)k * VH_NBINS);
        wait_for_hist(h, h->scan_seq);   // (scan_core has counted the pass: its flag value is scan_seq)
        memcpy(hist.data(), h->hist(slot), hist.size() * 8);
        for (int j = 0; j < k; ++j) {
            out[j].density_fx = (int64_t)sm[4 * j + 0];
            for (int b = 0; b < VH_NBINS; ++b) out[j].hist_fx[b] = (int64_t)hist[(size_t)j * VH_NBINS + b];
            out[j].n_within = (int64_t)sm[4 * j + 1];
            out[j].n_lt = (int64_t)sm[4 * j + 2];
        }
    });
}

}  // extern "C"

namespace {

// count + the first kSelXCap rows of a local select into one block (and the select counter re-armed): the common case -- a
// cluster of fewer than kSelXCap members per shard -- then needs ONE all-gather
constexpr int kSelXCap = 1023;
__global__ __launch_bounds__(kBlock) void clu_sel_block_kernel(unsigned int* __restrict__ count, const int32_t* __restrict__ rows,
                                                               uint32_t* __restrict__ block) {
    const unsigned int cnt = *count;
    const unsigned int ncopy = cnt < (unsigned int)kSelXCap ? cnt : (unsigned int)kSelXCap;
    for (unsigned int i = threadIdx.x; i < ncopy; i += kBlock) block[1 + i] = (uint32_t)rows[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        block[0] = cnt;
        *count = 0u;
    }
}

// sharded cluster.py:_smaller_indices.  query == nullptr: the owner of the medoid contributes its vector (all-reduce);
// otherwise `query` is a host [L] vector every rank passes.  Returns the GLOBAL rows, ascending, in h->h_sel64.
int64_t select_sharded_core(vh_clu* h, int64_t local_row, const float* query, float threshold, int remove,
                            const int64_t* row_offsets, std::vector<int64_t>& out) {
    VH_REQUIRE(h->comm != nullptr, "vh_clu_attach_comm has not been called");
    VH_REQUIRE(local_row >= -1 && local_row < h->n_rows, "medoid row out of range");
    vh_comm* comm = h->comm;
    const int world = comm->world;
    h->sel_rows.ensure((size_t)std::max<int64_t>(h->ld, h->max_shard_ld));
    if (query) {
        for (int c = 0; c < h->L4; ++c) h->h_q.p[c] = c < h->L ? query[c] : 0.0f;
        VH_HIP(hipMemcpyAsync(h->q.p, h->h_q.p, (size_t)h->L4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    } else {   // query vector from its owner
        MedoidRows med;
        for (int j = 0; j < kMaxMedoids; ++j) med.row[j] = j == 0 ? local_row : -1;
        hipLaunchKernelGGL(clu_gather_owned_kernel, dim3(1), dim3(kBlock), 0, h->stream, h->Mt.p, h->ld, h->L4, med, 1, h->q.p);
        VH_HIP(hipGetLastError());
        rccl_allreduce_sum_f32(comm, h->q.p, (size_t)h->L4, h->stream);
    }
    // local select (counts[0] is zero on entry)
    flush_pending_rm(h);
    h->timer.start(h->stream);
    hipLaunchKernelGGL(clu_select_kernel, dim3(scan_grid(h->n_rows)), dim3(kBlock), (size_t)h->L4 * 4, h->stream,
                       h->Mt.p, h->ld, h->L4, h->kept.p, h->ld, h->q.p, local_row, threshold, remove,
                       h->sel_rows.p, h->counts.p, h->ref_order ? h->L : 0, ref_slack(h->L), h->ref_filter ? 0 : 1, RmRows{});
    VH_HIP(hipGetLastError());
    h->timer.stop(h->stream);
    const size_t bw = 1 + (size_t)kSelXCap;
    h->xsend.ensure(bw);
    h->xrecv.ensure(bw * (size_t)world);
    hipLaunchKernelGGL(clu_sel_block_kernel, dim3(1), dim3(kBlock), 0, h->stream, h->counts.p, h->sel_rows.p, h->xsend.p);
    VH_HIP(hipGetLastError());
    rccl_allgather_bytes(comm, h->xsend.p, h->xrecv.p, bw * 4, h->stream);
    std::vector<uint32_t> blocks(bw * (size_t)world);
    VH_HIP(hipMemcpyAsync(blocks.data(), h->xrecv.p, 4 * blocks.size(), hipMemcpyDeviceToHost, h->stream));
    VH_HIP(hipStreamSynchronize(h->stream));
    if (h->timer.enabled) h->timer.collect();
    uint32_t maxc = 0;
    int64_t total = 0;
    for (int r = 0; r < world; ++r) {
        maxc = std::max(maxc, blocks[(size_t)r * bw]);
        total += blocks[(size_t)r * bw];
    }
    std::vector<uint32_t> gathered;
    if (maxc > (uint32_t)kSelXCap) {   // a long member list somewhere: the full lists in a second exchange
        VH_REQUIRE((int64_t)maxc <= std::max<int64_t>(h->ld, h->max_shard_ld), "internal: select count exceeds the shard size");
        h->xch_rows.ensure((size_t)world * maxc);
        rccl_allgather_u32(comm, reinterpret_cast<const uint32_t*>(h->sel_rows.p), h->xch_rows.p, maxc, h->stream);
        gathered.resize((size_t)world * maxc);
        VH_HIP(hipMemcpyAsync(gathered.data(), h->xch_rows.p, 4 * gathered.size(), hipMemcpyDeviceToHost, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
    }
    // rank order is global row order: sort every rank's local rows, shift them by its offset
    out.clear();
    out.reserve((size_t)total);
    for (int r = 0; r < world; ++r) {
        const uint32_t cnt = blocks[(size_t)r * bw];
        uint32_t* part = gathered.empty() ? blocks.data() + (size_t)r * bw + 1 : gathered.data() + (size_t)r * maxc;
        std::sort(part, part + cnt);
        for (uint32_t i = 0; i < cnt; ++i) out.push_back(row_offsets[r] + (int64_t)part[i]);
    }
    if (remove) h->n_live -= blocks[(size_t)comm->rank * bw];
    return total;
}

}  // namespace

extern "C" {

int vh_clu_select_sharded(vh_clu* h, int64_t local_row, float threshold, int remove, const int64_t* row_offsets,
                          int64_t* out_rows, int64_t cap, int64_t* n_out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && n_out != nullptr && row_offsets != nullptr, "NULL argument");
        VH_REQUIRE(cap >= 0 && (cap == 0 || out_rows != nullptr), "bad output buffer");
        std::vector<int64_t> rows;
        const int64_t total = select_sharded_core(h, local_row, nullptr, threshold, remove, row_offsets, rows);
        for (int64_t i = 0; i < std::min<int64_t>(cap, total); ++i) out_rows[i] = rows[(size_t)i];
        *n_out = total;
    });
}

int vh_clu_scan_seq(vh_clu* h, int64_t* seq) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && seq != nullptr, "NULL argument");
        *seq = (int64_t)h->scan_seq;
    });
}

int vh_clu_scan_list(vh_clu* h, int64_t seq, int j, int64_t* out_rows, int64_t cap, int64_t* n_out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && n_out != nullptr, "NULL argument");
        VH_REQUIRE(j >= 0 && j < kMaxMedoids, "medoid index out of range");
        *n_out = -1;
        // seq is the value vh_clu_scan_seq returned BEFORE the scan, i.e. the scan's own sequence number
        if (seq < 0 || (uint64_t)seq >= h->scan_seq || h->scan_seq - (uint64_t)seq > (uint64_t)kListRing) return;
        const int slot = (int)((uint64_t)seq % kListRing);
        if (h->last_counts[slot].empty()) return;
        const unsigned int cnt = h->last_counts[slot][j];
        if (cnt > (unsigned int)kListCap) return;   // the list overflowed: the caller falls back to vh_clu_select
        VH_REQUIRE(cap >= (int64_t)cnt && (cnt == 0 || out_rows != nullptr), "output buffer too small");
        const int32_t* src = h->lists + ((size_t)slot * kMaxMedoids + j) * kListCap;
        h->h_sel.assign(src, src + cnt);
        std::sort(h->h_sel.begin(), h->h_sel.end());
        for (unsigned int i = 0; i < cnt; ++i) out_rows[i] = h->h_sel[i];
        *n_out = cnt;
    });
}

int vh_clu_select(vh_clu* h, int64_t medoid_row, const float* query, float threshold, int remove, int64_t* out_rows,
                  int64_t cap, int64_t* n_out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && n_out != nullptr, "NULL argument");
        VH_REQUIRE(medoid_row >= -1 && medoid_row < h->n_rows, "medoid row out of range");
        VH_REQUIRE(query != nullptr || medoid_row >= 0, "medoid row -1 needs an explicit query vector");
        VH_REQUIRE(cap >= 0 && (cap == 0 || out_rows != nullptr), "bad output buffer");
        h->sel_rows.ensure((size_t)h->ld);
        const float* q_ext = nullptr;
        if (query) {
            for (int c = 0; c < h->L4; ++c) h->h_q.p[c] = c < h->L ? query[c] : 0.0f;
            VH_HIP(hipMemcpyAsync(h->q.p, h->h_q.p, (size_t)h->L4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
            q_ext = h->q.p;
        }
        // counts[0] is zero on entry (creation / re-armed by the publish kernel of the previous select)
        take_pending_rm(h);   // (up to kRmCap removed rows ride in the kernel's arguments, as in the scans; longer lists keep their launch)
        h->timer.start(h->stream);
        hipLaunchKernelGGL(clu_select_kernel, dim3(scan_grid(h->n_rows)), dim3(kBlock), (size_t)h->L4 * 4, h->stream,
                           h->Mt.p, h->ld, h->L4, h->kept.p, h->ld, q_ext, medoid_row, threshold, remove,
                           h->sel_rows.p, h->counts.p, h->ref_order ? h->L : 0, ref_slack(h->L), h->ref_filter ? 0 : 1, h->rm_pass);
        VH_HIP(hipGetLastError());
        h->timer.stop(h->stream);
        // count and (short) row list travel through host-mapped memory; the host spins on the sequence flag
        const unsigned long long seq = ++h->sel_seq;
        hipLaunchKernelGGL(clu_publish_select_kernel, dim3(1), dim3(kBlock), 0, h->stream, h->counts.p, h->sel_rows.p,
                           h->sel_host, h->sel_meta, seq);
        VH_HIP(hipGetLastError());
        {
            volatile unsigned long long* flag = h->sel_meta + 1;
            for (unsigned long long spins = 1;; ++spins) {
                if (*flag == seq) break;
                if ((spins & 0xFFFFull) == 0) {
                    const hipError_t qe = hipStreamQuery(h->stream);
                    if (qe == hipSuccess) {
                        if (*flag == seq) break;
                        throw ::vh::HipError{hipErrorUnknown, "select kernel retired without publishing its results", __FILE__, __LINE__};
                    }
                    if (qe != hipErrorNotReady) VH_HIP(qe);
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (h->timer.enabled) {
            VH_HIP(hipStreamSynchronize(h->stream));
            h->timer.collect();
        }
        const unsigned int cnt = (unsigned int)h->sel_meta[0];
        h->h_sel.resize(cnt);
        if (cnt) {
            if (cnt <= (unsigned int)kSelHostCap) {
                memcpy(h->h_sel.data(), h->sel_host, (size_t)cnt * sizeof(int32_t));
            } else {   // long list: the device copy is intact until the next select
                VH_HIP(hipMemcpyAsync(h->h_sel.data(), h->sel_rows.p, (size_t)cnt * sizeof(int32_t),
                                      hipMemcpyDeviceToHost, h->stream));
                VH_HIP(hipStreamSynchronize(h->stream));
            }
            std::sort(h->h_sel.begin(), h->h_sel.end());
        }
        const int64_t w = std::min<int64_t>(cap, cnt);
        for (int64_t i = 0; i < w; ++i) out_rows[i] = h->h_sel[i];
        *n_out = cnt;
        if (remove) h->n_live -= cnt;
    });
}

int vh_clu_remove(vh_clu* h, const int64_t* rows, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && (n == 0 || rows != nullptr), "NULL argument");
        VH_REQUIRE(n >= 0, "negative count");
        if (n == 0) return;
        for (int64_t i = 0; i < n; ++i)
            VH_REQUIRE(rows[i] >= 0 && rows[i] < h->n_rows, "row %lld out of range", (long long)rows[i]);
        flush_pending_rm(h);
        // count rows that are still live so that n_live stays exact
        std::vector<uint8_t> flags;
        h->row_idx.ensure((size_t)n);
        VH_HIP(hipMemcpyAsync(h->row_idx.p, rows, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        // read back current flags of these rows (small lists) to keep the live count exact
        std::vector<int64_t> uniq(rows, rows + n);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        int64_t live = 0;
        for (int64_t r : uniq) {
            uint8_t f = 0;
            VH_HIP(hipMemcpyAsync(&f, h->kept.p + r, 1, hipMemcpyDeviceToHost, h->stream));
            VH_HIP(hipStreamSynchronize(h->stream));
            live += f != 0;
        }
        hipLaunchKernelGGL(clu_remove_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, h->stream, h->kept.p,
                           h->row_idx.p, n);
        VH_HIP(hipGetLastError());
        VH_HIP(hipStreamSynchronize(h->stream));
        h->n_live -= live;
    });
}

int vh_clu_pack(vh_clu* h, int64_t* new_rows) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "handle is NULL");
        flush_pending_rm(h);
        const int nb = (int)(h->ld / kRowsPerBlock);
        unsigned int* block_counts = h->counts.p + 1;
        hipLaunchKernelGGL(clu_count_kept_kernel, dim3(nb), dim3(kBlock), 0, h->stream, h->kept.p, h->ld, block_counts);
        VH_HIP(hipGetLastError());
        hipLaunchKernelGGL(clu_exclusive_scan_kernel, dim3(1), dim3(1024), 0, h->stream, block_counts, nb, h->total.p);
        VH_HIP(hipGetLastError());
        unsigned long long total = 0;
        VH_HIP(hipMemcpyAsync(&total, h->total.p, sizeof(total), hipMemcpyDeviceToHost, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
        const int64_t n_new = (int64_t)total;
        VH_REQUIRE(n_new == h->n_live, "internal: live count mismatch (%lld vs %lld)", (long long)n_new,
                   (long long)h->n_live);
        if (n_new == h->n_rows) {
            if (new_rows) *new_rows = n_new;
            return;
        }
        h->sel_rows.ensure((size_t)h->ld);
        hipLaunchKernelGGL(clu_build_src_kernel, dim3(nb), dim3(kBlock), 0, h->stream, h->kept.p, h->ld, block_counts,
                           h->sel_rows.p);
        VH_HIP(hipGetLastError());
        const int64_t ld_new = round_up(std::max<int64_t>(n_new, 1), kRowsPerBlock);
        h->Mt_alt.ensure((size_t)h->L4 * ld_new);
        h->lengths_alt.ensure((size_t)ld_new);
        const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(ld_new, kBlock), 2048));
        hipLaunchKernelGGL(clu_gather_columns_kernel, dim3(gx, h->L4 + 1), dim3(kBlock), 0, h->stream, h->Mt.p, h->ld,
                           h->Mt_alt.p, ld_new, h->sel_rows.p, n_new, h->lengths.p, h->lengths_alt.p, h->L4);
        VH_HIP(hipGetLastError());
        if (h->LR) {
            h->Mr_alt.ensure((size_t)ld_new * h->LR);
            hipLaunchKernelGGL(clu_gather_rows_kernel, dim3(gx), dim3(kBlock), 0, h->stream, h->Mr.p, h->Mr_alt.p, h->sel_rows.p, n_new,
                               ld_new, h->LR);
            VH_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(clu_fill_kept_kernel, dim3(1024), dim3(256), 0, h->stream, h->kept.p, n_new, h->ld);
        VH_HIP(hipGetLastError());
        VH_HIP(hipStreamSynchronize(h->stream));
        std::swap(h->Mt, h->Mt_alt);
        if (h->LR) std::swap(h->Mr, h->Mr_alt);
        std::swap(h->lengths, h->lengths_alt);
        h->ld = ld_new;
        h->n_rows = n_new;
        if (new_rows) *new_rows = n_new;
    });
}

int vh_clu_get_rows(vh_clu* h, const int64_t* rows, int64_t k, float* out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && out != nullptr, "NULL argument");
        if (rows == nullptr) k = h->n_rows;
        VH_REQUIRE(k >= 0, "negative count");
        if (k == 0) return;
        if (rows) {
            for (int64_t i = 0; i < k; ++i)
                VH_REQUIRE(rows[i] >= 0 && rows[i] < h->n_rows, "row %lld out of range", (long long)rows[i]);
            h->row_idx.ensure((size_t)k);
            VH_HIP(hipMemcpyAsync(h->row_idx.p, rows, (size_t)k * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        }
        DevBuf<float> tmp;
        tmp.alloc((size_t)k * h->L);
        const int64_t tot = k * h->L;
        hipLaunchKernelGGL(clu_rows_to_rowmajor_kernel, dim3((unsigned)ceil_div(tot, 256)), dim3(256), 0, h->stream,
                           h->Mt.p, h->ld, h->L, rows ? h->row_idx.p : nullptr, k, tmp.p);
        VH_HIP(hipGetLastError());
        VH_HIP(hipMemcpyAsync(out, tmp.p, (size_t)tot * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
    });
}

int vh_clu_get_kept(vh_clu* h, uint8_t* out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && out != nullptr, "NULL argument");
        flush_pending_rm(h);
        VH_HIP(hipMemcpyAsync(out, h->kept.p, (size_t)h->n_rows, hipMemcpyDeviceToHost, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
    });
}

}  // extern "C"

// =============================================================================================
// Host state machine of ClusterGenerator (cluster.py:294-604) in C++: seed walk, wander_medoid,
// find_threshold, success window, packing policy.  It issues the same scans / selects as the Python
// class in vamb_amd/cluster.py (which remains the implementation behind the row-sharded multi-GPU
// backend) -- the Python interpreter was 60 % of a C1 sweep.
// =============================================================================================
namespace {

// CPython's random.Random for an int seed: MT19937 seeded with init_by_array(32-bit chunks of |seed|),
// getrandbits(k <= 32) = genrand_uint32() >> (32 - k), _randbelow_with_getrandbits and random.sample
// (Lib/random.py; cluster.py:269, 430, 445 use rng.sample only).
struct PyRandom {
    uint32_t mt[624];
    int idx = 625;
    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void seed(uint64_t a) {
        uint32_t key[2] = {(uint32_t)(a & 0xFFFFFFFFu), (uint32_t)(a >> 32)};
        const int klen = key[1] ? 2 : 1;
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = (624 > klen ? 624 : klen); k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
            if (++j >= klen) j = 0;
        }
        for (int k = 623; k; --k) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
    }
    uint32_t next32() {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7FFFFFFFu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9D2C5680u;
        y ^= (y << 15) & 0xEFC60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t randbelow(uint32_t n) {   // n >= 1
        int bits = 0;
        for (uint32_t t = n; t; t >>= 1) ++bits;
        uint32_t r = next32() >> (32 - bits);
        while (r >= n) r = next32() >> (32 - bits);
        return r;
    }
    // random.sample(population, k): pool swap for small populations, rejection set for large ones
    void sample(const std::vector<int64_t>& pop, int k, std::vector<int64_t>& out) {
        const size_t n = pop.size();
        out.assign((size_t)k, 0);
        double setsize = 21.0;
        if (k > 5) setsize += std::pow(4.0, std::ceil(std::log((double)k * 3.0) / std::log(4.0)));
        if ((double)n <= setsize) {
            std::vector<int64_t> pool(pop);
            for (int i = 0; i < k; ++i) {
                const uint32_t j = randbelow((uint32_t)(n - (size_t)i));
                out[i] = pool[j];
                pool[j] = pool[n - (size_t)i - 1];
            }
        } else {
            std::vector<uint32_t> chosen;
            for (int i = 0; i < k; ++i) {
                uint32_t j = randbelow((uint32_t)n);
                while (std::find(chosen.begin(), chosen.end(), j) != chosen.end()) j = randbelow((uint32_t)n);
                chosen.push_back(j);
                out[i] = pop[j];
            }
        }
    }
};

struct GenStats {
    double density = 0.0;
    int64_t n_within = 0, n_lt = 0;
    int64_t hist_fx[VH_NBINS];     // fetched on demand (only the medoid a cluster is built around needs it)
    bool have_hist = false;
    bool have_list = false;
    std::vector<int64_t> within;   // ascending rows inside the medoid radius (when have_list)
    uint64_t seq = 0;              // scan that produced the statistics; its results sit in ring slot seq % kListRing
    int slot_j = 0;
    unsigned int list_count = 0;   // > kListCap: the device list is incomplete
    bool spec = false;             // scanned ahead of need (an upcoming seed or a neighbour of one) and not used yet
    int64_t born = 0;              // emission count at its scan (age-based eviction)
    int64_t checked = 0;           // the near-field statistics are known to be exact as of this emission count (lazy validation)
    int64_t hist_checked = 0;      // ... and the histogram (range 0.3)
    std::vector<float> vec;        // the row itself (host copy): the per-emission validity check reads it from here
    bool hist_stale = false;       // rows were removed since the scan: the histogram (range 0.3) must be taken again
};

// float32(0.005) * float32(N(0, 0.01) pdf) -- the _NORMALPDF table of cluster.py:39-73
constexpr int kPdfLen = 31;
const float kPdfRaw[kPdfLen] = {
    2.43432053e-11f, 9.13472041e-10f, 2.66955661e-08f, 6.07588285e-07f, 1.07697600e-05f, 1.48671951e-04f,
    1.59837411e-03f, 1.33830226e-02f, 8.72682695e-02f, 4.43184841e-01f, 1.75283005e00f, 5.39909665e00f,
    1.29517596e01f, 2.41970725e01f, 3.52065327e01f, 3.98942280e01f, 3.52065327e01f, 2.41970725e01f,
    1.29517596e01f, 5.39909665e00f, 1.75283005e00f, 4.43184841e-01f, 8.72682695e-02f, 1.33830226e-02f,
    1.59837411e-03f, 1.48671951e-04f, 1.07697600e-05f, 6.07588285e-07f, 2.66955661e-08f, 9.13472041e-10f,
    2.43432053e-11f};

enum ThresholdKind { kLoner = 0, kNoThreshold = 1, kThreshold = 2 };

}  // namespace

struct vh_gen {
    vh_clu* clu = nullptr;          // borrowed
    // gen_speculative_fill asks for the physical row of the same few upcoming seeds pass after pass: a direct-mapped memo of
    // (original index -> row) saves the binary searches over `indices` (16 MB at C2: ~20 cache misses each); a pack renumbers
    // the rows and starts a new epoch
    struct RowMemo { int64_t orig = -1; int64_t row = 0; uint64_t epoch = 0; };
    RowMemo row_memo[256];
    uint64_t rows_epoch = 1;
    // Row-sharded execution (vh_gen_create_sharded): `clu` holds this rank's shard and every rank runs this same state machine
    // in lock step on GLOBAL physical rows (rank order = global row order; offsets[r] = first global row of rank r).  All
    // inputs of a decision are identical on every rank -- the pass results are exact integer sums over the shards -- so the
    // ranks take the same decisions and emit the same stream; no control-plane traffic is needed between them.
    vh_comm* comm = nullptr;
    std::vector<int64_t> offsets;   // [world + 1] (sharded only)
    std::vector<float> rows_global; // the whole normalised matrix by ORIGINAL row (sharded only; clu->host_rows is the shard's)
    int maxsteps = 25, minsuccesses = 15;
    size_t windowsize = 300;
    double pack_fraction = 0.5;
    int64_t pack_min_rows = 8192;
    PyRandom rng;
    std::vector<int64_t> order;     // contig indices by descending length, -1 = tombstone
    std::vector<int64_t> indices;   // original contig index of every physical row (ascending)
    std::vector<uint8_t> kept;      // host mirror of the device live mask (physical rows)
    std::vector<uint8_t> alive;     // the same by original contig index
    std::vector<int32_t> bit;       // Fenwick tree over kept[] (prefix counts of live rows)
    int64_t order_index = 0;
    int64_t n_emitted = 0, n_remaining = 0;
    double pvr = 0.1;
    std::deque<bool> attempts;
    int successes = 0;
    std::unordered_map<int64_t, GenStats> stats;
    uint64_t ring_rows_valid_from = 0;   // lists of scans older than this (clu->scan_seq at the last pack) name pre-pack rows
    // speculative results of the last pass that have not been turned into `stats` entries yet: that host work (a copy of the
    // row, of its list, a map insert per medoid; 1.2 s of a C2 sweep) runs while the GPU executes the NEXT pass, or at once for
    // an entry somebody asks for
    struct Pending { int64_t row; int slot; int j; uint64_t seq; int64_t born; };
    std::vector<Pending> pending;
    std::vector<uint8_t> pending_mark;   // [current rows]: 1 while the row is in `pending`
    std::vector<uint8_t> tried_mark;     // [current rows]: 1 while the row is in the running walk's `tried` list (gen_wander)
    bool defer_book = true;         // option gen.defer_bookkeeping
    // Removal log: one record per emitted cluster (index = emission count at the time), the rows it removed by ORIGINAL index.
    // Cached statistics are validated against it lazily, when they are looked at (gen_lookup), instead of eagerly at every
    // emission.
    struct Removal {
        int64_t first, count;     // into removed_orig
        float cos_near, cos_far;  // 2 <e, medoid> below this: no removed row can lie within 0.05 / 0.3 of e
    };
    std::vector<Removal> rlog;        // record of emission e is rlog[e - rlog_base]
    std::vector<float> rlog_medoid;   // [emission - rlog_base][L]: the emitted medoids (pivots of the triangle-inequality filter),
                                      // contiguous -- a lazy check walks consecutive records, never the 256 MB host matrix
    std::vector<int64_t> removed_orig;   // rows of record R: removed_orig[R.first - removed_base ...]
    // Records older than max_entry_age emissions are never read again (entries that old are dropped unseen), so the log is
    // trimmed now and then: a 10 M-row sweep that emits millions of clusters would otherwise hold ~1 GB of host memory for nothing.
    int64_t rlog_base = 0, removed_base = 0;
    int64_t lazy_checks = 0, lazy_cluster_tests = 0, lazy_point_tests = 0, lazy_invalid = 0, hist_kept = 0;
    // counters (bench accounting)
    int64_t scan_passes = 0, scan_medoids = 0, rows_streamed = 0;   // rows_streamed: RESIDENT rows per pass (what the kernels read)
    int64_t live_rows_streamed = 0;   // live rows per pass (algorithmic bytes count N_live, not the uncompacted matrix)
    double kernel_ms = 0.0;
    std::vector<int64_t> sel;       // scratch
    // speculative seed scans: upcoming seeds share the pass of whatever has to be scanned anyway
    int64_t spec_scanned = 0, spec_used = 0, spec_dropped = 0;
    bool speculate = true;
    int spec_window = kSpecWindow;
    int64_t max_entry_age = kMaxEntryAgeDefault;   // option gen.max_entry_age
    int spec_depth = 2;             // option gen.spec_depth: 1 = within-radius rows of upcoming seeds, 2 = also THEIR within-radius rows
    double t_validate = 0, t_fill = 0, t_book = 0, t_emit = 0, t_book_hidden = 0;   // profile: lazy validation, speculative fill, post-scan bookkeeping, emission
    // profile, INCLUSIVE times of the top-level pieces of one emission (scans and selects inside them included):
    double t_wander_incl = 0, t_hist_incl = 0, t_threshold = 0, t_members = 0, t_live = 0, t_pack_incl = 0, t_sample = 0, t_within = 0;
    // Speculative fill one pass AHEAD (option gen.prefill): while a pass runs on the GPU the host already collects the rows it
    // would add to the NEXT pass's free slots; that pass takes the rows of the list that are still unscanned and live and walks
    // the pools again only when the list comes up short although it had been cut at the slot count.  What is scanned ahead never
    // changes a result (lazy validation), only the number of passes.
    bool pass_in_flight = false;    // host code running UNDER a pass: the ring slot that pass writes is not to be read (gen_lookup)
    int prefill_on = 2;             // 2: a bounded fresh walk (the first prefill_fresh_seeds upcoming seeds, their own pools) in front of the list
    int prefill_fresh_seeds = 4;
    bool inline_rm = true;          // gen.inline_removals: removed rows ride in the next scan's kernel arguments (RmRows); 0 = one launch per emission
    std::vector<int64_t> prefill;
    uint64_t prefill_epoch = 0;
    bool prefill_ready = false, prefill_exhausted = false;
    double t_fill_hidden = 0;
    long long prefill_hits = 0, prefill_topups = 0;
    bool spec_neighbours = true;   // option gen.spec_neighbours: within-radius rows of cached upcoming seeds are scanned ahead too
    int spec_big_target = 0;      // experiment: widening target of passes over matrices above 600 k rows (0 = bucket fill)
    // optional wall-clock breakdown (VAMBHIP_GEN_PROFILE=1): scan calls, select calls, seed walk, logical index
    bool profile = false;
    double t_scan = 0, t_select = 0, t_seed = 0, t_logical = 0, t_total = 0;
    double t_km[33] = {0};          // scan wall time by medoid count of the pass (profile)
    int64_t pass_seed = 0, pass_cand = 0, pass_hist = 0, pass_select = 0, pass_listsel = 0;   // passes by purpose (profile)
    int64_t cand_rounds = 0, cand_rounds_cached = 0, cand_needed = 0, seeds_total = 0, seeds_cached = 0, wander_moves = 0;
    int64_t kept_entries = 0, kept_emissions = 0, full_checks = 0;
    int pass_purpose = 0;           // what the next scan pass is for: 0 seed, 1 candidate round
    int64_t n_km[33] = {0}, rows_km[33] = {0};
};

namespace {

struct GenTimer {
    double* acc;
    std::chrono::steady_clock::time_point t0;
    explicit GenTimer(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~GenTimer() { *acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

void gen_check(int rc) {
    if (rc != VH_OK) throw ::vh::HipError{hipErrorUnknown, g_last_error.c_str(), __FILE__, __LINE__};
}

void gen_collect_ms(vh_gen* g) {
    if (g->clu->timer.enabled) g->kernel_ms += g->clu->timer.last_ms;
}

// host copy of the normalised matrix, row-major by ORIGINAL (global) row
const float* gen_host_rows(const vh_gen* g) { return g->comm ? g->rows_global.data() : g->clu->host_rows.data(); }

// global physical row -> row of this rank's shard, or -1
int64_t gen_local_row(const vh_gen* g, int64_t row) {
    if (!g->comm) return row;
    const int64_t lo = g->offsets[(size_t)g->comm->rank], hi = g->offsets[(size_t)g->comm->rank + 1];
    return row >= lo && row < hi ? row - lo : -1;
}

// one scan pass for k medoids given by (global) physical row
int gen_scan(vh_gen* g, int k, const int64_t* rows, const std::function<void()>* while_waiting) {
    if (!g->comm) return scan_core(g->clu, k, rows, nullptr, 0, while_waiting);
    const int L = g->clu->L;
    int64_t local[kMaxMedoids];
    std::vector<float> q((size_t)k * L);
    const float* hm = gen_host_rows(g);
    for (int j = 0; j < k; ++j) {
        local[j] = gen_local_row(g, rows[j]);
        const float* v = hm + (size_t)g->indices[(size_t)rows[j]] * L;
        std::copy(v, v + L, q.begin() + (size_t)j * L);
    }
    return scan_core(g->clu, k, local, q.data(), 2, while_waiting, g->offsets[(size_t)g->comm->rank]);
}

float gen_dot(const float* a, const float* b, int L) {
    float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int c = 0;
    for (; c + 8 <= L; c += 8)
        for (int k = 0; k < 8; ++k) part[k] += a[c + k] * b[c + k];
    float dot = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    for (; c < L; ++c) dot += a[c] * b[c];
    return dot;
}

// Has any row removed by the emissions [from, n_emitted) come within `radius` (+ margin) of the row `vm`?  Per emitted cluster
// one dot product against its medoid decides whether the cluster can matter at all (triangle inequality on the sphere: rows are
// scaled to norm 1 / sqrt(2), cos(angle) = 2 <x, y>; the margins cover any float32 summation order); only then row by row.
bool gen_touched_since(vh_gen* g, const float* vm, int64_t from, bool far) {
    const int L = g->clu->L;
    const float* hm = gen_host_rows(g);
    const float limit = (far ? 0.3f : 0.05f) + 2e-3f;
    VH_REQUIRE(from >= g->rlog_base, "internal: the removal log was trimmed past an entry that is still in use");
    const float* piv = g->rlog_medoid.data() + (size_t)(from - g->rlog_base) * L;
    for (int64_t e = from - g->rlog_base; e < (int64_t)g->rlog.size(); ++e, piv += L) {
        const vh_gen::Removal& R = g->rlog[(size_t)e];
        g->lazy_cluster_tests++;
        if (2.0f * gen_dot(vm, piv, L) < (far ? R.cos_far : R.cos_near) - 1e-4f) continue;
        const int64_t* rows = g->removed_orig.data() + (R.first - g->removed_base);
        for (int64_t k = 0; k < R.count; ++k) __builtin_prefetch(hm + (size_t)rows[k] * L);
        for (int64_t k = 0; k < R.count; ++k) {
            g->lazy_point_tests++;
            if (0.5f - gen_dot(vm, hm + (size_t)rows[k] * L, L) <= limit) return true;
        }
    }
    return false;
}

// The cached statistics of a row, or nullptr.  What wander_medoid and the loner test read of an entry -- density, the two
// counts and the list of rows within the medoid radius 0.05 -- is a function of the live rows INSIDE that radius only
// (sample_medoid is pure, cluster.py:606-637), so an entry is exact as long as no row removed since its scan lies within the
// radius of its row.  The reference clears its cache at every emission (cluster.py:298-316); here every entry is checked
// against the removal log when it is looked at, once per emission it has not seen yet.
GenStats* gen_lookup(vh_gen* g, int64_t row);
// The cached entry of one scanned medoid (from the pass's summary, its ring slot and the host copy of the row)
GenStats& gen_materialise(vh_gen* g, int64_t row, int slot, int j, uint64_t seq, int64_t born, bool spec) {
    const std::vector<unsigned long long>& sm = g->clu->last_summary[slot];
    GenStats& st = g->stats[row];
    // the python float the reference gets from `.sum().item()` on a float32 tensor (cluster.py:629)
    st.density = (double)(float)((double)(int64_t)sm[4 * j] / VH_DENSITY_SCALE);
    st.n_within = (int64_t)sm[4 * j + 1];
    st.n_lt = (int64_t)sm[4 * j + 2];
    st.have_hist = false;
    // histogram and candidate list stay in the host-mapped ring; they are fetched only for the few
    // medoids that need them (host reads of that memory are slow: one bulk copy, on demand)
    st.seq = seq;
    st.slot_j = j;
    st.list_count = g->clu->last_counts[slot][j];
    st.have_list = false;
    st.within.clear();
    st.spec = spec;
    st.born = st.checked = st.hist_checked = born;
    st.hist_stale = false;
    {   // the row itself, for the validity checks of the emissions to come
        const float* v = gen_host_rows(g) + (size_t)g->indices[(size_t)row] * g->clu->L;
        st.vec.assign(v, v + g->clu->L);
    }
    if (spec) {
        g->spec_scanned++;
        // an entry scanned ahead may be used after its scan has left the ring, and its list names the rows to scan
        // ahead next (gen_speculative_fill): keep it now if it is short
        if (st.list_count <= (unsigned int)kKeepList) {
            const int32_t* src = g->clu->lists + ((size_t)slot * kMaxMedoids + j) * kListCap;
            st.within.assign(src, src + st.list_count);
            std::sort(st.within.begin(), st.within.end());
            st.have_list = true;
        }
    }
    return st;
}

void gen_flush_pending(vh_gen* g) {
    for (const vh_gen::Pending& p : g->pending) {
        g->pending_mark[(size_t)p.row] = 0;
        gen_materialise(g, p.row, p.slot, p.j, p.seq, p.born, true);
    }
    g->pending.clear();
}

bool gen_is_pending(const vh_gen* g, int64_t row) {
    return (size_t)row < g->pending_mark.size() && g->pending_mark[(size_t)row] != 0;
}

GenStats* gen_lookup(vh_gen* g, int64_t row) {
    if (gen_is_pending(g, row)) {   // scanned by the last pass, not turned into an entry yet
        for (size_t i = 0; i < g->pending.size(); ++i) {
            if (g->pending[i].row != row) continue;
            const vh_gen::Pending p = g->pending[i];
            g->pending[i] = g->pending.back();
            g->pending.pop_back();
            g->pending_mark[(size_t)row] = 0;
            gen_materialise(g, p.row, p.slot, p.j, p.seq, p.born, true);
            break;
        }
    }
    const auto it = g->stats.find(row);
    if (it == g->stats.end()) return nullptr;
    GenStats& st = it->second;
    if (st.checked == g->n_emitted) return &st;
    GenTimer tv(&g->t_validate);
    g->lazy_checks++;
    if (g->n_emitted - st.born > g->max_entry_age || gen_touched_since(g, st.vec.data(), st.checked, false)) {
        if (st.spec) g->spec_dropped++;
        g->lazy_invalid++;
        g->stats.erase(it);
        return nullptr;
    }
    st.checked = g->n_emitted;
    // an entry that outlives its emission will be asked for its list sooner or later: take it while its scan is in the ring.
    // (Under a running pass -- the fill one pass ahead -- the ring is one slot shorter: that pass is writing the slot of the
    // scan kListRing passes back.)
    if (!st.have_list && st.list_count <= (unsigned int)kListCap &&
        g->clu->scan_seq + (g->pass_in_flight ? 1u : 0u) - st.seq <= (uint64_t)kListRing && st.seq >= g->ring_rows_valid_from) {
        const int32_t* src = g->clu->lists + ((size_t)(st.seq % kListRing) * kMaxMedoids + st.slot_j) * kListCap;
        st.within.assign(src, src + st.list_count);
        std::sort(st.within.begin(), st.within.end());
        st.have_list = true;
    }
    return &st;
}

// kept[row] = 0 for rows the state machine knows to be live: stream-ordered launch, no read-back, no wait
void gen_remove_live(vh_gen* g, const int64_t* rows_in, int64_t n_in) {
    vh_clu* h = g->clu;
    std::vector<int64_t> mine;   // sharded: the rows of this rank's shard, as local rows
    const int64_t* rows = rows_in;
    int64_t n = n_in;
    if (g->comm) {
        for (int64_t i = 0; i < n_in; ++i) {
            const int64_t l = gen_local_row(g, rows_in[i]);
            if (l >= 0) mine.push_back(l);
        }
        rows = mine.data();
        n = (int64_t)mine.size();
    }
    // no launch: the rows wait on the handle and are cleared by the next scan's own prologue (RmRows), or by the remove kernel in
    // front of any other reader of the flags
    for (int64_t i = 0; i < n; ++i) h->pend_rm.push_back((int32_t)rows[i]);
    if (!g->inline_rm) flush_pending_rm(h);   // (A/B: one launch per emission, as until round 6)
    h->n_live -= n;
}

// What to put into the free medoid slots of a pass that has to run anyway.  sample_medoid is a pure function of the live
// matrix (cluster.py:606-637), so anything scanned now is exactly what a scan at the time of use returns -- as long as no
// row removed in between lies inside the medoid radius of the scanned row (vh_gen_next re-validates every cached entry at
// every emission).  In the order the walk will want them: the next live seeds (cluster.py:342-384, peeked at without
// touching the walk's state: dead stretches are not tombstoned), and for a seed whose statistics are already cached with
// its within-radius list, the rows of that list -- the pool wander_medoid draws its first candidates from
// (cluster.py:415-450).  A C2 sweep spent 446 k of its 586 k passes on candidate rounds (profiles/r03g_sweep_pass_purposes.txt);
// a round whose candidates were all scanned ahead needs no pass at all.
// seed_limit / depth_limit (0: the generator's own): a bounded walk over the first few upcoming seeds only
void gen_speculative_fill(vh_gen* g, size_t want, const std::vector<int64_t>& exclude, std::vector<int64_t>& out, int seed_limit = 0,
                          int depth_limit = 0) {
    const int64_t n_order = (int64_t)g->order.size();
    const int window = seed_limit > 0 ? std::min(seed_limit, g->spec_window) : g->spec_window;
    const int max_depth = depth_limit > 0 ? std::min(depth_limit, g->spec_depth) : g->spec_depth;
    auto taken = [&](int64_t row) {
        return gen_lookup(g, row) != nullptr || std::find(exclude.begin(), exclude.end(), row) != exclude.end() ||
               std::find(out.begin(), out.end(), row) != out.end();
    };
    // the upcoming live seeds, in walk order
    std::vector<int64_t> upcoming;
    int64_t looked = 0;
    for (int64_t i = g->order_index; i < n_order && (int)upcoming.size() < window && looked < 4096; ++i, ++looked) {
        const int64_t o = g->order[(size_t)i];
        if (o == -1 || !g->alive[(size_t)o]) continue;
        vh_gen::RowMemo& m = g->row_memo[(size_t)o & 255u];
        if (m.orig != o || m.epoch != g->rows_epoch) {
            m.orig = o;
            m.epoch = g->rows_epoch;
            m.row = (int64_t)(std::lower_bound(g->indices.begin(), g->indices.end(), o) - g->indices.begin());
        }
        upcoming.push_back(m.row);
    }
    // seeds first (most of them turn out to be loners: their scan is all they need) ...
    for (int64_t row : upcoming) {
        if (out.size() >= want) return;
        if (!taken(row)) out.push_back(row);
    }
    // ... then the candidate pools of the seeds whose statistics are already there (the first round of their hill climb), and
    // the pools of those candidates (where the climb can move to)
    if (!g->spec_neighbours) return;
    auto absent = [&](int64_t r) {   // (presence only: a stale entry is simply not refreshed ahead of time; a row removed since
                                     // the scan that listed it is never scheduled)
        return g->kept[(size_t)r] != 0 && g->stats.count(r) == 0 && !gen_is_pending(g, r) &&
               std::find(exclude.begin(), exclude.end(), r) == exclude.end() && std::find(out.begin(), out.end(), r) == out.end();
    };
    for (int depth = 1; depth <= max_depth; ++depth) {
        for (int64_t row : upcoming) {
            const GenStats* seed_st = gen_lookup(g, row);
            if (seed_st == nullptr || !seed_st->have_list) continue;
            // (no copy: gen_lookup below erases or inserts OTHER rows' entries only -- r != row -- and an unordered_map keeps
            // references to its elements valid across both)
            const std::vector<int64_t>& pool = seed_st->within;
            for (int64_t r : pool) {
                if (out.size() >= want) return;
                if (r == row) continue;
                if (depth == 1) {
                    if (absent(r)) out.push_back(r);
                    continue;
                }
                if (g->kept[(size_t)r] == 0) continue;
                const GenStats* nb = gen_lookup(g, r);   // (validated against the removal log: a stale list would name dead rows)
                if (nb == nullptr || !nb->have_list) continue;
                const std::vector<int64_t>& pool2 = nb->within;   // (nothing below touches the map)
                for (int64_t r2 : pool2) {
                    if (out.size() >= want) return;
                    if (r2 != r && r2 != row && absent(r2)) out.push_back(r2);
                }
            }
        }
    }
}

// sample_medoid's device half for every medoid not cached yet (one pass per <= 32 of them)
void gen_ensure_stats(vh_gen* g, const int64_t* medoids, size_t n) {
    std::vector<int64_t> missing;
    for (size_t i = 0; i < n; ++i) {
        const int64_t m = medoids[i];
        if (gen_lookup(g, m) != nullptr) continue;
        if (std::find(missing.begin(), missing.end(), m) != missing.end()) continue;
        missing.push_back(m);
    }
    if (missing.empty()) return;
    // Speculation: sample_medoid is a pure function of the live matrix, so the statistics of UPCOMING seeds can be taken
    // in the same pass (free slots of the medoid-count bucket; a lone seed scan is widened).  They are used only while no
    // row removed since lies within the histogram range of the seed (vh_gen_next), i.e. while they are exactly what a
    // scan at the time of use would return.
    // requested entries that were scanned ahead are in use from now on
    for (size_t i = 0; i < n; ++i) {
        const auto it = g->stats.find(medoids[i]);   // (validated by the loop above)
        if (it != g->stats.end() && it->second.spec) { it->second.spec = false; g->spec_used++; }
    }
    size_t n_needed = missing.size();
    if (g->speculate) {
        size_t target = (size_t)pick_km((int)std::min<size_t>(missing.size(), kMaxMedoids));
        if (g->clu->max_k < kMaxMedoids) target = std::min(missing.size(), (size_t)g->clu->max_k);   // wide latents: no widening
        else if (scan_uses_mfma(g->clu, (int)std::min<size_t>(missing.size(), kMaxMedoids))) target = kMaxMedoids;   // matrix-pipe pass: 32 medoids cost what 9 cost
        else if (g->clu->n_rows <= 600000) target = kMaxMedoids;  // latency-bound pass: extra medoids are free (the limit re-measured
                                                                   // in round 6 with the cheaper 32-slot pass: 400 k / 600 k / 900 k / 1.3 M /
                                                                   // 2.1 M rows give 11.55 / 10.7-11.4 / 10.9 / 11.1-11.2 / 11.3 s per C2 sweep)
        else if (g->spec_big_target > 0 && missing.size() <= 8) target = std::max(target, (size_t)g->spec_big_target);
        else if (missing.size() == 1) target = 8;
        if (missing.size() < target && missing.size() < (size_t)kMaxMedoids) {
            std::vector<int64_t> extra;
            GenTimer tf(&g->t_fill);
            const size_t want = target - missing.size();
            bool walk = true;
            if (g->prefill_ready && g->prefill_epoch == g->rows_epoch) {
                // what the list cannot know: the pools of the seeds whose statistics the pass it was collected under delivered
                if (g->prefill_on == 2) gen_speculative_fill(g, want, missing, extra, g->prefill_fresh_seeds, 1);
                for (int64_t r : g->prefill) {
                    if (extra.size() >= want) break;
                    if (g->kept[(size_t)r] != 0 && g->stats.count(r) == 0 && !gen_is_pending(g, r) &&
                        std::find(missing.begin(), missing.end(), r) == missing.end() &&
                        std::find(extra.begin(), extra.end(), r) == extra.end())
                        extra.push_back(r);
                }
                // a list that did not fill its own slots had walked every pool: walking them again one pass later finds little
                walk = extra.size() < want && !g->prefill_exhausted;
                (walk ? g->prefill_topups : g->prefill_hits)++;
            }
            g->prefill_ready = false;
            if (walk) gen_speculative_fill(g, want, missing, extra);
            missing.insert(missing.end(), extra.begin(), extra.end());
        }
    }
    const size_t max_k = (size_t)g->clu->max_k;
    for (size_t lo = 0; lo < missing.size(); lo += max_k) {
        const int k = (int)std::min<size_t>(max_k, missing.size() - lo);
        const uint64_t seq = g->clu->scan_seq;
        int slot;
        {
            GenTimer t(&g->t_scan);
            GenTimer t2(&g->t_km[k]);
            const bool last_chunk = lo + max_k >= missing.size();
            const std::function<void()> under_the_pass = [g, &missing, last_chunk] {
                {
                    GenTimer th(&g->t_book_hidden);
                    gen_flush_pending(g);
                }
                if (g->prefill_on && g->speculate && last_chunk) {   // the next pass's speculative rows, collected under this one
                    GenTimer tp(&g->t_fill_hidden);
                    g->prefill.clear();
                    g->pass_in_flight = true;
                    gen_speculative_fill(g, (size_t)kMaxMedoids, missing, g->prefill);
                    g->pass_in_flight = false;
                    g->prefill_exhausted = g->prefill.size() < (size_t)kMaxMedoids;
                    g->prefill_epoch = g->rows_epoch;
                    g->prefill_ready = true;
                }
            };
            slot = gen_scan(g, k, missing.data() + lo, &under_the_pass);
        }
        g->n_km[k]++;
        g->rows_km[k] += g->clu->n_rows;
        (g->pass_purpose == 0 ? g->pass_seed : g->pass_cand)++;
        g->scan_passes++;
        g->scan_medoids += k;
        g->rows_streamed += g->clu->n_rows;
        g->live_rows_streamed += g->clu->n_live;
        gen_collect_ms(g);
        GenTimer tb(&g->t_book);
        for (int j = 0; j < k; ++j) {
            const bool spec = lo + (size_t)j >= n_needed;
            // what the caller asked for becomes an entry now; what was scanned ahead waits for the next pass (gen_lookup
            // materialises the ones that are wanted before that)
            if (spec && g->defer_book) {
                if (g->pending_mark.size() < g->kept.size()) g->pending_mark.assign(g->kept.size(), 0);
                g->pending_mark[(size_t)missing[lo + j]] = 1;
                g->pending.push_back(vh_gen::Pending{missing[lo + j], slot, j, seq, g->n_emitted});
            }
            else gen_materialise(g, missing[lo + j], slot, j, seq, g->n_emitted, spec);
        }
    }
}

int64_t gen_select(vh_gen* g, int64_t medoid, float threshold, bool remove) {
    int64_t n = 0;
    GenTimer t(&g->t_select);
    (remove ? g->pass_select : g->pass_listsel)++;
    if (g->comm) {
        const float* q = gen_host_rows(g) + (size_t)g->indices[(size_t)medoid] * g->clu->L;
        n = select_sharded_core(g->clu, gen_local_row(g, medoid), q, threshold, remove ? 1 : 0, g->offsets.data(), g->sel);
    } else {
        g->sel.resize((size_t)std::max<int64_t>(1, g->clu->n_rows));
        gen_check(vh_clu_select(g->clu, medoid, nullptr, threshold, remove ? 1 : 0, g->sel.data(), (int64_t)g->sel.size(), &n));
    }
    g->scan_passes++;
    g->rows_streamed += g->clu->n_rows;
    g->live_rows_streamed += g->clu->n_live + (remove ? n : 0);   // the rows that were live when the pass ran
    gen_collect_ms(g);
    return n;
}

const std::vector<int64_t>& gen_within(vh_gen* g, int64_t medoid) {
    GenStats& st = g->stats.at(medoid);
    if (!st.have_list) {
        const bool in_ring = g->clu->scan_seq - st.seq <= (uint64_t)kListRing && st.seq >= g->ring_rows_valid_from;
        if (in_ring && st.list_count <= (unsigned int)kListCap) {
            const int slot = (int)(st.seq % kListRing);
            g->clu->h_sel.resize(st.list_count);
            if (st.list_count)
                memcpy(g->clu->h_sel.data(), g->clu->lists + ((size_t)slot * kMaxMedoids + st.slot_j) * kListCap,
                       (size_t)st.list_count * sizeof(int32_t));
            std::sort(g->clu->h_sel.begin(), g->clu->h_sel.end());
            st.within.assign(g->clu->h_sel.begin(), g->clu->h_sel.end());
        } else {
            const int64_t n = gen_select(g, medoid, 0.05f, false);
            st.within.assign(g->sel.begin(), g->sel.begin() + n);
        }
        st.have_list = true;
    }
    return st.within;
}

// cluster.py:342-384
int64_t gen_next_seed(vh_gen* g) {
    GenTimer t(&g->t_seed);
    int64_t n_order = (int64_t)g->order.size();
    int64_t i = g->order_index - 1;
    while (true) {
        i = (i + 1) % n_order;
        if (i == 0 && g->n_emitted > 0) {   // pack_order (cluster.py:337-340)
            g->order.erase(std::remove(g->order.begin(), g->order.end(), (int64_t)-1), g->order.end());
            n_order = (int64_t)g->order.size();
            if (n_order == 0) throw ::vh::HipError{hipErrorUnknown, "seed order exhausted", __FILE__, __LINE__};
        }
        const int64_t o = g->order[i];
        if (o == -1) continue;
        if (!g->alive[o]) { g->order[i] = -1; continue; }
        g->order_index = i + 1;
        const auto it = std::lower_bound(g->indices.begin(), g->indices.end(), o);
        return (int64_t)(it - g->indices.begin());
    }
}

// cluster.py:386-413
void gen_update_successes(vh_gen* g, bool success) {
    if (g->attempts.size() == g->windowsize) {
        g->successes -= g->attempts.front() ? 1 : 0;
        g->attempts.pop_front();
    }
    g->successes += success ? 1 : 0;
    g->attempts.push_back(success);
    if (g->attempts.size() == g->windowsize && g->successes < g->minsuccesses) {
        g->pvr += 0.1;
        g->attempts.clear();
        g->successes = 0;
        g->order_index = 0;
    }
}

// cluster.py:415-450
int64_t gen_wander(vh_gen* g, int64_t seed) {
    int64_t medoid = seed;
    // rows this walk has sampled so far (cluster.py:425 `tried`): a list to undo the marks with, a byte per row to test membership --
    // a pool of a large cluster is up to kListCap rows, and each round used to compare every one of them with the whole list
    std::vector<int64_t> tried{medoid};
    if (g->tried_mark.size() < g->kept.size()) g->tried_mark.assign(g->kept.size(), 0);
    g->tried_mark[(size_t)medoid] = 1;
    struct Unmark {
        vh_gen* g; std::vector<int64_t>& t;
        ~Unmark() { for (int64_t r : t) g->tried_mark[(size_t)r] = 0; }
    } unmark{g, tried};
    g->seeds_total++;
    if (gen_lookup(g, seed) != nullptr) g->seeds_cached++;
    g->pass_purpose = 0;
    gen_ensure_stats(g, &seed, 1);
    double local_density = g->stats.at(seed).density;
    auto untried = [&](const std::vector<int64_t>& rows) {
        std::vector<int64_t> c;
        c.reserve(rows.size());
        for (int64_t r : rows)
            if (g->tried_mark[(size_t)r] == 0) c.push_back(r);
        return c;
    };
    std::vector<int64_t> pool, candidates;
    auto draw = [&](int64_t from) {
        const std::vector<int64_t>* w;
        {
            GenTimer tw(&g->t_within);
            w = &gen_within(g, from);
        }
        GenTimer ts(&g->t_sample);
        pool = untried(*w);
        g->rng.sample(pool, (int)std::min<size_t>(pool.size(), (size_t)g->maxsteps), candidates);
    };
    draw(seed);
    size_t i = 0;
    while (i < candidates.size()) {
        // look ahead: every not-yet-scanned candidate of this round shares one matrix pass
        if (i == 0) {
            size_t miss = 0;
            for (int64_t c : candidates) miss += gen_lookup(g, c) != nullptr ? 0 : 1;
            g->cand_rounds++;
            g->cand_needed += (int64_t)miss;
            g->cand_rounds_cached += miss == 0 ? 1 : 0;
        }
        // ONE call per round: it leaves every candidate of the round with valid statistics (all the missing ones are scanned, in
        // chunks if need be), and nothing inside a walk can change that -- entries are validated against emissions, and none
        // happens here.  (Until round 6 it was repeated in front of every candidate; the lookups it repeated are cached and
        // cheap -- the sweep measured the same, profiles/r06y4_sweep_ab.txt -- so this is tidiness, not speed.)
        if (i == 0) {
            g->pass_purpose = 1;
            gen_ensure_stats(g, candidates.data(), candidates.size());
        }
        const int64_t sampled = candidates[i];
        tried.push_back(sampled);
        g->tried_mark[(size_t)sampled] = 1;
        const double d = g->stats.at(sampled).density;
        if (d > local_density) {
            medoid = sampled;
            local_density = d;
            g->wander_moves++;
            draw(sampled);
            i = 0;
        } else {
            ++i;
        }
    }
    return medoid;
}

// cluster.py:452-543 on the exact histogram of the scan
void gen_fetch_hist(vh_gen* g, int64_t medoid, GenStats& st) {
    // the histogram reaches out to 0.3: it is the scan's as long as no row removed since lies within that range of the medoid
    if (st.hist_checked != g->n_emitted) {
        if (!st.hist_stale && gen_touched_since(g, st.vec.data(), st.hist_checked, true)) st.hist_stale = true;
        if (st.hist_stale) st.have_hist = false;
        st.hist_checked = g->n_emitted;
    }
    if (st.have_hist) return;
    if (st.hist_stale || g->clu->scan_seq - st.seq > (uint64_t)kListRing) {
        // rows in range were removed, or the scan has left the ring: one more pass for this medoid alone
        const uint64_t seq = g->clu->scan_seq;
        {
            GenTimer t(&g->t_scan);
            const std::function<void()> under_the_pass = [g] {   // (entries are never moved by an insert: `st` stays valid)
                GenTimer th(&g->t_book_hidden);
                gen_flush_pending(g);
            };
            (void)gen_scan(g, 1, &medoid, &under_the_pass);
        }
        g->scan_passes++;
        g->pass_hist++;
        g->scan_medoids += 1;
        g->rows_streamed += g->clu->n_rows;
        g->live_rows_streamed += g->clu->n_live;
        gen_collect_ms(g);
        if (!st.have_list) {   // (the near field is unchanged: the new scan's list is the same list, and it is in the ring)
            st.seq = seq;
            st.slot_j = 0;
            st.list_count = g->clu->last_counts[(int)(seq % kListRing)][0];
        }
        unsigned long long tmp0[VH_NBINS];
        wait_for_hist(g->clu, seq + 1);
        memcpy(tmp0, g->clu->hist((int)(seq % kListRing)), sizeof(tmp0));
        for (int b = 0; b < VH_NBINS; ++b) st.hist_fx[b] = (int64_t)tmp0[b];
        st.have_hist = true;
        st.hist_stale = false;
        return;
    }
    if (st.born != g->n_emitted) g->hist_kept++;
    unsigned long long tmp[VH_NBINS];
    wait_for_hist(g->clu, st.seq + 1);   // (the scan's pass flag has been seen; its histograms may still be on their way)
    memcpy(tmp, g->clu->hist((int)(st.seq % kListRing)) + (size_t)st.slot_j * VH_NBINS, sizeof(tmp));
    for (int b = 0; b < VH_NBINS; ++b) st.hist_fx[b] = (int64_t)tmp[b];
    st.have_hist = true;
}

ThresholdKind gen_find_threshold(const vh_gen* g, const GenStats& st, double* threshold, double* observed_pvr) {
    if (st.n_lt == 1) return kLoner;
    float hist[VH_NBINS];
    for (int b = 0; b < VH_NBINS; ++b) hist[b] = (float)((double)st.hist_fx[b] / VH_HIST_SCALE);
    // densities[k] = sum_i pdf[k - i] * hist[i]: float32 products, float32 running sum, i ascending
    // (cluster.py:495-500), then the middle 60 of the 90 values
    float dens[VH_NBINS];
    for (int k = 15; k < 15 + VH_NBINS; ++k) {
        float acc = 0.0f;
        for (int i = std::max(0, k - (kPdfLen - 1)); i <= std::min(VH_NBINS - 1, k); ++i) {
            const float w = 0.005f * kPdfRaw[k - i];
            const float prod = w * hist[i];
            acc = acc + prod;
        }
        dens[k - 15] = acc;
    }
    double peak_density = 0.0, minimum_x = 0.0, density_at_minimum = 0.0, thr = 0.0;
    bool peak_over = false, have_thr = false;
    const double delta_x = 0.3 / (double)VH_NBINS;
    double x = 0.0;
    for (int b = 0; b < VH_NBINS; ++b) {
        const double density = (double)dens[b];
        if (!peak_over && density > peak_density) {
            if (x > 0.1) return kNoThreshold;
            peak_density = density;
        }
        if (!peak_over && density < 0.6 * peak_density) {
            peak_over = true;
            density_at_minimum = density;
        }
        if (peak_over && density > 1.5 * density_at_minimum) break;
        if (peak_over && density < density_at_minimum) {
            minimum_x = x;
            density_at_minimum = density;
            if (density < g->pvr * peak_density) { thr = minimum_x; have_thr = true; }
        }
        x += delta_x;
    }
    if (!have_thr || thr > 0.2 + g->pvr) return kNoThreshold;
    *threshold = thr;
    *observed_pvr = density_at_minimum / peak_density;
    return kThreshold;
}

// Cluster.seed is the seed's index in the reference's PACKED matrix = the number of live rows before it
// (cluster.py:553, 570, 590).  Counted with a Fenwick tree over the physical rows (O(log n) per query / removal;
// a linear count is 10 M byte-adds per cluster at config C4).
void gen_bit_build(vh_gen* g) {
    const size_t n = g->kept.size();
    g->bit.assign(n + 1, 0);
    for (size_t i = 1; i <= n; ++i) {
        g->bit[i] += 1;
        const size_t j = i + (i & (~i + 1));
        if (j <= n) g->bit[j] += g->bit[i];
    }
}
void gen_bit_remove(vh_gen* g, int64_t row) {
    for (size_t i = (size_t)row + 1; i < g->bit.size(); i += i & (~i + 1)) g->bit[i] -= 1;
}
int64_t gen_logical_index(vh_gen* g, int64_t row) {
    GenTimer t(&g->t_logical);
    int64_t c = 0;
    for (size_t i = (size_t)row; i > 0; i -= i & (~i + 1)) c += g->bit[i];
    return c;
}

// Before the resident matrix is compacted: everything that still refers to the ring of the scans (pending speculative results,
// within-radius lists not copied to the host yet) is brought to the host while the ring's row numbers are still the current ones.
void gen_prepare_pack(vh_gen* g) {
    gen_flush_pending(g);
    for (auto& kv : g->stats) {
        GenStats& st = kv.second;
        if (st.have_list || st.list_count > (unsigned int)kListCap) continue;
        if (g->clu->scan_seq - st.seq > (uint64_t)kListRing || st.seq < g->ring_rows_valid_from) continue;
        const int32_t* src = g->clu->lists + ((size_t)(st.seq % kListRing) * kMaxMedoids + st.slot_j) * kListCap;
        st.within.assign(src, src + st.list_count);
        std::sort(st.within.begin(), st.within.end());
        st.have_list = true;
    }
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

vh_gen* gen_create_common(vh_clu* clu, vh_comm* comm, const int64_t* order, int64_t n, int maxsteps, int windowsize, int minsuccesses,
                          uint64_t rng_seed, double pack_fraction, int64_t pack_min_rows) {
    VH_REQUIRE(maxsteps >= 1, "maxsteps must be a positive integer, not %d", maxsteps);
    VH_REQUIRE(windowsize >= 1, "windowsize must be at least 1, not %d", windowsize);
    VH_REQUIRE(minsuccesses >= 1 && minsuccesses <= windowsize, "minsuccesses must be between 1 and windowsize, not %d",
               minsuccesses);
    std::unique_ptr<vh_gen> g(new vh_gen());
    g->clu = clu;
    g->comm = comm;
    g->maxsteps = maxsteps;
    g->windowsize = (size_t)windowsize;
    g->minsuccesses = minsuccesses;
    g->pack_fraction = pack_fraction;
    g->pack_min_rows = pack_min_rows;
    g->rng.seed(rng_seed);
    g->profile = option("gen.profile", 0) != 0;
    g->speculate = option("gen.speculate", 1) != 0;
    g->spec_window = (int)option("gen.spec_window", kSpecWindow);
    g->prefill_on = (int)option("gen.prefill", 2);
    g->prefill_fresh_seeds = 4;
    g->inline_rm = option("gen.inline_removals", 1) != 0;
    g->spec_neighbours = true;
    g->max_entry_age = kMaxEntryAgeDefault;
    g->spec_depth = 2;
    g->defer_book = true;
    g->spec_big_target = 0;
    g->order.assign(order, order + n);
    g->indices.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) g->indices[(size_t)i] = i;
    g->kept.assign((size_t)n, 1);
    g->alive.assign((size_t)n, 1);
    gen_bit_build(g.get());
    g->n_remaining = n;
    return g.release();
}

}  // namespace

extern "C" {

int vh_gen_create(vh_clu* clu, const int64_t* order, int64_t n, int maxsteps, int windowsize, int minsuccesses,
                  uint64_t rng_seed, double pack_fraction, int64_t pack_min_rows, vh_gen** out) {
    return guarded([&] {
        VH_REQUIRE(clu != nullptr && order != nullptr && out != nullptr, "NULL argument");
        VH_REQUIRE(n == clu->n_rows && n >= 1, "order length does not match the matrix");
        *out = gen_create_common(clu, nullptr, order, n, maxsteps, windowsize, minsuccesses, rng_seed, pack_fraction, pack_min_rows);
    });
}

// The state machine over a ROW-SHARDED matrix: `clu` is this rank's shard with a communicator attached (vh_clu_attach_comm);
// `order` = np.argsort(lengths)[::-1] of the GLOBAL lengths (n_global entries, identical on every rank).  Collective: every
// rank calls it, and afterwards vh_gen_next, in lock step; every rank receives the same clusters with GLOBAL row indices.
int vh_gen_create_sharded(vh_clu* clu, const int64_t* order, int64_t n_global, int maxsteps, int windowsize, int minsuccesses,
                          uint64_t rng_seed, double pack_fraction, int64_t pack_min_rows, vh_gen** out) {
    return guarded([&] {
        VH_REQUIRE(clu != nullptr && order != nullptr && out != nullptr, "NULL argument");
        VH_REQUIRE(clu->comm != nullptr, "vh_clu_attach_comm has not been called");
        VH_REQUIRE(clu->n_rows == clu->n_live && (int64_t)clu->host_rows.size() == clu->n_rows * clu->L,
                   "the sharded state machine needs a freshly created handle");
        vh_comm* comm = clu->comm;
        const int world = comm->world, L = clu->L;
        // shard sizes -> offsets
        clu->xch_counts.ensure((size_t)world + 1);
        const uint32_t mine = (uint32_t)clu->n_rows;
        VH_HIP(hipMemcpyAsync(clu->xch_counts.p + world, &mine, 4, hipMemcpyHostToDevice, clu->stream));
        rccl_allgather_u32(comm, clu->xch_counts.p + world, clu->xch_counts.p, 1, clu->stream);
        std::vector<uint32_t> sizes((size_t)world);
        VH_HIP(hipMemcpyAsync(sizes.data(), clu->xch_counts.p, 4 * sizes.size(), hipMemcpyDeviceToHost, clu->stream));
        VH_HIP(hipStreamSynchronize(clu->stream));
        std::vector<int64_t> off((size_t)world + 1, 0);
        for (int r = 0; r < world; ++r) off[(size_t)r + 1] = off[(size_t)r] + sizes[(size_t)r];
        VH_REQUIRE(off.back() == n_global, "order length %lld does not match the %lld rows of all shards", (long long)n_global,
                   (long long)off.back());
        VH_REQUIRE(n_global < ((int64_t)1 << 31) - 4096, "more than 2^31 rows in total are not supported");
        std::unique_ptr<vh_gen> g(gen_create_common(clu, comm, order, n_global, maxsteps, windowsize, minsuccesses, rng_seed,
                                                    pack_fraction, pack_min_rows));
        g->offsets = off;
        // The whole normalised matrix in HOST memory on every rank: the state machine reads rows of ANY shard all the time -- the query
        // vector of every medoid it scans (most live in another shard), the row-by-row tests of the lazy validation against the
        // rows an emission removed -- so fetching them on demand would put a collective on every such read.  Cost: n_global * L * 4
        // bytes of host memory per rank (C3: 256 MB, C4 10 M x 64: 2.56 GB; an 8-GPU node holds 8 copies).  Device memory is NOT
        // part of that: the gather goes through a bounded staging buffer, row chunk by row chunk (the one-shot gather
        // held the whole matrix on every device while the generator was created).
        const size_t max_rows = *std::max_element(sizes.begin(), sizes.end());
        const size_t kStageBytes = (size_t)std::max<int64_t>(1, option("gen.gather_stage_bytes", (int64_t)64 << 20));   // receive staging per rank
        const size_t chunk_rows = std::max<size_t>(1, std::min<size_t>(std::max<size_t>(max_rows, 1),
                                                                        kStageBytes / ((size_t)world * (size_t)L * sizeof(float))));
        const size_t block = chunk_rows * (size_t)L;
        DevBuf<float> send, recv;
        send.alloc(block);
        recv.alloc(block * (size_t)world);
        g->rows_global.resize((size_t)n_global * L);
        for (size_t r0 = 0; r0 < max_rows; r0 += chunk_rows) {
            const size_t mine_rows = (size_t)clu->n_rows > r0 ? std::min(chunk_rows, (size_t)clu->n_rows - r0) : 0;
            VH_HIP(hipMemsetAsync(send.p, 0, block * sizeof(float), clu->stream));
            if (mine_rows)
                VH_HIP(hipMemcpyAsync(send.p, clu->host_rows.data() + r0 * (size_t)L, mine_rows * (size_t)L * sizeof(float),
                                      hipMemcpyHostToDevice, clu->stream));
            rccl_allgather_bytes(comm, send.p, recv.p, block * sizeof(float), clu->stream);
            for (int r = 0; r < world; ++r) {
                const size_t have = (size_t)sizes[(size_t)r];
                if (have <= r0) continue;
                const size_t take = std::min(chunk_rows, have - r0);
                VH_HIP(hipMemcpyAsync(g->rows_global.data() + ((size_t)off[(size_t)r] + r0) * L, recv.p + (size_t)r * block,
                                      take * (size_t)L * sizeof(float), hipMemcpyDeviceToHost, clu->stream));
            }
            VH_HIP(hipStreamSynchronize(clu->stream));   // (the staging buffers are reused by the next chunk)
        }
        VH_HIP(hipStreamSynchronize(clu->stream));
        *out = g.release();
    });
}

int vh_gen_destroy(vh_gen* g) {
    if (g && g->profile)
        fprintf(stderr, "[vambhip] generator: total %.1f ms = scans %.1f + selects %.1f + seed walk %.1f + logical index %.1f + rest %.1f; "
                "%lld passes, %lld medoids; speculative seed scans %lld, used %lld, invalidated %lld\n",
                g->t_total, g->t_scan, g->t_select, g->t_seed, g->t_logical,
                g->t_total - g->t_scan - g->t_select - g->t_seed - g->t_logical, (long long)g->scan_passes,
                (long long)g->scan_medoids, (long long)g->spec_scanned, (long long)g->spec_used, (long long)g->spec_dropped);
    if (g && g->profile) {
        fprintf(stderr, "[vambhip]   host time inside 'rest': lazy validation %.1f ms (part of it inside the fill), speculative fill %.1f ms, "
                "post-scan bookkeeping %.1f ms (+ %.1f ms under the next pass), removal log + eviction %.1f ms; fill one pass ahead: %.1f ms "
                "under the passes, %lld lists used as they were, %lld topped up\n", g->t_validate, g->t_fill, g->t_book, g->t_book_hidden, g->t_emit,
                g->t_fill_hidden, g->prefill_hits, g->prefill_topups);
        fprintf(stderr, "[vambhip]   inclusive times of an emission's pieces: walk %.1f ms (of it: within-radius lists %.1f, untried + sample %.1f), "
                "histogram %.1f, threshold %.1f, member copy %.1f, live flags + rank tree %.1f, rest of the emission incl. compaction %.1f\n",
                g->t_wander_incl, g->t_within, g->t_sample, g->t_hist_incl, g->t_threshold, g->t_members, g->t_live, g->t_pack_incl);
        fprintf(stderr, "[vambhip]   passes by purpose: seed scans %lld, candidate rounds %lld, histogram re-scans %lld, selects %lld (+ %lld list selects); "
                "seeds %lld (cached at arrival %lld), candidate rounds %lld (fully cached %lld, %lld candidates to scan), medoid moves %lld; "
                "cached entries per emission %.1f; lazy validations %lld (%lld cluster tests, %lld row tests, %lld invalid), histograms reused %lld\n",
                (long long)g->pass_seed, (long long)g->pass_cand, (long long)g->pass_hist, (long long)g->pass_select, (long long)g->pass_listsel,
                (long long)g->seeds_total, (long long)g->seeds_cached, (long long)g->cand_rounds, (long long)g->cand_rounds_cached,
                (long long)g->cand_needed, (long long)g->wander_moves,
                g->kept_emissions ? (double)g->kept_entries / (double)g->kept_emissions : 0.0, (long long)g->lazy_checks,
                (long long)g->lazy_cluster_tests, (long long)g->lazy_point_tests, (long long)g->lazy_invalid, (long long)g->hist_kept);
        for (int k = 1; k <= 32; ++k)
            if (g->n_km[k])
                fprintf(stderr, "[vambhip]   passes with %2d medoids: %8lld, %7.1f ms, avg %6.1f us, avg rows %9.0f\n", k,
                        (long long)g->n_km[k], g->t_km[k], 1e3 * g->t_km[k] / (double)g->n_km[k],
                        (double)g->rows_km[k] / (double)g->n_km[k]);
    }
    delete g;
    return VH_OK;
}

}  // extern "C"

namespace {

// __next__ (cluster.py:298-316) + find_cluster (cluster.py:545-604).  info->n_members == 0: StopIteration.
void gen_next_impl(vh_gen* g, vh_cluster_info* info, int64_t* members, int64_t cap) {
    {
        VH_REQUIRE(g != nullptr && info != nullptr && members != nullptr, "NULL argument");
        memset(info, 0, sizeof(*info));
        if (g->n_remaining == 0) return;
        GenTimer t_all(&g->t_total);
        int64_t n_points = 0;
        std::vector<int64_t> points;
        int64_t emitted_medoid = -1;   // physical row and radius of the cluster being emitted (validity check below)
        double emitted_radius = 0.0;
        while (true) {
            const int64_t seed = gen_next_seed(g);
            int64_t medoid;
            {
                GenTimer tw(&g->t_wander_incl);
                medoid = gen_wander(g, seed);
            }
            GenStats& st = g->stats.at(medoid);
            double threshold = 0.0, observed = 0.0;
            if (st.n_lt != 1) {
                GenTimer th(&g->t_hist_incl);
                gen_fetch_hist(g, medoid, st);
            }
            ThresholdKind kind;
            {
                GenTimer tt(&g->t_threshold);
                kind = gen_find_threshold(g, st, &threshold, &observed);
            }
            const int64_t original = g->indices[(size_t)medoid];
            info->medoid = original;
            info->maximal_pvr = g->pvr;
            if (kind == kLoner) {
                info->seed = gen_logical_index(g, seed);
                info->kind = 1;
                info->successes = g->successes;
                info->attempts = (int64_t)g->attempts.size();
                points.assign(1, medoid);
                gen_remove_live(g, points.data(), 1);
                emitted_medoid = medoid;
                emitted_radius = 0.0;
                break;
            }
            if (kind == kNoThreshold) {
                if (g->pvr > 0.55) {
                    info->seed = gen_logical_index(g, seed);
                    info->kind = 2;
                    info->radius = 0.06;
                    info->successes = g->successes;
                    info->attempts = (int64_t)g->attempts.size();
                    n_points = gen_select(g, medoid, (float)0.06, true);
                    points.assign(g->sel.begin(), g->sel.begin() + n_points);
                    emitted_medoid = medoid;
                    emitted_radius = 0.06;
                    break;
                }
                gen_update_successes(g, false);
                continue;
            }
            info->seed = gen_logical_index(g, seed);
            info->kind = 0;
            info->radius = threshold;
            info->observed_pvr = observed;
            info->successes = g->successes;
            info->attempts = (int64_t)g->attempts.size();
            n_points = gen_select(g, medoid, (float)threshold, true);
            points.assign(g->sel.begin(), g->sel.begin() + n_points);
            emitted_medoid = medoid;
            emitted_radius = threshold;
            if (g->pvr < 0.55) gen_update_successes(g, true);
            break;
        }
        VH_REQUIRE((int64_t)points.size() <= cap, "members buffer too small");
        {
            GenTimer tm(&g->t_members);
            for (size_t i = 0; i < points.size(); ++i) members[i] = g->indices[(size_t)points[i]];
        }
        info->n_members = (int64_t)points.size();
        // __next__ bookkeeping.  The reference clears its sample_medoid cache here because a removal may change any cached
        // result (cluster.py:298-316).  Here the removal goes into a log and cached entries are validated against it when they
        // are looked at (gen_lookup / gen_fetch_hist): an entry nobody asks for again costs nothing, one that is asked for pays
        // one dot product per emission since its last check (+ a row-by-row comparison for the few clusters near it).
        {
            GenTimer te(&g->t_emit);
            const double t_cluster = std::min(0.5, std::max(0.0, emitted_radius) + 2e-3);
            const double a_t = std::acos(1.0 - 2.0 * t_cluster);
            const double a_near = a_t + std::acos(1.0 - 2.0 * (0.05 + 2e-3)) + 1e-3;
            const double a_far = a_t + std::acos(1.0 - 2.0 * (0.3 + 2e-3)) + 1e-3;
            vh_gen::Removal R;
            R.first = g->removed_base + (int64_t)g->removed_orig.size();
            R.count = (int64_t)points.size();
            {
                const float* mv = gen_host_rows(g) + (size_t)g->indices[(size_t)emitted_medoid] * g->clu->L;
                g->rlog_medoid.insert(g->rlog_medoid.end(), mv, mv + g->clu->L);
            }
            R.cos_near = a_near < 3.14 ? (float)std::cos(a_near) : -3.0f;
            R.cos_far = a_far < 3.14 ? (float)std::cos(a_far) : -3.0f;
            for (int64_t r : points) g->removed_orig.push_back(g->indices[(size_t)r]);
            g->rlog.push_back(R);   // rlog_base + rlog.size() == n_emitted + 1 from here on
            g->kept_entries += (int64_t)g->stats.size();
            g->kept_emissions++;
            // entries the walk has left behind: dropped in bulk now and then (their lazy check would walk a long log)
            if ((g->n_emitted & 63) == 63 || g->stats.size() > kMaxCached) {
                for (auto it = g->stats.begin(); it != g->stats.end();)
                    it = (g->n_emitted - it->second.born > g->max_entry_age || g->stats.size() > 2 * kMaxCached) ? g->stats.erase(it) : std::next(it);
            }
            // ... and the log records nobody can ask for any more (every surviving entry was born, hence last checked, at most
            // max_entry_age emissions ago; pending entries are younger still)
            if ((g->n_emitted & 8191) == 8191) {
                const int64_t keep_from = g->n_emitted - g->max_entry_age - 64;
                if (keep_from > g->rlog_base) {
                    const size_t drop = (size_t)(keep_from - g->rlog_base);
                    const int64_t first_kept = g->rlog[drop].first;
                    g->removed_orig.erase(g->removed_orig.begin(), g->removed_orig.begin() + (first_kept - g->removed_base));
                    g->removed_base = first_kept;
                    g->rlog.erase(g->rlog.begin(), g->rlog.begin() + drop);
                    g->rlog_medoid.erase(g->rlog_medoid.begin(), g->rlog_medoid.begin() + drop * (size_t)g->clu->L);
                    g->rlog_base = keep_from;
                }
            }
        }
        g->n_emitted++;
        g->n_remaining -= (int64_t)points.size();
        {
            GenTimer tl(&g->t_live);
            for (int64_t r : points) {
                g->kept[(size_t)r] = 0;
                g->alive[(size_t)g->indices[(size_t)r]] = 0;
                gen_bit_remove(g, r);
            }
        }
        GenTimer tpk(&g->t_pack_incl);   // (to the end of the emission: the compaction, when one is due)
        const int64_t n_rows = (int64_t)g->kept.size();
        if (g->n_remaining > 0 && n_rows >= g->pack_min_rows && (double)g->n_remaining < g->pack_fraction * (double)n_rows) {
            int64_t new_n = 0;
            gen_prepare_pack(g);
            gen_check(vh_clu_pack(g->clu, &new_n));
            if (g->comm) {   // every rank packed its own shard: the new offsets follow from the (replicated) live mask
                std::vector<int64_t> off(g->offsets.size(), 0);
                for (size_t r = 0; r + 1 < g->offsets.size(); ++r) {
                    int64_t live = 0;
                    for (int64_t i = g->offsets[r]; i < g->offsets[r + 1]; ++i) live += g->kept[(size_t)i] != 0;
                    off[r + 1] = off[r] + live;
                }
                VH_REQUIRE(off[(size_t)g->comm->rank + 1] - off[(size_t)g->comm->rank] == new_n, "pack bookkeeping mismatch (shard)");
                g->offsets = off;
                new_n = off.back();
            }
            // Physical row numbers change.  The cached statistics survive the renumbering (a row's new number is its rank among
            // the live rows: the Fenwick tree still describes the old numbering here): an entry is kept if its row and every
            // row of its within-radius list are live -- a dead row in the list means a neighbour was removed since the scan,
            // which the lazy validation would have found -- and its list is translated; lists that are not on the host yet
            // were fetched from the ring before the pack (gen_prepare_pack), the rest fall back to a select pass.
            {
                std::unordered_map<int64_t, GenStats> kept_stats;
                kept_stats.reserve(g->stats.size());
                for (auto& kv : g->stats) {
                    GenStats& st = kv.second;
                    if (!g->kept[(size_t)kv.first]) continue;
                    bool ok = true;
                    if (st.have_list) {
                        for (int64_t& r : st.within) {
                            if (!g->kept[(size_t)r]) { ok = false; break; }
                            r = gen_logical_index(g, r);
                        }
                    } else {
                        st.list_count = (unsigned int)kListCap + 1u;   // no list: the caller selects
                    }
                    if (ok) kept_stats.emplace(gen_logical_index(g, kv.first), std::move(st));
                }
                g->stats.swap(kept_stats);
            }
            g->ring_rows_valid_from = g->clu->scan_seq;
            g->pending_mark.assign(g->pending_mark.size(), 0);
            size_t w = 0;
            for (size_t r = 0; r < g->kept.size(); ++r)
                if (g->kept[r]) g->indices[w++] = g->indices[r];
            VH_REQUIRE((int64_t)w == new_n, "pack bookkeeping mismatch");
            g->indices.resize(w);
            g->rows_epoch++;   // physical rows renumbered
            g->kept.assign(w, 1);
            gen_bit_build(g);
        }
        info->pvr_after = g->pvr;
        info->successes_after = g->successes;
        info->attempts_after = (int64_t)g->attempts.size();
        info->order_index_after = g->order_index;
    }
}

}  // namespace

extern "C" {

int vh_gen_next(vh_gen* g, vh_cluster_info* info, int64_t* members, int64_t cap) {
    return guarded([&] { gen_next_impl(g, info, members, cap); });
}

// Up to max_clusters consecutive clusters in one call: infos[i] describes cluster i, whose members follow those of cluster
// i - 1 in `members`.  A members buffer of the generator's row count always holds whatever is left.  *n_out < max_clusters:
// the generator is exhausted (or the buffer was too small for the next cluster).  The state machine is exactly vh_gen_next's;
// the batch only saves the per-cluster call overhead of the binding (~3 us of a ctypes round trip, 0.2 M times per C2 sweep).
int vh_gen_next_batch(vh_gen* g, int max_clusters, vh_cluster_info* infos, int64_t* members, int64_t cap, int* n_out) {
    return guarded([&] {
        VH_REQUIRE(g != nullptr && infos != nullptr && members != nullptr && n_out != nullptr && max_clusters >= 1, "bad argument");
        *n_out = 0;
        int64_t used = 0;
        for (int i = 0; i < max_clusters; ++i) {
            if (g->n_remaining == 0 || g->n_remaining > cap - used) break;   // (a cluster never has more members than are left)
            gen_next_impl(g, &infos[i], members + used, cap - used);
            if (infos[i].n_members == 0) break;
            used += infos[i].n_members;
            ++*n_out;
        }
    });
}

// test hook (host only, no GPU): find_threshold of the native state machine on a given exact histogram
int vh_debug_find_threshold(const int64_t* hist_fx, int64_t n_lt, double pvr, int* kind, double* threshold,
                            double* observed_pvr) {
    return guarded([&] {
        VH_REQUIRE(hist_fx != nullptr && kind != nullptr && threshold != nullptr && observed_pvr != nullptr, "NULL argument");
        vh_gen g;
        g.pvr = pvr;
        GenStats st;
        st.n_lt = n_lt;
        for (int b = 0; b < VH_NBINS; ++b) st.hist_fx[b] = hist_fx[b];
        *threshold = 0.0;
        *observed_pvr = 0.0;
        *kind = (int)gen_find_threshold(&g, st, threshold, observed_pvr);
    });
}

// Diagnostic (option debug.guard_bytes > 0): the live device allocations whose trailing canary is no longer intact, one line each
// (allocation number, payload bytes, first damaged byte past the end, damaged bytes, allocation call stack as offsets into this
// library: resolve with llvm-symbolizer -e libvambhip.so).  *damaged = how many; the report is cut at cap bytes.
int vh_debug_check_guards(char* report, int cap, int* damaged) {
    return guarded([&] {
        VH_REQUIRE(report != nullptr && cap >= 1 && damaged != nullptr, "bad argument");
        VH_HIP(hipDeviceSynchronize());
        std::string out;
        *damaged = 0;
        std::lock_guard<std::mutex> lock(::vh::g_guard_mutex);
        std::vector<unsigned char> host;
        Dl_info self{};
        (void)dladdr(reinterpret_cast<void*>(&vh_debug_check_guards), &self);
        char line[1024];
        snprintf(line, sizeof(line), "%zu tracked allocations\n", ::vh::guard_table().size());
        out += line;
        for (const auto& kv : ::vh::guard_table()) {
            const ::vh::GuardRec& r = kv.second;
            host.resize(r.guard);
            VH_HIP(hipMemcpy(host.data(), static_cast<const char*>(kv.first) + r.bytes, r.guard, hipMemcpyDeviceToHost));
            size_t first = r.guard, count = 0, last = 0;
            for (size_t i = 0; i < r.guard; ++i)
                if (host[i] != ::vh::kGuardByte) { if (first == r.guard) first = i; last = i; ++count; }
            if (count == 0) continue;
            ++*damaged;
            int o = snprintf(line, sizeof(line), "allocation #%llu: %zu bytes; canary damaged at +%zu .. +%zu (%zu bytes); first words:",
                             (unsigned long long)r.id, r.bytes, first, last, count);
            for (size_t i = first; i < std::min(first + 16, r.guard) && o < (int)sizeof(line) - 8; i += 4)
                o += snprintf(line + o, sizeof(line) - o, " %02x%02x%02x%02x", host[i + 3 < r.guard ? i + 3 : i], host[i + 2 < r.guard ? i + 2 : i],
                              host[i + 1 < r.guard ? i + 1 : i], host[i]);
            o += snprintf(line + o, sizeof(line) - o, "; stack:");
            for (int f = 1; f < r.n_frames && o < (int)sizeof(line) - 24; ++f) {
                Dl_info di{};
                if (dladdr(r.frames[f], &di) && di.dli_fbase == self.dli_fbase)
                    o += snprintf(line + o, sizeof(line) - o, " 0x%llx", (unsigned long long)(static_cast<char*>(r.frames[f]) - static_cast<char*>(di.dli_fbase)));
            }
            out += line;
            out += "\n";
        }
        snprintf(report, (size_t)cap, "%s", out.c_str());
    });
}

// Diagnostic: the phase stamps of the LAST scan pass of a handle (a library built with -DVAMBHIP_TIMING_EXPERIMENTS only)
int vh_debug_scan_timeline(vh_clu* h, unsigned long long* stamps, int cap_rows, int* n_rows, double* host_us) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && stamps != nullptr && n_rows != nullptr && host_us != nullptr && cap_rows >= 1, "bad argument");
#ifdef VAMBHIP_TIMING_EXPERIMENTS
        VH_HIP(hipStreamSynchronize(h->stream));
        VH_REQUIRE(cap_rows >= kStampRows, "stamps: room for %d rows of 8 words (the last row is the publish kernel's)", kStampRows);
        VH_HIP(hipMemcpy(stamps, h->stamps.p, (size_t)kStampRows * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        *n_rows = kStampRows;
        for (int i = 0; i < 4; ++i) host_us[i] = h->host_us[i];
        VH_HIP(hipMemset(h->stamps.p, 0, h->stamps.bytes()));
#else
        (void)h; (void)stamps; (void)cap_rows; (void)host_us;
        *n_rows = 0;
        VH_REQUIRE(false, "vh_debug_scan_timeline needs a library built with -DVAMBHIP_TIMING_EXPERIMENTS");
#endif
    });
}

// test hook (host only, no GPU): consecutive random.Random(seed).sample(range(ns[i]), ks[i]) calls on ONE generator
int vh_debug_pyrandom_sample(uint64_t seed, int n_calls, const int64_t* ns, const int64_t* ks, int64_t* out) {
    return guarded([&] {
        VH_REQUIRE(ns != nullptr && ks != nullptr && out != nullptr && n_calls >= 0, "bad argument");
        PyRandom rng;
        rng.seed(seed);
        std::vector<int64_t> pop, got;
        for (int i = 0; i < n_calls; ++i) {
            VH_REQUIRE(ks[i] >= 0 && ks[i] <= ns[i] && ns[i] < (1ll << 31), "need 0 <= k <= n < 2^31");
            pop.resize((size_t)ns[i]);
            for (int64_t v = 0; v < ns[i]; ++v) pop[(size_t)v] = v;
            rng.sample(pop, (int)ks[i], got);
            for (int64_t v : got) *out++ = v;
        }
    });
}

int vh_gen_state(vh_gen* g, double* peak_valley_ratio, int64_t* successes, int64_t* attempts, int64_t* order_index) {
    return guarded([&] {
        VH_REQUIRE(g != nullptr, "NULL argument");
        if (peak_valley_ratio) *peak_valley_ratio = g->pvr;
        if (successes) *successes = g->successes;
        if (attempts) *attempts = (int64_t)g->attempts.size();
        if (order_index) *order_index = g->order_index;
    });
}

int vh_gen_live_rows(vh_gen* g, int64_t* live_rows_streamed) {
    return guarded([&] {
        VH_REQUIRE(g != nullptr && live_rows_streamed != nullptr, "NULL argument");
        *live_rows_streamed = g->live_rows_streamed;
    });
}

int vh_gen_counters(vh_gen* g, int64_t* scan_passes, int64_t* scan_medoids, int64_t* rows_streamed, double* kernel_ms,
                    int64_t* n_emitted, int64_t* n_remaining) {
    return guarded([&] {
        VH_REQUIRE(g != nullptr, "NULL argument");
        if (scan_passes) *scan_passes = g->scan_passes;
        if (scan_medoids) *scan_medoids = g->scan_medoids;
        if (rows_streamed) *rows_streamed = g->rows_streamed;
        if (kernel_ms) *kernel_ms = g->kernel_ms;
        if (n_emitted) *n_emitted = g->n_emitted;
        if (n_remaining) *n_remaining = g->n_remaining;
    });
}

}  // extern "C"
