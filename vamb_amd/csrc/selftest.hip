// selftest.hip -- start-up self-test of the kernels whose correctness rests on hand-counted `s_waitcnt vmcnt(n)` behind inline-asm
// loads: the row-major many-medoid scan K6r (cluster.hip, clu_scan_mfma_rm_kernel) and the deep-prefetch /
// K-group tiles of the fp32 GEMM (gemm.hpp, PF = 4 / KS = 4).  The compiler treats the destination registers of such a load as
// defined at once; a register copy or spill it placed between the load and the manual wait -- legal for it, and a matter of
// the hipcc version and flags -- would read stale data, silently.  The GPU tests guard the build that was tested; this guards
// the build that RUNS: each of those kernels is run once beside its compiler-scheduled twin on the same input, and a
// disagreement switches the option that selects it off for the process (and says so on stderr):
//   scan  : 32 medoids x 20 000 rows x 32 columns, scan.mfma_rowmajor = 1 against 0     -> every int64 accumulator equal
//   GEMM  : 256 x 512 x 512, tile 6 (four K-tiles in flight) against tile 3             -> bit-identical
//           tile 7 (four K groups per workgroup, another summation order) against tile 3 -> 1e-5 of the largest entry
// Written on top of the library's own C ABI (vambhip.h / vambhip_debug.h); ~40 ms, once per process (vamb_amd/_lib.py calls it
// before the first handle is created; a C caller calls vh_selftest itself).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/vambhip.h"
#include "../../include/vambhip_debug.h"

namespace {

// xorshift64*: deterministic inputs without <random>'s implementation-defined distributions
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint64_t next() {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    float uniform() { return (float)((next() >> 40) * (1.0 / 16777216.0)); }   // [0, 1)
    float normal() {                                                           // sum of 12 uniforms - 6
        float t = 0.f;
        for (int i = 0; i < 12; ++i) t += uniform();
        return t - 6.0f;
    }
};

int scan_once(const std::vector<float>& mat, const std::vector<float>& len, int64_t n, int L, int rowmajor,
              const std::vector<int64_t>& medoids, std::vector<vh_scan_result>& out) {
    int st = vh_set_option("scan.mfma_rowmajor", rowmajor);
    if (st != VH_OK) return st;
    vh_clu* h = nullptr;
    st = vh_clu_create(mat.data(), len.data(), n, L, 0, nullptr, &h);
    if (st != VH_OK) return st;
    out.resize(medoids.size());
    st = vh_clu_scan(h, (int)medoids.size(), medoids.data(), nullptr, out.data());
    vh_clu_destroy(h);
    return st;
}

}  // namespace

extern "C" int vh_selftest(int* fallbacks) {
    int mask = 0;
    // ---- scan: blobs (20 rows per centre, sigma 0.05: every medoid has rows inside its radius and in the histogram range)
    {
        const int64_t n = 20000;
        const int L = 32;
        Rng rng(7);
        std::vector<float> centres((size_t)(n / 20) * L), mat((size_t)n * L), len((size_t)n);
        for (auto& c : centres) c = rng.normal();
        for (int64_t i = 0; i < n; ++i) {
            const float* c = &centres[(size_t)(i % (n / 20)) * L];
            for (int j = 0; j < L; ++j) mat[(size_t)i * L + j] = c[j] + 0.05f * rng.normal();
            len[(size_t)i] = 2000.0f + 1000.0f * rng.uniform();
        }
        std::vector<int64_t> medoids(32);
        for (int j = 0; j < 32; ++j) medoids[(size_t)j] = (int64_t)(rng.next() % (uint64_t)n);
        int64_t user = 1;
        vh_get_option("scan.mfma_rowmajor", &user);
        std::vector<vh_scan_result> a, b;
        int st = scan_once(mat, len, n, L, 1, medoids, a);
        if (st == VH_OK) st = scan_once(mat, len, n, L, 0, medoids, b);
        if (st != VH_OK) return st;
        const bool same = memcmp(a.data(), b.data(), a.size() * sizeof(vh_scan_result)) == 0;
        int64_t hits = 0;
        for (const auto& r : b) hits += r.n_within;
        if (!same || hits < 32) {
            mask |= 1;
            vh_set_option("scan.mfma_rowmajor", 0);
            fprintf(stderr, "[vambhip] SELF-TEST: the row-major many-medoid scan kernel disagrees with the column-major one on this "
                            "build (%s); falling back to scan.mfma_rowmajor = 0 for this process\n",
                    same ? "no row inside a medoid radius: input generator broken" : "accumulators differ");
        } else if (user != 1) {
            vh_set_option("scan.mfma_rowmajor", user);
        } else {
            vh_unset_option("scan.mfma_rowmajor");
        }
    }
    // ---- fp32 GEMM tiles
    {
        const int M = 256, N = 512, K = 512;
        Rng rng(11);
        std::vector<float> A((size_t)M * K), B((size_t)N * K), c3((size_t)M * N), c6(c3.size()), c7(c3.size());
        for (auto& v : A) v = rng.normal();
        for (auto& v : B) v = rng.normal() * 0.05f;
        float ms = 0.f;
        int st = vh_debug_gemm(3, 1, 1, A.data(), B.data(), nullptr, c3.data(), M, N, K, 1, &ms);
        if (st == VH_OK) st = vh_debug_gemm(6, 1, 1, A.data(), B.data(), nullptr, c6.data(), M, N, K, 1, &ms);
        if (st == VH_OK) st = vh_debug_gemm(7, 1, 1, A.data(), B.data(), nullptr, c7.data(), M, N, K, 1, &ms);
        if (st != VH_OK) return st;
        float big = 0.f, err7 = 0.f;
        for (size_t i = 0; i < c3.size(); ++i) {
            big = std::fmax(big, std::fabs(c3[i]));
            err7 = std::fmax(err7, std::fabs(c7[i] - c3[i]));
        }
        const bool ok6 = memcmp(c3.data(), c6.data(), c3.size() * sizeof(float)) == 0;
        const bool ok7 = big > 0.f && err7 <= 1e-5f * big;
        if (!ok6 || !ok7) {
            mask |= 2;
            vh_set_option("vae.gemm_prefetch", 1);
            vh_set_option("vae.gemm_kgroups", 1);
            fprintf(stderr, "[vambhip] SELF-TEST: the deep-prefetch / K-group tiles of the fp32 GEMM disagree with the plain tile on this "
                            "build (prefetch %s, K groups %s); falling back to vae.gemm_prefetch = 1, vae.gemm_kgroups = 1\n",
                    ok6 ? "ok" : "DIFFERENT", ok7 ? "ok" : "DIFFERENT");
        }
    }
    if (fallbacks) *fallbacks = mask;
    return VH_OK;
}
