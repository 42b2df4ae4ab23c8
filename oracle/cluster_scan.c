/* TEST INFRASTRUCTURE ONLY -- defined-order CPU restatement of the cluster scan arithmetic.
 *
 * Part of oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (vamb_amd/) never links or calls it.
 *
 * Restates, with a FIXED floating-point evaluation order that the HIP kernels reproduce bit-for-bit:
 *   - vamb/cluster.py:653-669  _normalize          -> vo_normalize
 *   - vamb/cluster.py:672-676  _calc_distances     -> vo_distances
 *   - vamb/cluster.py:606-637  sample_medoid       -> vo_scan (within list, local density)
 *   - vamb/cluster.py:452-481  find_threshold head -> vo_scan (loner count, weighted histogram)
 *   - vamb/cluster.py:640-650  _smaller_indices    -> vo_select
 *
 * Defined order (the contract shared with vamb_amd/csrc/cluster.hip):
 *   dot(i)   = fmaf chain over k = 0..L-1 starting from +0.0f          (one rounding per step)
 *   d(i)     = 0.5f - dot(i);  d(medoid) = 0
 *   density  = sum_i llrint( (double)( len_i * (0.05f - d_i) ) * 2^16 )   exact int64, order-free
 *   hist[b]  = sum_i llrint( (double) len_i * 2^8 )                      exact int64, order-free
 *   bin b    : edge[b] <= d < edge[b+1] (last bin closed), edge = float32 linspace(0, 0.3, 61) as
 *              torch.linspace produces it (== torch.histogram's searchsorted-right-minus-one rule)
 * The reference sums density / histogram in fp32 in an unspecified (MKL / vectorised) order; the
 * integer accumulation here is the correctly rounded version of the same sum.
 *
 * vo_set_order(1 | 2) -- the REFERENCE's own order, measured (round 3; oracle/probe_reference_order.py): the evaluation order of
 * `matrix.matmul(matrix[index])` and of `matrix.norm(dim=1)` on the torch 2.10 / oneMKL 2024.2 / AVX-512 CPU build of this
 * container, identified by probing which partial sums meet first and then confirmed bit for bit on 10^5-row random matrices for
 * every latent width from 1 to 257, 1 and 8 threads.  Order 2 (the DEFAULT since round 4, and the default of libvambhip's
 * scan.reference_order) keeps the exact integer density / histogram sums; order 1 additionally makes cluster_oracle.py sum them in
 * float32 in the reference's own order (vo_torch_sum / vo_reference_sums), which reproduces even the reported observed_pvr bit
 * for bit.  Order 0 is the ascending fmaf chain of the "defined order" paragraph above (libvambhip: scan.reference_order = 0).
 *   dot(i)   : s = a0 x0;  16 lanes {s, 0, ...}; every full block of 16 columns from column 1 on is accumulated lane-wise with
 *              fma; halving tree (p + 8, p + 4, p + 2, p + 1); the (L - 1) % 16 remaining columns form one more 16-lane block
 *              whose lane 0 starts from the running sum, reduced by the same tree
 *   norm(i)  : 8 lanes accumulated with fma over full blocks of 8 columns, lanes summed 0..7 in order; of the L % 8 remaining
 *              columns the first four (if there are four) add their rounded squares one by one, the last <= 3 are fused
 *              multiply-adds; sqrtf;  row / (norm * 1.41421354f)
 * With this order the defined arithmetic IS the reference's arithmetic for distances and normalisation on that build.
 * vo_torch_sum / vo_reference_sums add the two float32 sums the reference reports (density, histogram bins) in ITS order.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define VO_NBINS 60
#define VO_DENSITY_SCALE 65536.0
#define VO_HIST_SCALE 256.0

/* torch.linspace(0.0, 0.3, 61) in float32, bit patterns as ATen's (vectorised) linspace kernel
 * produces them; these are also the edges torch.histogram(range=(0,0.3), bins=60) writes
 * (cluster.py:288,475-481).  tests/test_oracle_cluster.py asserts table == torch.linspace. */
static const uint32_t g_edge_bits[VO_NBINS + 1] = {
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
static float g_edges[VO_NBINS + 1];
static int g_edges_ready = 0;

static void make_edges(void) {
    memcpy(g_edges, g_edge_bits, sizeof(g_edges));
    g_edges_ready = 1;
}

void vo_edges(float* out) {
    if (!g_edges_ready) make_edges();
    memcpy(out, g_edges, sizeof(g_edges));
}

/* bin index or -1 when d is outside [edge[0], edge[60]] (torch.histogram skips those) */
int vo_bin(float d) {
    if (!g_edges_ready) make_edges();
    if (!(d >= g_edges[0]) || !(d <= g_edges[VO_NBINS])) return -1;
    int b = (int)(d * 200.0f);
    if (b < 0) b = 0;
    if (b > VO_NBINS - 1) b = VO_NBINS - 1;
    while (b > 0 && d < g_edges[b]) --b;
    while (b < VO_NBINS - 1 && d >= g_edges[b + 1]) ++b;
    return b;
}

static int g_order = 2;   /* 0: ascending fmaf chain, 1 / 2: the reference build's order for matmul and norm (see header); 1 also selects
                            * the reference's float32 density / histogram sums in cluster_oracle.py, 2 (default = what libvambhip
                            * computes by default) keeps the exact integer sums */
void vo_set_order(int order) { g_order = order; }
int vo_get_order(void) { return g_order; }

static inline float tree16(const float* v) {
    float a[8], b[4], c[2];
    for (int p = 0; p < 8; ++p) a[p] = v[p] + v[p + 8];
    for (int p = 0; p < 4; ++p) b[p] = a[p] + a[p + 4];
    for (int p = 0; p < 2; ++p) c[p] = b[p] + b[p + 2];
    return c[0] + c[1];
}

/* <row, q> as torch CPU (oneMKL sgemv, AVX-512) evaluates it */
static inline float dot_ref(const float* row, const float* q, int L) {
    float s = row[0] * q[0];
    const int nfull = (L - 1) / 16, rem = (L - 1) % 16;
    int k = 1;
    float acc[16];
    if (nfull) {
        acc[0] = s;
        for (int p = 1; p < 16; ++p) acc[p] = 0.0f;
        for (int b = 0; b < nfull; ++b, k += 16)
            for (int p = 0; p < 16; ++p) acc[p] = fmaf(row[k + p], q[k + p], acc[p]);
        s = tree16(acc);
    }
    if (rem) {
        acc[0] = s;
        for (int p = 1; p < 16; ++p) acc[p] = 0.0f;
        for (int p = 0; p < rem; ++p) acc[p] = fmaf(row[k + p], q[k + p], acc[p]);
        s = tree16(acc);
    }
    return s;
}

/* ||row|| as torch's norm(dim=1) evaluates it (ATen norm_reduce, 8-lane vectors) */
static inline float norm_ref(const float* row, int L) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int d = 0;
    for (; d + 8 <= L; d += 8)
        for (int p = 0; p < 8; ++p) acc[p] = fmaf(row[d + p], row[d + p], acc[p]);
    float s = acc[0];
    for (int p = 1; p < 8; ++p) s = s + acc[p];
    if (L - d >= 4) {   /* four remaining columns: their ROUNDED squares are added one after the other */
        for (int p = 0; p < 4; ++p) { const float sq = row[d + p] * row[d + p]; s = s + sq; }
        d += 4;
    }
    for (; d < L; ++d) s = fmaf(row[d], row[d], s);   /* the last <= 3 columns: fused */
    return sqrtf(s);
}

/* torch.sum of a contiguous float32 vector as ATen evaluates it on the same build (SumKernel.cpp cascade_sum ->
 * vectorized_inner_sum / scalar_inner_sum; measured with the same probe and confirmed on random vectors of 1..20000 elements):
 * items are 8-lane vectors (n >= 8) or scalars (n < 8); four interleaved rows of items (item i -> row i % 4) are accumulated by
 * a cascade (sixteen items into level 0, level 0 into level 1, ...; levels added up at the end), the items left over by the
 * interleave go into row 0, rows are added 0..3; the n % 8 trailing scalars are summed first, then lanes 0..7 are added to that. */
#define VO_W 8
static int ceil_log2_i64(int64_t x) { int b = 0; while (((int64_t)1 << b) < x) ++b; return b; }
static void vo_row_sum(const float* x, int64_t size, int W, float* out /* [W] */) {
    const int ilp = 4, num_levels = 4;
    const int64_t size_ilp = size / ilp;
    int level_power = ceil_log2_i64(size_ilp) / num_levels;
    if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power, level_mask = level_step - 1;
    float acc[4][4][VO_W];
    memset(acc, 0, sizeof(acc));
    int64_t i = 0;
    while (i + level_step <= size_ilp) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < ilp; ++k)
                for (int p = 0; p < W; ++p) acc[0][k][p] = acc[0][k][p] + x[((i * ilp + k) * W) + p];
        for (int j = 1; j < num_levels; ++j) {
            for (int k = 0; k < ilp; ++k)
                for (int p = 0; p < W; ++p) { acc[j][k][p] = acc[j][k][p] + acc[j - 1][k][p]; acc[j - 1][k][p] = 0.0f; }
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size_ilp; ++i)
        for (int k = 0; k < ilp; ++k)
            for (int p = 0; p < W; ++p) acc[0][k][p] = acc[0][k][p] + x[((i * ilp + k) * W) + p];
    for (int j = 1; j < num_levels; ++j)
        for (int k = 0; k < ilp; ++k)
            for (int p = 0; p < W; ++p) acc[0][k][p] = acc[0][k][p] + acc[j][k][p];
    for (int64_t t = size_ilp * ilp; t < size; ++t)
        for (int p = 0; p < W; ++p) acc[0][0][p] = acc[0][0][p] + x[t * W + p];
    for (int k = 1; k < ilp; ++k)
        for (int p = 0; p < W; ++p) acc[0][0][p] = acc[0][0][p] + acc[0][k][p];
    for (int p = 0; p < W; ++p) out[p] = acc[0][0][p];
}
float vo_torch_sum(const float* x, int64_t n) {
    float v[VO_W];
    if (n < VO_W) {
        vo_row_sum(x, n, 1, v);
        return v[0];
    }
    const int64_t nb = n / VO_W;
    vo_row_sum(x, nb, VO_W, v);
    float s = 0.0f;
    for (int64_t k = nb * VO_W; k < n; ++k) s = s + x[k];
    for (int p = 0; p < VO_W; ++p) s = s + v[p];
    return s;
}

/* cluster.py:653-669.  Row-major [n][L], in place. */
void vo_normalize(float* m, int64_t n, int L) {
    const float inv_l = (float)(1.0 / (double)L);
    const float sqrt2 = (float)1.4142135623730951; /* tensor * (2**0.5): python double cast to f32 */
    for (int64_t i = 0; i < n; ++i) {
        float* row = m + i * (int64_t)L;
        int allzero = 1;
        for (int k = 0; k < L; ++k) if (row[k] != 0.0f) { allzero = 0; break; }
        if (allzero) for (int k = 0; k < L; ++k) row[k] = inv_l;
        float nrm;
        if (g_order != 0) {
            nrm = norm_ref(row, L);
        } else {
            float ss = 0.0f;
            for (int k = 0; k < L; ++k) ss = fmaf(row[k], row[k], ss);
            nrm = sqrtf(ss);
        }
        const float denom = nrm * sqrt2;
        for (int k = 0; k < L; ++k) row[k] = row[k] / denom;
    }
}

static inline float dist_to(const float* row, const float* q, int L) {
    if (g_order != 0) return 0.5f - dot_ref(row, q, L);
    float acc = 0.0f;
    for (int k = 0; k < L; ++k) acc = fmaf(row[k], q[k], acc);
    return 0.5f - acc;
}

/* cluster.py:672-676 */
void vo_distances(const float* m, int64_t n, int L, int64_t medoid, float* dist) {
    const float* q = m + medoid * (int64_t)L;
    for (int64_t i = 0; i < n; ++i) dist[i] = dist_to(m + i * (int64_t)L, q, L);
    dist[medoid] = 0.0f;
}

/* One sample_medoid + the head of find_threshold, restricted to rows with kept[i] != 0
 * (kept == NULL means all rows live, i.e. the reference's packed CPU path).
 * q: the query vector (NULL: row `medoid` of m); medoid: the row whose distance is forced to 0, or -1
 * when the medoid is not part of this matrix (row-sharded execution).
 * within_idx receives the ascending row indices with d <= 0.05f (at most cap written; the true
 * count is returned in *n_within).  dist (optional) receives every distance. */
void vo_scan_q(const float* m, const float* lengths, const uint8_t* kept, int64_t n, int L,
               int64_t medoid, const float* q, float* dist, int64_t* hist_fx, int64_t* density_fx,
               int64_t* n_within, int64_t* n_lt, int64_t* within_idx, int64_t cap) {
    if (!g_edges_ready) make_edges();
    if (!q) q = m + medoid * (int64_t)L;
    int64_t dens = 0, nw = 0, nlt = 0;
    for (int b = 0; b < VO_NBINS; ++b) hist_fx[b] = 0;
    for (int64_t i = 0; i < n; ++i) {
        float d = dist_to(m + i * (int64_t)L, q, L);
        if (i == medoid) d = 0.0f;
        if (dist) dist[i] = d;
        if (kept && !kept[i]) continue;
        if (d < 0.05f) ++nlt;
        if (d <= 0.05f) {
            const float p = lengths[i] * (0.05f - d);
            dens += llrint((double)p * VO_DENSITY_SCALE);
            if (within_idx && nw < cap) within_idx[nw] = i;
            ++nw;
        }
        const int b = vo_bin(d);
        if (b >= 0) hist_fx[b] += llrint((double)lengths[i] * VO_HIST_SCALE);
    }
    *density_fx = dens;
    *n_within = nw;
    *n_lt = nlt;
}

/* What the reference itself reports for one scan when its distances are `dist` (live rows only): local_density =
 * (lengths[within] * (0.05f - dist[within])).sum() (cluster.py:628-629, torch.sum's order above) and the histogram as
 * torch.histogram with ONE thread accumulates it (cluster.py:475-481: float32, rows in ascending order).  scratch: n floats. */
void vo_reference_sums(const float* dist, const float* lengths, const uint8_t* kept, int64_t n, float* scratch,
                       float* density, float* hist /* [VO_NBINS] */) {
    if (!g_edges_ready) make_edges();
    int64_t m = 0;
    for (int b = 0; b < VO_NBINS; ++b) hist[b] = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        if (kept && !kept[i]) continue;
        const float d = dist[i];
        if (d <= 0.05f) scratch[m++] = lengths[i] * (0.05f - d);
        const int b = vo_bin(d);
        if (b >= 0) hist[b] = hist[b] + lengths[i];
    }
    *density = vo_torch_sum(scratch, m);
}

void vo_scan(const float* m, const float* lengths, const uint8_t* kept, int64_t n, int L,
             int64_t medoid, float* dist, int64_t* hist_fx, int64_t* density_fx,
             int64_t* n_within, int64_t* n_lt, int64_t* within_idx, int64_t cap) {
    vo_scan_q(m, lengths, kept, n, L, medoid, 0, dist, hist_fx, density_fx, n_within, n_lt, within_idx, cap);
}

/* cluster.py:640-650: ascending indices of live rows with d <= threshold (float32 compare) */
int64_t vo_select_q(const float* m, const uint8_t* kept, int64_t n, int L, int64_t medoid, const float* q,
                    float threshold, int64_t* out_idx, int64_t cap) {
    if (!q) q = m + medoid * (int64_t)L;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (kept && !kept[i]) continue;
        float d = dist_to(m + i * (int64_t)L, q, L);
        if (i == medoid) d = 0.0f;
        if (d <= threshold) {
            if (cnt < cap) out_idx[cnt] = i;
            ++cnt;
        }
    }
    return cnt;
}

int64_t vo_select(const float* m, const uint8_t* kept, int64_t n, int L, int64_t medoid,
                  float threshold, int64_t* out_idx, int64_t cap) {
    return vo_select_q(m, kept, n, L, medoid, 0, threshold, out_idx, cap);
}

/* vambcore.overwrite_matrix contract (vambtools.py:291-321): order-preserving row compaction */
int64_t vo_compact_rows(float* m, const uint8_t* mask, int64_t n, int L) {
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!mask[i]) continue;
        if (w != i) memmove(m + w * (int64_t)L, m + i * (int64_t)L, sizeof(float) * (size_t)L);
        ++w;
    }
    return w;
}
