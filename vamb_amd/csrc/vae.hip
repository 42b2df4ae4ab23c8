// vae.hip -- MI355X (gfx950) implementation of vamb.encode.VAE's numerics behind the C ABI:
// forward (encode.py:259-314), loss (316-357), backward (autograd of the same), BatchNorm1d,
// dropout, D-Adapt-Adam (dadaptation 3.2, encode.py:578) and the eval-mode encode pass (442-484).
//
// HBM layout (per handle)
//   X      [n][D_p]    the normalised feature matrix depths|tnf|abundance, rows padded to D_p = ru32(D)
//   w      [n]         contig weights
//   P/M1/M2/S  flat    parameters and D-Adapt-Adam moments; every tensor stored padded
//                      ([rows_p][cols_p], multiples of 32, zero padding) in a 1024-element aligned slot
//   per-batch workspaces sized for the current batch (bs_p = ru128(bs))
// All dense contractions run on the fp32-input MFMA kernel of gemm.hpp; everything else is a small
// bandwidth-bound kernel from vae_kernels.hpp.  One stream, no host synchronisation inside an epoch.
#include "comm.hpp"
#include "common.hpp"
#include "dataset.hpp"

#include <hip/hip_ext.h>
#include "gemm.hpp"
#include "gemm_bf16.hpp"
#include "gemm_bf16_tn.hpp"
#include "gemm_skinny16.hpp"
#include "vae_kernels.hpp"
#include "vae_kernels16.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <functional>
#include <map>
#include <memory>
#include <random>

using namespace vh;

namespace {

constexpr int kColPad = kDatasetColPad;
constexpr int kRowPad = 128;
constexpr int kProbeRing = 512;
constexpr int kSkinnySplits = 8;   // split-K slabs for the GEMMs whose output is only nlatent wide

struct Tensor {
    std::string name;
    int rows = 1, cols = 1;      // logical shape ([cols] vectors have rows == 1)
    int rows_p = 1, cols_p = 1;  // padded shape
    size_t off = 0;              // offset in the flat buffers (optimised tensors) or in bnbuf
    size_t slot = 0;             // allocated elements (multiple of 1024)
    bool optimised = true;
    bool matrix = false;         // weight matrix [rows][cols] (vectors are [1][cols])
    // gradient slabs for the current batch size
    float* slab = nullptr;
    int nslab = 0;
    int64_t stride = 0;
    const double* dsrc = nullptr;   // gradient lives in an fp64 accumulator instead of slabs
    bool dsrc_allrank = false;      // ... that SyncBN turns into an all-rank sum (BatchNorm gamma / beta)
    int64_t logical() const { return (int64_t)rows * cols; }
    int64_t padded() const { return (int64_t)rows_p * cols_p; }
};

struct Hidden {
    int nin = 0, nout = 0, nin_p = 0, nout_p = 0;
    int tW = -1, tb = -1, tG = -1, tB = -1, tRM = -1, tRV = -1;
    DevBuf<float> H, A, DA, DZ;         // post-dropout activations, eval-mode BN outputs, grad wrt the BN output,
                                        // grad wrt the pre-activation
    double* fstat = nullptr;            // [2][nout_p] batch sums of h, h^2        (inside vh_vae::statbuf)
    double* bstat = nullptr;            // [2][nout_p] batch sums of dA, dA*xhat
    double* dbias = nullptr;            // [nout_p]    column sums of dZ
    DevBuf<float> mean, invstd, scale, shift;
    DevBuf<uint8_t> mask;               // injected dropout keep-mask (parity mode)
    long long batches_tracked = 0;
    // bf16-storage step (vae_step16.hpp): activations / gradients row-major and transposed, BatchNorm-folded weights
    DevBuf<bf16_t> H16, DA16, DZ16, Wf16;
    DevBuf<float> biasf;
};

constexpr size_t kMaxDynLds = 120 * 1024;

// When set, the next launch_gemm records the kernel's own begin / end timestamps into this event pair
// (hipExtLaunchKernelGGL: the dispatch packet's timestamps, the same source rocprofv3 reads) and clears it.
thread_local hipEvent_t t_probe_start = nullptr, t_probe_stop = nullptr;
// When set (and no probe is armed), the next launch_gemm signals this event from the kernel's own completion
// signal -- a cross-stream fork without an extra barrier packet in the producing stream (fork_after below).
thread_local hipEvent_t t_fork_stop = nullptr;

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int EPI, int XFA = XF_NONE, int XFB = XF_NONE,
          int BK = 32, int DT = 0, int PF = 1, int KS = 1>
void launch_gemm(hipStream_t stream, const GemmArgs& g_in, int splits) {
    static bool attr_set = false;
    GemmArgs g = g_in;
    // K groups: one copy of the LDS layout per group (operand tiles + the coefficient tables of the group's K slab)
    const size_t group_bytes = round_up(gemm_smem_bytes<BM, BN, AKC, BKC, EPI, XFA, XFB, BK, DT>(g.k_per_split / KS), 16);
    g.group_floats = (int)(group_bytes / sizeof(float));
    const size_t smem = group_bytes * KS;
    VH_REQUIRE(smem <= kMaxDynLds, "layer too wide for the fused GEMM (needs %zu bytes of LDS)", smem);
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, AKC, BKC, EPI, XFA, XFB, BK, DT, PF, KS>;
    if (!attr_set) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kMaxDynLds));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)splits);
    if (t_probe_start) {
        hipExtLaunchKernelGGL(kern, grid, dim3(WM * WN * KS * 64), smem, stream, t_probe_start, t_probe_stop, 0, g);
        t_probe_start = t_probe_stop = nullptr;
    } else if (t_fork_stop) {
        hipExtLaunchKernelGGL(kern, grid, dim3(WM * WN * KS * 64), smem, stream, nullptr, t_fork_stop, 0, g);
        t_fork_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, grid, dim3(WM * WN * KS * 64), smem, stream, g);
    }
    VH_HIP(hipGetLastError());
}

// Small launches (at most one workgroup per CU, e.g. every GEMM of the joint TaxVamb step at batch 256) with a K loop that is a
// multiple of four tiles deep run the deep-prefetch instantiation of the fp32 kernel (gemm.hpp, PF = 4; option vae.gemm_prefetch).
int g_gemm_prefetch = 4;
int g_gemm_kgroups = 4;
// ... and those that would be at most 128 workgroups of 64 x 64 take 32 x 32 tiles with four wavefront groups over the K range
// (gemm.hpp, KS = 4; option vae.gemm_kgroups), each group with its four K-tiles in flight
// the PF / KS tiles fetch row-contiguous operands as clamped 16-byte quads (gemm.hpp asm_load_x4): a row count that is not a
// multiple of 4 would shift the last quad while the keep mask still takes it, and the loads need 16-byte aligned rows
bool quad_operands(const GemmArgs& g) {
    return g.M >= 4 && g.N >= 4 && g.M % 4 == 0 && g.N % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0;
}
bool k_groups(const GemmArgs& g, int splits) {
    if (g_gemm_kgroups != 4 || g_gemm_prefetch != 4 || g.bf16) return false;
    const int64_t wgs64 = ceil_div(g.M, 64) * ceil_div(g.N, 64) * splits;
    const int kp = g.k_per_split;
    return wgs64 <= 128 && kp % 512 == 0 && (int64_t)kp * splits == g.K && quad_operands(g);
}
bool deep_prefetch(const GemmArgs& g, int bm, int bn, int splits) {
    if (g_gemm_prefetch != 4 || g.bf16) return false;
    const int64_t wgs = ceil_div(g.M, bm) * ceil_div(g.N, bn) * splits;
    const int kp = g.k_per_split;
    return wgs <= 256 && kp % 128 == 0 && kp >= 256 && (int64_t)kp * splits == g.K && quad_operands(g);
}

// (64-wide K-tiles for the forward GEMMs of the step -- tile 4 of vh_debug_gemm -- measured +4-7 % for an isolated GEMM
// and nothing for the step, 351.5 vs 351.0 us, and slower on the backward GEMMs: not wired into the step.)
// production tiles: 3 = 64x64 (2x2 waves, 2 workgroups per CU), 2 = 128x32 (4x1) for latent-wide outputs
template <bool AKC, bool BKC, int EPI, int XFA = XF_NONE, int XFB = XF_NONE>
void gemm_tile(hipStream_t s, int tile, const GemmArgs& g, int splits) {
    if (g.bf16) {   // bf16 operands, fp32 accumulation (BASELINE configs C2+)
        if (tile == 2) launch_gemm<128, 32, 4, 1, AKC, BKC, EPI, XFA, XFB, 32, 1>(s, g, splits);
        else launch_gemm<64, 64, 2, 2, AKC, BKC, EPI, XFA, XFB, 32, 1>(s, g, splits);
        return;
    }
    if (tile == 2) launch_gemm<128, 32, 4, 1, AKC, BKC, EPI, XFA, XFB>(s, g, splits);
    else if (tile == 1) launch_gemm<128, 128, 2, 2, AKC, BKC, EPI, XFA, XFB>(s, g, splits);
    else if (tile == 0) launch_gemm<64, 128, 2, 2, AKC, BKC, EPI, XFA, XFB>(s, g, splits);
    else if (k_groups(g, splits)) launch_gemm<32, 32, 1, 1, AKC, BKC, EPI, XFA, XFB, 32, 0, 4, 4>(s, g, splits);
    else if (deep_prefetch(g, 64, 64, splits)) launch_gemm<64, 64, 2, 2, AKC, BKC, EPI, XFA, XFB, 32, 0, 4>(s, g, splits);
    else launch_gemm<64, 64, 2, 2, AKC, BKC, EPI, XFA, XFB>(s, g, splits);
}

// every tile shape, no transforms (vh_debug_gemm): 0 = 64x128, 1 = 128x128, 2 = 128x32, 3 = 64x64,
// 5 = one free-running wave per 32x32 tile
template <bool AKC, bool BKC, int EPI>
void gemm_tile_debug(hipStream_t s, int tile, const GemmArgs& g, int splits) {
    if (g.bf16) {
        switch (tile) {
            case 0: launch_gemm<64, 128, 2, 2, AKC, BKC, EPI, XF_NONE, XF_NONE, 32, 1>(s, g, splits); break;
            case 1: launch_gemm<128, 128, 2, 2, AKC, BKC, EPI, XF_NONE, XF_NONE, 32, 1>(s, g, splits); break;
            case 2: launch_gemm<128, 32, 4, 1, AKC, BKC, EPI, XF_NONE, XF_NONE, 32, 1>(s, g, splits); break;
            case 5: launch_gemm<32, 32, 1, 1, AKC, BKC, EPI, XF_NONE, XF_NONE, 32, 1>(s, g, splits); break;
            default: launch_gemm<64, 64, 2, 2, AKC, BKC, EPI, XF_NONE, XF_NONE, 32, 1>(s, g, splits); break;
        }
        return;
    }
    switch (tile) {
        case 0: launch_gemm<64, 128, 2, 2, AKC, BKC, EPI>(s, g, splits); break;
        case 1: launch_gemm<128, 128, 2, 2, AKC, BKC, EPI>(s, g, splits); break;
        case 2: launch_gemm<128, 32, 4, 1, AKC, BKC, EPI>(s, g, splits); break;
        case 4: launch_gemm<64, 64, 2, 2, AKC, BKC, EPI, XF_NONE, XF_NONE, 64>(s, g, splits); break;
        case 6: launch_gemm<64, 64, 2, 2, AKC, BKC, EPI, XF_NONE, XF_NONE, 32, 0, 4>(s, g, splits); break;   // deep prefetch
        case 7: launch_gemm<32, 32, 1, 1, AKC, BKC, EPI, XF_NONE, XF_NONE, 32, 0, 4, 4>(s, g, splits); break;   // + four K groups
        case 5: launch_gemm<32, 32, 1, 1, AKC, BKC, EPI>(s, g, splits); break;
        default: launch_gemm<64, 64, 2, 2, AKC, BKC, EPI>(s, g, splits); break;
    }
}

// Output [M][N] of a forward / input-gradient GEMM.  the option vae.big_tiles picks the largest tile that still yields
// two workgroups per CU (a 64x64 accumulator per wavefront needs one fresh LDS operand per MFMA instead of two:
// isolated, 16384x512x512 runs at 99 instead of 86 TF/s and 8192x512x1120 at 95 instead of 87).  Inside the real
// training step at C2 (batch 8192) it measured SLOWER (665 vs 634 us): opt-in, parity-tested
// (tests/test_vae_gpu.py::test_large_batch_tiles_match_oracle).
// Process-wide tuning read from the option table (vh_set_option) whenever a VAE handle is created
struct VaeTuning {
    bool big_tiles = false;   // vae.big_tiles
    int xcd_remap = 1;        // vae.xcd_remap
    int dw_workgroups = 256;  // vae.dw_workgroups: workgroups wanted per weight-gradient GEMM (split-K target)
    bool fused_skinny = true; // vae.fused_skinny: bf16 step: the two latent-wide products (mu, the first decoder layer's input gradient) and
                              // their elementwise consumers (reparameterisation, latent backward) as ONE launch each
                              // (gemm_skinny16.hpp) instead of split-K launch + slab-summing kernel.  Same bits.
    bool fused_finalize = true;  // vae.fused_finalize: bf16 step: d / k / counters / clearing of the fp64 accumulators by the LAST workgroup of the
                              // update kernel (arrivals behind drained write-through stores, no fence) instead of a one-workgroup launch.
                              // Bit-identical.  Round 5 measured it SLOWER with ONE arrival word (272.5 vs 267.4 us per step at C2: ~900
                              // arrivals on a word that takes ~88 atomics per us); with two levels of words (32 groups, round 6) it is
                              // 237.1 vs 238.6 us at C2 and neutral at the C3 shape (profiles/r06f_*): on.  The data-parallel schedule
                              // keeps the separate launch.
    int fork_plan = -1;       // vae.fork_plan: the two-stream schedule of the bf16 step, bit mask; -1 = by input width (fork_plan_for,
                              // vae_step16.hpp: 2 up to 512 padded input columns, 6 above).  The side stream always starts at the loss
                              // kernel and forks again at the top decoder layer's and encoder layer 1's BatchNorm backward.
                              // 1 = one more fork at the first decoder layer (its weight gradient and the mu layer's start there instead
                              // of at encoder layer 1); 2 = encoder layer 1's weight gradient on the MAIN stream with layer 0's (one paired
                              // launch, vae.dw_pair); 4 = running statistics + loss reduction at the END of the side stream's work instead
                              // of in front of its first weight gradient; 8 = the mu layer's weight gradient on the main stream as well;
                              // 16 = no fork at the top decoder layer
    bool prefetch_batch = true; // vae.prefetch_batch: bf16 step: the gather of step t + 1 (10 MB read, 15 MB written at C2: 8.6 us on the
                              // critical path) runs on the side stream during step t, FIRST in the batch of work the loss-kernel fork hands
                              // over; the join in front of the optimiser -- already there -- covers it, so the main stream pays no extra
                              // event (round 3 tried this with an event of its own and measured it neutral).  Same bits.
    int prefetch_max_cols = 512;   // vae.prefetch_max_cols: widest (padded) input the next-batch prefetch is used for
    bool loss_from_dataset = true; // vae.loss_from_dataset: bf16 step of the plain VAE: the loss kernel reads its targets from the dataset
                              // rows of the batch; the gather kernel then writes no fp32 copy of the batch (a third of its traffic)
    bool dw_pair = true;      // vae.dw_pair: bf16 step: the last two weight gradients of the backward pass (encoder layers 0 and 1, both on
                              // the main stream behind the first layer's BatchNorm backward) as ONE launch (gemm_bf16_tn_pair_kernel).  Same bits.
    bool loss_registers = true; // vae.loss_registers: bf16 step of the plain VAE: the loss kernel holds a wavefront's two rows in registers
                              // (vae_loss16_reg_kernel) instead of staging them in LDS; same bits but for the last digit of the reported sse
} g_tuning;

void refresh_tuning() {
    g_tuning.loss_registers = option("vae.loss_registers", 1) != 0;
    g_tuning.big_tiles = option("vae.big_tiles", 0) != 0;
    g_tuning.xcd_remap = (int)option("vae.xcd_remap", 1);
    g_tuning.dw_workgroups = (int)option("vae.dw_workgroups", 256);
    g_tuning.dw_pair = option("vae.dw_pair", 1) != 0;
    g_tuning.fused_skinny = option("vae.fused_skinny", 1) != 0;
    g_tuning.fused_finalize = option("vae.fused_finalize", 1) != 0;
    g_tuning.loss_from_dataset = option("vae.loss_from_dataset", 1) != 0;
    g_tuning.prefetch_max_cols = (int)option("vae.prefetch_max_cols", 512);
    g_gemm_prefetch = (int)option("vae.gemm_prefetch", 4);
    g_gemm_kgroups = (int)option("vae.gemm_kgroups", 4);
    g_tuning.prefetch_batch = option("vae.prefetch_batch", 1) != 0;
    g_tuning.fork_plan = (int)option("vae.fork_plan", -1);
}

int fwd_tile(int M, int N) {
    if (N <= 32) return 2;
    if (g_tuning.big_tiles) {
        if (ceil_div(M, 128) * ceil_div(N, 128) >= 512) return 1;
        if (ceil_div(M, 64) * ceil_div(N, 128) >= 512) return 0;
    }
    return 3;
}

GemmArgs base_args(bool bf16 = false) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.bf16 = bf16 ? 1 : 0;
    g.drop_scale = 1.0f;
    g.xcd_remap = g_tuning.xcd_remap;
    return g;
}

}  // namespace

struct vh_vae {
    vh_vae_config cfg;
    bool fork_ext = true;   // forks ride on the producing kernel's completion signal (option vae.fork_events: records)
    bool bf16 = false;   // GEMM operands rounded to bf16, fp32 accumulation (vh_vae_set_precision)
    int nl = 0;       // hidden layers per side
    int S = 0, D = 0, D_p = 0, L = 0, L_p = 0;
    float ce_w = 0, ab_w = 0, sse_w = 0, kld_w = 0;
    // input / reconstruction columns: S depths | ntnf TNF | nab total abundance | NL label logits (semisupervised_encode.py:
    // VAEConcat 438-698 appends the one-hot labels, VAELabels 189-436 has ONLY the label block)
    int kind = VH_VAE_PLAIN;
    int NL = 0, lab0 = 0, ntnf = VH_NTNF, nab = 1;
    float adam_lr = 0.0f;             // > 0: torch.optim.Adam(lr) instead of D-Adapt-Adam (VAELabels.trainmodel, :405)
    int64_t ld_src = 0;               // row width of the dataset in use (0: labels-only dataset)
    const int32_t* labels = nullptr;  // labels of the dataset in use
    DevBuf<int32_t> Lb;               // labels of the batch rows
    DevBuf<float> lab_part;           // per loss workgroup: label cross-entropy sum, correct predictions
    std::vector<double> label_stats;  // (mean label cross-entropy, correct predictions) per epoch of the last training call
    // hierarchical label loss (vh_vae_set_hierarchy; taxvamb_encode.py:277-538): labels are nodes of a taxonomy, the loss sees the
    // first n_leaves logits of the label block
    int n_nodes = 0, n_leaves = 0;
    DevBuf<uint8_t> leaf_masks;       // [n_nodes][n_leaves]
    // joint trainer (vaevae.hpp): a PASS REPLICA shares the parameters, moments, running statistics, streams and counters of the
    // network it belongs to and owns only its activations, gradient slabs and random-stream seed
    bool owns_streams = true;
    const unsigned long long* step_src = nullptr;   // the network's step counter (random streams)
    const long long* batch_src = nullptr;           // the network's batch cursor (row gather)
    hipStream_t stream = nullptr;
    // weight-gradient GEMMs are off the critical path of backward (only the optimiser needs them): they
    // run on a second stream, forked after each layer's dZ is ready and joined before the update
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // joint trainer (vaevae.hpp), whose passes run on their own stream pairs: recorded on `stream` when forward() has the latent
    // code (mu, z) complete, so that the passes fed by it need not wait for the decoder; and the running-statistics update of a
    // pass left to the trainer, which orders the passes of one network as the reference's step does

    std::vector<Tensor> tensors;
    std::map<std::string, int> tindex;
    std::vector<Hidden> hidden;   // encoder layers then decoder layers
    int tWmu = -1, tbmu = -1, tWo = -1, tbo = -1;
    size_t flat_elems = 0, bn_elems = 0;
    DevBuf<float> P, M1, M2, Sv, bnbuf;

    // dataset
    int64_t n = 0;
    vh_dataset own;                   // storage of vh_vae_set_dataset
    struct { float* p = nullptr; } X, w;   // the dataset in use (own or shared)
    DevBuf<int64_t> perm;
    // pinned staging: hipMemcpyAsync from pageable memory costs milliseconds (page pinning) per call
    PinnedBuf<int64_t> h_perm;
    PinnedBuf<float> h_gwsum;
    PinnedBuf<StepState> h_state;

    // per-batch workspaces
    int bs = 0, bs_p = 0;
    DevBuf<float> Xb, Wb, MU, Z, EPS, R, dR, dMUk, DA, dMU, loss_part, slabs, out_sm, skinny;
    DevBuf<double> statbuf;          // every hidden layer's fstat | bstat | dbias, zeroed once per step
    OptTable opt_tab, opt_tab_flat;  // parameter tensors -> gradient sources (slabs / fp64 accumulators / flat G)
    bool stat_clean = false;         // statbuf is all zero (left so by the optimiser's finalize kernel)
    bool keep_grads = false;         // single-step API: leave the accumulators for vh_vae_get_grad
    DevBuf<double> opt_part;
    DevBuf<StepState> state;
    int opt_blocks = 0;
    int loss_blocks = 0;

    // (hipGraph replay of the step measured 16 % slower than eager launches -- 36.3 vs 31.3 ms per C1 epoch,
    // profiles/README.md -- and was removed; everything that changes from step to step lives in device memory.)
    bool warmed_up = false;   // one eager step has run (kernel attributes set, workspaces touched)

    // bf16-storage step (configs C2-C4; vae_step16.hpp)
    DevBuf<bf16_t> W16, W16T, zeros16;       // bf16 shadows of the flat parameter buffer (plain / transposed matrices)
    DevBuf<bf16_t> Xb16, Z16, dR16, dMU16, Wf16_mu, Wf16_out;
    // next-batch prefetch (vae.prefetch_batch): the second set of batch buffers, what the running step was asked to fill, and whether
    // the set holds the batch the next step needs
    DevBuf<float> Xb_n, Wb_n;
    DevBuf<bf16_t> Xb16_n;
    DevBuf<int32_t> Lb_n;
    DevBuf<long long> Rb, Rb_n;      // dataset rows of the batch (vae.loss_from_dataset: the loss kernel's targets)
    bool batch_from_gather = false;  // the batch in Xb16 came from the gather kernel (Rb is valid), not from an uploaded batch
    bool prefetch_next = false, batch_prefetched = false;
    DevBuf<float> biasf_mu, biasf_out;
    double* dbias_mu = nullptr;              // fp64 column sums of dMU / dR (inside statbuf)
    double* dbias_out = nullptr;
    DevBuf<Opt16Tensor> opt16_tab, opt16_tab_flat;
    DevBuf<uint8_t> opt16_blk2t;             // optimiser workgroup -> entry of the table
    DevBuf<unsigned int> opt_ticket;         // arrival counter of the update kernel's workgroups (scalar tail by the last one)
    int opt16_n = 0, opt16_blocks = 0;

    // data parallelism (one process per GPU; gradients all-reduced over RCCL on `stream`)
    vh_comm* comm = nullptr;
    bool syncbn = true;              // BatchNorm batch statistics over the ALL-RANK batch (reference semantics, encode.py:238,246)
    int opt16_bucketA_blk0 = 0;      // bf16 step: first optimiser workgroup / flat offset of the decoder-side tensors
    size_t opt16_bucketA_off = 0;    // (gradient bucket that is all-reduced while the encoder's backward still runs)
    DevBuf<float> G, gwsum;          // flat gradient buffer; per-batch global weight sums
    const float* gwsum_src = nullptr; // per-batch all-rank weight sums of the running epoch (or nullptr)
    ShuffleSpec shuffle{0, 0, 1};    // device-side epoch shuffle (key 0: explicit row list / identity)
    uint64_t epoch_counter = 0;
    int global_bs = 0;               // rows of the all-rank batch (== bs without a communicator)

    // probe
    bool probe_on = false;
    int probe_layer = 0;
    std::vector<hipEvent_t> ev_a, ev_b;
    int probe_used = 0, probe_head = 0;   // armed pairs form the circular range [head, head + used)
    int64_t probe_calls = 0;              // launches of the probed GEMM seen; every probe_every-th one is timed
    int probe_every = 16;                 // option vae.probe_every: a timed launch carries a start AND a stop event (two barrier
                                          // packets around the kernel); timing every step taxes the step it measures
    PinnedBuf<double> h_epoch_loss;        // per-epoch loss sums of vh_vae_train_epochs
    double probe_ms = 0.0;
    int64_t probe_launches = 0;
    double probe_flops = 0.0;

    ~vh_vae() {
        for (auto e : ev_a) (void)hipEventDestroy(e);
        for (auto e : ev_b) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (owns_streams) {
            if (side && side != stream) (void)hipStreamDestroy(side);
            if (stream) (void)hipStreamDestroy(stream);
        }
    }

    int add_tensor(const std::string& name, int rows, int cols, bool optimised, bool matrix = false) {
        Tensor t;
        t.name = name;
        t.rows = rows;
        t.cols = cols;
        t.rows_p = matrix ? (int)round_up(rows, kColPad) : 1;   // a 1-row weight matrix is still padded to 32 rows
        t.matrix = matrix;
        t.cols_p = (int)round_up(cols, kColPad);
        t.optimised = optimised;
        t.slot = (size_t)round_up(t.padded(), 1024);
        if (optimised) { t.off = flat_elems; flat_elems += t.slot; }
        else { t.off = bn_elems; bn_elems += t.slot; }
        tensors.push_back(t);
        tindex[name] = (int)tensors.size() - 1;
        return (int)tensors.size() - 1;
    }
    float* pptr(int t) { return (tensors[t].optimised ? P.p : bnbuf.p) + tensors[t].off; }
};

namespace {

namespace step16 {   // bf16-storage step, defined in vae_step16.hpp (included below, after the shared helpers)
int dw_splits16(int M, int N, int K);
void prepare_batch16(vh_vae* h);
void build_opt16_table(vh_vae* h);
void refresh_shadows(vh_vae* h, int only);
struct SideQueue;
void forward16(vh_vae* h, bool training, bool eps_injected, bool masks_injected, bool add_noise, SideQueue* defer);
void train_step16(vh_vae* h, const int64_t* dev_idx, bool eps_injected, bool masks_injected);
void encode16(vh_vae* h, float* latent);
}  // namespace step16

// ---- host <-> padded device layout ---------------------------------------------------------------
void upload_tensor(vh_vae* h, int ti, const float* data) {
    Tensor& t = h->tensors[ti];
    std::vector<float> buf((size_t)t.slot, 0.0f);
    for (int r = 0; r < t.rows; ++r)
        memcpy(buf.data() + (size_t)r * t.cols_p, data + (size_t)r * t.cols, sizeof(float) * t.cols);
    VH_HIP(hipMemcpyAsync(h->pptr(ti), buf.data(), sizeof(float) * t.slot, hipMemcpyHostToDevice, h->stream));
    VH_HIP(hipStreamSynchronize(h->stream));
}

void download_padded(vh_vae* h, const Tensor& t, const float* dev, float* out) {
    std::vector<float> buf((size_t)t.padded());
    VH_HIP(hipMemcpyAsync(buf.data(), dev, sizeof(float) * buf.size(), hipMemcpyDeviceToHost, h->stream));
    VH_HIP(hipStreamSynchronize(h->stream));
    for (int r = 0; r < t.rows; ++r)
        memcpy(out + (size_t)r * t.cols, buf.data() + (size_t)r * t.cols_p, sizeof(float) * t.cols);
}

void init_parameters(vh_vae* h) {
    std::mt19937_64 rng(h->cfg.seed * 0x9E3779B97F4A7C15ull + 12345);
    auto uniform_fill = [&](int ti, double bound) {
        Tensor& t = h->tensors[ti];
        std::uniform_real_distribution<double> dist(-bound, bound);
        std::vector<float> v((size_t)t.logical());
        for (auto& x : v) x = (float)dist(rng);
        upload_tensor(h, ti, v.data());
    };
    auto const_fill = [&](int ti, float value) {
        std::vector<float> v((size_t)h->tensors[ti].logical(), value);
        upload_tensor(h, ti, v.data());
    };
    // torch.nn.Linear default: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for weight and bias
    for (auto& hl : h->hidden) {
        const double b = 1.0 / std::sqrt((double)hl.nin);
        uniform_fill(hl.tW, b);
        uniform_fill(hl.tb, b);
        const_fill(hl.tG, 1.0f);
        const_fill(hl.tB, 0.0f);
        const_fill(hl.tRM, 0.0f);
        const_fill(hl.tRV, 1.0f);
    }
    const double bmu = 1.0 / std::sqrt((double)h->hidden[h->nl - 1].nout);
    uniform_fill(h->tWmu, bmu);
    uniform_fill(h->tbmu, bmu);
    const double bo = 1.0 / std::sqrt((double)h->hidden[2 * h->nl - 1].nout);
    uniform_fill(h->tWo, bo);
    uniform_fill(h->tbo, bo);
}

int dw_splits(int M, int N, int K, int tile) {
    const int bm = (tile == 0 || tile == 3) ? 64 : 128, bn = tile == 2 ? 32 : (tile == 3 ? 64 : 128);
    const int tiles = (int)(ceil_div(M, bm) * ceil_div(N, bn));
    int want = (int)std::max<int64_t>(1, ceil_div(g_tuning.dw_workgroups, tiles));
    want = std::min(want, K / 32);
    return std::max(1, want);
}
int dw_tile(int M, int N) {
    if (N <= 32) return 2;
    return 3;
}

// (re)allocate everything that depends on the batch size
void prepare_batch(vh_vae* h, int bs) {
    if (bs == h->bs) return;
    VH_HIP(hipStreamSynchronize(h->stream));
    const int bs_p = (int)round_up(bs, kRowPad);
    h->bs = bs;
    h->bs_p = bs_p;
    int maxw = std::max(h->D_p, h->L_p);
    for (auto& hl : h->hidden) maxw = std::max(maxw, hl.nout_p);
    h->Xb.ensure((size_t)bs_p * h->D_p);
    h->Wb.ensure(bs_p);
    h->MU.ensure((size_t)bs_p * h->L_p);
    h->Z.ensure((size_t)bs_p * h->L_p);
    h->EPS.ensure((size_t)bs_p * h->L_p);
    h->R.ensure((size_t)bs_p * h->D_p);
    h->dR.ensure((size_t)bs_p * h->D_p);
    h->dMUk.ensure((size_t)bs_p * h->L_p);
    h->DA.ensure((size_t)bs_p * maxw);
    h->dMU.ensure((size_t)bs_p * h->L_p);
    const int nrb = bs_p / kRB;
    h->loss_blocks = bs_p / 4;
    h->loss_part.ensure((size_t)h->loss_blocks * 4);
    h->Lb.ensure(bs_p);
    if (h->NL > 0) h->lab_part.ensure((size_t)h->loss_blocks * 2);
    h->out_sm.ensure((size_t)bs_p * std::max(1, h->S));
    h->skinny.ensure((size_t)kSkinnySplits * bs_p * h->L_p);
    for (auto& hl : h->hidden) {
        hl.H.ensure((size_t)bs_p * hl.nout_p);
        hl.A.ensure((size_t)bs_p * hl.nout_p);
        hl.DA.ensure((size_t)bs_p * hl.nout_p);
        hl.DZ.ensure((size_t)bs_p * hl.nout_p);
    }
    {   // fp64 accumulators of every hidden layer, contiguous so that one memset per step clears them
        size_t tot = 0;
        for (auto& hl : h->hidden) tot += (size_t)5 * hl.nout_p;
        tot += (size_t)h->L_p + h->D_p;   // bf16 step: bias gradients of mu / output layer
        h->statbuf.ensure(tot);
        size_t off = 0;
        for (auto& hl : h->hidden) {
            hl.fstat = h->statbuf.p + off;
            hl.bstat = hl.fstat + (size_t)2 * hl.nout_p;
            hl.dbias = hl.bstat + (size_t)2 * hl.nout_p;
            off += (size_t)5 * hl.nout_p;
        }
        h->dbias_mu = h->statbuf.p + off;
        h->dbias_out = h->dbias_mu + h->L_p;
    }
    // gradient slabs: weights get split-K slabs, biases row-block partials, BN affine a single slab
    size_t total = 0;
    auto plan = [&](int ti, int nslab, int64_t stride) {
        Tensor& t = h->tensors[ti];
        t.nslab = nslab;
        t.stride = stride;
        t.slab = reinterpret_cast<float*>(total);  // offset for now
        total += (size_t)round_up((int64_t)nslab * stride, 256);
    };
    for (auto& hl : h->hidden) {
        const int tile = dw_tile(hl.nout_p, hl.nin_p);
        plan(hl.tW, h->bf16 ? step16::dw_splits16(hl.nout_p, hl.nin_p, bs_p) : dw_splits(hl.nout_p, hl.nin_p, bs_p, tile),
             (int64_t)hl.nout_p * hl.nin_p);
        // bias / gamma / beta gradients are the fp64 accumulators filled by the GEMM loaders / epilogues
        h->tensors[hl.tb].dsrc = hl.dbias;
        h->tensors[hl.tG].dsrc = hl.bstat + hl.nout_p;   // sum dA * xhat
        h->tensors[hl.tB].dsrc = hl.bstat;               // sum dA
        h->tensors[hl.tG].dsrc_allrank = h->tensors[hl.tB].dsrc_allrank = true;
        for (int ti : {hl.tb, hl.tG, hl.tB}) { h->tensors[ti].nslab = 0; h->tensors[ti].stride = 0; h->tensors[ti].slab = nullptr; }
    }
    {
        const int hlast = h->hidden[h->nl - 1].nout_p;
        plan(h->tWmu, h->bf16 ? step16::dw_splits16(h->L_p, hlast, bs_p) : dw_splits(h->L_p, hlast, bs_p, dw_tile(h->L_p, hlast)),
             (int64_t)h->L_p * hlast);
        const int hdec = h->hidden[2 * h->nl - 1].nout_p;
        plan(h->tWo, h->bf16 ? step16::dw_splits16(h->D_p, hdec, bs_p) : dw_splits(h->D_p, hdec, bs_p, dw_tile(h->D_p, hdec)),
             (int64_t)h->D_p * hdec);
        if (h->bf16) {   // bf16 step: fp64 accumulators filled by the transposing kernels
            h->tensors[h->tbmu].dsrc = h->dbias_mu;
            h->tensors[h->tbo].dsrc = h->dbias_out;
            for (int ti : {h->tbmu, h->tbo}) { h->tensors[ti].nslab = 0; h->tensors[ti].stride = 0; h->tensors[ti].slab = nullptr; }
        } else {
            h->tensors[h->tbmu].dsrc = nullptr;
            h->tensors[h->tbo].dsrc = nullptr;
            plan(h->tbmu, nrb, h->L_p);
            plan(h->tbo, nrb, h->D_p);
        }
    }
    h->slabs.ensure(total);
    OptTable tab;
    memset(&tab, 0, sizeof(tab));
    int nblk = 0;
    for (auto& t : h->tensors) {
        if (!t.optimised) continue;
        VH_REQUIRE(tab.n < kMaxOptTensors, "too many parameter tensors");
        if (!t.dsrc) t.slab = h->slabs.p + reinterpret_cast<size_t>(t.slab);
        TensorDesc& d = tab.d[tab.n];
        d.dsrc = t.dsrc;
        d.dscale = t.dsrc_allrank && h->comm && h->comm->world > 1 ? 1.0f / (float)h->comm->world : 1.0f;
        d.slab = t.slab;
        d.nslab = t.nslab;
        d.stride = t.stride;
        d.p_off = (int64_t)t.off;
        d.size = t.padded();
        tab.blk_start[tab.n] = nblk;
        nblk += (int)ceil_div(t.padded(), 1024);
        tab.n++;
    }
    tab.blk_start[tab.n] = nblk;
    h->opt_tab = tab;
    h->G.ensure(h->flat_elems);
    h->opt_tab_flat = tab;
    for (int i = 0; i < tab.n; ++i) {
        TensorDesc& d = h->opt_tab_flat.d[i];
        d.slab = h->G.p + d.p_off; d.nslab = 1; d.stride = 0; d.dsrc = nullptr;
    }
    h->opt_blocks = nblk;
    h->opt_part.ensure((size_t)h->opt_blocks * 2);
    if (h->bf16) {
        step16::prepare_batch16(h);
        step16::build_opt16_table(h);
    }
    VH_HIP(hipMemset(h->statbuf.p, 0, h->statbuf.bytes()));
    h->stat_clean = true;
}

struct DropCfg {
    float scale = 1.0f;
    uint32_t thresh = 0;
    bool injected = false;
};

DropCfg drop_cfg(vh_vae* h, bool training, bool injected) {
    DropCfg d;
    if (!training || h->cfg.dropout <= 0.0f) return d;
    d.scale = 1.0f / (1.0f - h->cfg.dropout);
    d.thresh = (uint32_t)std::min<double>(4294967295.0, (double)h->cfg.dropout * 4294967296.0);
    d.injected = injected;
    return d;
}

uint64_t layer_key(vh_vae* h, int layer) {
    const uint64_t rank = h->comm ? (uint64_t)h->comm->rank : 0ull;
    return (h->cfg.seed * 0xD1342543DE82EF95ull) ^ (uint64_t)layer ^ (rank << 52);
}

const unsigned long long* step_ptr(vh_vae* h) { return h->step_src ? h->step_src : &h->state.p->step; }
const long long* batch_ptr(vh_vae* h) { return h->batch_src ? h->batch_src : &h->state.p->batch; }

// injected dropout keep-masks of the hidden layers [first, last): per layer [bs][nout] bytes, concatenated in layer order
void upload_masks(vh_vae* h, const uint8_t* masks, int bs, int first = 0, int last = -1) {
    size_t off = 0;
    if (last < 0) last = (int)h->hidden.size();
    for (int li = first; li < last; ++li) {
        Hidden& hl = h->hidden[li];
        std::vector<uint8_t> buf((size_t)h->bs_p * hl.nout_p, 0);
        for (int r = 0; r < bs; ++r)
            memcpy(buf.data() + (size_t)r * hl.nout_p, masks + off + (size_t)r * hl.nout, hl.nout);
        off += (size_t)bs * hl.nout;
        hl.mask.ensure(buf.size());
        VH_HIP(hipMemcpyAsync(hl.mask.p, buf.data(), buf.size(), hipMemcpyHostToDevice, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
    }
}

// arm the exact-timestamp event pair for the next GEMM launch (circular ring of kProbeRing pairs; a launch is
// simply not sampled while the ring is full)
void probe_arm(vh_vae* h) {
    if (!h->probe_on || h->probe_used >= kProbeRing) return;
    if ((h->probe_calls++ % h->probe_every) != 0) return;
    if ((int)h->ev_a.size() < kProbeRing) {
        hipEvent_t a, b;
        VH_HIP(hipEventCreate(&a));
        VH_HIP(hipEventCreate(&b));
        h->ev_a.push_back(a);
        h->ev_b.push_back(b);
    }
    const int slot = (h->probe_head + h->probe_used) % kProbeRing;
    if (slot >= (int)h->ev_a.size()) return;   // ring still growing and wrapped: skip this sample
    t_probe_start = h->ev_a[slot];
    t_probe_stop = h->ev_b[slot];
    h->probe_used++;
}

void probe_pop(vh_vae* h) {
    float ms = 0.f;
    VH_HIP(hipEventElapsedTime(&ms, h->ev_a[h->probe_head], h->ev_b[h->probe_head]));
    h->probe_ms += ms;
    h->probe_launches++;
    h->probe_head = (h->probe_head + 1) % kProbeRing;
    h->probe_used--;
}

// after a stream synchronisation: every armed pair is complete
void probe_collect(vh_vae* h) {
    while (h->probe_used > 0) probe_pop(h);
}

// without waiting: pop the pairs whose kernels have already retired (oldest first)
void probe_collect_ready(vh_vae* h) {
    while (h->probe_used > 0 && hipEventQuery(h->ev_b[h->probe_head]) == hipSuccess) probe_pop(h);
}

// The side stream carries everything that is off the critical path of a step (weight-gradient GEMMs,
// running statistics): fork after the producing kernel, join before the optimiser.
void fork_side(vh_vae* h) {
    if (h->side == h->stream) return;   // single-stream mode
    VH_HIP(hipEventRecord(h->ev_fork, h->stream));
    VH_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
}
// An event record costs the recording stream one more barrier packet (~6 us before its next kernel starts, 7
// forks per step).  Instead the producing kernel is launched with the fork event as its stop event
// (hipExtLaunchKernelGGL: the event is the dispatch's own completion signal) and the side stream waits on that.
bool fork_from_kernel(const vh_vae* h) { return h->side != h->stream && h->fork_ext; }
template <class K, class... Args>
void launch_forking(vh_vae* h, K kernel, dim3 grid, dim3 block, size_t smem, Args... args) {
    if (fork_from_kernel(h)) {
        hipExtLaunchKernelGGL(kernel, grid, block, smem, h->stream, nullptr, h->ev_fork, 0, args...);
        VH_HIP(hipGetLastError());
        VH_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    } else {
        hipLaunchKernelGGL(kernel, grid, block, smem, h->stream, args...);
        VH_HIP(hipGetLastError());
        fork_side(h);
    }
}
void join_side(vh_vae* h) {
    if (h->side == h->stream) return;
    VH_HIP(hipEventRecord(h->ev_join, h->side));
    VH_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
}

// ---- forward ---------------------------------------------------------------------------------------
// Synchronised BatchNorm under data parallelism: the fp64 batch sums every GEMM epilogue accumulates are all-reduced
// over the ranks (forward: sum h, sum h^2; backward: sum dA, sum dA xhat -- 2 x n doubles each, on the main stream) and
// every consumer divides by the all-rank batch, so the data-parallel step computes the statistics of the whole batch
// like the single-process reference (encode.py:238,246,264).
bool syncbn_active(const vh_vae* h) { return h->comm != nullptr && h->syncbn && h->global_bs > 0; }
int stat_bs(const vh_vae* h) { return syncbn_active(h) ? h->global_bs : h->bs; }
// 1 when this step's BatchNorm sums really are all-rank sums: only then are the gamma / beta gradient accumulators scaled
// by 1 / world before the gradient all-reduce (same predicate as sync_stats; a step without a global batch keeps
// per-rank statistics and per-rank gradients)
int allrank_stats(const vh_vae* h) { return syncbn_active(h) && h->comm->world > 1 ? 1 : 0; }
void sync_stats(vh_vae* h, double* stats, int n_p) {
    if (syncbn_active(h) && h->comm->world > 1) rccl_allreduce_sum_f64(h->comm, stats, (size_t)2 * n_p, h->stream);
}

BnSrc bn_src(vh_vae* h, const Hidden& hl) {
    BnSrc b;
    b.fstat = hl.fstat;
    b.gamma = h->pptr(hl.tG);
    b.beta = h->pptr(hl.tB);
    b.n_p = hl.nout_p;
    b.bs = stat_bs(h);
    return b;
}

// Xb/Wb must hold the batch.
// training: every hidden layer is ONE launch -- the GEMM applies the previous layer's BatchNorm while
//   staging its A operand (XF_BN), and its epilogue does bias + LeakyReLU + dropout, stores H and
//   accumulates the batch sums of H for its own BatchNorm.  The normalised activations are never written.
// eval: running statistics folded into the epilogue (EPI_HIDDEN_EVAL), activations in hl.A.
// Parts of a pass (joint trainer, vaevae.hpp): PASS_DECODER starts at the latent code -- `z_src` [bs_p][L_p] plays mu, the
// noise is added to it (semisupervised_encode.py:903-908: VAEVamb._decode(VAEVamb.reparameterize(mu_sup))) -- and touches
// neither the encoder's activations nor its running statistics; PASS_ENCODER (backward only) starts at h->dMUk.
enum { PASS_FULL = 0, PASS_DECODER = 1, PASS_ENCODER = 2 };

// BatchNorm1d's running statistics after a training-mode forward of the layers `part` names (momentum 0.1, unbiased variance),
// from the batch sums the forward GEMMs' epilogues left in statbuf
void update_running(vh_vae* h, int part, hipStream_t st) {
    RunningTable rt;
    memset(&rt, 0, sizeof(rt));
    int maxn = 0;
    for (int li = part == PASS_DECODER ? h->nl : 0; li < 2 * h->nl; ++li) {
        Hidden& hl = h->hidden[li];
        rt.fstat[rt.n] = hl.fstat; rt.rm[rt.n] = h->pptr(hl.tRM); rt.rv[rt.n] = h->pptr(hl.tRV);
        rt.n_p[rt.n] = hl.nout_p;
        maxn = std::max(maxn, hl.nout_p);
        rt.n++;
    }
    hipLaunchKernelGGL(vae_bn_running_kernel, dim3((unsigned)ceil_div(maxn, 256), rt.n), dim3(256), 0, st, rt, stat_bs(h));
    VH_HIP(hipGetLastError());
}

void forward(vh_vae* h, bool training, bool eps_injected, bool masks_injected, bool add_noise, int part = PASS_FULL,
             const float* z_src = nullptr, const float* zero_bias = nullptr) {
    const int bs = h->bs, bs_p = h->bs_p;
    hipStream_t s = h->stream;
    const DropCfg dc = drop_cfg(h, training, masks_injected);
    if (training) {
        // the optimiser's finalize kernel leaves the accumulators zeroed; only a forward-only call dirties them
        if (!h->stat_clean) VH_HIP(hipMemsetAsync(h->statbuf.p, 0, h->statbuf.bytes(), s));
        h->stat_clean = false;
    }
    const float* in = h->Xb.p;
    int in_w = h->D_p;
    const Hidden* prev = nullptr;   // the layer whose BatchNorm still has to be applied to `in` (training)
    auto hidden_layer = [&](int li) {
        Hidden& hl = h->hidden[li];
        const int tile = fwd_tile(bs_p, hl.nout_p);
        GemmArgs g = base_args(h->bf16);
        g.A = in; g.lda = in_w;
        g.B = h->pptr(hl.tW); g.ldb = hl.nin_p;
        g.M = bs_p; g.N = hl.nout_p; g.K = hl.nin_p; g.k_per_split = g.K;
        g.bias = h->pptr(hl.tb);
        g.m_real = bs;
        g.wide_k = 1;
        if (training) {
            const bool probed = h->probe_on && li == h->probe_layer;
            g.C = hl.H.p; g.ldc = hl.nout_p;
            g.fstat_out = hl.fstat;
            g.drop_scale = dc.scale; g.drop_thresh = dc.thresh; g.drop_key = layer_key(h, li);
            g.step_ptr = step_ptr(h);
            g.drop_mask = dc.injected ? hl.mask.p : nullptr; g.ld_mask = hl.nout_p;
            if (probed) { probe_arm(h); h->probe_flops = 2.0 * bs * (double)hl.nin * hl.nout; }
            if (prev) {
                g.bnA = bn_src(h, *prev);
                gemm_tile<true, true, EPI_HIDDEN_TRAIN, XF_BN>(s, tile, g, 1);
            } else {
                gemm_tile<true, true, EPI_HIDDEN_TRAIN>(s, tile, g, 1);
            }
            sync_stats(h, hl.fstat, hl.nout_p);
            in = hl.H.p;
            prev = &hl;
        } else {
            hipLaunchKernelGGL(vae_bn_eval_coeff_kernel, dim3((unsigned)ceil_div(hl.nout_p, 256)), dim3(256), 0, s,
                               hl.nout_p, h->pptr(hl.tG), h->pptr(hl.tB), h->pptr(hl.tRM), h->pptr(hl.tRV),
                               hl.scale.p, hl.shift.p);
            VH_HIP(hipGetLastError());
            g.C = hl.A.p; g.ldc = hl.nout_p;
            g.scale = hl.scale.p; g.shift = hl.shift.p;
            gemm_tile<true, true, EPI_HIDDEN_EVAL>(s, tile, g, 1);
            in = hl.A.p;
        }
        in_w = hl.nout_p;
    };
    if (part != PASS_DECODER)
        for (int li = 0; li < h->nl; ++li) hidden_layer(li);
    int mu_slabs = 1;
    if (part != PASS_DECODER) {   // mu = a * Wmu^T + bmu  (encode.py:268).  The output is only nlatent wide, so the contraction is
        // split over up to 8 workgroup slices (slabs); bias and the slab sum are folded into the
        // reparameterisation kernel below.
        GemmArgs g = base_args(h->bf16);
        g.A = in; g.lda = in_w;
        g.B = h->pptr(h->tWmu); g.ldb = in_w;
        g.C = h->skinny.p; g.ldc = h->L_p;
        g.M = bs_p; g.N = h->L_p; g.K = in_w;
        const int want = std::max(1, std::min(kSkinnySplits, in_w / 64));
        g.k_per_split = (int)round_up(ceil_div(in_w, want), 32);
        mu_slabs = (int)ceil_div(in_w, g.k_per_split);
        g.slab_stride = (int64_t)bs_p * h->L_p;
        if (prev) {
            g.bnA = bn_src(h, *prev);
            gemm_tile<true, true, EPI_SPLITK, XF_BN>(s, fwd_tile(bs_p, h->L_p), g, mu_slabs);
        } else {
            gemm_tile<true, true, EPI_SPLITK>(s, fwd_tile(bs_p, h->L_p), g, mu_slabs);
        }
    }
    {   // latent = mu + eps  (encode.py:276-286; sigma == 1); eps injected (parity) or generated in place
        const int64_t tot = (int64_t)bs_p * h->L_p;
        const bool ext = part == PASS_DECODER;
        VH_REQUIRE(!ext || (z_src != nullptr && zero_bias != nullptr), "a decoder-only pass needs its latent input");
        hipLaunchKernelGGL(vae_reparam_kernel, dim3((unsigned)ceil_div(tot, 256)), dim3(256), 0, s, ext ? z_src : h->skinny.p,
                           mu_slabs, (int64_t)bs_p * h->L_p, ext ? zero_bias : h->pptr(h->tbmu),
                           eps_injected ? h->EPS.p : nullptr, layer_key(h, 0xEE), step_ptr(h), add_noise ? 1 : 0, h->MU.p,
                           h->Z.p, bs, h->L, h->L_p, bs_p);
        VH_HIP(hipGetLastError());
    }
    in = h->Z.p;
    in_w = h->L_p;
    prev = nullptr;
    for (int li = h->nl; li < 2 * h->nl; ++li) hidden_layer(li);
    {   // reconstruction = a * Wo^T + bo  (encode.py:294)
        GemmArgs g = base_args(h->bf16);
        g.A = in; g.lda = in_w;
        g.B = h->pptr(h->tWo); g.ldb = in_w;
        g.C = h->R.p; g.ldc = h->D_p;
        g.M = bs_p; g.N = h->D_p; g.K = in_w; g.k_per_split = g.K;
        g.bias = h->pptr(h->tbo);
        g.wide_k = 1;
        const bool ext_fork = training && fork_from_kernel(h);
        if (ext_fork) t_fork_stop = h->ev_fork;   // the running-statistics kernel below forks off this GEMM
        if (prev) {
            g.bnA = bn_src(h, *prev);
            gemm_tile<true, true, EPI_BIAS, XF_BN>(s, fwd_tile(bs_p, h->D_p), g, 1);
        } else {
            gemm_tile<true, true, EPI_BIAS>(s, fwd_tile(bs_p, h->D_p), g, 1);
        }
        if (ext_fork) VH_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    }
    if (training) {
        // running statistics (momentum 0.1, unbiased variance): off the critical path, on the side stream
        if (!fork_from_kernel(h)) fork_side(h);
        update_running(h, part, h->side);
    }
}

// kld_w < 0: the model's own weight; 0: a pass whose Kullback-Leibler term lives elsewhere (calc_loss_joint)
void loss_and_seed(vh_vae* h, float kld_w = -1.0f) {
    const int bs_global = h->global_bs > 0 ? h->global_bs : h->bs;
    LossArgs a;
    a.R = h->R.p; a.X = h->Xb.p; a.ld = h->D_p;
    a.MU = h->MU.p; a.ldl = h->L_p;
    a.inv_b2 = (float)(1.0 / ((double)bs_global * (double)bs_global));
    a.bs = h->bs; a.bs_p = h->bs_p; a.S = h->S; a.L = h->L;
    a.ce_w = h->ce_w; a.ab_w = h->ab_w; a.sse_w = h->sse_w; a.kld_w = kld_w < 0.f ? h->kld_w : kld_w;
    a.dR = h->dR.p; a.dMUk = h->dMUk.p; a.part = h->loss_part.p;
    a.NL = h->NL; a.lab0 = h->lab0; a.ntnf = h->ntnf; a.nab = h->nab; a.Lb = h->Lb.p;
    a.lab_part = h->NL > 0 ? h->lab_part.p : nullptr;
    a.leaf_masks = h->leaf_masks.p; a.n_leaves = h->n_leaves;
    // the scalar reduction (loss means, sum of weights) is only needed by the optimiser: side stream; the
    // output layer's weight gradient (backward) forks off the same point
    launch_forking(h, vae_loss_kernel, dim3(h->loss_blocks), dim3(256), 0, a);
    hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(kLossFinThreads), 0, h->side, h->loss_part.p, h->loss_blocks,
                       h->Wb.p, h->bs, h->gwsum_src, bs_global, h->state.p, (const float*)a.lab_part);
    VH_HIP(hipGetLastError());
}

// dW slabs = dZ^T * In on the side stream (both operands row-contiguous along the batch).
//   inbn: when given, B = raw H of the previous hidden layer and its BatchNorm is applied on load (XF_BN)
void grad_weight(vh_vae* h, int tW, const float* dZ, int out_p, const float* In, int in_p, const BnSrc* inbn,
                 hipStream_t st = nullptr) {
    if (!st) st = h->side;
    Tensor& t = h->tensors[tW];
    const int tile = dw_tile(out_p, in_p);
    GemmArgs g = base_args(h->bf16);
    g.A = dZ; g.lda = out_p;
    g.B = In; g.ldb = in_p;
    g.C = t.slab; g.ldc = in_p;
    g.M = out_p; g.N = in_p; g.K = h->bs_p;
    g.k_per_split = (int)round_up(ceil_div(h->bs_p, t.nslab), 32);
    g.slab_stride = t.stride;
    const int splits = (int)ceil_div(h->bs_p, g.k_per_split);
    if (splits < t.nslab)  // unused slabs must read as zero
        VH_HIP(hipMemsetAsync(t.slab + (int64_t)splits * t.stride, 0, sizeof(float) * (t.nslab - splits) * t.stride,
                              st));
    if (inbn) {
        g.bnB = *inbn;
        gemm_tile<false, false, EPI_SPLITK, XF_NONE, XF_BN>(st, tile, g, splits);
    } else {
        gemm_tile<false, false, EPI_SPLITK>(st, tile, g, splits);
    }
}

// dIn = dZ * W on the main stream (dZ K-contiguous over the layer's outputs, W row-contiguous [out][in]).
//   below: the hidden layer that produced this layer's input; the epilogue then also accumulates the two
//          batch sums its BatchNorm backward needs (EPI_STORE_BNRED) and writes dIn into below->DA
//   to_latent: the input is the latent code: split-K slabs into h->skinny (returns their count)
int grad_input(vh_vae* h, const float* dZ, int out_p, int tW, int in_p, float* dIn, const Hidden* below,
               bool to_latent) {
    GemmArgs g = base_args(h->bf16);
    g.A = dZ; g.lda = out_p;
    g.B = h->pptr(tW); g.ldb = in_p;
    g.M = h->bs_p; g.N = in_p; g.K = out_p;
    g.m_real = h->bs;
    const int tile = fwd_tile(h->bs_p, in_p);
    if (to_latent) {
        int splits = 1;
        g.k_per_split = g.K;
        g.C = dIn; g.ldc = in_p;
        if (in_p <= 32 && out_p >= 128) {
            const int want = std::max(1, std::min(kSkinnySplits, out_p / 64));
            g.k_per_split = (int)round_up(ceil_div(out_p, want), 32);
            splits = (int)ceil_div(out_p, g.k_per_split);
            g.C = h->skinny.p;
            g.slab_stride = (int64_t)h->bs_p * in_p;
        }
        gemm_tile<true, false, EPI_SPLITK>(h->stream, tile, g, splits);
        return splits;
    }
    g.C = dIn; g.ldc = in_p;
    g.k_per_split = g.K;
    g.Hbelow = below->H.p;
    g.bnC = bn_src(h, *below);
    g.bstat_out = below->bstat;
    gemm_tile<true, false, EPI_STORE_BNRED>(h->stream, tile, g, 1);
    sync_stats(h, below->bstat, below->nout_p);
    return 1;
}

// Backward of one step.  Critical path (main stream) per hidden layer: one bandwidth-bound dZ kernel and
// one dIn GEMM whose epilogue leaves dA plus the BatchNorm-backward sums of the layer below (so BatchNorm
// backward has no reduction / finalize kernels).  Weight gradients run on the side stream.
// part (joint trainer): PASS_DECODER stops at the latent code -- h->dMU then holds d loss / d z and nothing upstream of it is
// touched; PASS_ENCODER starts there: h->dMUk holds d loss / d mu (put there by the caller), the decoder is skipped.
void backward(vh_vae* h, bool masks_injected, int part = PASS_FULL) {
    const int bs = h->bs, bs_p = h->bs_p, nrb = bs_p / kRB;
    const DropCfg dc = drop_cfg(h, true, masks_injected);
    const int nl = h->nl;
    if (part != PASS_ENCODER) {   // output layer: dR is ready (loss kernel)
        Hidden& last = h->hidden[2 * nl - 1];
        const BnSrc inbn = bn_src(h, last);
        // (the side stream already waits on the loss kernel: loss_and_seed)
        grad_weight(h, h->tWo, h->dR.p, h->D_p, last.H.p, last.nout_p, &inbn);
        hipLaunchKernelGGL(vae_colsum_partial_kernel, dim3((unsigned)ceil_div(h->D_p, kCT), nrb), dim3(kCT, kRL), 0,
                           h->side, h->dR.p, (int64_t)h->D_p, h->D_p, bs_p, h->tensors[h->tbo].slab);
        VH_HIP(hipGetLastError());
        grad_input(h, h->dR.p, h->D_p, h->tWo, last.nout_p, last.DA.p, &last, false);
    }
    int latent_slabs = 1;
    // hidden layer li: its dA (hl.DA) and the sums in hl.bstat are complete on the main stream
    auto hidden_bwd = [&](int li) {
        Hidden& hl = h->hidden[li];
        DzArgs a;
        a.DA = hl.DA.p; a.H = hl.H.p; a.DZ = hl.DZ.p;
        a.n_p = hl.nout_p; a.bs = bs; a.bs_p = bs_p;
        a.bn = bn_src(h, hl);
        a.bstat = hl.bstat;
        a.drop_scale = dc.scale;
        a.drop_mask = dc.injected ? hl.mask.p : nullptr; a.ld_mask = hl.nout_p;
        a.dbias = hl.dbias;
        // The first encoder layer is the end of the chain: nothing is left on the main stream for its weight
        // gradient to hide behind, so it runs there too (no fork, and the optimiser does not wait for a
        // cross-stream hop: dZ -> dW -> join was 25 us longer on the side stream).
        const bool tail = li == 0;   // nothing left on the main stream to hide the last weight-gradient GEMM behind
        if (tail) {
            hipLaunchKernelGGL(vae_dz_kernel, dim3((unsigned)ceil_div(hl.nout_p, kDzCols), (unsigned)ceil_div(bs_p, kDzRows)),
                               dim3(32, kRL), 0, h->stream, a);
            VH_HIP(hipGetLastError());
        } else {
            launch_forking(h, vae_dz_kernel, dim3((unsigned)ceil_div(hl.nout_p, kDzCols), (unsigned)ceil_div(bs_p, kDzRows)),
                           dim3(32, kRL), 0, a);
        }
        const bool from_input = (li == 0) || (li == nl);          // input is Xb / Z: no BatchNorm to apply
        const Hidden* below = from_input ? nullptr : &h->hidden[li - 1];
        const float* In = li == 0 ? h->Xb.p : (li == nl ? h->Z.p : below->H.p);
        const int in_p = li == 0 ? h->D_p : (li == nl ? h->L_p : below->nout_p);
        if (below) {
            const BnSrc inbn = bn_src(h, *below);
            grad_weight(h, hl.tW, hl.DZ.p, hl.nout_p, In, in_p, &inbn);
        } else {
            grad_weight(h, hl.tW, hl.DZ.p, hl.nout_p, In, in_p, nullptr, tail ? h->stream : nullptr);
        }
        if (li == nl) {
            // (running the decoder half of the optimiser here, on the side stream, measured slower -- 364 vs 353 us per
            // step: it competes with the GEMMs -- and was removed)
            latent_slabs = grad_input(h, hl.DZ.p, hl.nout_p, hl.tW, in_p, h->DA.p, nullptr, true);
        } else if (li > 0) grad_input(h, hl.DZ.p, hl.nout_p, hl.tW, in_p, below->DA.p, below, false);
        // li == 0: the input gradient is never needed
    };
    if (part != PASS_ENCODER)
        for (int li = 2 * nl - 1; li >= nl; --li) hidden_bwd(li);
    else latent_slabs = 0;   // nothing arrives from a decoder: dMU = dMUk
    {   // latent: dMU = dZlat + d(KLD)/dmu; mu layer
        Hidden& enc_last = h->hidden[nl - 1];
        // latent_slabs == 1: the first decoder layer wrote dZlat into DA; otherwise split-K slabs in skinny
        const float* src = latent_slabs == 1 ? h->DA.p : h->skinny.p;
        launch_forking(h, vae_latent_bwd_kernel, dim3((unsigned)ceil_div(h->L_p, kCT), nrb), dim3(kCT, kRL), 0, src,
                       latent_slabs, (int64_t)bs_p * h->L_p, (const float*)h->dMUk.p, h->dMU.p, h->L_p, bs, bs_p,
                       h->tensors[h->tbmu].slab);
        if (part == PASS_DECODER) {
            join_side(h);
            return;
        }
        const BnSrc inbn = bn_src(h, enc_last);
        grad_weight(h, h->tWmu, h->dMU.p, h->L_p, enc_last.H.p, enc_last.nout_p, &inbn);
        grad_input(h, h->dMU.p, h->L_p, h->tWmu, enc_last.nout_p, enc_last.DA.p, &enc_last, false);
    }
    for (int li = nl - 1; li >= 0; --li) hidden_bwd(li);
    join_side(h);  // every weight gradient (and the running statistics) is complete before the optimiser
}

// combined: the flat buffer h->G already holds the complete, fully scaled gradient (joint trainer: the sum over the passes of a
// network) -- no slab reduction, no collective, no sum(w) factor
void optimizer_step(vh_vae* h, bool combined = false) {
    const OptTable* tab = &h->opt_tab;
    if (combined) {
        tab = &h->opt_tab_flat;
    } else if (h->comm) {
        // sum this rank's slabs into the flat buffer, all-reduce it over the ranks (RCCL, same stream,
        // no host synchronisation), then every rank applies the identical update
        hipLaunchKernelGGL(vae_reduce_slabs_kernel, dim3(h->opt_blocks), dim3(256), 0, h->stream, h->opt_tab, h->G.p,
                           allrank_stats(h));
        VH_HIP(hipGetLastError());
        rccl_allreduce_sum_f32(h->comm, h->G.p, h->flat_elems, h->stream);
        tab = &h->opt_tab_flat;
    }
    const int nblk = h->opt_blocks;
    if (nblk > 0) {
        hipLaunchKernelGGL(vae_dadapt_kernel, dim3(nblk), dim3(256), 0, h->stream, *tab, h->P.p, h->M1.p, h->M2.p,
                           h->Sv.p, h->state.p, h->opt_part.p, 0, h->adam_lr, combined ? 1.0f : 0.0f);
        VH_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(vae_dadapt_finalize_kernel, dim3(1), dim3(256), 0, h->stream, h->opt_part.p, h->opt_blocks,
                       h->state.p, h->statbuf.p, h->keep_grads ? 0 : (int)h->statbuf.n, h->adam_lr > 0.f ? 1 : 0);
    VH_HIP(hipGetLastError());
    h->stat_clean = !h->keep_grads;
}

void gather_rows(vh_vae* h, const int64_t* dev_idx) {
    auto kern = h->kind == VH_VAE_PLAIN ? vae_gather_kernel<false> : vae_gather_kernel<true>;
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(h->bs_p, 4)), dim3(64, 4), 0, h->stream, h->X.p,
                       h->ld_src, (int64_t)h->D_p, h->w.p, dev_idx, h->shuffle, batch_ptr(h), (int64_t)0, h->bs,
                       h->bs_p, h->Xb.p, h->Wb.p, LabelSrc{h->labels, h->lab0}, h->Lb.p);
    VH_HIP(hipGetLastError());
}

#include "vae_step16.hpp"

// One optimisation step on the batch `state->batch` of the row list dev_idx.  Every launch argument
// is identical from step to step, so the same sequence can be captured once and replayed.
void train_step_device(vh_vae* h, const int64_t* dev_idx, bool eps_injected, bool masks_injected) {
    if (h->bf16) { step16::train_step16(h, dev_idx, eps_injected, masks_injected); return; }
    gather_rows(h, dev_idx);
    forward(h, true, eps_injected, masks_injected, true);
    loss_and_seed(h);
    backward(h, masks_injected);
    optimizer_step(h);
}

// BatchNorm1d.num_batches_tracked: one per training-mode forward
void count_batches(vh_vae* h, long long n) {
    for (auto& hl : h->hidden) hl.batches_tracked += n;
}

// the epoch's batch cursor restarts at 0 (stream-ordered, no host synchronisation)
void reset_batch_index(vh_vae* h) {
    VH_HIP(hipMemsetAsync(&h->state.p->batch, 0, sizeof(long long), h->stream));
}

// Run the n_batches steps of the epoch whose row list is dev_idx: eager launches, no host synchronisation.
void run_epoch_steps(vh_vae* h, const int64_t* dev_idx, int64_t n_batches) {
    count_batches(h, n_batches);
    const bool dbg = option("vae.debug_timing", 0) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t b = 0; b < n_batches; ++b) {
        // bf16 step: the NEXT batch of the epoch is assembled on the side stream during this step (vae.prefetch_batch)
        // (narrow inputs only: at the C3 shape -- 1120 columns, 36 MB read + 55 MB written per batch -- the gather on the side stream
        // delays the weight gradients the optimiser waits for: 380.5 vs 366.4 us per step, profiles/r05g_step_prefetch_batch.txt)
        h->prefetch_next = h->bf16 && g_tuning.prefetch_batch && h->D_p <= g_tuning.prefetch_max_cols && b + 1 < n_batches && h->side != h->stream;
        train_step_device(h, dev_idx, false, false);
    }
    h->prefetch_next = false;
    if (dbg) {
        const auto t1 = std::chrono::steady_clock::now();
        VH_HIP(hipStreamSynchronize(h->stream));
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[vambhip] epoch: host enqueue %.2f ms, drain after enqueue %.2f ms (%lld steps)\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count(), (long long)n_batches);
    }
}

void read_state(vh_vae* h, StepState* out) {
    h->h_state.ensure(1);
    VH_HIP(hipMemcpyAsync(h->h_state.p, h->state.p, sizeof(StepState), hipMemcpyDeviceToHost, h->stream));
    VH_HIP(hipStreamSynchronize(h->stream));
    *out = *h->h_state.p;
}

void reset_epoch_sums(vh_vae* h) {
    const size_t off = offsetof(StepState, epoch_loss);
    VH_HIP(hipMemsetAsync(reinterpret_cast<char*>(h->state.p) + off, 0, sizeof(StepState) - off, h->stream));
}

int find_tensor(vh_vae* h, const char* name) {
    VH_REQUIRE(name != nullptr, "name is NULL");
    auto it = h->tindex.find(name);
    VH_REQUIRE(it != h->tindex.end(), "unknown parameter name '%s'", name);
    return it->second;
}

}  // namespace

namespace {

// VAE.__init__ (encode.py:171-257) and its two subclasses (semisupervised_encode.py:207-226, 457-478): same stack of layers,
// different input / reconstruction columns, loss terms and optimiser
int create_vae(const vh_vae_config* cfg, const vh_vae_labels_config* lab, vh_vae** out) {
    return guarded([&] {
        VH_REQUIRE(cfg != nullptr && out != nullptr, "NULL argument");
        *out = nullptr;
        const int kind = lab ? lab->kind : VH_VAE_PLAIN;
        VH_REQUIRE(kind == VH_VAE_PLAIN || kind == VH_VAE_CONCAT || kind == VH_VAE_LABELS, "unknown model kind %d", kind);
        VH_REQUIRE(kind == VH_VAE_PLAIN || lab->nlabels >= 1, "nlabels must be > 0, not %d", lab ? lab->nlabels : 0);
        VH_REQUIRE(!lab || lab->optimizer == VH_OPT_DADAPT_ADAM || (lab->optimizer == VH_OPT_ADAM && lab->lrate > 0),
                   "optimizer must be D-Adapt-Adam or Adam with a positive learning rate");
        // encode.py:182-208
        VH_REQUIRE(cfg->nlatent >= 1, "Minimum 1 latent neuron, not %d", cfg->nlatent);
        VH_REQUIRE(cfg->nsamples >= 1, "nsamples must be > 0, not %d", cfg->nsamples);
        VH_REQUIRE(cfg->nlayers >= 1 && cfg->nlayers <= VH_MAX_HIDDEN_LAYERS, "between 1 and %d hidden layers",
                   VH_MAX_HIDDEN_LAYERS);
        for (int i = 0; i < cfg->nlayers; ++i)
            VH_REQUIRE(cfg->nhiddens[i] >= 1, "Minimum 1 neuron per layer, not %d", cfg->nhiddens[i]);
        VH_REQUIRE(cfg->beta > 0, "beta must be > 0, not %g", (double)cfg->beta);
        VH_REQUIRE(cfg->alpha > 0 && cfg->alpha < 1, "alpha must be 0 < alpha < 1, not %g", (double)cfg->alpha);
        VH_REQUIRE(cfg->dropout >= 0 && cfg->dropout < 1, "dropout must be 0 <= dropout < 1, not %g",
                   (double)cfg->dropout);
        std::unique_ptr<vh_vae> h(new vh_vae());
        h->cfg = *cfg;
        h->nl = cfg->nlayers;
        h->S = cfg->nsamples;
        h->kind = kind;
        if (kind == VH_VAE_LABELS) {   // VAELabels: the network sees nothing but the one-hot labels
            h->S = 0; h->ntnf = 0; h->nab = 0;
        }
        if (kind != VH_VAE_PLAIN) {
            h->NL = lab->nlabels;
            h->lab0 = h->S + h->ntnf + h->nab;
            if (lab->optimizer == VH_OPT_ADAM) h->adam_lr = lab->lrate;
        }
        h->D = h->S + h->ntnf + h->nab + h->NL;
        h->D_p = (int)round_up(h->D, kColPad);
        h->L = cfg->nlatent;
        h->L_p = (int)round_up(h->L, kColPad);
        // encode.py:333-343
        const double a = cfg->alpha, S = cfg->nsamples;
        h->ce_w = cfg->nsamples == 1 ? 0.0f : (float)(((1 - a) * (S - 1)) / (S * std::log(S)));
        h->ab_w = (float)((1 - a) * (1 / S));
        h->sse_w = (float)(a / VH_NTNF);
        if (kind == VH_VAE_LABELS) h->ce_w = h->ab_w = h->sse_w = 0.0f;   // semisupervised_encode.py:248-257: CE(labels) + KLD
        h->kld_w = (float)(1.0 / ((double)cfg->nlatent * cfg->beta));
        // the main stream carries the critical path of a step (forward chain, dX chain, optimiser): highest
        // priority, so that its workgroups are dispatched ahead of the side stream's weight-gradient GEMMs
        // whenever both have work queued
        int prio_lo = 0, prio_hi = 0;
        VH_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));   // lo = least urgent (largest number)
        refresh_tuning();
        VH_HIP(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi));
        if (option("vae.single_stream", 0) != 0) h->side = h->stream;   // A/B: weight-gradient GEMMs on the main stream
        else VH_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_lo));
        // device-scope release: these events only order our two streams on the same GPU; the default
        // (system-scope) release flushes L2 for a host that never waits on them
        h->fork_ext = option("vae.fork_events", 0) == 0;
        const unsigned ev_flags = hipEventDisableTiming | hipEventReleaseToDevice;
        VH_HIP(hipEventCreateWithFlags(&h->ev_fork, ev_flags));
        VH_HIP(hipEventCreateWithFlags(&h->ev_join, ev_flags));

        h->hidden.resize(2 * h->nl);
        auto make_hidden = [&](int li, const std::string& lin, const std::string& norm, int nin, int nout) {
            Hidden& hl = h->hidden[li];
            hl.nin = nin; hl.nout = nout;
            hl.nin_p = (int)round_up(nin, kColPad);
            hl.nout_p = (int)round_up(nout, kColPad);
            hl.tW = h->add_tensor(lin + ".weight", nout, nin, true, true);
            hl.tb = h->add_tensor(lin + ".bias", 1, nout, true);
            hl.tG = h->add_tensor(norm + ".weight", 1, nout, true);
            hl.tB = h->add_tensor(norm + ".bias", 1, nout, true);
            hl.tRM = h->add_tensor(norm + ".running_mean", 1, nout, false);
            hl.tRV = h->add_tensor(norm + ".running_var", 1, nout, false);
            hl.mean.alloc(hl.nout_p); hl.invstd.alloc(hl.nout_p); hl.scale.alloc(hl.nout_p); hl.shift.alloc(hl.nout_p);
        };
        int nin = h->D;
        for (int i = 0; i < h->nl; ++i) {
            make_hidden(i, "encoderlayers." + std::to_string(i), "encodernorms." + std::to_string(i), nin,
                        cfg->nhiddens[i]);
            nin = cfg->nhiddens[i];
        }
        h->tWmu = h->add_tensor("mu.weight", h->L, nin, true, true);
        h->tbmu = h->add_tensor("mu.bias", 1, h->L, true);
        nin = h->L;
        for (int i = 0; i < h->nl; ++i) {
            const int nout = cfg->nhiddens[h->nl - 1 - i];
            make_hidden(h->nl + i, "decoderlayers." + std::to_string(i), "decodernorms." + std::to_string(i), nin, nout);
            nin = nout;
        }
        h->tWo = h->add_tensor("outputlayer.weight", h->D, nin, true, true);
        h->tbo = h->add_tensor("outputlayer.bias", 1, h->D, true);

        h->P.alloc(h->flat_elems); h->M1.alloc(h->flat_elems); h->M2.alloc(h->flat_elems); h->Sv.alloc(h->flat_elems);
        h->bnbuf.alloc(h->bn_elems);
        VH_HIP(hipMemsetAsync(h->P.p, 0, h->P.bytes(), h->stream));
        VH_HIP(hipMemsetAsync(h->M1.p, 0, h->M1.bytes(), h->stream));
        VH_HIP(hipMemsetAsync(h->M2.p, 0, h->M2.bytes(), h->stream));
        VH_HIP(hipMemsetAsync(h->Sv.p, 0, h->Sv.bytes(), h->stream));
        VH_HIP(hipMemsetAsync(h->bnbuf.p, 0, h->bnbuf.bytes(), h->stream));
        h->state.alloc(1);
        StepState st;
        memset(&st, 0, sizeof(st));
        st.d = 1e-6;  // DAdaptAdam d0
        VH_HIP(hipMemcpyAsync(h->state.p, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
        init_parameters(h.get());
        *out = h.release();
    });
}

}  // namespace

extern "C" {

int vh_vae_create(const vh_vae_config* cfg, vh_vae** out) { return create_vae(cfg, nullptr, out); }

int vh_vae_create_labelled(const vh_vae_config* cfg, const vh_vae_labels_config* lab, vh_vae** out) {
    if (lab == nullptr) return guarded([&] { VH_REQUIRE(false, "NULL argument"); });
    return create_vae(cfg, lab, out);
}

int vh_vae_destroy(vh_vae* h) {
    return guarded([&] { delete h; });
}

int vh_vae_param_size(vh_vae* h, const char* name, int64_t* n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && n != nullptr, "NULL argument");
        std::string s(name ? name : "");
        const std::string suffix = ".num_batches_tracked";
        if (s.size() > suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0) {
            find_tensor(h, (s.substr(0, s.size() - suffix.size()) + ".running_mean").c_str());
            *n = 1;
            return;
        }
        *n = h->tensors[find_tensor(h, name)].logical();
    });
}

int vh_vae_set_param(vh_vae* h, const char* name, const float* data, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && data != nullptr, "NULL argument");
        std::string s(name ? name : "");
        const std::string suffix = ".num_batches_tracked";
        if (s.size() > suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0) {
            const int ti = find_tensor(h, (s.substr(0, s.size() - suffix.size()) + ".running_mean").c_str());
            VH_REQUIRE(n == 1, "num_batches_tracked has one element");
            for (auto& hl : h->hidden)
                if (hl.tRM == ti) hl.batches_tracked = (long long)data[0];
            return;
        }
        const int ti = find_tensor(h, name);
        VH_REQUIRE(n == h->tensors[ti].logical(), "parameter '%s' has %lld elements, got %lld", name,
                   (long long)h->tensors[ti].logical(), (long long)n);
        upload_tensor(h, ti, data);
        if (h->bf16) {
            step16::refresh_shadows(h, ti);
            VH_HIP(hipStreamSynchronize(h->stream));
        }
    });
}

int vh_vae_get_param(vh_vae* h, const char* name, float* data, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && data != nullptr, "NULL argument");
        std::string s(name ? name : "");
        const std::string suffix = ".num_batches_tracked";
        if (s.size() > suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0) {
            const int ti = find_tensor(h, (s.substr(0, s.size() - suffix.size()) + ".running_mean").c_str());
            VH_REQUIRE(n == 1, "num_batches_tracked has one element");
            for (auto& hl : h->hidden)
                if (hl.tRM == ti) data[0] = (float)hl.batches_tracked;
            return;
        }
        const int ti = find_tensor(h, name);
        const Tensor& t = h->tensors[ti];
        VH_REQUIRE(n == t.logical(), "parameter '%s' has %lld elements, got %lld", name, (long long)t.logical(),
                   (long long)n);
        download_padded(h, t, h->pptr(ti), data);
    });
}

int vh_vae_get_grad(vh_vae* h, const char* name, float* data, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && data != nullptr, "NULL argument");
        const int ti = find_tensor(h, name);
        const Tensor& t = h->tensors[ti];
        VH_REQUIRE(t.optimised, "'%s' is a buffer, not a parameter", name);
        VH_REQUIRE(t.slab != nullptr || t.dsrc != nullptr, "no training step has run yet");
        VH_REQUIRE(h->bs > 0, "no training step has run yet");
        VH_REQUIRE(n == t.logical(), "parameter '%s' has %lld elements, got %lld", name, (long long)t.logical(),
                   (long long)n);
        StepState st;
        read_state(h, &st);
        if (h->bf16) {
            // the complete gradient (slab sums + BatchNorm completion) as the optimiser forms it, through the flat buffer
            hipLaunchKernelGGL(vae_grad16_kernel, dim3(h->opt16_blocks), dim3(256), 0, h->stream, h->opt16_tab.p, (const uint8_t*)h->opt16_blk2t.p,
                               stat_bs(h), h->G.p, 0, allrank_stats(h));
            VH_HIP(hipGetLastError());
            std::vector<float> buf((size_t)t.padded());
            VH_HIP(hipMemcpyAsync(buf.data(), h->G.p + t.off, sizeof(float) * buf.size(), hipMemcpyDeviceToHost, h->stream));
            VH_HIP(hipStreamSynchronize(h->stream));
            for (int r = 0; r < t.rows; ++r)
                for (int c = 0; c < t.cols; ++c)
                    data[(size_t)r * t.cols + c] = buf[(size_t)r * t.cols_p + c] * (float)st.wsum;
            return;
        }
        if (t.dsrc) {   // bias / gamma / beta of a hidden layer: fp64 accumulator
            std::vector<double> acc((size_t)t.padded());
            VH_HIP(hipMemcpy(acc.data(), t.dsrc, sizeof(double) * acc.size(), hipMemcpyDeviceToHost));
            for (int c = 0; c < t.cols; ++c) data[c] = (float)acc[c] * (float)st.wsum;
            return;
        }
        std::vector<float> slab((size_t)t.nslab * t.stride);
        VH_HIP(hipMemcpyAsync(slab.data(), t.slab, sizeof(float) * slab.size(), hipMemcpyDeviceToHost, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
        const float gscale = (float)st.wsum;  // the loss' sum(w) factor is applied by the optimiser kernel
        for (int r = 0; r < t.rows; ++r)
            for (int c = 0; c < t.cols; ++c) {
                float g = 0.f;  // same slab order as the optimiser kernel
                for (int s = 0; s < t.nslab; ++s) g += slab[(size_t)s * t.stride + (size_t)r * t.cols_p + c];
                data[(size_t)r * t.cols + c] = g * gscale;
            }
    });
}

int vh_vae_get_hidden(vh_vae* h, int layer, float* out, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && out != nullptr, "NULL argument");
        VH_REQUIRE(layer >= 0 && layer < 2 * h->nl, "layer %d out of range", layer);
        const Hidden& hl = h->hidden[layer];
        VH_REQUIRE(h->bs > 0 && (hl.H.p != nullptr || hl.H16.p != nullptr), "no training step has run yet");
        VH_REQUIRE(n == (int64_t)h->bs * hl.nout, "layer %d holds %d x %d activations, got %lld", layer, h->bs, hl.nout,
                   (long long)n);
        std::vector<float> buf((size_t)h->bs * hl.nout_p);
        if (h->bf16) {
            VH_REQUIRE(hl.H16.p != nullptr, "no training step has run yet");
            std::vector<bf16_t> b16(buf.size());
            VH_HIP(hipMemcpyAsync(b16.data(), hl.H16.p, sizeof(bf16_t) * b16.size(), hipMemcpyDeviceToHost, h->stream));
            VH_HIP(hipStreamSynchronize(h->stream));
            for (size_t i = 0; i < buf.size(); ++i) {
                const uint32_t u = (uint32_t)b16[i] << 16;
                memcpy(&buf[i], &u, 4);
            }
        } else {
            VH_HIP(hipMemcpyAsync(buf.data(), hl.H.p, sizeof(float) * buf.size(), hipMemcpyDeviceToHost, h->stream));
            VH_HIP(hipStreamSynchronize(h->stream));
        }
        for (int r = 0; r < h->bs; ++r)
            memcpy(out + (size_t)r * hl.nout, buf.data() + (size_t)r * hl.nout_p, sizeof(float) * hl.nout);
    });
}

}  // extern "C"

namespace {

void upload_dataset(vh_dataset* d, int S, int D_p, const float* depths, const float* tnf, const float* abundance,
                    const float* weights, int64_t n) {
    VH_REQUIRE(depths && tnf && abundance && weights, "NULL argument");
    VH_REQUIRE(n >= 1, "empty dataset");
    d->n = n;
    d->S = S;
    d->D_p = D_p;
    d->X.alloc((size_t)n * D_p);
    d->w.alloc((size_t)n);
    // assemble padded rows on the host in chunks (one H2D per chunk)
    const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / (D_p * 4));
    std::vector<float> buf((size_t)std::min(chunk, n) * D_p);
    for (int64_t lo = 0; lo < n; lo += chunk) {
        const int64_t hi = std::min(n, lo + chunk);
        std::fill(buf.begin(), buf.end(), 0.0f);
        for (int64_t r = lo; r < hi; ++r) {
            float* dst = buf.data() + (size_t)(r - lo) * D_p;
            memcpy(dst, depths + (size_t)r * S, sizeof(float) * S);
            memcpy(dst + S, tnf + (size_t)r * VH_NTNF, sizeof(float) * VH_NTNF);
            dst[S + VH_NTNF] = abundance[r];
        }
        VH_HIP(hipMemcpy(d->X.p + (size_t)lo * D_p, buf.data(), sizeof(float) * (size_t)(hi - lo) * D_p,
                         hipMemcpyHostToDevice));
    }
    VH_HIP(hipMemcpy(d->w.p, weights, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
}

}  // namespace

extern "C" {

int vh_vae_set_dataset(vh_vae* h, const float* depths, const float* tnf, const float* abundance, const float* weights,
                       int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        VH_REQUIRE(h->kind == VH_VAE_PLAIN, "models with a label block take a shared dataset (vh_dataset_set_labels + vh_vae_use_dataset)");
        upload_dataset(&h->own, h->S, h->D_p, depths, tnf, abundance, weights, n);
        h->n = n;
        h->X.p = h->own.X.p;
        h->w.p = h->own.w.p;
        h->ld_src = h->D_p;
        h->labels = nullptr;
    });
}

namespace {
void upload_labels(vh_dataset* d, const int32_t* labels, int64_t n, int32_t nlabels) {
    VH_REQUIRE(labels != nullptr, "NULL argument");
    VH_REQUIRE(nlabels >= 1, "nlabels must be > 0, not %d", nlabels);
    for (int64_t i = 0; i < n; ++i)
        VH_REQUIRE(labels[i] >= 0 && labels[i] < nlabels, "label %d of row %lld is outside [0, %d)", labels[i], (long long)i, nlabels);
    d->labels.alloc((size_t)n);
    VH_HIP(hipMemcpy(d->labels.p, labels, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    d->NL = nlabels;
    d->max_label = -1;
    for (int64_t i = 0; i < n; ++i) d->max_label = std::max(d->max_label, labels[i]);
}
}  // namespace

int vh_dataset_set_labels(vh_dataset* d, const int32_t* labels, int64_t n, int32_t nlabels) {
    return guarded([&] {
        VH_REQUIRE(d != nullptr, "NULL argument");
        VH_REQUIRE(n == d->n, "%lld labels for a dataset of %lld rows", (long long)n, (long long)d->n);
        upload_labels(d, labels, n, nlabels);
    });
}

int vh_dataset_create_labels(const int32_t* labels, int64_t n, int32_t nlabels, vh_dataset** out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr, "NULL argument");
        VH_REQUIRE(n >= 1, "empty dataset");
        std::unique_ptr<vh_dataset> d(new vh_dataset());
        d->n = n;
        upload_labels(d.get(), labels, n, nlabels);
        // VAELabels.calc_loss (semisupervised_encode.py:248-257) has no per-contig weights: unit weights
        std::vector<float> ones((size_t)n, 1.0f);
        d->w.alloc((size_t)n);
        VH_HIP(hipMemcpy(d->w.p, ones.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
        *out = d.release();
    });
}

int vh_dataset_create(const float* depths, const float* tnf, const float* abundance, const float* weights, int64_t n,
                      int nsamples, vh_dataset** out) {
    return guarded([&] {
        VH_REQUIRE(out != nullptr, "NULL argument");
        VH_REQUIRE(nsamples >= 1, "nsamples must be positive");
        std::unique_ptr<vh_dataset> d(new vh_dataset());
        upload_dataset(d.get(), nsamples, (int)round_up(nsamples + VH_NTNF + 1, kColPad), depths, tnf, abundance, weights, n);
        *out = d.release();
    });
}

int vh_dataset_destroy(vh_dataset* d) {
    delete d;
    return VH_OK;
}

int vh_vae_use_dataset(vh_vae* h, vh_dataset* d) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && d != nullptr, "NULL argument");
        VH_REQUIRE(d->S == h->S, "dataset has %d samples, the model %d", d->S, h->S);
        if (h->kind == VH_VAE_PLAIN) VH_REQUIRE(d->D_p == h->D_p, "dataset rows are %d wide, the model's %d", d->D_p, h->D_p);
        else VH_REQUIRE(d->labels.p != nullptr && d->NL == h->NL, "the model has %d label columns, the dataset %d", h->NL, d->NL);
        VH_REQUIRE(h->n_leaves == 0 || d->max_label < h->n_nodes, "label %d is not a node of the model's taxonomy (%d nodes)",
                   d->max_label, h->n_nodes);
        VH_HIP(hipStreamSynchronize(h->stream));
        h->own.X.release();
        h->own.w.release();
        h->n = d->n;
        h->X.p = d->X.p;
        h->w.p = d->w.p;
        h->ld_src = d->D_p;
        h->labels = h->kind == VH_VAE_PLAIN ? nullptr : d->labels.p;
    });
}

int vh_vae_train_step(vh_vae* h, const int64_t* rows, int64_t batch, const float* eps, const uint8_t* masks,
                      double losses[5]) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && rows != nullptr, "NULL argument");
        VH_REQUIRE(h->n > 0, "no dataset: call vh_vae_set_dataset first");
        VH_REQUIRE(batch >= 2, "BatchNorm1d needs more than 1 value per channel when training (batch=%lld)",
                   (long long)batch);
        VH_REQUIRE(batch <= (1 << 24), "batch too large");
        for (int64_t i = 0; i < batch; ++i)
            VH_REQUIRE(rows[i] >= 0 && rows[i] < h->n, "row %lld out of range", (long long)rows[i]);
        prepare_batch(h, (int)batch);
        h->perm.ensure((size_t)batch);
        h->h_perm.ensure((size_t)batch);
        memcpy(h->h_perm.p, rows, sizeof(int64_t) * batch);
        VH_HIP(hipMemcpyAsync(h->perm.p, h->h_perm.p, sizeof(int64_t) * batch, hipMemcpyHostToDevice, h->stream));
        if (eps) {
            std::vector<float> e((size_t)h->bs_p * h->L_p, 0.f);
            for (int r = 0; r < batch; ++r) memcpy(e.data() + (size_t)r * h->L_p, eps + (size_t)r * h->L, sizeof(float) * h->L);
            VH_HIP(hipMemcpyAsync(h->EPS.p, e.data(), sizeof(float) * e.size(), hipMemcpyHostToDevice, h->stream));
            VH_HIP(hipStreamSynchronize(h->stream));
        }
        if (masks && h->cfg.dropout > 0) upload_masks(h, masks, (int)batch);
        reset_epoch_sums(h);
        reset_batch_index(h);
        h->gwsum_src = nullptr;
        h->global_bs = 0;
        h->keep_grads = true;   // vh_vae_get_grad may be called afterwards
        train_step_device(h, h->perm.p, eps != nullptr, masks != nullptr && h->cfg.dropout > 0);
        h->keep_grads = false;
        count_batches(h, 1);
        StepState st;
        read_state(h, &st);
        probe_collect(h);
        if (losses)
            for (int i = 0; i < 5; ++i) losses[i] = st.step_loss[i];
        h->label_stats.assign({st.step_label[0], st.step_label[1]});
    });
}

}  // extern "C"

namespace {

// Enqueue one epoch (validation, shuffle key, per-batch weight sums, every step, the per-epoch collectives of the
// data-parallel path).  Nothing here waits for the GPU.
void enqueue_epoch(vh_vae* h, const int64_t* perm, int64_t n_batches, int64_t batch, int64_t global_batch,
                   const float* global_wsum) {
    {
        VH_REQUIRE(h != nullptr, "NULL argument");
        VH_REQUIRE(h->n > 0, "no dataset: call vh_vae_set_dataset first");
        VH_REQUIRE(n_batches >= 1, "no batches");
        VH_REQUIRE(batch >= 2, "BatchNorm1d needs more than 1 value per channel when training (batch=%lld)",
                   (long long)batch);
        VH_REQUIRE(batch <= (1 << 24), "batch too large");
        VH_REQUIRE(n_batches * batch <= h->n || perm != nullptr, "epoch needs %lld rows but the dataset has %lld",
                   (long long)(n_batches * batch), (long long)h->n);
        const bool dp = h->comm != nullptr && h->comm->world > 1;
        const bool dbg = option("vae.debug_timing", 0) != 0;
        const auto T0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (dbg) fprintf(stderr, "[vambhip] %-18s %.3f ms\n", what,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
        };
        if (global_batch <= 0) global_batch = batch;
        VH_REQUIRE(global_batch >= batch, "global batch smaller than the local batch");
        VH_REQUIRE(!(global_batch != batch && h->comm == nullptr), "a global batch needs a communicator");
        const int64_t total = n_batches * batch;
        prepare_batch(h, (int)batch);
        const int64_t* dev_idx = nullptr;
        if (perm) {
            h->perm.ensure((size_t)total);
            h->h_perm.ensure((size_t)total);
            {   // validate while staging into pinned memory (one pass over the row list)
                int64_t bad = -1;
                const int64_t n = h->n;
                for (int64_t i = 0; i < total; ++i) {
                    const int64_t r = perm[i];
                    if (r < 0 || r >= n) bad = r;
                    h->h_perm.p[i] = r;
                }
                VH_REQUIRE(bad == -1, "row %lld out of range", (long long)bad);
            }
            VH_HIP(hipMemcpyAsync(h->perm.p, h->h_perm.p, sizeof(int64_t) * total, hipMemcpyHostToDevice, h->stream));
            dev_idx = h->perm.p;
            h->shuffle.key = 0ull;
        } else {
            // device-side shuffle: a fresh keyed bijection of [0, n) per epoch (vae_kernels.hpp)
            h->epoch_counter++;
            const uint64_t rank = h->comm ? (uint64_t)h->comm->rank : 0ull;
            uint64_t key = (h->cfg.seed + 0x632BE59BD9B4E019ull) * 0x9E3779B97F4A7C15ull;
            key ^= (h->epoch_counter * 0xD6E8FEB86659FD93ull) ^ (rank << 40);
            key ^= key >> 29;
            if (key == 0ull) key = 1ull;
            h->shuffle.key = key;
            h->shuffle.n = (unsigned long long)h->n;
            int bits = 1;
            while ((1ull << bits) < (unsigned long long)h->n) ++bits;
            h->shuffle.bits = bits;
        }
        h->gwsum_src = nullptr;
        h->global_bs = 0;
        if (global_batch != batch || global_wsum != nullptr) {
            h->gwsum.ensure((size_t)n_batches);
            if (global_wsum) {
                h->h_gwsum.ensure((size_t)n_batches);
                memcpy(h->h_gwsum.p, global_wsum, sizeof(float) * n_batches);
                VH_HIP(hipMemcpyAsync(h->gwsum.p, h->h_gwsum.p, sizeof(float) * n_batches, hipMemcpyHostToDevice,
                                      h->stream));
            } else {
                // plan on the device: local weight sum of every batch, all-reduced over the ranks
                hipLaunchKernelGGL(vae_batch_wsum_kernel, dim3((unsigned)n_batches), dim3(256), 0, h->stream, h->w.p,
                                   dev_idx, h->shuffle, (int)batch, h->gwsum.p);
                VH_HIP(hipGetLastError());
                if (h->comm) rccl_allreduce_sum_f32(h->comm, h->gwsum.p, (size_t)n_batches, h->stream);
            }
            h->gwsum_src = h->gwsum.p;
            h->global_bs = (int)global_batch;
        }
        lap("staged");
        reset_epoch_sums(h);
        reset_batch_index(h);
        run_epoch_steps(h, dev_idx, n_batches);
        lap("enqueued");
        h->gwsum_src = nullptr;
        h->global_bs = 0;
        h->shuffle.key = 0ull;
        if (dp) {
            // epoch log line: every rank holds local_sum / B_global, the sum over ranks is the global mean
            StepState* st = h->state.p;
            rccl_allreduce_sum_f64(h->comm, st->epoch_loss, 5, h->stream);
            if (h->NL > 0) rccl_allreduce_sum_f64(h->comm, st->epoch_label, 2, h->stream);
            // BatchNorm running statistics are per-rank (local batch statistics): average them so that
            // every rank encodes with the same eval-mode network
            rccl_allreduce_sum_f32(h->comm, h->bnbuf.p, h->bn_elems, h->stream);
            hipLaunchKernelGGL(vae_scale_kernel, dim3(64), dim3(256), 0, h->stream, h->bnbuf.p, (int64_t)h->bn_elems,
                               1.0f / (float)h->comm->world);
            VH_HIP(hipGetLastError());
        }
    }
}

}  // namespace

extern "C" {

int vh_vae_train_epoch_dp(vh_vae* h, const int64_t* perm, int64_t n_batches, int64_t batch, int64_t global_batch,
                          const float* global_wsum, double loss_means[5]) {
    return guarded([&] {
        enqueue_epoch(h, perm, n_batches, batch, global_batch, global_wsum);
        StepState st;
        read_state(h, &st);
        probe_collect(h);
        if (loss_means)
            for (int i = 0; i < 5; ++i) loss_means[i] = st.epoch_loss[i] / (double)n_batches;
        h->label_stats.assign({st.epoch_label[0] / (double)n_batches, st.epoch_label[1]});
    });
}

// Several epochs of the same shape (device-side shuffle) with ONE host synchronisation at the end: each epoch's
// loss sums are copied into pinned memory in stream order, so the GPU never idles at an epoch boundary
// (the per-epoch read-back cost ~0.4 ms of 17.4 at C1).
int vh_vae_train_epochs(vh_vae* h, int64_t n_epochs, int64_t n_batches, int64_t batch, int64_t global_batch,
                        double* loss_means /* [n_epochs][5] */) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && loss_means != nullptr, "NULL argument");
        VH_REQUIRE(n_epochs >= 1, "no epochs");
        h->h_epoch_loss.ensure((size_t)n_epochs * 7);   // per epoch: the five loss sums, then (label cross-entropy, correct)
        double* lab_sums = h->h_epoch_loss.p + 5 * n_epochs;
        for (int64_t e = 0; e < n_epochs; ++e) {
            enqueue_epoch(h, nullptr, n_batches, batch, global_batch, nullptr);
            VH_HIP(hipMemcpyAsync(h->h_epoch_loss.p + 5 * e, h->state.p->epoch_loss, 5 * sizeof(double),
                                  hipMemcpyDeviceToHost, h->stream));
            if (h->NL > 0)
                VH_HIP(hipMemcpyAsync(lab_sums + 2 * e, h->state.p->epoch_label, 2 * sizeof(double), hipMemcpyDeviceToHost,
                                      h->stream));
            probe_collect_ready(h);
        }
        VH_HIP(hipStreamSynchronize(h->stream));
        probe_collect(h);
        for (int64_t i = 0; i < n_epochs * 5; ++i) loss_means[i] = h->h_epoch_loss.p[i] / (double)n_batches;
        h->label_stats.clear();
        for (int64_t e = 0; e < n_epochs && h->NL > 0; ++e) {
            h->label_stats.push_back(lab_sums[2 * e] / (double)n_batches);
            h->label_stats.push_back(lab_sums[2 * e + 1]);
        }
    });
}

int vh_vae_train_epoch(vh_vae* h, const int64_t* perm, int64_t n_batches, int64_t batch, double loss_means[5]) {
    return vh_vae_train_epoch_dp(h, perm, n_batches, batch, 0, nullptr, loss_means);
}

int vh_vae_set_syncbn(vh_vae* h, int enable) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        if (h->syncbn != (enable != 0)) {
            VH_HIP(hipStreamSynchronize(h->stream));
            h->syncbn = enable != 0;
            h->bs = 0;
        }
    });
}

int vh_vae_attach_comm(vh_vae* h, vh_comm* comm) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        if (h->comm != comm) {
            VH_HIP(hipStreamSynchronize(h->stream));
            h->comm = comm;
            h->bs = 0;   // the optimiser tables depend on the communicator (SyncBN scaling of the gamma / beta gradients)
        }
    });
}

int vh_vae_forward(vh_vae* h, const float* depths, const float* tnf, const float* abundance, int64_t batch,
                   int training, const float* eps, const uint8_t* masks, float* depths_out, float* tnf_out,
                   float* abundance_out, float* mu_out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && depths && tnf && abundance, "NULL argument");
        VH_REQUIRE(h->kind == VH_VAE_PLAIN, "models with a label block go through vh_vae_forward_rows");
        VH_REQUIRE(batch >= 1 && batch <= (1 << 24), "bad batch size");
        VH_REQUIRE(!training || batch >= 2, "Expected more than 1 value per channel when training");
        prepare_batch(h, (int)batch);
        std::vector<float> xb((size_t)h->bs_p * h->D_p, 0.f);
        for (int64_t r = 0; r < batch; ++r) {
            float* dst = xb.data() + (size_t)r * h->D_p;
            memcpy(dst, depths + (size_t)r * h->S, sizeof(float) * h->S);
            memcpy(dst + h->S, tnf + (size_t)r * VH_NTNF, sizeof(float) * VH_NTNF);
            dst[h->S + VH_NTNF] = abundance[r];
        }
        VH_HIP(hipMemcpyAsync(h->Xb.p, xb.data(), sizeof(float) * xb.size(), hipMemcpyHostToDevice, h->stream));
        std::vector<float> e;
        if (eps) {
            e.assign((size_t)h->bs_p * h->L_p, 0.f);
            for (int r = 0; r < batch; ++r) memcpy(e.data() + (size_t)r * h->L_p, eps + (size_t)r * h->L, sizeof(float) * h->L);
            VH_HIP(hipMemcpyAsync(h->EPS.p, e.data(), sizeof(float) * e.size(), hipMemcpyHostToDevice, h->stream));
        }
        VH_HIP(hipStreamSynchronize(h->stream));
        const bool inj_masks = training && masks != nullptr && h->cfg.dropout > 0;
        if (inj_masks) upload_masks(h, masks, (int)batch);
        h->batch_from_gather = false;   // (an uploaded batch: no dataset rows behind it)
        if (h->bf16) {
            const int64_t n4 = (int64_t)h->bs_p * h->D_p / 4;
            hipLaunchKernelGGL(vae_cast16_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n4, 256), 4096)), dim3(256), 0,
                               h->stream, h->Xb.p, h->Xb16.p, n4);
            VH_HIP(hipGetLastError());
            step16::forward16(h, training != 0, eps != nullptr, inj_masks, true, nullptr);
        } else {
            forward(h, training != 0, eps != nullptr, inj_masks, true);
        }
        if (training) { join_side(h); count_batches(h, 1); }
        hipLaunchKernelGGL(vae_advance_step_kernel, dim3(1), dim3(1), 0, h->stream, h->state.p);
        VH_HIP(hipGetLastError());
        // outputs
        std::vector<float> r((size_t)batch * h->D_p);
        VH_HIP(hipMemcpyAsync(r.data(), h->R.p, sizeof(float) * r.size(), hipMemcpyDeviceToHost, h->stream));
        if (depths_out) {
            hipLaunchKernelGGL(vae_softmax_out_kernel, dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, h->stream,
                               h->R.p, (int64_t)h->D_p, (int)batch, h->S, h->out_sm.p);
            VH_HIP(hipGetLastError());
            VH_HIP(hipMemcpyAsync(depths_out, h->out_sm.p, sizeof(float) * (size_t)batch * h->S, hipMemcpyDeviceToHost,
                                  h->stream));
        }
        std::vector<float> mu;
        if (mu_out) {
            mu.resize((size_t)batch * h->L_p);
            VH_HIP(hipMemcpyAsync(mu.data(), h->MU.p, sizeof(float) * mu.size(), hipMemcpyDeviceToHost, h->stream));
        }
        VH_HIP(hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < batch; ++i) {
            if (tnf_out) memcpy(tnf_out + (size_t)i * VH_NTNF, r.data() + (size_t)i * h->D_p + h->S, sizeof(float) * VH_NTNF);
            if (abundance_out) abundance_out[i] = r[(size_t)i * h->D_p + h->S + VH_NTNF];
            if (mu_out) memcpy(mu_out + (size_t)i * h->L, mu.data() + (size_t)i * h->L_p, sizeof(float) * h->L);
        }
    });
}

int vh_vae_label_stats(vh_vae* h, int64_t n_epochs, double* out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && out != nullptr, "NULL argument");
        VH_REQUIRE(h->NL > 0, "the model has no label block");
        VH_REQUIRE((int64_t)h->label_stats.size() == 2 * n_epochs, "the last training call covered %lld epochs, not %lld",
                   (long long)(h->label_stats.size() / 2), (long long)n_epochs);
        memcpy(out, h->label_stats.data(), sizeof(double) * h->label_stats.size());
    });
}

int vh_vae_row_width(vh_vae* h, int32_t* width) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && width != nullptr, "NULL argument");
        *width = h->D;
    });
}

// forward() on explicit input rows in the model's own column order; R = the reconstruction with the softmax applied to the
// depths block (what `_decode` returns, encode.py:283-304 / semisupervised_encode.py:228-237, 480-502)
int vh_vae_forward_rows(vh_vae* h, const float* X, int64_t batch, int training, const float* eps, const uint8_t* masks,
                        float* R_out, float* mu_out) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && X != nullptr, "NULL argument");
        VH_REQUIRE(batch >= 1 && batch <= (1 << 24), "bad batch size");
        VH_REQUIRE(!training || batch >= 2, "Expected more than 1 value per channel when training");
        prepare_batch(h, (int)batch);
        std::vector<float> xb((size_t)h->bs_p * h->D_p, 0.f);
        for (int64_t r = 0; r < batch; ++r) memcpy(xb.data() + (size_t)r * h->D_p, X + (size_t)r * h->D, sizeof(float) * h->D);
        VH_HIP(hipMemcpyAsync(h->Xb.p, xb.data(), sizeof(float) * xb.size(), hipMemcpyHostToDevice, h->stream));
        std::vector<float> e;
        if (eps) {
            e.assign((size_t)h->bs_p * h->L_p, 0.f);
            for (int r = 0; r < batch; ++r) memcpy(e.data() + (size_t)r * h->L_p, eps + (size_t)r * h->L, sizeof(float) * h->L);
            VH_HIP(hipMemcpyAsync(h->EPS.p, e.data(), sizeof(float) * e.size(), hipMemcpyHostToDevice, h->stream));
        }
        VH_HIP(hipStreamSynchronize(h->stream));
        const bool inj_masks = training && masks != nullptr && h->cfg.dropout > 0;
        if (inj_masks) upload_masks(h, masks, (int)batch);
        h->batch_from_gather = false;   // (an uploaded batch: no dataset rows behind it)
        if (h->bf16) {
            const int64_t n4 = (int64_t)h->bs_p * h->D_p / 4;
            hipLaunchKernelGGL(vae_cast16_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n4, 256), 4096)), dim3(256), 0,
                               h->stream, h->Xb.p, h->Xb16.p, n4);
            VH_HIP(hipGetLastError());
            step16::forward16(h, training != 0, eps != nullptr, inj_masks, true, nullptr);
        } else {
            forward(h, training != 0, eps != nullptr, inj_masks, true);
        }
        if (training) { join_side(h); count_batches(h, 1); }
        hipLaunchKernelGGL(vae_advance_step_kernel, dim3(1), dim3(1), 0, h->stream, h->state.p);
        VH_HIP(hipGetLastError());
        std::vector<float> r((size_t)batch * h->D_p), sm, mu;
        VH_HIP(hipMemcpyAsync(r.data(), h->R.p, sizeof(float) * r.size(), hipMemcpyDeviceToHost, h->stream));
        if (R_out && h->S > 1) {
            sm.resize((size_t)batch * h->S);
            hipLaunchKernelGGL(vae_softmax_out_kernel, dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, h->stream,
                               h->R.p, (int64_t)h->D_p, (int)batch, h->S, h->out_sm.p);
            VH_HIP(hipGetLastError());
            VH_HIP(hipMemcpyAsync(sm.data(), h->out_sm.p, sizeof(float) * sm.size(), hipMemcpyDeviceToHost, h->stream));
        }
        if (mu_out) {
            mu.resize((size_t)batch * h->L_p);
            VH_HIP(hipMemcpyAsync(mu.data(), h->MU.p, sizeof(float) * mu.size(), hipMemcpyDeviceToHost, h->stream));
        }
        VH_HIP(hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < batch; ++i) {
            if (R_out) {
                memcpy(R_out + (size_t)i * h->D, r.data() + (size_t)i * h->D_p, sizeof(float) * h->D);
                if (!sm.empty()) memcpy(R_out + (size_t)i * h->D, sm.data() + (size_t)i * h->S, sizeof(float) * h->S);
            }
            if (mu_out) memcpy(mu_out + (size_t)i * h->L, mu.data() + (size_t)i * h->L_p, sizeof(float) * h->L);
        }
    });
}

int vh_vae_encode(vh_vae* h, float* latent) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && latent != nullptr, "NULL argument");
        VH_REQUIRE(h->n > 0, "no dataset: call vh_vae_set_dataset first");
        if (h->bf16) { step16::encode16(h, latent); return; }
        hipStream_t s = h->stream;
        const int64_t chunk = 16384;
        int maxw = 0;
        for (int li = 0; li < h->nl; ++li) maxw = std::max(maxw, h->hidden[li].nout_p);
        DevBuf<float> a0, a1, lat, xin;
        a0.alloc((size_t)chunk * maxw);
        a1.alloc((size_t)chunk * maxw);
        lat.alloc((size_t)chunk * h->L);
        if (h->kind != VH_VAE_PLAIN) xin.alloc((size_t)chunk * h->D_p);   // rows with their one-hot label block
        for (int li = 0; li < h->nl; ++li) {
            Hidden& hl = h->hidden[li];
            hipLaunchKernelGGL(vae_bn_eval_coeff_kernel, dim3((unsigned)ceil_div(hl.nout_p, 256)), dim3(256), 0, s,
                               hl.nout_p, h->pptr(hl.tG), h->pptr(hl.tB), h->pptr(hl.tRM), h->pptr(hl.tRV),
                               hl.scale.p, hl.shift.p);
            VH_HIP(hipGetLastError());
        }
        for (int64_t lo = 0; lo < h->n; lo += chunk) {
            const int m = (int)std::min<int64_t>(chunk, h->n - lo);
            const float* in = h->X.p + (size_t)lo * h->D_p;
            if (h->kind != VH_VAE_PLAIN) {
                hipLaunchKernelGGL(vae_gather_kernel<true>, dim3((unsigned)ceil_div(m, 4)), dim3(64, 4), 0, s, h->X.p, h->ld_src,
                                   (int64_t)h->D_p, h->w.p, (const int64_t*)nullptr, ShuffleSpec{0, 0, 1},
                                   (const long long*)nullptr, lo, m, m, xin.p, (float*)nullptr, LabelSrc{h->labels, h->lab0},
                                   (int32_t*)nullptr);
                VH_HIP(hipGetLastError());
                in = xin.p;
            }
            int in_w = h->D_p;
            float* bufs[2] = {a0.p, a1.p};
            for (int li = 0; li < h->nl; ++li) {
                Hidden& hl = h->hidden[li];
                GemmArgs g = base_args(h->bf16);
                g.A = in; g.lda = in_w;
                g.B = h->pptr(hl.tW); g.ldb = hl.nin_p;
                g.C = bufs[li & 1]; g.ldc = hl.nout_p;
                g.M = m; g.N = hl.nout_p; g.K = hl.nin_p; g.k_per_split = g.K;
                g.bias = h->pptr(hl.tb); g.scale = hl.scale.p; g.shift = hl.shift.p; g.m_real = m;
                gemm_tile<true, true, EPI_HIDDEN_EVAL>(s, fwd_tile(m, hl.nout_p), g, 1);
                in = bufs[li & 1];
                in_w = hl.nout_p;
            }
            GemmArgs g = base_args(h->bf16);
            g.A = in; g.lda = in_w;
            g.B = h->pptr(h->tWmu); g.ldb = in_w;
            g.C = lat.p; g.ldc = h->L;          // compact [m][L]: only the logical columns are stored
            g.M = m; g.N = h->L; g.K = in_w; g.k_per_split = g.K;
            g.bias = h->pptr(h->tbmu);
            gemm_tile<true, true, EPI_LATENT_MASK>(s, fwd_tile(m, h->L_p), g, 1);
            VH_HIP(hipMemcpyAsync(latent + (size_t)lo * h->L, lat.p, sizeof(float) * (size_t)m * h->L,
                                  hipMemcpyDeviceToHost, s));
            VH_HIP(hipStreamSynchronize(s));
        }
    });
}

int vh_vae_opt_state(vh_vae* h, double* d, double* numerator_weighted, int64_t* k) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        StepState st;
        read_state(h, &st);
        if (d) *d = st.d;
        if (numerator_weighted) *numerator_weighted = st.numerator_weighted;
        if (k) *k = st.k;
    });
}

int vh_vae_opt_set_state(vh_vae* h, double d, double numerator_weighted, int64_t k) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        VH_REQUIRE(d > 0 && k >= 0, "d must be positive and k non-negative");
        StepState st;
        read_state(h, &st);
        st.d = d;
        st.numerator_weighted = numerator_weighted;
        st.k = k;
        VH_HIP(hipMemcpyAsync(h->state.p, &st, offsetof(StepState, step), hipMemcpyHostToDevice, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
    });
}

int vh_vae_reset_optimizer(vh_vae* h) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        VH_HIP(hipMemsetAsync(h->M1.p, 0, h->M1.bytes(), h->stream));
        VH_HIP(hipMemsetAsync(h->M2.p, 0, h->M2.bytes(), h->stream));
        VH_HIP(hipMemsetAsync(h->Sv.p, 0, h->Sv.bytes(), h->stream));
        StepState st;
        read_state(h, &st);
        st.d = 1e-6;
        st.numerator_weighted = 0.0;
        st.k = 0;
        VH_HIP(hipMemcpyAsync(h->state.p, &st, offsetof(StepState, step), hipMemcpyHostToDevice, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
    });
}

int vh_vae_set_optimizer(vh_vae* h, int optimizer, float lrate) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        VH_REQUIRE(optimizer == VH_OPT_DADAPT_ADAM || optimizer == VH_OPT_ADAM, "unknown optimizer %d", optimizer);
        VH_REQUIRE(optimizer == VH_OPT_DADAPT_ADAM || lrate > 0, "Learning rate must be positive, not %g", (double)lrate);
        VH_HIP(hipStreamSynchronize(h->stream));
        h->adam_lr = optimizer == VH_OPT_ADAM ? lrate : 0.0f;
    });
}

namespace {
float* moment_ptr(vh_vae* h, int which) {
    VH_REQUIRE(which >= 0 && which <= 2, "which: 0 exp_avg, 1 exp_avg_sq, 2 s");
    return which == 0 ? h->M1.p : (which == 1 ? h->M2.p : h->Sv.p);
}
}  // namespace

int vh_vae_get_opt_moment(vh_vae* h, const char* name, int which, float* data, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && data != nullptr, "NULL argument");
        const int ti = find_tensor(h, name);
        const Tensor& t = h->tensors[ti];
        VH_REQUIRE(t.optimised, "'%s' is a buffer, not a parameter", name);
        VH_REQUIRE(n == t.logical(), "parameter '%s' has %lld elements, got %lld", name, (long long)t.logical(), (long long)n);
        download_padded(h, t, moment_ptr(h, which) + t.off, data);
    });
}

int vh_vae_set_opt_moment(vh_vae* h, const char* name, int which, const float* data, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && data != nullptr, "NULL argument");
        const int ti = find_tensor(h, name);
        const Tensor& t = h->tensors[ti];
        VH_REQUIRE(t.optimised, "'%s' is a buffer, not a parameter", name);
        VH_REQUIRE(n == t.logical(), "parameter '%s' has %lld elements, got %lld", name, (long long)t.logical(), (long long)n);
        std::vector<float> buf((size_t)t.slot, 0.0f);
        for (int r = 0; r < t.rows; ++r) memcpy(buf.data() + (size_t)r * t.cols_p, data + (size_t)r * t.cols, sizeof(float) * t.cols);
        VH_HIP(hipMemcpyAsync(moment_ptr(h, which) + t.off, buf.data(), sizeof(float) * t.slot, hipMemcpyHostToDevice, h->stream));
        VH_HIP(hipStreamSynchronize(h->stream));
    });
}

int vh_vae_set_precision(vh_vae* h, int bf16_operands) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL handle");
        const bool on = bf16_operands != 0;
        if (on == h->bf16) return;
        VH_REQUIRE(!on || h->n_leaves == 0, "the hierarchical label loss is implemented by the fp32 step");
        // the bf16 step's loss kernel stages 4 x (reconstruction row + target row) in LDS: checked HERE (VH_ERR_INVALID), not at
        // the first training step -- VAEConcat / VAELabels with a few thousand classes reach this width (step16::loss_and_seed16)
        VH_REQUIRE(!on || (size_t)8 * h->D_p * sizeof(float) <= 160 * 1024 - 256,
                   "%d input columns are too wide for the bf16 step (its loss kernel holds 8 rows in LDS: at most 5112 columns); "
                   "use the fp32 step", h->D);
        VH_HIP(hipStreamSynchronize(h->stream));
        h->bf16 = on;
        h->bs = 0;   // the per-batch plan (gradient slabs, optimiser table, workspaces) depends on the mode
        if (on) {
            h->W16.ensure(h->flat_elems);
            h->W16T.ensure(h->flat_elems);
            h->zeros16.ensure(128);
            VH_HIP(hipMemsetAsync(h->W16.p, 0, h->W16.bytes(), h->stream));
            VH_HIP(hipMemsetAsync(h->W16T.p, 0, h->W16T.bytes(), h->stream));
            VH_HIP(hipMemsetAsync(h->zeros16.p, 0, h->zeros16.bytes(), h->stream));
            step16::refresh_shadows(h, -1);
            VH_HIP(hipStreamSynchronize(h->stream));
        }
    });
}

// taxvamb_encode.py:326-330 / 474-478: Hierarchy(table_parent) + FlatSoftmaxNLL(tree).  table_parent[0] = -1 (the root),
// 0 <= table_parent[i] < i (taxvamb_encode.py:49-61 emits nodes in BFS order); leaves = nodes nobody names as a parent, in
// node order (hloss_misc.py:51-58); leaf_masks[i][j] = leaf j is node i or below it (hloss_misc.py:98-115, 1110-1113).
int vh_vae_set_hierarchy(vh_vae* h, const int32_t* table_parent, int32_t n_nodes) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr && table_parent != nullptr, "NULL argument");
        VH_REQUIRE(h->kind != VH_VAE_PLAIN, "the model has no label block");
        VH_REQUIRE(!h->bf16, "the hierarchical label loss is implemented by the fp32 step");
        VH_REQUIRE(n_nodes >= 1 && n_nodes <= h->NL, "%d nodes do not fit the model's %d label columns", n_nodes, h->NL);
        VH_REQUIRE(table_parent[0] == -1, "node 0 must be the root (parent -1), not a child of %d", table_parent[0]);
        std::vector<int> nchild((size_t)n_nodes, 0);
        for (int i = 1; i < n_nodes; ++i) {
            VH_REQUIRE(table_parent[i] >= 0 && table_parent[i] < i, "parent %d of node %d: parents must precede their children",
                       table_parent[i], i);
            nchild[table_parent[i]]++;
        }
        std::vector<int> leaf_of((size_t)n_nodes, -1);
        int n_leaves = 0;
        for (int i = 0; i < n_nodes; ++i)
            if (nchild[i] == 0) leaf_of[i] = n_leaves++;
        std::vector<uint8_t> mask((size_t)n_nodes * n_leaves, 0);
        for (int j = 0; j < n_nodes; ++j) {
            if (leaf_of[j] < 0) continue;
            for (int i = j; i >= 0; i = table_parent[i]) mask[(size_t)i * n_leaves + leaf_of[j]] = 1;   // j itself and every ancestor
        }
        VH_HIP(hipStreamSynchronize(h->stream));
        h->leaf_masks.alloc(mask.size());
        VH_HIP(hipMemcpy(h->leaf_masks.p, mask.data(), mask.size(), hipMemcpyHostToDevice));
        h->n_nodes = n_nodes;
        h->n_leaves = n_leaves;
    });
}

int vh_vae_set_probe(vh_vae* h, int enable, int layer) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        VH_REQUIRE(layer >= 0 && layer < 2 * h->nl, "layer out of range");
        h->probe_on = enable != 0;
        h->probe_layer = layer;
        h->probe_ms = 0.0;
        h->probe_launches = 0;
        h->probe_used = 0;
        h->probe_head = 0;
        h->probe_calls = 0;
        h->probe_every = (int)std::max<int64_t>(1, option("vae.probe_every", 16));
    });
}

int vh_vae_probe_result(vh_vae* h, double* ms_total, int64_t* launches, double* flops_per_launch) {
    return guarded([&] {
        VH_REQUIRE(h != nullptr, "NULL argument");
        if (ms_total) *ms_total = h->probe_ms;
        if (launches) *launches = h->probe_launches;
        if (flops_per_launch) *flops_per_launch = h->probe_flops;
    });
}

int vh_debug_gemm(int tile, int a_kc, int b_kc, const float* A, const float* B, const float* bias, float* C, int M,
                  int N, int K, int splits, float* ms) {
    return guarded([&] {
        VH_REQUIRE(A && B && C, "NULL argument");
        const bool use_bf16 = tile >= 100;   // tile + 100: the bf16-operand instantiation of that tile
        if (use_bf16) tile -= 100;
        VH_REQUIRE(tile >= 0 && tile <= 7 && !((tile == 4 || tile >= 6) && use_bf16), "tile in {0..7} (+100 for bf16 operands, not tiles 4, 6, 7)");
        VH_REQUIRE(tile != 6 || ((K / std::max(1, splits)) % 128 == 0 && K % std::max(1, splits) == 0), "tile 6 needs K / splits multiple of 128");
        VH_REQUIRE(tile != 7 || ((K / std::max(1, splits)) % 512 == 0 && K % std::max(1, splits) == 0), "tile 7 needs K / splits multiple of 512");
        VH_REQUIRE(tile != 4 || (K % 64 == 0 && (K / std::max(1, splits)) % 64 == 0), "tile 4 needs K multiple of 64");
        VH_REQUIRE(M >= 1 && N >= 1 && K >= 32 && K % 32 == 0 && M % 4 == 0 && N % 4 == 0,
                   "need K multiple of 32 and M, N multiples of 4");
        VH_REQUIRE(splits >= 1 && (K / 32) >= splits, "bad split count");
        VH_REQUIRE(!(bias && (splits > 1 || !a_kc || !b_kc)), "bias only with K-contiguous operands and one split");
        VH_REQUIRE(!(a_kc == 0 && b_kc == 1), "layout (row-contiguous A, K-contiguous B) is not instantiated");
        hipStream_t s;
        VH_HIP(hipStreamCreate(&s));
        DevBuf<float> dA, dB, dC, dbias;
        dA.alloc((size_t)M * K); dB.alloc((size_t)N * K);
        const int k_per = (int)round_up(ceil_div(K, splits), 32);
        const int nsplit = (int)ceil_div(K, k_per);
        dC.alloc((size_t)nsplit * M * N);
        VH_HIP(hipMemcpy(dA.p, A, sizeof(float) * (size_t)M * K, hipMemcpyHostToDevice));
        VH_HIP(hipMemcpy(dB.p, B, sizeof(float) * (size_t)N * K, hipMemcpyHostToDevice));
        if (bias) { dbias.alloc(N); VH_HIP(hipMemcpy(dbias.p, bias, sizeof(float) * N, hipMemcpyHostToDevice)); }
        GemmArgs g = base_args(use_bf16);
        g.A = dA.p; g.lda = a_kc ? K : M;
        g.B = dB.p; g.ldb = b_kc ? K : N;
        g.C = dC.p; g.ldc = N;
        g.M = M; g.N = N; g.K = K; g.k_per_split = k_per; g.slab_stride = (int64_t)M * N;
        g.bias = dbias.p; g.m_real = M;
        hipEvent_t e0, e1;
        VH_HIP(hipEventCreate(&e0));
        VH_HIP(hipEventCreate(&e1));
        auto run = [&] {
            if (a_kc && b_kc) {
                if (bias) gemm_tile_debug<true, true, EPI_BIAS>(s, tile, g, 1);
                else gemm_tile_debug<true, true, EPI_SPLITK>(s, tile, g, nsplit);
            } else if (a_kc && !b_kc) {
                gemm_tile_debug<true, false, EPI_SPLITK>(s, tile, g, nsplit);
            } else {
                gemm_tile_debug<false, false, EPI_SPLITK>(s, tile, g, nsplit);
            }
        };
        run();  // warm-up (also sets the LDS attribute)
        const int reps = std::max(1, (int)option("debug.gemm_reps", 1));   // back-to-back launches timed together (ms per launch)
        VH_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) run();
        VH_HIP(hipEventRecord(e1, s));
        VH_HIP(hipStreamSynchronize(s));
        float t = 0.f;
        VH_HIP(hipEventElapsedTime(&t, e0, e1));
        if (ms) *ms = t / (float)reps;
        std::vector<float> hc((size_t)nsplit * M * N);
        VH_HIP(hipMemcpy(hc.data(), dC.p, sizeof(float) * hc.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            float acc = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) acc += hc[(size_t)sp * M * N + i];
            C[i] = acc;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(s);
    });
}

int vh_debug_gemm16_timeline(int epi, int M, int N, int K, int variant, unsigned long long* stamps, int cap_blocks,
                             int* n_blocks, float* ms) {
    return guarded([&] {
        VH_REQUIRE(stamps && n_blocks, "NULL argument");
        VH_REQUIRE(epi == E16_SPLITK || epi == E16_BIAS || epi == E16_HIDDEN_TRAIN, "epi in {0 split-K, 1 bias, 3 hidden}");
        VH_REQUIRE(M >= 8 && N >= 8 && K >= 8 && K % 8 == 0 && N % 8 == 0 && M % 8 == 0, "need M, N, K multiples of 8");
        hipStream_t s;
        VH_HIP(hipStreamCreate(&s));
        DevBuf<bf16_t> dA, dB, dC16, dz;
        DevBuf<float> dC32, dbias;
        DevBuf<double> dstat;
        DevBuf<unsigned long long> dts;
        std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
        uint32_t x = 12345u;
        auto rnd = [&] { x = x * 1664525u + 1013904223u; return (bf16_t)(0x3C00u + ((x >> 9) & 0x3FFu) + ((x >> 3) & 0x8000u)); };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd();
        dA.alloc(hA.size()); dB.alloc(hB.size()); dz.alloc(128);
        VH_HIP(hipMemcpy(dA.p, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        VH_HIP(hipMemcpy(dB.p, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        VH_HIP(hipMemset(dz.p, 0, dz.bytes()));
        dC32.alloc((size_t)M * N); dC16.alloc((size_t)M * N); dstat.alloc((size_t)2 * N); dbias.alloc(N);
        VH_HIP(hipMemset(dbias.p, 0, dbias.bytes()));
        VH_HIP(hipMemset(dstat.p, 0, dstat.bytes()));
        const int max_blocks = (int)(ceil_div(M, 32) * ceil_div(N, 32));
        dts.alloc((size_t)max_blocks * 8);
        VH_HIP(hipMemset(dts.p, 0, dts.bytes()));
        Gemm16Args g;
        memset(&g, 0, sizeof(g));
        g.A = dA.p; g.lda = K; g.B = dB.p; g.ldb = K; g.M = M; g.N = N; g.K = K;
        g.k_per_split = K; g.slab_stride = (int64_t)M * N; g.zeros = dz.p;
        g.C32 = dC32.p; g.ldc32 = N; g.C16 = dC16.p; g.ldc16 = N;
        g.bias = dbias.p; g.m_real = M; g.fstat_out = dstat.p; g.drop_scale = 1.25f; g.drop_thresh = 858993459u; g.drop_key = 77;
        g.xcd_remap = 1;
        const int tile = variant & 0xFF;
        auto run = [&] {
            if (epi == E16_SPLITK) step16::gemm16_variant<E16_SPLITK>(s, tile, g, 1);
            else if (epi == E16_BIAS) step16::gemm16_variant<E16_BIAS>(s, tile, g, 1);
            else step16::gemm16_variant<E16_HIDDEN_TRAIN>(s, tile, g, 1);
        };
        for (int i = 0; i < 3; ++i) run();   // warm-up: attributes, caches
        hipEvent_t e0, e1;
        VH_HIP(hipEventCreate(&e0));
        VH_HIP(hipEventCreate(&e1));
        VH_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < 20; ++r) run();
        VH_HIP(hipEventRecord(e1, s));
        VH_HIP(hipStreamSynchronize(s));
        float t = 0.f;
        VH_HIP(hipEventElapsedTime(&t, e0, e1));
        if (ms) *ms = t / 20.0f;
        g.tstamps = dts.p;
        run();
        VH_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)max_blocks * 8);
        VH_HIP(hipMemcpy(h.data(), dts.p, h.size() * 8, hipMemcpyDeviceToHost));
        int nb = 0;
        for (int b = 0; b < max_blocks; ++b)
            if (h[(size_t)b * 8] != 0) nb = b + 1;
        *n_blocks = nb;
        for (int b = 0; b < std::min(nb, cap_blocks); ++b)
            for (int k = 0; k < 8; ++k) stamps[(size_t)b * 8 + k] = h[(size_t)b * 8 + k];
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(s);
    });
}

int vh_debug_gemm16_tn(const float* A, const float* B, float* C, double* colsum, int M, int N, int K, int k_real, int splits,
                       int reps, int tile, int pipeline, float* ms) {
    return guarded([&] {
        VH_REQUIRE(A && B && C, "NULL argument");
        VH_REQUIRE(M >= 8 && N >= 8 && K >= 1 && M % 8 == 0 && N % 8 == 0, "need M, N multiples of 8");
        VH_REQUIRE(splits >= 1 && reps >= 1, "bad split / repetition count");
        VH_REQUIRE(tile == 0 || tile == 1 || tile == 3, "tile: 0 (by shape), 1 (128x128) or 3 (64x128)");
        VH_REQUIRE(pipeline == 0 || pipeline == 2, "pipeline: 0 or 2");
        auto to_bf16 = [](const float* src, size_t n) {
            std::vector<bf16_t> out(n);
            for (size_t i = 0; i < n; ++i) {
                uint32_t u;
                memcpy(&u, &src[i], 4);
                u += 0x7FFFu + ((u >> 16) & 1u);
                out[i] = (bf16_t)(u >> 16);
            }
            return out;
        };
        hipStream_t s;
        VH_HIP(hipStreamCreate(&s));
        DevBuf<bf16_t> dA, dB, dz;
        DevBuf<float> dC;
        DevBuf<double> dsum;
        const std::vector<bf16_t> hA = to_bf16(A, (size_t)K * M), hB = to_bf16(B, (size_t)K * N);
        dA.alloc(hA.size()); dB.alloc(hB.size()); dz.alloc(128);
        VH_HIP(hipMemcpy(dA.p, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        VH_HIP(hipMemcpy(dB.p, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        VH_HIP(hipMemset(dz.p, 0, dz.bytes()));
        const int k_per = splits == 1 ? K : (int)round_up(ceil_div(K, splits), 64);
        const int nsplit = (int)ceil_div(K, k_per);
        dC.alloc((size_t)nsplit * M * N);
        dsum.alloc((size_t)M);
        Gemm16TnArgs g;
        memset(&g, 0, sizeof(g));
        g.A = dA.p; g.lda = M; g.B = dB.p; g.ldb = N; g.M = M; g.N = N; g.K = K; g.k_real = k_real;
        g.k_per_split = k_per; g.slab_stride = (int64_t)M * N; g.zeros = dz.p;
        g.C32 = dC.p; g.ldc = N; g.colsum = colsum ? dsum.p : nullptr; g.xcd_remap = 1;
        auto run = [&] {
            if (colsum) step16::gemm16_tn<1>(s, g, nsplit, tile, pipeline);
            else step16::gemm16_tn<0>(s, g, nsplit, tile, pipeline);
        };
        run();   // warm-up (sets the LDS attribute)
        VH_HIP(hipMemsetAsync(dsum.p, 0, dsum.bytes(), s));
        hipEvent_t e0, e1;
        VH_HIP(hipEventCreate(&e0));
        VH_HIP(hipEventCreate(&e1));
        VH_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) run();
        VH_HIP(hipEventRecord(e1, s));
        VH_HIP(hipStreamSynchronize(s));
        float t = 0.f;
        VH_HIP(hipEventElapsedTime(&t, e0, e1));
        if (ms) *ms = t / (float)reps;
        std::vector<float> hc((size_t)nsplit * M * N);
        VH_HIP(hipMemcpy(hc.data(), dC.p, sizeof(float) * hc.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            float acc = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) acc += hc[(size_t)sp * M * N + i];
            C[i] = acc;
        }
        if (colsum) {
            std::vector<double> hs((size_t)M);
            VH_HIP(hipMemcpy(hs.data(), dsum.p, hs.size() * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < hs.size(); ++i) colsum[i] = hs[i] / (double)reps;   // accumulated once per timed launch
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(s);
    });
}

int vh_debug_gemm16(int epi, const float* A, const float* B, const float* bias, float* C, float* CT, double* stats, int M,
                    int N, int K, int splits, int reps, int variant, float* ms) {
    return guarded([&] {
        VH_REQUIRE(A && B && C, "NULL argument");
        VH_REQUIRE(epi == E16_SPLITK || epi == E16_BIAS || epi == E16_HIDDEN_TRAIN, "epi in {0 split-K, 1 bias, 3 hidden}");
        VH_REQUIRE(M >= 1 && N >= 1 && K >= 8 && K % 8 == 0 && N % 8 == 0 && M % 8 == 0, "need M, N, K multiples of 8");
        VH_REQUIRE(splits >= 1 && reps >= 1, "bad split / repetition count");
        VH_REQUIRE(epi == E16_SPLITK || splits == 1, "only the split-K epilogue takes splits > 1");
        VH_REQUIRE(epi == E16_SPLITK || bias != nullptr, "bias is NULL");
        auto to_bf16 = [](const float* src, size_t n) {
            std::vector<bf16_t> out(n);
            for (size_t i = 0; i < n; ++i) {
                uint32_t u;
                memcpy(&u, &src[i], 4);
                u += 0x7FFFu + ((u >> 16) & 1u);
                out[i] = (bf16_t)(u >> 16);
            }
            return out;
        };
        hipStream_t s;
        VH_HIP(hipStreamCreate(&s));
        DevBuf<bf16_t> dA, dB, dC16, dC16T, dz;
        DevBuf<float> dC32, dbias;
        DevBuf<double> dstat;
        const std::vector<bf16_t> hA = to_bf16(A, (size_t)M * K), hB = to_bf16(B, (size_t)N * K);
        dA.alloc(hA.size()); dB.alloc(hB.size()); dz.alloc(128);
        VH_HIP(hipMemcpy(dA.p, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        VH_HIP(hipMemcpy(dB.p, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        VH_HIP(hipMemset(dz.p, 0, dz.bytes()));
        const int k_per = splits == 1 ? K : (int)round_up(ceil_div(K, splits), 64);
        const int nsplit = (int)ceil_div(K, k_per);
        dC32.alloc((size_t)nsplit * M * N);
        dC16.alloc((size_t)M * N); dC16T.alloc((size_t)M * N);
        dstat.alloc((size_t)2 * N);
        if (bias) { dbias.alloc(N); VH_HIP(hipMemcpy(dbias.p, bias, sizeof(float) * N, hipMemcpyHostToDevice)); }
        Gemm16Args g;
        memset(&g, 0, sizeof(g));
        g.A = dA.p; g.lda = K; g.B = dB.p; g.ldb = K; g.M = M; g.N = N; g.K = K;
        g.k_per_split = k_per; g.slab_stride = (int64_t)M * N; g.zeros = dz.p;
        g.C32 = dC32.p; g.ldc32 = N; g.C16 = dC16.p; g.ldc16 = N; g.C16T = CT ? dC16T.p : nullptr; g.ldc16t = M;   // without CT: the lean epilogue
        g.bias = dbias.p; g.m_real = M; g.fstat_out = dstat.p; g.drop_scale = 1.0f; g.xcd_remap = 1;
        const int tile = variant & 0xFF;
        g.dbg = variant >> 8;
        if (g.dbg & 32) {   // timing of the training epilogue as the step runs it: hashed dropout, p = 0.2 (values then differ from the host's)
            g.drop_scale = 1.25f; g.drop_thresh = (uint32_t)(0.2 * 4294967296.0); g.drop_key = 0x1234567ull;
            g.dbg &= ~32;
        }
        VH_REQUIRE(tile == 0 || tile == 1 || tile == 3 || tile == 4 || tile == 7 || tile == 11 || tile == 13 || tile == 17 ||
                       tile == 21 || tile == 23 || tile == 24 || tile == 27 || tile == 28,
                   "variant: tile 0, 1, 3, 4, 7; 11, 13, 17 (register-staged); 21, 23, 24, 27, 28 (interleaved DMA) (+ 256 * timing-experiment flags)");
        auto run = [&] {
            if (epi == E16_SPLITK) step16::gemm16_variant<E16_SPLITK>(s, tile, g, nsplit);
            else if (epi == E16_BIAS) step16::gemm16_variant<E16_BIAS>(s, tile, g, 1);
            else step16::gemm16_variant<E16_HIDDEN_TRAIN>(s, tile, g, 1);
        };
        run();   // warm-up (sets the LDS attribute)
        VH_HIP(hipMemsetAsync(dstat.p, 0, dstat.bytes(), s));
        hipEvent_t e0, e1;
        VH_HIP(hipEventCreate(&e0));
        VH_HIP(hipEventCreate(&e1));
        VH_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) run();
        VH_HIP(hipEventRecord(e1, s));
        VH_HIP(hipStreamSynchronize(s));
        float t = 0.f;
        VH_HIP(hipEventElapsedTime(&t, e0, e1));
        if (ms) *ms = t / (float)reps;
        if (epi == E16_HIDDEN_TRAIN) {
            std::vector<bf16_t> hc((size_t)M * N), hct((size_t)M * N);
            VH_HIP(hipMemcpy(hc.data(), dC16.p, hc.size() * 2, hipMemcpyDeviceToHost));
            VH_HIP(hipMemcpy(hct.data(), dC16T.p, hct.size() * 2, hipMemcpyDeviceToHost));
            auto tof = [](bf16_t b) { const uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
            for (size_t i = 0; i < hc.size(); ++i) C[i] = tof(hc[i]);
            if (CT) for (size_t i = 0; i < hct.size(); ++i) CT[i] = tof(hct[i]);
            if (stats) {
                std::vector<double> hs((size_t)2 * N);
                VH_HIP(hipMemcpy(hs.data(), dstat.p, hs.size() * 8, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hs.size(); ++i) stats[i] = hs[i] / (double)reps;   // accumulated once per timed launch
            }
        } else {
            std::vector<float> hc((size_t)nsplit * M * N);
            VH_HIP(hipMemcpy(hc.data(), dC32.p, sizeof(float) * hc.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < (size_t)M * N; ++i) {
                float acc = 0.f;
                for (int sp = 0; sp < nsplit; ++sp) acc += hc[(size_t)sp * M * N + i];
                C[i] = acc;
            }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(s);
    });
}

}  // extern "C"

#include "vaevae.hpp"   // the joint TaxVamb trainer (vh_vaevae_*), built from the step above
