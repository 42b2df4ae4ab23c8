// vae_step16.hpp -- host side of the bf16-storage training / inference step (BASELINE configs C2-C4).
// Included by vae.hip inside its anonymous namespace, after the helpers it shares with the fp32 step
// (drop_cfg, layer_key, step_ptr, bn_src, launch_forking, fork_side, join_side, probe_arm).
//
// Tensors (per batch size, see prepare_batch16):
//   Xb16, Z16, dR16, dMU16   bf16, row-major [bs_p][w]
//   per hidden layer: H16 (post-dropout activations), DA16 (grad wrt the BatchNorm output), DZ16
//   W16 / W16T: bf16 shadows of every parameter tensor at the offsets of the flat fp32 buffer (transposed for
//   matrices); Wf16 / biasf: the BatchNorm-folded weights of the layers that consume normalised activations.
// Stream plan: the main stream carries gather -> forward -> loss -> dX chain -> optimiser; the side stream carries
// the transposes of the narrow tensors, the running statistics and every weight-gradient GEMM but the last.
#pragma once

namespace step16 {

constexpr int kDwTargetWgs = 256;   // workgroups wanted per weight-gradient GEMM (they share the chip with the dX chain)

template <int BM, int BN, int WM, int WN, int EPI, int STG = 0, int TAG = 0>
void launch_gemm16(hipStream_t stream, const Gemm16Args& g, int splits) {
    static bool attr_set = false;
    constexpr size_t smem = gemm16_smem_bytes<BM, BN, WM, WN, EPI, STG>();
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, EPI, STG, TAG>;
    if (!attr_set) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kMaxDynLds));
        attr_set = true;
    }
    static_assert(smem <= kMaxDynLds, "tile does not fit the LDS budget");
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)splits);
    if (t_probe_start) {
        hipExtLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, t_probe_start, t_probe_stop, 0, g);
        t_probe_start = t_probe_stop = nullptr;
    } else if (t_fork_stop) {
        hipExtLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, nullptr, t_fork_stop, 0, g);
        t_fork_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, g);
    }
    VH_HIP(hipGetLastError());
}

template <int BM, int BN, int WM, int WN, int COLSUM, int STG>
void launch_gemm16_tn(hipStream_t stream, const Gemm16TnArgs& g, int splits) {
    static bool attr_set = false;
    constexpr size_t smem = gemm16_tn_smem_bytes<BM, BN, STG>();
    auto kern = gemm_bf16_tn_kernel<BM, BN, WM, WN, COLSUM, STG>;
    if (!attr_set) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kMaxDynLds));
        attr_set = true;
    }
    static_assert(smem <= kMaxDynLds, "tile does not fit the LDS budget");
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)splits);
    if (t_probe_start) {
        hipExtLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, t_probe_start, t_probe_stop, 0, g);
        t_probe_start = t_probe_stop = nullptr;
    } else if (t_fork_stop) {
        hipExtLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, nullptr, t_fork_stop, 0, g);
        t_fork_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, stream, g);
    }
    VH_HIP(hipGetLastError());
}

// latent-wide product with its elementwise consumer in one launch (gemm_skinny16.hpp); false: the shape does not fit one
// workgroup's LDS and the caller keeps the split-K launch + slab kernel
template <int EPI>
bool launch_skinny16(hipStream_t stream, const Skinny16Args& a, int N) {
    const size_t smem = skinny16_smem_bytes(N, a.k_per_wave, a.nslab);
    if (smem == 0 || (a.M & 31) != 0) return false;
    const int waves = std::max(kSkinnyMinWaves, a.nslab);
    auto go = [&](auto kern) {
        static bool attr_set = false;
        if (!attr_set) {
            VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(a.M / 32)), dim3(64 * waves), smem, stream, a);
        VH_HIP(hipGetLastError());
    };
    if (N == 32) go(gemm_skinny16_kernel<1, EPI>);
    else go(gemm_skinny16_kernel<2, EPI>);
    return true;
}

// weight-gradient tile by output shape (same rule as the K-contiguous kernel; dw_splits16 plans the slabs for it):
// tile 0 = by shape, 1 = 128x128 / 8 waves, 3 = 64x128 / 4 waves (A/B measurements through vh_debug_gemm16_tn)
template <int COLSUM, int STG>
void gemm16_tn_stg(hipStream_t s, const Gemm16TnArgs& g, int splits, int tile) {
    if (tile == 1) launch_gemm16_tn<128, 128, 2, 4, COLSUM, STG>(s, g, splits);
    else if (tile == 3) launch_gemm16_tn<64, 128, 2, 2, COLSUM, STG>(s, g, splits);
    else if (g.N <= 32) launch_gemm16_tn<128, 32, 4, 1, COLSUM, STG>(s, g, splits);
    else if (g.M <= 32) launch_gemm16_tn<32, 128, 1, 4, COLSUM, STG>(s, g, splits);
    else launch_gemm16_tn<64, 128, 2, 2, COLSUM, STG>(s, g, splits);
}
template <int COLSUM>
void gemm16_tn(hipStream_t s, const Gemm16TnArgs& g, int splits, int tile = 0, int pipeline = -1) {
    if (pipeline < 0) pipeline = 0;   // 4-wave tiles: the two-buffer loop (see gemm16)
    if (pipeline == 2) gemm16_tn_stg<COLSUM, 2>(s, g, splits, tile);
    else gemm16_tn_stg<COLSUM, 0>(s, g, splits, tile);
}

// Tile by output shape (measured, profiles/r02_gemm16_variants.json): batch-tall outputs take 128x128 tiles with 8 waves
// (2x4, two per SIMD: one wave's DMA issue and fragment reads overlap the other's MFMAs), weight gradients 64x128 tiles
// (twice the workgroups per split), 128x32 / 32x128 for latent-wide / latent-tall outputs.
template <int EPI, int STG>
void gemm16_stg(hipStream_t s, const Gemm16Args& g, int splits) {
    if constexpr (EPI == E16_SPLITK) {
        if (g.N <= 32) launch_gemm16<128, 32, 4, 1, EPI, STG>(s, g, splits);
        else if (g.M <= 32) launch_gemm16<32, 128, 1, 4, EPI, STG>(s, g, splits);
        else launch_gemm16<64, 128, 2, 2, EPI, STG>(s, g, splits);
    } else if constexpr (EPI == E16_LATENT_MASK) {
        launch_gemm16<128, 32, 4, 1, EPI, STG>(s, g, splits);
    } else {
        launch_gemm16<128, 128, 2, 4, EPI, STG>(s, g, splits);
    }
}
// The interleaved three-buffer loop pays on the 8-wave 128 x 128 tiles (two waves per SIMD: one wave's DMA issue hides under the
// other's MFMAs); on the 4-wave split-K tiles it measured slower (profiles/r03b_gemm16_variants.txt, r03c_step_timeline_*.txt:
// 64x128 weight gradients 16.8 -> 22.6 us, latent-wide 6.3 -> 10.8 us), so those stay on the two-buffer loop.
template <int EPI>
void gemm16(hipStream_t s, const Gemm16Args& g, int splits) {
    if constexpr (EPI == E16_SPLITK || EPI == E16_LATENT_MASK) {
        gemm16_stg<EPI, 0>(s, g, splits);
    } else {
        gemm16_stg<EPI, 2>(s, g, splits);
    }
}

// every tile / pipeline variant of one epilogue (vh_debug_gemm16): 0 = the production choice by output shape,
// 1 = 128x128 / 8 waves (2x4), 3 = 64x128 / 4 waves, 4 = 128x64 / 4 waves, 7 = 128x128 / 4 waves.  (Measured and
// removed, profiles/r02_gemm16_variants.json: 256x128 / 8 waves -- half the workgroups, 20-30 % slower at M = 8192,
// N = 512; three LDS buffers with counted vmcnt across a raw barrier -- 0-8 % slower: the K loop is bound by the
// per-CU LDS-DMA rate, ~17 B/clk, not by DMA latency.)
template <int EPI>
void gemm16_variant(hipStream_t s, int tile, const Gemm16Args& g, int splits) {
    switch (tile) {
        case 1: launch_gemm16<128, 128, 2, 4, EPI>(s, g, splits); break;
        case 3: launch_gemm16<64, 128, 2, 2, EPI>(s, g, splits); break;
        case 4: launch_gemm16<128, 64, 2, 2, EPI>(s, g, splits); break;
        case 7: launch_gemm16<128, 128, 2, 2, EPI>(s, g, splits); break;
        case 11: launch_gemm16<128, 128, 2, 4, EPI, 1>(s, g, splits); break;
        case 13: launch_gemm16<64, 128, 2, 2, EPI, 1>(s, g, splits); break;
        case 17: launch_gemm16<128, 128, 2, 2, EPI, 1>(s, g, splits); break;
        case 21: launch_gemm16<128, 128, 2, 4, EPI, 2>(s, g, splits); break;   // interleaved DMA, three buffers
        case 23: launch_gemm16<64, 128, 2, 2, EPI, 2>(s, g, splits); break;
        // round 6: tiles of 72 KB (three buffers) -- TWO workgroups per CU, one's prologue / epilogue under the
        // other's K loop.  8192 x 512 x 320 with the training epilogue: 9.82 us against 10.38 (variant 21), equal at K = 512, 22.2
        // against 18.6 us at K = 1120 (profiles/r06l_gemm16_twowg.txt); inside the step the K = D launch on tile 24 measured
        // 242.8 against 241.3 us per step (profiles/r06n_step_fold_narrow_c2.txt): not wired into the step
        case 24: launch_gemm16<128, 64, 2, 2, EPI, 2>(s, g, splits); break;   // 4 waves of 64 x 32
        case 28: launch_gemm16<128, 64, 4, 2, EPI, 2>(s, g, splits); break;   // 8 waves of 32 x 32 (16 waves per CU)
        case 27: launch_gemm16<128, 128, 2, 2, EPI, 2>(s, g, splits); break;
        default: gemm16<EPI>(s, g, splits); break;
    }
}

Gemm16Args args16(vh_vae* h) {
    Gemm16Args g;
    memset(&g, 0, sizeof(g));
    g.zeros = h->zeros16.p;
    g.drop_scale = 1.0f;
    g.xcd_remap = 1;
    return g;
}

bf16_t* w16(vh_vae* h, int t) { return h->W16.p + h->tensors[t].off; }
bf16_t* w16t(vh_vae* h, int t) { return h->W16T.p + h->tensors[t].off; }

// split-K plan of a weight-gradient GEMM C[M][N] over K = bs_p: slabs of >= 512 batch rows, ~kDwTargetWgs workgroups
int dw_splits16(int M, int N, int K) {
    const int bm = M <= 32 ? 32 : (N <= 32 ? 128 : 64), bn = N <= 32 ? 32 : 128;
    const int tiles = (int)(ceil_div(M, bm) * ceil_div(N, bn));
    int want = (int)std::max<int64_t>(1, ceil_div(kDwTargetWgs, tiles));
    want = std::min(want, std::max(1, K / 512));
    return std::max(1, want);
}

// Work that is off the critical path of a step (transposes of the narrow tensors, running statistics, the loss
// reduction, weight-gradient GEMMs) is queued in program order and handed to the side stream at a few fork points
// only: every cross-stream fork costs the main stream ~5 us (measured, profiles/r02_c_step_timeline.txt: one fork per
// producing kernel = 9 forks = 45 us of a 350 us step).  An item may be flushed at any fork that follows the launch of
// its last producer on the main stream.
// The two-stream schedule of a step (vae.fork_plan, bit mask: VaeTuning).  Unset, it follows the input width (round 6, with the
// paired launch of the last two weight gradients; profiles/r06f_step_forkplan2_c{2,3}.txt): up to 512 padded input columns (C2:
// D_p = 320) the two one-workgroup kernels -- running statistics, loss reduction -- run FIRST on the side stream (plan 2: 238.6
// against 251.0 us per step with them last), wider inputs (the C3 shape: D_p = 1120, a 3.5 x larger output-layer weight gradient
// in front of them) keep them LAST (plan 6: 341.5 against 350.8 us).
int fork_plan_for(const vh_vae* h) { return g_tuning.fork_plan >= 0 ? g_tuning.fork_plan : (h->D_p <= 512 ? 2 : 6); }

struct SideQueue {
    int plan = 6;
    std::vector<std::function<void(hipStream_t)>> items;
    std::vector<std::function<void(hipStream_t)>> tail;   // small items nobody but the optimiser waits for (vae.fork_plan & 4)
    void add(std::function<void(hipStream_t)> f) { items.push_back(std::move(f)); }
    void add_small(std::function<void(hipStream_t)> f) {
        if (plan & 4) tail.push_back(std::move(f));
        else items.push_back(std::move(f));
    }
    void flush(hipStream_t s) {
        for (auto& f : items) f(s);
        items.clear();
    }
    // behind everything else the side stream was given: the producers of these items precede every fork point of the step
    void flush_tail(hipStream_t s) {
        for (auto& f : tail) f(s);
        tail.clear();
    }
};

// bf16 shadows of one parameter tensor (or all of them) from the fp32 masters: init, set_param, precision switch
void refresh_shadows(vh_vae* h, int only) {
    for (int ti = 0; ti < (int)h->tensors.size(); ++ti) {
        const Tensor& t = h->tensors[ti];
        if (!t.optimised || !t.matrix || (only >= 0 && only != ti)) continue;
        const int64_t n = t.padded();
        hipLaunchKernelGGL(vae_shadow_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, h->stream, h->pptr(ti),
                           t.rows_p, t.cols_p, w16(h, ti), w16t(h, ti));
        VH_HIP(hipGetLastError());
    }
}

// everything of the bf16 step that depends on the batch size (called from prepare_batch)
void prepare_batch16(vh_vae* h) {
    const int bs_p = h->bs_p;
    h->Xb16.ensure((size_t)bs_p * h->D_p);
    h->Z16.ensure((size_t)bs_p * h->L_p);
    h->dR16.ensure((size_t)bs_p * h->D_p);
    h->dMU16.ensure((size_t)bs_p * h->L_p);
    for (auto& hl : h->hidden) {
        const size_t n = (size_t)bs_p * hl.nout_p;
        hl.H16.ensure(n); hl.DA16.ensure(n); hl.DZ16.ensure(n);
        hl.Wf16.ensure((size_t)hl.nout_p * hl.nin_p);
        hl.biasf.ensure((size_t)hl.nout_p);
    }
    h->Wf16_mu.ensure((size_t)h->L_p * h->hidden[h->nl - 1].nout_p);
    h->biasf_mu.ensure((size_t)h->L_p);
    h->Wf16_out.ensure((size_t)h->D_p * h->hidden[2 * h->nl - 1].nout_p);
    h->biasf_out.ensure((size_t)h->D_p);
}

// gradient sources of the bf16 step: split-K slabs for matrices (planned by prepare_batch), fp64 accumulators for
// every vector; BatchNorm completion for the weights that consume normalised activations
void build_opt16_table(vh_vae* h) {
    std::vector<Opt16Tensor> tab;
    int nblk = 0;
    const int nl = h->nl;
    auto add = [&](int ti, const Hidden* prev_bn, const double* dbias) {
        const Tensor& t = h->tensors[ti];
        Opt16Tensor d;
        memset(&d, 0, sizeof(d));
        d.rows_p = t.rows_p; d.cols_p = t.cols_p;
        d.p_off = (int64_t)t.off;
        d.blk_start = nblk;
        d.dscale = t.dsrc_allrank && h->comm && h->comm->world > 1 ? 1.0f / (float)h->comm->world : 1.0f;
        if (!t.matrix) {
            d.dsrc = t.dsrc;
            nblk += (int)ceil_div(t.cols_p, 1024);
        } else {
            d.slab = t.slab; d.nslab = t.nslab; d.stride = t.stride;
            d.w16 = w16(h, ti); d.w16t = w16t(h, ti);
            if (prev_bn) {
                d.bn_scale = prev_bn->scale.p;
                d.bn_shift = prev_bn->shift.p;
                d.dbias = dbias;
            }
            nblk += (t.rows_p / 32) * (t.cols_p / 32);
        }
        tab.push_back(d);
    };
    // creation order of the tensors (encoder hidden layers, mu, decoder hidden layers, output)
    for (int li = 0; li < 2 * nl; ++li) {
        if (li == nl) {
            add(h->tWmu, &h->hidden[nl - 1], h->dbias_mu);
            add(h->tbmu, nullptr, nullptr);
        }
        Hidden& hl = h->hidden[li];
        const bool from_input = (li == 0) || (li == nl);
        if (li == nl) {   // decoder-side tensors (the tail of the flat buffer): gradient bucket A
            h->opt16_bucketA_blk0 = nblk;
            h->opt16_bucketA_off = h->tensors[hl.tW].off;
        }
        add(hl.tW, from_input ? nullptr : &h->hidden[li - 1], hl.dbias);
        add(hl.tb, nullptr, nullptr);
        add(hl.tG, nullptr, nullptr);
        add(hl.tB, nullptr, nullptr);
    }
    add(h->tWo, &h->hidden[2 * nl - 1], h->dbias_out);
    add(h->tbo, nullptr, nullptr);
    VH_REQUIRE((int)tab.size() <= kMaxOpt16, "too many parameter tensors");
    h->opt16_n = (int)tab.size();
    h->opt16_blocks = nblk;
    h->opt16_tab.ensure(tab.size());
    VH_HIP(hipMemcpy(h->opt16_tab.p, tab.data(), tab.size() * sizeof(Opt16Tensor), hipMemcpyHostToDevice));
    {   // workgroup -> tensor (one load instead of a walk over the table)
        std::vector<uint8_t> map((size_t)nblk);
        for (size_t t = 0; t < tab.size(); ++t) {
            const int end = t + 1 < tab.size() ? tab[t + 1].blk_start : nblk;
            for (int b = tab[t].blk_start; b < end; ++b) map[(size_t)b] = (uint8_t)t;
        }
        h->opt16_blk2t.ensure(map.size());
        VH_HIP(hipMemcpy(h->opt16_blk2t.p, map.data(), map.size(), hipMemcpyHostToDevice));
    }
    // data-parallel variant: gradients already complete (and all-reduced) in the flat buffer G
    for (auto& d : tab) {
        d.dsrc = nullptr;
        d.slab = h->G.p + d.p_off; d.nslab = 1; d.stride = 0;
        d.bn_scale = nullptr; d.bn_shift = nullptr; d.dbias = nullptr;
    }
    h->opt16_tab_flat.ensure(tab.size());
    VH_HIP(hipMemcpy(h->opt16_tab_flat.p, tab.data(), tab.size() * sizeof(Opt16Tensor), hipMemcpyHostToDevice));
    h->opt_part.ensure((size_t)std::max(h->opt_blocks, nblk) * 2);
    if (h->opt_ticket.n == 0) {
        h->opt_ticket.alloc((size_t)kOptTicketStride * (1 + kOptTicketGroups));
        VH_HIP(hipMemset(h->opt_ticket.p, 0, h->opt_ticket.bytes()));
    }
}

void fold_bn(vh_vae* h, hipStream_t s, int tW, int tb, int n_rows, int K, const Hidden& prev, bool training,
             bf16_t* Wf16, float* biasf) {
    BnSrc bn = bn_src(h, prev);
    hipLaunchKernelGGL(vae_fold_bn_kernel, dim3((unsigned)ceil_div(n_rows, 4 * kFoldRowsPerWave)), dim3(256), (size_t)2 * K * sizeof(float), s,
                       h->pptr(tW), (int64_t)K, n_rows, K, h->pptr(tb), bn, training ? nullptr : prev.scale.p,
                       training ? nullptr : prev.shift.p, Wf16, biasf, training ? prev.scale.p : nullptr,
                       training ? prev.shift.p : nullptr, training ? prev.mean.p : nullptr, training ? prev.invstd.p : nullptr);
    VH_HIP(hipGetLastError());
}

// Xb / Xb16 / Wb must hold the batch.
void forward16(vh_vae* h, bool training, bool eps_injected, bool masks_injected, bool add_noise, SideQueue* defer) {
    const int bs = h->bs, bs_p = h->bs_p, nl = h->nl;
    hipStream_t s = h->stream;
    const DropCfg dc = drop_cfg(h, training, masks_injected);
    if (training) {
        if (!h->stat_clean) VH_HIP(hipMemsetAsync(h->statbuf.p, 0, h->statbuf.bytes(), s));
        h->stat_clean = false;
    } else {
        for (int li = 0; li < 2 * nl; ++li) {
            Hidden& hl = h->hidden[li];
            hipLaunchKernelGGL(vae_bn_eval_coeff_kernel, dim3((unsigned)ceil_div(hl.nout_p, 256)), dim3(256), 0, s,
                               hl.nout_p, h->pptr(hl.tG), h->pptr(hl.tB), h->pptr(hl.tRM), h->pptr(hl.tRV),
                               hl.scale.p, hl.shift.p);
            VH_HIP(hipGetLastError());
        }
    }
    const bf16_t* in = h->Xb16.p;
    int in_w = h->D_p;
    const Hidden* prev = nullptr;   // training: the layer whose BatchNorm still has to be folded into the consumer
    auto hidden_layer = [&](int li) {
        Hidden& hl = h->hidden[li];
        Gemm16Args g = args16(h);
        g.A = in; g.lda = in_w;
        g.M = bs_p; g.N = hl.nout_p; g.K = hl.nin_p; g.k_per_split = g.K;
        g.m_real = bs;
        g.C16 = hl.H16.p; g.ldc16 = hl.nout_p;
        if (training) {
            if (prev) {
                fold_bn(h, s, hl.tW, hl.tb, hl.nout_p, hl.nin_p, *prev, true, hl.Wf16.p, hl.biasf.p);
                g.B = hl.Wf16.p; g.bias = hl.biasf.p;
            } else {
                g.B = w16(h, hl.tW); g.bias = h->pptr(hl.tb);
            }
            g.ldb = hl.nin_p;
            g.fstat_out = hl.fstat;
            g.drop_scale = dc.scale; g.drop_thresh = dc.thresh; g.drop_key = layer_key(h, li);
            g.step_ptr = step_ptr(h);
            g.drop_mask = dc.injected ? hl.mask.p : nullptr; g.ld_mask = hl.nout_p;
            if (h->probe_on && li == h->probe_layer) { probe_arm(h); h->probe_flops = 2.0 * bs * (double)hl.nin * hl.nout; }
            if (li == 0) launch_gemm16<128, 128, 2, 4, E16_HIDDEN_TRAIN, 2, 1>(s, g, 1);   // the K = D launch under its own name (TAG 1)
            else gemm16<E16_HIDDEN_TRAIN>(s, g, 1);
            sync_stats(h, hl.fstat, hl.nout_p);
            prev = &hl;
        } else {
            g.B = w16(h, hl.tW); g.ldb = hl.nin_p;
            g.bias = h->pptr(hl.tb);
            g.scale = hl.scale.p; g.shift = hl.shift.p;
            gemm16<E16_HIDDEN_EVAL>(s, g, 1);
        }
        in = hl.H16.p;
        in_w = hl.nout_p;
    };
    for (int li = 0; li < nl; ++li) hidden_layer(li);
    int mu_slabs = 1;
    const float* mu_bias = h->pptr(h->tbmu);
    {   // mu (encode.py:268): latent-wide output, contraction split over up to 8 slabs; reparameterisation (encode.py:276-286).
        // One launch when the operands of 32 output rows fit a workgroup's LDS (gemm_skinny16.hpp: the slabs are the waves of a
        // workgroup and meet in LDS); else the split-K launch + the slab-summing kernel.  Same bits either way.
        const bf16_t* Bmu = w16(h, h->tWmu);
        if (training && prev) {
            fold_bn(h, s, h->tWmu, h->tbmu, h->L_p, in_w, *prev, true, h->Wf16_mu.p, h->biasf_mu.p);
            Bmu = h->Wf16_mu.p;
            mu_bias = h->biasf_mu.p;
        }
        const int want = std::max(1, std::min(kSkinnySplits, in_w / 128));
        const int k_per_split = (int)round_up(ceil_div(in_w, want), 64);
        mu_slabs = (int)ceil_div(in_w, k_per_split);
        const float* eps_ptr = eps_injected ? h->EPS.p : nullptr;
        bool fused = false;
        if (g_tuning.fused_skinny) {
            Skinny16Args a;
            memset(&a, 0, sizeof(a));
            a.A = in; a.lda = in_w;
            a.B = Bmu; a.ldb = in_w;
            a.M = bs_p; a.K = in_w; a.k_per_wave = k_per_split; a.nslab = mu_slabs;
            a.zeros = h->zeros16.p; a.bs = bs;
            a.bias = mu_bias; a.E = eps_ptr; a.key = layer_key(h, 0xEE); a.step_ptr = step_ptr(h); a.noise = add_noise ? 1 : 0;
            a.L = h->L; a.MU = h->MU.p; a.Z16 = h->Z16.p;
            fused = launch_skinny16<SK16_REPARAM>(s, a, h->L_p);
        }
        if (!fused) {
            Gemm16Args g = args16(h);
            g.A = in; g.lda = in_w;
            g.B = Bmu; g.ldb = in_w;
            g.C32 = h->skinny.p; g.ldc32 = h->L_p;
            g.M = bs_p; g.N = h->L_p; g.K = in_w;
            g.k_per_split = k_per_split;
            g.slab_stride = (int64_t)bs_p * h->L_p;
            gemm16<E16_SPLITK>(s, g, mu_slabs);
            const int64_t tot = (int64_t)bs_p * h->L_p;
            hipLaunchKernelGGL(vae_reparam16_kernel, dim3((unsigned)ceil_div(tot, 256)), dim3(256), 0, s,
                               (const float*)h->skinny.p, mu_slabs, (int64_t)bs_p * h->L_p, mu_bias, eps_ptr,
                               layer_key(h, 0xEE), step_ptr(h), add_noise ? 1 : 0, h->MU.p, h->Z16.p, bs, h->L, h->L_p, bs_p);
            VH_HIP(hipGetLastError());
        }
    }
    in = h->Z16.p;
    in_w = h->L_p;
    prev = nullptr;
    for (int li = nl; li < 2 * nl; ++li) hidden_layer(li);
    {   // reconstruction (encode.py:294), fp32
        Gemm16Args g = args16(h);
        g.A = in; g.lda = in_w;
        g.B = w16(h, h->tWo); g.ldb = in_w;
        g.bias = h->pptr(h->tbo);
        if (training && prev) {
            fold_bn(h, s, h->tWo, h->tbo, h->D_p, in_w, *prev, true, h->Wf16_out.p, h->biasf_out.p);
            g.B = h->Wf16_out.p;
            g.bias = h->biasf_out.p;
        }
        g.C32 = h->R.p; g.ldc32 = h->D_p;
        g.M = bs_p; g.N = h->D_p; g.K = in_w; g.k_per_split = g.K;
        gemm16<E16_BIAS>(s, g, 1);
    }
    if (training) {
        // running statistics (momentum 0.1, unbiased variance): off the critical path
        auto running = [h](hipStream_t st) {
            RunningTable rt;
            memset(&rt, 0, sizeof(rt));
            int maxn = 0;
            for (auto& hl : h->hidden) {
                rt.fstat[rt.n] = hl.fstat; rt.rm[rt.n] = h->pptr(hl.tRM); rt.rv[rt.n] = h->pptr(hl.tRV);
                rt.n_p[rt.n] = hl.nout_p;
                maxn = std::max(maxn, hl.nout_p);
                rt.n++;
            }
            hipLaunchKernelGGL(vae_bn_running_kernel, dim3((unsigned)ceil_div(maxn, 256), rt.n), dim3(256), 0, st, rt, stat_bs(h));
            VH_HIP(hipGetLastError());
        };
        if (defer) {
            defer->add_small(running);
        } else {   // forward-only call: run it now
            fork_side(h);
            running(h->side);
        }
    }
}

// plain VAE: the loss kernel takes its targets from the dataset rows the gather recorded instead of from an fp32 copy of the batch
bool loss_reads_dataset(const vh_vae* h) { return g_tuning.loss_from_dataset && h->kind == VH_VAE_PLAIN && h->batch_from_gather; }

void loss_and_seed16(vh_vae* h, SideQueue& q) {
    const int bs_global = h->global_bs > 0 ? h->global_bs : h->bs;
    Loss16Args a;
    a.R = h->R.p; a.X = h->Xb.p; a.rows = nullptr; a.ld = h->D_p;
    if (loss_reads_dataset(h)) { a.X = h->X.p; a.rows = h->Rb.p; }
    a.MU = h->MU.p; a.ldl = h->L_p;
    a.inv_b2 = (float)(1.0 / ((double)bs_global * (double)bs_global));
    a.bs = h->bs; a.bs_p = h->bs_p; a.S = h->S; a.L = h->L;
    a.ce_w = h->ce_w; a.ab_w = h->ab_w; a.sse_w = h->sse_w; a.kld_w = h->kld_w;
    a.dR16 = h->dR16.p; a.dMUk = h->dMUk.p; a.part = h->loss_part.p;
    a.NL = h->NL; a.lab0 = h->lab0; a.ntnf = h->ntnf; a.nab = h->nab; a.Lb = h->Lb.p;
    a.lab_part = h->NL > 0 ? h->lab_part.p : nullptr;
    const size_t loss_lds = (size_t)8 * h->D_p * sizeof(float);   // 4 waves x (reconstruction row + target row)
    VH_REQUIRE(loss_lds <= 160 * 1024 - 256, "%d input columns are too wide for the bf16 step's loss kernel (fp32 mode has no limit)", h->D);
    static bool loss_attr = false;
    if (!loss_attr) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_loss16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024 - 256));
        loss_attr = true;
    }
    auto kern = vae_loss16_kernel<true>;
    // the side stream starts here: its first items (see backward16) only need what this kernel leaves
    // plain VAE, up to 1152 padded input columns: the rows of a wavefront in registers (vae_loss16_reg_kernel; vae.loss_registers)
    const int nv = (int)ceil_div(h->D_p, 64);
    if (g_tuning.loss_registers && h->NL == 0 && nv <= 18) {
        if (nv <= 4) launch_forking(h, vae_loss16_reg_kernel<4>, dim3(h->loss_blocks), dim3(256), 0, a);
        else if (nv <= 6) launch_forking(h, vae_loss16_reg_kernel<6>, dim3(h->loss_blocks), dim3(256), 0, a);
        else if (nv <= 8) launch_forking(h, vae_loss16_reg_kernel<8>, dim3(h->loss_blocks), dim3(256), 0, a);
        else if (nv <= 12) launch_forking(h, vae_loss16_reg_kernel<12>, dim3(h->loss_blocks), dim3(256), 0, a);
        else launch_forking(h, vae_loss16_reg_kernel<18>, dim3(h->loss_blocks), dim3(256), 0, a);
    } else {
        launch_forking(h, kern, dim3(h->loss_blocks), dim3(256), loss_lds, a);
    }
    // the scalar reduction (loss means, sum of weights) is only needed by the optimiser
    const float* gw = h->gwsum_src;
    const float* lab_part = a.lab_part;
    q.add_small([h, bs_global, gw, lab_part](hipStream_t st) {
        hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(kLossFinThreads), 0, st, h->loss_part.p, h->loss_blocks, h->Wb.p,
                           h->bs, gw, bs_global, h->state.p, lab_part);
        VH_HIP(hipGetLastError());
    });
}

// dW slabs = dZ^T In straight from the ROW-major tensors dZ [bs_p][out_p], In [bs_p][in_p] (gemm_bf16_tn.hpp); dbias: the
// fp64 column sums of dZ over the real rows, for the layers whose bias gradient no other kernel produces (output, mu)
struct DwSpec {   // one weight-gradient product of the row-major dataflow
    int tW = -1;
    const bf16_t* dZ = nullptr;
    int out_p = 0;
    const bf16_t* In = nullptr;
    int in_p = 0;
    double* dbias = nullptr;
};
Gemm16TnArgs dw_args16_rm(vh_vae* h, const DwSpec& d, int& splits, hipStream_t st) {
    Tensor& t = h->tensors[d.tW];
    Gemm16TnArgs g;
    memset(&g, 0, sizeof(g));
    g.zeros = h->zeros16.p;
    g.xcd_remap = 1;
    g.A = d.dZ; g.lda = d.out_p;
    g.B = d.In; g.ldb = d.in_p;
    g.C32 = t.slab; g.ldc = d.in_p;
    g.M = d.out_p; g.N = d.in_p; g.K = h->bs_p; g.k_real = h->bs;
    g.k_per_split = (int)round_up(ceil_div(h->bs_p, t.nslab), 64);
    g.slab_stride = t.stride;
    g.colsum = d.dbias;
    splits = (int)ceil_div(h->bs_p, g.k_per_split);
    if (splits < t.nslab)
        VH_HIP(hipMemsetAsync(t.slab + (int64_t)splits * t.stride, 0, sizeof(float) * (t.nslab - splits) * t.stride, st));
    return g;
}
void grad_weight16_rm(vh_vae* h, int tW, const bf16_t* dZ, int out_p, const bf16_t* In, int in_p, double* dbias,
                      hipStream_t st) {
    int splits = 1;
    const Gemm16TnArgs g = dw_args16_rm(h, DwSpec{tW, dZ, out_p, In, in_p, dbias}, splits, st);
    if (dbias) gemm16_tn<1>(st, g, splits);
    else gemm16_tn<0>(st, g, splits);
}
// The last two weight gradients of the backward pass (encoder layers 0 and 1 on the main stream) as ONE launch
// (gemm_bf16_tn_pair_kernel): both must take the 64 x 128 tile of gemm16_tn_stg and no column sums.  false: launch them one by one.
bool grad_weight16_rm_pair(vh_vae* h, const DwSpec& a, const DwSpec& b, hipStream_t st) {
    auto fits = [](const DwSpec& d) { return d.tW >= 0 && d.dbias == nullptr && d.out_p > 32 && d.in_p > 32; };
    if (!g_tuning.dw_pair || !fits(a) || !fits(b)) return false;
    Gemm16TnPair pr;
    memset(&pr, 0, sizeof(pr));
    int splits[2];
    pr.p[0] = dw_args16_rm(h, a, splits[0], st);
    pr.p[1] = dw_args16_rm(h, b, splits[1], st);
    for (int i = 0; i < 2; ++i) {
        pr.gx[i] = (int)ceil_div(pr.p[i].N, 128);
        pr.gy[i] = (int)ceil_div(pr.p[i].M, 64);
        pr.gz[i] = splits[i];
    }
    pr.nwg0 = pr.gx[0] * pr.gy[0] * pr.gz[0];
    const int nwg = pr.nwg0 + pr.gx[1] * pr.gy[1] * pr.gz[1];
    auto kern = gemm_bf16_tn_pair_kernel<64, 128, 2, 2, 0>;
    constexpr size_t smem = gemm16_tn_smem_bytes<64, 128, 0>();
    static bool attr_set = false;
    if (!attr_set) {
        VH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), smem, st, pr);
    VH_HIP(hipGetLastError());
    return true;
}

// dIn16 = dZ16 [bs_p][out] x W16T [in][out]; the epilogue leaves the BatchNorm-backward sums of the layer below
void grad_input16(vh_vae* h, const bf16_t* dZ, int out_p, int tW, int in_p, Hidden& below) {
    Gemm16Args g = args16(h);
    g.A = dZ; g.lda = out_p;
    g.B = w16t(h, tW); g.ldb = out_p;
    g.M = h->bs_p; g.N = in_p; g.K = out_p; g.k_per_split = g.K;
    g.m_real = h->bs;
    g.C16 = below.DA16.p; g.ldc16 = in_p;
    g.Hbelow = below.H16.p; g.ldh = in_p;
    g.bnC = bn_src(h, below);
    g.bn_mean = below.mean.p; g.bn_istd = below.invstd.p;   // left by the fold of this layer's BatchNorm in the forward pass
    g.bstat_out = below.bstat;
    gemm16<E16_STORE_BNRED>(h->stream, g, 1);
    sync_stats(h, below.bstat, below.nout_p);
}

void backward16(vh_vae* h, bool masks_injected, SideQueue& q) {
    const int bs = h->bs, bs_p = h->bs_p, nl = h->nl;
    const DropCfg dc = drop_cfg(h, true, masks_injected);
    {   // output layer: dR16 is ready
        Hidden& last = h->hidden[2 * nl - 1];
        q.add([h, &last](hipStream_t st) {
            grad_weight16_rm(h, h->tWo, h->dR16.p, h->D_p, last.H16.p, last.nout_p, h->dbias_out, st);
        });
        // running statistics, the loss reduction and the output layer's weight gradient depend on nothing later than the
        // loss kernel: hand them to the side stream now (forked on that kernel's completion, loss_and_seed16)
        q.flush(h->side);
        grad_input16(h, h->dR16.p, h->D_p, h->tWo, last.nout_p, last);
    }
    int latent_slabs = 1;
    bool latent_fused = false;
    std::function<void(hipStream_t)> late_dw, late_dw_mu;
    DwSpec late_spec;   // what late_dw launches (row-major dataflow), for the paired launch with the first layer's
    auto hidden_bwd = [&](int li) {
        Hidden& hl = h->hidden[li];
        Dz16Args a;
        a.DA = hl.DA16.p; a.H = hl.H16.p; a.DZ = hl.DZ16.p; 
        a.n_p = hl.nout_p; a.bs = bs; a.bs_p = bs_p;
        a.bn = bn_src(h, hl);
        a.mean = hl.mean.p; a.istd = hl.invstd.p;
        a.bstat = hl.bstat;
        a.drop_scale = dc.scale;
        a.drop_mask = dc.injected ? hl.mask.p : nullptr; a.ld_mask = hl.nout_p;
        constexpr bool rm = true;   // weight gradients contract the row-major tensors (gemm_bf16_tn.hpp)
        const int in_p = li == 0 ? h->D_p : (li == nl ? h->L_p : h->hidden[li - 1].nout_p);
        // bias gradient = column sums of dZ, taken by this kernel from the values it writes (one fp64 atomic per column from each of
        // its bs_p / 64 row blocks).  (The weight-gradient GEMM could sum the dZ it streams instead -- COLSUM, as the output and mu
        // layers do: the dz kernel 12.1 -> 10.2 us, the dW GEMMs 16.9 -> 20.4 us, step 291 -> 293 us, profiles/r03zd_dz_colsum.txt.)
        constexpr bool colsum_in_gemm = false;
        a.dbias = hl.dbias;
        const bf16_t* InT = li == 0 ? h->Xb16.p : (li == nl ? h->Z16.p : h->hidden[li - 1].H16.p);   // row-major input of the layer
        auto dw = [h, &hl, InT, in_p](hipStream_t st) {
            grad_weight16_rm(h, hl.tW, hl.DZ16.p, hl.nout_p, InT, in_p, colsum_in_gemm ? hl.dbias : nullptr, st);
        };
        // Fork points: the top decoder layer and encoder layer 1 (and layer 0 when something is still queued): each
        // hands everything queued so far to the side stream.  The first layer's weight gradient is the end of the
        // chain -- nothing is left on the main stream for it to hide behind -- so it runs there.
        // (vae.fork_plan & 2: encoder layer 1's weight gradient waits for layer 0's on the main stream)
        const bool dw_on_main_late = li == 1 && nl >= 2 && (q.plan & 2) != 0;
        if (dw_on_main_late) {
            late_dw = dw;
            if (rm) late_spec = DwSpec{hl.tW, hl.DZ16.p, hl.nout_p, InT, in_p, colsum_in_gemm ? hl.dbias : nullptr};
        }
        else if (li > 0) q.add(dw);
        if (li == nl && h->comm) {
            // data parallel: every decoder-side gradient is now queued -- materialise that bucket of the flat gradient and
            // all-reduce it on the side stream while the encoder's backward still runs on the main stream
            q.add([h](hipStream_t st) {
                const int nb = h->opt16_blocks - h->opt16_bucketA_blk0;
                hipLaunchKernelGGL(vae_grad16_kernel, dim3(nb), dim3(256), 0, st, h->opt16_tab.p, (const uint8_t*)h->opt16_blk2t.p, stat_bs(h), h->G.p,
                                   h->opt16_bucketA_blk0, allrank_stats(h));
                VH_HIP(hipGetLastError());
                rccl_allreduce_sum_f32(h->comm, h->G.p + h->opt16_bucketA_off, h->flat_elems - h->opt16_bucketA_off, st);
            });
        }
        // Fork points: the top decoder layer and encoder layer 1 (and layer 0 when something is still queued): each hands
        // everything queued so far to the side stream (vae.fork_at_loss adds one at the loss kernel, loss_and_seed16).  A fork
        // costs the main stream ~5 us before its next kernel (the producing kernel's completion signal).  Variants measured at
        // C2 (profiles/r03w_*, r03zb_*, r03zc_*): these two 286 us per step; + the loss kernel 296; loss + LAST decoder layer +
        // encoder layer 1 299 against 297 on the box of that run.
        // (plan bit 16: NO fork at the top decoder layer -- its weight gradient waits for the fork at encoder layer 1)
        const bool fork = (li == 2 * nl - 1 && (q.plan & 16) == 0) || li == 1 || (li == 0 && !q.items.empty()) ||
                          (li == nl && nl >= 2 && (q.plan & 1) != 0);
        auto launch_dz = [&](auto kern, int cols, int rows) {
            const dim3 grid((unsigned)ceil_div(hl.nout_p, cols), (unsigned)ceil_div(bs_p, rows));
            if (fork) {
                launch_forking(h, kern, grid, dim3(256), 0, a);
                q.flush(h->side);
            } else {
                hipLaunchKernelGGL(kern, grid, dim3(256), 0, h->stream, a);
                VH_HIP(hipGetLastError());
            }
        };
        if (a.drop_mask) launch_dz(vae_dz16_kernel<kDz16Cols, kDz16Rows, true>, kDz16Cols, kDz16Rows);   // injected masks (parity tests)
        else launch_dz(vae_dz16_kernel<kDz16Cols, kDz16Rows, false>, kDz16Cols, kDz16Rows);
        if (li == 0) {
            const DwSpec first{hl.tW, hl.DZ16.p, hl.nout_p, InT, in_p, colsum_in_gemm ? hl.dbias : nullptr};
            if (!(rm && late_dw && grad_weight16_rm_pair(h, first, late_spec, h->stream))) {
                dw(h->stream);
                if (late_dw) late_dw(h->stream);
            }
            if (late_dw_mu) late_dw_mu(h->stream);
        } else if (li == nl) {
            // first decoder layer -> latent: latent-wide output; dMU = dZlat + d(KLD)/dmu fused into the same launch when the
            // operands of 32 rows fit a workgroup's LDS (gemm_skinny16.hpp), else split-K slabs summed by the latent kernel
            const int want = std::max(1, std::min(kSkinnySplits, hl.nout_p / 128));
            const int k_per_split = (int)round_up(ceil_div(hl.nout_p, want), 64);
            latent_slabs = (int)ceil_div(hl.nout_p, k_per_split);
            if (g_tuning.fused_skinny) {
                Skinny16Args sa;
                memset(&sa, 0, sizeof(sa));
                sa.A = hl.DZ16.p; sa.lda = hl.nout_p;
                sa.B = w16t(h, hl.tW); sa.ldb = hl.nout_p;
                sa.M = bs_p; sa.K = hl.nout_p; sa.k_per_wave = k_per_split; sa.nslab = latent_slabs;
                sa.zeros = h->zeros16.p; sa.bs = bs;
                sa.dMUk = h->dMUk.p; sa.dMU16 = h->dMU16.p;
                latent_fused = launch_skinny16<SK16_LATENT_BWD>(h->stream, sa, in_p);
            }
            if (!latent_fused) {
                Gemm16Args g = args16(h);
                g.A = hl.DZ16.p; g.lda = hl.nout_p;
                g.B = w16t(h, hl.tW); g.ldb = hl.nout_p;
                g.M = bs_p; g.N = in_p; g.K = hl.nout_p;
                g.k_per_split = k_per_split;
                g.C32 = h->skinny.p; g.ldc32 = in_p;
                g.slab_stride = (int64_t)bs_p * in_p;
                gemm16<E16_SPLITK>(h->stream, g, latent_slabs);
            }
        } else {
            grad_input16(h, hl.DZ16.p, hl.nout_p, hl.tW, in_p, h->hidden[li - 1]);
        }
    };
    for (int li = 2 * nl - 1; li >= nl; --li) hidden_bwd(li);
    {   // latent: dMU = dZlat + d(KLD)/dmu; mu layer
        Hidden& enc_last = h->hidden[nl - 1];
        if (!latent_fused) {
            const int64_t tot = (int64_t)bs_p * h->L_p;
            hipLaunchKernelGGL(vae_latent_bwd16_kernel, dim3((unsigned)ceil_div(tot, 256)), dim3(256), 0, h->stream,
                               (const float*)h->skinny.p, latent_slabs, (int64_t)bs_p * h->L_p, (const float*)h->dMUk.p,
                               h->dMU16.p, h->L_p, bs, bs_p);
            VH_HIP(hipGetLastError());
        }
        auto dw_mu = [h, &enc_last](hipStream_t st) {
            grad_weight16_rm(h, h->tWmu, h->dMU16.p, h->L_p, enc_last.H16.p, enc_last.nout_p, h->dbias_mu, st);
        };
        if ((q.plan & 8) && nl >= 2) late_dw_mu = dw_mu;   // (on the main stream, behind the first layer's)
        else q.add(dw_mu);
        grad_input16(h, h->dMU16.p, h->L_p, h->tWmu, enc_last.nout_p, enc_last);
    }
    for (int li = nl - 1; li >= 0; --li) hidden_bwd(li);
    if (h->side != h->stream) q.flush_tail(h->side);
    else q.flush_tail(h->stream);
    join_side(h);   // every weight gradient, the loss reduction and the running statistics are complete
}

void optimizer_step16(vh_vae* h) {
    const Opt16Tensor* tab = h->opt16_tab.p;
    if (h->comm) {
        // bucket B (encoder + mu; bucket A went out on the side stream during the encoder's backward, joined by now)
        hipLaunchKernelGGL(vae_grad16_kernel, dim3(h->opt16_bucketA_blk0), dim3(256), 0, h->stream, h->opt16_tab.p, (const uint8_t*)h->opt16_blk2t.p,
                           stat_bs(h), h->G.p, 0, allrank_stats(h));
        VH_HIP(hipGetLastError());
        rccl_allreduce_sum_f32(h->comm, h->G.p, h->opt16_bucketA_off, h->stream);
        tab = h->opt16_tab_flat.p;
    }
    const int nblk = h->opt16_blocks;
    // the scalar tail (d, k, counters, clearing the accumulators) rides on the last workgroup of the update kernel when ONE launch
    // covers every tensor of the step (vae.fused_finalize; the split / data-parallel schedules keep the separate launch)
    Opt16Tail tail{};
    const bool fuse_tail = g_tuning.fused_finalize && nblk == h->opt16_blocks;
    if (fuse_tail) {
        tail.ticket = h->opt_ticket.p; tail.st = h->state.p; tail.statbuf = h->statbuf.p;
        tail.nstat = h->keep_grads ? 0 : (int)h->statbuf.n; tail.nblocks = h->opt16_blocks; tail.adam = h->adam_lr > 0.f ? 1 : 0;
    }
    hipLaunchKernelGGL(vae_dadapt16_kernel, dim3(nblk), dim3(256), 0, h->stream, tab, (const uint8_t*)h->opt16_blk2t.p, stat_bs(h), h->P.p,
                       h->M1.p, h->M2.p, h->Sv.p, h->state.p, h->opt_part.p, h->adam_lr, 0, tail);
    VH_HIP(hipGetLastError());
    if (!fuse_tail) {
        hipLaunchKernelGGL(vae_dadapt_finalize_kernel, dim3(1), dim3(256), 0, h->stream, h->opt_part.p, h->opt16_blocks,
                           h->state.p, h->statbuf.p, h->keep_grads ? 0 : (int)h->statbuf.n, h->adam_lr > 0.f ? 1 : 0);
        VH_HIP(hipGetLastError());
    }
    h->stat_clean = !h->keep_grads;
}

// rows [base + batch * bs, ...) of the epoch's order -> (Xb, Xb16, Wb, Lb) on stream st; base = bs: the batch AFTER the cursor's
void gather_launch16(vh_vae* h, const int64_t* dev_idx, hipStream_t st, int64_t base, float* Xb, bf16_t* Xb16, float* Wb, int32_t* Lb,
                     long long* Rb) {
    auto kern = h->kind == VH_VAE_PLAIN ? vae_gather16_kernel<false> : vae_gather16_kernel<true>;
    const bool direct = g_tuning.loss_from_dataset && h->kind == VH_VAE_PLAIN;   // (then nobody reads the fp32 copy of the batch)
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(h->bs_p, 4)), dim3(64, 4), 0, st,
                       (const float*)h->X.p, h->ld_src, (int64_t)h->D_p, (const float*)h->w.p, dev_idx, h->shuffle,
                       (const long long*)&h->state.p->batch, base, h->bs, h->bs_p, direct ? (float*)nullptr : Xb, Xb16, Wb,
                       LabelSrc{h->labels, h->lab0}, Lb, direct ? Rb : (long long*)nullptr);
    VH_HIP(hipGetLastError());
}

void gather_rows16(vh_vae* h, const int64_t* dev_idx, SideQueue& q) {
    if (h->batch_prefetched) {
        // the previous step assembled this batch on the side stream (joined before its optimiser ran): take its buffers
        std::swap(h->Xb, h->Xb_n); std::swap(h->Xb16, h->Xb16_n); std::swap(h->Wb, h->Wb_n); std::swap(h->Lb, h->Lb_n);
        std::swap(h->Rb, h->Rb_n);
        h->batch_prefetched = false;
    } else {
        h->Rb.ensure((size_t)h->bs_p);
        gather_launch16(h, dev_idx, h->stream, 0, h->Xb.p, h->Xb16.p, h->Wb.p, h->Lb.p, h->Rb.p);
    }
    h->batch_from_gather = true;
    if (h->prefetch_next) {
        // first in the side stream's queue: nothing on it depends on this batch, and it is ready long before the join
        h->Xb_n.ensure(h->Xb.n); h->Xb16_n.ensure(h->Xb16.n); h->Wb_n.ensure(h->Wb.n); h->Lb_n.ensure(h->Lb.n);
        h->Rb_n.ensure((size_t)h->bs_p);
        q.add([h, dev_idx](hipStream_t st) {
            gather_launch16(h, dev_idx, st, (int64_t)h->bs, h->Xb_n.p, h->Xb16_n.p, h->Wb_n.p, h->Lb_n.p, h->Rb_n.p);
        });
        h->batch_prefetched = true;
    }
}

void train_step16(vh_vae* h, const int64_t* dev_idx, bool eps_injected, bool masks_injected) {
    SideQueue q;
    q.plan = fork_plan_for(h);
    gather_rows16(h, dev_idx, q);
    forward16(h, true, eps_injected, masks_injected, true, &q);
    loss_and_seed16(h, q);
    backward16(h, masks_injected, q);
    optimizer_step16(h);
}

// VAE.encode in bf16: the resident feature matrix is fp32; each chunk is cast once, then runs the eval-mode encoder
void encode16(vh_vae* h, float* latent) {
    hipStream_t s = h->stream;
    const int64_t chunk = 16384;
    int maxw = h->D_p;
    for (int li = 0; li < h->nl; ++li) maxw = std::max(maxw, h->hidden[li].nout_p);
    DevBuf<bf16_t> a0, a1;
    DevBuf<float> lat;
    a0.alloc((size_t)chunk * maxw);
    a1.alloc((size_t)chunk * maxw);
    lat.alloc((size_t)chunk * h->L);
    for (int li = 0; li < h->nl; ++li) {
        Hidden& hl = h->hidden[li];
        hipLaunchKernelGGL(vae_bn_eval_coeff_kernel, dim3((unsigned)ceil_div(hl.nout_p, 256)), dim3(256), 0, s, hl.nout_p,
                           h->pptr(hl.tG), h->pptr(hl.tB), h->pptr(hl.tRM), h->pptr(hl.tRV), hl.scale.p, hl.shift.p);
        VH_HIP(hipGetLastError());
    }
    for (int64_t lo = 0; lo < h->n; lo += chunk) {
        const int m = (int)std::min<int64_t>(chunk, h->n - lo);
        if (h->kind == VH_VAE_PLAIN) {
            const int64_t n4 = (int64_t)m * h->D_p / 4;
            hipLaunchKernelGGL(vae_cast16_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n4, 256), 4096)), dim3(256), 0, s,
                               h->X.p + (size_t)lo * h->D_p, a1.p, n4);
        } else {   // rows with their one-hot label block, straight to bf16
            hipLaunchKernelGGL(vae_gather16_kernel<true>, dim3((unsigned)ceil_div(m, 4)), dim3(64, 4), 0, s, (const float*)h->X.p,
                               h->ld_src, (int64_t)h->D_p, (const float*)h->w.p, (const int64_t*)nullptr, ShuffleSpec{0, 0, 1},
                               (const long long*)nullptr, lo, m, m, (float*)nullptr, a1.p, (float*)nullptr,
                               LabelSrc{h->labels, h->lab0}, (int32_t*)nullptr, (long long*)nullptr);
        }
        VH_HIP(hipGetLastError());
        const bf16_t* in = a1.p;
        int in_w = h->D_p;
        bf16_t* bufs[2] = {a0.p, a1.p};
        for (int li = 0; li < h->nl; ++li) {
            Hidden& hl = h->hidden[li];
            Gemm16Args g = args16(h);
            g.A = in; g.lda = in_w;
            g.B = w16(h, hl.tW); g.ldb = hl.nin_p;
            g.C16 = bufs[li & 1]; g.ldc16 = hl.nout_p;
            g.M = m; g.N = hl.nout_p; g.K = hl.nin_p; g.k_per_split = g.K;
            g.bias = h->pptr(hl.tb); g.scale = hl.scale.p; g.shift = hl.shift.p; g.m_real = m;
            gemm16<E16_HIDDEN_EVAL>(s, g, 1);
            in = bufs[li & 1];
            in_w = hl.nout_p;
        }
        Gemm16Args g = args16(h);
        g.A = in; g.lda = in_w;
        g.B = w16(h, h->tWmu); g.ldb = in_w;
        g.C32 = lat.p; g.ldc32 = h->L;
        g.M = m; g.N = h->L; g.K = in_w; g.k_per_split = g.K;
        g.bias = h->pptr(h->tbmu);
        gemm16<E16_LATENT_MASK>(s, g, 1);
        VH_HIP(hipMemcpyAsync(latent + (size_t)lo * h->L, lat.p, sizeof(float) * (size_t)m * h->L, hipMemcpyDeviceToHost, s));
        VH_HIP(hipStreamSynchronize(s));
    }
}

}  // namespace step16
