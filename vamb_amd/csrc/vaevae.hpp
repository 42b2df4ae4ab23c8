// vaevae.hpp -- the joint TaxVamb trainer (included by vae.hip, which owns the VAE step it is built from).
//
// Replaces VAEVAE.trainepoch / trainmodel (/root/reference/vamb/semisupervised_encode.py:829-1084) as VAEVAEHLoss configures it
// (/root/reference/vamb/taxvamb_encode.py:551-743): three networks -- VAEVamb (vamb.encode.VAE), VAELabels and VAEJoint
// (VAEConcat) -- and, per batch, SEVEN passes through them, one backward over the sum of three losses, one torch.optim.Adam
// step over all parameters:
//
//   pass        network    rows          forward                     what flows back
//   joint       VAEJoint   supervised    encoder + decoder           encoder only (its decoder's outputs are discarded, :899)
//   vamb_x      VAEVamb    supervised    decoder on mu_sup + noise   decoder, and d/d mu_sup into VAEJoint's encoder (:903)
//   labels_x    VAELabels  supervised    decoder on mu_sup + noise   decoder, and d/d mu_sup into VAEJoint's encoder (:906)
//   vamb_u      VAEVamb    unsupervised  encoder + decoder           everything: VAE.calc_loss (:936)
//   vamb_s      VAEVamb    supervised    encoder + decoder           encoder only: mu_vamb_sup_s enters kld_gauss (:919, :807)
//   labels_u    VAELabels  unsupervised  encoder + decoder           everything: VAELabels(HLoss).calc_loss (:952)
//   labels_s    VAELabels  supervised    encoder + decoder           encoder only (:928, :808)
//
// Nobody calls .train() / .eval() in that loop: the modules are in training mode, so EVERY pass applies dropout, normalises with
// its own batch statistics and updates the running statistics, in the order above.
//
// Design: a pass is a vh_vae handle.  The three networks' own handles play joint / vamb_u / labels_u; the other four are PASS
// REPLICAS -- same configuration, parameters / moments / running statistics BORROWED from their network (DevBuf::borrow), own
// activations, own gradient slabs, own dropout / noise seed.  Every pass runs the fp32 step's forward / loss / backward
// (restricted to the decoder or the encoder where the table says so); each handle's slabs are reduced into its flat gradient
// buffer (the data-parallel path's kernel), the passes of a network are summed with their loss scales (sum of the batch's
// weights, see LossArgs::inv_b2) and the network's Adam kernel runs on the sum.
//
// Scheduling.  All seven handles share ONE stream pair (VAEVamb's) for the duration of a call: stream order is program order.
// (Round 5 measured the obvious alternative -- every pass on its handle's OWN stream pair, the passes meeting through events
// where the step's dataflow says so: 2.55 ms per step against 1.68 ms on the shared pair, 14 HIP streams over the process's
// hardware queues; a two-lane variant 3.69 ms and one fault in three runs: profiles/r05h_taxvamb_lanes.txt, r05k_*.  Both were
// removed in round 6; DESIGN.md section 4.8 keeps the numbers.)
//
// The Kullback-Leibler terms of calc_loss_joint: every logsigma in the reference is a zero tensor (:241, :507, :916, :922), so
// kld_gauss(p, 0, q, 0) = 0.5 * mean((p - q)^2) over all batch x nlatent elements (taxvamb_encode.py:541-548).
#pragma once

namespace {

struct JointState {
    double step[8];    // loss_joint, ce_joint, sse_joint, ce_labels_joint, kld_vamb_joint, kld_labels_joint (last step)
    double epoch[8];   // sums over the epoch's steps
};

constexpr int kVvPasses = 7;   // joint, vamb_x, labels_x, vamb_u, vamb_s, labels_u, labels_s: the order of the reference's step

// d(kld terms)/d mu in the units of every other gradient of the supervised loss (x sum(w_sup) later, LossArgs::inv_b2):
//   gk = kld_w / (L * B^2);  a = mu_sup - mu_vs, b = mu_sup - mu_ls
//   dj = dz_v + dz_l + gk (a + b)   (VAEJoint's mu: the two decoders' input gradients + both KLD terms)
//   dv = -gk a, dl = -gk b          (the mu of the two *_s passes)
// and per workgroup the sums of a^2 and b^2 (the two KLD values).  Padding rows / columns get zeros.
__global__ __launch_bounds__(256) void vv_kld_kernel(const float* __restrict__ mu_sup, const float* __restrict__ mu_vs,
                                                     const float* __restrict__ mu_ls, const float* __restrict__ dz_v,
                                                     const float* __restrict__ dz_l, float gk, int bs, int L, int L_p, int bs_p,
                                                     float* __restrict__ dj, float* __restrict__ dv, float* __restrict__ dl,
                                                     float* __restrict__ part) {
    __shared__ float red[2][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float sa = 0.f, sb = 0.f;
    if (i < (int64_t)bs_p * L_p) {
        const int r = (int)(i / L_p), c = (int)(i % L_p);
        float a = 0.f, b = 0.f, z = 0.f;
        if (r < bs && c < L) {
            const float m = mu_sup[i];
            a = m - mu_vs[i];
            b = m - mu_ls[i];
            z = dz_v[i] + dz_l[i];
        }
        dj[i] = z + gk * (a + b);
        dv[i] = -gk * a;
        dl[i] = -gk * b;
        sa = a * a;
        sb = b * b;
    }
    sa = wave_sum(sa);
    sb = wave_sum(sb);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = sa; red[1][wave] = sb; }
    __syncthreads();
    if (threadIdx.x < 2) part[(int64_t)blockIdx.x * 2 + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// calc_loss_joint's return values (taxvamb_encode.py:735-743) from the pieces the passes left behind: the weighted row means of
// the vamb_x loss (ab, ce, sse), the label loss of labels_x, the two KLD sums, the mean weight of the supervised batch.
__global__ __launch_bounds__(256) void vv_joint_finalize_kernel(const float* __restrict__ kld_part, int nblk,
                                                                const StepState* __restrict__ vx, const StepState* __restrict__ lx,
                                                                float ce_w, float sse_w, float kld_w, int bs, int L,
                                                                JointState* __restrict__ js) {
    __shared__ double red[2][4];
    double sa = 0.0, sb = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) { sa += (double)kld_part[2 * b]; sb += (double)kld_part[2 * b + 1]; }
    sa = wave_sum_f64(sa);
    sb = wave_sum_f64(sb);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = sa; red[1][wave] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)bs * (double)L;
        const double kld_v = 0.5 * ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / n;
        const double kld_l = 0.5 * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / n;
        const double wmean = vx->wsum / (double)bs;
        const double ab = vx->step_loss[1], ce = vx->step_loss[2], sse = vx->step_loss[3], cel = lx->step_label[0];
        const double loss = ((((ce + ab) + sse) + cel) + (kld_v + kld_l) * (double)kld_w) * wmean;
        // ce_joint / sse_joint are logged UNWEIGHTED (ce.mean(), sse.mean()).  One sample: the weight is 0, and the value IS 0 in
        // float32 -- VAE._decode's softmax over a single column is the constant 1, so ce = -log(1 + 1e-9) * x (encode.py:302, 329)
        const double v[6] = {loss, ce_w > 0.f ? ce / (double)ce_w : 0.0, sse / (double)sse_w, cel, kld_v, kld_l};
        for (int t = 0; t < 6; ++t) { js->step[t] = v[t]; js->epoch[t] += v[t]; }
    }
}

// dst[i] = sum over the sources whose flat range holds i of scale_k * src_k[i]  (scale_k = that pass's sum of batch weights, read
// from device memory); elements no source covers become 0 -- a pass's buffer outside its range is never read (stale slabs).
struct VvSource {
    const float* g;
    const double* wsum;
    int64_t lo, hi;
};
__global__ __launch_bounds__(256) void vv_combine_kernel(float* __restrict__ dst, int64_t n, VvSource a, VvSource b, VvSource c) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    if (a.g && i >= a.lo && i < a.hi) v += a.g[i] * (float)*a.wsum;
    if (b.g && i >= b.lo && i < b.hi) v += b.g[i] * (float)*b.wsum;
    if (c.g && i >= c.lo && i < c.hi) v += c.g[i] * (float)*c.wsum;
    dst[i] = v;
}

__global__ void vv_zero_joint_epoch_kernel(JointState* js) {
    if (threadIdx.x < 8) js->epoch[threadIdx.x] = 0.0;
}

}  // namespace

struct vh_vaevae {
    vh_vae *vamb = nullptr, *labels = nullptr, *joint = nullptr;   // the three networks (not owned)
    std::unique_ptr<vh_vae> vamb_x, vamb_s, labels_x, labels_s;    // pass replicas
    DevBuf<float> zero_bias, kld_part;
    DevBuf<JointState> jstate;
    DevBuf<int64_t> perm;
    PinnedBuf<int64_t> h_perm;
    int64_t n = 0;         // rows of the attached datasets (0: none)
    int kld_blocks = 0;
    bool entered = false;
    hipStream_t saved[kVvPasses][2] = {};    // the handles' own stream pairs while a call borrows VAEVamb's

    vh_vae* pass(int p) {
        vh_vae* v[kVvPasses] = {joint, vamb_x.get(), labels_x.get(), vamb, vamb_s.get(), labels, labels_s.get()};
        return v[p];
    }
};

namespace {

std::unique_ptr<vh_vae> vv_make_replica(vh_vae* m, uint64_t salt) {
    vh_vae_config cfg = m->cfg;
    cfg.seed = m->cfg.seed * 0x9E3779B97F4A7C15ull + salt;   // its own dropout / noise streams
    vh_vae_labels_config lab;
    memset(&lab, 0, sizeof(lab));
    lab.kind = m->kind;
    lab.nlabels = m->NL;
    lab.optimizer = VH_OPT_DADAPT_ADAM;
    vh_vae* raw = nullptr;
    const int rc = create_vae(&cfg, m->kind == VH_VAE_PLAIN ? nullptr : &lab, &raw);
    if (rc != VH_OK) throw InvalidArg{g_last_error};
    std::unique_ptr<vh_vae> r(raw);
    VH_REQUIRE(r->flat_elems == m->flat_elems && r->bn_elems == m->bn_elems, "replica layout differs from its network");
    VH_HIP(hipStreamSynchronize(r->stream));
    r->P.borrow(m->P);
    r->M1.borrow(m->M1);
    r->M2.borrow(m->M2);
    r->Sv.borrow(m->Sv);
    r->bnbuf.borrow(m->bnbuf);
    r->step_src = &m->state.p->step;
    r->batch_src = &m->state.p->batch;
    return r;
}

// For the duration of a call all seven passes run on VAEVamb's stream pair (stream order is program order); replicas follow their
// network's taxonomy.
void vv_enter(vh_vaevae* t) {
    VH_REQUIRE(!t->entered, "trainer is busy");
    vh_vae* nets[3] = {t->vamb, t->labels, t->joint};
    // every check BEFORE the first handle is touched: a refusal must leave the networks as they were
    for (vh_vae* m : nets) {
        VH_REQUIRE(!m->bf16, "the joint trainer runs the fp32 step: vh_vae_set_precision(h, 0) on its three networks");
        VH_REQUIRE(m->comm == nullptr, "the joint trainer is a single-process path");
        VH_REQUIRE(m->adam_lr > 0.f, "the joint trainer optimises with torch.optim.Adam: vh_vae_set_optimizer(h, VH_OPT_ADAM, lr)");
    }
    VH_REQUIRE(t->joint->n_leaves == t->labels->n_leaves && t->joint->n_nodes == t->labels->n_nodes,
               "VAEJoint and VAELabels must share one taxonomy");
    for (int p = 0; p < kVvPasses; ++p) {
        vh_vae* h = t->pass(p);
        VH_HIP(hipStreamSynchronize(h->stream));
        if (h->side != h->stream) VH_HIP(hipStreamSynchronize(h->side));
        t->saved[p][0] = h->stream;
        t->saved[p][1] = h->side;
    }
    t->entered = true;
    for (int p = 0; p < kVvPasses; ++p) {
        vh_vae* h = t->pass(p);
        h->stream = t->vamb->stream;
        h->side = t->vamb->side;
        h->gwsum_src = nullptr;
        h->global_bs = 0;
        h->shuffle.key = 0ull;   // explicit row lists
    }
    for (vh_vae* r : {t->labels_x.get(), t->labels_s.get()}) {
        r->leaf_masks.borrow(t->labels->leaf_masks);
        r->n_leaves = t->labels->n_leaves;
        r->n_nodes = t->labels->n_nodes;
    }
}

void vv_exit(vh_vaevae* t) {
    if (!t->entered) return;
    for (int p = 0; p < kVvPasses; ++p) {
        vh_vae* h = t->pass(p);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->side);
    }
    for (int p = 0; p < kVvPasses; ++p) {
        vh_vae* h = t->pass(p);
        h->stream = t->saved[p][0];
        h->side = t->saved[p][1];
    }
    t->entered = false;
}

struct VvScope {   // exception-safe enter / exit
    vh_vaevae* t;
    explicit VvScope(vh_vaevae* t_) : t(t_) { vv_enter(t); }
    ~VvScope() { vv_exit(t); }
};

void vv_prepare(vh_vaevae* t, int bs) {
    for (int p = 0; p < kVvPasses; ++p) prepare_batch(t->pass(p), bs);
    vh_vae* j = t->joint;
    t->kld_blocks = (int)ceil_div((int64_t)j->bs_p * j->L_p, 256);
    t->kld_part.ensure((size_t)t->kld_blocks * 2);
}

// One batch of VAEVAE.trainepoch (semisupervised_encode.py:864-997): rows of the three datasets at the networks' batch cursor.
// All passes share one stream pair: the order below is the order on the device.
void vv_step(vh_vaevae* t, const int64_t* dev_idx, bool eps_inj, bool masks_inj) {
    enum { P_J = 0, P_VX = 1, P_LX = 2, P_V = 3, P_VS = 4, P_LB = 5, P_LS = 6 };
    vh_vae *J = t->joint, *V = t->vamb, *Lb = t->labels, *Vx = t->vamb_x.get(), *Vs = t->vamb_s.get(), *Lx = t->labels_x.get(),
           *Ls = t->labels_s.get();
    for (int p = 0; p < kVvPasses; ++p) gather_rows(t->pass(p), dev_idx);
    // ---- the seven forward passes (each updates nothing but its own activations and batch sums)
    forward(J, true, eps_inj, masks_inj, true);
    forward(Vx, true, eps_inj, masks_inj, true, PASS_DECODER, J->MU.p, t->zero_bias.p);
    forward(Lx, true, eps_inj, masks_inj, true, PASS_DECODER, J->MU.p, t->zero_bias.p);
    forward(V, true, eps_inj, masks_inj, true);
    forward(Vs, true, eps_inj, masks_inj, true);
    forward(Lb, true, eps_inj, masks_inj, true);
    forward(Ls, true, eps_inj, masks_inj, true);
    auto reduce = [&](vh_vae* h) {   // the pass's gradient slabs -> its flat buffer
        hipLaunchKernelGGL(vae_reduce_slabs_kernel, dim3(h->opt_blocks), dim3(256), 0, h->stream, h->opt_tab, h->G.p, 0);
        VH_HIP(hipGetLastError());
    };
    // ---- calc_loss_joint's reconstruction terms through the two one-modality decoders (no KLD of their own): the chain the KLD
    // kernel and VAEJoint's backward wait for, enqueued first
    loss_and_seed(Vx, 0.0f);
    backward(Vx, masks_inj, PASS_DECODER);
    loss_and_seed(Lx, 0.0f);
    backward(Lx, masks_inj, PASS_DECODER);
    // ---- VAEVamb.calc_loss / VAELabels.calc_loss on the unsupervised rows: the ordinary step's loss and backward
    loss_and_seed(V);
    backward(V, masks_inj);
    loss_and_seed(Lb);
    backward(Lb, masks_inj);
    // ---- ... and the two kld_gauss terms, which tie VAEJoint's mu to the mu of the two *_s passes
    const int bs = J->bs;
    const float gk = (float)((double)V->kld_w / ((double)J->L * (double)bs * (double)bs));   // 1 / (VAEVamb.nlatent * VAEVamb.beta), taxvamb_encode.py:730
    hipLaunchKernelGGL(vv_kld_kernel, dim3((unsigned)t->kld_blocks), dim3(256), 0, J->stream, (const float*)J->MU.p,
                       (const float*)Vs->MU.p, (const float*)Ls->MU.p, (const float*)Vx->dMU.p, (const float*)Lx->dMU.p, gk, bs, J->L,
                       J->L_p, J->bs_p, J->dMUk.p, Vs->dMUk.p, Ls->dMUk.p, t->kld_part.p);
    VH_HIP(hipGetLastError());
    backward(J, masks_inj, PASS_ENCODER);
    backward(Vs, masks_inj, PASS_ENCODER);
    backward(Ls, masks_inj, PASS_ENCODER);
    hipLaunchKernelGGL(vv_joint_finalize_kernel, dim3(1), dim3(256), 0, J->stream, (const float*)t->kld_part.p, t->kld_blocks,
                       (const StepState*)Vx->state.p, (const StepState*)Lx->state.p, V->ce_w, V->sse_w, V->kld_w, bs, J->L,
                       t->jstate.p);
    VH_HIP(hipGetLastError());
    // ---- one gradient per network: every pass's slabs -> its flat buffer, the passes summed with their loss scales
    for (int p : {P_VX, P_LX, P_VS, P_LS}) reduce(t->pass(p));
    auto combine = [&](vh_vae* net, VvSource a, VvSource b, VvSource c) {
        const int64_t n = (int64_t)net->flat_elems;
        hipLaunchKernelGGL(vv_combine_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, net->stream, net->G.p, n, a, b, c);
        VH_HIP(hipGetLastError());
    };
    auto split_of = [](vh_vae* h) { return (int64_t)h->tensors[h->hidden[h->nl].tW].off; };   // encoder + mu | decoder + output
    const double* w_sup = &Vx->state.p->wsum;   // sum of the supervised batch's weights (the labels_x pass holds the same value)
    const VvSource none{nullptr, nullptr, 0, 0};
    // Each network on its own stream: its passes' running statistics in the order of the reference's step (BatchNorm1d updates
    // them in every training-mode forward: decoder pass, unsupervised pass, supervised pass), the gradient sum, torch.optim.Adam
    // (per-element: three launches are one optimiser).  The update's finalize kernel clears the network handle's batch sums, so
    // the running statistics go first.
    reduce(J);
    combine(J, VvSource{J->G.p, w_sup, 0, split_of(J)}, none, none);
    optimizer_step(J, true);
    reduce(V);
    combine(V, VvSource{V->G.p, &V->state.p->wsum, 0, (int64_t)V->flat_elems},
            VvSource{Vx->G.p, w_sup, split_of(V), (int64_t)V->flat_elems}, VvSource{Vs->G.p, w_sup, 0, split_of(V)});
    optimizer_step(V, true);
    reduce(Lb);
    combine(Lb, VvSource{Lb->G.p, &Lb->state.p->wsum, 0, (int64_t)Lb->flat_elems},
            VvSource{Lx->G.p, w_sup, split_of(Lb), (int64_t)Lb->flat_elems}, VvSource{Ls->G.p, w_sup, 0, split_of(Lb)});
    optimizer_step(Lb, true);
    // BatchNorm1d.num_batches_tracked: one per training-mode forward of the layer
    for (vh_vae* m : {V, Lb})
        for (int li = 0; li < 2 * m->nl; ++li) m->hidden[li].batches_tracked += li < m->nl ? 2 : 3;
    count_batches(J, 1);
}

void vv_reset_epoch(vh_vaevae* t) {
    for (vh_vae* h : {t->vamb, t->labels, t->joint, t->vamb_x.get(), t->labels_x.get()}) reset_epoch_sums(h);
    for (vh_vae* h : {t->vamb, t->labels, t->joint}) reset_batch_index(h);
    hipLaunchKernelGGL(vv_zero_joint_epoch_kernel, dim3(1), dim3(64), 0, t->vamb->stream, t->jstate.p);
    VH_HIP(hipGetLastError());
}

// the 17 numbers of trainepoch's log line (:830-848), from the step (per_epoch false) or the epoch sums / n_batches
void vv_metrics(vh_vaevae* t, bool per_epoch, int64_t n_batches, int64_t batch, double out[17]) {
    StepState v, l;
    JointState js;
    read_state(t->vamb, &v);
    read_state(t->labels, &l);
    VH_HIP(hipMemcpy(&js, t->jstate.p, sizeof(js), hipMemcpyDeviceToHost));
    const double nb = per_epoch ? (double)n_batches : 1.0;
    const double* lv = per_epoch ? v.epoch_loss : v.step_loss;
    const double* ll = per_epoch ? l.epoch_loss : l.step_loss;
    const double* lab = per_epoch ? l.epoch_label : l.step_label;
    const double* jj = per_epoch ? js.epoch : js.step;
    for (int i = 0; i < 5; ++i) out[i] = lv[i] / nb;                         // loss_vamb, ab_vamb, ce_vamb, sse_vamb, kld_vamb
    out[5] = ll[0] / nb;                                                     // loss_labels
    out[6] = lab[0] / nb;                                                    // ce_labels_labels
    out[7] = ll[4] / nb / (double)t->labels->kld_w;                          // kld_labels (unweighted, :252)
    // correct_*: the reference divides the epoch's count by the batch size, then by the number of batches (:999-1004); the
    // hierarchical losses return a constant 0 (taxvamb_encode.py:355, 743)
    out[8] = t->labels->n_leaves > 0 ? 0.0 : lab[1] / (per_epoch ? (double)batch : 1.0) / nb;
    for (int i = 0; i < 6; ++i) out[9 + i] = jj[i] / nb;                      // loss_joint .. kld_labels_joint
    out[15] = 0.0;
    if (t->labels->n_leaves == 0) {
        StepState lx;
        read_state(t->labels_x.get(), &lx);
        out[15] = (per_epoch ? lx.epoch_label[1] / (double)batch : lx.step_label[1]) / nb;
    }
    out[16] = out[9] + out[0] + out[5];                                      // loss
}

}  // namespace

extern "C" {

int vh_vaevae_create(vh_vae* vamb, vh_vae* labels, vh_vae* joint, vh_vaevae** out) {
    return guarded([&] {
        VH_REQUIRE(vamb != nullptr && labels != nullptr && joint != nullptr && out != nullptr, "NULL argument");
        *out = nullptr;
        VH_REQUIRE(vamb->kind == VH_VAE_PLAIN && labels->kind == VH_VAE_LABELS && joint->kind == VH_VAE_CONCAT,
                   "expected (VAE, VAELabels, VAEConcat) handles");
        VH_REQUIRE(vamb->L == labels->L && vamb->L == joint->L, "the three networks must share one latent width");
        VH_REQUIRE(vamb->S == joint->S, "VAEVamb has %d samples, VAEJoint %d", vamb->S, joint->S);
        VH_REQUIRE(labels->NL == joint->NL, "VAELabels has %d label columns, VAEJoint %d", labels->NL, joint->NL);
        VH_REQUIRE(vamb->nl == labels->nl && vamb->nl == joint->nl, "the three networks must have the same depth");
        std::unique_ptr<vh_vaevae> t(new vh_vaevae());
        t->vamb = vamb; t->labels = labels; t->joint = joint;
        t->vamb_x = vv_make_replica(vamb, 0x11);
        t->vamb_s = vv_make_replica(vamb, 0x12);
        t->labels_x = vv_make_replica(labels, 0x21);
        t->labels_s = vv_make_replica(labels, 0x22);
        t->zero_bias.alloc((size_t)vamb->L_p);
        VH_HIP(hipMemset(t->zero_bias.p, 0, t->zero_bias.bytes()));
        t->jstate.alloc(1);
        VH_HIP(hipMemset(t->jstate.p, 0, sizeof(JointState)));
        *out = t.release();
    });
}

int vh_vaevae_destroy(vh_vaevae* t) {
    return guarded([&] { delete t; });
}

// The three row-aligned datasets behind the 10-tensor loader of make_dataloader_semisupervised[_hloss] (taxvamb_encode.py:192-239),
// named as THAT function names them: `unsup` = tensors[0:4] (dataloader_vamb: features + weights), `unsup_labels` = tensors[4]
// (dataloader_labels), `sup` = tensors[5:10] (dataloader_joint: features + weights + labels).  VAEVAE.trainepoch unpacks the ten
// tensors BY POSITION (semisupervised_encode.py:864-875) and its names are the other way round: tensors[0:5] are its `*_sup`
// batch -- VAEJoint's input, the targets of the two decoders fed with mu_sup, calc_loss_joint's weights, the two `_sup_s`
// passes -- and tensors[5:10] its `*_unsup` batch (VAEVamb.calc_loss, VAELabels.calc_loss).  The passes are bound the way
// trainepoch uses them.  (`vamb bin taxvamb` builds the three loaders from the same contigs with one permutation seed, so both
// halves hold the same rows there; the fixture vaevae_tree_split pins the binding with halves that differ.  Until round 5 the
// passes followed the loader's names -- ADVICE r4.)
int vh_vaevae_set_datasets(vh_vaevae* t, vh_dataset* unsup, vh_dataset* unsup_labels, vh_dataset* sup) {
    return guarded([&] {
        VH_REQUIRE(t != nullptr && unsup != nullptr && unsup_labels != nullptr && sup != nullptr, "NULL argument");
        VH_REQUIRE(unsup->n == unsup_labels->n && unsup->n == sup->n, "the three datasets must have the same number of rows");
        VH_REQUIRE(unsup->X.p != nullptr && sup->X.p != nullptr && unsup->S == t->vamb->S && sup->S == t->vamb->S &&
                       unsup->D_p == t->vamb->D_p && sup->D_p == t->vamb->D_p,
                   "feature datasets do not match the model (%d samples)", t->vamb->S);
        VH_REQUIRE(unsup_labels->labels.p != nullptr && sup->labels.p != nullptr && unsup_labels->NL == t->labels->NL &&
                       sup->NL == t->labels->NL, "label datasets do not match the model's %d label columns", t->labels->NL);
        if (t->labels->n_leaves > 0)
            VH_REQUIRE(unsup_labels->max_label < t->labels->n_nodes && sup->max_label < t->labels->n_nodes,
                       "a label is not a node of the taxonomy (%d nodes)", t->labels->n_nodes);
        for (int p = 0; p < kVvPasses; ++p) {
            vh_vae* h = t->pass(p);
            if (h->stream) VH_HIP(hipStreamSynchronize(h->stream));
            // the networks' own handles play the `*_unsup` passes (vamb_u, labels_u): tensors[5:10]; VAEJoint and the four
            // replicas (vamb_x, labels_x, vamb_s, labels_s) see trainepoch's `*_sup` batch: tensors[0:4] + tensors[4]
            const bool unsup_pass = h == t->vamb || h == t->labels;
            vh_dataset* feat = unsup_pass ? sup : unsup;           // features + weights
            vh_dataset* lab = unsup_pass ? sup : unsup_labels;     // labels
            h->own.X.release();
            h->own.w.release();
            h->n = feat->n;
            // (VAELabels.calc_loss has no contig weights: its passes take the unit weights of the labels-only dataset, whichever
            // half their labels come from)
            h->w.p = h->kind == VH_VAE_LABELS ? unsup_labels->w.p : feat->w.p;
            if (h->kind == VH_VAE_LABELS) { h->X.p = nullptr; h->ld_src = 0; }
            else { h->X.p = feat->X.p; h->ld_src = feat->D_p; }
            h->labels = h->kind == VH_VAE_PLAIN ? nullptr : lab->labels.p;
        }
        t->n = sup->n;
    });
}

// One optimisation step on the rows `rows` (parity interface).  eps: 7 x [batch][nlatent] in pass order (joint, vamb_x, labels_x,
// vamb_u, vamb_s, labels_u, labels_s) or NULL (generated); masks: dropout keep-masks, passes concatenated in the same order, each
// pass its hidden layers in order ([batch][width] bytes per layer; the _x passes have decoder layers only), or NULL.
int vh_vaevae_train_step(vh_vaevae* t, const int64_t* rows, int64_t batch, const float* eps, const uint8_t* masks,
                         double metrics[17]) {
    return guarded([&] {
        VH_REQUIRE(t != nullptr && rows != nullptr, "NULL argument");
        VH_REQUIRE(t->n > 0, "no datasets: call vh_vaevae_set_datasets first");
        VH_REQUIRE(batch >= 2, "BatchNorm1d needs more than 1 value per channel when training (batch=%lld)", (long long)batch);
        VH_REQUIRE(batch <= (1 << 24), "batch too large");
        for (int64_t i = 0; i < batch; ++i) VH_REQUIRE(rows[i] >= 0 && rows[i] < t->n, "row %lld out of range", (long long)rows[i]);
        VvScope scope(t);
        vv_prepare(t, (int)batch);
        hipStream_t s = t->vamb->stream;
        t->perm.ensure((size_t)batch);
        t->h_perm.ensure((size_t)batch);
        memcpy(t->h_perm.p, rows, sizeof(int64_t) * batch);
        VH_HIP(hipMemcpyAsync(t->perm.p, t->h_perm.p, sizeof(int64_t) * batch, hipMemcpyHostToDevice, s));
        const bool drop = t->vamb->cfg.dropout > 0;
        size_t moff = 0;
        for (int p = 0; p < kVvPasses; ++p) {
            vh_vae* h = t->pass(p);
            if (eps) {
                std::vector<float> e((size_t)h->bs_p * h->L_p, 0.f);
                const float* src = eps + (size_t)p * batch * h->L;
                for (int r = 0; r < batch; ++r) memcpy(e.data() + (size_t)r * h->L_p, src + (size_t)r * h->L, sizeof(float) * h->L);
                VH_HIP(hipMemcpyAsync(h->EPS.p, e.data(), sizeof(float) * e.size(), hipMemcpyHostToDevice, s));
                VH_HIP(hipStreamSynchronize(s));
            }
            if (masks && drop) {
                const bool dec_only = p == 1 || p == 2;
                const int first = dec_only ? h->nl : 0;
                upload_masks(h, masks + moff, (int)batch, first, 2 * h->nl);
                for (int li = first; li < 2 * h->nl; ++li) moff += (size_t)batch * h->hidden[li].nout;
            }
        }
        vv_reset_epoch(t);
        for (int p = 0; p < kVvPasses; ++p) t->pass(p)->keep_grads = true;
        vv_step(t, t->perm.p, eps != nullptr, masks != nullptr && drop);
        for (int p = 0; p < kVvPasses; ++p) t->pass(p)->keep_grads = false;
        VH_HIP(hipStreamSynchronize(s));
        if (metrics) vv_metrics(t, false, 1, batch, metrics);
    });
}

// One epoch: batch b takes rows[b * batch .. (b + 1) * batch) of the three datasets (the caller draws the order:
// sequential before the first batch-size doubling, shuffled after it, semisupervised_encode.py:852-862).  Everything is enqueued
// without a host synchronisation; metrics = the 17 epoch means of trainepoch's log line.
int vh_vaevae_train_epoch(vh_vaevae* t, const int64_t* rows, int64_t n_batches, int64_t batch, double metrics[17]) {
    return guarded([&] {
        VH_REQUIRE(t != nullptr && rows != nullptr, "NULL argument");
        VH_REQUIRE(t->n > 0, "no datasets: call vh_vaevae_set_datasets first");
        VH_REQUIRE(n_batches >= 1, "no batches");
        VH_REQUIRE(batch >= 2, "BatchNorm1d needs more than 1 value per channel when training (batch=%lld)", (long long)batch);
        VH_REQUIRE(batch <= (1 << 24), "batch too large");
        const int64_t total = n_batches * batch;
        VvScope scope(t);
        vv_prepare(t, (int)batch);
        hipStream_t s = t->vamb->stream;
        t->perm.ensure((size_t)total);
        t->h_perm.ensure((size_t)total);
        int64_t bad = -1;
        for (int64_t i = 0; i < total; ++i) {
            const int64_t r = rows[i];
            if (r < 0 || r >= t->n) bad = r;
            t->h_perm.p[i] = r;
        }
        VH_REQUIRE(bad == -1, "row %lld out of range", (long long)bad);
        VH_HIP(hipMemcpyAsync(t->perm.p, t->h_perm.p, sizeof(int64_t) * total, hipMemcpyHostToDevice, s));
        vv_reset_epoch(t);
        for (int64_t b = 0; b < n_batches; ++b) vv_step(t, t->perm.p, false, false);
        VH_HIP(hipStreamSynchronize(s));
        if (metrics) vv_metrics(t, true, n_batches, batch, metrics);
    });
}

// The complete gradient (d of the summed loss / d parameter) of the last vh_vaevae_train_step: network 0 VAEVamb, 1 VAELabels,
// 2 VAEJoint.
int vh_vaevae_get_grad(vh_vaevae* t, int network, const char* name, float* data, int64_t n) {
    return guarded([&] {
        VH_REQUIRE(t != nullptr && data != nullptr, "NULL argument");
        VH_REQUIRE(network >= 0 && network <= 2, "network: 0 VAEVamb, 1 VAELabels, 2 VAEJoint");
        vh_vae* m = network == 0 ? t->vamb : (network == 1 ? t->labels : t->joint);
        VH_REQUIRE(m->bs > 0 && m->G.p != nullptr, "no training step has run yet");
        const int ti = find_tensor(m, name);
        const Tensor& tt = m->tensors[ti];
        VH_REQUIRE(tt.optimised, "'%s' is a buffer, not a parameter", name);
        VH_REQUIRE(n == tt.logical(), "parameter '%s' has %lld elements, got %lld", name, (long long)tt.logical(), (long long)n);
        download_padded(m, tt, m->G.p + tt.off, data);
    });
}

}  // extern "C"
