/* vambhip.h -- C ABI of libvambhip.so, the MI355X (gfx950) hot path of Vamb's
 * VAE-train -> encode -> cluster pipeline.
 *
 * The reference (RasmussenLab/vamb v5.0.x) has no FFI for this path: its boundary is the Python class
 * surface `vamb.encode.VAE` / `vamb.encode.make_dataloader` / `vamb.cluster.ClusterGenerator`, whose
 * native arithmetic lives in torch, `dadaptation` and the Rust wheel `vambcore`.  This header is the
 * C boundary a maintainer binds instead (ctypes stub: vamb_amd/_lib.py).  Each entry
 * point names the reference code it replaces (file:line into /root/reference).
 *
 * Conventions
 *   - every function returns 0 (VH_OK) or a negative vh_status; vh_last_error() gives the message of
 *     the last failure on the calling thread.  No C++ exception crosses the boundary.
 *   - host pointers passed in are owned by the caller and only read/written during the call;
 *     all device memory is owned by the handle and released by *_destroy.
 *   - one host thread per handle; the library serialises its work on one HIP stream per handle.
 *   - float means IEEE binary32, row-major, C-contiguous.
 */
#ifndef VAMBHIP_H
#define VAMBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    VH_OK = 0,
    VH_ERR_INVALID = -1,  /* bad argument (the Python layer raises ValueError)        */
    VH_ERR_HIP = -2,      /* HIP runtime error (message carries hipGetErrorString)    */
    VH_ERR_NOMEM = -3,    /* device allocation failed                                 */
    VH_ERR_STATE = -4     /* call not valid in the handle's current state             */
} vh_status;

const char* vh_last_error(void);
const char* vh_version(void);

/* Process-wide options, read when a handle is created.  The library itself reads NO environment variables (the Python layer
 * forwards its VAMBHIP_* variables through these calls, vamb_amd/_lib.py).  Round 6 removed the measured-slower variants that
 * were kept "for the A/B" (stream-memory forks, the split optimiser, per-pass streams of the joint trainer, the transposed-copy
 * weight-gradient dataflow, the bias sums inside the weight-gradient GEMM, the two-buffer K loop, the non-DPP loss reductions,
 * the timing switches that produced wrong results): what is left selects between code paths the tests cross-check bit for bit,
 * or sets a policy.  Integer options:
 *   scan.reference_order (2)  evaluation order of the two float32 reductions behind every cluster decision, `matrix.matmul` and
 *                          `matrix.norm` (cluster.py:668, 674).  2 (default): the order measured on the reference's own torch /
 *                          oneMKL AVX-512 CPU build (oracle/probe_reference_order.py) -- the cluster stream equals the reference's
 *                          on every golden fixture; the tuned kernels evaluate the ascending fmaf chain as a filter and
 *                          re-evaluate in the reference's order only the pairs within the rounding slack of a decision boundary.
 *                          1: the same order on a plain one-pair-per-lane kernel (cross-check, ~3x the time per pass).
 *                          0: the ascending fmaf chain (the default of rounds 1-3; differs from the reference at near-ties)
 *   scan.column_loop (1)   0: runtime-width column loop in every scan kernel; 1: unrolled loads up to 8 medoids
 *   scan.mfma (1)          passes with more than 8 medoids on the matrix-pipe kernels
 *   scan.mfma_rowmajor (1) ... on the row-major copy of the matrix (K6r); 0 = the column-major kernel, its compiler-scheduled twin
 *                          (the start-up self-test's fallback, vh_selftest)
 *   scan.publish_split (1) a pass publishes the four summary words of every medoid first and the histograms behind a second sequence
 *                          flag (0: everything in one step); same results
 *   gen.speculate (1), gen.spec_window (16)   medoid statistics scanned ahead of need in the free slots of a pass
 *   gen.prefill (2)        the speculative fill of a pass collected one pass ahead, under the running pass
 *   gen.inline_removals (1)  rows of an emitted cluster are cleared by the NEXT scan's own prologue (kernel arguments) instead of
 *                          by a launch of their own; same stream
 *   gen.profile (0)        wall-clock breakdown of the native cluster state machine on stderr
 *   gen.gather_stage_bytes (64 MiB)  staging buffer of vh_gen_create_sharded's matrix gather
 *   vae.single_stream (0)  weight-gradient GEMMs on the main stream;  vae.fork_events (0)  forks as event records instead of kernel
 *                          completion signals (both: scheduling cross-checks, bit-identical)
 *   vae.fork_plan (-1)     two-stream schedule of the bf16 step, bit mask (csrc/vae.hip VaeTuning); -1 = by input width
 *   vae.dw_pair (1)        the last two weight gradients of a step as one launch;  vae.fused_skinny (1)  latent-wide products with
 *                          their elementwise consumer in one launch;  vae.fused_finalize (1)  the optimiser's scalar tail on the last
 *                          workgroup of the update kernel;  vae.prefetch_batch (1), vae.prefetch_max_cols (512)  the next batch
 *                          assembled during the running step;  vae.loss_from_dataset (1)  loss targets read from the dataset rows;
 *                          vae.loss_registers (1)  the bf16 loss kernel holds its rows in registers (0: staged in LDS; same bits)
 *   vae.gemm_prefetch (4), vae.gemm_kgroups (4)   fp32 GEMM: four K-tiles in flight / four K groups per workgroup for small launches
 *                          (1 = the plain tile: the self-test's fallback)
 *   vae.big_tiles (0), vae.xcd_remap (1), vae.dw_workgroups (256), vae.debug_timing (0)
 *   vae.probe_every (16)   the roofline probe (vh_vae_set_probe) times every n-th launch of the probed GEMM
 *   debug.guard_bytes (0)  diagnostic: canary bytes behind every device allocation (vh_debug_check_guards, vambhip_debug.h)
 * String options: comm.rccl_library (path of librccl), comm.rocm_path (default /opt/rocm). */
int vh_set_option(const char* name, int64_t value);
int vh_unset_option(const char* name);
int vh_get_option(const char* name, int64_t* value /* in: default, out: the value in force */);
int vh_set_option_string(const char* name, const char* value);
/* Start-up self-test of the kernels that rest on hand-counted waits behind inline-asm loads (csrc/selftest.hip): the row-major
 * many-medoid scan and the deep-prefetch / K-group tiles of the fp32 GEMM are run once beside their compiler-scheduled twins; a
 * disagreement switches the selecting option off for the process (scan.mfma_rowmajor = 0; vae.gemm_prefetch = vae.gemm_kgroups
 * = 1) and reports on stderr.  *fallbacks: bit 0 = scan fell back, bit 1 = GEMM tiles fell back.  ~40 ms; call once, before the
 * first handle is created (vamb_amd/_lib.py does).  No reference counterpart: a property of this build, not of Vamb. */
int vh_selftest(int* fallbacks);
/* number of visible HIP devices (0 with an error message when there is no GPU) */
int vh_device_count(int* n);
/* bind the calling process to a device (LOCAL_RANK under torch.distributed.run) */
int vh_set_device(int device);

/* =============================================================================================
 * Cluster scan  (replaces the torch/MKL + vambcore arithmetic inside vamb/cluster.py)
 * ============================================================================================= */
#define VH_NBINS 60
/* fixed-point scales of the exact integer accumulators (see DESIGN.md "defined-order arithmetic") */
#define VH_DENSITY_SCALE 65536.0
#define VH_HIST_SCALE 256.0

typedef struct vh_clu vh_clu;

/* Raw accumulators of one medoid scan.  density = density_fx / VH_DENSITY_SCALE rounded to float;
 * histogram[b] = hist_fx[b] / VH_HIST_SCALE rounded to float. */
typedef struct {
    int64_t density_fx;       /* sum len_i * (0.05f - d_i) over live rows with d_i <= 0.05f   (cluster.py:628-629) */
    int64_t hist_fx[VH_NBINS];/* length-weighted histogram of live d_i in [0, 0.3], torch.histogram edge rule (cluster.py:467-481) */
    int64_t n_within;         /* live rows with d_i <= 0.05f                                   (cluster.py:621-626) */
    int64_t n_lt;             /* live rows with d_i <  0.05f  -> loner test                    (cluster.py:457)     */
} vh_scan_result;

/* ClusterGenerator.__init__ (cluster.py:234-292): upload [n][L] latent rows + float32 lengths,
 * normalise on device unless `normalized` (cluster.py:653-669), keep the matrix resident in HBM in
 * column-major (SoA) form.  If normalized_out != NULL it receives the normalised row-major matrix
 * (the reference normalises the caller's array in place when destroy=True). */
int vh_clu_create(const float* matrix, const float* lengths, int64_t n, int L, int normalized,
                  float* normalized_out, vh_clu** out);
int vh_clu_destroy(vh_clu* h);
/* current number of physical rows (live + masked-out) */
int vh_clu_rows(vh_clu* h, int64_t* n_rows, int64_t* n_live);
/* medoids one vh_clu_scan pass takes for this handle's latent width: 32, fewer for very wide latent spaces (the
 * query vectors are staged in LDS; the reference has no limit on the latent width, cluster.py:204-222) */
int vh_clu_max_medoids(vh_clu* h, int* k);

/* sample_medoid (cluster.py:606-637) + head of find_threshold (cluster.py:452-481) for k medoids in
 * ONE pass over the matrix.  medoid_rows[j] is the physical row whose distance is forced to 0
 * (cluster.py:675), or -1 when that row lives in another shard.  queries == NULL: the query vectors
 * are the rows medoid_rows[j] of this handle; otherwise queries is a host [k][L] array. 1 <= k <= 32. */
int vh_clu_scan(vh_clu* h, int k, const int64_t* medoid_rows, const float* queries, vh_scan_result* out);

/* The scan also keeps, per medoid, the ascending rows within the medoid radius (sample_medoid's `cluster`
 * tensor, cluster.py:621-626) on the device for the last 16 scans.  vh_clu_scan_seq returns the sequence
 * number the NEXT scan will get; vh_clu_scan_list(h, seq, j, ...) returns medoid j's list of scan `seq`, or
 * *n_out = -1 when that scan has left the ring or the list overflowed its 2048 entries (then use
 * vh_clu_select with threshold 0.05f). */
/* ---- row-sharded clustering over RCCL (one process per GPU; rank order = global row order) -------------------------
 * vh_clu_attach_comm binds the communicator (vh_comm_create) whose ranks hold the other row shards.  A sharded pass is
 * issued by EVERY rank with the same medoids: local_rows[j] = the medoid's LOCAL row on its owner rank, -1 elsewhere.
 * On the handle's stream, without a host round trip in between: the owners' query vectors are all-reduced (sum with
 * zeros: exact), the shard is scanned, the exact int64 accumulators (density, 60 histogram bins, two counts per medoid)
 * are all-reduced -- integer sums are order-free, so `out` is identical on every rank and for any number of shards. */
typedef struct vh_comm vh_comm;
int vh_clu_attach_comm(vh_clu* h, vh_comm* comm);
int vh_clu_scan_sharded(vh_clu* h, int k, const int64_t* local_rows, vh_scan_result* out);
/* sharded cluster.py:_smaller_indices: local select (+ removal), all-gather of the counts, all-gather of the rows;
 * out_rows = GLOBAL rows, ascending (row_offsets[r] = first global row of rank r, world + 1 entries) */
int vh_clu_select_sharded(vh_clu* h, int64_t local_row, float threshold, int remove, const int64_t* row_offsets,
                          int64_t* out_rows, int64_t cap, int64_t* n_out);
int vh_clu_scan_seq(vh_clu* h, int64_t* seq);
int vh_clu_scan_list(vh_clu* h, int64_t seq, int j, int64_t* out_rows, int64_t cap, int64_t* n_out);

/* _smaller_indices (cluster.py:640-650) / the `within` list of sample_medoid (cluster.py:621-626):
 * ascending physical rows that are live and have d <= threshold (float32 compare).  Writes at most
 * cap indices, returns the true count in *n_out.  remove != 0 also clears their live flag
 * (kept_mask[point] = 0, cluster.py:308-309). */
int vh_clu_select(vh_clu* h, int64_t medoid_row, const float* query, float threshold, int remove,
                  int64_t* out_rows, int64_t cap, int64_t* n_out);
/* kept_mask[point] = 0 for explicit rows (cluster.py:308-309) */
int vh_clu_remove(vh_clu* h, const int64_t* rows, int64_t n);
/* pack (cluster.py:318-335; vambcore.overwrite_matrix, vambtools.py:291-321): order-preserving
 * compaction of matrix / lengths by the live mask; afterwards every row is live. */
int vh_clu_pack(vh_clu* h, int64_t* new_rows);
/* copy rows out (row-major [k][L]); rows == NULL: the whole physical matrix [n_rows][L] */
int vh_clu_get_rows(vh_clu* h, const int64_t* rows, int64_t k, float* out);
int vh_clu_get_kept(vh_clu* h, uint8_t* out /* [n_rows] */);
/* bytes of matrix + lengths + mask one scan pass reads (for roofline accounting) and the duration in
 * milliseconds of the last scan/select kernel measured with HIP events on the handle's stream */
int vh_clu_last_kernel_ms(vh_clu* h, float* ms);
int vh_clu_set_timing(vh_clu* h, int enable);


/* ---------------------------------------------------------------------------------------------
 * The host state machine of ClusterGenerator in native code: __next__ / find_cluster / wander_medoid /
 * find_threshold / get_next_seed / update_successes / the packing policy (cluster.py:294-604) on top of a
 * vh_clu handle, including CPython's random.Random(seed).sample (cluster.py:269,430,445).  `order` is
 * np.argsort(lengths)[::-1] (cluster.py:275; computed by the caller because numpy's unstable sort order is part
 * of the reference's behaviour).  The handle borrows `clu`; destroy it first.
 * --------------------------------------------------------------------------------------------- */
typedef struct vh_gen vh_gen;
typedef struct {
    int64_t medoid;        /* original contig index (Cluster.medoid) */
    int64_t seed;          /* index of the seed in the reference's packed matrix (Cluster.seed) */
    int64_t n_members;     /* 0: the generator is exhausted (StopIteration) */
    int32_t kind;          /* 0 normal, 1 loner, 2 fallback (Cluster.kind_str) */
    int32_t pad_;
    double maximal_pvr;    /* peak_valley_ratio when the cluster was emitted */
    double observed_pvr;   /* valid for kind 0 */
    double radius;         /* valid for kind 0 and 2 */
    int64_t successes;
    int64_t attempts;
    /* the generator's state AFTER this cluster (what the reference's attributes show to a caller between two
     * __next__ calls: peak_valley_ratio, successes, len(attempts), order_index) */
    double pvr_after;
    int64_t successes_after;
    int64_t attempts_after;
    int64_t order_index_after;
} vh_cluster_info;

int vh_gen_create(vh_clu* clu, const int64_t* order, int64_t n, int maxsteps, int windowsize, int minsuccesses,
                  uint64_t rng_seed, double pack_fraction, int64_t pack_min_rows, vh_gen** out);
/* The same state machine over a ROW-SHARDED matrix (BASELINE.json north_star: "the cluster seed search partitions across the 8
 * GPUs"; the reference has one process).  `clu` = this rank's shard with a communicator attached (vh_clu_attach_comm; rank
 * order = global row order); `order` = np.argsort(lengths)[::-1] of the GLOBAL lengths (n_global entries, identical on every
 * rank).  Collective: every rank calls it and then vh_gen_next in lock step, and receives the same clusters with GLOBAL
 * contig indices.  A pass = the shard's scan with explicit query vectors (every rank keeps a host copy of the whole normalised
 * matrix) + ONE all-gather of the exact integer accumulators and the within-radius list parts; the sums are order-free, so
 * the stream is bit-identical for any number of shards.  No control-plane traffic: all ranks take the same decisions. */
int vh_gen_create_sharded(vh_clu* clu, const int64_t* order, int64_t n_global, int maxsteps, int windowsize, int minsuccesses,
                          uint64_t rng_seed, double pack_fraction, int64_t pack_min_rows, vh_gen** out);
int vh_gen_destroy(vh_gen* g);
/* one Cluster (cluster.py:298-316, 545-604); members = original contig indices, ascending */
int vh_gen_next(vh_gen* g, vh_cluster_info* info, int64_t* members, int64_t cap);
/* up to max_clusters consecutive clusters in one call (the same state machine; it saves the binding's per-call overhead):
 * infos[i] describes cluster i, its members follow those of cluster i - 1 in `members` (a buffer of the generator's row count
 * always holds whatever is left).  *n_out < max_clusters: exhausted. */
int vh_gen_next_batch(vh_gen* g, int max_clusters, vh_cluster_info* infos, int64_t* members, int64_t cap, int* n_out);
/* the generator's mutable search state AFTER the last vh_gen_next (ClusterGenerator.peak_valley_ratio / successes /
 * len(attempts) / order_index, cluster.py:282-283, 386-413): what repr() and callers inspecting the attributes see */
int vh_gen_state(vh_gen* g, double* peak_valley_ratio, int64_t* successes, int64_t* attempts, int64_t* order_index);
/* accounting: passes over the matrix, medoids scanned, rows streamed, summed kernel time (when timing is on) */
/* Sum over all scan / select passes so far of the LIVE rows at the time of the pass (vh_gen_counters' rows_streamed counts
 * the resident rows, dead-but-uncompacted ones included): the N_live of SURVEY.md section 8d's algorithmic bytes. */
int vh_gen_live_rows(vh_gen* g, int64_t* live_rows_streamed);
int vh_gen_counters(vh_gen* g, int64_t* scan_passes, int64_t* scan_medoids, int64_t* rows_streamed,
                    double* kernel_ms, int64_t* n_emitted, int64_t* n_remaining);


/* =============================================================================================
 * VAE  (replaces the torch / dadaptation arithmetic inside vamb/encode.py)
 * ============================================================================================= */
#define VH_MAX_HIDDEN_LAYERS 8
#define VH_NTNF 103

typedef struct vh_vae vh_vae;

/* VAE.__init__ arguments after defaulting (encode.py:171-223).  Validation (ValueError cases of
 * encode.py:182-208) is repeated by the library and reported as VH_ERR_INVALID. */
typedef struct {
    int32_t nsamples;
    int32_t nlatent;
    int32_t nlayers;                          /* len(nhiddens) */
    int32_t nhiddens[VH_MAX_HIDDEN_LAYERS];
    float alpha;
    float beta;
    float dropout;
    uint64_t seed;                            /* seeds parameter init, dropout and reparameterisation noise */
} vh_vae_config;

int vh_vae_create(const vh_vae_config* cfg, vh_vae** out);
int vh_vae_destroy(vh_vae* h);

/* The two subclasses of VAE the semi-supervised / TaxVamb workflows train: the same stack of layers on
 * other input / reconstruction columns, with a label cross-entropy term.
 *   VH_VAE_CONCAT  semisupervised_encode.py:438-698 VAEConcat: columns depths | TNF | abundance | one-hot labels; calc_loss
 *                  (:515-569) = VAE.calc_loss + CrossEntropyLoss(label logits, class) with weight 1; trained by the inherited
 *                  VAE.trainmodel, i.e. D-Adapt-Adam.
 *   VH_VAE_LABELS  semisupervised_encode.py:189-436 VAELabels: the one-hot labels are the ONLY columns; calc_loss (:248-257) =
 *                  CrossEntropyLoss + KLD / (nlatent * beta), no per-contig weights; trainmodel (:362-436) uses
 *                  torch.optim.Adam(lr = lrate).  cfg->nsamples only documents the reference's `nlabels - 104`. */
enum { VH_VAE_PLAIN = 0, VH_VAE_CONCAT = 1, VH_VAE_LABELS = 2 };
enum { VH_OPT_DADAPT_ADAM = 0, VH_OPT_ADAM = 1 };
typedef struct {
    int32_t kind;        /* VH_VAE_* */
    int32_t nlabels;     /* width of the one-hot block: max(number of classes, 105), semisupervised_encode.py:25-47 */
    int32_t optimizer;   /* VH_OPT_* */
    float lrate;         /* Adam only */
} vh_vae_labels_config;
int vh_vae_create_labelled(const vh_vae_config* cfg, const vh_vae_labels_config* lab, vh_vae** out);
/* number of input (= reconstruction) columns of the model */
int vh_vae_row_width(vh_vae* h, int32_t* width);
/* forward() on rows given in the model's own column order ([batch][width], one-hot labels included): R = what `_decode`
 * returns, concatenated (softmax on the depths block when nsamples > 1, raw label logits), mu [batch][nlatent].
 * eps / masks as vh_vae_forward. */
int vh_vae_forward_rows(vh_vae* h, const float* X, int64_t batch, int training, const float* eps, const uint8_t* masks,
                        float* R_out, float* mu_out);
/* Label statistics of the most recent vh_vae_train_step / _epoch / _epochs call, per epoch: the mean over the batches of
 * CrossEntropyLoss (`ce_labels`) and the number of rows whose argmax(label logits) is their class (`correct_labels`,
 * semisupervised_encode.py:257, 563-569).  n_epochs must equal the epochs of that call (1 for a step / an epoch). */
int vh_vae_label_stats(vh_vae* h, int64_t n_epochs, double* out /* [n_epochs][2] */);

/* Parameters and buffers by their torch state_dict names (encode.py:226-249): e.g.
 * "encoderlayers.0.weight" [nh0][D], "encodernorms.1.running_var" [nh1], "mu.bias" [L],
 * "outputlayer.weight" [D][nh0], "...num_batches_tracked" [1] (as float).  n is the logical element
 * count (row-major); VH_ERR_INVALID if name or n does not match.  VAE.save / VAE.load (encode.py:486-541). */
int vh_vae_param_size(vh_vae* h, const char* name, int64_t* n);
int vh_vae_set_param(vh_vae* h, const char* name, const float* data, int64_t n);
int vh_vae_get_param(vh_vae* h, const char* name, float* data, int64_t n);
/* gradient of the last training step for a parameter (autograd's p.grad after loss.backward(), encode.py:418) */
int vh_vae_get_grad(vh_vae* h, const char* name, float* data, int64_t n);

/* The four tensors of make_dataloader's TensorDataset (encode.py:129-137), uploaded once and kept
 * resident: depths [n][nsamples], tnf [n][103], abundance [n][1], weights [n][1]. */
int vh_vae_set_dataset(vh_vae* h, const float* depths, const float* tnf, const float* abundance,
                       const float* weights, int64_t n);

/* One optimisation step = the body of trainepoch's batch loop (encode.py:390-425): forward, loss,
 * backward, DAdaptAdam.step (dadaptation==3.2, encode.py:578), on dataset rows `rows[0..batch)`.
 * eps   : NULL (device generator) or host [batch][nlatent] reparameterisation noise (encode.py:277)
 * masks : NULL (device generator) or host keep-masks, for every hidden layer in application order
 *         (encoder then decoder) a [batch][nhidden] uint8 block, concatenated
 * losses: out, the five means calc_loss returns (loss, ab_sse, ce, sse, kld) (encode.py:350-356) */
int vh_vae_train_step(vh_vae* h, const int64_t* rows, int64_t batch, const float* eps, const uint8_t* masks,
                      double losses[5]);
/* A whole epoch (encode.py:390-437): perm holds n_batches*batch dataset rows; the five per-batch means
 * are averaged over the batches exactly like the epoch log line.  One host sync per epoch.
 * perm == NULL: the library shuffles on the device (a fresh keyed bijection of [0, n) per epoch, the
 * counterpart of the DataLoader's RandomSampler, encode.py:33-50,129-144) -- no host permutation, no upload. */
int vh_vae_train_epoch(vh_vae* h, const int64_t* perm, int64_t n_batches, int64_t batch, double loss_means[5]);

/* VAE.forward on explicit host inputs (encode.py:306-314); training != 0 uses batch statistics,
 * dropout and noise like a torch module in train() mode (and updates the BatchNorm running stats).
 * Outputs are host arrays [batch][nsamples], [batch][103], [batch][1], [batch][nlatent]; any may be NULL. */
int vh_vae_forward(vh_vae* h, const float* depths, const float* tnf, const float* abundance, int64_t batch,
                   int training, const float* eps, const uint8_t* masks, float* depths_out, float* tnf_out,
                   float* abundance_out, float* mu_out);

/* VAE.encode (encode.py:442-484): eval-mode encoder over the resident dataset, low 12 mantissa bits
 * cleared (vambtools.py:324-330); latent is a host [n][nlatent] array. */
int vh_vae_encode(vh_vae* h, float* latent);

/* post-dropout activations dropout(leaky_relu(x W^T + b)) of hidden layer `layer` (encoder layers first) for the
 * batch of the last training-mode forward: [batch][nhidden] (tests: dropout statistics) */
int vh_vae_get_hidden(vh_vae* h, int layer, float* out, int64_t n);

/* D-Adapt-Adam group state (d, numerator_weighted, k) */
int vh_vae_opt_state(vh_vae* h, double* d, double* numerator_weighted, int64_t* k);
/* restore the group state (resuming a run whose moments were restored with vh_vae_set_opt_moments) */
int vh_vae_opt_set_state(vh_vae* h, double d, double numerator_weighted, int64_t k);
/* a fresh optimiser, as `DAdaptAdam(self.parameters(), decouple=True)` at the top of trainmodel (encode.py:578):
 * exp_avg = exp_avg_sq = s = 0, d = 1e-6, numerator_weighted = 0, k = 0 */
int vh_vae_reset_optimizer(vh_vae* h);
/* the optimiser of the following steps: VH_OPT_DADAPT_ADAM (VAE.trainmodel, encode.py:578) or VH_OPT_ADAM =
 * torch.optim.Adam(lr = lrate) with its default betas / eps (VAELabels.trainmodel, semisupervised_encode.py:405).  Both use
 * the exp_avg / exp_avg_sq buffers and the step count k; call vh_vae_reset_optimizer for a fresh state. */
int vh_vae_set_optimizer(vh_vae* h, int optimizer, float lrate);
/* per-parameter optimiser state by state_dict name: which = 0 exp_avg, 1 exp_avg_sq, 2 s */
int vh_vae_get_opt_moment(vh_vae* h, const char* name, int which, float* data, int64_t n);
int vh_vae_set_opt_moment(vh_vae* h, const char* name, int which, const float* data, int64_t n);

/* A dataset that outlives / is shared between VAE handles (one upload per `vamb bin default` run, however many
 * models are trained on it).  The handle must stay alive while a VAE uses it. */
typedef struct vh_dataset vh_dataset;
int vh_dataset_create(const float* depths, const float* tnf, const float* abundance, const float* weights, int64_t n,
                      int nsamples, vh_dataset** out);
int vh_dataset_destroy(vh_dataset* d);
int vh_vae_use_dataset(vh_vae* h, vh_dataset* d);
/* Labels of the semi-supervised loaders (semisupervised_encode.py:111-175: `np.unique(labels, return_inverse=True)[1]`,
 * one-hot to `nlabels` = max(classes, 105) columns by the collate functions): one int32 class per row.  The one-hot block
 * is produced by the batch gather on the device and never stored.  vh_dataset_set_labels adds them to a feature dataset
 * (make_dataloader_concat), vh_dataset_create_labels makes the labels-only dataset of make_dataloader_labels. */
int vh_dataset_set_labels(vh_dataset* d, const int32_t* labels, int64_t n, int32_t nlabels);
int vh_dataset_create_labels(const int32_t* labels, int64_t n, int32_t nlabels, vh_dataset** out);

/* ---- hierarchical label loss and the joint TaxVamb trainer (SURVEY.md 8f row N4 remainder) -----------------------------
 * vh_vae_set_hierarchy replaces Hierarchy(table_parent) + FlatSoftmaxNLL(tree) of VAELabelsHLoss / VAEConcatHLoss
 * (/root/reference/vamb/taxvamb_encode.py:326-330, 474-478; vamb/hloss_misc.py:20-124, 1102-1133): the labels of the model's
 * dataset become NODES of the taxonomy (table_parent[0] = -1, 0 <= table_parent[i] < i: the BFS order make_graph emits,
 * taxvamb_encode.py:29-61), the label loss is -log of the softmax mass -- over the FIRST n_leaves label logits -- on the leaves
 * at or below the row's node, and "correct predictions" is a constant 0 (taxvamb_encode.py:355).  fp32 step only. */
int vh_vae_set_hierarchy(vh_vae* h, const int32_t* table_parent, int32_t n_nodes);

/* The joint trainer replaces VAEVAE.trainepoch / trainmodel (/root/reference/vamb/semisupervised_encode.py:829-1084) as
 * VAEVAEHLoss uses them (taxvamb_encode.py:551-743): per batch seven passes through three networks (VAEVamb = a plain handle,
 * VAELabels = VH_VAE_LABELS, VAEJoint = VH_VAE_CONCAT; all fp32, all set to VH_OPT_ADAM), the sum of VAEVamb.calc_loss,
 * VAELabels.calc_loss and calc_loss_joint, one Adam step.  The handles stay usable on their own (encode, state_dict). */
typedef struct vh_vaevae vh_vaevae;
int vh_vaevae_create(vh_vae* vamb, vh_vae* labels, vh_vae* joint, vh_vaevae** out);
int vh_vaevae_destroy(vh_vaevae* t);
/* The row-aligned tensors of make_dataloader_semisupervised_hloss's TensorDataset (taxvamb_encode.py:192-239): `unsup` =
 * tensors 0-3 (features + weights), `unsup_labels` = tensor 4 (vh_dataset_create_labels), `sup` = tensors 5-9 (features +
 * weights + labels: vh_dataset_create + vh_dataset_set_labels).  Same number of rows. */
int vh_vaevae_set_datasets(vh_vaevae* t, vh_dataset* unsup, vh_dataset* unsup_labels, vh_dataset* sup);
/* The 17 values of trainepoch's log line, in its order (semisupervised_encode.py:830-848): loss_vamb, ab_vamb, ce_vamb,
 * sse_vamb, kld_vamb, loss_labels, ce_labels_labels, kld_labels, correct_labels_labels, loss_joint, ce_joint, sse_joint,
 * ce_labels_joint, kld_vamb_joint, kld_labels_joint, correct_labels_joint, loss.
 * train_step: one batch on rows[0..batch) (parity interface).  eps: NULL or 7 x [batch][nlatent] in pass order (joint, vamb_x,
 *   labels_x, vamb_u, vamb_s, labels_u, labels_s: the order in which the reference's step calls reparameterize); masks: NULL or
 *   the keep-masks of the same passes concatenated, each pass its hidden layers in application order ([batch][width] bytes per
 *   layer; the two _x passes run decoders only).
 * train_epoch: batch b takes rows[b*batch .. (b+1)*batch); metrics are the epoch means.  One host synchronisation. */
int vh_vaevae_train_step(vh_vaevae* t, const int64_t* rows, int64_t batch, const float* eps, const uint8_t* masks,
                         double metrics[17]);
int vh_vaevae_train_epoch(vh_vaevae* t, const int64_t* rows, int64_t n_batches, int64_t batch, double metrics[17]);
/* p.grad of the last vh_vaevae_train_step (after the single loss.backward(), :993): network 0 VAEVamb, 1 VAELabels, 2 VAEJoint */
int vh_vaevae_get_grad(vh_vaevae* t, int network, const char* name, float* data, int64_t n);

/* ---- make_dataloader on the device -------------------------------------------------
 * The matrix passes of vamb/encode.py:98-119 and vamb/vambtools.py:250-288 (column sums of the abundances, per-row
 * scaling + total + relative abundance, column z-score of the TNF block) run on the RAW matrices after ONE upload and
 * leave the normalised features in the resident matrix the VAE trains on; the O(n) / O(columns) vector work (1e6 / sums,
 * log + z-score of the totals, mean / std from the sums, contig weights) stays in numpy on the host, fed by these calls,
 * so every value is bit-identical to the reference's numpy result (summation orders reproduced, see csrc/prep.hip).
 * Call order: create -> upload -> column_sums(0) -> normalise_rows -> column_sums(1) -> column_sums(1, centre = mean)
 * -> zscore_tnf -> finish (which hands the handle's matrix over as a vh_dataset; destroy the vh_prep afterwards). */
typedef struct vh_prep vh_prep;
int vh_prep_create(int64_t n, int nsamples, vh_prep** out);
int vh_prep_destroy(vh_prep* p);
/* raw abundance [n][nsamples] and tnf [n][103], C-contiguous float32 */
int vh_prep_upload(vh_prep* p, const float* abundance, const float* tnf);
/* out[j] = sum over rows IN ROW ORDER (numpy's a.sum(axis=0)) of block 0 (depths, nsamples columns) or 1 (tnf, 103);
 * with centre != NULL the summand is (x - centre[j])^2 (numpy's _var) */
int vh_prep_column_sums(vh_prep* p, int block, const float* centre, float* out);
/* depths[r][j] *= scale[j]; totals[r] = numpy pairwise row sum (program: n_ops x {kind, start, len} int32, postfix:
 * kind 0 = leaf over [start, start + len), 1 = add); depths[r] /= totals[r], or = uniform where totals[r] == 0 */
int vh_prep_normalise_rows(vh_prep* p, const float* scale, const int32_t* program, int n_ops, float uniform,
                           float* totals);
/* tnf[r][c] = (tnf[r][c] - mean[c]) / stdev[c] */
int vh_prep_zscore_tnf(vh_prep* p, const float* mean, const float* stdev);
/* total_abundance [n] (log + z-scored on the host) and weights [n] complete the dataset */
int vh_prep_finish(vh_prep* p, const float* total_abundance, const float* weights, vh_dataset** out);
int vh_dataset_shape(vh_dataset* d, int64_t* n, int* nsamples);
/* host copies of the four tensors of make_dataloader's TensorDataset (any pointer may be NULL) */
int vh_dataset_download(vh_dataset* d, float* depths, float* tnf, float* total_abundance, float* weights);

/* ---- tetranucleotide frequencies: the step upstream of make_dataloader --------------------
 * vh_tnf_create takes the reference's projection kernel (vamb/kernel.npz, float32 [256][103], vamb/parsecontigs.py:9-15). */
typedef struct vh_tnf vh_tnf;
int vh_tnf_create(const float* kernel, vh_tnf** out);
int vh_tnf_destroy(vh_tnf* t);
/* vambcore.kmercounts (vamb/vambtools.py:444-447) for n sequences: sequence i = bases[offsets[i] .. offsets[i + 1]);
 * counts [n][256] uint32 (NULL: keep them on the device for vh_tnf_project).  4-mers containing a byte other than
 * A C G T (either case) are skipped (the definition test/test_vambtools.py:137-151 pins vambcore.kmercounts to). */
int vh_tnf_kmercounts(vh_tnf* t, const uint8_t* bases, const int64_t* offsets, int64_t n, uint32_t* counts);
/* Composition._project (vamb/parsecontigs.py:140-150) + mask_lower_bits(., mask_bits) (parsecontigs.py:211):
 * fourmers [n][256] float32 (raw counts), or NULL = the counts of the last vh_tnf_kmercounts call; tnf [n][103] */
int vh_tnf_project(vh_tnf* t, const float* fourmers, int64_t n, int mask_bits, float* tnf);

/* n_epochs consecutive epochs of the same shape with the device-side shuffle and a single host synchronisation
 * at the end (the loop of trainmodel, encode.py:598-601, between two batch-size changes).  global_batch = 0
 * without a communicator.  loss_means: [n_epochs][5] = the five means trainepoch logs per epoch. */
int vh_vae_train_epochs(vh_vae* h, int64_t n_epochs, int64_t n_batches, int64_t batch, int64_t global_batch,
                        double* loss_means);

/* Arithmetic of the dense contractions: 0 (default) = fp32 operands on the fp32 MFMA (BASELINE config C1);
 * 1 = operands rounded to bf16 while they are staged, bf16 MFMA with fp32 accumulation (configs C2-C4).
 * Tensors in memory, BatchNorm, loss and optimiser stay fp32 either way.  May be changed between calls. */
int vh_vae_set_precision(vh_vae* h, int bf16_operands);

/* Optional HIP-event probe around the forward GEMM of encoder layer `layer` (bench.py roofline):
 * after vh_vae_train_epoch, *ms_total is the summed duration of the probed launches and *launches
 * their number; *flops_per_launch = 2*batch*K*N of that GEMM. */
int vh_vae_set_probe(vh_vae* h, int enable, int layer);
int vh_vae_probe_result(vh_vae* h, double* ms_total, int64_t* launches, double* flops_per_launch);

/* =============================================================================================
 * Data-parallel training (one process per GPU, RCCL over xGMI).  The reference has no distributed
 * path; these entry points are what a multi-GPU host (vamb_amd/parallel.py) binds.
 * ============================================================================================= */
/* rank 0: 128 opaque bytes (ncclUniqueId) to hand to every rank out of band */
int vh_comm_unique_id(unsigned char* out128);
/* collective over all ranks: ncclCommInitRank on the calling process' current device */
int vh_comm_create(int rank, int world, const unsigned char* id128, vh_comm** out);
int vh_comm_destroy(vh_comm* c);
/* Host data plane: a communicator whose collectives are two caller-supplied functions on HOST memory (what a
 * torch.distributed gloo group offers); the library stages device buffers through pinned memory around them.  For running
 * and checking the multi-rank paths with several processes on ONE GPU (RCCL refuses two ranks per device) or where RCCL
 * cannot be loaded -- slow by construction.  allreduce(ctx, buf, count, dtype): in-place sum, dtype 0 = float32,
 * 1 = float64, 2 = uint64; allgather(ctx, send, recv, bytes): the ranks' blocks in rank order.  Both return 0 on success. */
typedef int (*vh_comm_allreduce_fn)(void* ctx, void* buf, int64_t count, int dtype);
typedef int (*vh_comm_allgather_fn)(void* ctx, const void* send, void* recv, int64_t bytes);
int vh_comm_create_host(int rank, int world, vh_comm_allreduce_fn allreduce, vh_comm_allgather_fn allgather, void* ctx,
                        vh_comm** out);
/* rank / world of a communicator, the rank count the collective library itself reports (ncclCommCount; = world on the host
 * plane) and whether it is an RCCL communicator.  Any output pointer may be NULL. */
int vh_comm_info(vh_comm* c, int* rank, int* world, int* reported_ranks, int* is_rccl);
/* hipDeviceSynchronize of the library's HIP runtime (the timing fence used by bench.py) */
int vh_device_synchronize(void);
/* Synchronised BatchNorm under data parallelism (default ON): the BatchNorm batch sums (forward: sum h, sum h^2;
 * backward: sum dy, sum dy*xhat) are all-reduced over the ranks, so the statistics span the ALL-RANK batch exactly as
 * the single-process reference computes them (encode.py:238,246,264).  0 = per-rank statistics (torch DDP without
 * SyncBatchNorm): 8 small all-reduces per step fewer, different mathematics. */
int vh_vae_set_syncbn(vh_vae* h, int enable);
/* From now on every optimisation step all-reduces (sum) the flat gradient over `comm` before the
 * D-Adapt-Adam update; comm == NULL detaches.  All ranks must hold identical parameters. */
int vh_vae_attach_comm(vh_vae* h, vh_comm* comm);
/* Epoch over this rank's shard: perm holds n_batches*batch LOCAL dataset rows; the loss of every
 * step is normalised by the all-rank batch (global_batch rows, global_wsum[b] = sum of the weights
 * of global batch b) so that the summed gradients equal the single-GPU gradient of the global batch
 * (BatchNorm statistics stay per-rank).  loss_means are the all-rank epoch means.
 * perm == NULL: device-side shuffle of this rank's shard; global_wsum == NULL with global_batch > batch: the
 * per-batch weight sums are computed on the device and all-reduced over the communicator.
 * global_batch <= 0: plain single-GPU epoch. */
int vh_vae_train_epoch_dp(vh_vae* h, const int64_t* perm, int64_t n_batches, int64_t batch, int64_t global_batch,
                          const float* global_wsum, double loss_means[5]);





#ifdef __cplusplus
}
#endif
#endif /* VAMBHIP_H */
