/* vambhip_debug.h -- test hooks and kernel diagnostics of libvambhip.so.
 *
 * NOT part of the drop-in boundary (include/vambhip.h): nothing the reference's interface needs is declared here.  These
 * entry points exist so that the parity tests can pin host-side pieces of the state machine in isolation (the threshold
 * walk, CPython's random.sample) and so that single GEMM instantiations can be checked and timed (tests/test_vae_gpu.py,
 * tools/gpu).  Same calling convention as vambhip.h: plain pointers and sizes, status code + vh_last_error(). */
#ifndef VAMBHIP_DEBUG_H
#define VAMBHIP_DEBUG_H

#include "vambhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook (host only): find_threshold (cluster.py:452-543) on exact histogram accumulators; kind 0 loner,
 * 1 no threshold, 2 threshold (then *threshold and *observed_pvr are set) */
int vh_debug_find_threshold(const int64_t* hist_fx, int64_t n_lt, double pvr, int* kind, double* threshold,
                            double* observed_pvr);

/* test hook (host only): n_calls consecutive random.Random(seed).sample(range(ns[i]), ks[i]) on one generator,
 * results concatenated into out (sum of ks entries) */
int vh_debug_pyrandom_sample(uint64_t seed, int n_calls, const int64_t* ns, const int64_t* ks, int64_t* out);

/* Diagnostic: run one GEMM instantiation on host data.  C[M][N] = sum_k A(m,k) B(n,k) (+bias[n] if
 * bias != NULL).  a_kc / b_kc: operand stored [rows][K] (1) or [K][rows] (0).  tile: 0 = 64x128,
 * 1 = 128x128, 2 = 128x32.  splits > 1 exercises the split-K slabs (summed on the host side of the call). */
int vh_debug_gemm(int tile, int a_kc, int b_kc, const float* A, const float* B, const float* bias, float* C,
                  int M, int N, int K, int splits, float* ms);

/* Diagnostic: one launch configuration of the bf16-storage GEMM (gemm_bf16.hpp: bf16 operands in memory, LDS-DMA
 * staging, bf16 MFMA with fp32 accumulation) on host data.  A [M][K] and B [N][K] are rounded to bf16 (nearest even) on
 * the host; C[m][n] = sum_k A[m][k] B[n][k].  epi: 0 = split-K fp32 slabs (summed on return), 1 = fp32 + bias,
 * 3 = hidden-layer epilogue without dropout: C = bf16(leaky_relu(acc + bias)) as float, CT (optional) the transposed
 * bf16 copy [N][M] as float, stats (optional) [2][N] the fp64 batch sums of C and C^2.  The tile is chosen from the
 * output shape as in the training step (variant 0) or forced (variant 1..6: the tile / pipeline variants listed in
 * csrc/vae_step16.hpp; + 256 * flags switches parts of the epilogue off for timing experiments, results then wrong).
 * *ms = average duration of `reps` back-to-back launches. */
int vh_debug_gemm16(int epi, const float* A, const float* B, const float* bias, float* C, float* CT, double* stats, int M,
                    int N, int K, int splits, int reps, int variant, float* ms);

/* Diagnostic: the row-major weight-gradient GEMM (gemm_bf16_tn.hpp; reference shape: dW = dZ^T In of a Linear layer's
 * backward, vamb/encode.py:226-249,419) on host data.  A [K][M] and B [K][N] are rounded to bf16 on the host;
 * C[m][n] = sum_k A[k][m] B[k][n] (split-K slabs summed on return); colsum (optional) [M] = sum over k < k_real of the
 * rounded A[k][m] (the fused bias gradient).  tile: 0 = by output shape as in the training step, 1 = 128x128 / 8 waves,
 * 3 = 64x128 / 4 waves; pipeline: 0 = two LDS buffers, 2 = three buffers with interleaved DMA issue.
 * *ms = average duration of `reps` back-to-back launches. */
int vh_debug_gemm16_tn(const float* A, const float* B, float* C, double* colsum, int M, int N, int K, int k_real, int splits,
                       int reps, int tile, int pipeline, float* ms);

/* Diagnostic: one launch of the bf16-storage GEMM on device-generated data with in-kernel time stamps (s_memtime ticks,
 * 100 MHz constant clock or shader clock -- compare differences only).  stamps[b][0..4] of workgroup b = kernel entry,
 * first K-tile landed, K loop done, epilogue phase 1 done (image in LDS), stores issued.  epi / variant as vh_debug_gemm16
 * (variant < 256).  Returns the number of workgroups in *n_blocks (<= cap_blocks rows are written). */
int vh_debug_gemm16_timeline(int epi, int M, int N, int K, int variant, unsigned long long* stamps, int cap_blocks,
                             int* n_blocks, float* ms);

/* Diagnostic: guard mode of the device allocator.  With the option debug.guard_bytes = g > 0 (vh_set_option, before the handles
 * are created) every device allocation is followed by g canary bytes; this call synchronises the device and lists the live
 * allocations whose canary was overwritten (text, one line each: allocation number, payload size, damaged range, the
 * allocation's call stack as offsets into the library).  *damaged = their number.  The tool for finding a kernel that stores
 * past the end of a buffer: GPU AddressSanitizer is not available on the test pool. */
int vh_debug_check_guards(char* report, int cap, int* damaged);

/* Diagnostic (a library built with -DVAMBHIP_TIMING_EXPERIMENTS only; VH_ERR_INVALID otherwise): constant-clock (100 MHz) stamps
 * of the LAST scan pass of a handle.  stamps[b][0..6] of workgroup b (b < 4095): kernel entry, prologue done, first row block
 * evaluated (VALU kernels), row loop done, drain done, flush begun, flush retired; row 4095 = the publish kernel (entry, copies
 * added up, results in host memory).  host_us[0..3]: scan_core entry, scan launched, publish launched, flag seen (microseconds
 * on the host clock).  Rows are zero where no workgroup ran.  cap_rows >= 4096. */
int vh_debug_scan_timeline(vh_clu* h, unsigned long long* stamps, int cap_rows, int* n_rows, double* host_us);

#ifdef __cplusplus
}
#endif
#endif /* VAMBHIP_DEBUG_H */
