/*
 * dig_hip.h -- C ABI of libdig_hip.so: the MI355X (gfx950) implementation of DIGDriver's
 * mutation-rate + burden-test hot path.
 *
 * The reference (maxwellsh/DIGDriver) is pure Python and has no FFI layer; the entry points
 * below are the ones a binding for this path would need, one per vectorised reference
 * function (cited as file:line relative to the reference tree).  See INTEGRATION.md for the
 * ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C symbols, caller-owned buffers, no allocation ownership crosses the boundary;
 *   - every function returns 0 on success, a negative DIG_E* code on failure; the message of
 *     the last failure on the calling thread is available from dig_last_error();
 *   - functions WITHOUT the _host suffix take DEVICE pointers valid on the current HIP device
 *     and enqueue on `stream` (a hipStream_t passed as void*; NULL = default stream); they do
 *     not synchronise and do not allocate, so they may be captured into a hipGraph;
 *   - _host twins take HOST pointers, stage through device memory on `device`, run the same
 *     kernels and synchronise before returning;
 *   - NaN/inf inputs propagate to NaN outputs exactly as the reference's numpy/scipy
 *     expressions do; there is no CPU fallback: without a HIP device every call fails.
 *   - matrices are row-major; "[E, C]" means cohort index fastest.
 */
#ifndef DIG_HIP_H
#define DIG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIG_OK 0
#define DIG_EINVAL (-1)   /* bad argument */
#define DIG_EHIP (-2)     /* HIP runtime error */
#define DIG_ENODEV (-3)   /* no usable gfx950 device */

#define DIG_ABI_VERSION 12  /* 2: + join, contexts, scale factors, pipeline entry points; 3: + chunked suff-stats, tile front half, RBF passes;
                             * a statistics stage leaves its worklist length in the header; 4: + dig_element_pipeline_prepare / DIG_PIPE_COMPACT_L;
                             * 5: + dig_bin_records_pack, `bin_records` argument of dig_element_pipeline; 6: + dig_count_contexts2 (2-bit genome), dig_write_tsv_host;
                             * 7: + dig_mutation_file_*_host; 8: + dig_stage_timer_*; 9: + DIG_PIPE_RECORDS, dig_element_records_*;
                             * 12 (round 6): + dig_sort_rows, dig_bh_qvalues_ragged; - dig_element_pipeline_scaled*, dig_element_pipeline_pack_counts /
                             * DIG_PIPE_PACKED_COUNTS (round 5's measured losers: tools/rejected/r05_*.patch) */

/* dtype codes for dig_gather_bins */
#define DIG_F32 0
#define DIG_F64 1
#define DIG_I16 2
#define DIG_BF16 3

/* indices of the seven result planes written by dig_element_stats */
#define DIG_ES_EXP_SNV 0
#define DIG_ES_PVAL_SNV_BURDEN 1
#define DIG_ES_PVAL_SAMPLE_BURDEN 2
#define DIG_ES_THETA_INDEL 3
#define DIG_ES_EXP_INDEL 4
#define DIG_ES_PVAL_INDEL_BURDEN 5
#define DIG_ES_PVAL_MUT_BURDEN 6
#define DIG_ES_NPLANES 7

int dig_abi_version(void);
const char *dig_last_error(void);
/* number of HIP devices whose gcnArchName starts with "gfx950"; negative on HIP error */
int dig_device_count(void);

/* ---- negative-binomial tests (DIGDriver/sequence_model/nb_model.py) ------------------ */

/* nb_pvalue_greater_midp(k, alpha, p)  nb_model.py:271-278
 *   out[i] = 0.5 * nbinom.pmf(k[i]; alpha[i], p[i]) + betainc(k[i] + 1, alpha[i], 1 - p[i]) */
int dig_nb_midp_upper(const double *k, const double *alpha, const double *p, double *out, int64_t n, void *stream);
int dig_nb_midp_upper_host(const double *k, const double *alpha, const double *p, double *out, int64_t n, int device);

/* nb_pvalue_exact(k, alpha, p)  nb_model.py:298-314 (mu = alpha (1-p)/p) */
int dig_nb_exact(const double *k, const double *alpha, const double *p, double *out, int64_t n, void *stream);
int dig_nb_exact_host(const double *k, const double *alpha, const double *p, double *out, int64_t n, int device);

/* nb_pvalue_greater(k, alpha, p)  nb_model.py:243-256 */
int dig_nb_greater(const double *k, const double *alpha, const double *p, double *out, int64_t n, void *stream);
int dig_nb_greater_host(const double *k, const double *alpha, const double *p, double *out, int64_t n, int device);

/* nb_pvalue_midp(k, alpha, p)  nb_model.py:316-337 */
int dig_nb_midp_twosided(const double *k, const double *alpha, const double *p, double *out, int64_t n, void *stream);
int dig_nb_midp_twosided_host(const double *k, const double *alpha, const double *p, double *out, int64_t n,
                              int device);

/* Fisher combination  chi2.sf(-2 (ln p1 + ln p2), df=4)
 * DIGDriver/driver_model/transfer_tools.py:860-861,1086-1087; onthefly_tools.py:182-187 */
int dig_fisher(const double *p1, const double *p2, double *out, int64_t n, void *stream);
int dig_fisher_host(const double *p1, const double *p2, double *out, int64_t n, int device);

/* normal_params_to_gamma(mu, sigma)  nb_model.py:237-241: alpha = mu^2/sigma^2, theta = sigma^2/mu */
int dig_normal_params_to_gamma(const double *mu, const double *sigma, double *alpha, double *theta, int64_t n,
                               void *stream);
int dig_normal_params_to_gamma_host(const double *mu, const double *sigma, double *alpha, double *theta, int64_t n,
                                    int device);

/* ---- element statistics block (DIGDriver/driver_model/transfer_tools.py) ------------- *
 * One (element, cohort) pair per work item, dense [E, C]:
 *   ALPHA, THETA            load_pretrained_model            :17-19,46-48
 *   THETA *= cj[c]          transfer_element_model_with_indels :300
 *   EXP_SNV                 element_expected_muts_nb         :343-344
 *   PVAL_SNV_BURDEN         element_pvalue_burden_nb         :473-482
 *   PVAL_SAMPLE_BURDEN      element_pvalue_burden_nb_by_sample :594-615
 *   THETA_INDEL, EXP_INDEL, PVAL_INDEL_BURDEN  element_pvalue_indel :731-747
 *   PVAL_MUT_BURDEN         Fisher                           :1086-1087
 * mu_indel/sigma_indel may be NULL (= mu/sigma, the reference's indels_direct=False route,
 * genic_driver_tools.py:383-386).  pi_indel is [E] when pi_indel_per_cohort == 0, else [E, C].
 * out holds DIG_ES_NPLANES planes of E*C doubles: out[plane * E * C + e * C + c].
 * The same entry point serves the gene twins (transfer_tools.py:331-340,425-454,554-583,
 * 709-727): call it once per mutation class with that class's Pi_* / OBS_* / N_SAMP_*.
 * workspace: optional device scratch of at least dig_element_stats_workspace(E, C) bytes.  With it the
 * rare expensive tests (k > 64 or p-value < 1e-3: lgamma + continued fraction) are compacted into a
 * queue in LDS and finished by the same kernel at the end of each workgroup (what does not fit there goes through the
 * workspace); with workspace == NULL they are resolved inline (same results, more wave divergence).
 * The function never allocates. */
int64_t dig_element_stats_workspace(int64_t E, int64_t C);
int dig_element_stats(const double *mu, const double *sigma, const double *mu_indel, const double *sigma_indel,
                      const double *pi_sum, const double *pi_indel, int pi_indel_per_cohort, const int32_t *obs_snv,
                      const int32_t *obs_samples, const int32_t *obs_indel, const double *cj, const double *cj_indel,
                      double *out, int64_t E, int64_t C, void *workspace, int64_t workspace_bytes, void *stream);
int dig_element_stats_host(const double *mu, const double *sigma, const double *mu_indel, const double *sigma_indel,
                           const double *pi_sum, const double *pi_indel, int pi_indel_per_cohort,
                           const int32_t *obs_snv, const int32_t *obs_samples, const int32_t *obs_indel,
                           const double *cj, const double *cj_indel, double *out, int64_t E, int64_t C, int device);

/* ---- per-element accumulation (DIGDriver/sequence_model/genic_driver_tools.py) ------- *
 * nonc_model :300-431 (n_class = 1), genic_model :31-203 (n_class = 4: silent, missense,
 * nonsense, splice columns of L_data), tiled_nonc_model :599-690 (one bin per element), and the
 * loop body of DIG_onthefly (driver_model/onthefly_tools.py:109-164), for all C cohorts at once:
 *   MU = sum Y_PRED, SIGMA = sqrt(sum STD^2), R_OBS = sum Y_TRUE, FLAG = OR of FLAG over the
 *   element's overlapped bins (the reference adds numpy bools: logical OR, result 0/1)
 *                                                     get_region_params_direct :258-272
 *   region_counts = sum of the bins' 64 context counts, each repeated x3, reverse-complement
 *   permuted for '-' strand elements                  sequence_tools.py:630-634
 *   t_pi = d_pr / sum(region_counts * d_pr);  P = sum(t_pi * L)        :361-367
 *   R_SIZE = int(sum(region_counts)/3); ELT_SIZE = int(sum(L)/3); P_INDEL = ELT_SIZE/R_SIZE :375-381
 *   (gene_length != NULL: P_INDEL = gene_length / R_SIZE, genic_driver_tools.py:158-159)
 * Inputs (device): bin_mu, bin_std f64 [N, C]; bin_y i32 [N, C]; bin_flag u8 [N, C];
 *   bin_ctx i32 [N, 64] (contexts in sorted ACGT^3 order); ov_ptr i64 [E+1], ov_idx i32 [nnz]
 *   (CSR of overlapped bin rows, see dig_ideal_overlaps_host); L i32 [E, n_class, 192] in sorted
 *   substitution order ("XYZ>XaZ"); strand_minus u8 [E]; d_pr f64 [C, 192] = FREQ re-indexed by
 *   sorted substitution string (genic_driver_tools.py:321-325).
 * Outputs (device): MU, SIGMA f64 [E, C]; R_OBS, FLAG i32 [E, C]; P f64 [E, n_class, C];
 *   R_SIZE, ELT_SIZE i32 [E]; P_INDEL f64 [E].
 * workspace: device scratch of at least dig_accumulate_workspace(E, C) bytes, 256-byte aligned
 *   (strand-permuted context counts per element + transposed parameter tables).  With
 *   workspace == NULL a slower single-kernel LDS variant runs.  The function never allocates. */
int64_t dig_accumulate_workspace(int64_t E, int64_t C);
int dig_accumulate_elements(const double *bin_mu, const double *bin_std, const int32_t *bin_y,
                            const uint8_t *bin_flag, const int32_t *bin_ctx, const int64_t *ov_ptr,
                            const int32_t *ov_idx, const int32_t *L, int n_class, const uint8_t *strand_minus,
                            const int32_t *gene_length, const double *d_pr, double *MU, double *SIGMA,
                            int32_t *R_OBS, int32_t *FLAG, double *P, int32_t *R_SIZE, int32_t *ELT_SIZE,
                            double *P_INDEL, int64_t N, int64_t E, int64_t C, void *workspace,
                            int64_t workspace_bytes, void *stream);
int dig_accumulate_elements_host(const double *bin_mu, const double *bin_std, const int32_t *bin_y,
                                 const uint8_t *bin_flag, const int32_t *bin_ctx, const int64_t *ov_ptr,
                                 const int32_t *ov_idx, const int32_t *L, int n_class, const uint8_t *strand_minus,
                                 const int32_t *gene_length, const double *d_pr, double *MU, double *SIGMA,
                                 int32_t *R_OBS, int32_t *FLAG, double *P, int32_t *R_SIZE, int32_t *ELT_SIZE,
                                 double *P_INDEL, int64_t N, int64_t E, int64_t C, int device);

/* ---- accumulation + statistics block as one operation --------------------------------------- *
 * dig_element_pipeline == dig_accumulate_elements (n_class = 1) followed by dig_element_stats(MU, SIGMA, NULL, NULL,
 * P, P_INDEL [E], obs..., cj, cj_indel, out): the elementDriver / tiledModel route from bin tables to p-values
 * (genic_driver_tools.py:300-431 then transfer_tools.py:1069-1087).  One fusion across the two: MU / SIGMA / R_OBS /
 * FLAG are summed inside the statistics streaming kernel instead of being written by one kernel and read back by
 * the next.  All outputs of both operations are written; results are bit-identical to the two separate calls.
 * stages: DIG_PIPE_ALL, or the stages as separate calls in the order CONTEXTS, DOT, STATISTICS on the same workspace
 *   (only STATISTICS reads cj / cj_indel / obs_*: a caller can form the scale factors on another stream meanwhile,
 *   e.g. beside the MFMA-bound DOT kernel, and make `stream` wait for them before the statistics call).
 * workspace: dig_element_pipeline_workspace(E, C) bytes (0 = not available: E * C >= 2^32 - 1), 256-byte aligned. */
#define DIG_PIPE_CONTEXTS 1   /* acc_region_kernel: context rows, parameter table, R_SIZE */
#define DIG_PIPE_DOT 2        /* acc_dot_mfma_kernel: P, ELT_SIZE, P_INDEL */
#define DIG_PIPE_ACCUMULATE 3
#define DIG_PIPE_STATISTICS 4 /* MU, SIGMA, R_OBS, FLAG and the seven statistics planes */
#define DIG_PIPE_ALL 7
/* A call WITHOUT the CONTEXTS stage clears the worklist header of the statistics stage with a memset node of its own.
 * OR this flag into `stages` to skip that node when the caller knows the header is clear: after a CONTEXTS call on the
 * same workspace with no STATISTICS call since (a statistics stage leaves counts in the header: dword 2 the pairs it
 * finished from the global worklist, dword 3 the pairs the fused stream pass finished from its workgroups' LDS queues). */
#define DIG_PIPE_WORKLIST_CLEAN 8
/* Context-repeated L (ABI 4).  The reference builds the L_counts of every element / tile / on-the-fly region as its 64
 * trinucleotide context counts, each written to the three substitutions of that context (sequence_tools.py:560-564): then
 * sum(t_pi * L) runs over the same 64 per-context sums of d_pr as the denominator sum(region_counts * d_pr), and contexts +
 * dot become ONE kernel with half the matrix work (acc_dot_ctx_kernel; no acc_region_kernel, no parameter-table pass).
 *   dig_element_pipeline_prepare  PLAN TIME, once per element set: checks L[e, 3 j] == L[e, 3 j + 1] == L[e, 3 j + 2] for
 *       every element and context on the device, writes the compact [E, 64] counts into `workspace`, waits for `stream`
 *       and reports *compact_ok (host int) = 1 / 0.  Genic (n_class = 4) and --f-sites sets give 0.
 *   DIG_PIPE_COMPACT_L  OR into `stages` of dig_element_pipeline calls on a workspace prepared with compact_ok = 1 for
 *       this very L.  The CONTEXTS stage is then empty; the DOT stage is the fused kernel and clears the worklist header.
 *       Denominators are bit-identical to the general form, numerators differ by the rounding of a regrouped sum (P within
 *       a few ulp; tests/test_gpu_parity.py).  Without the flag the general K = 256 form runs, as before. */
#define DIG_PIPE_COMPACT_L 16
/* Record-major outputs of the statistics stage (ABI 9).  The stage has eleven outputs per (element, cohort) pair -- the seven
 * planes of dig_element_stats and MU, SIGMA, R_OBS, FLAG of the accumulation (transfer_tools.py:343-344,473-482,594-615,
 * 731-747,1086-1087; genic_driver_tools.py:404-417) -- i.e. eleven store streams whose tile starts fall on odd 8-byte
 * boundaries (E * C is rarely a multiple of 16).  With DIG_PIPE_RECORDS in `stages`, `out` is instead TILE-BLOCKED RECORDS:
 *     double out[ceil(E * C / 64)][DIG_REC_DOUBLES / 2][64][2]      (256-byte aligned)
 * the ten fields of pair i = e * C + c -- [DIG_ES_EXP_SNV .. DIG_ES_PVAL_MUT_BURDEN] the seven statistics in plane order,
 * [DIG_REC_MU], [DIG_REC_SIGMA], and at [DIG_REC_ROBS_FLAG] the two int32 R_OBS (low word) and FLAG (high word) -- live in
 * block i / 64: field f at out[i / 64][f / 2][i % 64][f % 2].  A lane of the kernel writes its ten fields as five 16-byte
 * pieces; a wave's five store instructions write 1 KB each, ONE aligned 5 120-byte run per 64-pair tile from one base
 * address, and nothing passes through LDS.  MU / SIGMA / R_OBS / FLAG arguments are not written (may be NULL); the lanes
 * past E * C of the last block are scratch.  Needs `bin_records`.  Values are bit-identical to the plane form.
 * dig_element_records_unpack turns the blocks into the plane form (out7 [7, E, C], MU, SIGMA [E, C] doubles, R_OBS, FLAG
 * [E, C] int32; any destination may be NULL; cohort_major: every plane as [C, E]). */
#define DIG_PIPE_RECORDS 32
#define DIG_REC_DOUBLES 10
#define DIG_REC_MU 7
#define DIG_REC_SIGMA 8
#define DIG_REC_ROBS_FLAG 9
int64_t dig_element_records_bytes(int64_t E, int64_t C);
int dig_element_records_unpack(const double *records, int64_t E, int64_t C, double *out7, double *MU, double *SIGMA,
                               int32_t *R_OBS, int32_t *FLAG, int cohort_major, void *stream);
int64_t dig_element_pipeline_workspace(int64_t E, int64_t C);
/* Stage timers (ABI 8; measurement only).  How long did the dot kernel / the statistics kernel of a dig_element_pipeline call
 * run?  Two events around a stage on the stream measure more than the kernel (each is a packet of its own: ~6 us, and the
 * packets change what runs beside the kernel on other streams).  A timer armed for DIG_PIPE_DOT or DIG_PIPE_STATISTICS makes
 * the NEXT launch of that stage's kernel by the calling thread record its own begin and end (hipExtLaunchKernelGGL: the
 * times of the dispatch itself, what rocprofv3's kernel trace shows; nothing is added to the stream).  dig_stage_timer_read
 * waits for that kernel and returns its duration in milliseconds; DIG_EINVAL if no such launch followed the arming. */
int dig_stage_timer_create(void **timer);
int dig_stage_timer_arm(void *timer, int stage);
int dig_stage_timer_read(void *timer, double *ms);
/* what the timer reads for a kernel that does nothing (one wave, launched here on `stream`): the part of a reading that is dispatch, not kernel */
int dig_stage_timer_selftest(void *timer, void *stream);
int dig_stage_timer_destroy(void *timer);
int dig_element_pipeline_prepare(const int32_t *L, int64_t E, int64_t C, void *workspace, int64_t workspace_bytes,
                                 int *compact_ok, void *stream);
/* host twin: all pointers in host memory; stages the arrays, checks L (the compact form runs when it repeats), runs
 * DIG_PIPE_ALL on the null stream and copies every output of accumulation and statistics back */
int dig_element_pipeline_host(const double *bin_mu, const double *bin_std, const int32_t *bin_y, const uint8_t *bin_flag,
                              const int32_t *bin_ctx, const int64_t *ov_ptr, const int32_t *ov_idx, const int32_t *L,
                              const uint8_t *strand_minus, const int32_t *gene_length, const double *d_pr,
                              const int32_t *obs_snv, const int32_t *obs_samples, const int32_t *obs_indel, const double *cj,
                              const double *cj_indel, double *MU, double *SIGMA, int32_t *R_OBS, int32_t *FLAG, double *P,
                              int32_t *R_SIZE, int32_t *ELT_SIZE, double *P_INDEL, double *out, int64_t N, int64_t E, int64_t C,
                              int device);
/* Packed bin records (ABI 5), PLAN TIME, once per set of bin tables: the statistics stage gathers Y_PRED, STD, Y_TRUE and
 * FLAG of every bin an element overlaps (get_region_params_direct, genic_driver_tools.py:258-272) -- four arrays, four
 * random row segments per bin.  dig_bin_records_pack rewrites them as {Y_PRED, STD^2} pairs (16 bytes) and
 * Y_TRUE | (FLAG != 0) << 31 (4 bytes) per (bin, cohort): two gathers per bin, the square of :266 taken once.  Pass the
 * records as `bin_records` of dig_element_pipeline (NULL: the four tables are read as before; they stay required either
 * way).  Same bits as without.  The call waits for `stream`; Y_TRUE must be >= 0.  Repack when a table changes. */
int64_t dig_bin_records_bytes(int64_t N, int64_t C);
int dig_bin_records_pack(const double *bin_mu, const double *bin_std, const int32_t *bin_y, const uint8_t *bin_flag, int64_t N,
                         int64_t C, void *records, int64_t records_bytes, void *stream);
int dig_element_pipeline(const double *bin_mu, const double *bin_std, const int32_t *bin_y, const uint8_t *bin_flag,
                         const int32_t *bin_ctx, const int64_t *ov_ptr, const int32_t *ov_idx, const int32_t *L,
                         const uint8_t *strand_minus, const int32_t *gene_length, const double *d_pr,
                         const int32_t *obs_snv, const int32_t *obs_samples, const int32_t *obs_indel, const double *cj,
                         const double *cj_indel, double *MU, double *SIGMA, int32_t *R_OBS, int32_t *FLAG, double *P,
                         int32_t *R_SIZE, int32_t *ELT_SIZE, double *P_INDEL, double *out, int64_t N, int64_t E, int64_t C,
                         const void *bin_records, int stages, void *workspace, int64_t workspace_bytes, void *stream);
/* ---- the gene route's statistics block as one launch (ABI 4) -------------------------------- *
 * gene_expected_muts_nb :331-340, gene_pvalue_burden_nb :394-456, gene_pvalue_burden_nb_by_sample :554-583,
 * gene_pvalue_indel :709-729 and the Fisher combination :860-861 of driver_model/transfer_tools.py, for G genes x C cohorts;
 * classes SYN, MIS, NONS, SPL, TRUNC = NONS + SPL, NONSYN = MIS + TRUNC.
 *   mu, sigma f64 [G, C] (ALPHA, THETA = normal_params_to_gamma; THETA *= cj[c]); mu_indel, sigma_indel: the indel pair or NULL
 *   pi f64 [G, n_pi, C]: n_pi = 6 (Pi_SYN .. Pi_NONSYN as load_pretrained_model hands them) or 4 (P_SILENT, P_MIS, P_NONS,
 *       P_SPLICE straight from dig_accumulate_elements(n_class = 4): TRUNC and NONSYN are added here)
 *   pi_indel f64 [G] or [G, C]; obs i32 [G, 5, C] = OBS_SYN, MIS, NONS, SPL, INDEL (TRUNC / NONSYN are added here);
 *   n_samp i32 [G, 6, C] = N_SAMP_c (distinct samples per class); cj, t_indel f64 [C] (t_indel: the indel calibration of
 *       :720-721, formed by the caller over its null gene set); with_indel = 0 leaves the four indel planes NaN.
 *   out f64 [22, G, C]: EXP_c x 6, PVAL_c_BURDEN x 6, PVAL_c_BURDEN_SAMPLE x 6, THETA_INDEL, EXP_INDEL, PVAL_INDEL_BURDEN,
 *       PVAL_MUT_BURDEN (Fisher of PVAL_TRUNC_BURDEN and PVAL_INDEL_BURDEN).
 * dig_gene_pipeline == dig_accumulate_elements(n_class = 4, gene_length) followed by dig_gene_stats(MU, SIGMA, NULL, NULL, P, 4,
 *   P_INDEL [G], ...): genic_model (genic_driver_tools.py:31-203) to p-values in one call; workspace as dig_accumulate_elements. */
int dig_gene_stats(const double *mu, const double *sigma, const double *mu_indel, const double *sigma_indel, const double *pi,
                   int n_pi, const double *pi_indel, int pi_indel_per_cohort, const int32_t *obs, const int32_t *n_samp,
                   const double *cj, const double *t_indel, int with_indel, double *out, int64_t G, int64_t C, void *stream);
int dig_gene_stats_host(const double *mu, const double *sigma, const double *mu_indel, const double *sigma_indel,
                        const double *pi, int n_pi, const double *pi_indel, int pi_indel_per_cohort, const int32_t *obs,
                        const int32_t *n_samp, const double *cj, const double *t_indel, int with_indel, double *out, int64_t G,
                        int64_t C, int device);
int dig_gene_pipeline(const double *bin_mu, const double *bin_std, const int32_t *bin_y, const uint8_t *bin_flag,
                      const int32_t *bin_ctx, const int64_t *ov_ptr, const int32_t *ov_idx, const int32_t *L,
                      const uint8_t *strand_minus, const int32_t *gene_length, const double *d_pr, const int32_t *obs,
                      const int32_t *n_samp, const double *cj, const double *t_indel, int with_indel, double *MU, double *SIGMA,
                      int32_t *R_OBS, int32_t *FLAG, double *P, int32_t *R_SIZE, int32_t *ELT_SIZE, double *P_INDEL, double *out,
                      int64_t N, int64_t G, int64_t C, void *workspace, int64_t workspace_bytes, void *stream);
int dig_gene_pipeline_host(const double *bin_mu, const double *bin_std, const int32_t *bin_y, const uint8_t *bin_flag,
                           const int32_t *bin_ctx, const int64_t *ov_ptr, const int32_t *ov_idx, const int32_t *L,
                           const uint8_t *strand_minus, const int32_t *gene_length, const double *d_pr, const int32_t *obs,
                           const int32_t *n_samp, const double *cj, const double *t_indel, int with_indel, double *MU,
                           double *SIGMA, int32_t *R_OBS, int32_t *FLAG, double *P, int32_t *R_SIZE, int32_t *ELT_SIZE,
                           double *P_INDEL, double *out, int64_t N, int64_t G, int64_t C, int device);

/* ---- sufficient statistics in canonical chunks (bin-sharded runs) --------------------------- *
 * Same quantity as dig_scale_suffstats / dig_scale_factors, defined so that it does not depend on the sharding: the bins
 * are cut into K canonical chunks of the GLOBAL grid (boundaries floor(N j / K)); a rank computes the chunk sums of the
 * chunks it owns (out_chunks [n_chunks, C], each in an order fixed by the chunk's rows and C alone), the chunk sums of
 * all ranks are all-gathered in chunk order and added first to last; obs f64 [world, 2, C] hold every rank's observed
 * SNV / indel counts (integers).  Bit-identical scale factors for any number of ranks.
 *   chunk_rows int64 [n_chunks + 1], HOST memory: first row of every chunk in this rank's table (+ end); n_chunks <= 256;
 *   C <= 256.  workspace: dig_scale_suffstats_chunked_workspace(chunk_rows, n_chunks, C) bytes.
 *   bin_flag may be NULL (ABI 5): bin_mu then holds +0.0 in the flagged entries already (a plan-time copy saves the flag
 *   bytes of every step); the same additions, the same bits. */
int64_t dig_scale_suffstats_chunked_workspace(const int64_t *chunk_rows, int n_chunks, int64_t C);
int dig_scale_suffstats_chunked(const double *bin_mu, const uint8_t *bin_flag, int64_t C, const int64_t *chunk_rows,
                                int n_chunks, double *out_chunks, void *workspace, int64_t workspace_bytes, void *stream);
int dig_scale_factors_chunked(const double *chunk_sums, int n_chunks, const double *obs, int world, int64_t C,
                              double *out_sum, double *cj, double *cj_indel, void *stream);

/* ---- per-base / tiled route, front half -------------------------------------------------- *
 * base_probabilities_by_region (sequence_model/sequence_tools.py:292-317) + the tiling of apply_nb_to_region
 * (sequence_model/nb_model.py:126-186) for trinucleotide contexts (n_up = n_down = 1), all cohorts at once; the back
 * half is dig_tiled_nb_test.  Genome layout as dig_count_contexts.
 *   region r: positions max(start, 1) .. min(end, chrom_len - 1) - 1 (fetch_sequence :21-29); tile t = positions
 *       first + t * binsize .. (the last tile of a region may be shorter); n_valid[r] = number of tiles the region has
 *       (<= n_tiles), first_pos[r] = its first position.
 *   s_prob f64 [C, 64]: S_prob of cohort c by context index 16 b0 + 4 b1 + b2; a position whose window holds a non-ACGT
 *       base has probability 0.
 *   pt f64 [C, R, n_tiles]: (sum of S_prob over the tile's positions) / (sum over the region's positions); tiles past
 *       n_valid[r] are NaN.  Sums are regrouped by context (exact integer counts x table): a few ulp from the reference's
 *       normalise-then-np.sum order.
 * dig_tile_mut_counts: k i32 [C, R, n_tiles] (cleared by the call) from (mutation, region) pairs as produced by
 *   dig_overlap_join_count/fill with the regions as blocks: a pair counts when the mutation's START is one of the
 *   region's positions (value_counts of START, nb_model.py:135-136,160-163); pairs whose mut_cohort is outside [0, C)
 *   are skipped. */
int dig_base_tile_probs(const uint32_t *genome_words, int64_t n_words, const int64_t *chrom_off, const int64_t *chrom_len,
                        int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start, const int64_t *reg_end, int64_t R,
                        const double *s_prob, int64_t C, int binsize, int64_t n_tiles, double *pt, int64_t *first_pos,
                        int32_t *n_valid, void *stream);
int dig_tile_mut_counts(const int32_t *pair_mut, const int32_t *pair_reg, int64_t n_pairs, const int64_t *mut_start,
                        const int32_t *mut_cohort, const int64_t *first_pos, const int32_t *n_valid, int binsize,
                        int64_t n_tiles, int64_t R, int64_t C, int32_t *k, void *stream);
/* General-context form (ABI 4): n_up = n_down = 1 (forwards to dig_base_tile_probs) or 2 -- penta-nucleotide contexts, the
 * DEFAULT signature of the reference's per-base functions (sequence_tools.py:292, nb_model.py:126,188).  s_prob f64
 * [C, 4^(2 n_up + 1)] by context index (itertools.product('ACGT', repeat = 2 n_up + 1) order); positions
 * (start == 0 ? n_up : start) .. min(end, chrom_len - n_up) - 1 (fetch_sequence :21-29).  A region of more than 16 384
 * positions is NOT evaluated by the device entry point (n_valid[r] = -1, its pt NaN; ABI 6 -- before, such a region
 * overran a staging buffer); a caller that holds the coordinates on the host refuses such regions, and regions that
 * start inside (0, n_up), before the launch (the _host twin does).  Everything else as dig_base_tile_probs. */
int dig_base_tile_probs_ctx(const uint32_t *genome_words, int64_t n_words, const int64_t *chrom_off, const int64_t *chrom_len,
                            int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start, const int64_t *reg_end, int64_t R,
                            const double *s_prob, int64_t C, int n_up, int binsize, int64_t n_tiles, double *pt,
                            int64_t *first_pos, int32_t *n_valid, void *stream);
int dig_base_tile_probs_ctx_host(const uint32_t *genome_words, int64_t n_words, const int64_t *chrom_off,
                                 const int64_t *chrom_len, int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start,
                                 const int64_t *reg_end, int64_t R, const double *s_prob, int64_t C, int n_up, int binsize,
                                 int64_t n_tiles, double *pt, int64_t *first_pos, int32_t *n_valid, int device);
/* host twins (n_mut: rows of mut_start / mut_cohort, so that the twin knows how much to stage) */
int dig_base_tile_probs_host(const uint32_t *genome_words, int64_t n_words, const int64_t *chrom_off,
                             const int64_t *chrom_len, int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start,
                             const int64_t *reg_end, int64_t R, const double *s_prob, int64_t C, int binsize, int64_t n_tiles,
                             double *pt, int64_t *first_pos, int32_t *n_valid, int device);
int dig_tile_mut_counts_host(const int32_t *pair_mut, const int32_t *pair_reg, int64_t n_pairs, const int64_t *mut_start,
                             int64_t n_mut, const int32_t *mut_cohort, const int64_t *first_pos, const int32_t *n_valid,
                             int binsize, int64_t n_tiles, int64_t R, int64_t C, int32_t *k, int device);

/* ---- per-cohort sufficient statistics for the scale factors --------------------------- *
 * calc_scale_factor_efficient, genome mode (driver_model/transfer_tools.py:148-156):
 *   out_sum[c] = sum over bins with FLAG == 0 of Y_PRED[bin, c]   (N_SNV_EXP per cohort);
 *   cj = N_SNV_OBS / out_sum, cj_indel = N_IND_OBS / out_sum are formed by the caller (after the
 *   all-gather of the per-shard sums when bins are sharded over GPUs).
 * bin_mu f64 [N, C], bin_flag u8 [N, C]; fixed summation order (bit-reproducible).
 * workspace: at least dig_scale_suffstats_workspace(N, C) bytes, 8-byte aligned. */
int64_t dig_scale_suffstats_workspace(int64_t N, int64_t C);
int dig_scale_suffstats(const double *bin_mu, const uint8_t *bin_flag, int64_t N, int64_t C, double *out_sum,
                        void *workspace, int64_t workspace_bytes, void *stream);
int dig_scale_suffstats_host(const double *bin_mu, const uint8_t *bin_flag, int64_t N, int64_t C, double *out_sum,
                             int device);

/* Scale factors from the (all-gathered) per-shard statistics, transfer_tools.py:153-154:
 *   parts f64 [world, 3, C]: row 0 = sum(Y_PRED[~FLAG]) of the shard, row 1 = its N_SNV_OBS, row 2 = its N_IND_OBS;
 *   cj[c] = sum_r parts[r][1][c] / sum_r parts[r][0][c],  cj_indel[c] = sum_r parts[r][2][c] / sum_r parts[r][0][c],
 *   both sums taken in rank order r = 0 .. world-1 (bit-reproducible on every rank).  world = 1: a plain division. */
int dig_scale_factors(const double *parts, int world, int64_t C, double *cj, double *cj_indel, void *stream);
/* Single-shard form of the two calls above (nothing to all-gather): out_sum as dig_scale_suffstats, and
 * cj[c] = n_snv_obs[c] / out_sum[c], cj_indel[c] = n_ind_obs[c] / out_sum[c] from the same final reduction kernel.
 * Same bits as dig_scale_suffstats + dig_scale_factors(world = 1).  Workspace as dig_scale_suffstats. */
int dig_scale_factors_local(const double *bin_mu, const uint8_t *bin_flag, int64_t N, int64_t C, const double *n_snv_obs,
                            const double *n_ind_obs, double *out_sum, double *cj, double *cj_indel, void *workspace,
                            int64_t workspace_bytes, void *stream);

/* ---- trinucleotide context counting from sequence ---------------------------------------- *
 * count_sequence_context over fetch_sequence (sequence_model/sequence_tools.py:21-29,42-55,65-80), for a batch of
 * regions: count_contexts_by_regions (:82-99), nonc_elt_context_count (:527-566), DIG_onthefly
 * (driver_model/onthefly_tools.py:70-71,120).
 *   genome_words u32 [n_words], 16-byte aligned (device entry point): 4 bits per base (A=0 C=1 G=2 T=3, anything else 4), 8 bases per word, base 0 in the
 *       low nibble; word 0 and word n_words-1 are all-N pad words; chromosome c occupies bases
 *       chrom_off[c] .. chrom_off[c] + chrom_len[c] - 1 counted from word 1, chrom_off[c] % 8 == 0.
 *   region r: centre positions max(start, 1) .. min(end, chrom_len - 1) - 1 of chromosome reg_chrom[r] (the fetch is
 *       widened by one base, START == 0 becomes 1, truncated at the chromosome end); a triplet holding a non-ACGT
 *       base is skipped; reg_minus[r] != 0 counts the reverse-complemented sequence ('-' strand elements).
 *   out i32 [R, 64], context index 16 b0 + 4 b1 + b2 (= itertools.product('ACGT', repeat=3) order).  Bit-exact. */
int dig_count_contexts(const uint32_t *genome_words, int64_t n_words, const int64_t *chrom_off,
                       const int64_t *chrom_len, int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start,
                       const int64_t *reg_end, const uint8_t *reg_minus, int64_t R, int32_t *out, void *stream);
int dig_count_contexts_host(const uint32_t *genome_words, int64_t n_words, const int64_t *chrom_off,
                            const int64_t *chrom_len, int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start,
                            const int64_t *reg_end, const uint8_t *reg_minus, int64_t R, int32_t *out, int device);

/* The same counts from the 2-BIT genome (ABI 6) -- half the bytes, no per-base test for unknown letters in the scan:
 *   words2 u32 [n_words2], 16-byte aligned: 2 bits per base (A=0 C=1 G=2 T=3; EVERY OTHER LETTER STORED AS A), 16 bases
 *       per word, base 0 in the low bits; array base g of chromosome position p is 64 + chrom_off[c] + p (64 pad bases
 *       = one 4-word group in front; chrom_off as above; at least 24 pad words behind the last chromosome).
 *   nint_start / nint_end i64 [n_int]: the maximal runs [start, end) of letters other than ACGT, in ARRAY bases, sorted,
 *       disjoint and not touching; nint_bucket i32 [n_buckets]: index of the first run that ends behind array base
 *       b << 12 (b = 0 .. n_buckets - 1, n_buckets > (last array base) >> 12).  n_int == 0: the three may be NULL.
 *   regions, strand flag and out exactly as dig_count_contexts; the same bits. */
int dig_count_contexts2(const uint32_t *words2, int64_t n_words2, const int64_t *nint_start, const int64_t *nint_end,
                        int64_t n_int, const int32_t *nint_bucket, int64_t n_buckets, const int64_t *chrom_off,
                        const int64_t *chrom_len, int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start,
                        const int64_t *reg_end, const uint8_t *reg_minus, int64_t R, int32_t *out, void *stream);
int dig_count_contexts2_host(const uint32_t *words2, int64_t n_words2, const int64_t *nint_start, const int64_t *nint_end,
                             int64_t n_int, const int32_t *nint_bucket, int64_t n_buckets, const int64_t *chrom_off,
                             const int64_t *chrom_len, int n_chrom, const int32_t *reg_chrom, const int64_t *reg_start,
                             const int64_t *reg_end, const uint8_t *reg_minus, int64_t R, int32_t *out, int device);

/* ---- result files (ABI 6; host code only) ------------------------------------------------- *
 * The text DataFrame.to_csv(path, header=True, index=True, sep="\t") writes for a frame (DigDriver.py:115-118): `header`
 * (a complete first line, no newline), then n_rows rows  label TAB col_0 TAB ... col_{n_cols-1}.
 *   labels: the row labels as one UTF-8 blob, label r = bytes label_off[r] .. label_off[r + 1] - 1;
 *   col_kind[j]: 0 = float64, written as Python's repr (shortest round-trip digits; positional for 1e-4 <= |x| < 1e16, else
 *       d.ddde-XX; NaN = empty field, inf / -inf), 1 = int64, 2 = bool as uint8 (True / False);
 *   n_threads: rows are formatted by up to 16 threads and written in order. */
int dig_write_tsv_host(const char *path, const char *header, const char *labels, const int64_t *label_off, int64_t n_rows,
                       int n_cols, const void *const *col_ptr, const int *col_kind, int n_threads);

/* ---- annotated mutation files (ABI 7; host code only) -------------------------------------- *
 * What mutation_tools.read_mutation_file (mutation_tools.py:45-104) + the encoding of the many-cohort driver do for one file
 * (tab-separated, no header: CHROM START END REF ALT SAMPLE GENE ANNOT [...]), without the interpreter:
 *   rows whose CHROM (one leading "chr" removed) is not "1" .. "22" are dropped; sample = id in order of first appearance
 *   among the kept rows; gene = id in order of first appearance in the file; indel = (ANNOT == "INDEL"); uid = dense rank of
 *   (CHROM, START, END, REF, ALT) among the kept rows (REF / ALT compared as ids of first appearance).
 * parse: *n_rows = kept rows, *n_samples, *names_bytes = length of the '\n'-joined sample names; *n_rows = -1 and no handle
 *   when the file holds something the parser does not cover (a double quote, a non-integer coordinate, ragged rows): the
 *   caller parses such a file as before.  fetch: copies the seven int64 columns and the names into the caller's buffers.
 *   free: releases the handle (NULL allowed). */
int dig_mutation_file_parse_host(const char *path, void **handle, int64_t *n_rows, int64_t *n_samples, int64_t *names_bytes);
int dig_mutation_file_fetch_host(void *handle, int64_t *chrom, int64_t *start, int64_t *end, int64_t *uid, int64_t *sample,
                                 int64_t *indel, int64_t *gene, char *sample_names);
/* flags (round 5): the reference's two de-duplications of a cohort's rows -- drop_duplicate_mutations, then get_unique_indels
 * (mutation_tools.py:106-117) -- as per-row 0 / 1 flags in file order: first_row[i] = no earlier row has the same mutation and
 * sample; first_indel[i] = row i is such a row, is an INDEL, and no earlier such row has the same mutation and GENE label.  The
 * genome-mode scale factors (calc_scale_factor_efficient, transfer_tools.py:129-159) count flagged rows: no sort on the device. */
int dig_mutation_file_flags_host(void *handle, int64_t *first_row, int64_t *first_indel);
int dig_mutation_file_free_host(void *handle);

/* ---- Benjamini-Hochberg q-values (nb_model.get_q_vals, nb_model.py:340-342 = statsmodels fdrcorrection, method 'indep') ---- *
 * For `rows` lists of n p-values each, every list ALREADY in ascending order (the caller sorts: torch.sort / rocPRIM; row r at
 * p_sorted + r n): q_sorted[i] = min(1, min_{j >= i} p[j] / ((j + 1) / n)), the same IEEE operations in the same order as the host
 * form (a NaN -- sorted last -- makes every q of its list NaN, as statsmodels does).  workspace: dig_bh_workspace(n, rows) bytes.  One HBM-bound pass (round 5: torch.cummin took 21 ms per 7.2 M values, 99 % of the
 * per-base route of BASELINE configs[4]). */
int64_t dig_bh_workspace(int64_t n, int64_t rows);
int dig_bh_qvalues_sorted(const double *p_sorted, int64_t n, int64_t rows, double *q_sorted, void *workspace, int64_t workspace_bytes,
                          void *stream);
/* Ranking p-values on the device (ABI 11, round 6; csrc/dig_sort.hip): a batched LSD radix sort written for this step (63-bit
 * keys in 9-bit digits, four passes over the upper 36 bits + a fix-up of the short runs that share them; ranges of 65 536
 * elements: histogram, scan, scatter -- no workgroup waits for another one) with the Benjamini-Hochberg pass behind it -- what
 * nb_model.get_q_vals (nb_model.py:340-342) needs for every cohort of the per-base route at once.
 *   rows: ragged lists in one array, row r = elements row_ptr[r] .. row_ptr[r + 1] - 1 (row_ptr: HOST array of rows + 1 offsets;
 *       a row holds fewer than 2^30 elements).  p, q, p_sorted, order: DEVICE arrays indexed like p.
 *   dig_sort_rows: p_sorted = the row's values ascending (every NaN last), order[j] = position in the row of the j-th smallest;
 *       either may be NULL.
 *   dig_bh_qvalues_ragged: q[i] = Benjamini-Hochberg q-value of p[i] among its row (statsmodels' operations, the bits of
 *       dig_bh_qvalues_sorted behind a stable sort; equal p-values have equal q-values, so the order among them is free).
 *       n_global, rank0, carry (HOST arrays of `rows`, each may be NULL): the row is the ranks rank0[r] + 1 .. of a list of
 *       n_global[r] values whose elements behind the row have the running minimum carry[r] (+inf when NULL) -- a rank of a
 *       sample sort finishes its range of the global order with them.  row_min (DEVICE, `rows`, may be NULL): the minimum of
 *       p / (rank / n) over the row, without carry (what the ranks in front take as their carry).  q may be NULL when only
 *       row_min is wanted.  sorted_out != 0: q leaves in ascending order of p instead of in place.
 *       The way back to the elements' places: q is a step function of p (one step per record of the reverse running minimum),
 *       so only the keys are sorted, the records go into a table per row and every element finds its q by its own value; a row
 *       with more records than half its length makes the call sort again with a 32-bit payload and write q through it.  The
 *       call synchronises `stream` (row tables go up, one flag word comes back).
 *   workspace: dig_bh_ragged_workspace(row_ptr, rows) bytes (24.2 bytes per element + tables). */
int64_t dig_bh_ragged_workspace(const int64_t *row_ptr, int64_t rows);
int dig_sort_rows(const double *p, const int64_t *row_ptr, int64_t rows, double *p_sorted, uint32_t *order, void *workspace,
                  int64_t workspace_bytes, void *stream);
int dig_bh_qvalues_ragged(const double *p, const int64_t *row_ptr, int64_t rows, const double *n_global, const int64_t *rank0,
                          const double *carry, double *q, double *row_min, int sorted_out, void *workspace, int64_t workspace_bytes,
                          void *stream);

/* get_ideal_overlaps(chrom, intervals, window)  genic_driver_tools.py:275-283, for a batch of
 * elements (host-side index construction, integer only): block b of element e covers bins
 * floor(start/w)*w ... ceil(end/w)*w; duplicates removed; rows are looked up in the sorted bin
 * table (bin_chrom, bin_start) of N rows.  Two-call protocol: with ov_idx == NULL only ov_ptr
 * (E+1) is filled so the caller can size ov_idx = ov_ptr[E].  A bin that is not in the table
 * is an error (the reference raises KeyError at df.loc, genic_driver_tools.py:265). */
int dig_ideal_overlaps_host(const int32_t *elt_chrom, const int64_t *blk_ptr, const int64_t *blk_start,
                            const int64_t *blk_end, int64_t E, int64_t window, const int32_t *bin_chrom,
                            const int64_t *bin_start, int64_t N, int64_t *ov_ptr, int32_t *ov_idx);

/* ---- mutation x element-block interval join (data_tools/mutation_tools.py:191-230) ----------- *
 * Replaces `bedtools intersect -wa -wb` of the mutation file with the bed6 blocks of the elements: half-open
 * overlap  m.start < b.end && b.start < m.end  on the same chromosome (a zero-length mutation is tested as
 * [start, start+1)).  Blocks are sorted by (chrom, start) and given as composite keys:
 *   blk_start_key[i] = chrom << 40 | start,  blk_runmax_key[i] = chrom << 40 | running max of end within the chrom,
 *   blk_end[i] = end.   Mutations: mut_chrom, mut_start, mut_end (int64, any order).
 * Two-call protocol: dig_overlap_join_count fills counts[n_mut] (overlapped blocks per mutation); the caller
 * forms the exclusive prefix sum `offsets`; dig_overlap_join_fill writes the pairs (mutation row, block row) at
 * offsets[m] ..., mutation-major with blocks ascending -- the order bedtools reports them in.  The integer
 * bookkeeping after the join (de-duplication, per-sample caps, blacklist; :208-226, :155-189) is sort/unique work. */
int dig_overlap_join_count(const int64_t *blk_start_key, const int64_t *blk_runmax_key, const int64_t *blk_end,
                           int64_t n_blk, const int64_t *mut_chrom, const int64_t *mut_start, const int64_t *mut_end,
                           int64_t n_mut, int32_t *counts, void *stream);
int dig_overlap_join_fill(const int64_t *blk_start_key, const int64_t *blk_runmax_key, const int64_t *blk_end,
                          int64_t n_blk, const int64_t *mut_chrom, const int64_t *mut_start, const int64_t *mut_end,
                          int64_t n_mut, const int64_t *offsets, int32_t *pair_mut, int32_t *pair_blk, void *stream);
/* host twins (n_pairs = offsets[n_mut - 1] + counts[n_mut - 1]: the size of pair_mut / pair_blk) */
int dig_overlap_join_count_host(const int64_t *blk_start_key, const int64_t *blk_runmax_key, const int64_t *blk_end,
                                int64_t n_blk, const int64_t *mut_chrom, const int64_t *mut_start, const int64_t *mut_end,
                                int64_t n_mut, int32_t *counts, int device);
int dig_overlap_join_fill_host(const int64_t *blk_start_key, const int64_t *blk_runmax_key, const int64_t *blk_end,
                               int64_t n_blk, const int64_t *mut_chrom, const int64_t *mut_start, const int64_t *mut_end,
                               int64_t n_mut, const int64_t *offsets, int64_t n_pairs, int32_t *pair_mut, int32_t *pair_blk,
                               int device);

/* ---- per-bin epigenomic-track gather (region_model/data_aux/mut_dataset.py:76-81) ---- *
 * out[b, l, t] = (float) x_data[bin_rows[b], l, tracks[t]]      (transpose_out == 0)
 * out[b, t, l] = ...                                            (transpose_out != 0: the
 *   channels-first layout SimpleMultiTaskResNet.forward makes with transpose(x, 1, 2),
 *   region_model/nets/cnn_predictors.py:131)
 * x_data is [N, L, T] of src_dtype (DIG_F32 | DIG_F64 | DIG_I16); out is DIG_F32 or DIG_BF16.
 * tracks == NULL selects all tracks in order (T_sel must equal T; the reference's default when no track file is
 * given, dataset_generator.py:52-55): a bin is then one contiguous block and is copied with 8 / 16-byte accesses. */
int dig_gather_bins(const void *x_data, int src_dtype, int64_t N, int64_t L, int64_t T, const int64_t *bin_rows,
                    int64_t B, const int32_t *tracks, int64_t T_sel, void *out, int out_dtype, int transpose_out,
                    void *stream);
int dig_gather_bins_host(const void *x_data, int src_dtype, int64_t N, int64_t L, int64_t T, const int64_t *bin_rows,
                         int64_t B, const int32_t *tracks, int64_t T_sel, void *out, int out_dtype, int transpose_out,
                         int device);

/* ---- per-base tiled NB test (nb_model.py:126-186, arithmetic of apply_nb_to_region) --- *
 * For cohort c, bin b, tile t:  alpha, theta = normal_params_to_gamma(mu[c,b], sigma[c,b]);
 *   p = 1/(pt * theta + 1);  pval = nb_pvalue_exact(k, alpha, p);  exp = pt * mu   (:141-178)
 * pt f64 [C or 1, n_bins, n_tiles] (pt_per_cohort selects), k i32 [C, n_bins, n_tiles],
 * mu, sigma f64 [C, n_bins]; pval, exp f64 [C, n_bins, n_tiles]. */
int dig_tiled_nb_test(const double *pt, int pt_per_cohort, const int32_t *k, const double *mu, const double *sigma,
                      double *pval, double *exp_out, int64_t C, int64_t n_bins, int64_t n_tiles, void *stream);
int dig_tiled_nb_test_host(const double *pt, int pt_per_cohort, const int32_t *k, const double *mu,
                           const double *sigma, double *pval, double *exp_out, int64_t C, int64_t n_bins,
                           int64_t n_tiles, int device);

/* ---- RBF cross-covariance of the sparse-GP calibration (gp_trainer.py:28-45: ScaleKernel(RBFKernel) between the m
 * inducing points and the n training rows) as single passes over the m x n matrix ----------------------------------- *
 * dig_rbf_cross: K[i][j] = outputscale * exp(-|z_i - x_j|^2 / (2 lengthscale^2)), Z f64 [m, d], X f64 [n, d], d <= 32,
 *   K f64 [m, n] row-major.
 * dig_rbf_backward: W = g o K and, per row and 1024-column chunk, partial[row][chunk] = {sum W, sum W d2}
 *   (dig_rbf_backward_partials(m, n) doubles; summed by the caller: d outputscale = sum W / outputscale,
 *   d lengthscale = sum W d2 / lengthscale^3, dZ = (W X - rowsum(W) o Z) / lengthscale^2). */
int dig_rbf_cross(const double *Z, const double *X, int64_t m, int64_t n, int64_t d, double lengthscale, double outputscale,
                  double *K, void *stream);
int64_t dig_rbf_backward_partials(int64_t m, int64_t n);
int dig_rbf_backward(const double *g, const double *K, int64_t m, int64_t n, double lengthscale, double outputscale,
                     double *W, double *partial, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DIG_HIP_H */
