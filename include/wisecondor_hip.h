/*
 * wisecondor_hip.h -- C ABI of the MI355X (gfx950) implementation of WISECONDOR's
 * numeric hot path: `newref` reference-bin selection and the `test` path
 * (PCA-apply, masked z-score repeats, Stouffer window segmentation).
 *
 * The reference (VUmcCGP/wisecondor) is pure Python/numpy and has no FFI of its
 * own; each entry point below replaces one numpy function of the reference and
 * cites it as file:line under the upstream tree.  INTEGRATION.md shows the
 * ctypes stub a maintainer would add to wisetools.py to bind them.
 *
 * Conventions
 *   - plain C, no C++/torch types; every array is a dense row-major buffer.
 *   - functions ending in `_dev` take DEVICE pointers and enqueue work on the
 *     given hipStream_t (passed as void*; NULL = default stream) without
 *     synchronising; the others take HOST pointers, copy, run and synchronise.
 *   - return value 0 = success, negative = WC_E_* below; wc_last_error() gives
 *     a message.  Nothing here falls back to a CPU implementation.
 *   - IEEE specials propagate as in the reference (np.seterr('ignore'),
 *     wisetools.py:34): NaN/inf inputs yield NaN/inf outputs, never a trap.
 */
#ifndef WISECONDOR_HIP_H
#define WISECONDOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WC_OK 0
#define WC_E_ARG (-1)      /* invalid argument (shape, range, NULL)            */
#define WC_E_HIP (-2)      /* a HIP runtime call failed                        */
#define WC_E_LIMIT (-3)    /* size beyond a documented implementation limit    */
#define WC_E_INTERNAL (-4) /* internal consistency check failed                */

/* Summation order of the float64 distance, which numpy derives from the memory
 * layout of correctedData (wisetools.py:302): a C-contiguous [bins, samples]
 * array is reduced row by row with numpy's pairwise summation; the
 * Fortran-contiguous array that np.load returns for a prep file (trainPCA
 * returns corrected.T, wisetools.py:101) is reduced sample by sample, i.e. a
 * plain left-to-right sum.  The buffers handed to this library are always
 * row-major [bins, samples]; the flag only selects the rounding order.        */
#define WC_SUM_PAIRWISE 0
#define WC_SUM_SEQUENTIAL 1

typedef struct wc_ctx wc_ctx; /* per-device context: workspaces, scratch, counters */

/* ---- context -------------------------------------------------------------- */
wc_ctx *wc_create(int device);
void wc_destroy(wc_ctx *ctx);
const char *wc_last_error(void);
const char *wc_version(void);
/* counters of the last newref call: [0] rows finished by the fast path,
 * [1] rows that took the exact fallback path, [2] symmetric tiles launched,
 * [3] sample columns M, [4] candidates re-scored in float64 (sum over rows)   */
int wc_newref_stats(wc_ctx *ctx, int64_t out[8]);

/* ---- newref: reference-bin selection -------------------------------------- */
/* getPart, wisetools.py:358-361: rows [start,end) of zero-based part p of n.  */
void wc_get_part(int64_t partnum, int64_t outof, int64_t bincount, int64_t *start, int64_t *end);

/*
 * getReference + getRefForBins, wisetools.py:298-325 and :364-398.
 * corrected      [n_bins, n_samples] float64 (the prep file's correctedData)
 * chrom_bins     [n_chrom] bins per chromosome (maskedChromBins); sum == n_bins
 * k              selectRefAmount (reference default 100)
 * sum_order      WC_SUM_PAIRWISE / WC_SUM_SEQUENTIAL (see above)
 * row_begin/end  target rows to produce (getPart of part, splitParts)
 * idx_out        [row_end-row_begin, k] int32: positions in the
 *                "all bins not on the target's chromosome" concatenation
 *                (wisetools.py:386-387), -1 padded
 * dist_out       [row_end-row_begin, k] float64 ascending, 1e10 padded
 * Results equal the reference bit for bit (stable (distance, position) order,
 * numpy pairwise-summed float64 distances).  k <= 1024; above 256 every row is
 * scanned exactly (the candidate lists are sized for refsize <= 256): slower, same bits.
 */
int wc_get_reference(wc_ctx *ctx, const double *corrected, int64_t n_bins, int64_t n_samples,
                     const int64_t *chrom_bins, int n_chrom, int k, int sum_order,
                     int64_t row_begin, int64_t row_end, int32_t *idx_out, double *dist_out);
int wc_get_reference_dev(wc_ctx *ctx, void *stream, const double *corrected, int64_t n_bins,
                         int64_t n_samples, const int64_t *chrom_bins_host, int n_chrom, int k,
                         int sum_order, int64_t row_begin, int64_t row_end, int32_t *idx_out,
                         double *dist_out);

/*
 * Multi-GPU building blocks of the same computation (one process per GPU; the
 * host exchanges `thr` / candidate lists with RCCL between the calls).
 *   stage A  wc_newref_prepare_dev    centre + float16 operand image + norm bounds (all rows)
 *   stage B  wc_newref_thresholds_dev per-row admission thresholds for rows
 *            [row_begin,row_end) from a fixed pseudo-random column sample
 *   stage C  wc_newref_collect_dev    symmetric MFMA distance tiles
 *            `tile_rank` of `tile_ranks` (round-robin), appending candidates
 *            whose lower-bound distance passes the row threshold
 *   stage D  wc_newref_finish_dev     float64 re-score in numpy order, stable
 *            sort, certificate, exact fallback -> idx/dist for the row range
 * wc_get_reference_dev == A, B(all), C(0 of 1), D(range).
 */
int wc_newref_prepare_dev(wc_ctx *ctx, void *stream, const double *corrected, int64_t n_bins,
                          int64_t n_samples, const int64_t *chrom_bins_host, int n_chrom, int k,
                          int sum_order);
int wc_newref_thresholds_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end);
/* copy thresholds of rows [row_begin,row_end) out of / into the context (device float[rows]) */
int wc_newref_get_thresholds_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end, float *out);
int wc_newref_set_thresholds_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end, const float *in);
/* the error interval every decision of the fast path rests on (device float[rows] each): a listed
 * key of the pair (i, j) obeys  key <= distance(i, j) <= key + slack[i] + slack[j];  lo[i] is the
 * row's lower norm bound (key = lo[i] + lo[j] - 2 dot).  For tests of the bound itself. */
int wc_newref_get_bounds_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end, float *lo_out,
                             float *slack_out);
int wc_newref_collect_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                          int tile_rank, int tile_ranks);
/* candidate-list exchange: pack the lists of rows [row_begin,row_end) into
 * dst_cnt int32[rows] / dst_list u64[rows, dst_cap] (device), and merge lists
 * received from another rank into this context's lists for those rows.  A source
 * row with more than `cap` entries marks the row for the exact fallback path.  */
int64_t wc_newref_list_capacity(wc_ctx *ctx);
int wc_newref_export_lists_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                               int64_t dst_cap, int32_t *dst_cnt, uint64_t *dst_list);
int wc_newref_import_lists_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                               int64_t src_cap, const int32_t *src_cnt, const uint64_t *src_list);
int wc_newref_finish_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                         int32_t *idx_out, double *dist_out);
/* stage D in its two halves, for callers that time them apart: the per-row fast path, then
 * the exact path for the rows it handed over (same arguments; finish == rescore + fallback) */
int wc_newref_rescore_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                          int32_t *idx_out, double *dist_out);
int wc_newref_fallback_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                           int32_t *idx_out, double *dist_out);
/* the fast path's own two kernels apart (rescore == pick + rescore_pairs): candidate selection with
 * the certificate, then the exact float64 distances and their order (wisetools.py:302, 305-324) */
int wc_newref_pick_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                       int32_t *idx_out, double *dist_out);
int wc_newref_rescore_pairs_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end,
                                int32_t *idx_out, double *dist_out);
/* The exact path for EVERY row of [row_begin, row_end) of a prepared job (wc_newref_prepare_dev): float64
 * distances to all candidates in numpy's order and the stable selection of wisetools.py:305-321, without
 * matrix cores, bounds or candidate lists.  It is what a row takes when its certificate fails and what
 * refsize > 256 runs for every row; exported so that callers (the full-size tests) can hold the fast path
 * against it row by row. */
int wc_newref_exact_dev(wc_ctx *ctx, void *stream, int64_t row_begin, int64_t row_end, int32_t *idx_out,
                        double *dist_out);
/* measurement helper: microseconds a chain of n dependent empty kernel launches takes on `stream`
 * (mean over reps) -- the launch floor bench.py prices the one-sample `test` latency against */
int wc_launch_floor_us(wc_ctx *ctx, void *stream, int n, int reps, double *out);

/*
 * Native sample ingest and result output for many files (replaces the
 * np.load loop of wisecondor.py:75-80 / 193-196 and the np.savez_compressed of wisecondor.py:270-280
 * for batches).  Host only, plain threads, no GPU.
 *
 * wc_read_samples: file i of `paths` (a converted sample: members `sample` = pickled dict
 *   chromosome -> integer array, `arguments` = pickled dict with 'binsize') -> row i of counts_out
 *   (int32 [n_files, row_stride]): chromosomes '1'..'n_chrom', each padded with zeros / truncated to
 *   chrom_sizes[c] (toNumpyRefFormat, wisetools.py:268-274), bins merged when to_binsize is a whole
 *   multiple of the file's own bin size (scaleSample, wisetools.py:220-237; to_binsize <= 0: keep).
 *   binsize_out[i] = the file's own bin size.  status[i]: WC_OK, or WC_NPZ_* -- then the row is
 *   undefined and the caller reads that file the slow way (np.load), which also words the error.
 * wc_write_test_results: one `test` output file per row (keys / dtypes / shapes of the reference's,
 *   SURVEY.md App. B): arguments (args_npy[i]: the ready-made .npy member bytes, a pickled dict),
 *   runtime (shared bytes), binsize (int64 when binsize_is_int -- the reference stores the Python value it read
 *   from the reference file unchanged --, else float64), results_r / results_z (object arrays of n_chrom float64 arrays cut
 *   from row i of r / z at chrom_sizes), results_cwz [n_sel], results_calls [n, 5] (shape (0,) when
 *   empty), threshold_z, asdef, aasdef = asdef * threshold_z.  level: zlib level of the members
 *   (0 = stored).  status[i]: WC_OK or WC_NPZ_IO.
 * Both return WC_OK unless an argument is unusable; per-file outcomes are in status.
 */
#define WC_NPZ_UNSUPPORTED 1   /* not a file this reader understands (layout, pickle opcode, dtype) */
#define WC_NPZ_IO 2            /* could not read / write the file */
#define WC_NPZ_BINSIZE 3       /* the file's bin size cannot be scaled to to_binsize */
int wc_read_samples(const char *const *paths, int n_files, int n_threads, const int64_t *chrom_sizes, int n_chrom,
                    double to_binsize, int32_t *counts_out, int64_t row_stride, double *binsize_out, int *status);
/* chromosome lengths of every file at to_binsize (int64 [n_files, n_chrom]) and its own bin size: what
 * `newref`'s toNumpyArray needs to size the dense matrix (per-chromosome maximum over the samples,
 * wisetools.py:243-250) before wc_read_samples fills it */
int wc_read_sample_lengths(const char *const *paths, int n_files, int n_threads, int n_chrom, double to_binsize,
                           int64_t *lengths_out, double *binsize_out, int *status);
int wc_write_test_results(int n_files, int n_threads, const char *const *out_paths,
                          const unsigned char *const *args_npy, const int64_t *args_len,
                          const unsigned char *runtime_npy, int64_t runtime_len, double binsize, int binsize_is_int,
                          double threshold_z,
                          const int64_t *chrom_sizes, int n_chrom, const double *z, const double *r,
                          int64_t row_stride, const double *cwz, int n_sel, const double *calls,
                          const int32_t *n_calls, int max_calls, const double *asdef, int level, int *status);

/*
 * newref prep (upstream of the hot path): toNumpyArray's
 * normalisation + all-zero-bin mask (wisetools.py:240-264) and trainPCA
 * (wisetools.py:89-101) as a deterministic exact PCA (float64 Gram matrix on the GPU's float64
 * matrix cores, its [samples, samples] eigenproblem by the direct solver of csrc/eigh.hip on the GPU --
 * wc_newref_prep_eig; callers may also solve the fetched matrix with their own LAPACK --, everything
 * bins-sized on the GPU).
 *   counts [n_samples, n_total_bins] int32: per chromosome padded with zeros to
 *          chromosome_bins[c] (the longest sample, wisetools.py:245-250)
 *   mask_out [n_total_bins], masked_chrom_bins_out [n_chrom], *n_masked_out = B
 *   masked_data_out [B, n_samples]; corrected_t_out [n_samples, B] (the reference's
 *   correctedData is its transpose VIEW, i.e. Fortran-ordered [B, n_samples]);
 *   pca_components_out [n_comp, B]; pca_mean_out [B].
 * Call once with the four data outputs NULL to learn B, then again with buffers.
 * wc_newref_prep is the one-call form.  The steps are also exported: _gram builds the Gram
 * matrix G [n_samples, n_samples] of the centred data in HBM (and copies it to gram_out unless
 * that is NULL), _eig solves its eigenproblem on the GPU for the n_pairs leading pairs
 * (Householder tridiagonalisation, Sturm multisection, inverse iteration: a direct method,
 * 3 <= n_samples <= 4096, n_pairs <= 8; eigenvalues descending, unit eigenvectors as rows, host
 * outputs), _finish takes such pairs -- from _eig or from the caller's own LAPACK on gram_out.
 * wc_sym_eigh_leading_dev is the same solver on any device-resident symmetric float64 matrix
 * (left untouched).
 */
int wc_newref_prep_gram(wc_ctx *ctx, const int32_t *counts, int64_t n_samples, int64_t n_total_bins,
                        const int64_t *chromosome_bins, int n_chrom, uint8_t *mask_out,
                        int64_t *masked_chrom_bins_out, int64_t *n_masked_out, double *gram_out);
int wc_newref_prep_eig(wc_ctx *ctx, int n_pairs, double *eigvals_out, double *eigvecs_out);
int wc_sym_eigh_leading_dev(wc_ctx *ctx, const double *matrix_dev, int64_t n, int n_pairs, double *eigvals_out,
                            double *eigvecs_out);
int wc_newref_prep_finish(wc_ctx *ctx, int n_comp, const double *eigvecs, const double *eigvals,
                          double *masked_data_out, double *corrected_t_out, double *pca_components_out,
                          double *pca_mean_out);
/* Device-resident finish: the bins-sized results stay in HBM for wc_newref_*_dev (no 2 x B x S x 8 B
 * trip through the host).  corrected_bs_dev [n_masked, n_samples] row-major device memory holding the
 * values of the reference's Fortran-ordered correctedData (wisetools.py:101: pass WC_SUM_SEQUENTIAL
 * to newref); masked_dev [n_masked, n_samples] device memory; pca_components_out / pca_mean_out host
 * memory.  Every output may be NULL. */
int wc_newref_prep_finish_dev(wc_ctx *ctx, int n_comp, const double *eigvecs, const double *eigvals,
                              double *masked_dev, double *corrected_bs_dev, double *pca_components_out,
                              double *pca_mean_out);
int wc_newref_prep(wc_ctx *ctx, const int32_t *counts, int64_t n_samples, int64_t n_total_bins,
                   const int64_t *chromosome_bins, int n_chrom, int n_comp, uint8_t *mask_out,
                   int64_t *masked_chrom_bins_out, int64_t *n_masked_out, double *masked_data_out,
                   double *corrected_t_out, double *pca_components_out, double *pca_mean_out);

/* ---- test: per-reference state -------------------------------------------- */
typedef struct wc_reference wc_reference;

/*
 * Everything toolTest derives from the reference file alone
 * (wisecondor.py:177-201): uploads indexes/distances/PCA, computes
 * getOptimalCutoff(distances, cutoff_repeats) (wisetools.py:328-336) on the
 * GPU and the per-bin reference lists `index[i][distances[i] < cutoff]`
 * (wisetools.py:424) once, since they are sample independent.
 *   indexes [n_bins,k] int32, distances [n_bins,k] float64,
 *   chromosome_sizes/masked_sizes [n_chrom] int64, mask [sum(chromosome_sizes)] uint8,
 *   pca_mean [n_bins], pca_components [n_comp, n_bins] float64.
 * cutoff_override: NULL to compute the cutoff, else the value to use (the
 * reference passes it explicitly to repeatTest, wisetools.py:438).
 * Limits: k <= 1024 (lists above 128 entries take a slower generic kernel), n_comp <= 8.
 */
wc_reference *wc_reference_create(wc_ctx *ctx, const int32_t *indexes, const double *distances,
                                  int64_t n_bins, int k, const int64_t *chromosome_sizes,
                                  const int64_t *masked_sizes, int n_chrom, const uint8_t *mask,
                                  const double *pca_mean, const double *pca_components, int n_comp,
                                  int cutoff_repeats, const double *cutoff_override);
void wc_reference_destroy(wc_reference *ref);
double wc_reference_cutoff(const wc_reference *ref);

/* getOptimalCutoff alone, wisetools.py:328-336 (host pointers). */
int wc_optimal_cutoff(wc_ctx *ctx, const double *distances, int64_t count, int repeats, double *cutoff);
/* ... with the function's second return value (wisetools.py:332, :336): mask [count] uint8 =
 * `distances < cutoff of the iteration before the last` (all finite values when repeats == 1).
 * repeats >= 1; the reference's repeats == 0 case (an all-zero float array) is host marshalling. */
int wc_optimal_cutoff_mask(wc_ctx *ctx, const double *distances, int64_t count, int repeats, double *cutoff,
                           uint8_t *mask);

/* applyPCA alone, wisetools.py:104-113, for a batch of already normalised and
 * masked vectors: samples/out [n_samples, n_bins] float64 (host pointers).    */
int wc_apply_pca(wc_ctx *ctx, const double *samples, int64_t n_samples, int64_t n_bins,
                 const double *pca_mean, const double *pca_components, int n_comp, double *out);

/*
 * toNumpyRefFormat + applyPCA, wisetools.py:267-278 and :104-113, for a batch.
 * counts [n_samples, sum(chromosome_sizes)] int32: per chromosome already
 * padded/truncated to the reference length (host marshalling of the dict).
 * out    [n_samples, n_bins] float64 (PCA-corrected, masked, unit-sum).
 * raw    optional [n_samples, n_bins] float64: the vector before PCA.
 */
int wc_prepare_samples(wc_ctx *ctx, const wc_reference *ref, const int32_t *counts,
                       int64_t n_samples, double *out, double *raw);

/*
 * repeatTest/trySample, wisetools.py:407-448, for a batch of samples.
 * data [n_samples, n_bins] float64 -> z, r, ref_sizes [n_samples, n_bins]
 * float64 and sd_avg [n_samples] (stdDevAvg).  Means and standard deviations
 * are summed in numpy's pairwise order.
 */
int wc_repeat_test(wc_ctx *ctx, const wc_reference *ref, const double *data, int64_t n_samples,
                   double threshold, int repeats, double *z, double *r, double *ref_sizes,
                   double *sd_avg);

/*
 * stdDevAvg of trySample alone (wisetools.py:428-435): the mean of the non-NaN per-bin standard
 * deviations, summed bin by bin like the reference's Python loop (sequential float64 rounding).
 * sd [n_samples, n_bins] host -> out [n_samples].  The sum runs as an exact parallel scan
 * (testpath.hip, k_sd_fast); *serial_samples (optional) receives how many samples needed the
 * serial chain instead.
 */
int wc_std_dev_avg(wc_ctx *ctx, const double *sd, int64_t n_samples, int64_t n_bins, double *out,
                   int32_t *serial_samples);

/*
 * fillTri / fillTriMin (wisetools.py:466-487) + TriArr.segmentTri (triarray.py:59-84)
 * on a batch of independent regions without materialising the triangle.
 * z [total] float64: concatenated regions; region_offsets [n_regions+1].
 * min_effect != 0 enables fillTriMin's filter: a window keeps its value only if
 * abs(median(ratio[x..y]) - 1) >= min_effect (ratio laid out like z; may be NULL
 * when min_effect == 0).
 * Outputs per region: whole-region Stouffer z (getValue(0,n-1),
 * wisecondor.py:237) and up to max_calls segments (value, x, y inclusive) in
 * ascending position order; n_calls[region] holds the number found (if it
 * exceeds max_calls the call fails with WC_E_LIMIT).
 */
int wc_stouffer_segments(wc_ctx *ctx, const double *z, const double *ratio, double min_effect,
                         const int64_t *region_offsets, int64_t n_regions, double threshold,
                         int min_search, int max_calls, double *region_z, int32_t *n_calls,
                         double *call_value, int32_t *call_x, int32_t *call_y);

/*
 * The numeric content of toolTest (wisecondor.py:199-268) for a batch:
 * prepare -> repeatTest -> minrefbins cleaning -> Stouffer segmentation ->
 * call coordinate mapping and median effect -> inflated per-bin outputs.
 *   counts        [n_samples, n_total_bins] int32 (see wc_prepare_samples)
 *   chromosomes   [n_sel] 1-based chromosome numbers to segment (-chromosomes)
 *   results_z/r   [n_samples, n_total_bins] float64 (r is ratio-1), zeros at
 *                 masked / removed bins
 *   results_cwz   [n_samples, n_sel]
 *   calls         [n_samples, max_calls, 5] rows [chrom, start, end, z, effect]
 *   n_calls       [n_samples]
 *   asdef         [n_samples]
 * min_effect is -mineffectsize (0 = the reference default, no filter).
 */
int wc_test_batch(wc_ctx *ctx, const wc_reference *ref, const int32_t *counts, int64_t n_samples,
                  double threshold, int min_ref_bins, int repeats, double min_effect,
                  const int32_t *chromosomes, int n_sel, int max_calls, double *results_z,
                  double *results_r, double *results_cwz, double *calls, int32_t *n_calls,
                  double *asdef);

/* Device-resident variant used by bench.py: counts already on the GPU, outputs
 * stay on the GPU (any output pointer may be NULL to skip it).                */
int wc_test_batch_dev(wc_ctx *ctx, void *stream, const wc_reference *ref, const int32_t *counts,
                      int64_t n_samples, double threshold, int min_ref_bins, int repeats,
                      double min_effect, const int32_t *chromosomes_host, int n_sel, int max_calls,
                      double *results_z, double *results_r, double *results_cwz, double *calls,
                      int32_t *n_calls, double *asdef);

/*
 * Optional stage timing of wc_test_batch_dev for the measurement harness (bench.py): with
 * profiling enabled every call records HIP events between its stages on the launch stream.
 * wc_test_profile_read synchronises the device and returns, for the LAST call, milliseconds of
 *   [0] prepare (toNumpyRefFormat + applyPCA)   [1] z-score repeats (repeatTest)
 *   [2] reshaping, inflation, cleaning          [3] Stouffer segmentation (fillTri + segmentTri)
 *   [4] call mapping and outputs                [5] of [3]: the certificate + window-search launches
 * and the work those launches executed: [6] windows evaluated by the search kernel (4 float64
 * operations each), [7] window / bound evaluations of the quiet-job certificate.
 */
int wc_test_profile(wc_ctx *ctx, int enable);
/* Development aid: phase stamps (shader clock) of the latency-mode kernels' workgroup
 * `block_plus_one - 1` (0: off); returns the 64 stamps of the calls since the last request. */
int wc_debug_times(wc_ctx *ctx, int block_plus_one, unsigned long long *out64);
int wc_test_profile_read(wc_ctx *ctx, double out[8]);

#ifdef __cplusplus
}
#endif
#endif /* WISECONDOR_HIP_H */
