/*
 * secedo_simmat.h -- C-ABI of the MI355X similarity-matrix path.
 *
 * Drop-in boundary for the reference's
 *     Matd computeSimilarityMatrix(pos_data, num_cells, max_fragment_length, group_id_to_pos,
 *                                  mutation_rate, homozygous_rate, seq_error_rate, num_threads,
 *                                  marker, normalization)
 * (reference: similarity_matrix.hpp:51-60, implementation similarity_matrix.cpp:295-433, sole
 * caller spectral_clustering.cpp:354-356). Plain pointers and sizes only; no C++ or torch types.
 * include/secedo_simmat.hpp keeps the C++ signature on top of these entry points, and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Flat pileup layout (replaces std::vector<std::vector<PosData>>, sequenced_data.hpp:11-47):
 *   chr_locus_off[n_chr+1]  loci of chromosome c are [chr_locus_off[c], chr_locus_off[c+1])
 *   locus_pos[L]            PosData::position                      (sequenced_data.hpp:26)
 *   locus_entry_off[L+1]    entries of locus l are [off[l], off[l+1])
 *   read_ids[E]             PosData::read_ids                      (sequenced_data.hpp:28)
 *   id_base16[E]            PosData::group_ids_bases, group_id<<2|base in 16 bits (:29-37), OR
 *   id_base32[E]            the same packing in 32 bits for more than 16383 groups
 *                           (exactly one of the two pointers is non-NULL)
 *
 * Every function returns SECEDO_OK (0) or a negative SECEDO_E_* code; the message of the last
 * error on the calling thread is available from secedo_simmat_last_error().
 * There is no CPU fallback: without a usable HIP device every compute entry point fails with
 * SECEDO_E_NO_DEVICE.
 */
#ifndef SECEDO_SIMMAT_H
#define SECEDO_SIMMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SECEDO_OK 0
#define SECEDO_E_INVALID_ARG (-1)
#define SECEDO_E_INVALID_NORMALIZATION (-2) /* reference: std::logic_error, similarity_matrix.cpp:264 */
#define SECEDO_E_NO_DEVICE (-3)
#define SECEDO_E_HIP (-4)
#define SECEDO_E_STATE (-5)
#define SECEDO_E_LIMIT (-6)

/* enum class Normalization (reference: similarity_matrix.hpp:9-17) */
#define SECEDO_NORM_ADD_MIN 0
#define SECEDO_NORM_EXPONENTIATE 1
#define SECEDO_NORM_SCALE_MAX_1 2

/* to_enum (reference: similarity_matrix.cpp:256-266): "ADD_MIN" | "EXPONENTIATE" | "SCALE_MAX_1"
 * -> SECEDO_NORM_*, anything else -> SECEDO_E_INVALID_NORMALIZATION. */
int secedo_simmat_normalization_from_string(const char *name);

const char *secedo_simmat_last_error(void);
const char *secedo_simmat_version(void);
/* Number of HIP devices visible to this process (0 when there is none; never fails). */
int secedo_simmat_device_count(void);

/* ------------------------------------------------------------------------------------------
 * One-shot entry point: host buffers in, host matrix out. This is what the C++ shim calls.
 * Replaces computeSimilarityMatrix (similarity_matrix.cpp:295-433) including normalize
 * (:271-293). `out` is caller-allocated, num_cells*num_cells doubles, row-major (the layout of
 * Mat<double>, util/mat.hpp:17-37). num_threads is the reference's num_threads: it does not
 * set any parallelism here, it is the semantic input of the flush rule (similarity_matrix.cpp:
 * 354-356: a batch of completed reads is compared once 4*num_threads of them are complete).
 * Devices: secedo_simmat_set_devices() below, else the environment's SECEDO_GPUS, else SECEDO_DEVICE
 * (default: device 0). With more than one device the N x N output is block-partitioned across them
 * behind this same call (north_star): one host thread per device, every device packs the
 * pileup, takes a contiguous range of upper-triangular tiles, pulls the others' tiles over xGMI
 * (hipMemcpyPeerAsync, chunk by chunk behind the accumulation), normalises its block of rows and
 * downloads it into its rows of `out`. Integer accumulators: the matrix is bit-identical for any
 * number of devices.
 * ---------------------------------------------------------------------------------------- */
int secedo_simmat_compute(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
                          const uint64_t *locus_entry_off, const uint32_t *read_ids,
                          const uint16_t *id_base16, const uint32_t *id_base32,
                          const uint32_t *group_id_to_pos, uint32_t n_groups, uint32_t num_cells,
                          uint32_t max_fragment_length, double mutation_rate,
                          double homozygous_rate, double seq_error_rate, uint32_t num_threads,
                          int normalization, double *out);
/* The devices secedo_simmat_compute (and so the C++ shim with the reference's signature, whose argument list
 * has no room for them) spreads one matrix over: device_ids[0 .. n_devices), each < device_count(); a device
 * may be listed more than once (rehearsal of the N-device path on fewer GPUs: every entry is a lane of its
 * own). n_devices = 0 returns to the environment: SECEDO_GPUS = a count N (devices 0 .. N - 1) or a
 * comma-separated list of ids; else SECEDO_DEVICE; else device 0. Process-wide; at most 16 devices.
 * get_devices writes up to `capacity` ids and returns how many devices the next call would use (< 0: error). */
int secedo_simmat_set_devices(const int *device_ids, uint32_t n_devices);
int secedo_simmat_get_devices(int *device_ids, uint32_t capacity);

/* Pinned host staging for callers that assemble the flat pileup themselves (the C++ shim flattens the
 * reference's vector<vector<PosData>> straight into it, several threads at a time): five buffers of at least
 * bytes[0..4] bytes -- for chr_locus_off, locus_pos, locus_entry_off, read_ids, id_base in this order --
 * kept by the library between calls (page-locked, so secedo_simmat_compute uploads them by DMA without an
 * intermediate copy, and no fresh pages are touched per call). One caller at a time: returns SECEDO_E_STATE
 * while another thread holds them (fall back to buffers of your own). release() hands them back (they stay
 * allocated until secedo_simmat_release_cache()). */
int secedo_simmat_staging_acquire(const uint64_t bytes[5], void *ptrs[5]);
void secedo_simmat_staging_release(void);

/* secedo_simmat_compute keeps its device buffers between calls (one set per device; the reference calls
 * computeSimilarityMatrix once per sub-cluster of its recursion), and so does secedo_em_refine* with its
 * scratch (secedo_em.h). This frees both. Never required. */
void secedo_simmat_release_cache(void);

/* ------------------------------------------------------------------------------------------
 * Staged interface (device-resident data, explicit stream, tile partition for multi-GPU).
 *
 *   create -> set_pileup -> prepare -> [accumulate -> finalize]* -> destroy
 *
 * prepare()    read assembly (similarity_matrix.cpp:376-403), flush schedule and tail rule
 *              (:342-373, :407-408), packing into cell-block tiles, upload to HBM.
 * accumulate() the pair enumeration of compare_with_reads (:189-243) + apply_updates (:246-254)
 *              for the tiles [tile_begin, tile_end) into a tile-major int64 fixed-point
 *              accumulator in HBM (d_acc, secedo_simmat_acc_elems() elements, device pointer).
 *              Tiles outside the range are left untouched (zero them first: see zero_acc).
 * finalize()   mat_diff - mat_same (:428) + normalize (:271-293): accumulator -> dense
 *              num_cells x num_cells fp64 matrix in HBM (d_out, device pointer).
 * `stream` is a hipStream_t (NULL = the default stream); calls only enqueue work unless noted.
 * ---------------------------------------------------------------------------------------- */
typedef struct secedo_simmat secedo_simmat_t;

int secedo_simmat_create(secedo_simmat_t **handle, int device_id);
void secedo_simmat_destroy(secedo_simmat_t *handle);

/* Borrows the host arrays until secedo_simmat_prepare() returns. */
int secedo_simmat_set_pileup(secedo_simmat_t *handle, const uint32_t *chr_locus_off, uint32_t n_chr,
                             const uint32_t *locus_pos, const uint64_t *locus_entry_off,
                             const uint32_t *read_ids, const uint16_t *id_base16,
                             const uint32_t *id_base32, const uint32_t *group_id_to_pos,
                             uint32_t n_groups);

/* Same, for a raw flat pileup that already lives in HBM (device pointers, borrowed until
 * secedo_simmat_prepare() returns). n_loci = chr_locus_off[n_chr], n_entries = locus_entry_off[n_loci]. */
int secedo_simmat_set_pileup_device(secedo_simmat_t *handle, const uint32_t *d_chr_locus_off,
                                    uint32_t n_chr, const uint32_t *d_locus_pos,
                                    const uint64_t *d_locus_entry_off, const uint32_t *d_read_ids,
                                    const uint16_t *d_id_base16, const uint32_t *d_id_base32,
                                    const uint32_t *d_group_id_to_pos, uint32_t n_groups,
                                    uint32_t n_loci, uint64_t n_entries);

/* Where prepare() packs: 0 = on the GPU, on the host only when the pileup requires it (a read
 * spanning >= max_fragment_length is split at flushes, reference similarity_matrix.cpp:368-382; the
 * empty pileup; >= 2^31 entries); 1 = always on the host; 2 = on the GPU or fail (SECEDO_E_LIMIT).
 * Both paths produce the same packed pileup up to the order of entries inside one (cell block,
 * locus) group, which the integer accumulation does not depend on. Env SECEDO_PACKING=host|device
 * overrides. */
int secedo_simmat_set_packing(secedo_simmat_t *handle, int mode);
/* 1 if the last prepare() packed on the GPU, else 0. */
int secedo_simmat_used_device_packing(const secedo_simmat_t *handle);

/* Read assembly, flush schedule, tile packing (see above). Work is enqueued on `stream` with a few
 * stream synchronisations for scalar read-backs; host packing is synchronous.
 * block_cells: cells per tile edge, 0 = choose (64 or 128). */
int secedo_simmat_prepare(secedo_simmat_t *handle, uint32_t num_cells, uint32_t max_fragment_length,
                          uint32_t num_threads, uint32_t block_cells, void *stream);

/* Geometry after prepare(). Tiles are the upper-triangular (I <= J) cell-block pairs. */
uint32_t secedo_simmat_num_tiles(const secedo_simmat_t *handle);
uint32_t secedo_simmat_block_cells(const secedo_simmat_t *handle);
uint64_t secedo_simmat_acc_elems(const secedo_simmat_t *handle); /* num_tiles * block_cells^2 */
/* Work counters of the prepared pileup: kept entries, live reads (segments), loci. */
uint64_t secedo_simmat_num_entries(const secedo_simmat_t *handle);
uint64_t secedo_simmat_num_reads(const secedo_simmat_t *handle);
uint64_t secedo_simmat_num_loci(const secedo_simmat_t *handle);

/* Fixed-point scale of the accumulator. The int64 accumulators hold D * 2^scale_log2; the scale is 44 unless
 * an int64 could overflow: |sum of one cell pair| <= pair bound (an upper bound on the (read pair, shared
 * locus) incidences one cell pair can collect: the largest per-row sum over loci of squared entry counts) x
 * the largest |D(x_s, x_d)| / (x_s + x_d) among the table entries the pileup can reach (x_s + x_d <= kept
 * entries of its longest read). With the reference's default rates that leaves 44 up to ~1e7 incidences per
 * cell pair. Accumulators that are ADDED UP across handles -- chromosome shards on several GPUs -- must
 * share one scale: every rank reads its shard's per-row squares (cell_squares: num_cells uint64 in device
 * memory) and longest read after prepare(), the ranks sum the vectors and take the maximum (the exact bound
 * of the union) and the maximum of the read lengths, and every rank sets both before accumulate(), ranks with
 * an empty shard included. (0, 0) restores the handle's own; so does setting a pileup of other sizes or
 * arrays. set_pair_bound sets the first alone. scale_log2() is the scale of the last accumulate().
 * ORDER: the bounds belong to the pileup the handle holds WHEN THEY ARE SET -- set them after set_pileup /
 * prepare, never before (bounds set first are taken away by the set_pileup that follows), and set them again
 * when the pileup is set from other arrays (a fresh copy of equal data counts as another pileup: the handle
 * knows a pileup by its sizes and the address of its entry array). scale_bounds_state() tells which holds:
 * 0 = no bounds set (the handle's own scale), 1 = bounds in force for the pileup held, 2 = bounds were set and a
 * later pileup took them away -- a caller that adds accumulators across handles must not accumulate in state 2
 * (secedo_amd.distributed.chromosome_sharded_accumulate refuses to). */
uint64_t secedo_simmat_pair_bound(const secedo_simmat_t *handle);
uint32_t secedo_simmat_max_read_entries(const secedo_simmat_t *handle);
int secedo_simmat_cell_squares(secedo_simmat_t *handle, uint64_t *d_out /* num_cells, device */, void *stream);
int secedo_simmat_set_scale_bounds(secedo_simmat_t *handle, uint64_t pair_bound, uint32_t max_read_entries);
int secedo_simmat_set_pair_bound(secedo_simmat_t *handle, uint64_t pair_bound);
int secedo_simmat_scale_bounds_state(const secedo_simmat_t *handle);
int secedo_simmat_scale_log2(const secedo_simmat_t *handle);

int secedo_simmat_zero_acc(secedo_simmat_t *handle, int64_t *d_acc, void *stream);
int secedo_simmat_accumulate(secedo_simmat_t *handle, double mutation_rate, double homozygous_rate,
                             double seq_error_rate, uint32_t tile_begin, uint32_t tile_end,
                             int64_t *d_acc, void *stream);
/* accumulate() into tiles that need not be zeroed first: acc[tile] = the tile's sum for the tiles of the
 * launch, whatever they held (the reference's matrix starts from zero, similarity_matrix.cpp:306-307; the
 * additive form above is for shards that are summed). Saves the pass that zeroes and the read of the zeroes. */
int secedo_simmat_assign(secedo_simmat_t *handle, double mutation_rate, double homozygous_rate,
                         double seq_error_rate, uint32_t tile_begin, uint32_t tile_end, int64_t *d_acc,
                         void *stream);
int secedo_simmat_finalize(secedo_simmat_t *handle, int normalization, const int64_t *d_acc,
                           double *d_out, void *stream);
/* assign(all tiles) + finalize in one call, the single-GPU form of the reference function after prepare():
 * the maximum that ADD_MIN / SCALE_MAX_1 need (similarity_matrix.cpp:275-278, :286-289) is taken while the tiles are
 * stored, which saves finalize its pass over the accumulator. */
int secedo_simmat_assign_finalize(secedo_simmat_t *handle, double mutation_rate, double homozygous_rate,
                                  double seq_error_rate, int normalization, int64_t *d_acc, double *d_out,
                                  void *stream);
/* A rank that keeps its row block of the matrix without ever receiving other ranks' tiles (BASELINE
 * config 5, no all-gather) accumulates every tile that touches its rows itself -- each off-diagonal
 * tile is then computed by two ranks, and nothing but one scalar is exchanged:
 *   tiles_of_rows       the tiles (global indices, ascending) with their row block or their column
 *                       block inside [row_begin, row_end); tile_ids == NULL only counts
 *   accumulate_list     accumulate() for a list of tiles instead of a range
 *   max_of_tiles        max(0, max D) over the listed tiles, to be max-reduced over the ranks
 *                       (synchronises `stream`)
 *   finalize_rows_max   finalize_rows with that maximum instead of the one of the whole accumulator */
int secedo_simmat_tiles_of_rows(const secedo_simmat_t *handle, uint32_t row_begin, uint32_t row_end,
                                uint32_t *tile_ids, uint32_t *n_tile_ids);
int secedo_simmat_accumulate_list(secedo_simmat_t *handle, double mutation_rate, double homozygous_rate,
                                  double seq_error_rate, const uint32_t *tile_ids, uint32_t n_tile_ids,
                                  int64_t *d_acc, void *stream);
int secedo_simmat_assign_list(secedo_simmat_t *handle, double mutation_rate, double homozygous_rate,
                              double seq_error_rate, const uint32_t *tile_ids, uint32_t n_tile_ids,
                              int64_t *d_acc, void *stream);
int secedo_simmat_max_of_tiles(secedo_simmat_t *handle, const int64_t *d_acc, const uint32_t *tile_ids,
                               uint32_t n_tile_ids, double *max_value, void *stream);
int secedo_simmat_finalize_rows_max(secedo_simmat_t *handle, int normalization, const int64_t *d_acc,
                                    uint32_t row_begin, uint32_t row_end, double max_value, double *d_out_rows,
                                    void *stream);
/* finalize for a row block only: d_out_rows[(row_end - row_begin) * num_cells] receives rows
 * [row_begin, row_end) of the normalised matrix (the maximum that ADD_MIN / SCALE_MAX_1 need is still
 * taken over the whole accumulator). For ranks that keep the matrix sharded by rows (BASELINE config
 * 5) and hand their block to secedo_spectral_eigs_rows_device (secedo_spectral.h). */
int secedo_simmat_finalize_rows(secedo_simmat_t *handle, int normalization, const int64_t *d_acc,
                                uint32_t row_begin, uint32_t row_end, double *d_out_rows, void *stream);
/* Same as finalize with SECEDO_NORM_*, but writes the un-normalised D = logP_diff - logP_same
 * (similarity_matrix.cpp:428), mirrored to both triangles, zero diagonal. For parity checks. */
int secedo_simmat_finalize_raw(secedo_simmat_t *handle, const int64_t *d_acc, double *d_out,
                               void *stream);

/* Counters of the last accumulate() on this handle, read back after synchronising the stream:
 * updates = (read pair, shared locus) incidences examined, read_pairs = pairs that contributed
 * one log-likelihood-ratio term. Synchronises the device. */
int secedo_simmat_last_counts(secedo_simmat_t *handle, uint64_t *updates, uint64_t *read_pairs);
/* Device time of the accumulate kernel of the last accumulate() call, measured with hipEvents
 * on the stream it ran on. Synchronises on the end event. */
int secedo_simmat_last_accumulate_ms(secedo_simmat_t *handle, float *ms);
/* ... and of its dominant kernel by itself when the sparse-loci kernels ran (accumulate_counts, from the start
 * of the accumulate to the launch that follows it); SECEDO_E_STATE otherwise. */
int secedo_simmat_last_pair_kernel_ms(secedo_simmat_t *handle, float *ms);
/* Which pair kernel the prepared pileup runs: "accumulate_counts" (sparse loci: count tile, followed by
 * correct_tiles), "accumulate_masks" (clustered loci: window masks staged, followed by wide_pairs when reads reach
 * beyond their windows) or "accumulate_tiles" (deep pileups, and the A/B switches). Static string. */
const char *secedo_simmat_pair_kernel(const secedo_simmat_t *handle);

/* log-likelihood ratio D(x_s, x_d) = log P(x_s,x_d | different) - log P(x_s,x_d | same) as the
 * matrix path adds it (host-only, no device): what the reference's nested sums return
 * (similarity_matrix.cpp:117-170), including the wrap-around of its uint64_t binomial products from
 * x_s + x_d ~ 48 on -- up to 128 shared loci from the device's table, beyond from the host, which evaluates
 * the few (x_s, x_d) a launch meets there with the same sums (O(x_s^2 x_d^2) terms each: a pair sharing 300
 * loci takes about a second on eight threads, as it takes the reference on its first use). Under
 * SECEDO_LLR_EXACT=1 the closed form of the sums everywhere, i.e. the reference's formula in exact
 * arithmetic (secedo_simmat_llr_closed_form). */
double secedo_simmat_llr(uint32_t x_s, uint32_t x_d, double mutation_rate, double homozygous_rate,
                         double seq_error_rate);
double secedo_simmat_llr_closed_form(uint32_t x_s, uint32_t x_d, double mutation_rate, double homozygous_rate,
                                     double seq_error_rate);

/* ------------------------------------------------------------------------------------------
 * Locus filter, the step immediately upstream of the similarity matrix.
 * Replaces Filter::filter / Filter::is_significant (reference: util/is_significant.hpp:68-72,
 * util/is_significant.cpp:149-193 and :78-138; caller spectral_clustering.cpp:336-337).
 * id_to_pos[g] == 16383 (NO_POS, util/is_significant.hpp:11) marks a group outside the current
 * sub-cluster: its entries are dropped; a locus is kept iff the base counts of the remaining entries
 * pass the reference's significance test for sequencing error rate `seq_error_rate` and split
 * `cell_proportion` (0..4 = 10-90 .. 50-50, util/is_significant.hpp:30-40). Outputs use the flat
 * layout of the input and are caller-allocated with the input's capacities (n_chr + 1, L, L + 1, E, E;
 * out_id_base has the width of the given id_base array). avg_coverage = kept entries / kept loci.
 * The counting, the test and the compaction run on the GPU; loci whose statistic lies within 1e-9 of
 * its threshold are re-decided on the host with the C library (decisions equal the reference's).
 * ---------------------------------------------------------------------------------------- */
int secedo_is_significant(const uint16_t *base_count /* A, C, G, T */, double seq_error_rate,
                          uint32_t cell_proportion);

/* host buffers in, host buffers out: what a binding of Filter::filter calls */
int secedo_filter(const uint32_t *chr_locus_off, uint32_t n_chr, const uint32_t *locus_pos,
                  const uint64_t *locus_entry_off, const uint32_t *read_ids, const uint16_t *id_base16,
                  const uint32_t *id_base32, const uint32_t *id_to_pos, uint32_t n_groups,
                  double seq_error_rate, uint32_t cell_proportion, uint32_t *out_chr_locus_off,
                  uint32_t *out_locus_pos, uint64_t *out_locus_entry_off, uint32_t *out_read_ids,
                  void *out_id_base, uint64_t *out_n_loci, uint64_t *out_n_entries, double *avg_coverage);

/* device buffers in, device buffers out (pileup stays in HBM across recursion levels; the outputs
 * feed secedo_simmat_set_pileup_device). Synchronises `stream`. */
int secedo_filter_device(const uint32_t *d_chr_locus_off, uint32_t n_chr, const uint32_t *d_locus_pos,
                         const uint64_t *d_locus_entry_off, const uint32_t *d_read_ids,
                         const uint16_t *d_id_base16, const uint32_t *d_id_base32, const uint32_t *d_id_to_pos,
                         uint32_t n_groups, uint32_t n_loci, uint64_t n_entries, double seq_error_rate,
                         uint32_t cell_proportion, uint32_t *d_out_chr_locus_off, uint32_t *d_out_locus_pos,
                         uint64_t *d_out_locus_entry_off, uint32_t *d_out_read_ids, void *d_out_id_base,
                         uint64_t *out_n_loci, uint64_t *out_n_entries, double *avg_coverage, void *stream);

/* ------------------------------------------------------------------------------------------
 * Pileup files -> flat layout. Replaces read_pileup (reference:
 * util/pileup_reader.hpp:33-39, util/pileup_reader.cpp:12-257) for one file = one chromosome: a path
 * ending in ".bin" is the reference's binary format (u32 position, u16 coverage, u32 read_ids[],
 * u16 cell<<2|base [] per locus), anything else its 6-column text format. Host only.
 * id_to_group maps cell ids to groups (get_grouping, util/pileup_reader.cpp:273-291); loci with more
 * than max_coverage entries are skipped; a non-empty sorted `positions` list restricts the loci;
 * write_bin != 0 makes the text reader also write `path + ".bin"` as the reference always does.
 * Call with the four arrays NULL to get the sizes in *info, then again with buffers
 * (n_loci, n_loci + 1, n_entries, n_entries). Errors: SECEDO_E_INVALID_ARG + secedo_pileup_last_error().
 * ---------------------------------------------------------------------------------------- */
typedef struct secedo_pileup_info {
    uint64_t n_loci;
    uint64_t n_entries;
    uint32_t num_cells;        /* distinct cell ids (text) resp. largest cell id + 1 (binary) */
    uint32_t max_read_length;  /* longest (last - first) position of one read id; 1000 if not computed */
} secedo_pileup_info;

int secedo_pileup_read(const char *path, const uint16_t *id_to_group, uint32_t n_ids, uint32_t max_coverage,
                       const uint32_t *positions, uint64_t n_positions, int compute_max_read_len,
                       int write_bin, secedo_pileup_info *info, uint32_t *locus_pos,
                       uint64_t *locus_entry_off, uint32_t *read_ids, uint16_t *id_base16);
const char *secedo_pileup_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SECEDO_SIMMAT_H */
