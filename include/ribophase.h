/*
 * ribophase.h -- C ABI of libribophase.so, the MI355X (gfx950) phase-score engine
 * that sits behind ribotricer's `detect-orfs` hot loop.
 *
 * The reference (smithlabcode/ribotricer v1.5.0) is pure Python and has no FFI
 * seam; the entry points below are what a binding for this path would bind.
 * Each one names the reference interface it replaces (paths relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes stub a ribotricer
 * maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types, no exceptions.
 *   - every function returns 0 (RP_OK) or a negative rp_status; rp_last_error()
 *     returns a thread-local message for the last failure on the calling thread.
 *   - pointers prefixed d_ are DEVICE pointers owned by the caller; the library
 *     never allocates or frees device memory.  Inputs are read-only, outputs are
 *     fully overwritten.  Work is enqueued on the caller's HIP stream and is
 *     asynchronous with respect to the host unless stated otherwise.
 *   - ORF P-site profiles are CSR-packed: counts[offsets[i] .. offsets[i+1]) is
 *     the 5'->3' profile of ORF i (the list `cov` of detect_orfs.py:277);
 *     offsets has n_orfs+1 monotone entries with offsets[0] == 0.
 *   - counts must satisfy 0 <= counts[k] <= RP_MAX_COUNT.
 *   - an entry point taking `device` makes it current for the duration of the call and
 *     restores the calling thread's previous HIP device before it returns.
 *   - the *_dev entry points never fall back to the CPU: without a usable HIP device they fail with
 *     RP_ERR_DEVICE / RP_ERR_HIP.  The *_host entry points (no GPU involved) restate the reference's own float64
 *     arithmetic in C++; the Python layer routes through them ONLY when asked to (RIBOTRICER_AMD_BACKEND=cpu, or
 *     "auto" on a machine where no HIP device is visible) -- never behind a failing device call.
 */
#ifndef RIBOPHASE_H
#define RIBOPHASE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RP_VERSION_STRING "0.4.0"

/* largest admissible P-site count per nucleotide (exact in fp32; codon sums stay inside
 * int32 and a lane's 45-position partial read count inside uint32) */
#define RP_MAX_COUNT 16777215 /* 2^24 - 1: counts convert to fp32 exactly */

/* value of min_codon_cov for an ORF with an empty profile (min over no codons;
 * numpy.all([]) is True in detect_orfs.py:288,293) */
#define RP_MIN_CODON_COV_EMPTY 2147483647

typedef enum rp_status {
    RP_OK = 0,
    RP_ERR_NULL = -1,      /* required pointer is NULL */
    RP_ERR_SIZE = -2,      /* negative / inconsistent size */
    RP_ERR_OFFSETS = -3,   /* offsets[0] != 0, not monotone, or offsets[n] != total_nt */
    RP_ERR_HIP = -4,       /* HIP runtime error (see rp_last_error) */
    RP_ERR_WORKSPACE = -5, /* workspace missing, misaligned or too small */
    RP_ERR_DEVICE = -6,    /* no such device / no HIP device available */
    RP_ERR_COUNTS = -7,    /* a count is negative, passes 2^31 - 1, or (strict callers) exceeds RP_MAX_COUNT */
    RP_ERR_ARG = -8,       /* invalid enum / option value */
    RP_ERR_INDEX_COLUMNS = -9, /* index line without exactly 11 tab-separated fields (orf.py:143-152) */
    RP_ERR_INDEX_COORD = -10,  /* malformed "start-end,..." coordinate field */
    RP_ERR_BAM = -11,          /* BAM file cannot be opened / is not BGZF-compressed BAM / is truncated */
    RP_ERR_INTERVALS = -12     /* interval table with an empty or off-array interval: no gather plan (use rp_gather_profiles_dev) */
} rp_status;

/* bits of the per-ORF flags byte */
#define RP_FLAG_TIE 0x01u      /* two candidate frames score within RP_TIE_RTOL with different N: the
                                  reference's pick is decided by the last bits of what numpy / scipy
                                  computed.  Such ORFs are re-scored by a
                                  replay of exactly that arithmetic (RP_FLAG_REPLAY) */
#define RP_FLAG_RECHECK64 0x02u /* frame decision was re-derived in float64 on device */
#define RP_FLAG_SPLIT 0x04u    /* profile spanned more than one tile (several segment records) */
#define RP_FLAG_REPLAY 0x08u   /* phase / valid_codons of this tie-flagged ORF come from the on-device
                                  replay of the reference's own float64 (numpy / scipy) arithmetic */
#define RP_FLAG_BIGCOUNT 0x20u /* set by the Python layer (never by a kernel): the ORF holds a count beyond RP_MAX_COUNT and
                                  its results were recomputed in float64 / int64 (engine.rescore_big_count_orfs) */
#define RP_FLAG_UNRESOLVED 0x40u /* only with RP_FILTER_PRINTED_ONLY: too close to call in fp32 and left at that, because
                                   no resolution could make the ORF translating (status is 0 for certain) */
#define RP_FLAG_BIGTIE 0x10u   /* the replay met a codon with a count >= 16: the reference squares through
                                  the host C library's pow() (statistics.py:83), which the device cannot
                                  restate past its host-filled table; phase / valid_codons stand on x*x
                                  there.  rp_tie_replay_host on the profile gives the reference's bits
                                  (the Python layer does that for every such ORF: engine.resolve_big_ties) */

/* two frame scores count as tied when they differ by no more than RP_TIE_RTOL * (the larger
 * one) + RP_TIE_ATOL: relative, because the reference's own rounding noise is relative (~1e-15);
 * the absolute floor covers the frames whose unit vectors cancel exactly (scores that are 0 in
 * exact arithmetic and ~1e-32 in float64) */
#define RP_TIE_RTOL 1e-9
#define RP_TIE_ATOL 1e-24

/* kernel family selector */
typedef enum rp_algo {
    RP_ALGO_AUTO = 0, /* RP_ALGO_TILE, or RP_ALGO_WAVE for batches under 2 Mi nucleotides */
    RP_ALGO_WAVE = 1, /* one wavefront per ORF, streaming straight from HBM */
    RP_ALGO_TILE = 2  /* LDS-staged flat tiles, ragged lane packing -> one record per (ORF, tile)
                         segment in the workspace -> one thread per ORF scores and filters */
} rp_algo;

/*
 * Thresholds of the status predicate, detect_orfs.py:289-299 (defaults const.py:20-39;
 * CLI flags cli.py:173-219).  status = 1 ("translating") iff
 *   phase >= phase_score_cutoff  and  valid >= min_valid_codons
 *   and  min_codon_cov >= min_reads_per_codon
 *   and  valid / n_codons >= min_valid_codons_ratio
 *   and  read_count / n_codons >= min_density_over_orf,      n_codons = max(1, L // 3).
 */
typedef struct rp_filter_params {
    double phase_score_cutoff;     /* const.py:20  CUTOFF = 0.428571428571 */
    double min_valid_codons_ratio; /* const.py:35  0 */
    double min_density_over_orf;   /* const.py:39  0.0 */
    double min_reads_per_codon;    /* const.py:32  0 */
    int32_t min_valid_codons;      /* const.py:27  5 */
    int32_t flags;                 /* RP_FILTER_* bits; 0 = every ORF fully resolved */
} rp_filter_params;

/*
 * rp_filter_params.flags.  RP_FILTER_PRINTED_ONLY: the caller prints translating ORFs only (the reference's default,
 * detect_orfs.py:301-302: `if status == "nontranslating" and not report_all: continue`).  An ORF whose frame decision is too close to
 * call in fp32 is then NOT re-walked in float64 / replayed when no outcome of that decision could make it translating
 * (every frame's N below min_valid_codons, every frame's score below the cutoff by more than the fp32 margin, or an
 * integer condition failing): its status is 0 either way.  Such ORFs carry RP_FLAG_UNRESOLVED; their phase is the fp32
 * tile sums' (within 1e-5 of the reference), their valid_codons one of the tied frames' N.  Everything else -- every ORF
 * with status 1, every unflagged ORF -- is exactly what flags = 0 gives.  Device tile path only; the host entry points and
 * the wave kernel resolve everything regardless.
 */
#define RP_FILTER_PRINTED_ONLY 0x1

/* Library / error introspection. */
const char *rp_version(void);
const char *rp_last_error(void);
const char *rp_status_string(int status);

/* Number of visible HIP devices (RP_ERR_DEVICE if the runtime reports none). */
int rp_device_count(int *n_devices);

/* Fill *out with the reference defaults (const.py:20-39). */
int rp_filter_defaults(rp_filter_params *out);

/*
 * Bytes of device workspace rp_phase_score_csr_dev needs for a batch of this
 * shape (one 48-byte record per ORF and per tile, the long re-walk queue, and -- without a plan -- the tile index, descriptors and head rows).  16-byte aligned
 * pointer required.
 * Placement matters on MI355X: the scorer writes its records into the workspace while it streams
 * the counts, and a write stream costs a read stream ~10 % when the two buffers lie in different
 * classes of the physical address space (runs of 16-32 GiB) and ~23 % when they share one -- 2.6 vs
 * 3.0 ms per 4 G nt.  HIP has no placement hint; a caller that scores many batches can allocate a
 * few candidate workspaces some GiB apart, time a call on each and keep the fastest (what
 * ribotricer_amd's engine.tune_workspace does).
 */
int rp_workspace_bytes(int64_t n_orfs, int64_t total_nt, int algo, size_t *bytes);

/*
 * THE HOT PATH.  Replaces, for a whole batch of ORFs at once, the per-ORF body of
 * export_orf_coverages (detect_orfs.py:274-299):
 *     count = sum(cov); length = len(cov)                      detect_orfs.py:278-279
 *     coh, valid_codons = phasescore(cov)                      detect_orfs.py:280
 *                                                              -> statistics.py:48-115
 *     codon_coverage = collapse_coverage_to_codon(cov)         detect_orfs.py:284
 *                                                              -> common.py:164-180
 *     valid_codons_ratio, orf_density, status                  detect_orfs.py:285-299
 *
 * Outputs per ORF i (arrays of n_orfs elements):
 *     d_phase[i]          float64  phase score  (np.sqrt(coh), statistics.py:115)
 *     d_valid[i]          int32    valid codons of the winning frame
 *     d_read_count[i]     int64    sum(cov)
 *     d_min_codon_cov[i]  int32    min over codon sums incl. the partial last codon
 *                                  (RP_MIN_CODON_COV_EMPTY if the profile is empty)
 *     d_flags[i]          uint8    RP_FLAG_* bits
 *     d_status[i]         uint8    1 = translating, 0 = nontranslating; optional:
 *                                  pass d_status = NULL or filter = NULL to skip
 * total_nt must equal offsets[n_orfs] (the caller built the CSR and knows it);
 * `filter` is a HOST pointer read before the call returns.
 */
int rp_phase_score_csr_dev(int device, const int32_t *d_counts, const int64_t *d_offsets,
                           int64_t n_orfs, int64_t total_nt, double *d_phase, int32_t *d_valid,
                           int64_t *d_read_count, int32_t *d_min_codon_cov, uint8_t *d_flags,
                           uint8_t *d_status, const rp_filter_params *filter, void *d_workspace,
                           size_t workspace_bytes, int algo, void *hip_stream);

/*
 * Positions per tile of the tile path for an index of n_orfs ORFs and total_nt nucleotides:
 * 7 936, or 6 144 for indexes of short ORFs (mean length < 180 nt), where a tile then holds
 * fewer segments.  Informational (RP_FLAG_SPLIT marks ORFs that span tiles); every sizing
 * function and kernel derives the same value from the same two numbers.
 */
int rp_tile_positions(int64_t n_orfs, int64_t total_nt, int32_t *positions);

/*
 * Tile plans.  export_orf_coverages scores ONE candidate-ORF index against many samples
 * (detect_orfs.py:510-520 is called once per BAM with the same ribotricer_index), and
 * everything the tile path derives from the offsets alone -- the tile index, and the check
 * that the offsets are a valid CSR index -- is the same for all of them.  A plan holds that
 * part: build it once per index, then call rp_phase_score_csr_plan_dev per sample.
 *
 *   rp_plan_bytes        device bytes the plan needs (caller-owned memory, 16-byte aligned)
 *   rp_plan_create_dev   validates d_offsets (RP_ERR_OFFSETS: offsets[0] != 0, a decreasing
 *                        step, offsets[n] != total_nt), builds the tile index in d_plan_mem,
 *                        SYNCHRONISES the stream and returns a small host handle.
 *                        counts_phase = (address of the counts array / 4) % 4 that the plan
 *                        will be used with (tiles live on the 16-byte address grid).
 *   rp_plan_free         frees the host handle only; d_plan_mem stays the caller's.
 * rp_phase_score_csr_plan_dev = rp_phase_score_csr_dev(RP_ALGO_TILE) minus the index pass;
 * RP_ERR_ARG when d_counts has a different 16-byte phase than the plan was built for.
 */
typedef struct rp_plan rp_plan;

int rp_plan_bytes(int64_t n_orfs, int64_t total_nt, size_t *bytes);
int rp_plan_create_dev(int device, const int64_t *d_offsets, int64_t n_orfs, int64_t total_nt,
                       int counts_phase, void *d_plan_mem, size_t plan_bytes, void *hip_stream,
                       rp_plan **out);
void rp_plan_free(rp_plan *plan);
int rp_phase_score_csr_plan_dev(const rp_plan *plan, const int32_t *d_counts, const int64_t *d_offsets,
                                double *d_phase, int32_t *d_valid, int64_t *d_read_count,
                                int32_t *d_min_codon_cov, uint8_t *d_flags, uint8_t *d_status,
                                const rp_filter_params *filter, void *d_workspace,
                                size_t workspace_bytes, void *hip_stream);

/*
 * Per-frame diagnostics in float64: score_f (NaN when M_f == 0, 0 when N_f == 0),
 * N_f, M_f for the three reading frames of every ORF, laid out [n_orfs][3].
 * These are the quantities statistics.py:67-108 computes per frame and never
 * exposes; used to analyse tie-flagged ORFs.
 */
int rp_phase_score_frames_dev(int device, const int32_t *d_counts, const int64_t *d_offsets,
                              int64_t n_orfs, double *d_frame_score, int32_t *d_frame_n,
                              int32_t *d_frame_m, void *hip_stream);

/*
 * float64 profiles (the metagene caller passes float sums: metagene.py:243-244
 * -> statistics.py:48).  Same CSR convention, float64 values, float64 arithmetic.
 */
int rp_phase_score_f64_csr_dev(int device, const double *d_values, const int64_t *d_offsets,
                               int64_t n_profiles, double *d_phase, int32_t *d_valid,
                               uint8_t *d_flags, void *hip_stream);

/*
 * Exact frame ties the device cannot finish with the reference's bits (HOST functions, no GPU
 * involved): statistics.py:83 squares through the C library's pow(), which is not x*x and
 * belongs to the machine the reference runs on -- so this step is taken with this host's own
 * libm.  For every profile of the CSR batch: phase[i], valid[i] = what statistics.py:48-115
 * returns, float64 operation for float64 operation (csrc/rp_replay.hpp).  Meant for the few
 * profiles per sample flagged RP_FLAG_BIGTIE (integer profiles, a tie involving a count >= 16)
 * and for tie-flagged float profiles (metagene.py:243-244); it is NOT a scoring path -- a
 * single thread, ~1 microsecond per codon.
 */
int rp_tie_replay_host(const int32_t *counts, const int64_t *offsets, int64_t n_profiles,
                       double *phase, int32_t *valid);
int rp_tie_replay_f64_host(const double *values, const int64_t *offsets, int64_t n_profiles,
                           double *phase, int32_t *valid);

/*
 * The per-ORF loop body of detect_orfs.py:274-299 for a whole CSR batch ON THE HOST, in the
 * reference's own float64 arithmetic (the sequence of rp_tie_replay_host for every profile, not
 * only the ties) -- SURVEY.md section 8(b) lists a host entry point next to the device one.  Outputs
 * as rp_phase_score_csr_dev's (host pointers): phase and valid_codons carry the reference's bits on
 * EVERY ORF, read_count / min_codon_cov are the integer results, flags = 0, status (may be NULL)
 * the predicate of `filter` (may be NULL: no status).  n_threads <= 0: all hardware threads.
 * The scoring step of the GPU-less backend (RIBOTRICER_AMD_BACKEND=cpu: export_orf_coverages / phasescore on a
 * machine without a HIP device -- BASELINE configs[0], "CPU path, plumbing"); never used while the hip backend is
 * selected, and never as a fallback behind a failing device call.  ~1 microsecond per codon and thread.
 */
int rp_phase_score_csr_host(const int32_t *counts, const int64_t *offsets, int64_t n_orfs, double *phase,
                            int32_t *valid, int64_t *read_count, int32_t *min_codon_cov, uint8_t *flags,
                            uint8_t *status, const rp_filter_params *filter, int n_threads);

/*
 * Profile gather: builds the CSR counts array on the device from
 * dense P-site coverage and the ORFs' exon intervals.  Replaces orf_coverage() for every
 * ORF at once (detect_orfs.py:134-203): the positions of the intervals in ascending order,
 * reversed for '-' strand ORFs (detect_orfs.py:201-202); positions outside d_coverage
 * count 0 (the reference's missing-key case, detect_orfs.py:176-187).
 *   d_coverage  int32[coverage_len]  all (strand, chrom) coverage arrays, concatenated
 *   d_iv_start  int64[n_intervals]   index into d_coverage of each interval's first position
 *   d_iv_len    int32[n_intervals]   interval lengths (end - start + 1, interval.py:60-62)
 *   d_orf_iv    int64[n_orfs+1]      which intervals belong to which ORF (CSR, ascending)
 *   d_reverse   uint8[n_orfs]        1 for '-' strand ORFs
 *   d_offsets   int64[n_orfs+1]      output CSR offsets = prefix sum of the ORFs' lengths
 *   d_counts    int32[offsets[n]]    output, ready for rp_phase_score_csr_dev
 */
int rp_gather_profiles_dev(int device, const int32_t *d_coverage, int64_t coverage_len,
                           const int64_t *d_iv_start, const int32_t *d_iv_len,
                           const int64_t *d_orf_iv, const uint8_t *d_reverse,
                           const int64_t *d_offsets, int64_t n_orfs, int32_t *d_counts,
                           void *hip_stream);

/*
 * Gather plan (fused form): the profile space of a whole index -- every
 * ORF's exon intervals in transcript order, '-' strand ORFs reversed (detect_orfs.py:134-203)
 * -- as a sorted list of pieces of the dense coverage plus one fixed-stride row of clipped
 * pieces per 7 936-position tile.  Depends on the index (interval table + the coverage layout
 * derived from it) only: built once per index, reused for every sample.  Arguments as for
 * rp_gather_profiles_dev; every interval must be non-empty and lie inside the coverage array
 * (RP_ERR_INTERVALS otherwise -- rp_gather_profiles_dev handles such tables), the intervals of
 * an ORF must add up to its profile length (RP_ERR_OFFSETS).
 * d_plan_mem: device memory of rp_gather_plan_bytes() bytes, 16-byte aligned, caller-owned,
 * must outlive the plan.  Synchronous on hip_stream.
 */
typedef struct rp_gather_plan rp_gather_plan;
int rp_gather_plan_bytes(int64_t n_orfs, int64_t n_intervals, int64_t total_nt, size_t *bytes);
int rp_gather_plan_create_dev(int device, const int64_t *d_iv_start, const int32_t *d_iv_len,
                              const int64_t *d_orf_iv, const uint8_t *d_reverse, const int64_t *d_offsets,
                              int64_t n_orfs, int64_t n_intervals, int64_t total_nt, int64_t coverage_len,
                              void *d_plan_mem, size_t plan_bytes, void *hip_stream, rp_gather_plan **out);
void rp_gather_plan_free(rp_gather_plan *plan);

/*
 * The profiles of a SUBSET of the index's ORFs through the plan (default mode of the export prints the translating
 * ORFs only: detect_orfs.py:301-303 skips the others before `profile` is formatted, :323): one wave per chosen ORF.
 *   d_chosen       int64[n_chosen]      ORF numbers (any order; each < n_orfs of the plan -- not checked)
 *   d_out_offsets  int64[n_chosen]      where each chosen ORF's profile starts in d_counts (the caller's prefix sum of
 *                                       the chosen ORFs' lengths)
 *   d_counts       int32[sum of the chosen lengths]
 * Asynchronous.  Same bytes as rp_gather_profiles_dev over the sub-table of the chosen ORFs.
 */
int rp_gather_selected_plan_dev(const rp_gather_plan *plan, const int32_t *d_coverage, int64_t coverage_len,
                                const int64_t *d_chosen, int64_t n_chosen, const int64_t *d_out_offsets,
                                int32_t *d_counts, void *hip_stream);

/*
 * rp_gather_profiles_dev through a gather plan: one workgroup per tile stages its positions
 * from the coverage with LDS-DMA (64 consecutive positions of one piece per instruction) and
 * writes them out 16 bytes per lane.  d_counts: int32[total_nt], 16-byte aligned.  Asynchronous.
 */
int rp_gather_profiles_plan_dev(const rp_gather_plan *plan, const int32_t *d_coverage, int64_t coverage_len,
                                int32_t *d_counts, void *hip_stream);

/*
 * Fused gather + score (the default, non-report_all mode of detect_orfs.py:274-324, where only
 * the translating minority's profiles are printed): rp_phase_score_csr_dev with the tiles
 * staged straight from the dense coverage through the gather plan -- the CSR counts array is
 * never written or read; the float64 re-walks and tie replays of the too-close-to-call ORFs
 * read the coverage through the plan as well.  Results are bit-identical to
 * rp_gather_profiles_dev + rp_phase_score_csr_plan_dev.  plan: the tile plan of d_offsets built
 * with counts_phase 0 (may be NULL: built per call).  ms: NULL, or float[4] receiving HIP-event
 * times {plan kernels, scoring kernel, finish, whole call} (then synchronous).
 */
int rp_phase_score_coverage_dev(int device, const int32_t *d_coverage, int64_t coverage_len,
                                const int64_t *d_offsets, int64_t n_orfs, int64_t total_nt, double *d_phase,
                                int32_t *d_valid, int64_t *d_read_count, int32_t *d_min_codon_cov,
                                uint8_t *d_flags, uint8_t *d_status, const rp_filter_params *filter,
                                void *d_workspace, size_t workspace_bytes, const rp_plan *plan,
                                const rp_gather_plan *gather, void *hip_stream, float *ms);

/*
 * Dense P-site coverage from columnar alignments: replaces the
 * Counter arithmetic of merge_read_lengths (detect_orfs.py:54-83: one dict update per
 * (read length, strand, chrom, pos) key) and the per-nucleotide dict lookups of orf_coverage
 * (detect_orfs.py:176-187) with one launch:
 *     coverage[group_start[g] + pos - group_lo[g]] += count     for group_lo[g] <= pos <= group_hi[g]
 *   d_group   int32[n]  (strand, chrom) group of the candidate-ORF index (rp_index_view.group
 *                       numbering); negative = not in the index, entry ignored
 *   d_pos     int64[n]  1-based position, ALREADY shifted by the read length's P-site offset
 *                       (+offset on '+', -offset on '-': detect_orfs.py:76-80)
 *   d_count   int32[n]  reads; entries hitting one position add up (any order: integer atomics)
 *   d_group_* int64[n_groups]  first coverage index and [lo, hi] extent of each group
 * d_coverage must be zero-filled (or hold a partial sum) on entry.  Synchronous.
 * Counts beyond RP_MAX_COUNT (round 4; the reference has no limit: detect_orfs.py:176-187, 278-280):
 *   big_counts != NULL   *big_counts = 1 when an accumulated count passed RP_MAX_COUNT, else 0; RP_OK either
 *                        way.  The scoring kernels' fp32 codon arithmetic is exact only up to RP_MAX_COUNT, so the
 *                        caller must finish the ORFs that hold such a position in float64
 *                        (rp_coverage_big_positions_dev lists the positions; rp_phase_score_f64_csr_dev scores
 *                        their profiles; integer sums from the int32 profile -- ribotricer_amd.engine.
 *                        rescore_big_count_orfs does all three).
 *   big_counts == NULL   the strict contract: RP_ERR_COUNTS when a count passes RP_MAX_COUNT.
 * RP_ERR_COUNTS in both cases for a negative count or a sum past 2^31 - 1 (the coverage is int32).
 */
int rp_coverage_build_dev(int device, const int32_t *d_group, const int64_t *d_pos, const int32_t *d_count,
                          int64_t n_entries, const int64_t *d_group_start, const int64_t *d_group_lo,
                          const int64_t *d_group_hi, int32_t n_groups, int32_t *d_coverage,
                          int64_t coverage_len, void *hip_stream, int32_t *big_counts);

/*
 * The same from the columns merge_read_lengths hands over (detect_orfs.py:54-83 as columns: strand
 * uint8 0 '+' / 1 '-', chromosome code int32, shifted position int64, count int64), with the
 * (strand, chromosome) -> group lookup on the device: d_lut[strand * n_chroms + chrom] = group of the
 * candidate-ORF index, or -1 where no ORF lives.  Rows outside every group's extent are dropped
 * whatever their count (the reference never looks them up).  big_counts / RP_ERR_COUNTS as for
 * rp_coverage_build_dev, for rows that do land and their sums.  Synchronous.
 */
int rp_coverage_build_rows_dev(int device, const uint8_t *d_strand, const int32_t *d_chrom, const int64_t *d_pos,
                               const int64_t *d_count, int64_t n_rows, const int32_t *d_lut, int32_t n_chroms,
                               const int64_t *d_group_start, const int64_t *d_group_lo, const int64_t *d_group_hi,
                               int32_t n_groups, int32_t *d_coverage, int64_t coverage_len, void *hip_stream,
                               int32_t *big_counts, const void *d_block_map, int64_t dense_len, int32_t block_positions);

/*
 * Compact coverage (round 4).  The dense layout above gives every position of every (strand, chromosome) extent a slot
 * -- 25 GB for a human index -- although only positions under an exon interval are ever read (the reference looks nothing
 * else up: detect_orfs.py:176-187).  rp_coverage_map_create_dev keeps the blocks of block_positions (a power of two, 1 ...
 * 64) positions that an interval touches and packs them in order (one bit per block + a running count per 64 blocks, in
 * caller-owned d_map_mem of rp_coverage_map_bytes(dense_len, block_positions) bytes), REWRITES d_iv_start from dense to
 * compact coordinates (an interval stays contiguous) and returns the compact length.  block_positions 64: the map takes
 * dense_len / 256 bytes and the coverage keeps up to 126 unread positions per interval; block_positions 1: dense_len / 4
 * bytes of map and nothing but exonic positions in the coverage -- exons that face each other across an intron become
 * neighbours in memory, so the gather plan merges a spliced ORF's pieces into one run and consecutive pieces share their
 * cache lines.  rp_coverage_build_rows_dev with that d_block_map (and the dense_len and block_positions it was built for)
 * accumulates straight into a compact coverage of compact_len positions; rows under no exon are dropped.
 * Everything downstream (gather plan, fused scoring, gathers) takes the compact coverage and the rewritten table as
 * they are.  Synchronous.  RP_ERR_INTERVALS for an empty or off-layout interval, RP_ERR_ARG for another block size.
 */
int rp_coverage_map_bytes(int64_t dense_len, int32_t block_positions, size_t *bytes);
int rp_coverage_map_create_dev(int device, int64_t *d_iv_start, const int32_t *d_iv_len, int64_t n_intervals, int64_t dense_len,
                               int32_t block_positions, void *d_map_mem, size_t map_bytes, void *hip_stream, int64_t *compact_len);
/* Positions of the dense layout that lie under an exon (e.g. interval starts) -> their slots in the compact coverage, in
 * place, through a map that rp_coverage_map_create_dev built (same dense_len and block_positions).  Asynchronous. */
int rp_coverage_map_remap_dev(int device, int64_t *d_positions, int64_t n_positions, const void *d_map_mem, int64_t dense_len,
                              int32_t block_positions, void *hip_stream);

/*
 * The positions of a dense coverage whose count passes RP_MAX_COUNT (after a coverage build reported
 * big_counts): up to `capacity` indices into d_coverage are written to d_positions (any order),
 * *n_found receives how many there are in all (call with capacity 0 to size the buffer).  One pass over
 * the array; synchronous.  Replaces nothing in the reference -- it is what lets the fp32 kernels serve an
 * input range the reference's Python ints cover (detect_orfs.py:176-187).
 */
int rp_coverage_big_positions_dev(int device, const int32_t *d_coverage, int64_t coverage_len, int64_t *d_positions,
                                  int64_t capacity, int64_t *n_found, void *hip_stream);

/*
 * Metagene profiles of one read length: replaces the per-ORF pandas
 * loop of metagene_coverage (metagene.py:203-228).  d_counts / d_offsets: the profiles
 * "leader + ORF + trailer, first max_positions nucleotides, transcript orientation" of the
 * annotated ORFs (orf_coverage_length, metagene.py:97-157), CSR-packed in index order --
 * rp_gather_profiles_dev builds them from the read length's dense coverage.  Outputs:
 *   d_mean  float64[n_orfs]          mean coverage of each profile (0 for an empty one)
 *   d_sum   float64[2*max_positions] [0..max) start side: sum over ORFs with mean > 0 of
 *                                    profile[j] / mean, ORFs added in index order (the float64
 *                                    operations of the reference, so the same bits);
 *                                    [max..2max) stop side, slot m = m-th nucleotide from the end
 *   d_seen  int32[2*max_positions]   ORFs that contributed to each slot (position_counter)
 * Asynchronous on hip_stream.
 */
int rp_metagene_dev(int device, const int32_t *d_counts, const int64_t *d_offsets, int64_t n_orfs,
                    int32_t max_positions, double *d_mean, double *d_sum, int32_t *d_seen, void *hip_stream);

/* rp_metagene_dev on the HOST (the GPU-less backend): the same float64 operations in the same order, same outputs. */
int rp_metagene_host(const int32_t *counts, const int64_t *offsets, int64_t n_orfs, int32_t max_positions, double *mean,
                     double *sum, int32_t *seen);

/*
 * Synchronous input check (one pass over offsets and counts on the device, then a
 * host sync): RP_ERR_OFFSETS / RP_ERR_COUNTS as documented above.
 */
int rp_validate_csr_dev(int device, const int32_t *d_counts, const int64_t *d_offsets,
                        int64_t n_orfs, int64_t total_nt, void *hip_stream);

/*
 * ---- host side: TSV row rendering ------------------------------
 *
 * rp_format_rows_host replaces the per-ORF `formatter.format(...)` of
 * detect_orfs.py:301-324 for a whole batch: plain host memory in, text out, no GPU
 * involved.  Every rendering is byte-identical to CPython's ('{}'.format of a float is
 * repr(): shortest round-trip digits, fixed notation for 1e-4 <= |x| < 1e16; of a list of
 * ints "[a, b, c]").  Column order, detect_orfs.py:304-323:
 *   head_i  status  phase_score  read_count  length  valid_codons  valid_codons_ratio
 *   read_density  tail_i  profile
 * where head_i = "ORF_ID\tORF_type" and tail_i = "transcript_id\t...\tstart_codon" are
 * the caller's bytes (index columns, orf.py:122-182), valid_codons_ratio =
 * valid / max(1, length // 3) (detect_orfs.py:281-285) and read_density likewise (:287).
 * Rows of ORFs with status[i] == 0 are skipped unless report_all (detect_orfs.py:301-303).
 *
 * Streaming: rows of ORFs first, first+1, ... are written while they fit in out[0..cap);
 * *next = first ORF not written (n_orfs when finished), *out_len = bytes written.  If the
 * first row to be written does not fit on its own: RP_ERR_SIZE, *next = its ORF (skipped
 * rows before it are consumed), *out_len = bytes it needs.  Thread-safe (no shared state): callers may format disjoint ORF ranges
 * concurrently into separate buffers.
 */
int rp_format_rows_host(const int32_t *counts, const int64_t *offsets, int64_t n_orfs,
                        const double *phase, const int32_t *valid, const int64_t *read_count,
                        const uint8_t *status, const char *head, const int64_t *head_off,
                        const char *tail, const int64_t *tail_off, int report_all, int64_t first,
                        char *out, size_t out_cap, int64_t *next, size_t *out_len);

/*
 * ---- host side: index parser -----------------------------------
 *
 * rp_index_parse_host reads the text of a `{prefix}_candidate_orfs.tsv` index
 * (prepare_orfs.py:370-404) in one pass with the line semantics of ORF.from_string /
 * ORF.__init__ (orf.py:88-182): 11 tab-separated fields or RP_ERR_INDEX_COLUMNS (the
 * reference exits there), "s-e,s-e" coordinates sorted by start, ORF_ID recomputed as
 * tid_start_end_length, start_codon = first three characters of field 9 or None.  It
 * replaces parse_ribotricer_index (detect_orfs.py:86-131) and the per-line
 * ORF.from_string calls of the export loop (detect_orfs.py:273-278).
 *
 * The result is an opaque object; rp_index_view_host lends out its arrays (valid until
 * rp_index_free):
 *   per ORF       orf_iv[n+1], length[n], group[n] (index of its (strand, chrom)), reverse[n]
 *   per interval  iv_start[], iv_end[]   1-based closed, ascending inside each ORF
 *   per group     group_names/off ("strand\tchrom"), group_lo/hi = extent of its ORFs
 *   string tables head ("ORF_ID\tORF_type") and tail ("transcript_id ... start_codon")
 *                 exactly as rp_format_rows_host takes them
 * On a malformed line *error_line (may be NULL) receives its 1-based line number.
 */
typedef struct rp_index rp_index;

typedef struct rp_index_view {
    int64_t n_orfs, n_intervals, n_groups;
    const int64_t *orf_iv;
    const int64_t *length;
    const int32_t *group;
    const uint8_t *reverse;
    const int64_t *iv_start;
    const int64_t *iv_end;
    const char *group_names;
    const int64_t *group_off;
    const int64_t *group_lo;
    const int64_t *group_hi;
    const char *head;
    const int64_t *head_off;
    const char *tail;
    const int64_t *tail_off;
} rp_index_view;

int rp_index_parse_host(const char *text, size_t len, int skip_header, rp_index **out, int64_t *error_line);
int rp_index_view_host(const rp_index *index, rp_index_view *view);
void rp_index_free(rp_index *index);

/*
 * The interval table of rp_gather_profiles_dev / rp_gather_plan_create_dev from a parsed index and
 * a coverage layout (host, one pass): interval k of an ORF of group g starts at coverage index
 * group_start[g] + (iv_start[k] - group_lo[g]) and is iv_end[k] - iv_start[k] + 1 long
 * (interval.py:60-62); out_offsets = prefix sum of the ORF lengths.  Arrays as in rp_index_view.
 */
int rp_interval_table_host(const int64_t *iv_start, const int64_t *iv_end, const int64_t *orf_iv, const int32_t *group,
                           const int64_t *length, int64_t n_orfs, int64_t n_intervals, const int64_t *group_start,
                           const int64_t *group_lo, int64_t n_groups, int64_t *out_iv_start, int32_t *out_iv_len,
                           int64_t *out_offsets);

/*
 * orf_coverage for every ORF of a parsed index ON THE HOST (detect_orfs.py:134-203; the gather of the GPU-less backend,
 * RIBOTRICER_AMD_BACKEND=cpu -- the device paths are rp_gather_profiles_dev / the gather plan): the merged P-site counts
 * (merge_read_lengths, detect_orfs.py:54-83) come as a sorted table instead of a dict,
 *   keys  int64[n_keys]  (group << 40) | position, ascending and unique: group = the (strand, chrom) group of the index
 *                        (rp_index_view.group numbering), position 1-based and already shifted by the P-site offset
 *   vals  int64[n_keys]  reads at that position (rows of one position added up by the caller)
 * and every position of every exon interval (rp_index_view: iv_start / iv_end 1-based closed, orf_iv, group, reverse) is
 * looked up in it: absent = 0 (the missing-key case of detect_orfs.py:176-187), '-' strand profiles reversed (:201-202).
 *   offsets int64[n_orfs + 1]  prefix sums of the ORF lengths;  counts int32[offsets[n_orfs]]  output (CSR).
 * RP_ERR_COUNTS for a count that is negative or passes 2^31 - 1 (the CSR array is int32), RP_ERR_OFFSETS when an ORF's
 * intervals do not add up to its length.  n_threads <= 0: all usable cores.
 */
int rp_gather_profiles_host(const int64_t *keys, const int64_t *vals, int64_t n_keys, const int64_t *iv_start,
                            const int64_t *iv_end, const int64_t *orf_iv, const int32_t *group, const uint8_t *reverse,
                            const int64_t *offsets, int64_t n_orfs, int32_t *counts, int n_threads);

/*
 * Default-mode bookkeeping of export_orf_coverages (detect_orfs.py:301-303 prints the translating ORFs only): from the
 * per-ORF `keep` flags and profile lengths, in one pass -- the ids of the kept ORFs (chosen[k]), where each one's profile
 * starts in the packed array of kept profiles (chosen_off[k], k <= n_chosen), and CSR offsets over ALL ORFs in which
 * every other ORF has an empty range (offsets[n_orfs + 1]: what the row formatter takes).  chosen / chosen_off need
 * room for n_orfs (+ 1) entries.
 */
int rp_select_profiles_host(const uint8_t *keep, const int64_t *lengths, int64_t n_orfs, int64_t *chosen, int64_t *chosen_off,
                            int64_t *offsets, int64_t *n_chosen);

/*
 * The part of a coverage array that a set of intervals touches, as a few WINDOWS (what a GPU that scores one slice of
 * the index needs of the whole coverage: engine.CoverageShards; replaces nothing in the reference -- its loop runs on
 * one process).  Intervals that lie closer than 2^gap_shift positions share a window; windows ascend, start and length
 * rounded to 16 positions; win_base[k] = where window k sits in the compacted array of *total positions.  With
 * out_iv_start non-NULL the interval starts are also re-based onto the compacted array (start - win_start[k] +
 * win_base[k]).  One pass over the intervals, no sort (a block of 2^gap_shift positions can only hold one window).
 * RP_ERR_SIZE when more than `capacity` windows exist (*n_windows says how many), RP_ERR_INTERVALS for an empty or
 * negative interval.
 */
int rp_coverage_windows_host(const int64_t *iv_start, const int32_t *iv_len, int64_t n_intervals, int32_t gap_shift,
                             int64_t *win_start, int64_t *win_len, int64_t *win_base, int64_t capacity, int64_t *n_windows,
                             int64_t *total, int64_t *out_iv_start);

/*
 * ---- host side: BAM front end ----------------------------------
 *
 * rp_bam_split_host replaces split_bam (bam.py:33-153) without pysam: the BGZF file is read
 * once, every alignment goes through the reference's decision list (qcfail, duplicate,
 * secondary, unmapped, then is_read_uniq_mapping of common.py:33-70: NH == 1, or MAPQ == 255
 * when there is no NH tag), the aligned length is the number of reference positions under
 * M/=/X, strand and 5' end follow the protocol (0 = forward, 1 = reverse; bam.py:108-128), and
 * the key (length, strand, chrom, pos + 1) is counted.  read_lengths (may be NULL / 0): keep
 * only these aligned lengths (bam.py:104-106).  The result -- the reference's nested Counter
 * as five columns sorted by (length, strand, chrom, pos), the reference names of the header
 * and the counters of {prefix}_bam_summary.txt -- is lent out by rp_bam_view_host until
 * rp_bam_free.
 */
typedef struct rp_bam rp_bam;

typedef struct rp_bam_view {
    int64_t n_rows, n_refs;
    const int32_t *length;
    const uint8_t *strand; /* 0 '+', 1 '-' */
    const int32_t *chrom;  /* reference id of the BAM header */
    const int64_t *pos;    /* 1-based 5' end */
    const int64_t *count;
    const char *ref_names;
    const int64_t *ref_off; /* name r = ref_names[ref_off[r] .. ref_off[r+1]) */
    int64_t n_lengths;
    const int32_t *length_order; /* aligned lengths in first-met order (the key order of the
                                    reference's read_length_counts dict) */
    int64_t total, valid, qcfail, duplicate, secondary, unmapped, multi;
} rp_bam_view;

int rp_bam_split_host(const char *path, int protocol, const int32_t *read_lengths, int32_t n_lengths, rp_bam **out);
int rp_bam_view_host(const rp_bam *bam, rp_bam_view *view);
void rp_bam_free(rp_bam *bam);

/*
 * Measurement aid: while the tag is on (process-wide), scoring launches use a second instantiation of the
 * scoring kernel (rp::k_tile_score_probe -- same code, another name), so that a profiler's per-kernel
 * statistics of rp::k_tile_score hold the production launches only.  ribotricer_amd's
 * engine.tune_workspace brackets its candidate timings with it.  Returns the previous state.
 */
int rp_measurement_tag(int on);

/* repr(float) of CPython 3 into buf (>= 32 bytes, not NUL-terminated); returns the length. */
int rp_format_double_repr(double value, char *buf);

/* str(list_of_int) of CPython 3 ("[a, b, c]") into out (>= 2 + 13*n bytes); returns the length. */
size_t rp_format_int_list(const int32_t *values, int64_t n, char *out);

/*
 * The body of one variableStep block of export_wig (detect_orfs.py:346-351): "{pos}\t{count}\n" for n
 * positions into out (>= 42*n bytes); returns the length.  The "variableStep chrom=..." header lines
 * stay with the caller.
 */
size_t rp_format_wig_rows_host(const int64_t *pos, const int64_t *count, int64_t n, char *out);

/*
 * export_wig's ordering work (detect_orfs.py:327-345: the reference sorts the Counter's (chrom, pos) keys in Python)
 * for the rows of ONE strand of the merged P-site columns, in two native passes around a plain value sort:
 *   rp_wig_pack_host    (rank of the row's chromosome name, position, count) of every row with strand == strand_code
 *                       into one 64-bit word each (10 + 32 + 22 bits: the sort order of the words is the file's order);
 *                       in threads.  RP_ERR_ARG when a row does not fit (position outside [0, 2^32), count outside
 *                       [0, 2^22), >= 1024 chromosomes, a chromosome code outside the table): the caller sorts by columns.
 *   rp_wig_render_host  words [lo, hi) of the SORTED array (lo and hi on position boundaries) to text: the rows of one
 *                       position are added up (several read lengths can land on one P-site), "{pos}\t{count}\n" per
 *                       position, and "variableStep chrom={name}\n" (detect_orfs.py:343-351) in front of the first position
 *                       of every chromosome that STARTS in the range (names: the chromosome names in rank order,
 *                       concatenated; name_off[r] .. name_off[r + 1] is rank r's).  out must hold 42 * (hi - lo) bytes +
 *                       24 + the name's length for every rank; returns the bytes written.  Ranges are independent: the
 *                       caller renders them in threads and writes them in order.
 */
int rp_wig_pack_host(const uint8_t *strand, const int32_t *chrom, const int64_t *pos, const int64_t *count, int64_t n_rows,
                     int32_t strand_code, const int32_t *rank_of_chrom, int32_t n_chroms, uint64_t *packed, int64_t *n_packed);
size_t rp_wig_render_host(const uint64_t *sorted_words, int64_t lo, int64_t hi, const char *names, const int64_t *name_off,
                          char *out);

/*
 * Same as rp_phase_score_csr_dev (plan == NULL) or rp_phase_score_csr_plan_dev (plan given)
 * but brackets each internal launch with HIP events on `hip_stream`, synchronises, and
 * reports milliseconds: ms[0] tile-index pass (0 with a plan), ms[1] main scoring kernel,
 * ms[2] per-ORF finish kernel, ms[3] whole call.
 * For bench.py's roofline figure; not for production use (it blocks the host).
 */
int rp_phase_score_csr_dev_timed(int device, const int32_t *d_counts, const int64_t *d_offsets,
                                 int64_t n_orfs, int64_t total_nt, double *d_phase,
                                 int32_t *d_valid, int64_t *d_read_count,
                                 int32_t *d_min_codon_cov, uint8_t *d_flags, uint8_t *d_status,
                                 const rp_filter_params *filter, void *d_workspace,
                                 size_t workspace_bytes, int algo, const rp_plan *plan,
                                 void *hip_stream, float ms[4]);

#ifdef __cplusplus
}
#endif
#endif /* RIBOPHASE_H */
