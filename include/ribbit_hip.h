/*
 * ribbit_hip.h -- C ABI of the MI355X (gfx950) implementation of ribbit's per-sequence
 * shift-XOR tandem-repeat scan.
 *
 * The reference (SowpatiLab/ribbit @ 2024_10_08) has no plugin/FFI interface; its de-facto
 * boundary is the set of free functions processSequence() calls (fasta_utils.cpp:59-250).
 * Every entry point below names the reference interface it replaces.  All functions are
 * plain C: pointers and sizes only, no C++ or torch types.  INTEGRATION.md shows the glue a
 * ribbit maintainer would add to fasta_utils.cpp to call them.
 *
 * Conventions
 *   - every call returns 0 on success or a negative RIBBIT_E_* code; ribbit_hip_last_error()
 *     returns a human-readable message for the last failure on the calling thread;
 *   - a handle is bound to one GPU and one HIP stream and is single-thread-affine;
 *   - there is NO CPU fallback: if no gfx950 device is usable every call fails loudly;
 *   - host arrays returned through `T **out` are owned by the handle and stay valid until the
 *     next call that produces the same kind of output, or ribbit_hip_close();
 *   - positions are 0-based sequence positions p (the reference stores p at bit L-1-p,
 *     fasta_utils.cpp:93; that reversal is not part of this ABI).
 */
#ifndef RIBBIT_HIP_H
#define RIBBIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIBBIT_ABI_VERSION 1

enum {
    RIBBIT_OK = 0,
    RIBBIT_E_ARG = -1,       /* bad argument */
    RIBBIT_E_DEVICE = -2,    /* no usable gfx950 device / HIP runtime error */
    RIBBIT_E_STATE = -3,     /* call made in the wrong order (e.g. scan before load) */
    RIBBIT_E_NOMEM = -4,     /* host or device allocation failed */
    RIBBIT_E_OVERFLOW = -5,  /* event buffer could not be grown far enough */
    RIBBIT_E_INTERNAL = -6
};

/* seed ranks, global_variables.cpp:28-34 */
enum { RIBBIT_RANK_P = 5, RIBBIT_RANK_Q = 4, RIBBIT_RANK_S = 3, RIBBIT_RANK_F = 2,
       RIBBIT_RANK_C = 1, RIBBIT_RANK_A = 0, RIBBIT_RANK_N = -1 };

/* Scan parameters: the reference's process-wide globals that the path reads
 * (global_variables.h:29-34, ribbit.cpp:191,240-243, fasta_utils.cpp:165). */
typedef struct RibbitScanParams {
    int32_t min_motif;        /* MINIMUM_MLEN, -m (default 2) */
    int32_t max_motif;        /* MAXIMUM_MLEN, -M (default 100) */
    int32_t window_length;    /* 8  (ribbit.cpp:191) -- only 8 is supported */
    int32_t subst_threshold;  /* 7  (ribbit.cpp:191) */
    int32_t anchor_threshold; /* 6  (fasta_utils.cpp:165) */
    int32_t anchor_length;    /* 3  (ribbit.cpp:191) */
} RibbitScanParams;

/* The tuple<int,int,int,int> every reference stage exchanges: (start, end, motif length, type). */
typedef struct RibbitSeed { int32_t start, end, mlen, type; } RibbitSeed;

/* One maximal run found by the perfect scan, before addSeedToSeedPositionsPerfect:
 * [start, end) is a run of X_m one-bits over non-N bases; term says what closed it. */
enum { RIBBIT_TERM_ZERO = 0, RIBBIT_TERM_N = 1, RIBBIT_TERM_EOS = 2 };
typedef struct RibbitRun { int32_t start, end, mlen, term; } RibbitRun;

/* One top-level addSeedToSeedPositions* call a reference scanner would make, in call order:
 * pos is the scan position of the call (sequence length for the end-of-sequence flush). */
typedef struct RibbitCall { int32_t pos, mlen, start, end; } RibbitCall;

typedef struct RibbitHandle RibbitHandle;

/* Fills *p with the reference defaults for -m min_motif -M max_motif. */
void ribbit_scan_params_default(RibbitScanParams *p, int32_t min_motif, int32_t max_motif);

const char *ribbit_hip_last_error(void);
int ribbit_hip_abi_version(void);

/* Number of usable gfx950 devices (0 if none; never initialises a device context). */
int ribbit_hip_device_count(void);
/* The PCI bus id of a device ("0000:c1:00.0"), e.g. to find its counters under /sys/class/drm on a host whose other GPUs belong to
 * other jobs (tools/m500_probe.py). */
int ribbit_hip_device_pci_bus_id(int device, char *out, size_t cap);

/* Open a handle on `device`.  Replaces the globals set up in main() (ribbit.cpp:237-243). */
int ribbit_hip_open(const RibbitScanParams *params, int device, RibbitHandle **out);
int ribbit_hip_close(RibbitHandle *h);

/* Use an existing HIP stream (hipStream_t passed as void*) for all work; NULL = handle's own. */
int ribbit_hip_set_stream(RibbitHandle *h, void *hip_stream);

/*
 * Load one FASTA record (ASCII bases, no newlines).  Replaces the 2-bit encode of
 * fasta_utils.cpp:78-115: H2D copy + pack kernel -> device bit planes (left, right, N).
 * Must not be called between ribbit_hip_scan_perfect_begin and _end on the same handle (RIBBIT_E_STATE).
 * The shift-XOR sweep of fasta_utils.cpp:117-122 is never materialised; each scan kernel
 * recomputes X_s words in registers.
 */
int ribbit_hip_load_record(RibbitHandle *h, const char *ascii, int64_t length);
/* Same, but the ASCII bases are already in device memory (length bytes at dev_ascii).  dev_ascii stays the caller's and
 * is READ AGAIN by later calls on this record (the refinement kernels and the fetch of the bases for alignment): it must
 * stay valid and unchanged until the next load on this handle or ribbit_hip_close(). */
int ribbit_hip_load_record_device(RibbitHandle *h, const void *dev_ascii, int64_t length);
/* Same as ribbit_hip_load_record from page-locked host memory (ribbit_hip_host_alloc) that the caller keeps valid and
 * unchanged until the next load on this handle or ribbit_hip_close(): the upload runs asynchronously at link speed on
 * the handle's upload stream (the previous record's kernels keep running), and refinement reads the bases in place --
 * no host copy of the record is made.  This is what a streaming FASTA reader hands over (ribbit.cpp:269-280 is the
 * loop it replaces: getline + string += into one pageable std::string per record). */
int ribbit_hip_load_record_pinned(RibbitHandle *h, const char *pinned_ascii, int64_t length);
/* Page-locked host memory for ribbit_hip_load_record_pinned (hipHostMalloc / hipHostFree). */
int ribbit_hip_host_alloc(size_t bytes, void **out);
int ribbit_hip_host_free(void *p);

/*
 * Device hot loop of processShiftXORsPerfect (parse_perfect_shiftxor.cpp:173-223) without the
 * addSeed merge: all maximal runs with length >= min(cutoff, 32), sorted by (mlen, start).
 */
int ribbit_hip_scan_perfect_runs(RibbitHandle *h, const RibbitRun **out, size_t *n);

/*
 * The addSeedToSeedPositionsPerfect calls processShiftXORsPerfect would make, in its call
 * order (position-major, motif-minor, end-of-sequence flush last; cutoffs of :179,:193,:216).
 */
int ribbit_hip_perfect_calls(RibbitHandle *h, const RibbitCall **out, size_t *n);

/*
 * processShiftXORsPerfect (parse_perfect_shiftxor.h:10; called at fasta_utils.cpp:132):
 * scan + addSeedToSeedPositionsPerfect -> seed_positions_perfect.
 */
int ribbit_hip_seeds_perfect(RibbitHandle *h, const RibbitSeed **out, size_t *n);

/*
 * The addSeedToSeedPositionsSubstitutions calls processShiftXORswithSubstitutions would make
 * (parse_substitute_shiftxor.cpp:430-574), in its call order: window-scan kernel (8-wide window,
 * >= 7 matches) + per-motif state machine replay.
 */
int ribbit_hip_subst_calls(RibbitHandle *h, const RibbitCall **out, size_t *n);

/*
 * processShiftXORswithSubstitutions (parse_substitute_shiftxor.h:9; called at fasta_utils.cpp:136).
 * Runs the perfect stage first if it has not run.  Returns seed_positions_substut and the perfect
 * list as this stage leaves it (entries may have been re-typed to RIBBIT_RANK_N).
 */
int ribbit_hip_seeds_substitutions(RibbitHandle *h, const RibbitSeed **perfect, size_t *n_perfect,
                                   const RibbitSeed **subst, size_t *n_subst);

/*
 * The addSeedToSeedPositionsAnchored calls processShiftXORsAnchored would make
 * (parse_anchored_shiftxor.cpp:580-723), in its call order.  Runs the anchored stage's two kernels: the planes
 * kernel -- generateAnchoredShiftXORs (parse_anchored_shiftxor.h:10, called at fasta_utils.cpp:144) and the plane
 * composition of fasta_utils.cpp:146-160, written to device memory -- and the 6-of-8 window scan of those planes,
 * then the window state machine on the device.  After this call "plane m" means the composed plane for every motif
 * length m (as in the reference, fasta_utils.cpp:159).  Motif sizes up to 990 (the planes kernel widens its per-wave
 * halo with max_motif).
 */
int ribbit_hip_anchored_calls(RibbitHandle *h, const RibbitCall **out, size_t *n);

/*
 * processShiftXORsAnchored (parse_anchored_shiftxor.h:14; called at fasta_utils.cpp:166).  Runs the
 * earlier stages first if needed.  Returns all three seed lists as the stage leaves them.
 */
int ribbit_hip_seeds_anchored(RibbitHandle *h, const RibbitSeed **perfect, size_t *n_perfect,
                              const RibbitSeed **subst, size_t *n_subst,
                              const RibbitSeed **anchored, size_t *n_anchored);

/*
 * The 3-way merge by start of fasta_utils.cpp:187-224: the seeds handed to refinement
 * (processSeedMotifWise / processSeed), in order, RANK_N entries and seeds shorter than 0.9*m dropped.
 */
int ribbit_hip_dispatch_seeds(RibbitHandle *h, const RibbitSeed **out, size_t *n);

/*
 * Refinement of ONE record on several GPUs (fasta_utils.cpp:211-242 handles the dispatched seeds one after the other, and each
 * independently of the others): a handle on another GPU that has the same record loaded takes over a SLICE of the dispatch list --
 * `seeds` = n consecutive entries of the list ribbit_hip_dispatch_seeds returned on the handle that ran the stages (they are
 * copied; they may also be a slice of this handle's own list) -- makes the composed planes on its own device (the planes kernel
 * alone: no scan, no merge) and is then ready for ribbit_hip_refine_bed, whose text is the BED rows of exactly those seeds.  The
 * slices' texts, in order, are the record's BED -- with ONE exception the caller must check: an alignment with an empty query sees
 * the previous seed's CIGAR (the reference's shared Alignment object); inside a slice that is resolved exactly, across a slice's
 * first seed it cannot be, so if ribbit_hip_refine_met_empty_query is 1 for any slice the record is refined again in one piece.
 * After this call the handle's seed lists are not available (ribbit_hip_seeds_* run the stages again from the next load on).
 */
int ribbit_hip_adopt_dispatch(RibbitHandle *h, const RibbitSeed *seeds, size_t n);
/* 1 if the last ribbit_hip_refine_bed on this handle met an alignment with an empty query */
int ribbit_hip_refine_met_empty_query(const RibbitHandle *h);

/*
 * Thresholds the refinement scans read from the reference's globals: MINIMUM_LENGTH / PERFECT_UNITS
 * (global_variables.h:36-38, filled at ribbit.cpp:143-174,219-235; index = motif size, 0 = the value
 * operator[] default-inserts for a missing key), PURITY_THRESHOLD (always 0.85, -p is ignored) and
 * cones_threshold (3, ribbit.cpp:191).
 */
#define RIBBIT_TABLE 1024
typedef struct RibbitRefineParams {
    int32_t min_length[RIBBIT_TABLE];
    int32_t perfect_units[RIBBIT_TABLE];
    float purity_threshold;
    int32_t continuous_ones_threshold;
} RibbitRefineParams;
/* defaults for -m min_motif -M max_motif with no -l / --min-units / --perfect-units */
void ribbit_refine_params_default(RibbitRefineParams *p, int32_t min_motif, int32_t max_motif);

/*
 * One Smith-Waterman job as processSeedMotifWise (parse_smallmotif_seed.cpp:255-270) or the first level
 * of processSeed (parse_seed.cpp:379-404) sets it up: query = sequence.substr(query_start, query_length);
 * reference = the motif (motif_pool + motif_offset, `atomicity` characters) repeated until longer than
 * ppr_length; Align(query, ref, ppr_length, filter, &alignment, 15).
 */
typedef struct RibbitAlignJob {
    int32_t seed_index;      /* index into the dispatch list */
    int32_t seed_type, motif_length, atomicity;
    int32_t query_start, query_length, ppr_length;
    int32_t small;           /* 1: processSeedMotifWise (m <= 10); 0: processSeed */
    int32_t motif_offset;
} RibbitAlignJob;

/*
 * What the two striped passes of an alignment determine (ssw.c:843-891: score, end point, second best score outside
 * the mask window, begin point): everything of StripedSmithWaterman::Alignment except the CIGAR.
 * flag: 0 ok, 2 the reverse pass scored less than the forward pass, -1 not computed (job too large for the
 * GPU kernels: query_length > 8192 or ppr_length > 16384) -- align those with ribbit_ssw_align.
 */
typedef struct RibbitSswEnds {
    int32_t score, ref_end, query_end, score2, ref_end2, ref_begin, query_begin, flag;
} RibbitSswEnds;

/* The striped passes of n alignment jobs on the loaded record, batched on the GPU (one alignment per 16-lane DPP
 * row, the library's stripe order: results are the library's, Aligner::Align at parse_seed.cpp:404 /
 * parse_smallmotif_seed.cpp:270 with mask length mask_len).  out[j] belongs to jobs[j]. */
int ribbit_hip_ssw_passes(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const char *motif_pool, size_t pool_len,
                          int32_t mask_len, RibbitSswEnds *out);

/* longestContinuousMatches (parse_seed.cpp:26-44; calls at parse_seed.cpp:366, parse_smallmotif_seed.cpp:234)
 * for every dispatched seed at once, on the GPU: out[i] belongs to dispatch seed i. */
int ribbit_hip_seed_longest_runs(RibbitHandle *h, const int32_t **out, size_t *n);

/* Host-only twin (no GPU): the same scan over seeds[0..n) of a record given by its packed planes (LSB-first words as
 * ribbit_hip_packed_plane returns them, zero-padded by max_motif/32 + 4 words past word L/32).  The composed planes
 * XA_m (fasta_utils.cpp:143-161; generateAnchoredShiftXORs, parse_anchored_shiftxor.cpp:20-56) are not needed: the slice
 * a seed covers is recomputed from the packed planes, as the host merges of the GPU path do. */
int ribbit_host_longest_runs(const RibbitScanParams *params, int64_t length, const uint32_t *hi, const uint32_t *lo,
                             const uint32_t *brk, size_t nwords, const RibbitSeed *seeds, size_t n, int32_t *out);

/*
 * The refinement scans between dispatch and alignment for every dispatched seed: seed validity,
 * possibleMotifs / calculateRepeatClass / calculateAtomicity (parse_smallmotif_seed.cpp:76-188,
 * bitseq_utils.cpp:139-221) for m <= 10, mostFrequentLongerMotif / calculateAtomicityLongMotif
 * (parse_seed.cpp:153-256, bitseq_utils.cpp:116-137) for m > 10, calculateMotif (bitseq_utils.cpp:14-38).
 * Jobs come in the order the reference would run its alignments (the recursion of processSeed on
 * flanks happens after alignment and is not part of this list).
 */
int ribbit_hip_refine_jobs(RibbitHandle *h, const RibbitRefineParams *prm, const RibbitAlignJob **jobs, size_t *n,
                           const char **motif_pool);

/* Host-only variant (no GPU): same jobs from a dispatch list and host planes (xa may be NULL: recomputed); *jobs and
 * *motif_pool are malloc'ed, release with ribbit_refine_jobs_free(). */
int ribbit_host_refine_jobs(const RibbitScanParams *params, const RibbitRefineParams *prm, int64_t length,
                            const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                            const uint32_t *xa, size_t xa_stride, const RibbitSeed *dispatch, size_t n_dispatch,
                            RibbitAlignJob **jobs, size_t *n_jobs, char **motif_pool, size_t *pool_len);
void ribbit_refine_jobs_free(RibbitAlignJob *jobs, char *motif_pool);

/*
 * The alignment step of refinement: StripedSmithWaterman::Aligner().Align(query, ref, ref_len, Filter(),
 * &alignment, mask_len) as called at parse_seed.cpp:404 and parse_smallmotif_seed.cpp:270 (default scores:
 * match 2, mismatch 2, gap open 3, gap extend 1).  Own implementation with the library's exact results
 * (scores, coordinates, CIGAR with = X I D S).  Host function, no GPU needed.  The CIGAR is written to
 * cigar[0..cap) NUL-terminated; cigar_len receives its full length (> cap-1 means truncated).
 */
typedef struct RibbitAlignment {
    int32_t sw_score, sw_score_next_best;
    int32_t ref_begin, ref_end, query_begin, query_end, ref_end_next_best;
    int32_t mismatches;
    int32_t flag;        /* Align's return value */
    int32_t cigar_len;
} RibbitAlignment;
int ribbit_ssw_align(const char *query, int32_t query_len, const char *ref, int32_t ref_len, int32_t mask_len,
                     RibbitAlignment *out, char *cigar, size_t cap);
/* Test hook, host only: the same alignment against `motif` (atom bases) repeated past ref_len, made the way refinement finishes an
 * alignment whose end points and path come from the GPU -- passes, path as run-length operations, then the finish that reads the
 * reference through the motif's period instead of a spelt-out string.  Must equal ribbit_ssw_align on that string. */
int ribbit_debug_ssw_align_periodic(const char *query, int32_t query_len, const char *motif, int32_t atom, int32_t ref_len, int32_t mask_len,
                                    RibbitAlignment *out, char *cigar, size_t cap);

/* Whole alignments of n jobs on the loaded record with as much on the GPU as it takes (striped passes: ssw_kernels.hip; the
 * banded path search of ssw.c:590-775: ssw_path.hip; the host writes the CIGAR text and aligns what the kernels leave alone):
 * out[j] and the NUL-terminated CIGAR at cigars + cigar_off[j] are Aligner::Align's for jobs[j].  on_gpu[j] (may be NULL):
 * 0 aligned on the host, 1 passes on the GPU, 2 passes and path on the GPU. */
int ribbit_hip_ssw_align_jobs(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const char *motif_pool, size_t pool_len, int32_t mask_len,
                              RibbitAlignment *out, char *cigars, size_t cap, int64_t *cigar_off, int32_t *on_gpu);


/*
 * The rest of processSequence (fasta_utils.cpp:211-242): processSeedMotifWise (parse_smallmotif_seed.cpp:190-288)
 * / processSeed (parse_seed.cpp:318-464) for every dispatched seed -- motif discovery, alignment
 * (ribbit_ssw_align), processCIGARMotifWise / processCIGARWithPruning (process_cigar.cpp:126-336),
 * calculateMotifUnits, the recursion on flanks -- and the BED rows they print (11 tab-separated columns,
 * parse_seed.cpp:434-436).  *text points at handle-owned memory holding the rows of this record.
 */
int ribbit_hip_refine_bed(RibbitHandle *h, const RibbitRefineParams *prm, const char *sequence_id,
                          const char **text, size_t *len);


/* Host-only variant (no GPU; xa may be NULL: recomputed); *text is malloc'ed, release with ribbit_text_free(). */
int ribbit_host_refine_bed(const RibbitScanParams *params, const RibbitRefineParams *prm, const char *sequence, int64_t length,
                           const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                           const uint32_t *xa, size_t xa_stride, const RibbitSeed *dispatch, size_t n_dispatch,
                           const char *sequence_id, char **text, size_t *len);
void ribbit_text_free(char *text);

/*
 * ---- streaming FASTA ingest ---------------------------------------------------------------------------------------
 * Replaces the reader loop of ribbit.cpp:269-280 (getline + `sequence += line` into one pageable std::string per
 * record).  The file is read in 16-MB blocks; line bodies are copied once, straight into a page-locked buffer
 * (pinned != 0; plain malloc otherwise, for hosts without a GPU) that ribbit_hip_load_record_pinned uploads from
 * asynchronously and refinement reads in place.  Record boundaries follow the reference loop exactly: a '>' line ends
 * the previous record if it has any bases and names the next one (text up to the first space); other lines are appended
 * without their '\n'; the last record is handed out even when it is empty (*is_last = 1: ribbit.cpp:280 processes it
 * without the "Processing sequence" line).
 * ribbit_fasta_next returns 1 with a record, 0 after the last one, < 0 on error; *bases stays valid until
 * ribbit_fasta_release (any thread) or ribbit_fasta_close; released buffers are reused.
 */
typedef struct RibbitFastaReader RibbitFastaReader;
int ribbit_fasta_open(const char *path, int pinned, RibbitFastaReader **out);
int ribbit_fasta_next(RibbitFastaReader *r, const char **name, const char **bases, int64_t *length, int *is_last);
int ribbit_fasta_release(RibbitFastaReader *r, const char *bases);
int ribbit_fasta_close(RibbitFastaReader *r);
const char *ribbit_fasta_last_error(void);

/* How often the defined-divergence guards fired in the merges of this record (the
 * reference has undefined behaviour there; 0 on ordinary inputs). */
int64_t ribbit_hip_guard_hits(const RibbitHandle *h);

/*
 * Bits [start, end) of shift plane `shift` (X_shift; for a motif length, the composed plane XA_shift once the anchored
 * stage has run on this record -- ribbit_hip_anchored_calls / ribbit_hip_seeds_anchored), one byte per base.  Replaces reads of
 * lshift_xor_bsets[shift-MINIMUM_SHIFT][L-1-p] (fasta_utils.cpp:220-222, parse_seed.cpp:366).
 */
int ribbit_hip_plane_bits(RibbitHandle *h, int32_t shift, int64_t start, int64_t end, uint8_t *out);

/* Popcount of plane `shift` over [start, end): the loop of retainNestedSeed /
 * retainIdenticalSeeds (parse_perfect_shiftxor.cpp:18-43). */
int ribbit_hip_range_popcount(RibbitHandle *h, int32_t shift, int64_t start, int64_t end, int32_t *count);

/* Packed device planes copied back to the host (LSB-first: base p is bit p%32 of word p/32);
 * each array holds ribbit_hip_plane_words() words.  which: 0 left bit, 1 right bit, 2 N mask. */
int64_t ribbit_hip_plane_words(const RibbitHandle *h);
int ribbit_hip_packed_plane(RibbitHandle *h, int which, uint32_t *out_words);

/*
 * Host-only replay of scanner call lists through the order-dependent seed-list merges
 * (addSeedToSeedPositionsPerfect / ...Substitutions).  This is the host half of
 * ribbit_hip_seeds_*: it needs no GPU, only the packed planes (LSB-first words as returned by
 * ribbit_hip_packed_plane, padded with at least max_motif/32 + 4 zero words past word L/32) for
 * the range popcounts of retainNestedSeed / retainIdenticalSeeds.  Used when the call lists come
 * from elsewhere: other ranks in chunk-sharded multi-GPU runs, or the CPU tests of the merges.
 * The arrays in *out are malloc'ed; release them with ribbit_seed_lists_free().
 */
typedef struct RibbitSeedLists {
    RibbitSeed *perfect;  size_t n_perfect;
    RibbitSeed *subst;    size_t n_subst;
    RibbitSeed *anchored; size_t n_anchored;
    RibbitSeed *dispatch; size_t n_dispatch;   /* fasta_utils.cpp:187-224 order */
    int64_t guard_hits;   /* defined-divergence guards that fired */
} RibbitSeedLists;
/* xa: composed planes XA_m for m = min_motif..max_motif, xa_stride words each; may be NULL: the slices the merges
 * read are then recomputed from the packed planes (as in the GPU path, where the composed planes stay in HBM).
 * anchored_calls == NULL and xa == NULL: the anchored stage is not run (no dispatch list). */
int ribbit_host_replay_calls(const RibbitScanParams *params, int64_t length,
                             const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                             const uint32_t *xa, size_t xa_stride,
                             const RibbitCall *perfect_calls, size_t n_perfect_calls,
                             const RibbitCall *subst_calls, size_t n_subst_calls,
                             const RibbitCall *anchored_calls, size_t n_anchored_calls,
                             RibbitSeedLists *out);
void ribbit_seed_lists_free(RibbitSeedLists *lists);

/*
 * ---- chunk-sharded operation: one long record scanned by several GPUs ----------------------------
 * Every scan kernel's output is a stream of events whose values depend on the sequence only within a
 * bounded distance (perfect: 32 + M+2 bases to the right, 33 to the left; window scans: 8 + M+2 / 1;
 * anchored: 4*(M+2) + 16 / 2*(M+2) + 8).  A rank therefore loads its chunk plus halos as a record of
 * its own and runs the whole device side of every stage on it -- scan, pairing of the events, window state
 * machines, length filters -- keeping the run records and the addSeed calls it owns; those sparse 16-byte
 * records are gathered (gather-v over RCCL) and one host runs the order-dependent merges on them exactly
 * as for a single GPU (ribbit_hip_scan_perfect_chunk, ribbit_hip_stage_calls_chunk, ribbit_host_merge_chunks).
 *
 * Events are 64-bit: bits 0-31 position, 32-47 motif length, 48-51 kind (0 START, 1 END closed by a
 * mismatch / failing window, 2 END at an N, 3 END at the end of the loaded record).
 */
#define RIBBIT_EVENT_POS(e)  ((uint32_t)(e))
#define RIBBIT_EVENT_MLEN(e) ((uint32_t)((e) >> 32) & 0xffffu)
#define RIBBIT_EVENT_KIND(e) ((uint32_t)((e) >> 48) & 0xfu)
enum { RIBBIT_STAGE_PERFECT = 0, RIBBIT_STAGE_SUBST = 1, RIBBIT_STAGE_ANCHORED = 2 };

/* Perfect stage of one chunk: scan, keep the events owned ([own_lo, own_hi) local, shifted by pos_offset) and
 * pair them locally.  runs = complete runs; halves = the unmatched events at the chunk's edges (a motif's
 * leading END / trailing START: that run continues in a neighbouring chunk).  After gathering, the halves of
 * all ranks sorted by (motif, position) alternate START, END and pair up into the remaining runs. */
int ribbit_hip_perfect_runs_partial(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset,
                                    const RibbitRun **runs, size_t *n_runs, const uint64_t **halves, size_t *n_halves);

/* The same on the device end to end (no event ever reaches the host): scan the loaded piece, pair on the GPU, and
 * deliver one record per run START of the piece, ordered by (motif, start), into dst (caller's buffer of dst_cap
 * records; ideally pinned, see ribbit_hip_host_register) or, when dst is NULL, into handle-owned pinned memory;
 * *out is where they are.  A record is either a complete run owned by this chunk (START in [own_lo, own_hi), END
 * before own_hi; term = RIBBIT_TERM_*) or a place holder to skip (term = RIBBIT_RUN_NOT_OWNED: the run belongs to
 * a neighbour or is cut by the own range).  The cut runs -- at most two per motif -- go to half_dst / *halves:
 *   term RIBBIT_RUN_HALF_START     START owned, the END lies in a later chunk (end = -1);
 *   term RIBBIT_RUN_HALF_END + t   END owned (terminator t), the START lies in an earlier chunk (start = -1).
 * Over all chunks of a record the halves, ordered by (motif, position), pair up START, END into the remaining runs.
 * Positions are shifted by pos_offset.  A whole record is own_lo = 0, own_hi = INT64_MAX, pos_offset = 0: only
 * complete runs, no halves (that is what ribbit_hip_scan_perfect_runs does).
 * Replaces the run bookkeeping of parse_perfect_shiftxor.cpp:173-223 for one chunk of a chunk-sharded record. */
enum { RIBBIT_RUN_NOT_OWNED = -1, RIBBIT_RUN_HALF_START = 3, RIBBIT_RUN_HALF_END = 4 };
int ribbit_hip_scan_perfect_chunk(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset,
                                  RibbitRun *dst, size_t dst_cap, RibbitRun *half_dst, size_t half_dst_cap,
                                  const RibbitRun **out, size_t *n, const RibbitRun **halves, size_t *n_halves);

/* The same scan split in two, for callers that keep several handles (each has its own HIP stream) busy: _begin
 * enqueues the load's pack, the scan and the pairing kernels and returns at once; _end waits for them and
 * copies the records down (arguments as for ribbit_hip_scan_perfect_chunk).  With two handles, one record's
 * kernels run while the previous record's runs cross PCIe -- the double-buffered streaming of a multi-record
 * FASTA (ribbit.cpp:269-280 is the loop being pipelined).  A whole record: own_lo 0, own_hi INT64_MAX, offset 0. */
int ribbit_hip_scan_perfect_begin(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset);
/* wait != 0: the records are in place when the call returns.  wait == 0: the counts are final but the records are
 * still being copied; ribbit_hip_scan_perfect_wait (or the next _begin on the handle) completes the copy -- lets the
 * caller enqueue the next record's kernels while this one's results cross PCIe. */
int ribbit_hip_scan_perfect_end(RibbitHandle *h, RibbitRun *dst, size_t dst_cap, RibbitRun *half_dst, size_t half_dst_cap,
                                int wait, const RibbitRun **out, size_t *n, const RibbitRun **halves, size_t *n_halves);
int ribbit_hip_scan_perfect_wait(RibbitHandle *h);
/* _end without the copy to the host: the records stay where the pairing kernels wrote them.  *dev_runs / *dev_halves
 * are DEVICE pointers to *n / *n_halves RibbitRun records (complete: the call has waited for the kernels), valid until
 * the next scan on this handle.  For callers that move them on over RCCL / xGMI (the chunks' candidate seed intervals
 * of a chunk-sharded record, gathered to the rank that runs the host merge) instead of through host memory. */
int ribbit_hip_scan_perfect_end_device(RibbitHandle *h, const void **dev_runs, size_t *n, const void **dev_halves, size_t *n_halves);

/*
 * ---- chunk-sharded operation, window stages: only the calls that reach a merge leave a GPU ---------------------
 * The substitution / anchored stage of ONE CHUNK of a longer record, on the device end to end: the loaded record is the
 * chunk plus halos (a "piece", starting pos_offset bases into a record of record_length bases); the window scan, the
 * pairing of its pass-streaks, the per-motif window state machine (parse_substitute_shiftxor.cpp:430-574,
 * parse_anchored_shiftxor.cpp:580-723), the stage's length filter and the call order all run on this GPU, exactly as
 * for a whole record, and the chunk keeps the addSeed calls whose SCAN POSITION it owns: own_lo <= pos - pos_offset <
 * own_hi (piece coordinates; the chunks' own ranges partition 0..record_length, the last one includes record_length,
 * the position of the end-of-sequence calls).  Results are in record coordinates; concatenated over the chunks in order
 * they are the record's kept calls in the reference's call order -- 16 bytes per kept call is all that travels.
 *
 * Exactness.  A call is a function of the sequence between the start of its group of pass-streaks and the first
 * evaluated window behind it -- bounded by the locus, not by a constant.  The piece must reach
 *   right: 4 * (max_motif + 2) + 16 bases beyond own_hi (unless it ends where the record ends): checked, RIBBIT_E_ARG;
 *   left:  far enough that (a) an evaluated window lies between the first exact position (2 * (max_motif + 2) + 8 bases
 *          into the piece) + 16 and own_lo - 8, and (b) no owned call's group starts within 8 positions of that first exact
 *          position; own_lo itself must lie at least 2 * (max_motif + 2) + 8 + 32 + 16 bases into the piece (RIBBIT_E_ARG
 *          otherwise: the last 16 are the span of the anchored scan's group filter, whose dropped groups (b) never sees).
 *          (a) and (b) are CHECKED on every call: *inexact = 1 means a repeat or a block of N reaches further left than
 *          the halo (nothing else can) and the caller loads the chunk again with a longer left halo (the results of such a
 *          call must not be used).  A piece that starts where the record starts (pos_offset == 0) is always exact.
 * calls / pend / flush: handle-owned page-locked memory, valid until the next call for the same stage on this handle;
 * dev_calls / dev_pend: the same arrays in device memory (for RCCL), valid until the next scan on the handle.
 * pend[i] (NULL: none in this chunk): for a kept call made at an N or right behind a blocked stretch, the largest end of
 * any of THIS chunk's calls before it (-1 otherwise); tail_pend: the largest end of any call of this chunk (-1: none).
 * Across chunks the bound of such a call is the larger of pend[i] and the earlier chunks' tail_pend
 * (ribbit_host_merge_chunks does that).  flush: the end-of-sequence calls, last chunk only.
 */
typedef struct RibbitChunkCalls {
    const RibbitCall *calls; size_t n;
    const int32_t *pend;
    int32_t tail_pend;
    int32_t inexact;
    const RibbitCall *flush; size_t n_flush;
    const void *dev_calls, *dev_pend;
    int64_t streaks;          /* pass-streaks the piece's scan found (diagnostic) */
} RibbitChunkCalls;
int ribbit_hip_stage_calls_chunk(RibbitHandle *h, int stage /* RIBBIT_STAGE_SUBST | RIBBIT_STAGE_ANCHORED */, int64_t own_lo, int64_t own_hi,
                                 int64_t pos_offset, int64_t record_length, RibbitChunkCalls *out);

/* Words [word_lo, word_hi) of every composed plane XA_m of the loaded piece into out + motif_index * out_stride: a chunk
 * writes its own words straight into its place in the record's planes (e.g. a page-locked segment every rank of the node
 * maps) -- ribbit_hip_xa_words with a destination stride. */
int ribbit_hip_xa_words_strided(RibbitHandle *h, int64_t word_lo, int64_t word_hi, uint32_t *out, int64_t out_stride);

/* Host-only: the merging rank's half of the chunk-sharded path.  parts[0..nparts) in chunk order: the perfect stage's run
 * records and halves of every chunk (ribbit_hip_scan_perfect_chunk; place holders are skipped, halves paired across
 * chunks) and its kept calls of both window stages (ribbit_hip_stage_calls_chunk; only the host pointers are read).
 * Planes cover the whole record (see ribbit_host_replay_calls); xa is READ IN PLACE (3 GB for a chromosome), it must stay
 * valid during the call.  Runs addSeedToSeedPositionsPerfect / ...Substitutions / ...Anchored + mergeAllLists
 * (parse_perfect_shiftxor.cpp:47-142, parse_substitute_shiftxor.cpp:18-388, parse_anchored_shiftxor.cpp:113-534,
 * merge_types.cpp:11-189) and the dispatch merge of fasta_utils.cpp:187-224 exactly as for one GPU. */
typedef struct RibbitChunkPart {
    const RibbitRun *runs; size_t n_runs;
    const RibbitRun *halves; size_t n_halves;
    RibbitChunkCalls subst, anchored;
} RibbitChunkPart;
int ribbit_host_merge_chunks(const RibbitScanParams *params, int64_t length,
                             const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                             const uint32_t *xa, size_t xa_stride, const RibbitChunkPart *parts, size_t nparts,
                             RibbitSeedLists *out);

/* Test hook: run the device-side pairing on a caller-made event stream of a record of `length` bases,
 * as if one scan had left it in one region.  *flags = 0 for a well-formed stream, otherwise the PAIR_* bits of
 * device_planes.h (1 malformed event, 2 duplicate chunk, 4 starts and ends do not alternate, 8 unterminated run,
 * 16 no room); runs receives min(*n_runs, runs_cap) records. */
int ribbit_hip_debug_pair_events(RibbitHandle *h, const uint64_t *events, size_t n, int64_t length, RibbitRun *runs, size_t runs_cap,
                                 size_t *n_runs, uint32_t *flags);

/* Test hook: the first guess of the event-buffer capacity of the scans (0 = automatic).  A guess that is too small
 * makes a scan overflow its regions, after which it is sized for the fullest region and run again. */
int ribbit_hip_debug_set_event_capacity(RibbitHandle *h, size_t events);

/* Test hook: the window stages' merges run as independent position ranges on host threads (RIBBIT_THREADS, default
 * min(cores, 16)) wherever the call sequence can be cut; this sets the smallest number of calls per range (default 4096). */
void ribbit_debug_set_merge_min_range(size_t calls);
/* Test hook: what the last merge of a stage (0 substitution, 1 anchored) on the calling thread did: out = {ranges, ranges
 * merged again (after validation, or because their writes to list heads change an entry; low 16 bits) | ranges run again
 * behind a list-head change only because the changed entry lay within sight of their walks << 16, whole stage redone in order (0/1)
 * | range runs of the anchored stage's parallel passes, all passes together << 1,
 * list-head writes that changed an entry, first range empty (0/1) | parallel passes of the stage << 8}. */
void ribbit_debug_last_merge(int stage, int32_t out[5]);
/* Test hook: the anchored stage's merge runs its first parallel pass on the GPU for stages of 2^20 kept calls and more
 * (anchored_merge.hip: one lane per range of some 64 calls, the host threads taking the ranges of more than 192 calls meanwhile;
 * environment RIBBIT_DEVICE_MERGE_MIN / RIBBIT_DEVICE_MERGE_RANGE / RIBBIT_DEVICE_MERGE_MAX_CALLS change the three numbers).
 * out = {ranges the device merged, ranges it was given but left to the host threads (candidate lists beyond its LDS, budget of
 * passes), ranges the host threads merged meanwhile, ranges of the stage, ranges merged again by the validation walk} of the
 * calling thread's last anchored merge; the first three are zero when the device pass did not run. */
void ribbit_debug_last_device_merge(int32_t out[5]);
/* Test hook: in how many independent ranges the calling thread's last dispatch merge (fasta_utils.cpp:187-224) ran; 1 = the
 * sequential merge (no cuts, or a list did not split cleanly at them). */
int32_t ribbit_debug_last_dispatch_ranges(void);

/* possibleMotifs (parse_smallmotif_seed.cpp:76-188) of every dispatched seed with m <= 10 that reaches it, computed
 * by one GPU launch (small_motifs.hip) -- what ribbit_hip_refine_jobs / ribbit_hip_refine_bed use for those seeds.
 * head: 4 ints per dispatched seed {first record, early reports, classes, flags}; flags != 0: the seed has no device
 * result (m > 10, filtered out by the continuous-ones threshold: -1; more than 64 classes or reports: 1; record arena
 * full: 2) and the library runs the host twin for it.  records: 4 words each {rotation class, first start, last end,
 * units}: a seed's early reports (already filtered by MINIMUM_LENGTH / PERFECT_UNITS, in the reference's push order),
 * then its classes with their final state: all of them in order of first appearance when two or more pass the filters
 * at the seed's end (the reference reports those in its unordered_map's iteration order, which every key determines),
 * otherwise only the one that passes, or none.  Valid until the next call on the handle. */
int ribbit_hip_small_motifs(RibbitHandle *h, const RibbitRefineParams *prm, const int32_t **head, size_t *n_seeds,
                            const uint32_t **records, size_t *n_records);
/* Test hook: cumulative, process-wide: small-motif seeds that refinement took from the GPU's table / computed on the host. */
void ribbit_debug_small_motif_counters(int64_t out[2]);
/* Test hook: cumulative, process-wide: alignments refinement made / of them with the striped passes from the GPU / with the
 * path from the GPU (ribbit_hip_refine_bed batches them on the GPU for records with 400,000 dispatched seeds or more;
 * RIBBIT_GPU_SSW=0 / 1 forces it off / on). */
void ribbit_debug_alignment_counters(int64_t out[3]);
/* Test hook: cumulative, process-wide: {levels run, nodes put off, alignments of those nodes} of the level-by-level GPU
 * refinement of long-motif seeds' recursion trees (processSeed's recursion on the flanks, parse_seed.cpp:443-463: nodes of
 * RIBBIT_DEFER_MIN bases and more, default 700, are not refined where they are met but batched on the GPU, level by level). */
void ribbit_debug_level_counters(int64_t out[3]);

/* HIP events behind ribbit_hip_last_timing_ms are recorded by default; every record is a barrier packet between two
 * kernels of the stream (~6 us each on MI355X).  A caller that streams many records can switch them off per handle
 * (ribbit_hip_last_timing_ms then fails with RIBBIT_E_STATE for the pack / scan / GPU-side intervals). */
int ribbit_hip_set_timing(RibbitHandle *h, int32_t enabled);

/* Worker threads the host-side stages of this handle may use (window state machines, refinement);
 * 0 = default (environment RIBBIT_THREADS, else min(cores, 16)).  A caller that keeps several handles busy at
 * once -- ribbit-hip does, one per in-flight FASTA record (ribbit.cpp:269-280 processes them one by one) --
 * divides the cores among them with this. */
int ribbit_hip_set_host_threads(RibbitHandle *h, int32_t threads);

/* Page-lock a host buffer the caller owns (e.g. a shared-memory segment several ranks of one node write their
 * chunk's records into) so that ribbit_hip_scan_perfect_chunk can DMA straight into it.  hipHostRegister. */
int ribbit_hip_host_register(void *p, size_t bytes);
int ribbit_hip_host_unregister(void *p);

/* Words [word_lo, word_hi) of every composed plane XA_m (after the anchored stage's kernel ran),
 * motif-major, into out[(max_motif-min_motif+1) * (word_hi-word_lo)]. */
int ribbit_hip_xa_words(RibbitHandle *h, int64_t word_lo, int64_t word_hi, uint32_t *out);

/* Host-only: pair gathered perfect-stage events into runs (sorted by motif, start); *runs is malloc'ed
 * (release with ribbit_runs_free). */
int ribbit_host_perfect_runs_from_events(const RibbitScanParams *params, size_t nparts, const uint64_t *events,
                                         const uint64_t *counts, RibbitRun **runs, size_t *n);
void ribbit_runs_free(RibbitRun *runs);

/* Timing of the last call, milliseconds.  what: 0 pack kernel, 1 last scan kernel, 2 GPU side of
 * the last scan (kernel + pairing + state machine + sort + read-back), all by HIP events on the launch stream;
 * 3 everything after the pairing of the last window stage (device state machine, sort, read-back; wall clock);
 * 4 the host merge of the last window stage (wall clock); 5 that of the substitution stage when
 * ribbit_hip_seeds_anchored ran both stages; 6 / 7 the scan kernel of the substitution / anchored stage (HIP events; the
 * anchored stage runs as two kernels, 7 is both); 8 / 9 its planes kernel (anchors + composition) / its window-scan kernel. */
int ribbit_hip_last_timing_ms(const RibbitHandle *h, int what, double *ms);
/* Profiling aid (no effect on results): streams `nbytes` of the loaded record's ASCII buffer /
 * planes through calib_stream_read_kernel so that a PMC pass contains a launch with a known byte
 * count in the scan kernels' access shape.  nbytes is clamped to what is resident. */
int ribbit_hip_debug_stream_read(RibbitHandle *h, int64_t nbytes, int64_t *bytes_read);

/* Number of raw device events (run starts + run ends) the last scan produced. */
int64_t ribbit_hip_last_event_count(const RibbitHandle *h);

#ifdef __cplusplus
}
#endif
#endif
