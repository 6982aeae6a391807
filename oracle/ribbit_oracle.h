/*
 * ribbit_oracle.h -- CPU restatement of ribbit's shift-XOR tandem-repeat scan.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product (ribbit_amd/csrc, libribbit_hip.so)
 * never links, imports or calls this code.
 *
 * PARITY UNPINNED: the reference (SowpatiLab/ribbit @ 2024_10_08) ships no
 * tests, golden vectors or fixtures for this path, and its sources cannot be
 * compiled in this image (every hot-path translation unit includes
 * boost/dynamic_bitset.hpp; Boost is absent and may not be substituted).  This
 * restatement therefore follows the reference text line by line (citations on
 * every function) but has not been checked against outputs of the reference.
 *
 * Coordinates: everything is in sequence space p = 0..L-1.  The reference
 * stores position p at bit index L-1-p (fasta_utils.cpp:93); that reversal is a
 * storage detail and is not reproduced.  Planes are one bit per base inside the
 * oracle; rbo_plane / rbo_anchor_plane hand out byte-per-base copies.
 */
#ifndef RIBBIT_ORACLE_H
#define RIBBIT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* seed ranks, global_variables.cpp:28-34 */
enum { RBO_RANK_P = 5, RBO_RANK_Q = 4, RBO_RANK_S = 3, RBO_RANK_F = 2,
       RBO_RANK_C = 1, RBO_RANK_A = 0, RBO_RANK_N = -1 };

/* the tuple<int,int,int,int> exchanged between all stages (start, end, motif length, type) */
typedef struct { int32_t start, end, mlen, type; } rbo_seed_t;

/* one top-level addSeedToSeedPositions* call made by a scanner, in call order.
 * pos = scan position p at which the call was made (L for the end-of-sequence flush). */
typedef struct { int32_t pos, mlen, start, end; } rbo_call_t;

enum { RBO_LIST_PERFECT = 0, RBO_LIST_SUBST = 1, RBO_LIST_ANCHORED = 2 };

typedef struct rbo_ctx rbo_ctx;

/* encode + shift-XOR sweep (fasta_utils.cpp:78-122) with the shift range of ribbit.cpp:240-243 */
rbo_ctx *rbo_open(const char *seq, int64_t len, int m_lo, int m_hi);
void rbo_close(rbo_ctx *c);

int rbo_min_shift(const rbo_ctx *c);
int rbo_max_shift(const rbo_ctx *c);
int64_t rbo_length(const rbo_ctx *c);

/* byte-per-base views (valid until rbo_close); plane(shift) is X_shift, or XA_shift once
 * rbo_run_anchor_planes has run and shift is a motif length */
const uint8_t *rbo_plane(rbo_ctx *c, int shift);
const uint8_t *rbo_anchor_plane(rbo_ctx *c, int shift);        /* valid after rbo_run_anchor_planes */
/* the plane as the oracle holds it: bit p & 63 of word p >> 6 (what the refinement half reads) */
const uint64_t *rbo_plane_bits(const rbo_ctx *c, int shift);
const uint8_t *rbo_nmask(const rbo_ctx *c);
const uint8_t *rbo_codes(const rbo_ctx *c);                    /* 2-bit code per base, N -> 0 */

/* the four stages of processSequence, in order (fasta_utils.cpp:132,136,144-160,166) */
int rbo_run_perfect(rbo_ctx *c);
int rbo_run_subst(rbo_ctx *c);
int rbo_run_anchor_planes(rbo_ctx *c);
int rbo_run_anchored(rbo_ctx *c);
/* 3-way merge + filters of fasta_utils.cpp:187-224: seeds in the order they reach refinement */
int rbo_run_dispatch(rbo_ctx *c);

int64_t rbo_seeds(const rbo_ctx *c, int which, const rbo_seed_t **out);
int64_t rbo_calls(const rbo_ctx *c, int which, const rbo_call_t **out);
int64_t rbo_dispatch(const rbo_ctx *c, const rbo_seed_t **out);

/* range popcount of a plane over [start,end) -- the loop of parse_perfect_shiftxor.cpp:22-25 */
int rbo_range_count(const rbo_ctx *c, int shift, int start, int end);

/* number of times the defined-divergence guards fired (Q9 empty-list reads, out-of-range [j] quirk) */
int64_t rbo_guard_hits(const rbo_ctx *c);
/* statistics: range popcounts requested by the merges so far */
int64_t rbo_range_queries(const rbo_ctx *c);

/* ---- refinement front half (rows a13-a15), ribbit_oracle_refine.cpp ---- */
#define RBO_TABLE 1024
typedef struct {
    int32_t min_length[RBO_TABLE];      /* MINIMUM_LENGTH[k]; 0 = the value operator[] default-inserts (Q13) */
    int32_t perfect_units[RBO_TABLE];   /* PERFECT_UNITS[k] */
    float purity_threshold;             /* PURITY_THRESHOLD, always 0.85 in the reference (Q1) */
    int32_t continuous_ones_threshold;  /* cones_threshold = 3, ribbit.cpp:191 */
} rbo_refine_params_t;

/* one Smith-Waterman job as processSeedMotifWise (parse_smallmotif_seed.cpp:255-270) or the first level
 * of processSeed (parse_seed.cpp:379-404) sets it up: query = sequence.substr(query_start, query_length),
 * reference = the motif (pool + motif_offset, `atomicity` characters) repeated past ppr_length */
typedef struct {
    int32_t seed_index;     /* index into the dispatch list */
    int32_t seed_type, motif_length, atomicity;
    int32_t query_start, query_length, ppr_length;
    int32_t small;          /* 1: processSeedMotifWise (m <= 10), 0: processSeed */
    int32_t motif_offset;
} rbo_job_t;

void rbo_refine_params_default(rbo_refine_params_t *p, int m_lo, int m_hi);
/* needs rbo_run_dispatch() to have run; arrays stay valid until the next call */
int64_t rbo_refine_jobs(rbo_ctx *c, const rbo_refine_params_t *prm, const rbo_job_t **jobs, const char **pool);

/* Rows f1/f4: the BED text processSequence writes for this record (fasta_utils.cpp:187-242 ->
 * processSeedMotifWise / processSeed).  Alignments are computed by the REFERENCE's own SSW
 * (oracle/_ref/libssw_ref.so); everything around them is restated.  seq = the record's bases. */
const char *rbo_refine_bed(rbo_ctx *c, const rbo_refine_params_t *prm, const char *seq, const char *seq_id, int64_t *len);

#ifdef __cplusplus
}
#endif
#endif
