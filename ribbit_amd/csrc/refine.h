// refine.h -- host side of the per-seed refinement scans that sit between seed dispatch and the
// Smith-Waterman alignment: motif discovery for small motifs
// (possibleMotifs, parse_smallmotif_seed.cpp:76-188), consensus motif for long ones
// (mostFrequentLongerMotif, parse_seed.cpp:153-256), atomicity (bitseq_utils.cpp:88-183) and the
// alignment-job set-up of processSeedMotifWise / processSeed.  Sparse per-seed work: host, as in the
// reference; the only bit scan (longestContinuousMatches, parse_seed.cpp:26-44) has a batched GPU
// kernel (seed_longest_run_kernel) and the host only consumes its per-seed results.
#pragma once
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "host_planes.h"
#include "ribbit_hip.h"
#include "seed_lists.h"
#include "ssw_exact.h"

namespace rb {

// possibleMotifs as the GPU computed it (small_motifs.hip): head = 4 ints per dispatched seed {first record, early
// reports, classes, flags}, flags != 0: not computed for this seed (the host twin runs); records = 4 words each
// {class, first start, last end, units}.
struct SmallMotifTable {
    const int32_t *head = nullptr;
    const uint32_t *records = nullptr;
};

// longest_runs[i] = longestContinuousMatches of dispatch seed i on its composed plane.
// best_rows (may be null): for seeds with m > 10, the window start mostFrequentLongerMotif selects
// (computed by long_motif_rows_kernel), or -1 to compute it on the host.
void build_align_jobs(const HostPlanes &hp, const RibbitRefineParams &prm, const SeedVec &dispatch,
                      const int32_t *longest_runs, const int32_t *best_rows, std::vector<RibbitAlignJob> &jobs,
                      std::string &motif_pool, unsigned host_threads = 1, size_t seed_lo = 0, size_t seed_hi = (size_t)-1,
                      const SmallMotifTable *small = nullptr);
// (seed_lo, seed_hi: only the seeds dispatch[seed_lo .. seed_hi); job.seed_index stays an index into dispatch)
// The same for the listed seeds only (increasing indices into dispatch); a seed's jobs are the same, in the same order, as in any other call
// profile: milliseconds the calling thread's build_align_jobs calls spent joining their chunks' results (sequential part)
double build_align_jobs_join_ms(bool reset);
double build_align_jobs_parallel_ms(bool reset);     // ... in their chunks' parallel region
// ... of several slices of the seed list (bounds[c] = [lo, hi) of slice c) in one parallel region; done(c, jobs, motif_pool) is called,
// on one of the worker threads, as soon as slice c is complete (roughly in order)
void build_align_jobs_slices(const HostPlanes &hp, const RibbitRefineParams &prm, const SeedVec &dispatch, const int32_t *longest_runs,
                             const int32_t *best_rows, unsigned host_threads, const SmallMotifTable *small,
                             const std::vector<std::pair<size_t, size_t>> &bounds,
                             const std::function<void(size_t, std::vector<RibbitAlignJob> &&, std::string &&)> &done);
void build_align_jobs_of(const HostPlanes &hp, const RibbitRefineParams &prm, const SeedVec &dispatch, const int32_t *longest_runs,
                         const int32_t *best_rows, const std::vector<uint32_t> &which, std::vector<RibbitAlignJob> &jobs, std::string &motif_pool,
                         unsigned host_threads, const SmallMotifTable *small);

// cumulative, process-wide: small-motif seeds refine_to_bed took from a SmallMotifTable / ran possibleMotifs for itself
void small_motif_counters(long &from_device, long &on_host);
// cumulative, process-wide: alignments refine_to_bed made / of them with the striped passes / with the path from the GPU
void alignment_counters(long &all, long &gpu_passes, long &gpu_paths);

// seed_sequence_length of parse_seed.cpp:342-349: seed + one motif, cut at the first N
int usable_length_host(const HostPlanes &hp, int start, int end, int m);

// host computation of the same quantity from the host copy of the composed planes (used by
// ribbit_host_refine_jobs, which has no device)
int longest_run_host(const HostPlanes &hp, int mlen, int start, int end);

// Everything processSequence does with the dispatched seeds (fasta_utils.cpp:211-242): processSeedMotifWise
// (parse_smallmotif_seed.cpp:190-288) for m <= 10, processSeed incl. its recursion on the flanks
// (parse_seed.cpp:318-464) for m > 10, alignment by ssw_exact, CIGAR processing (process_cigar.cpp:126-336),
// and the BED rows (11 tab-separated columns) appended to `bed`.  sequence = the record's bases.
// One run of BED rows: the rows of the seeds from `first_seed` on up to the next piece's first seed.  A record refined in
// several calls (slices of the seed list, seeds left for a later call) is put together by ordering the pieces by first_seed.
// path: empty for whole seeds; for the rows of a PART of a long-motif seed's recursion tree (below) the place of the part's
// first node in that tree -- pieces are ordered by (first_seed, path), which is the order processSeed prints in.
struct BedPiece { uint32_t first_seed; std::string text; std::string path; };

// processSeed (parse_seed.cpp:318-464) is a recursion: align the seed, print its row, then do the same with what is left on
// either side of the aligned repeat (:443-463).  The rows come out in pre-order (node, left subtree, right subtree), and a node
// depends on nothing but its own interval, so a node may be put off and done later, on the GPU, with many others -- the
// consensus-row kernel and the alignment kernels take it like any first-level seed -- as long as its rows end up in its place.
// A node's place is its path from the seed: '1' for the left flank, '2' for the right one; pre-order is the plain
// lexicographic order of the paths.  At -M 100 such nodes are few; at -M 500 (BASELINE.json configs[4]) the nodes of 700
// bases and more were 98 % of the consensus time and 92 % of the alignment time of refinement, on host threads.
struct DeferredNode {
    int32_t start, end, mlen, type;
    int32_t longest;        // longestContinuousMatches of the interval (computed before the node was put off: it decides whether the node exists)
    int32_t known_row;      // consensus row if known already (a first-level seed whose row the GPU scan made), else -1
    uint32_t root;          // dispatch index of the seed the node descends from
    std::string path;       // "" = the seed itself
};
struct Deferral {
    std::vector<DeferredNode> *out = nullptr;   // where nodes that are put off go (appended under `lock`); null: nothing is put off
    void *lock = nullptr;                        // std::mutex *
    int min_length = 0;                         // a node is put off when its usable length reaches this ...
    int max_query = 0, max_ref = 0;             // ... and its alignment fits the kernels (query, padded reference)
    const DeferredNode *nodes = nullptr;        // non-null: `dispatch` IS a list of nodes put off earlier (dispatch[k] = nodes[k]'s interval)
};

void refine_to_bed(const HostPlanes &hp, const char *sequence, const RibbitRefineParams &prm,
                   const SeedVec &dispatch, const int32_t *longest_runs, const int32_t *best_rows,
                   const std::string &sequence_id, std::string &bed, unsigned host_threads = 0,
                   const std::vector<RibbitAlignJob> *jobs = nullptr, const std::vector<SswEnds> *ends = nullptr,
                   const std::vector<SswPath> *paths = nullptr, size_t seed_lo = 0, size_t seed_hi = (size_t)-1,
                   bool *order_dependent = nullptr, const SmallMotifTable *small = nullptr,
                   const uint32_t *job_first = nullptr, const uint8_t *skip = nullptr, std::vector<BedPiece> *pieces = nullptr,
                   const std::vector<uint32_t> *only = nullptr, size_t job_first_base = 0, const Deferral *tree = nullptr);
// tree (optional, needs pieces): nodes of long-motif seeds' recursion trees that are worth a GPU batch are not done here but
// appended to tree->out, and the rows around them are cut into pieces that sort into place; with tree->nodes the call refines
// such nodes (one per entry of `dispatch`, with `only`).
// Whether a refine_to_bed call made by the CALLING THREAD since the last reset met an alignment with an empty query (the one order
// dependence between seeds: such an alignment sees the previous seed's CIGAR).  Inside a call that is resolved exactly; a caller that
// refines a record's seeds in SLICES of its own (several GPUs, include/ribbit_hip.h: ribbit_hip_adopt_dispatch) cannot resolve it
// across a slice's first seed and redoes the record in one piece when a slice reports it.
bool refine_met_empty_query(bool reset);

// job_first (optional, with jobs): job_first[i - job_first_base] = first job (index into jobs) of dispatch seed i, for every seed
// the call refines and the one after it (else it is worked out from the jobs' seed indices on every call).  With a base, `jobs`
// may hold the jobs of a slice of the seed list only.
// skip (optional, per dispatch seed): seeds left out of this call (their alignments are not ready); pieces must be given, and the
// output goes there instead of `bed`, cut at every seed left out.
// only (optional): refine exactly these seeds (indices into dispatch, in the order they are to be started: a piece carries its
// seed's index, so any order gives the same pieces), one piece each -- the seeds an earlier call left out.
// seed_lo, seed_hi: refine the seeds dispatch[seed_lo .. seed_hi) only (jobs, if given, are those of that range).
// order_dependent (optional): a slice cannot resolve the one order dependence between seeds (an empty query sees the
// previous seed's CIGAR) on its own; when it meets one it appends nothing, sets the flag and the caller redoes the
// whole record in one call.
// jobs / ends (optional): the first-level alignment jobs of build_align_jobs and the end points of their striped
// passes as the GPU computed them (flag -1 = not computed); such alignments only need the traceback here.

}  // namespace rb
