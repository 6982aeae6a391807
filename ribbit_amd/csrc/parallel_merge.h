// parallel_merge.h -- the window stages' seed-list merges (addSeedToSeedPositionsSubstitutions / ...Anchored,
// SURVEY.md 8a rows a7 / a11) over the calls a stage kept, run as independent position ranges on host threads.
//
// The reference makes the calls one by one; a call only ever reads or writes seeds whose interval intersects its
// own (or the union it grows into by merging with such seeds), plus the cursors, which are a function of the largest
// end seen so far.  Wherever the call sequence can be cut so that every interval before the cut lies left of some
// position p and every interval after it right of p -- p covered by no call and no seed of the earlier stages --
// the two halves do not interact: the right half sees the left half only as list entries that end before anything it
// looks for, which a single sentinel entry stands for.  The halves are merged in parallel into lists of their own and
// concatenated in call order, which is the order the reference appends in.
// One dependency crosses a cut: the candidate walk pushes the nearest earlier-stage seed to the LEFT of a call unless
// that seed is retired (merge_types.cpp:64-93), and it may lie in the range before.  Reads of such seeds' types are
// logged; after the parallel pass the ranges are checked in order and one that saw a type its left neighbour changed
// afterwards is merged again (its own retirements undone first).
// Two things a cut cannot localise at all:
//   * Q8: the anchored merge's coverage code reads and writes entries at the HEAD of the perfect / substitution lists by
//     loop counter, from anywhere in the record (parse_anchored_shiftxor.cpp:441-522).  Workers log the writes and the
//     entries they read that way.  After the parallel pass the ranges are walked in order: a range whose logged writes
//     would change an entry is merged again on its own with the writes made; if that changed what the by-counter reads see
//     (start, end, motif size) and a later range read such an entry, the ranges behind it run again in parallel against the
//     new heads -- one more pass per change that matters.  (Rounds 1-2 redid the whole stage in call order on the first such
//     write: never met on the test records, met on two of three chromosome-sized ones, 5 s each.)
//   * the first range appending nothing: later ranges assumed a non-empty list (merge_types.cpp:103 and
//     parse_substitute_shiftxor.cpp:48-116 take different paths for an empty list): the stage is then redone in call order.
// Product code: must never include anything from oracle/.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "ribbit_hip.h"
#include "seed_lists.h"

namespace rb {

// What a window stage hands its merge: the calls that pass the stage's length filter, in call order, each with
// the largest end among the earlier calls that can matter to its cursors (-1: none; pend == null: all -1), the
// largest end of any in-loop call, and the end-of-sequence calls (unfiltered, motif order).
struct KeptCalls {
    const RibbitCall *calls = nullptr;
    size_t n = 0;
    const int32_t *pend = nullptr;
    int32_t tail_pend = -1;
    const RibbitCall *flush = nullptr;
    size_t n_flush = 0;
};

struct MergeStats {
    unsigned ranges = 1;       // independent ranges the calls were cut into
    unsigned threads = 1;
    unsigned ranges_redone = 0;    // ranges that read a seed type an earlier range changed afterwards, merged again
    bool redone_in_order = false;
    long long head_writes = 0;     // Q8 writes to list heads that would change an entry (their ranges are done again with the writes made)
    unsigned passes = 1;           // parallel passes of the anchored stage (one more per head change that a later range read)
    unsigned ranges_run = 0;       // ranges run in those passes, all passes together (== ranges when there was one pass)
    unsigned stale_by_sight = 0;   // ranges run again behind a head change ONLY because the changed entry lay within sight of their walks (no by-counter read of it)
    bool first_range_empty = false;
    double prepare_ms = 0.0, merge_ms = 0.0, concat_ms = 0.0;      // concat_ms: the part of merge_ms spent joining the ranges' lists
    // anchored stage, inside merge_ms: wall time of the parallel passes, of the in-order walks behind them (with the ranges
    // they do again), and the ranges' own times in the parallel passes (sum over ranges, longest range)
    double pass_ms = 0.0, walk_ms = 0.0, range_ms_sum = 0.0, range_ms_max = 0.0;
    double device_ms = 0.0, device_apply_ms = 0.0;      // inside pass_ms: the device pass (AnchoredDevicePass::run) and putting its results into the ranges' states
    double device_meanwhile_ms = 0.0;                    // ... of which the host threads were merging their share of the ranges for this long
    unsigned device_host_share = 0;                      // ranges the host threads merged while the kernel ran (from the heavy end)
    unsigned device_ranges = 0, device_bailed = 0;       // ranges the device pass merged / left to the host (scratch overflow)
    double prep_parts[3] = {0, 0, 0};                    // inside prepare_ms, cumulative: the cuts; the ranges' cursors; the type snapshots (then: the ranges' states)
    double before_passes_ms = 0.0, flush_ms = 0.0;     // inside merge_ms too: what precedes the first pass; the end-of-sequence calls after the join
    std::vector<int> cut_pos;  // the positions the ranges were cut at (first entry INT32_MIN): uncovered by any call or earlier-stage seed
};

// dispatch_order (seed_lists.h: the 3-way merge of fasta_utils.cpp:187-224) over the ranges between `cut_pos`: a position
// that no seed of any list covers, with every seed before it in its list lying to its left and every seed after it to
// its right, splits the merge into independent halves (all heads left of it are smaller than all heads right of it).
// Whether the lists split that way at the given positions is CHECKED (split index by bisection, then every entry of
// every slice against its bounds, on the host threads); if any entry is out of place -- or negative, which the
// reference's unsigned comparison treats as huge -- the sequential merge runs instead.  Returns the number of ranges used.
unsigned dispatch_order_ranges(const SeedLists &sl, const std::vector<int> &cut_pos, unsigned threads, SeedVec &out);
unsigned last_dispatch_ranges();      // of the calling thread's last dispatch_order_ranges (1: the sequential merge ran)

// The first parallel pass of the anchored stage's merge as DEVICE work (anchored_merge.hip: one lane per range; the API layer
// supplies `run`, this file stays free of HIP).  The stage then cuts the calls into ranges of about `calls_per_range` calls (a
// few hundred: 10^5 ranges for a chromosome) instead of eight ranges per host thread; what comes back per range is exactly what a
// host worker leaves in its RangeState (parallel_merge.cpp), so the in-order validation walk, the ranges done again and every later
// pass are the host code unchanged.  A range with a non-zero status was not merged (its scratch lists overflowed): the host
// merges it.  run() returns false when the device pass could not be made at all (the host pass runs instead).
struct AnchoredDevicePass {
    struct RangeResult {
        const RibbitSeed *own = nullptr;      // the range's part of the anchored list (entry 0 of every range but the first: the sentinel)
        uint32_t own_n = 0, status = 0;
        int64_t guard_hits = 0;
        Cursor2 cursor;
        uint64_t head_reads[2] = {0, 0};
    };
    struct LogEntry { uint32_t range, list, index; int32_t value; };      // list 0 perfect, 1 substitution; value: old type / was live
    struct HeadEntry { uint32_t range, list, index; RibbitSeed value; bool changed_then; };      // changed_then: ListRefs::HeadWrite
    size_t min_calls = 1u << 20;          // smaller stages stay on the host threads
    size_t calls_per_range = 64;          // what the stage's calls are cut into (wherever a cut is valid)
    size_t max_range_passes = 6000;       // a range expected to take a lane more passes of its loop than this is the host threads' from the start
    // order: all ranges, by the work they are expected to be; the lanes take them from the front, the first device_limit of them
    // at most.  meanwhile(from_back, from_front):
    // called once while the kernel runs, if it could be launched -- the host threads take ranges from the back of `order` in it;
    // *from_back (page-locked, read by the lanes) = entries of `order` still left to the device, *from_front (written by the lanes) =
    // entries they have taken.  A range both sides took is the host's.  ranges[k].status != 0 for every range the device did not merge.
    std::function<bool(const SeedLists &lists, const KeptCalls &kc, const std::vector<size_t> &first, const std::vector<int> &cut_pos,
                       const std::vector<Cursor2> &start_cursor, const std::vector<uint32_t> &order, size_t device_limit,
                       const std::function<void(uint32_t *from_back, const uint32_t *from_front)> &meanwhile,
                       std::vector<RangeResult> &ranges, std::vector<LogEntry> &undo, std::vector<LogEntry> &reads, std::vector<HeadEntry> &heads)> run;
};

// lists.subst is rebuilt from kc (lists.perfect as the perfect stage left it)
void merge_subst_stage(SeedLists &lists, const KeptCalls &kc, unsigned threads, MergeStats *stats = nullptr);
// lists.anchored is rebuilt from kc (lists.perfect / lists.subst as the substitution stage left them)
void merge_anchored_stage(SeedLists &lists, const KeptCalls &kc, unsigned threads, MergeStats *stats = nullptr, const AnchoredDevicePass *device = nullptr);

// Replay of a full anchored call list (ribbit_hip_anchored_calls order: in-loop calls, then the end-of-sequence flush)
void replay_anchored_calls(SeedLists &lists, const RibbitCall *calls, size_t n, int64_t length);
// ... and of a full substitution call list
void replay_subst_calls(SeedLists &lists, const RibbitCall *calls, size_t n);

// The same two stages from a FULL call list in the scanner's call order (in-loop calls, then the end-of-sequence
// flush with pos == length): filtered and bounded here, then merged as above.
void merge_subst_stage_full(SeedLists &lists, const RibbitCall *calls, size_t n, unsigned threads, MergeStats *stats = nullptr);
void merge_anchored_stage_full(SeedLists &lists, const RibbitCall *calls, size_t n, unsigned threads, MergeStats *stats = nullptr, const AnchoredDevicePass *device = nullptr);

// threads the merges may use: `asked` if non-zero, else environment RIBBIT_THREADS, else min(cores, 16)
unsigned merge_threads(unsigned asked);

// test hook: smallest number of calls per range (default 4096); small values cut wherever a cut is valid
void set_merge_min_range(size_t calls);
// test hook: what the last merge of a stage (0 substitution, 1 anchored) on the calling thread did
MergeStats last_merge_stats(int stage);

}  // namespace rb
