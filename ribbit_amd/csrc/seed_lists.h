// seed_lists.h -- host half of the scanners: the order-dependent seed-list merges that the
// reference runs inside processShiftXORs* (sparse, sequential, recursive: they stay on the host,
// SURVEY.md 8a rows a4/a7/a11).  Product code: must never include anything from oracle/.
#pragma once
#include <stdint.h>

#include <functional>
#include <vector>

#include "ribbit_hip.h"

namespace rb {

// (Round 2 tried an allocator whose resize() does not zero, so that the ranges of the parallel merges could fill their
// slices of the joined list on all threads: the joining halved, 90 -> 45 ms at chromosome size, but the merges proper ran
// 30-90 ms slower with it -- measured with both builds alternating on one box -- so the lists are plain vectors.)
using SeedVec = std::vector<RibbitSeed>;

// popcount of plane `shift` over [start, end)  (retainNestedSeed's loop); supplied by the caller
// so that this file has no device dependency (the API layer passes a GPU query).
using RangeCount = std::function<int(int shift, int start, int end)>;

struct SeedLists {
    int64_t length = 0;       // bset_size
    int min_motif = 2, max_motif = 100, min_shift = 1;
    SeedVec perfect, subst, anchored;
    RangeCount range_count;
    int64_t guard_hits = 0;   // defined-divergence guards
    // Where range_count's answers of the anchored stage come from, when the composed planes are stored (host_planes.h):
    // plane m = plane_words + (m - plane_lo) * plane_stride, for m = plane_lo..plane_hi.  Only a hint: the range-parallel
    // merge prefetches the words a coming call's queries will read (parallel_merge.cpp); null when there is nothing stored.
    const uint32_t *plane_words = nullptr;
    int64_t plane_stride = 0;
    int plane_lo = 0, plane_hi = -1;
};

// The same lists by reference, for the window stages' merges: when a record's calls are merged as independent
// position ranges in parallel (parallel_merge.h), a worker appends to a list of its own while the lists of the
// earlier stages are shared.  Member names follow SeedLists.
struct ListRefs {
    SeedVec &perfect, &subst, &anchored;
    const RangeCount &range_count;
    int64_t length;
    int max_motif;
    int64_t guard_hits = 0;
    // Q8 (parse_anchored_shiftxor.cpp:511-522): the coverage code writes list entries it indexes with a loop counter,
    // i.e. entries at the HEAD of the perfect / substitution list, wherever in the record the seed lies.
    // Parallel workers log the write instead of performing it; the merge then checks that it changes nothing (the
    // usual case: the entry is retired already and is given its own coordinates) or redoes the stage in order.
    // changed_then: the write would have changed its target as the target was WHEN it was logged, and the target lies in the worker's
    // own range -- the calls of that range behind it then ran against the wrong entry, whatever the entry looks like by the time the
    // ranges are walked (a later, ordinary retirement in the same range can make the logged write look like a no-op: fuzz seed 430991,
    // found late in round 4 with ranges of one call; the reference makes the write at once, parse_anchored_shiftxor.cpp:511-522)
    struct HeadWrite { RibbitSeed *target; RibbitSeed value; bool changed_then; };
    std::vector<HeadWrite> *head_write_log = nullptr;
    // What lets the parallel merge stay parallel when such a write does change an entry (round 3: it happens on most
    // chromosome-sized records).  head_reads[x] (x = 0 perfect, 1 substitution list): bit min(j, 63) set = the coverage code
    // read entry j of list x by loop counter.  head_changes[x]: when the writes are performed (no log), bit min(j, 63) set =
    // a write changed start, end or motif size of entry j (what those reads see; a change of the type alone is a retirement
    // like any other).
    uint64_t *head_reads = nullptr, *head_changes = nullptr;
    // with head_changes: the largest end among the old and new values of the entries such writes changed (-1: none)
    int *head_change_reach = nullptr;
    // Parallel workers (parallel_merge.h) share the lists of the earlier stages.  The only field that changes there is
    // `type` (a seed is retired), and the only place where the type of a seed OUTSIDE the worker's range steers a
    // decision is the candidate walk, which pushes the nearest seed to the left unless it is retired
    // (merge_types.cpp:64-93, parse_substitute_shiftxor.cpp:92-115).  Those reads and all retirements are logged so
    // that a range that saw a stale type can be redone.
    struct TypeWrite { RibbitSeed *seed; int32_t old_type; };
    struct TypeRead { const RibbitSeed *seed; bool live; };
    std::vector<TypeWrite> *undo = nullptr;
    std::vector<TypeRead> *foreign_reads = nullptr;
    int range_lo = INT32_MIN;         // seeds that end before this position belong to an earlier range
    // Seeds that start beyond range_hi belong to a LATER range; in call order nothing has touched them yet when this
    // range's calls are made, so their type is read from the snapshot taken before the stage (the walk meets the
    // nearest such seed at its cursor, and whether it is pushed decides which candidate the loop sees last: Q8).
    int range_hi = INT32_MAX;
    const int32_t *initial_types_perfect = nullptr, *initial_types_subst = nullptr;
    ListRefs(SeedVec &p, SeedVec &s, SeedVec &a, const RangeCount &rc, int64_t len, int mm)
        : perfect(p), subst(s), anchored(a), range_count(rc), length(len), max_motif(mm) {}
    explicit ListRefs(SeedLists &sl) : perfect(sl.perfect), subst(sl.subst), anchored(sl.anchored), range_count(sl.range_count), length(sl.length), max_motif(sl.max_motif) {}
};

// addSeedToSeedPositionsPerfect, parse_perfect_shiftxor.cpp:47-142
void perfect_add(SeedLists &sl, int seed_start, int seed_end, int mlen);

// seedlen_cutoffs of processShiftXORswithSubstitutions (parse_substitute_shiftxor.cpp:423)
inline int subst_seedlen_cutoff(int mlen) { return mlen > 30 ? mlen / 3 : 10; }
// seedlen_cutoffs of processShiftXORsAnchored (parse_anchored_shiftxor.cpp:572-573)
inline int anchored_seedlen_cutoff(int mlen) { return mlen >= 10 ? (int)(0.9 * mlen) : (mlen > 6 ? mlen : 10); }

// The cursor loop that opens addSeedToSeedPositionsSubstitutions / ...Anchored
// (parse_substitute_shiftxor.cpp:34-42): first index >= from whose start exceeds seed_end, capped at the
// last element.  advance(advance(i, a), b) == advance(i, max(a, b)) for any list whose starts do not
// change in between, so a run of calls that only move the cursor (they fail the length filter right
// after this loop, :44) can be replaced by one advance with the largest seed_end among them.
inline int advance_cursor(const SeedVec &list, int from, int seed_end) {
    while ((size_t)from < list.size() && list[from].start <= seed_end && (size_t)from != list.size() - 1) ++from;
    return from;
}

// addSeedToSeedPositionsSubstitutions, parse_substitute_shiftxor.cpp:18-388; returns the new cursor
int subst_add(ListRefs &sl, int seed_start, int seed_end, int mlen, int from_index, int seed_type);
inline int subst_add(SeedLists &sl, int seed_start, int seed_end, int mlen, int from_index, int seed_type) {
    ListRefs l(sl);
    const int r = subst_add(l, seed_start, seed_end, mlen, from_index, seed_type);
    sl.guard_hits += l.guard_hits;
    return r;
}


// cursors into the perfect and substitution lists carried between addSeedToSeedPositionsAnchored calls
struct Cursor2 { int perfect = 0, subst = 0; };

// addSeedToSeedPositionsAnchored (parse_anchored_shiftxor.cpp:113-534) incl. mergeAllLists
// (merge_types.cpp:11-189); range_count must answer on the COMPOSED planes XA_m.
Cursor2 anchored_add(ListRefs &sl, int seed_start, int seed_end, int mlen, Cursor2 from, int seed_type);
inline Cursor2 anchored_add(SeedLists &sl, int seed_start, int seed_end, int mlen, Cursor2 from, int seed_type) {
    ListRefs l(sl);
    const Cursor2 r = anchored_add(l, seed_start, seed_end, mlen, from, seed_type);
    sl.guard_hits += l.guard_hits;
    return r;
}

// 3-way merge by start + filters of fasta_utils.cpp:187-224: the seeds that reach refinement, in order
void dispatch_order(const SeedLists &sl, SeedVec &out);
// ... of three slices given by pointer
void dispatch_order_slices(const RibbitSeed *P, size_t np, const RibbitSeed *S, size_t ns, const RibbitSeed *A, size_t na, SeedVec &out);

}  // namespace rb
