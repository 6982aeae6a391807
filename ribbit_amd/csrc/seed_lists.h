// seed_lists.h -- host half of the scanners: the order-dependent seed-list merges that the
// reference runs inside processShiftXORs* (sparse, sequential, recursive: they stay on the host,
// SURVEY.md 8a rows a4/a7/a11).  Product code: must never include anything from oracle/.
#pragma once
#include <stdint.h>

#include <functional>
#include <vector>

#include "ribbit_hip.h"

namespace rb {

// popcount of plane `shift` over [start, end)  (retainNestedSeed's loop); supplied by the caller
// so that this file has no device dependency (the API layer passes a GPU query).
using RangeCount = std::function<int(int shift, int start, int end)>;

struct SeedLists {
    int64_t length = 0;       // bset_size
    int min_motif = 2, max_motif = 100, min_shift = 1;
    std::vector<RibbitSeed> perfect, subst, anchored;
    RangeCount range_count;
    int64_t guard_hits = 0;   // defined-divergence guards
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
inline int advance_cursor(const std::vector<RibbitSeed> &list, int from, int seed_end) {
    while ((size_t)from < list.size() && list[from].start <= seed_end && (size_t)from != list.size() - 1) ++from;
    return from;
}

// addSeedToSeedPositionsSubstitutions, parse_substitute_shiftxor.cpp:18-388; returns the new cursor
int subst_add(SeedLists &sl, int seed_start, int seed_end, int mlen, int from_index, int seed_type);


// cursors into the perfect and substitution lists carried between addSeedToSeedPositionsAnchored calls
struct Cursor2 { int perfect = 0, subst = 0; };

// addSeedToSeedPositionsAnchored (parse_anchored_shiftxor.cpp:113-534) incl. mergeAllLists
// (merge_types.cpp:11-189); range_count must answer on the COMPOSED planes XA_m.
Cursor2 anchored_add(SeedLists &sl, int seed_start, int seed_end, int mlen, Cursor2 from, int seed_type);

// 3-way merge by start + filters of fasta_utils.cpp:187-224: the seeds that reach refinement, in order
void dispatch_order(const SeedLists &sl, std::vector<RibbitSeed> &out);

}  // namespace rb
