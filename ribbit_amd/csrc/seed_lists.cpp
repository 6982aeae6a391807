// seed_lists.cpp -- host-side seed-list merges (see seed_lists.h).
#include "seed_lists.h"

#include <algorithm>

namespace rb {

namespace {
enum class Rel { Identical, NewInsideOld, OldInsideNew, Overlap };

inline Rel relate(int ns, int ne, int os, int oe) {
    if (os == ns && oe == ne) return Rel::Identical;
    if (os <= ns && oe >= ne) return Rel::NewInsideOld;
    if (ns <= os && ne >= oe) return Rel::OldInsideNew;
    return Rel::Overlap;
}
}  // namespace

// addSeedToSeedPositionsPerfect (parse_perfect_shiftxor.cpp:47-142).  The reference recurses in
// tail position with the merged interval (":97,:106,:118 ... return;"), dropping its pending
// removals; that is the `continue` of the outer loop here.
void perfect_add(SeedLists &sl, int seed_start, int seed_end, int mlen) {
    std::vector<RibbitSeed> &list = sl.perfect;
    std::vector<size_t> drop;
    for (;;) {
        drop.clear();
        const int rlen = seed_end - seed_start + mlen;          // seed_rlen (:52)
        bool restart = false;
        for (size_t i = list.size(); i-- > 0;) {                // newest first (:60)
            const RibbitSeed old = list[i];
            if (old.end < seed_start) break;                    // :70
            const int old_rlen = (old.end - old.start) + old.mlen;
            switch (relate(seed_start, seed_end, old.start, old.end)) {
                case Rel::Identical:                            // :73-76
                    if (old.mlen < mlen) return;
                    drop.push_back(i);
                    break;
                case Rel::NewInsideOld:                         // :79-82
                    if (rlen < old.mlen / 3) break;
                    return;
                case Rel::OldInsideNew:                         // :85-88
                    if (old_rlen < mlen / 3) break;
                    drop.push_back(i);
                    break;
                case Rel::Overlap: {                            // :91-125
                    const bool old_first = old.start < seed_start;
                    const int overlap = old_first ? old.end - seed_start + old.mlen : seed_end - old.start + mlen;
                    const int ms = old_first ? old.start : seed_start;
                    const int me = old_first ? seed_end : old.end;
                    bool merge = false;
                    if (old.mlen == mlen) {
                        merge = true;
                    } else if (old.mlen < mlen) {
                        if (mlen - overlap <= 1 && rlen / mlen < 3) merge = true;
                        else if (rlen - mlen - overlap <= old.mlen) return;
                    } else {
                        if (old.mlen - overlap <= 1 && old_rlen / old.mlen < 3) merge = true;
                        else if (old_rlen - old.mlen - overlap <= mlen) drop.push_back(i);
                    }
                    if (merge) { seed_start = ms; seed_end = me; mlen = old.mlen; restart = true; }
                    break;
                }
            }
            if (restart) break;
        }
        if (!restart) break;
    }
    for (size_t i : drop) list.erase(list.begin() + (ptrdiff_t)i);   // descending indices (:129-134)
    const int limit = (int)sl.length - mlen;                          // :137-139
    if (seed_end > limit) seed_end = limit;
    list.push_back(RibbitSeed{seed_start, seed_end, mlen, RIBBIT_RANK_P});
}


// ---------------------------------------------------------------------------------------------
// addSeedToSeedPositionsSubstitutions (parse_substitute_shiftxor.cpp:18-388).
//
// Every recursive call in the reference is in tail position (`from_index = add(...); return
// from_index;`), so it is a restart of this function with a new interval / motif / type and the
// already advanced cursor: the outer loop below.
namespace {

struct Cand { bool from_perfect; int idx; };

inline void retire(RibbitSeed &s) { s.type = RIBBIT_RANK_N; }

// retainNestedSeed (parse_perfect_shiftxor.cpp:18-29): keep the nested seed unless the parent plane
// has strictly more matches over [start, end)
inline bool keep_nested(const SeedLists &sl, int start, int end, int nested_mlen, int parent_mlen) {
    return !(sl.range_count(nested_mlen, start, end) < sl.range_count(parent_mlen, start, end));
}
// retainIdenticalSeeds (parse_perfect_shiftxor.cpp:31-43): ties go to the smaller plane index
inline bool keep_identical(const SeedLists &sl, int start, int end, int nested_mlen, int parent_mlen) {
    const int a = sl.range_count(nested_mlen, start, end), b = sl.range_count(parent_mlen, start, end);
    return a != b ? a > b : nested_mlen < parent_mlen;
}

// :48-116 -- perfect and substitution seeds that may touch [seed_start, ...), larger end first
void gather_candidates(const SeedLists &sl, int from_index, int seed_start, std::vector<Cand> &out) {
    const std::vector<RibbitSeed> &P = sl.perfect, &S = sl.subst;
    out.clear();
    bool more_p = !P.empty(), more_s = !S.empty();
    long pi = from_index, si = (long)S.size() - 1;
    while (more_p || more_s) {
        const bool take_p = more_p && (!more_s || !(S[si].end > P[pi].end));
        if (more_p && more_s) {
            // both lists live: the larger end goes first, and BOTH cursors are re-tested against
            // the ends read in this step (:92-115)
            const int p_end = P[pi].end, s_end = S[si].end;
            if (take_p) { if (P[pi].type != RIBBIT_RANK_N) out.push_back({true, (int)pi}); --pi; }
            else        { if (S[si].type != RIBBIT_RANK_N) out.push_back({false, (int)si}); --si; }
            if (pi < 0 || p_end < seed_start) more_p = false;
            if (si < 0 || s_end < seed_start) more_s = false;
        } else if (take_p) {                                  // :60-73
            const int p_end = P[pi].end;
            if (p_end >= seed_start) { if (P[pi].type != RIBBIT_RANK_N) out.push_back({true, (int)pi}); --pi; }
            if (pi < 0 || p_end < seed_start) more_p = false;
        } else {                                              // :76-89
            const int s_end = S[si].end;
            if (s_end >= seed_start) { if (S[si].type != RIBBIT_RANK_N) out.push_back({false, (int)si}); --si; }
            if (si < 0 || s_end < seed_start) more_s = false;
        }
    }
}

}  // namespace

int subst_add(SeedLists &sl, int seed_start, int seed_end, int mlen, int from_index, int seed_type) {
    std::vector<RibbitSeed> &P = sl.perfect, &S = sl.subst;
    std::vector<Cand> cands;
    constexpr int RP = RIBBIT_RANK_P, RQ = RIBBIT_RANK_Q, RS = RIBBIT_RANK_S;

    for (;;) {
        // :34-42 move the cursor to the first perfect seed that starts beyond seed_end
        while ((size_t)from_index < P.size() && P[from_index].start <= seed_end && (size_t)from_index != P.size() - 1) ++from_index;

        if (seed_end - seed_start < subst_seedlen_cutoff(mlen)) return from_index;      // :44

        gather_candidates(sl, from_index, seed_start, cands);

        const int seed_rend = seed_end + mlen, seed_len = seed_end - seed_start, seed_rlen = seed_len + mlen;
        bool restart = false;
        auto again = [&](int s, int e, int m, int t) { seed_start = s; seed_end = e; mlen = m; seed_type = t; restart = true; };

        for (const Cand &c : cands) {
            RibbitSeed &old = c.from_perfect ? P[c.idx] : S[c.idx];
            const int o_start = old.start, o_end = old.end, o_mlen = old.mlen, o_type = old.type;
            const int o_rend = o_end + o_mlen, o_len = o_end - o_start, o_rlen = o_rend - o_start;

            if (o_end < seed_start) break;                     // :150
            if (o_type == RIBBIT_RANK_N) continue;             // :152
            if (seed_end < o_start) continue;                  // :155

            const bool new_is_lower = (seed_type == RS && (o_type == RP || o_type == RQ)) || (seed_type == RQ && o_type == RP);
            const bool same_rank = (seed_type == RQ && o_type == RQ) || (seed_type == RS && o_type == RS);

            if (seed_start == o_start && seed_end == o_end) {                              // :158 identical
                if (new_is_lower) return from_index;
                if (seed_type == RQ && o_type == RS) { retire(S[c.idx]); }
                else if (same_rank) {
                    if (mlen % o_mlen == 0) return from_index;
                    if (o_mlen % mlen == 0) { retire(S[c.idx]); again(seed_start, seed_end, mlen, seed_type); break; }
                    if (!keep_identical(sl, seed_start, seed_end, mlen, o_mlen)) return from_index;
                    retire(S[c.idx]);
                    break;                                                                  // :188
                }
            } else if (o_start <= seed_start && seed_end <= o_end) {                        // :194 new inside old
                if (new_is_lower) return from_index;
                if ((seed_type == RQ && o_type == RS) || same_rank) {
                    const int merged_type = (seed_type == RS && o_type == RS) ? RS : RQ;    // :203
                    if (mlen == o_mlen) { S[c.idx].mlen = mlen; S[c.idx].type = merged_type; return from_index; }
                    if (mlen % o_mlen == 0) return from_index;
                    if (o_mlen % mlen == 0 || o_mlen < mlen) {
                        if (seed_rlen >= o_mlen - 1 || seed_rlen >= o_len - 1) {
                            S[c.idx].mlen = mlen; S[c.idx].type = merged_type; return from_index;
                        }
                    } else if (!keep_nested(sl, seed_start, seed_end, mlen, o_mlen)) {
                        return from_index;
                    }
                }
            } else if (seed_start <= o_start && o_end <= seed_end) {                        // :235 old inside new
                if (new_is_lower) {
                    if (o_mlen % mlen == 0) {                                               // :239
                        retire(old);
                        again(seed_start, seed_end, mlen, RQ); break;
                    }
                    if (mlen % o_mlen == 0 || o_mlen < mlen) {                              // :249
                        const bool many = seed_len / mlen > 3;
                        if ((many && o_rlen >= 3 * mlen - 1) ||
                            (!many && (o_rlen >= mlen - 1 || o_rlen >= seed_len - 1))) {
                            if (o_type != RP) retire(S[c.idx]);
                            again(seed_start, seed_end, o_mlen, RQ); break;
                        }
                    }
                } else if (seed_type == RQ && o_type == RS) {                               // :275
                    retire(S[c.idx]);
                    break;
                } else if (same_rank) {
                    if (o_mlen % mlen == 0) {                                               // :283
                        retire(S[c.idx]);
                    } else if (mlen % o_mlen == 0 || mlen > o_mlen) {                       // :288
                        if (o_rlen >= mlen - 1 || o_rlen >= seed_len - 1) {
                            retire(S[c.idx]);
                            again(seed_start, seed_end, o_mlen, seed_type); break;
                        }
                        if (keep_nested(sl, o_start, o_end, o_mlen, mlen)) continue;
                        retire(S[c.idx]);
                    } else if (o_mlen > mlen) {                                             // :303
                        if (keep_nested(sl, o_start, o_end, o_mlen, mlen)) continue;
                        retire(S[c.idx]);
                        again(seed_start, seed_end, mlen, seed_type); break;
                    }
                }
            } else {                                                                        // :318 partial overlap
                int overlap, ms, me;
                if (o_start < seed_start) {
                    const int reach = (o_mlen <= mlen) ? o_rend : o_end;
                    overlap = (seed_end <= reach ? seed_end : reach) - seed_start;
                    ms = o_start; me = seed_end;
                } else {
                    const int reach = (mlen <= o_mlen) ? seed_rend : seed_end;
                    overlap = (o_end <= reach ? o_end : reach) - o_start;
                    ms = seed_start; me = o_end;
                }
                if (o_mlen % mlen == 0 || o_mlen > mlen) {                                  // :343
                    const bool many = o_len / o_mlen > 3;
                    if ((many && overlap >= 3 * o_mlen - 1) ||
                        (!many && (overlap >= o_mlen - 1 || overlap >= o_len - 1))) {
                        retire(old);
                        again(ms, me, mlen, RQ); break;
                    }
                } else if (mlen % o_mlen == 0 || mlen > o_mlen) {                           // :362
                    const bool many = seed_len / mlen > 3;
                    if ((many && overlap >= 3 * mlen - 1) ||
                        (!many && (overlap >= mlen - 1 || overlap >= seed_len - 1))) {
                        if (o_type != RP) retire(S[c.idx]);
                        again(ms, me, o_mlen, RQ); break;
                    }
                }
            }
        }
        if (restart) continue;

        const int limit = (int)sl.length - mlen;                                            // :382-384
        if (seed_end > limit) seed_end = limit;
        S.push_back(RibbitSeed{seed_start, seed_end, mlen, seed_type});
        return from_index;
    }
}

}  // namespace rb
