// seed_lists.cpp -- host-side seed-list merges (see seed_lists.h).
#include "seed_lists.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

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
    SeedVec &list = sl.perfect;
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

// a seed of a list that parallel workers share: retirement and the "is it retired" test (see ListRefs)
inline void retire_shared(ListRefs &sl, RibbitSeed &s) {
    const int32_t was = __atomic_load_n(&s.type, __ATOMIC_RELAXED);
    if (sl.undo && was != RIBBIT_RANK_N) sl.undo->push_back({&s, was});
    __atomic_store_n(&s.type, (int32_t)RIBBIT_RANK_N, __ATOMIC_RELAXED);
}
inline bool live_shared(const ListRefs &sl, const RibbitSeed &s) {
    if (s.start > sl.range_hi) {
        const size_t ip = (size_t)(&s - sl.perfect.data());
        if (sl.initial_types_perfect && ip < sl.perfect.size()) return sl.initial_types_perfect[ip] != RIBBIT_RANK_N;
        const size_t is = (size_t)(&s - sl.subst.data());
        if (sl.initial_types_subst && is < sl.subst.size()) return sl.initial_types_subst[is] != RIBBIT_RANK_N;
    }
    const bool live = __atomic_load_n(&s.type, __ATOMIC_RELAXED) != RIBBIT_RANK_N;
    if (s.end < sl.range_lo && sl.foreign_reads) sl.foreign_reads->push_back({&s, live});
    return live;
}

// retainNestedSeed (parse_perfect_shiftxor.cpp:18-29): keep the nested seed unless the parent plane
// has strictly more matches over [start, end)
inline bool keep_nested(const ListRefs &sl, int start, int end, int nested_mlen, int parent_mlen) {
    return !(sl.range_count(nested_mlen, start, end) < sl.range_count(parent_mlen, start, end));
}
// retainIdenticalSeeds (parse_perfect_shiftxor.cpp:31-43): ties go to the smaller plane index
inline bool keep_identical(const ListRefs &sl, int start, int end, int nested_mlen, int parent_mlen) {
    const int a = sl.range_count(nested_mlen, start, end), b = sl.range_count(parent_mlen, start, end);
    return a != b ? a > b : nested_mlen < parent_mlen;
}

// :48-116 -- perfect and substitution seeds that may touch [seed_start, ...), larger end first
void gather_candidates(const ListRefs &sl, int from_index, int seed_start, std::vector<Cand> &out) {
    const SeedVec &P = sl.perfect, &S = sl.subst;
    out.clear();
    bool more_p = !P.empty(), more_s = !S.empty();
    long pi = from_index, si = (long)S.size() - 1;
    while (more_p || more_s) {
        const bool take_p = more_p && (!more_s || !(S[si].end > P[pi].end));
        if (more_p && more_s) {
            // both lists live: the larger end goes first, and BOTH cursors are re-tested against
            // the ends read in this step (:92-115)
            const int p_end = P[pi].end, s_end = S[si].end;
            if (take_p) { if (live_shared(sl, P[pi])) out.push_back({true, (int)pi}); --pi; }
            else        { if (S[si].type != RIBBIT_RANK_N) out.push_back({false, (int)si}); --si; }
            if (pi < 0 || p_end < seed_start) more_p = false;
            if (si < 0 || s_end < seed_start) more_s = false;
        } else if (take_p) {                                  // :60-73
            const int p_end = P[pi].end;
            if (p_end >= seed_start) { if (live_shared(sl, P[pi])) out.push_back({true, (int)pi}); --pi; }
            if (pi < 0 || p_end < seed_start) more_p = false;
        } else {                                              // :76-89
            const int s_end = S[si].end;
            if (s_end >= seed_start) { if (S[si].type != RIBBIT_RANK_N) out.push_back({false, (int)si}); --si; }
            if (si < 0 || s_end < seed_start) more_s = false;
        }
    }
}

}  // namespace

int subst_add(ListRefs &sl, int seed_start, int seed_end, int mlen, int from_index, int seed_type) {
    SeedVec &P = sl.perfect, &S = sl.subst;
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
            // (the type through an atomic load: another range may be retiring this seed right now -- what this range then
            // did with a stale value is found by the validation pass, parallel_merge.cpp -- ThreadSanitizer, round 4)
            const int o_start = old.start, o_end = old.end, o_mlen = old.mlen, o_type = __atomic_load_n(&old.type, __ATOMIC_RELAXED);
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
                        if (c.from_perfect) retire_shared(sl, old); else retire(old);
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
                        if (c.from_perfect) retire_shared(sl, old); else retire(old);
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

// ---------------------------------------------------------------------------------------------
// Anchored stage.
namespace {

enum Src : uint8_t { FROM_P, FROM_S, FROM_A };
struct Cand3 { Src src; int idx; };

inline const RibbitSeed &seed_of(const ListRefs &sl, Src src, int idx) {
    return src == FROM_P ? sl.perfect[idx] : src == FROM_S ? sl.subst[idx] : sl.anchored[idx];
}

// One backward walk over a single list as the reference writes it three times in mergeAllLists:
// take entries while their end reaches seed_start; stop (and mark the list done) at the first end
// below seed_start or at the front of the list.
template <typename Push>
void walk_back(const SeedVec &list, long &i, int seed_start, Push push) {
    for (;;) {
        const int end = list[i].end;
        if (end >= seed_start) { push(i); --i; }
        if (i < 0 || end < seed_start) return;
    }
}

// mergeAllLists (merge_types.cpp:11-189).  Phase 1 interleaves perfect and substitution seeds by
// descending end (starting at the two cursors); phase 2 interleaves that candidate list -- walked
// from its LAST element, i.e. by ascending end -- with the anchored list walked from its end.
// Divergence D1: an empty substitution list is treated as exhausted (the reference reads it: UB).
void merge_all_lists(ListRefs &sl, Cursor2 from, int seed_start, std::vector<Cand3> &out) {
    const SeedVec &P = sl.perfect, &S = sl.subst, &A = sl.anchored;
    static thread_local std::vector<Cand3> ps;       // scratch, reused across the millions of calls of a record
    ps.clear();
    bool p_done = P.empty(), s_done = false;
    if (S.empty()) { s_done = true; ++sl.guard_hits; }
    long pi = from.perfect, si = from.subst;
    auto push_p = [&](long i) { if (live_shared(sl, P[i])) ps.push_back({FROM_P, (int)i}); };
    auto push_s = [&](long i) { if (live_shared(sl, S[i])) ps.push_back({FROM_S, (int)i}); };
    while (!(p_done && s_done)) {
        if (s_done) { walk_back(P, pi, seed_start, push_p); p_done = true; }            // :30-45
        else if (p_done) { walk_back(S, si, seed_start, push_s); s_done = true; }        // :47-62
        else {                                                                           // :64-93
            const int p_end = P[pi].end, s_end = S[si].end;
            if (s_end > p_end) { push_s(si); --si; } else { push_p(pi); --pi; }
            if (pi < 0 || p_end < seed_start) p_done = true;
            if (si < 0 || s_end < seed_start) s_done = true;
        }
    }

    out.clear();
    if (A.empty()) { out = ps; return; }                                                 // :103-106
    long ai = (long)A.size() - 1;
    auto push_a = [&](long i) { if (A[i].type != RIBBIT_RANK_N) out.push_back({FROM_A, (int)i}); };
    if (ps.empty()) { walk_back(A, ai, seed_start, push_a); return; }                    // :107-122
    long ci = (long)ps.size() - 1;
    bool c_done = false, a_done = false;
    while (!(c_done && a_done)) {                                                        // :124-187
        if (a_done) {
            for (;;) {                                                                   // :127-140
                const Cand3 c = ps[ci];
                const int end = seed_of(sl, c.src, c.idx).end;
                if (end >= seed_start) { out.push_back(c); --ci; }
                if (ci < 0 || end < seed_start) break;
            }
            c_done = true;
        } else if (c_done) {
            walk_back(A, ai, seed_start, push_a);                                        // :143-157
            a_done = true;
        } else {                                                                         // :160-185
            const Cand3 c = ps[ci];
            const int c_end = seed_of(sl, c.src, c.idx).end, a_end = A[ai].end;
            if (a_end > c_end) { out.push_back({FROM_A, (int)ai}); --ai; }               // no RANK_N test here (:168-171)
            else { out.push_back(c); --ci; }
            if (ci < 0 || c_end < seed_start) c_done = true;
            if (ai < 0 || a_end < seed_start) a_done = true;
        }
    }
}

}  // namespace

Cursor2 anchored_add(ListRefs &sl, int seed_start, int seed_end, int mlen, const Cursor2 from, int seed_type) {
    SeedVec &P = sl.perfect, &S = sl.subst, &A = sl.anchored;
    constexpr int RA = RIBBIT_RANK_A, RC = RIBBIT_RANK_C, RP = RIBBIT_RANK_P, RS = RIBBIT_RANK_S, RQ = RIBBIT_RANK_Q;
    struct Child { int idx, mlen, type; };
    static thread_local std::vector<Cand3> cands;    // scratch, reused across the millions of calls of a record
    static thread_local std::vector<Child> factor_children, nonfactor_children;

    // state the reference keeps in function-scope variables across loop iterations; the coverage code
    // after the loop reads whatever was left in them (Q8)
    int o_start = 0, o_end = 0, o_rend = 0, o_mlen = 0, o_type = 0;

    for (;;) {
        // :133-151 both cursors restart from the caller's values on every (tail-)recursion
        Cursor2 cur = from;
        while ((size_t)cur.perfect < P.size() && P[cur.perfect].start <= seed_end && (size_t)cur.perfect != P.size() - 1) ++cur.perfect;
        while ((size_t)cur.subst < S.size() && S[cur.subst].start <= seed_end && (size_t)cur.subst != S.size() - 1) ++cur.subst;

        if (seed_end - seed_start < anchored_seedlen_cutoff(mlen)) return cur;              // :153

        merge_all_lists(sl, cur, seed_start, cands);                                         // :156
        factor_children.clear();
        nonfactor_children.clear();

        const int seed_rend = seed_end + mlen, seed_len = seed_end - seed_start, seed_rlen = seed_len + mlen;
        bool restart = false;
        auto again = [&](int s, int e, int m, int t) { seed_start = s; seed_end = e; mlen = m; seed_type = t; restart = true; };
        auto retire_by_type = [&](int type, int idx) {                                       // :270-271 and twins
            if (type == RP) retire_shared(sl, P[idx]);
            else if (type == RS || type == RQ) retire_shared(sl, S[idx]);
        };

        for (const Cand3 &c : cands) {
            const RibbitSeed &old = seed_of(sl, c.src, c.idx);
            o_start = old.start; o_mlen = old.mlen; o_end = old.end; o_rend = o_end + o_mlen; o_type = __atomic_load_n(&old.type, __ATOMIC_RELAXED);

            if (o_end < seed_start) break;                                                   // :203
            if (o_type == RIBBIT_RANK_N) continue;                                           // :205
            if (seed_end < o_start) continue;                                                // :208
            const int o_len = o_end - o_start, o_rlen = o_rend - o_start;
            const bool same_rank = (seed_type == RA && o_type == RA) || (seed_type == RC && o_type == RC);

            if (seed_start == o_start && seed_end == o_end) {                                // :215 identical
                if (seed_type == RA && o_type > RA) return cur;
                if (seed_type == RC && o_type == RA) A[c.idx].type = RIBBIT_RANK_N;
            } else if (o_start <= seed_start && seed_end <= o_end) {                         // :231 new inside old
                if (o_type > seed_type) return cur;
                if (seed_type == RC && o_type == RA) continue;
                if (same_rank) {
                    if (mlen % o_mlen == 0 && mlen != 4) return cur;                         // :241
                    if (o_mlen % mlen == 0 && o_mlen != 4) {                                 // :246
                        if (seed_rlen >= o_mlen - 1 || seed_rlen >= o_len) {
                            A[c.idx].type = RIBBIT_RANK_N;
                            again(o_start, o_end, mlen, seed_type); break;
                        }
                        continue;
                    }
                    if (!keep_nested(sl, seed_start, seed_end, mlen, o_mlen)) return cur;    // :257
                    continue;
                }
            } else if (seed_start <= o_start && o_end <= seed_end) {                         // :265 old inside new
                if (o_type > seed_type) {
                    if (mlen % o_mlen == 0) {                                                // :268
                        if (o_rlen >= mlen - 2 || o_rlen >= seed_len - 2) {
                            retire_by_type(o_type, c.idx);
                            again(seed_start, seed_end, o_mlen, RC); break;
                        }
                        factor_children.push_back({c.idx, o_mlen, o_type});
                    } else if (o_mlen % mlen == 0 || o_mlen > mlen) {                        // :285 / :301 (same action)
                        if (o_mlen >= 4 * mlen || o_len >= 4 * mlen) {
                            retire_by_type(o_type, c.idx);
                            again(seed_start, seed_end, mlen, RC); break;
                        }
                    } else {                                                                 // :312
                        nonfactor_children.push_back({c.idx, o_mlen, o_type});
                    }
                } else if (seed_type == RC && o_type == RA) {                                // :319
                    A[c.idx].type = RIBBIT_RANK_N;
                } else if (same_rank) {                                                      // :323
                    if (o_mlen == mlen || !keep_nested(sl, o_start, o_end, o_mlen, mlen)) {
                        A[c.idx].type = RIBBIT_RANK_N;
                    } else if (mlen % o_mlen == 0) {                                         // :332
                        if (o_rlen >= mlen - 2 || o_rlen >= seed_len - 2) {
                            A[c.idx].type = RIBBIT_RANK_N;
                            again(seed_start, seed_end, o_mlen, seed_type); break;
                        }
                    }
                }
            } else {                                                                         // :351 partial overlap
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
                if (seed_type == RA && o_type > RC) {                                        // :376
                    if (mlen == o_mlen && overlap >= 4 * mlen) {
                        retire_by_type(o_type, c.idx);
                        again(ms, me, mlen, RC); break;
                    }
                    if (!(mlen % o_mlen == 0 || o_mlen % mlen == 0) && (overlap >= mlen - 1 || overlap >= seed_len - 1)) return cur;
                } else if ((seed_type == RA || seed_type == RC) && (o_type == RA || o_type == RC)) {   // :398
                    if (mlen == o_mlen) {
                        // the `seed_type == ... ;` statements at :402,:410,:420,:428 compare and discard: no effect
                        bool merge;
                        if (o_len >= seed_len)
                            merge = (seed_len >= 3 * mlen) ? (overlap >= 3 * mlen - 1 || overlap >= seed_len - 1)
                                                           : (overlap >= mlen - 1 || overlap >= seed_len - 1);
                        else
                            merge = (o_len >= 3 * o_mlen && (overlap >= 3 * o_mlen - 1 || overlap >= o_len - 1)) ||
                                    (!(o_len >= 3 * o_mlen && (overlap >= 3 * o_mlen - 1 || overlap >= o_len - 1)) &&
                                     seed_len < 3 * o_mlen && (overlap >= o_mlen - 1 || overlap >= o_len - 1));
                        if (merge) {
                            A[c.idx].type = RIBBIT_RANK_N;
                            again(ms, me, o_mlen, seed_type); break;
                        }
                    }
                }
            }
        }
        if (restart) continue;

        // :441-468 coverage of the new seed by non-factor perfect/substitution children.  Q8: the lists
        // are indexed with the LOOP COUNTER j, and children typed Q leave the previous values in place.
        // Divergence D2: an out-of-range j keeps the stale values (the reference reads out of bounds).
        if (!nonfactor_children.empty()) {
            int coverage = 0;
            uint32_t prev_start = 0xffffffffu;
            for (size_t j = 0; j < nonfactor_children.size(); ++j) {
                const int t = nonfactor_children[j].type;
                const SeedVec *src = t == RP ? &P : t == RS ? &S : nullptr;
                if (src) {
                    if (j < src->size()) {
                        if (sl.head_reads) sl.head_reads[src == &P ? 0 : 1] |= 1ull << std::min<size_t>(j, 63);
                        o_start = (*src)[j].start; o_mlen = (*src)[j].mlen; o_end = (*src)[j].end; o_rend = o_end + o_mlen;
                    }
                    else ++sl.guard_hits;
                }
                if ((uint32_t)o_rend >= prev_start) coverage = (int)((uint32_t)coverage + (prev_start - (uint32_t)o_start));
                else if (o_rend < seed_end) coverage += o_rend - o_start;
                else coverage += seed_end - o_start;
                prev_start = (uint32_t)o_start;
            }
            if (coverage > 0.5 * seed_len) return cur;                                       // :467
        }

        // :471-526 coverage by factor children, per child motif size (the reference's two unordered_maps;
        // operator[] on a missing key inserts 0)
        if (!factor_children.empty()) {
            std::vector<int> prev_of(sl.max_motif + 8, 0), cov_of(sl.max_motif + 8, 0);
            std::vector<char> has_cov(sl.max_motif + 8, 0);
            for (const Child &ch : factor_children) { prev_of[ch.mlen] = -1; cov_of[ch.mlen] = 0; has_cov[ch.mlen] = 1; }
            for (size_t j = 0; j < factor_children.size(); ++j) {
                const int t = factor_children[j].type;
                const SeedVec *src = t == RP ? &P : t == RS ? &S : nullptr;
                if (src) {
                    if (j < src->size()) {
                        if (sl.head_reads) sl.head_reads[src == &P ? 0 : 1] |= 1ull << std::min<size_t>(j, 63);
                        o_start = (*src)[j].start; o_mlen = (*src)[j].mlen; o_end = (*src)[j].end; o_rend = o_end + o_mlen;
                    }
                    else ++sl.guard_hits;
                }
                const uint32_t prev_start = (uint32_t)prev_of[o_mlen];
                has_cov[o_mlen] = 1;
                if ((uint32_t)o_rend >= prev_start) cov_of[o_mlen] = (int)((uint32_t)cov_of[o_mlen] + (prev_start - (uint32_t)o_start));
                else if (o_rend < seed_end) cov_of[o_mlen] += o_rend - o_start;
                else cov_of[o_mlen] += seed_end - o_start;
                prev_of[o_mlen] = o_start;
            }
            for (int f = 0; f < (int)cov_of.size(); ++f) {                                   // ascending factor sizes (:504-507)
                if (!has_cov[f] || !(cov_of[f] >= 0.8 * seed_len)) continue;
                mlen = f; seed_type = RC;                                                    // :509
                for (size_t j = 0; j < factor_children.size(); ++j) {                        // :511-522, stale start/end written back
                    const int t = factor_children[j].type;
                    SeedVec *dst = t == RP ? &P : t == RS ? &S : nullptr;
                    if (!dst) continue;
                    if (j >= dst->size()) { ++sl.guard_hits; continue; }
                    if (sl.head_reads) sl.head_reads[dst == &P ? 0 : 1] |= 1ull << std::min<size_t>(j, 63);
                    o_mlen = (*dst)[j].mlen;
                    if (o_mlen == f) {
                        const RibbitSeed value{o_start, o_end, o_mlen, RIBBIT_RANK_N};
                        if (sl.head_write_log) {                                                    // parallel worker: decided later
                            const RibbitSeed &now = (*dst)[j];
                            const bool mine = now.end >= sl.range_lo && now.start <= sl.range_hi;
                            const bool changes = now.start != value.start || now.end != value.end || now.mlen != value.mlen ||
                                                 __atomic_load_n(&now.type, __ATOMIC_RELAXED) != value.type;
                            sl.head_write_log->push_back({&(*dst)[j], value, mine && changes});
                        }
                        else {
                            RibbitSeed &tgt = (*dst)[j];
                            if (sl.head_changes && (tgt.start != value.start || tgt.end != value.end || tgt.mlen != value.mlen)) {
                                sl.head_changes[dst == &P ? 0 : 1] |= 1ull << std::min<size_t>(j, 63);
                                if (sl.head_change_reach) *sl.head_change_reach = std::max(*sl.head_change_reach, std::max(tgt.end, value.end));
                            }
                            tgt = value;
                        }
                    }
                }
                break;
            }
        }

        const int limit = (int)sl.length - mlen;                                             // :529-531
        if (seed_end > limit) seed_end = limit;
        A.push_back(RibbitSeed{seed_start, seed_end, mlen, seed_type});
        return cur;
    }
}

// fasta_utils.cpp:187-224 on three slices given by pointer (the range-parallel form reads its slices of the lists in place).
// `smallest` is a uint64_t compared with int starts; the picked list persists across iterations when no head is smaller
// (cannot happen for starts >= 0).
void dispatch_order_slices(const RibbitSeed *P, size_t np, const RibbitSeed *S, size_t ns, const RibbitSeed *A, size_t na, SeedVec &out) {
    size_t ip = 0, is = 0, ia = 0;
    int pick = -1;
    out.clear();
    while (ip < np || is < ns || ia < na) {
        uint64_t smallest = ~(uint64_t)0;
        if (ip < np && smallest > (uint64_t)(int64_t)P[ip].start) { smallest = (uint64_t)(int64_t)P[ip].start; pick = 0; }
        if (is < ns && smallest > (uint64_t)(int64_t)S[is].start) { smallest = (uint64_t)(int64_t)S[is].start; pick = 1; }
        if (ia < na && smallest > (uint64_t)(int64_t)A[ia].start) { smallest = (uint64_t)(int64_t)A[ia].start; pick = 2; }
        RibbitSeed seed;
        if (pick == 0 && ip < np) seed = P[ip++];
        else if (pick == 1 && is < ns) seed = S[is++];
        else if (pick == 2 && ia < na) seed = A[ia++];
        else break;
        if (seed.type == RIBBIT_RANK_N) continue;                                            // :213
        if (seed.end - seed.start >= 0.9 * seed.mlen) out.push_back(seed);                   // :224
    }
}

void dispatch_order(const SeedLists &sl, SeedVec &out) {
    dispatch_order_slices(sl.perfect.data(), sl.perfect.size(), sl.subst.data(), sl.subst.size(), sl.anchored.data(), sl.anchored.size(), out);
}

}  // namespace rb
