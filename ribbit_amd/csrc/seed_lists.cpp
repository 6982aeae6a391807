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

}  // namespace rb
