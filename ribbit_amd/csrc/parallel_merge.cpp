// parallel_merge.cpp -- see parallel_merge.h.
#include "parallel_merge.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace rb {

namespace {

std::atomic<size_t> g_min_range{4096};
thread_local MergeStats tl_last_stats[2];      // [0] substitution, [1] anchored stage: last merge on this thread

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------- in call order
// Replay of anchored calls.  In the reference every in-loop call updates the cursor pair, but the end-of-sequence
// flush keeps the returned cursors only for the first of the two calls it makes when a motif has both a pending
// group and an unmerged open streak (parse_anchored_shiftxor.cpp:713 vs :688,:697,:706,:717).
template <class Lists>
struct AnchoredReplay {
    Lists &lists;
    Cursor2 cur;
    int pending_end = -1;   // largest seed_end among in-loop calls that only moved the cursors (:133-153)
    void catch_up() {
        if (pending_end < 0) return;
        cur.perfect = advance_cursor(lists.perfect, cur.perfect, pending_end);
        cur.subst = advance_cursor(lists.subst, cur.subst, pending_end);
        pending_end = -1;
    }
    void call(const RibbitCall &c, bool keeps_cursor) {
        if (c.end - c.start < anchored_seedlen_cutoff(c.mlen)) {
            if (keeps_cursor) pending_end = std::max(pending_end, c.end);
            return;
        }
        catch_up();
        const Cursor2 next = anchored_add(lists, c.start, c.end, c.mlen, cur, RIBBIT_RANK_A);
        if (keeps_cursor) cur = next;
    }
    // a call list (or its tail) that may hold the end-of-sequence flush
    void run(const RibbitCall *calls, size_t n, int64_t length) {
        for (size_t i = 0; i < n; ++i) {
            const RibbitCall &c = calls[i];
            const bool flush = c.pos == (int32_t)length;
            const bool first_of_two = flush && i + 1 < n && calls[i + 1].pos == c.pos && calls[i + 1].mlen == c.mlen;
            call(c, !flush || first_of_two);
        }
    }
};

// Calls that fail the length filter only move the perfect-list cursor (parse_substitute_shiftxor.cpp:34-44); their
// effect is folded into one advance with the running maximum of their ends (seed_lists.h: advance_cursor).
template <class Lists>
struct SubstReplay {
    Lists &lists;
    int from_index = 0;
    int pending_end = -1;
    void call(const RibbitCall &c) {
        if (c.end - c.start < subst_seedlen_cutoff(c.mlen)) { pending_end = std::max(pending_end, c.end); return; }
        if (pending_end >= 0) { from_index = advance_cursor(lists.perfect, from_index, pending_end); pending_end = -1; }
        from_index = subst_add(lists, c.start, c.end, c.mlen, from_index, RIBBIT_RANK_S);
    }
};

// ------------------------------------------------------------------------------------ where to cut
// positions covered by some interval [start, end] (closed: seeds that share a position interact)
struct Coverage {
    // (zero pages from the system, touched first by the threads that mark them: a vector zeroes its 31 MB per chromosome on one thread)
    struct Free { void operator()(uint64_t *p) const { std::free(p); } };
    std::unique_ptr<uint64_t[], Free> w;
    int64_t top;      // largest position
    explicit Coverage(int64_t length) : w(static_cast<uint64_t *>(std::calloc((size_t)(length / 64 + 2), sizeof(uint64_t)))), top(length) {
        if (!w) throw std::bad_alloc();
    }
    // (atomic: the intervals are marked by several threads at once, and neighbours in call order share words)
    void set(int64_t word, uint64_t bits) { __atomic_fetch_or(&w[(size_t)word], bits, __ATOMIC_RELAXED); }
    void mark(int64_t a, int64_t b) {
        a = std::max<int64_t>(a, 0);
        b = std::min<int64_t>(b, top);
        if (b < a) return;
        const int64_t wa = a >> 6, wb = b >> 6;
        const uint64_t lo = ~0ull << (a & 63), hi = ~0ull >> (63 - (b & 63));
        if (wa == wb) { set(wa, lo & hi); return; }
        set(wa, lo);
        for (int64_t k = wa + 1; k < wb; ++k) set(k, ~0ull);
        set(wb, hi);
    }
    // first uncovered position in [from, to], -1 if none
    int64_t first_clear(int64_t from, int64_t to) const {
        from = std::max<int64_t>(from, 0);
        to = std::min<int64_t>(to, top);
        for (int64_t p = from; p <= to;) {
            const uint64_t free_bits = ~w[(size_t)(p >> 6)] & (~0ull << (p & 63));
            if (free_bits) {
                const int64_t q = ((p >> 6) << 6) + __builtin_ctzll(free_bits);
                return q <= to ? q : -1;
            }
            p = ((p >> 6) + 1) << 6;
        }
        return -1;
    }
};

// fn(lo, hi, t) on `threads` threads over [0, n) cut into contiguous pieces (piece t = [n t / T, n (t+1) / T))
// smallest piece worth a thread: 64 K items, or 16 x the smallest range when the test hook has made the ranges tiny (so that
// the tests that force ranges of 1, 5 and 64 calls also run these scans in many pieces)
size_t piece_grain() { return std::max<size_t>(1, std::min<size_t>(65536, g_min_range.load() * 16)); }
unsigned pieces_for(size_t n, unsigned threads) { return (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / piece_grain() + 1)); }

template <class Fn>
void parallel_pieces(size_t n, unsigned threads, Fn fn) {
    threads = pieces_for(n, threads);
    if (threads == 1) { fn((size_t)0, n, 0u); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(fn, n * t / threads, n * (t + 1) / threads, t);
    fn((size_t)0, n / threads, 0u);
    for (std::thread &th : pool) th.join();
}

// Running extrema over the kept calls, built by two-pass scans on the host threads:
//   end_before[i]  = largest end of the calls before i (-1 if none): what a cut before call i has on its left;
//   seen_before[i] = the same with the filtered-out calls' ends (pend) folded in: what the merge's cursors have seen;
//   start_from[i]  = smallest start of the calls from i on (INT32_MAX at n): what a cut before call i has on its right.
struct CallExtrema {
    // plain arrays that only grow: a vector would zero a quarter of a gigabyte on one thread before the scans overwrite it
    std::unique_ptr<int32_t[]> store;
    size_t cap = 0;
    int32_t *end_before = nullptr, *seen_before = nullptr, *start_from = nullptr;
    void build(const KeptCalls &kc, unsigned threads) {
        const size_t n = kc.n;
        if (n + 1 > cap) { cap = (n + 1) + (n + 1) / 8; store.reset(new int32_t[3 * cap]); }
        end_before = store.get(); seen_before = end_before + cap; start_from = seen_before + cap;
        const unsigned T = pieces_for(n, threads);
        std::vector<int32_t> top_end(T, -1), top_seen(T, -1), low_start(T, INT32_MAX);
        parallel_pieces(n, T, [&](size_t lo, size_t hi, unsigned t) {          // pass 1: piece-local scans
            int32_t e = -1, sn = -1;
            for (size_t i = lo; i < hi; ++i) {
                end_before[i] = e; seen_before[i] = sn;
                e = std::max(e, kc.calls[i].end);
                sn = std::max(sn, std::max(kc.calls[i].end, kc.pend ? kc.pend[i] : -1));
            }
            top_end[t] = e; top_seen[t] = sn;
            int32_t st = INT32_MAX;
            for (size_t i = hi; i-- > lo;) { st = std::min(st, kc.calls[i].start); start_from[i] = st; }
            low_start[t] = st;
        });
        std::vector<int32_t> carry_end(T, -1), carry_seen(T, -1), carry_start(T, INT32_MAX);
        for (unsigned t = 1; t < T; ++t) { carry_end[t] = std::max(carry_end[t - 1], top_end[t - 1]); carry_seen[t] = std::max(carry_seen[t - 1], top_seen[t - 1]); }
        for (unsigned t = T - 1; t-- > 0;) carry_start[t] = std::min(carry_start[t + 1], low_start[t + 1]);
        parallel_pieces(n, T, [&](size_t lo, size_t hi, unsigned t) {          // pass 2: the pieces before / after
            const int32_t ce = carry_end[t], cs = carry_seen[t], cst = carry_start[t];
            for (size_t i = lo; i < hi; ++i) {
                end_before[i] = std::max(end_before[i], ce);
                seen_before[i] = std::max(seen_before[i], cs);
                start_from[i] = std::min(start_from[i], cst);
            }
        });
        end_before[n] = n ? std::max(carry_end[T - 1], top_end[T - 1]) : -1;
        seen_before[n] = n ? std::max(carry_seen[T - 1], top_seen[T - 1]) : -1;
        start_from[n] = INT32_MAX;
    }
};

// first call of every range; a range boundary before call i is valid iff some position p, covered by no call and no
// seed of the static lists, has every earlier call's interval to its left and every later call's to its right.
// The coverage bitmap is filled and the running extrema are scanned on the host threads; the search itself looks at
// the candidates in call order (the first valid one at or after every `per`-th call), as a sequential walk would.
std::vector<size_t> cut_ranges(const KeptCalls &kc, std::initializer_list<const SeedVec *> statics, int64_t length,
                               size_t want_ranges, std::vector<int> &cut_pos, unsigned threads, CallExtrema &ext, size_t min_range_cap = 0) {
    const size_t n = kc.n;
    std::vector<size_t> first{0};
    cut_pos.assign(1, INT32_MIN);
    // (min_range_cap: the device pass wants ranges far smaller than what is worth a host thread)
    const size_t min_range = std::max<size_t>(min_range_cap ? std::min(g_min_range.load(), min_range_cap) : g_min_range.load(), 1);
    const size_t per = std::max(min_range, (n + want_ranges - 1) / std::max<size_t>(want_ranges, 1));
    if (n < 2 * per) return first;
    Coverage cov(length);
    for (const SeedVec *list : statics) {
        const SeedVec &l = *list;
        parallel_pieces(l.size(), threads, [&](size_t lo, size_t hi, unsigned) { for (size_t i = lo; i < hi; ++i) cov.mark(l[i].start, l[i].end); });
    }
    parallel_pieces(n, threads, [&](size_t lo, size_t hi, unsigned) { for (size_t i = lo; i < hi; ++i) cov.mark(kc.calls[i].start, kc.calls[i].end); });
    ext.build(kc, threads);
    // the first valid cut at or after every `per`-th call, from call `from` up to `to`
    auto search = [&](size_t from, size_t to, std::vector<size_t> &at, std::vector<int> &pos) {
        for (size_t i = from; i < to && n - i >= min_range;) {
            const int64_t left = (int64_t)ext.end_before[i] + 1, right = (int64_t)ext.start_from[i] - 1;
            const int64_t p = left <= right ? cov.first_clear(left, right) : -1;
            if (p >= 0) {
                at.push_back(i);
                pos.push_back((int)p);
                i += per;
            } else {
                ++i;
            }
        }
    };
    const unsigned pieces = min_range_cap ? pieces_for(n, threads) : 1u;
    if (pieces <= 1) { search(per, n, first, cut_pos); return first; }
    // (the device pass's hundreds of thousands of ranges: the calls in pieces, each searched from `per` calls behind its start --
    // whether a cut is valid does not depend on the cuts before it, only which of the valid ones are taken does)
    std::vector<std::vector<size_t>> at(pieces);
    std::vector<std::vector<int>> pos(pieces);
    parallel_pieces(n, pieces, [&](size_t lo, size_t hi, unsigned t) { search(lo + per, hi, at[t], pos[t]); });
    for (unsigned t = 0; t < pieces; ++t) { first.insert(first.end(), at[t].begin(), at[t].end()); cut_pos.insert(cut_pos.end(), pos[t].begin(), pos[t].end()); }
    return first;
}

unsigned resolve_threads(unsigned threads) { return std::max(1u, std::min(threads ? threads : std::thread::hardware_concurrency(), 256u)); }

template <class Work>
void run_ranges(size_t n_ranges, unsigned threads, Work work) {
    std::atomic<size_t> next{0};
    const bool reverse = std::getenv("RIBBIT_MERGE_REVERSE_RANGES") != nullptr;     // debugging aid
    auto loop = [&]() { for (size_t k; (k = next.fetch_add(1)) < n_ranges;) work(reverse ? n_ranges - 1 - k : k); };
    std::vector<std::thread> pool;
    const unsigned nt = std::getenv("RIBBIT_MERGE_SERIAL_RANGES") ? 1u : (unsigned)std::min<size_t>(threads, n_ranges);   // debugging aid: ranges one after the other
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(loop);
    loop();
    for (std::thread &t : pool) t.join();
}

// the ranges' lists, joined in range order (each without its sentinel).  `out` is NOT cleared first: resize() then only
// constructs what the list grows by (nothing, for a handle whose previous record was as large), instead of writing half a
// gigabyte of zeros on one thread that the copies overwrite at once; the copies run on the host threads.
template <class States>
void join_ranges(const States &state, size_t nr, SeedVec &out, unsigned threads) {
    std::vector<size_t> at(nr + 1, 0);
    for (size_t k = 0; k < nr; ++k) at[k + 1] = at[k] + state[k].own_size() - (k > 0 ? 1 : 0);
    out.resize(at[nr]);
    // (pieces of ranges, not single ranges, per turn of a thread: the device pass makes a hundred thousand of them)
    const size_t grain = std::max<size_t>(1, nr / ((size_t)threads * 8 + 1));
    run_ranges((nr + grain - 1) / grain, threads, [&](size_t g) {
        for (size_t k = g * grain; k < std::min(nr, (g + 1) * grain); ++k) {
            const size_t n = at[k + 1] - at[k];
            if (n) std::memcpy(out.data() + at[k], state[k].own_data() + (k > 0 ? 1 : 0), n * sizeof(RibbitSeed));
        }
    });
}

const RibbitSeed SENTINEL{-1, -1, 0, RIBBIT_RANK_N};   // stands for everything earlier ranges appended: ends before any interval of this range

std::vector<int32_t> types_of(const SeedVec &list) {
    std::vector<int32_t> t(list.size());
    for (size_t i = 0; i < list.size(); ++i) t[i] = list[i].type;
    return t;
}
void restore_types(SeedVec &list, const std::vector<int32_t> &t) {
    for (size_t i = 0; i < list.size(); ++i) list[i].type = t[i];
}

// ---- the stages in call order (also the fallback of the parallel versions)
void subst_in_order(SeedLists &lists, const KeptCalls &kc) {
    lists.subst.clear();
    SubstReplay<SeedLists> r{lists};
    for (size_t i = 0; i < kc.n; ++i) {
        if (kc.pend) r.pending_end = std::max(r.pending_end, kc.pend[i]);
        r.call(kc.calls[i]);
    }
    r.pending_end = std::max(r.pending_end, kc.tail_pend);
    for (size_t i = 0; i < kc.n_flush; ++i) r.call(kc.flush[i]);
}

void anchored_in_order(SeedLists &lists, const KeptCalls &kc) {
    lists.anchored.clear();
    AnchoredReplay<SeedLists> r{lists, Cursor2{}};
    for (size_t i = 0; i < kc.n; ++i) {
        if (kc.pend) r.pending_end = std::max(r.pending_end, kc.pend[i]);
        r.call(kc.calls[i], true);
    }
    r.pending_end = std::max(r.pending_end, kc.tail_pend);
    r.run(kc.flush, kc.n_flush, lists.length);
}

}  // namespace

void set_merge_min_range(size_t calls) { g_min_range = std::max<size_t>(calls, 1); }
MergeStats last_merge_stats(int stage) { return tl_last_stats[stage ? 1 : 0]; }

unsigned merge_threads(unsigned asked) {
    if (asked) return asked;
    if (const char *env = std::getenv("RIBBIT_THREADS")) return (unsigned)std::max(1, std::atoi(env));
    return std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
}

namespace {
// full call list -> the kept calls with cumulative bounds (the largest end of ANY earlier call), the flush apart
struct Compacted {
    std::vector<RibbitCall> calls, flush;
    std::vector<int32_t> pend;
    KeptCalls view;
};
void compact_full(const RibbitCall *calls, size_t n, int64_t length, int (*min_span)(int), Compacted &out) {
    int32_t seen = -1;
    size_t i = 0;
    for (; i < n && calls[i].pos != (int32_t)length; ++i) {
        const RibbitCall &c = calls[i];
        if (c.end - c.start >= min_span(c.mlen)) { out.calls.push_back(c); out.pend.push_back(seen); }
        seen = std::max(seen, c.end);
    }
    out.flush.assign(calls + i, calls + n);
    out.view.calls = out.calls.data();
    out.view.n = out.calls.size();
    out.view.pend = out.pend.data();
    out.view.tail_pend = seen;
    out.view.flush = out.flush.data();
    out.view.n_flush = out.flush.size();
}
}  // namespace

void merge_subst_stage_full(SeedLists &lists, const RibbitCall *calls, size_t n, unsigned threads, MergeStats *stats) {
    Compacted c;
    compact_full(calls, n, lists.length, subst_seedlen_cutoff, c);
    merge_subst_stage(lists, c.view, threads, stats);
}

void merge_anchored_stage_full(SeedLists &lists, const RibbitCall *calls, size_t n, unsigned threads, MergeStats *stats, const AnchoredDevicePass *device) {
    Compacted c;
    compact_full(calls, n, lists.length, anchored_seedlen_cutoff, c);
    merge_anchored_stage(lists, c.view, threads, stats, device);
}

void replay_anchored_calls(SeedLists &lists, const RibbitCall *calls, size_t n, int64_t length) {
    AnchoredReplay<SeedLists> r{lists, Cursor2{}};
    r.run(calls, n, length);
}

void replay_subst_calls(SeedLists &lists, const RibbitCall *calls, size_t n) {
    SubstReplay<SeedLists> r{lists};
    for (size_t i = 0; i < n; ++i) r.call(calls[i]);
}

// One range's private state: the list it appends to and the logs that let it be validated and redone.
struct RangeState {
    SeedVec own;                    // the stage's list, this range's part (own[0] = SENTINEL for ranges > 0)
    // ... or, after the device pass (AnchoredDevicePass), where that left it; reset() returns to `own`
    const RibbitSeed *ext = nullptr;
    size_t ext_n = 0;
    size_t own_size() const { return ext ? ext_n : own.size(); }
    const RibbitSeed *own_data() const { return ext ? ext : own.data(); }
    bool own_empty() const { return own_size() == 0; }
    std::vector<ListRefs::TypeWrite> undo;
    std::vector<ListRefs::TypeRead> foreign_reads;
    std::vector<ListRefs::HeadWrite> head_writes;
    uint64_t head_reads[2] = {0, 0};      // which list-head entries the coverage code read by loop counter (seed_lists.h)
    int64_t guard_hits = 0;
    Cursor2 cursor;
    void reset(bool sentinel, size_t expect) {
        for (size_t i = undo.size(); i-- > 0;) undo[i].seed->type = undo[i].old_type;
        undo.clear();
        foreign_reads.clear();
        ext = nullptr; ext_n = 0;
        own.clear();
        own.reserve(expect + 1);
        if (sentinel) own.push_back(SENTINEL);
        guard_hits = 0;
        head_writes.clear();
        head_reads[0] = head_reads[1] = 0;
    }
    // every seed of an earlier range whose type steered a decision here still has that type (types only ever go to
    // "retired", and earlier ranges are final when this is asked)
    bool reads_still_valid() const {
        for (const ListRefs::TypeRead &r : foreign_reads)
            if ((r.seed->type != RIBBIT_RANK_N) != r.live) return false;
        return true;
    }
};

// (The ranges' states are made anew for every stage.  Keeping them from record to record -- their lists are half a gigabyte
// for a chromosome's anchored stage, released when the stage returns -- was tried at the end of round 3 and made the stages
// SLOWER by 50 ms at chromosome-1 size in the bench (as much as not freeing the record's own lists at load time gains), with
// the preparation and the dispatch merge paying for fresh memory instead.  Why a reused list should be slower than a fresh one
// was not found: memory placement across the host's two sockets was the suspicion, but confining the process to one socket's
// cores (taskset) made nothing faster.)
// Runs body(k, state) for every range on `threads` threads, then walks the ranges in order and redoes those that read
// the type of an earlier range's seed before that range retired it.  Returns the number of ranges redone.
template <class Body>
unsigned run_and_validate(size_t nr, unsigned threads, const std::vector<size_t> &first, std::vector<RangeState> &state, Body body) {
    run_ranges(nr, threads, [&](size_t k) {
        state[k].reset(k > 0, first[k + 1] - first[k]);
        body(k, state[k]);
    });
    unsigned redone = 0;
    for (size_t k = 1; k < nr; ++k) {
        if (state[k].reads_still_valid()) continue;
        state[k].reset(true, first[k + 1] - first[k]);
        body(k, state[k]);
        ++redone;
    }
    return redone;
}

void merge_subst_stage(SeedLists &lists, const KeptCalls &kc, unsigned threads, MergeStats *stats) {
    MergeStats st;
    threads = resolve_threads(threads);
    st.threads = threads;
    const double t0 = now_ms();
    std::vector<size_t> first{0};
    std::vector<int> cut_pos;
    static thread_local CallExtrema ext;      // scratch that lives across the records of the calling thread
    if (threads > 1) first = cut_ranges(kc, {&lists.perfect}, lists.length, (size_t)threads * 8, cut_pos, threads, ext);
    const size_t nr = first.size();
    st.ranges = (unsigned)nr;
    if (nr == 1) {
        subst_in_order(lists, kc);
        st.merge_ms = now_ms() - t0;
        tl_last_stats[0] = st;
        if (stats) *stats = st;
        return;
    }
    first.push_back(kc.n);
    // the cursor every range starts from: a function of the largest end among the earlier calls (advance_cursor)
    std::vector<int> start_cursor(nr, 0);
    {
        int cur = 0;
        for (size_t k = 0; k < nr; ++k) {
            const int seen = ext.seen_before[first[k]];
            if (seen >= 0) cur = advance_cursor(lists.perfect, cur, seen);
            start_cursor[k] = cur;
        }
    }
    const std::vector<int32_t> perfect_types = types_of(lists.perfect);
    std::vector<RangeState> state(nr);
    st.prepare_ms = now_ms() - t0;
    const double t1 = now_ms();
    SeedVec no_anchored;
    st.ranges_redone = run_and_validate(nr, threads, first, state, [&](size_t k, RangeState &me) {
        ListRefs l(lists.perfect, me.own, no_anchored, lists.range_count, lists.length, lists.max_motif);
        l.undo = &me.undo;
        l.foreign_reads = &me.foreign_reads;
        l.range_lo = cut_pos[k];
        l.range_hi = k + 1 < nr ? cut_pos[k + 1] : INT32_MAX;
        l.initial_types_perfect = perfect_types.data();
        SubstReplay<ListRefs> r{l, start_cursor[k]};
        for (size_t i = first[k]; i < first[k + 1]; ++i) {
            if (kc.pend) r.pending_end = std::max(r.pending_end, kc.pend[i]);
            r.call(kc.calls[i]);
        }
        me.guard_hits = l.guard_hits;
        me.cursor.perfect = r.from_index;
    });
    const bool force_redo = std::getenv("RIBBIT_MERGE_FORCE_REDO") != nullptr;      // test hook: exercise the fallback
    if (state[0].own.empty() || force_redo) {      // later ranges assumed a non-empty list
        restore_types(lists.perfect, perfect_types);
        subst_in_order(lists, kc);
        st.redone_in_order = true;
    } else {
        join_ranges(state, nr, lists.subst, threads);
        for (size_t k = 0; k < nr; ++k) lists.guard_hits += state[k].guard_hits;
        SubstReplay<SeedLists> r{lists, state[nr - 1].cursor.perfect};
        r.pending_end = kc.tail_pend;
        for (size_t i = 0; i < kc.n_flush; ++i) r.call(kc.flush[i]);
    }
    st.merge_ms = now_ms() - t1;
    tl_last_stats[0] = st;
    if (stats) *stats = st;
}

void merge_anchored_stage(SeedLists &lists, const KeptCalls &kc, unsigned threads, MergeStats *stats, const AnchoredDevicePass *device) {
    MergeStats st;
    threads = resolve_threads(threads);
    st.threads = threads;
    const double t0 = now_ms();
    std::vector<size_t> first{0};
    std::vector<int> cut_pos;
    static thread_local CallExtrema ext;      // scratch that lives across the records of the calling thread
    // the device pass: ranges of a few hundred calls, one lane each (parallel_merge.h)
    if (device && !(device->run && kc.n >= device->min_calls && threads > 1 && !std::getenv("RIBBIT_MERGE_FORCE_REDO"))) device = nullptr;
    if (device) first = cut_ranges(kc, {&lists.perfect, &lists.subst}, lists.length, std::max<size_t>(kc.n / std::max<size_t>(device->calls_per_range, 1), (size_t)threads * 8),
                                   cut_pos, threads, ext, std::max<size_t>(device->calls_per_range, 1));
    else if (threads > 1) first = cut_ranges(kc, {&lists.perfect, &lists.subst}, lists.length, (size_t)threads * 8, cut_pos, threads, ext);
    const size_t nr = first.size();
    st.ranges = (unsigned)nr;
    st.prep_parts[0] = now_ms() - t0;
    if (nr == 1) {
        anchored_in_order(lists, kc);
        st.merge_ms = now_ms() - t0;
        tl_last_stats[1] = st;
        if (stats) *stats = st;
        return;
    }
    first.push_back(kc.n);
    std::vector<Cursor2> start_cursor(nr);
    if (!device) {
        Cursor2 cur;
        for (size_t k = 0; k < nr; ++k) {
            const int seen = ext.seen_before[first[k]];
            if (seen >= 0) {
                cur.perfect = advance_cursor(lists.perfect, cur.perfect, seen);
                cur.subst = advance_cursor(lists.subst, cur.subst, seen);
            }
            start_cursor[k] = cur;
        }
    } else {
        // The same cursors for a hundred thousand ranges, on the host threads.  `seen` only grows from range to range, so the chain
        // of advances above ends, for every range, at the FIRST entry of the whole list whose start exceeds its `seen` (capped at
        // the last entry): everything before the previous cursor starts at or below the previous, smaller, `seen`.  The first
        // entry whose start exceeds a value is the first whose running maximum of starts does: a bisection per range.
        auto running_max = [&](const SeedVec &l) {
            std::vector<int32_t> m(l.size());
            int32_t top = INT32_MIN;
            for (size_t i = 0; i < l.size(); ++i) { top = std::max(top, l[i].start); m[i] = top; }
            return m;
        };
        std::vector<int32_t> mp, ms;
        bool other_failed = false;
        {
            std::thread other([&]() { try { ms = running_max(lists.subst); } catch (...) { other_failed = true; } });
            struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{other};      // (also on the way out of an exception)
            mp = running_max(lists.perfect);
        }
        if (other_failed) throw std::bad_alloc();
        auto first_beyond = [](const std::vector<int32_t> &m, int seen) {
            if (m.empty()) return 0;
            const size_t i = (size_t)(std::upper_bound(m.begin(), m.end(), seen) - m.begin());
            return (int)std::min(i, m.size() - 1);
        };
        const int32_t *seen_before = ext.seen_before;         // (ext is the CALLING thread's scratch: a worker must not name it)
        parallel_pieces(nr, threads, [&, seen_before](size_t lo, size_t hi, unsigned) {
            for (size_t k = lo; k < hi; ++k) {
                const int seen = seen_before[first[k]];
                if (seen >= 0) { start_cursor[k].perfect = first_beyond(mp, seen); start_cursor[k].subst = first_beyond(ms, seen); }
            }
        });
    }
    st.prep_parts[1] = now_ms() - t0;
    const std::vector<int32_t> perfect_types = types_of(lists.perfect), subst_types = types_of(lists.subst);
    st.prep_parts[2] = now_ms() - t0;
    const int64_t guard_before = lists.guard_hits;
    std::vector<RangeState> state(nr);
    st.prepare_ms = now_ms() - t0;
    const double t1 = now_ms();
    st.cut_pos = cut_pos;
    // One range's calls.  live == false (the parallel pass): Q8's writes to list heads are logged, not made.  live == true
    // (a range done again on its own, everything before it final): they are made, and `changes` tells which entries' start /
    // end / motif size they changed.
    // Every kept call goes through the candidate walk, and half of them count matches over their interval on their
    // motif's composed plane -- a random place in 12 bytes per base of planes, i.e. a memory access that nothing has asked
    // for before.  The call list says where, several calls ahead (RIBBIT_MERGE_PREFETCH: how many; 0 = none).
    const char *prefetch_env = std::getenv("RIBBIT_MERGE_PREFETCH");      // read per stage: tools/merge_prefetch_sweep.py changes it between runs
    const size_t ahead = lists.plane_words ? (prefetch_env ? (size_t)std::strtoul(prefetch_env, nullptr, 10) : (size_t)8) : 0;
    auto body = [&](size_t k, RangeState &me, bool live, uint64_t *changes, int *change_reach) {
        ListRefs l(lists.perfect, lists.subst, me.own, lists.range_count, lists.length, lists.max_motif);
        l.head_write_log = live ? nullptr : &me.head_writes;
        l.head_reads = me.head_reads;
        l.head_changes = live ? changes : nullptr;
        l.head_change_reach = live ? change_reach : nullptr;
        l.undo = &me.undo;
        l.foreign_reads = &me.foreign_reads;
        l.range_lo = cut_pos[k];
        l.range_hi = k + 1 < nr ? cut_pos[k + 1] : INT32_MAX;
        l.initial_types_perfect = perfect_types.data();
        l.initial_types_subst = subst_types.data();
        AnchoredReplay<ListRefs> r{l, start_cursor[k]};
        for (size_t i = first[k]; i < first[k + 1]; ++i) {
            if (kc.pend) r.pending_end = std::max(r.pending_end, kc.pend[i]);
            if (ahead && i + ahead < first[k + 1]) {
                // what the nested-seed tests of a coming call will count over: its own motif's plane, its own interval
                const RibbitCall &n = kc.calls[i + ahead];
                if (n.mlen >= lists.plane_lo && n.mlen <= lists.plane_hi) {
                    const uint32_t *w = lists.plane_words + (int64_t)(n.mlen - lists.plane_lo) * lists.plane_stride;
                    __builtin_prefetch(w + (n.start >> 5), 0, 1);
                    __builtin_prefetch(w + (n.end >> 5), 0, 1);
                }
            }
            r.call(kc.calls[i], true);
        }
        me.guard_hits = l.guard_hits;
        me.cursor = r.cur;
    };
    // Q8 (parse_anchored_shiftxor.cpp:511-522): the coverage code reads and writes entries at the HEAD of the perfect and
    // substitution lists by loop counter, from anywhere in the record.  Until round 3 a logged write that would change its
    // target sent the whole stage back to one thread (5 s instead of 0.4 for a chromosome) -- "never met" on the test
    // records, met on two of three chromosome-sized ones.  Now the ranges are walked in order after the parallel pass: a
    // range whose logged writes would change anything is done again on its own with the writes made; if that changed what
    // the by-counter reads see (start, end, motif size of an entry) and a later range read such an entry, the ranges behind
    // it run again in parallel against the new heads -- one more pass per change that matters, not a sequential stage.
    const size_t n_heads_p = std::min<size_t>(lists.perfect.size(), 4096), n_heads_s = std::min<size_t>(lists.subst.size(), 4096);
    const SeedVec saved_heads_p(lists.perfect.begin(), lists.perfect.begin() + (long)n_heads_p);
    const SeedVec saved_heads_s(lists.subst.begin(), lists.subst.begin() + (long)n_heads_s);
    auto differs = [](const RibbitSeed &t, const RibbitSeed &v) { return t.start != v.start || t.end != v.end || t.mlen != v.mlen || t.type != v.type; };
    size_t done = 0;
    std::vector<double> range_ms(nr, 0.0);
    // the ranges a parallel pass has to run: all of them in the first; in a later one those that can come out differently
    // after the head change that ended the walk (see below), the others keep what they have
    std::vector<char> stale(nr, 1);
    std::vector<size_t> todo;
    bool fallback = std::getenv("RIBBIT_MERGE_FORCE_REDO") != nullptr;      // (the variable: test hook)
    const bool rerun_all = std::getenv("RIBBIT_MERGE_RERUN_ALL") != nullptr;   // (test hook: every range behind a head change runs again, as until round 3)
    int64_t head_writes = 0;
    unsigned passes = 0;
    // The joined list's storage is made ready beside the passes: a vector constructs (zeroes) what it grows by, on one
    // thread -- 90 ms for a chromosome's half gigabyte when done at the join.  Nothing reads lists.anchored until then, so a
    // helper sizes it for the most the stage can append (one seed per kept call) while the ranges run; the join then only
    // shrinks it and copies on all threads.
    std::thread presize;
    if (!fallback) presize = std::thread([&lists, &kc]() {
        try { lists.anchored.clear(); lists.anchored.resize(kc.n + kc.n_flush + 1); }
        catch (...) { lists.anchored.clear(); }      // out of memory here: the join's own resize reports it on the calling thread
    });
    struct JoinPresize { std::thread &t; ~JoinPresize() { if (t.joinable()) t.join(); } } presize_guard{presize};
    st.before_passes_ms = now_ms() - t1;
    while (done < nr && !fallback) {
        if (++passes > 16) { fallback = true; break; }
        const size_t from = done;
        todo.clear();
        for (size_t k = from; k < nr; ++k) if (stale[k]) { todo.push_back(k); stale[k] = 0; }
        const double tp = now_ms();
        if (device && passes == 1) {
            // The first pass on the device -- and, while its kernel runs, on the host threads too.  A lane takes some 70 us per call, a
            // host thread 0.1: the GPU wins by numbers alone, and the time of its pass is that of its last ranges.  So the ranges are
            // put in the order of the work they are expected to be (passes of a lane's loop, from the calls of the range and how deep
            // they lie on one another: fitted on a 60-Mbp record, within 17 % for nine ranges in ten); the lanes take them from the
            // light end, the host threads from the heavy end -- the dense loci, where a lane needs thousands of passes and its
            // candidate lists may not fit -- until the two meet.  Neither side locks anything: each publishes how far it is (a word
            // in page-locked memory) and reads the other's; a range both took is the host's.  What the device could not merge
            // (candidate lists beyond its LDS, a budget of passes) follows on the host threads.
            std::vector<AnchoredDevicePass::RangeResult> res;
            std::vector<AnchoredDevicePass::LogEntry> undo, reads;
            std::vector<AnchoredDevicePass::HeadEntry> heads;
            std::vector<uint32_t> order(nr);                      // counting sort by expected work, 16 passes to a bucket
            size_t device_limit = 0;
            {
                constexpr uint32_t BUCKETS = 2048;
                std::vector<uint16_t> key(nr);
                parallel_pieces(nr, threads, [&](size_t lo, size_t hi, unsigned) {
                    for (size_t k = lo; k < hi; ++k) {
                        int64_t sum = 0; int32_t left = INT32_MAX, right = -1;
                        for (size_t i = first[k]; i < first[k + 1]; ++i) { const RibbitCall &c = kc.calls[i]; sum += c.end - c.start; left = std::min(left, c.start); right = std::max(right, c.end); }
                        const double calls = (double)(first[k + 1] - first[k]), depth = (double)sum / (double)std::max(1, right - left);
                        const double expected = 11.0 * calls + 0.285 * calls * depth;
                        key[k] = (uint16_t)std::min<double>(BUCKETS - 1, expected / 16.0);
                    }
                });
                std::vector<uint32_t> at(BUCKETS + 1, 0);
                for (size_t k = 0; k < nr; ++k) ++at[key[k] + 1u];
                for (uint32_t q = 1; q <= BUCKETS; ++q) at[q] += at[q - 1];
                for (size_t k = 0; k < nr; ++k) order[at[key[k]]++] = (uint32_t)k;
                // (at[b] is now the END of bucket b) the lanes' part of the list: ranges expected to take no more than the most a lane should
                device_limit = at[std::min<size_t>(BUCKETS - 2, device->max_range_passes / 16)];
            }
            std::vector<char> host_done(nr, 0);
            double meanwhile_ms = 0.0;
            std::atomic<size_t> host_took{0};
            const std::function<void(uint32_t *, const uint32_t *)> meanwhile = [&](uint32_t *from_back, const uint32_t *from_front) {
                const double tm = now_ms();
                // *from_back: how many entries of `order` the lanes may still take -- never more than device_limit, and nothing the
                // host threads have taken (they start at the very back); *from_front: how many the lanes have taken (as far as the
                // host has seen)
                std::atomic<uint32_t> back{(uint32_t)nr};
                run_ranges(threads, threads, [&](size_t) {
                    for (;;) {
                        uint32_t left = back.load(std::memory_order_relaxed);
                        do { if (left == 0) return; }                                          // (everything is taken)
                        while (!back.compare_exchange_weak(left, left - 1u, std::memory_order_relaxed));
                        const uint32_t q = left - 1u;
                        for (uint32_t seen = __atomic_load_n(from_back, __ATOMIC_RELAXED); q < seen;)       // (only ever lowered)
                            if (__atomic_compare_exchange_n(from_back, &seen, q, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                        if (q < __atomic_load_n(from_front, __ATOMIC_RELAXED)) return;          // the lanes are past this one
                        const size_t k = order[q];
                        const double tr = now_ms();
                        state[k].reset(k > 0, first[k + 1] - first[k]);
                        body(k, state[k], false, nullptr, nullptr);
                        range_ms[k] = now_ms() - tr;
                        host_done[k] = 1;
                        host_took.fetch_add(1, std::memory_order_relaxed);
                    }
                });
                meanwhile_ms = now_ms() - tm;
            };
            const double td = now_ms();
            const bool ran = device->run(lists, kc, first, cut_pos, start_cursor, order, device_limit, meanwhile, res, undo, reads, heads) && res.size() == nr;
            st.device_ms = now_ms() - td;
            st.device_meanwhile_ms = meanwhile_ms;
            st.device_host_share = (unsigned)host_took.load();
            std::vector<size_t> host_todo;
            if (ran) {
                auto seed_at = [&](uint32_t list, uint32_t index) -> RibbitSeed * { return (list ? lists.subst.data() : lists.perfect.data()) + index; };
                // the results into the ranges' states, on the host threads: each takes a slice of the ranges and looks through the whole
                // logs for its own (the entries come in the order the lanes happened to write them)
                const size_t slices = std::max<size_t>(1, std::min<size_t>(threads, nr / 1024 + 1));
                run_ranges(slices, threads, [&](size_t t) {
                    const size_t lo = nr * t / slices, hi = nr * (t + 1) / slices;
                    for (size_t k = lo; k < hi; ++k) {
                        if (res[k].status || host_done[k]) continue;
                        RangeState &me = state[k];
                        me.reset(false, 0);
                        me.ext = res[k].own; me.ext_n = res[k].own_n;
                        me.guard_hits = res[k].guard_hits; me.cursor = res[k].cursor;
                        me.head_reads[0] = res[k].head_reads[0]; me.head_reads[1] = res[k].head_reads[1];
                    }
                    // (a range the host merges makes its retirements itself; the others' are made here, as their workers would have)
                    for (const AnchoredDevicePass::LogEntry &e : undo) {
                        if (e.range < lo || e.range >= hi || res[e.range].status || host_done[e.range]) continue;
                        RibbitSeed *sd = seed_at(e.list, e.index);
                        state[e.range].undo.push_back({sd, e.value});
                        sd->type = RIBBIT_RANK_N;
                    }
                    for (const AnchoredDevicePass::LogEntry &e : reads)
                        if (e.range >= lo && e.range < hi && !res[e.range].status && !host_done[e.range]) state[e.range].foreign_reads.push_back({seed_at(e.list, e.index), e.value != 0});
                    for (const AnchoredDevicePass::HeadEntry &e : heads)
                        if (e.range >= lo && e.range < hi && !res[e.range].status && !host_done[e.range]) state[e.range].head_writes.push_back({seed_at(e.list, e.index), e.value, e.changed_then});
                });
                for (size_t k = 0; k < nr; ++k) if (res[k].status && !host_done[k]) host_todo.push_back(k);
                st.device_ranges = (unsigned)(nr - host_todo.size() - host_took.load());
                st.device_bailed = (unsigned)host_todo.size();
                st.device_apply_ms = now_ms() - td - st.device_ms;
            } else {
                for (size_t k = 0; k < nr; ++k) if (!host_done[k]) host_todo.push_back(k);
            }
            for (size_t k2 = 0; k2 < nr; ++k2) if (host_done[k2]) { st.range_ms_sum += range_ms[k2]; st.range_ms_max = std::max(st.range_ms_max, range_ms[k2]); }
            todo.swap(host_todo);
        }
        run_ranges(todo.size(), threads, [&](size_t q) {
            const size_t k = todo[q];
            const double tr = now_ms();
            state[k].reset(k > 0, first[k + 1] - first[k]);
            body(k, state[k], false, nullptr, nullptr);
            range_ms[k] = now_ms() - tr;
        });
        const double tw = now_ms();
        st.pass_ms += tw - tp;
        for (size_t k2 : todo) { st.range_ms_sum += range_ms[k2]; st.range_ms_max = std::max(st.range_ms_max, range_ms[k2]); }
        st.ranges_run += (unsigned)todo.size();
        bool again = false;
        size_t k = from;
        for (; k < nr; ++k) {
            bool redo = k > 0 && !state[k].reads_still_valid();
            int64_t changing = 0;
            for (const ListRefs::HeadWrite &w : state[k].head_writes) changing += w.changed_then || differs(*w.target, w.value);
            head_writes += changing;
            uint64_t ch[2] = {0, 0};
            int reach = -1;
            if (redo || changing) {
                state[k].reset(k > 0, first[k + 1] - first[k]);
                body(k, state[k], true, ch, &reach);
                ++st.ranges_redone;
            }
            if (k == 0 && state[0].own_empty()) { fallback = true; break; }      // later ranges assumed a non-empty list
            if (ch[0] | ch[1]) {
                // A later range comes out differently only if it read a changed entry by loop counter, or if the entry -- where
                // it was or where the write put it -- can be met by the range's ordinary walks: those stay right of the range's
                // left cut, so an entry that ends more than a motif before the cut is out of their sight, like every other seed
                // of the ranges before.  Bit 63 stands for every entry from the 64th on: then all ranges behind run again.
                // (Round 3 marked ranges only when some range had READ a changed entry by counter: a range with the entry
                // in sight of its walks but no such read kept a result made against the old entry -- a range does not log
                // reads inside its own territory as foreign, so nothing else caught it.  Each range is judged on its own now.)
                const bool wide = rerun_all || ((ch[0] | ch[1]) >> 63) != 0;
                bool any = false;
                for (size_t j = k + 1; j < nr; ++j) {
                    const bool reads = ((state[j].head_reads[0] & ch[0]) | (state[j].head_reads[1] & ch[1])) != 0;
                    const bool in_sight = (int64_t)cut_pos[j] <= (int64_t)reach + lists.max_motif + 2;
                    if (wide || reads || in_sight) { stale[j] = 1; any = true; }
                    if (in_sight && !reads && !wide) ++st.stale_by_sight;
                }
                if (any) { done = k + 1; again = true; break; }
            }
        }
        st.walk_ms += now_ms() - tw;
        if (fallback) break;
        if (!again) done = nr;
    }
    st.head_writes = head_writes;
    st.passes = passes;
    st.first_range_empty = state[0].own_empty();
    if (presize.joinable()) presize.join();
    if (fallback) {
        for (size_t k = nr; k-- > 0;) state[k].reset(false, 0);      // takes back the ranges' retirements
        std::copy(saved_heads_p.begin(), saved_heads_p.end(), lists.perfect.begin());
        std::copy(saved_heads_s.begin(), saved_heads_s.end(), lists.subst.begin());
        restore_types(lists.perfect, perfect_types);
        restore_types(lists.subst, subst_types);
        lists.guard_hits = guard_before;
        anchored_in_order(lists, kc);
        st.redone_in_order = true;
    } else {
        const double tc = now_ms();
        join_ranges(state, nr, lists.anchored, threads);
        for (size_t k = 0; k < nr; ++k) lists.guard_hits += state[k].guard_hits;
        st.concat_ms = now_ms() - tc;
        const double tf = now_ms();
        AnchoredReplay<SeedLists> r{lists, state[nr - 1].cursor};
        r.pending_end = kc.tail_pend;
        r.run(kc.flush, kc.n_flush, lists.length);
        st.flush_ms = now_ms() - tf;
    }
    st.merge_ms = now_ms() - t1;
    tl_last_stats[1] = st;
    if (stats) *stats = st;
}

namespace { thread_local unsigned tl_last_dispatch_ranges = 0; }
unsigned last_dispatch_ranges() { return tl_last_dispatch_ranges; }

unsigned dispatch_order_ranges(const SeedLists &sl, const std::vector<int> &all_cuts, unsigned threads, SeedVec &out) {
    threads = resolve_threads(threads);
    // (the device pass of the anchored merge cuts a chromosome into 10^5 ranges; sixteen per thread are plenty here, and any
    // subset of valid cuts is a valid set of cuts)
    std::vector<int> thinned;
    const size_t most = (size_t)threads * 16;
    if (all_cuts.size() > 2 * most) {
        const size_t step = (all_cuts.size() + most - 1) / most;
        for (size_t k = 0; k < all_cuts.size(); k += step) thinned.push_back(all_cuts[k]);
    }
    const std::vector<int> &cut_pos = thinned.empty() ? all_cuts : thinned;
    const size_t nr = cut_pos.size();
    tl_last_dispatch_ranges = 1;
    if (nr < 2 || threads < 2) { dispatch_order(sl, out); return 1; }
    const SeedVec *lists[3] = {&sl.perfect, &sl.subst, &sl.anchored};
    // split[x][k] = first entry of list x that belongs to range k (bisection on start > cut: valid iff the list splits
    // cleanly, which the workers check)
    std::vector<size_t> split[3];
    for (int x = 0; x < 3; ++x) {
        const SeedVec &l = *lists[x];
        split[x].assign(nr + 1, l.size());
        split[x][0] = 0;
        for (size_t k = 1; k < nr; ++k) {
            size_t lo = split[x][k - 1], hi = l.size();
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (l[mid].start > cut_pos[k]) hi = mid; else lo = mid + 1; }
            split[x][k] = lo;
        }
    }
    // the ranges' outputs, reserved for the most a range can yield (no regrowth); the slices of the three lists are read in place
    std::vector<SeedVec> part(nr);
    std::atomic<bool> out_of_place{false};
    std::atomic<size_t> next{0};
    auto loop = [&]() {
        for (size_t k; (k = next.fetch_add(1)) < nr && !out_of_place;) {
            const int64_t lo = k > 0 ? (int64_t)cut_pos[k] : -1, hi = k + 1 < nr ? (int64_t)cut_pos[k + 1] : INT64_MAX;
            for (int x = 0; x < 3 && !out_of_place; ++x) {
                const SeedVec &l = *lists[x];
                const size_t a = split[x][k], b = split[x][k + 1];
                for (size_t i = a; i < b; ++i)
                    if (!((int64_t)l[i].start > lo && (int64_t)l[i].start < hi)) { out_of_place = true; break; }
            }
            if (out_of_place) return;
            part[k].reserve((split[0][k + 1] - split[0][k]) + (split[1][k + 1] - split[1][k]) + (split[2][k + 1] - split[2][k]));
            dispatch_order_slices(sl.perfect.data() + split[0][k], split[0][k + 1] - split[0][k], sl.subst.data() + split[1][k], split[1][k + 1] - split[1][k],
                                  sl.anchored.data() + split[2][k], split[2][k + 1] - split[2][k], part[k]);
        }
    };
    {
        std::vector<std::thread> pool;
        const unsigned nt = (unsigned)std::min<size_t>(threads, nr);
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(loop);
        loop();
        for (std::thread &t : pool) t.join();
    }
    if (out_of_place) { dispatch_order(sl, out); return 1; }
    std::vector<size_t> at(nr + 1, 0);
    for (size_t k = 0; k < nr; ++k) at[k + 1] = at[k] + part[k].size();
    out.resize(at[nr]);
    next = 0;
    auto copy = [&]() {
        for (size_t k; (k = next.fetch_add(1)) < nr;)
            if (!part[k].empty()) std::memcpy(out.data() + at[k], part[k].data(), part[k].size() * sizeof(RibbitSeed));
    };
    {
        std::vector<std::thread> pool;
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, at[nr] / 65536 + 1));
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(copy);
        copy();
        for (std::thread &t : pool) t.join();
    }
    tl_last_dispatch_ranges = (unsigned)nr;
    return (unsigned)nr;
}

}  // namespace rb
