#include "event_stream.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "window_fsm.h"

namespace rb {

namespace {
inline bool call_order(const RibbitCall &a, const RibbitCall &b) { return a.pos != b.pos ? a.pos < b.pos : a.mlen < b.mlen; }
// c1 / c2 of parse_perfect_shiftxor.cpp:193 / :179
inline int cutoff_zero(int m) { return (m <= 6) ? 12 - m : m; }
inline int cutoff_n(int m, int min_shift) { return (m <= 6) ? 12 - m : m + (m - min_shift); }
}  // namespace

bool pair_perfect_runs(const EventSource &src, std::vector<RibbitRun> &runs, std::string *why) {
    runs.clear();
    for (size_t mi = 0; mi < src.nm; ++mi) {
        const int32_t mlen = src.m_lo + (int32_t)mi;
        int64_t open = -1;
        for (MotifCursor c(src, mi); !c.done(); c.next()) {
            const uint64_t e = c.peek();
            const int64_t pos = ev_pos(e);
            const uint32_t kind = ev_kind(e);
            if (kind == EV_START) {
                if (open != -1) { if (why) *why = "two run starts in a row for motif " + std::to_string(mlen); return false; }
                open = pos;
            } else {
                if (open == -1 || pos <= open) { if (why) *why = "run end without start for motif " + std::to_string(mlen); return false; }
                const int term = kind == EV_END_ZERO ? RIBBIT_TERM_ZERO : kind == EV_END_N ? RIBBIT_TERM_N : RIBBIT_TERM_EOS;
                runs.push_back(RibbitRun{(int32_t)open, (int32_t)pos, mlen, term});
                open = -1;
            }
        }
        if (open != -1) { if (why) *why = "unterminated run for motif " + std::to_string(mlen); return false; }
    }
    return true;
}

bool pair_perfect_runs_partial(const EventSource &src, int64_t own_lo, int64_t own_hi, int64_t pos_offset,
                               std::vector<RibbitRun> &runs, std::vector<uint64_t> &halves, std::string *why) {
    runs.clear();
    halves.clear();
    for (size_t mi = 0; mi < src.nm; ++mi) {
        const int32_t mlen = src.m_lo + (int32_t)mi;
        int64_t open = -1;
        bool first = true;
        for (MotifCursor c(src, mi); !c.done(); c.next()) {
            const uint64_t e = c.peek();
            const int64_t local = ev_pos(e);
            if (local < own_lo || local >= own_hi) continue;
            const int64_t pos = local + pos_offset;
            const uint32_t kind = ev_kind(e);
            if (kind == EV_START) {
                if (open != -1) { if (why) *why = "two run starts in a row for motif " + std::to_string(mlen); return false; }
                open = pos;
            } else if (open == -1) {
                if (!first) { if (why) *why = "run end without start for motif " + std::to_string(mlen); return false; }
                halves.push_back(ev_pack((uint32_t)pos, (uint32_t)mlen, kind));      // run began in an earlier chunk
            } else {
                const int term = kind == EV_END_ZERO ? RIBBIT_TERM_ZERO : kind == EV_END_N ? RIBBIT_TERM_N : RIBBIT_TERM_EOS;
                runs.push_back(RibbitRun{(int32_t)open, (int32_t)pos, mlen, term});
                open = -1;
            }
            first = false;
        }
        if (open != -1) halves.push_back(ev_pack((uint32_t)open, (uint32_t)mlen, EV_START));   // run ends in a later chunk
    }
    return true;
}

void perfect_calls_from_runs(const RibbitRun *runs, size_t n_runs, int64_t length, int min_shift, CallVec &calls) {
    calls.clear();
    const int32_t L = (int32_t)length;
    for (size_t i = 0; i < n_runs; ++i) {
        const RibbitRun &r = runs[i];
        const int len = r.end - r.start;
        if (r.term == RIBBIT_TERM_ZERO) {                       // parse_perfect_shiftxor.cpp:199-205
            if (len >= cutoff_zero(r.mlen)) calls.push_back(RibbitCall{r.end, r.mlen, r.start, r.end});
        } else if (r.term == RIBBIT_TERM_N) {                   // :175-186
            if (len >= cutoff_n(r.mlen, min_shift)) calls.push_back(RibbitCall{r.end, r.mlen, r.start, r.end});
        } else {                                                // :213-223, flushed with end = L-1
            if ((L - 1) - r.start >= cutoff_zero(r.mlen)) calls.push_back(RibbitCall{L, r.mlen, r.start, L - 1});
        }
    }
    // reference call order: scan position major, motif minor; the end-of-sequence flush (pos == L) last
    std::stable_sort(calls.begin(), calls.end(), call_order);
}

bool replay_window_events(const EventSource &src, const HostPlanes &hp, CallVec &calls, std::string *why, unsigned host_threads) {
    calls.clear();
    const size_t nm = src.nm;
    const int32_t m_lo = src.m_lo;
    constexpr int64_t TILE = 16384;          // ordering granule (any value works)
    const int64_t ntile = hp.length / TILE + 1;

    unsigned threads = host_threads ? host_threads : std::min(std::thread::hardware_concurrency(), 16u);
    if (!host_threads)
        if (const char *env = std::getenv("RIBBIT_THREADS")) threads = (unsigned)std::max(1, std::atoi(env));
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)threads, nm, (size_t)256}));

    // Phase 1 (parallel over motifs): every worker replays the state machines of its motifs over the whole
    // record, tile by tile, settling pending groups as soon as their reporting window is known to precede the
    // next tile, so that a call is generated while its tile (or the one before) is current.  Calls go to a
    // per-worker vector; cut[t] marks where tile t's calls start.  A worker's calls inside a tile are motif-major.
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // worker buffers live as long as the calling thread: a 250-Mbp chromosome's half gigabyte of calls is
    // page-faulted in once, not once per record
    struct Worker { CallVec calls; std::vector<size_t> cut; CallVec flush; std::string err; };
    static thread_local std::vector<Worker> tl_work;
    std::vector<Worker> &work = tl_work;      // the worker threads must see THIS thread's buffers, not their own
    if (work.size() < threads) work.resize(threads);
    for (unsigned w = 0; w < threads; ++w) { work[w].calls.clear(); work[w].flush.clear(); work[w].err.clear(); }
    auto phase1 = [&](unsigned w) {
        Worker &me = work[w];
        std::vector<size_t> mine;
        for (size_t mi = w; mi < nm; mi += threads) mine.push_back(mi);
        std::vector<WindowFsm> fsm;
        std::vector<MotifCursor> cur;
        for (size_t mi : mine) { fsm.emplace_back(hp, m_lo + (int32_t)mi); cur.emplace_back(src, mi); }
        me.cut.assign((size_t)ntile + 1, 0);
        for (int64_t t = 0; t < ntile; ++t) {
            me.cut[(size_t)t] = me.calls.size();
            const int64_t next_tile = (t + 1) * TILE;
            for (size_t k = 0; k < mine.size(); ++k) {
                WindowFsm &f = fsm[k];
                MotifCursor &c = cur[k];
                f.set_output(&me.calls);
                for (; !c.done(); c.next()) {
                    const uint64_t e = c.peek();
                    if ((int64_t)ev_pos(e) >= next_tile) break;
                    if (!f.event((int64_t)ev_pos(e), ev_kind(e))) {
                        me.err = "window START/END events of motif " + std::to_string(m_lo + (int)mine[k]) + " do not alternate";
                        return;
                    }
                }
                f.settle_up_to(next_tile);
            }
        }
        me.cut[(size_t)ntile] = me.calls.size();
        for (size_t k = 0; k < mine.size(); ++k) {
            if (!cur[k].done()) { me.err = "event beyond the end of the record"; return; }
            fsm[k].set_output(&me.flush);
            if (!fsm[k].finish()) { me.err = "window event stream ends inside a streak"; return; }
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < threads; ++w) pool.emplace_back(phase1, w);
        phase1(0);
        for (std::thread &t : pool) t.join();
    }
    for (unsigned w = 0; w < threads; ++w)
        if (!work[w].err.empty()) { if (why) *why = work[w].err; return false; }

    const double t_phase1 = now();
    // Phase 2: a call generated while tile t was current has pos in [t*TILE, (t+1)*TILE + 7), i.e. it belongs to
    // ordering bucket t or t+1.  Count per bucket, prefix-sum, then fill and order every bucket independently
    // (parallel over buckets): (pos, motif) order by two stable counting sorts.
    const size_t nb = (size_t)ntile + 1;
    std::vector<size_t> bucket_n(nb + 1, 0);          // all calls per bucket
    auto bucket_of = [&](const RibbitCall &c) { return (size_t)(c.pos / TILE); };
    {
        // every worker counts its own calls (hundreds of millions of them on a chromosome), then the counts are added up
        std::vector<std::vector<uint32_t>> all_w(threads);
        std::atomic<bool> beyond{false};
        auto count_worker = [&](unsigned wi) {
            all_w[wi].assign(nb, 0);
            for (const RibbitCall &c : work[wi].calls) {
                const size_t bk = bucket_of(c);
                if (bk >= nb) { beyond = true; return; }
                ++all_w[wi][bk];
            }
        };
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < threads; ++w) pool.emplace_back(count_worker, w);
        count_worker(0);
        for (std::thread &t : pool) t.join();
        if (beyond) { if (why) *why = "call beyond the end of the record"; return false; }
        for (unsigned wi = 0; wi < threads; ++wi)
            for (size_t k = 0; k < nb; ++k) bucket_n[k + 1] += all_w[wi][k];
    }
    for (size_t k = 0; k < nb; ++k) bucket_n[k + 1] += bucket_n[k];
    size_t n_flush = 0;
    for (unsigned wi = 0; wi < threads; ++wi) n_flush += work[wi].flush.size();
    calls.reserve(bucket_n[nb] + n_flush);      // the flush is appended below: no second half-gigabyte move
    calls.resize(bucket_n[nb]);
    const double t_count = now();
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    auto phase2 = [&]() {
        CallVec batch, by_motif;
        std::vector<uint32_t> count;
        for (size_t bk; (bk = next.fetch_add(1)) < nb;) {
            batch.clear();
            // bucket bk receives calls generated during tiles bk-1 and bk
            for (unsigned wi = 0; wi < threads; ++wi) {
                const Worker &w = work[wi];
                for (int64_t t = (int64_t)bk - 1; t <= (int64_t)bk; ++t) {
                    if (t < 0 || t >= ntile) continue;
                    for (size_t i = w.cut[(size_t)t]; i < w.cut[(size_t)t + 1]; ++i)
                        if (bucket_of(w.calls[i]) == bk) batch.push_back(w.calls[i]);
                }
            }
            if (batch.size() != bucket_n[bk + 1] - bucket_n[bk]) { bad = true; return; }
            if (batch.empty()) continue;
            by_motif.resize(batch.size());
            count.assign(nm + 1, 0);
            for (const RibbitCall &c : batch) ++count[(size_t)(c.mlen - m_lo) + 1];
            for (size_t k = 0; k < nm; ++k) count[k + 1] += count[k];
            for (const RibbitCall &c : batch) by_motif[count[(size_t)(c.mlen - m_lo)]++] = c;
            const int64_t base_pos = (int64_t)bk * TILE;
            count.assign((size_t)TILE + 1, 0);
            for (const RibbitCall &c : by_motif) ++count[(size_t)(c.pos - base_pos) + 1];
            for (int64_t k = 0; k < TILE; ++k) count[(size_t)k + 1] += count[(size_t)k];
            RibbitCall *dst = calls.data() + bucket_n[bk];
            for (const RibbitCall &c : by_motif) dst[count[(size_t)(c.pos - base_pos)]++] = c;
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned w = 1; w < threads; ++w) pool.emplace_back(phase2);
        phase2();
        for (std::thread &t : pool) t.join();
    }
    if (bad) { if (why) *why = "call generated outside its ordering buckets"; return false; }
    if (profile)
        std::fprintf(stderr, "[window replay] %u threads: state machines %.1f ms, bucket count %.1f ms, ordering %.1f ms, %zu calls%s\n",
                     threads, t_phase1 - t_begin, t_count - t_phase1, now() - t_count, bucket_n[nb], "");

    // the end-of-sequence flush comes last, in motif order
    CallVec flush;
    for (unsigned wi = 0; wi < threads; ++wi) flush.insert(flush.end(), work[wi].flush.begin(), work[wi].flush.end());
    std::stable_sort(flush.begin(), flush.end(), call_order);
    calls.insert(calls.end(), flush.begin(), flush.end());
    return true;
}

}  // namespace rb
