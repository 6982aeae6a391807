#include "event_stream.h"

#include <algorithm>

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

void perfect_calls_from_runs(const std::vector<RibbitRun> &runs, int64_t length, int min_shift, std::vector<RibbitCall> &calls) {
    calls.clear();
    const int32_t L = (int32_t)length;
    for (const RibbitRun &r : runs) {
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

bool replay_window_events(const EventSource &src, const HostPlanes &hp, std::vector<RibbitCall> &calls, std::string *why) {
    calls.clear();
    const size_t nm = src.nm;
    const int32_t m_lo = src.m_lo;
    constexpr int64_t TILE = 16384;          // batching granule of the replay (any value works)
    std::vector<WindowFsm> fsm;
    std::vector<MotifCursor> cur;
    fsm.reserve(nm); cur.reserve(nm);
    for (size_t mi = 0; mi < nm; ++mi) { fsm.emplace_back(hp, m_lo + (int32_t)mi); cur.emplace_back(src, mi); }
    std::vector<RibbitCall> batch, by_motif, carry;
    std::vector<uint32_t> count;
    int64_t flushed_below = 0;               // every call with pos < flushed_below has been written to `calls`
    const int64_t ntile = hp.length / TILE + 1;
    for (int64_t t = 0; t < ntile; ++t) {
        const int64_t next_tile = (t + 1) * TILE;
        batch.swap(carry);
        carry.clear();
        for (size_t mi = 0; mi < nm; ++mi) {
            WindowFsm &f = fsm[mi];
            MotifCursor &c = cur[mi];
            f.set_output(&batch);
            for (; !c.done(); c.next()) {
                const uint64_t e = c.peek();
                if ((int64_t)ev_pos(e) >= next_tile) break;
                if (!f.event((int64_t)ev_pos(e), ev_kind(e))) {
                    if (why) *why = "window START/END events of motif " + std::to_string(m_lo + (int)mi) + " do not alternate";
                    return false;
                }
            }
            f.settle_up_to(next_tile);
        }
        // calls generated from here on have pos >= next_tile + 7, so everything below next_tile is final
        if (batch.empty()) { flushed_below = next_tile; continue; }
        by_motif.resize(batch.size());
        count.assign(nm + 1, 0);
        for (const RibbitCall &c : batch) ++count[(size_t)(c.mlen - m_lo) + 1];
        for (size_t k = 0; k < nm; ++k) count[k + 1] += count[k];
        for (const RibbitCall &c : batch) by_motif[count[(size_t)(c.mlen - m_lo)]++] = c;
        const int64_t span = next_tile - flushed_below;
        count.assign((size_t)span + 2, 0);
        for (const RibbitCall &c : by_motif) {
            if (c.pos < flushed_below) { if (why) *why = "call generated out of order"; return false; }
            if (c.pos < next_tile) ++count[(size_t)(c.pos - flushed_below) + 1];
        }
        for (int64_t k = 0; k < span; ++k) count[(size_t)k + 1] += count[(size_t)k];
        const size_t base = calls.size();
        calls.resize(base + count[(size_t)span]);
        for (const RibbitCall &c : by_motif) {
            if (c.pos < next_tile) calls[base + count[(size_t)(c.pos - flushed_below)]++] = c;
            else carry.push_back(c);
        }
        flushed_below = next_tile;
    }
    for (size_t mi = 0; mi < nm; ++mi)
        if (!cur[mi].done()) { if (why) *why = "event beyond the end of the record"; return false; }
    // leftovers beyond the last tile boundary, then the end-of-sequence flush in motif order
    std::stable_sort(carry.begin(), carry.end(), call_order);
    calls.insert(calls.end(), carry.begin(), carry.end());
    batch.clear();
    for (size_t mi = 0; mi < nm; ++mi) {
        fsm[mi].set_output(&batch);
        if (!fsm[mi].finish()) { if (why) *why = "window event stream ends inside a streak"; return false; }
    }
    std::stable_sort(batch.begin(), batch.end(), call_order);
    calls.insert(calls.end(), batch.begin(), batch.end());
    return true;
}

}  // namespace rb
