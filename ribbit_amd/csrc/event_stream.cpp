#include "event_stream.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>


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

}  // namespace rb
