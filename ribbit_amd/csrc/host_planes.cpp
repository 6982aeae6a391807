#include "host_planes.h"

#include <algorithm>
#include <cstdlib>
#include <thread>

namespace rb {

std::shared_ptr<const std::vector<uint8_t>> HostPlanes::symbols(unsigned threads) const {
    std::lock_guard<std::mutex> lk(sym_mu_);
    // (80 entries of padding, all 4 = N: consensus_row compares up to 64 symbols at a time and may read that far past the record)
    if (sym_cache_ && sym_cache_->size() == (size_t)length + 80) return sym_cache_;
    auto sym = std::make_shared<std::vector<uint8_t>>((size_t)length + 80, (uint8_t)4);
    uint8_t *out = sym->data();
    const int64_t nw = (length + 31) / 32;
    auto decode = [&](int64_t w0, int64_t w1) {
        for (int64_t w = w0; w < w1; ++w) {
            const uint32_t h = hi[(size_t)w], l = lo[(size_t)w], b = brk[(size_t)w];
            const int64_t base = w * 32;
            const int n = (int)std::min<int64_t>(32, length - base);
            for (int k = 0; k < n; ++k)
                out[base + k] = ((b >> k) & 1u) ? (uint8_t)4 : (uint8_t)((((h >> k) & 1u) << 1) | ((l >> k) & 1u));
        }
    };
    if (!threads) {
        threads = std::min(std::thread::hardware_concurrency(), 16u);
        if (const char *env = std::getenv("RIBBIT_THREADS")) threads = (unsigned)std::max(1, std::atoi(env));
    }
    threads = (unsigned)std::max<int64_t>(1, std::min<int64_t>(threads, nw / 65536 + 1));
    if (threads == 1) decode(0, nw);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(decode, nw * t / threads, nw * (t + 1) / threads);
        for (std::thread &th : pool) th.join();
    }
    sym_cache_ = sym;
    return sym_cache_;
}

void HostPlanes::index_breaks() {
    blocked.clear();
    // N runs [n0, n1) among positions < L, from the brk plane
    int64_t run_start = -1;
    auto close_run = [&](int64_t n1) {
        const int64_t first = std::max<int64_t>(0, run_start - 7), last = n1 - 1;
        if (!blocked.empty() && first <= blocked.back().second + 1) blocked.back().second = last;
        else blocked.emplace_back(first, last);
        run_start = -1;
    };
    const int64_t nw = (length + 31) / 32;
    for (int64_t w = 0; w < nw; ++w) {
        uint32_t x = brk[w];
        const int64_t base = w * 32;
        if (base + 32 > length) x &= (length - base >= 32) ? 0xffffffffu : ((1u << (unsigned)(length - base)) - 1u);
        if (x == 0) { if (run_start != -1) close_run(base); continue; }
        if (x == 0xffffffffu) { if (run_start == -1) run_start = base; continue; }
        for (int b = 0; b < 32; ++b) {
            const bool n = (x >> b) & 1u;
            if (n && run_start == -1) run_start = base + b;
            else if (!n && run_start != -1) close_run(base + b);
        }
    }
    if (run_start != -1) close_run(length);
}

int HostPlanes::range_count(int shift, int start, int end) const {
    if (end <= start) return 0;
    int total = 0;
    const int64_t w0 = start >> 5, w1 = (end - 1) >> 5;
    for (int64_t w = w0; w <= w1; ++w) {
        uint32_t x = x_word(shift, w);
        if (w == w0) x &= 0xffffffffu << (start & 31);
        if (w == w1) { const int hi_bits = ((end - 1) & 31) + 1; if (hi_bits < 32) x &= (1u << hi_bits) - 1u; }
        total += __builtin_popcount(x);
    }
    return total;
}

void HostPlanes::anchor_slice(int shift, int start, int end, std::vector<uint32_t> &out) const {
    if (shift < 1 || end <= start) return;
    const int64_t L = length;
    // a run that overlaps [start, end) and is shorter than 2*shift lies inside this window; one that touches the
    // window's edge away from the record's ends is at least 2*shift + 1 long as seen from inside, i.e. no anchor
    const int64_t lo = std::max<int64_t>(0, (int64_t)start - 2 * shift), hi = std::min<int64_t>(L, (int64_t)end + 2 * shift);
    if (hi <= lo) return;
    const int64_t forced_from = L - shift;           // the reference stops at p = L-1-shift: a run still open there is dropped
    const int64_t base = (int64_t)(start >> 5) << 5;
    auto emit = [&](int64_t rs, int64_t re) {        // maximal run [rs, re) of ones, closed by a zero at re
        const int64_t len = re - rs;
        if (re >= L || len < 3 || len >= 2 * (int64_t)shift) return;     // re == L: cut by the record's end, i.e. still open
        const int64_t a = std::max<int64_t>(rs, start), b = std::min<int64_t>(re, end);
        for (int64_t p = a; p < b;) {
            const int64_t w = (p - base) >> 5, bit = (p - base) & 31;
            const int64_t n = std::min<int64_t>(32 - bit, b - p);
            out[(size_t)w] |= (n == 32 ? 0xffffffffu : ((1u << n) - 1u)) << bit;
            p += n;
        }
    };
    int64_t run_start = -1;
    for (int64_t w = lo >> 5; w <= (hi - 1) >> 5; ++w) {
        const int64_t p0 = w << 5;
        uint32_t x = x_word(shift, w);
        if (p0 + 32 > forced_from) x |= forced_from <= p0 ? 0xffffffffu : (0xffffffffu << (unsigned)(forced_from - p0));
        // positions outside [lo, hi) read as zeros (they close runs; see above why that is harmless) -- except the
        // record's end: a run reaching L is open, never an anchor
        if (p0 < lo) x &= 0xffffffffu << (unsigned)(lo - p0);
        if (p0 + 32 > hi) x &= (hi - p0 >= 32) ? 0xffffffffu : ((1u << (unsigned)(hi - p0)) - 1u);
        for (int64_t bit = 0; bit < 32;) {
            // next one (a run begins) or next zero (the run ends) at or after `bit`; the shift feeds zeros in from the top
            const uint32_t rest = (run_start == -1 ? x : ~x) >> bit;
            if (rest == 0) break;
            bit += __builtin_ctz(rest);
            if (run_start == -1) run_start = p0 + bit;
            else { emit(run_start, p0 + bit); run_start = -1; }
        }
    }
    if (run_start != -1 && hi < L) emit(run_start, hi);      // cut by the window: long (or irrelevant) by construction
}

void HostPlanes::xa_slice(int mlen, int start, int end, std::vector<uint32_t> &out) const {
    out.clear();
    if (end <= start) return;
    const int64_t w0 = start >> 5, w1 = (end - 1) >> 5;
    out.assign((size_t)(w1 - w0 + 1), 0u);
    if (xa_stored() && has_xa(mlen)) {
        const uint32_t *w = xa_words() + (int64_t)(mlen - xa_m_lo) * xa_stride;
        for (int64_t i = w0; i <= w1; ++i) out[(size_t)(i - w0)] = w[i];
    } else {
        for (int64_t i = w0; i <= w1; ++i) out[(size_t)(i - w0)] = x_word(mlen, i);
        if (has_xa(mlen))
            for (int s = mlen - 2; s <= mlen + 2; ++s)
                if (s != mlen) anchor_slice(s, start, end, out);
    }
    out.front() &= 0xffffffffu << (start & 31);
    const int hi_bits = ((end - 1) & 31) + 1;
    if (hi_bits < 32) out.back() &= (1u << hi_bits) - 1u;
}

int HostPlanes::range_count_xa(int mlen, int start, int end) const {
    if (end <= start) return 0;
    int total = 0;
    if (xa_stored()) {
        const uint32_t *w = xa_words() + (int64_t)(mlen - xa_m_lo) * xa_stride;
        const int64_t w0 = start >> 5, w1 = (end - 1) >> 5;
        for (int64_t i = w0; i <= w1; ++i) {
            uint32_t x = w[i];
            if (i == w0) x &= 0xffffffffu << (start & 31);
            if (i == w1) { const int hi_bits = ((end - 1) & 31) + 1; if (hi_bits < 32) x &= (1u << hi_bits) - 1u; }
            total += __builtin_popcount(x);
        }
        return total;
    }
    static thread_local std::vector<uint32_t> slice;
    xa_slice(mlen, start, end, slice);
    for (uint32_t x : slice) total += __builtin_popcount(x);
    return total;
}

int64_t HostPlanes::first_evaluated(int64_t from) const {
    int64_t q = std::max<int64_t>(from, 0);
    // last blocked interval starting at or before q
    auto it = std::upper_bound(blocked.begin(), blocked.end(), q,
                               [](int64_t v, const std::pair<int64_t, int64_t> &iv) { return v < iv.first; });
    if (it != blocked.begin()) {
        --it;
        if (q <= it->second) q = it->second + 1;   // intervals are merged, so q is now clear of every N
    }
    return (q + 7 <= length - 1) ? q : -1;
}

}  // namespace rb
