#include "host_planes.h"

#include <algorithm>

namespace rb {

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

int HostPlanes::range_count_xa(int mlen, int start, int end) const {
    if (end <= start) return 0;
    const uint32_t *w = xa_words() + (int64_t)(mlen - xa_m_lo) * xa_stride;
    int total = 0;
    const int64_t w0 = start >> 5, w1 = (end - 1) >> 5;
    for (int64_t i = w0; i <= w1; ++i) {
        uint32_t x = w[i];
        if (i == w0) x &= 0xffffffffu << (start & 31);
        if (i == w1) { const int hi_bits = ((end - 1) & 31) + 1; if (hi_bits < 32) x &= (1u << hi_bits) - 1u; }
        total += __builtin_popcount(x);
    }
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
