// ssw_exact.cpp -- own implementation of the striped Smith-Waterman alignment ribbit's refinement
// relies on (reference: the vendored Complete-Striped-Smith-Waterman-Library v1.2.5, ssw.c /
// ssw_cpp.cpp, used through Aligner::Align at parse_seed.cpp:404 and parse_smallmotif_seed.cpp:270).
//
// BED rows carry the CIGAR, so the result has to be identical to the library's down to its
// tie-breaks, and those depend on the striped evaluation order: the lazy-F loop
// stops early and never refreshes E, so scores are a function of the 16-/8-lane striping.  This
// file therefore evaluates the recurrences in exactly the library's striped order -- vector j,
// lane l  <->  query position j + l*segLen -- and restates the library's banded path search with its
// tie-breaks, band doubling and boundary handling (see banded_path).  The arithmetic is the library's by
// necessity; the code is this repository's (host SSE2 passes, own cell encoding and walker for the path).
// The library itself (MIT licence, Mengyao Zhao & Wan-Ping Lee) is not linked: oracle/_ref builds it from the
// reference tree only to pin this file in the tests.
// Pinned in tests/test_ssw.py against the reference library itself (oracle/_ref/libssw_ref.so).
#include "ssw_exact.h"

#include <emmintrin.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rb {

namespace {

constexpr int NSYM = 5;                    // A C G T other
constexpr int MATCH = 2, MISMATCH = 2;     // Aligner(): match 2, mismatch 2, gap open 3, gap extend 1 (ssw_cpp.cpp:230-242)
constexpr int GAP_OPEN = 3, GAP_EXTEND = 1;
constexpr int DISTANCE_FILTER = 32767;     // Filter(): score_filter 0, distance_filter 32767 (ssw_cpp.h:58-63)

inline int8_t score_of(int a, int b) { return (a == b && a < 4) ? MATCH : -MISMATCH; }

inline int8_t translate(char c) {          // kBaseTranslation, ssw_cpp.cpp:12-27
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

struct Ends { int score, ref, read, score2, ref2; };


// The two striped passes below are the inner loops of refinement (two of them per alignment over the
// whole pseudo-perfect repeat), so they are written with SSE2 intrinsics: one __m128i = one stripe of 16
// (8-bit pass) or 8 (16-bit pass) query positions, exactly the lane order described at the top.

inline int hmax_u8(__m128i v) {
    v = _mm_max_epu8(v, _mm_srli_si128(v, 8));
    v = _mm_max_epu8(v, _mm_srli_si128(v, 4));
    v = _mm_max_epu8(v, _mm_srli_si128(v, 2));
    v = _mm_max_epu8(v, _mm_srli_si128(v, 1));
    return _mm_extract_epi16(v, 0) & 0xff;
}
inline int hmax_i16(__m128i v) {
    v = _mm_max_epi16(v, _mm_srli_si128(v, 8));
    v = _mm_max_epi16(v, _mm_srli_si128(v, 4));
    v = _mm_max_epi16(v, _mm_srli_si128(v, 2));
    return _mm_extract_epi16(v, 0) & 0xffff;
}

// 8-bit striped pass (ssw.c:197-386).  dir 0: ref left to right, 1: right to left.
// Returns score 255 when the byte range overflowed (the caller then switches to 16 bits).
Ends striped_pass_u8(const int8_t *ref, int dir, int ref_len, const int8_t *read, int read_len, int terminate, int mask_len) {
    constexpr int W = 16;
    const uint8_t bias = MISMATCH;
    const int seg = (read_len + W - 1) / W;
    std::vector<__m128i> profile((size_t)NSYM * seg);
    {
        uint8_t *t = reinterpret_cast<uint8_t *>(profile.data());
        for (int nt = 0; nt < NSYM; ++nt)
            for (int j = 0; j < seg; ++j)
                for (int l = 0; l < W; ++l) {
                    const int q = j + l * seg;
                    *t++ = (uint8_t)(q >= read_len ? bias : score_of(nt, read[q]) + bias);
                }
    }
    const __m128i zero = _mm_setzero_si128();
    std::vector<__m128i> buf_a((size_t)seg, zero), buf_b((size_t)seg, zero), E((size_t)seg, zero), h_best((size_t)seg, zero);
    std::vector<uint8_t> col_max((size_t)ref_len, 0);
    __m128i *h_store = buf_a.data(), *h_load = buf_b.data();
    const __m128i v_gap_o = _mm_set1_epi8(GAP_OPEN), v_gap_e = _mm_set1_epi8(GAP_EXTEND), v_bias = _mm_set1_epi8((char)bias);
    __m128i run_max = zero, run_mark = zero;
    int best = 0, end_ref = -1;
    const int begin = dir ? ref_len - 1 : 0, stop = dir ? -1 : ref_len, step = dir ? -1 : 1;
    for (int i = begin; i != stop; i += step) {
        __m128i F = zero, cmax = zero;
        __m128i H = _mm_slli_si128(h_store[seg - 1], 1);            // last stripe, moved up one lane
        const __m128i *P = &profile[(size_t)ref[i] * seg];
        std::swap(h_store, h_load);
        for (int j = 0; j < seg; ++j) {
            H = _mm_subs_epu8(_mm_adds_epu8(H, P[j]), v_bias);
            __m128i e = E[(size_t)j];
            H = _mm_max_epu8(_mm_max_epu8(H, e), F);
            cmax = _mm_max_epu8(cmax, H);
            h_store[j] = H;
            H = _mm_subs_epu8(H, v_gap_o);
            E[(size_t)j] = _mm_max_epu8(_mm_subs_epu8(e, v_gap_e), H);
            F = _mm_max_epu8(_mm_subs_epu8(F, v_gap_e), H);
            H = h_load[j];
        }
        // lazy F: carry F across the lane boundary until it can no longer raise any H (E is not refreshed)
        bool settled = false;
        for (int k = 0; k < W && !settled; ++k) {
            F = _mm_slli_si128(F, 1);
            for (int j = 0; j < seg; ++j) {
                __m128i h = _mm_max_epu8(h_store[j], F);
                cmax = _mm_max_epu8(cmax, h);
                h_store[j] = h;
                h = _mm_subs_epu8(h, v_gap_o);
                F = _mm_subs_epu8(F, v_gap_e);
                if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_subs_epu8(F, h), zero)) == 0xffff) { settled = true; break; }
            }
        }
        run_max = _mm_max_epu8(run_max, cmax);
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(run_mark, run_max)) != 0xffff) {
            run_mark = run_max;
            const int top = hmax_u8(run_max);
            if (top > best) {
                best = top;
                if (best + bias >= 255) break;
                end_ref = i;
                std::memcpy(h_best.data(), h_store, (size_t)seg * sizeof(__m128i));
            }
        }
        col_max[(size_t)i] = (uint8_t)hmax_u8(cmax);
        if (col_max[(size_t)i] == (uint8_t)terminate) break;
    }
    int end_read = read_len - 1;
    const uint8_t *hb = reinterpret_cast<const uint8_t *>(h_best.data());
    for (int idx = 0; idx < seg * W; ++idx)
        if (hb[idx] == best) end_read = std::min(end_read, idx / W + idx % W * seg);
    Ends r{best + bias >= 255 ? 255 : best, end_ref, end_read, 0, 0};
    int edge = std::max(end_ref - mask_len, 0);
    for (int i = 0; i < edge; ++i) if (col_max[(size_t)i] > r.score2) { r.score2 = col_max[(size_t)i]; r.ref2 = i; }
    edge = std::min(end_ref + mask_len, ref_len);
    for (int i = edge + 1; i < ref_len; ++i) if (col_max[(size_t)i] > r.score2) { r.score2 = col_max[(size_t)i]; r.ref2 = i; }
    return r;
}

// 16-bit striped pass (ssw.c:412-588): signed add / max, unsigned saturating subtract.
Ends striped_pass_i16(const int8_t *ref, int dir, int ref_len, const int8_t *read, int read_len, int terminate, int mask_len) {
    constexpr int W = 8;
    const int seg = (read_len + W - 1) / W;
    std::vector<__m128i> profile((size_t)NSYM * seg);
    {
        int16_t *t = reinterpret_cast<int16_t *>(profile.data());
        for (int nt = 0; nt < NSYM; ++nt)
            for (int j = 0; j < seg; ++j)
                for (int l = 0; l < W; ++l) {
                    const int q = j + l * seg;
                    *t++ = (int16_t)(q >= read_len ? 0 : score_of(nt, read[q]));
                }
    }
    const __m128i zero = _mm_setzero_si128();
    std::vector<__m128i> buf_a((size_t)seg, zero), buf_b((size_t)seg, zero), E((size_t)seg, zero), h_best((size_t)seg, zero);
    std::vector<uint16_t> col_max((size_t)ref_len, 0);
    __m128i *h_store = buf_a.data(), *h_load = buf_b.data();
    const __m128i v_gap_o = _mm_set1_epi16(GAP_OPEN), v_gap_e = _mm_set1_epi16(GAP_EXTEND);
    __m128i run_max = zero, run_mark = zero;
    int best = 0, end_ref = 0;
    const int begin = dir ? ref_len - 1 : 0, stop = dir ? -1 : ref_len, step = dir ? -1 : 1;
    for (int i = begin; i != stop; i += step) {
        __m128i F = zero, cmax = zero;
        __m128i H = _mm_slli_si128(h_store[seg - 1], 2);
        const __m128i *P = &profile[(size_t)ref[i] * seg];
        std::swap(h_store, h_load);
        for (int j = 0; j < seg; ++j) {
            H = _mm_adds_epi16(H, P[j]);
            __m128i e = E[(size_t)j];
            H = _mm_max_epi16(_mm_max_epi16(H, e), F);
            cmax = _mm_max_epi16(cmax, H);
            h_store[j] = H;
            H = _mm_subs_epu16(H, v_gap_o);
            E[(size_t)j] = _mm_max_epi16(_mm_subs_epu16(e, v_gap_e), H);
            F = _mm_max_epi16(_mm_subs_epu16(F, v_gap_e), H);
            H = h_load[j];
        }
        bool settled = false;
        for (int k = 0; k < W && !settled; ++k) {
            F = _mm_slli_si128(F, 2);
            for (int j = 0; j < seg; ++j) {
                __m128i h = _mm_max_epi16(h_store[j], F);
                cmax = _mm_max_epi16(cmax, h);
                h_store[j] = h;
                h = _mm_subs_epu16(h, v_gap_o);
                F = _mm_subs_epu16(F, v_gap_e);
                if (!_mm_movemask_epi8(_mm_cmpgt_epi16(F, h))) { settled = true; break; }
            }
        }
        run_max = _mm_max_epi16(run_max, cmax);
        if (_mm_movemask_epi8(_mm_cmpeq_epi16(run_mark, run_max)) != 0xffff) {
            run_mark = run_max;
            const int top = hmax_i16(run_max);
            if (top > best) {
                best = top;
                end_ref = i;
                std::memcpy(h_best.data(), h_store, (size_t)seg * sizeof(__m128i));
            }
        }
        col_max[(size_t)i] = (uint16_t)hmax_i16(cmax);
        if (col_max[(size_t)i] == (uint16_t)terminate) break;
    }
    int end_read = read_len - 1;
    const uint16_t *hb = reinterpret_cast<const uint16_t *>(h_best.data());
    for (int idx = 0; idx < seg * W; ++idx)
        if (hb[idx] == (uint16_t)best) end_read = std::min(end_read, idx / W + idx % W * seg);
    Ends r{best, end_ref, end_read, 0, 0};
    int edge = std::max(end_ref - mask_len, 0);
    for (int i = 0; i < edge; ++i) if (col_max[(size_t)i] > r.score2) { r.score2 = col_max[(size_t)i]; r.ref2 = i; }
    edge = std::min(end_ref + mask_len, ref_len);
    for (int i = edge; i < ref_len; ++i) if (col_max[(size_t)i] > r.score2) { r.score2 = col_max[(size_t)i]; r.ref2 = i; }
    return r;
}

struct Op { char op; int len; };

// ---- the path between the located end points ----------------------------------------------------------------
// The library finds it with a banded dynamic programme over the rectangle between the end points (banded_sw,
// ssw.c:590-775), rows = query positions, columns = reference positions, |column - row| <= band, three states per
// cell (H: best score ending here; E: ... in a gap that consumes query = 'I'; F: ... in a gap that consumes
// reference = 'D').  The CIGAR in the BED row is that path, so everything that decides it is the library's and is
// kept as it is there: the recurrences and their tie-breaks (E/F open only when strictly better than extending; H
// takes the diagonal on a tie, E before F only when strictly larger), the band that starts at |ref_len -
// read_len| + 1 and doubles until the banded score reaches the striped score, the zero-valued out-of-band
// neighbours -- including the library's habit of clearing slot `edge` of the previous row, which for the first
// rows of a band that reaches past the reference is a live cell -- and the way the walk closes the path at
// column 0.  What is this file's own: one byte per cell instead of three direction codes (bits 0-1: H came from
// the diagonal / E / F; bit 2: E opened here; bit 3: F opened here), rows addressed relative to their first
// in-band column, an explicit walker over (state, cell), and scratch that lives across calls.
enum : uint8_t { FROM_DIAG = 0, FROM_E = 1, FROM_F = 2, SRC_MASK = 3, E_OPENS = 4, F_OPENS = 8, NO_CELL = 0xff };

struct BandScratch {
    std::vector<int32_t> h_above, e_above, h_row;     // slot u = column - first in-band column of the row + 1; slot 0 = out of band
    std::vector<uint8_t> cell;                        // [row][column - first in-band column]
};

bool banded_path(const int8_t *ref, const int8_t *read, int ref_len, int read_len, int score, int band, std::vector<Op> &out) {
    static thread_local BandScratch s;
    const int longest = std::max(ref_len, read_len);
    int best = 0, row_cells = 0;
    for (;;) {
        row_cells = 2 * band + 1;
        const int slots = row_cells + 2;
        s.h_above.assign((size_t)slots + 1, 0);
        s.e_above.assign((size_t)slots + 1, 0);
        s.h_row.assign((size_t)slots + 1, 0);
        if (s.cell.size() < (size_t)row_cells * read_len) s.cell.resize((size_t)row_cells * read_len);
        int32_t *h_above = s.h_above.data(), *e_above = s.e_above.data(), *h_row = s.h_row.data();
        for (int i = 0; i < read_len; ++i) {
            const int first = std::max(0, i - band), last = std::min(ref_len - 1, i + band);
            const int first_above = std::max(0, i - 1 - band);
            const int edge = std::min(last + 1, slots - 1);
            h_above[0] = e_above[0] = h_above[edge] = e_above[edge] = h_row[0] = 0;
            uint8_t *cells = s.cell.data() + (size_t)row_cells * i;
            int f = 0, u = 0;
            for (int j = first; j <= last; ++j) {
                u = j - first + 1;
                const int above = j - first_above + 1;            // same column, row above; above - 1: the diagonal
                uint8_t code = 0;
                // E: gap in the reference direction of the query ('I'), opened from H above or extended
                const int e_open = i == 0 ? -GAP_OPEN : h_above[above] - GAP_OPEN;
                const int e_ext = i == 0 ? -GAP_EXTEND : e_above[above] - GAP_EXTEND;
                const int e = std::max(e_open, e_ext);
                e_above[u] = e;
                if (e_open > e_ext) code |= E_OPENS;
                // F: gap that consumes reference ('D'), opened from H to the left or extended
                const int f_open = h_row[u - 1] - GAP_OPEN, f_ext = f - GAP_EXTEND;
                f = std::max(f_open, f_ext);
                if (f_open > f_ext) code |= F_OPENS;
                const int e0 = std::max(e, 0), f0 = std::max(f, 0);
                const int gap = std::max(e0, f0);
                const int diag = h_above[above - 1] + score_of(ref[j], read[i]);
                const int h = std::max(gap, diag);
                h_row[u] = h;
                best = std::max(best, h);
                code |= gap <= diag ? FROM_DIAG : (e0 > f0 ? FROM_E : FROM_F);
                cells[j - first] = code;
            }
            for (int k = last - first + 1; k < row_cells; ++k) cells[k] = NO_CELL;      // columns past the reference
            for (int k = 1; k <= u; ++k) h_above[k] = h_row[k];
        }
        if (!(best < score && band * 2 <= longest)) break;
        band *= 2;
    }

    // walk back from the end-point corner
    enum State { IN_E, IN_F, IN_H };
    const long n_cells = (long)row_cells * read_len;
    std::vector<Op> rev;
    int i = read_len - 1, j = ref_len - 1, count = 0;
    State state = IN_H;
    char op = 'M', prev = 'M';
    while (i >= 0 && j > 0) {
        // (a column outside the row's band addresses the neighbouring row's cell, as in the library's flat array)
        const long at = (long)row_cells * i + (j - std::max(0, i - band));
        if (at < 0 || at >= n_cells) return false;
        const uint8_t code = s.cell[(size_t)at];
        if (code == NO_CELL) return false;
        const State via = state != IN_H ? state : (code & SRC_MASK) == FROM_DIAG ? IN_H : (code & SRC_MASK) == FROM_E ? IN_E : IN_F;
        if (via == IN_H) { --i; --j; state = IN_H; op = 'M'; }
        else if (via == IN_E) { --i; state = (code & E_OPENS) ? IN_H : IN_E; op = 'I'; }
        else { --j; state = (code & F_OPENS) ? IN_H : IN_F; op = 'D'; }
        if (op == prev) ++count;
        else { rev.push_back({prev, count}); prev = op; count = 1; }
    }
    if (op == 'M') rev.push_back({op, count + 1});
    else { rev.push_back({op, count}); rev.push_back({'M', 1}); }
    out.assign(rev.rbegin(), rev.rend());
    return true;
}

}  // namespace

void ssw_passes(const char *query, int query_len, const char *ref, int ref_len, int mask_len, SswEnds &e) {
    e = SswEnds{};
    std::vector<int8_t> q((size_t)query_len), r((size_t)std::max(ref_len, 1));
    for (int i = 0; i < query_len; ++i) q[(size_t)i] = translate(query[i]);
    for (int i = 0; i < ref_len; ++i) r[(size_t)i] = translate(ref[i]);

    // forward pass: 8-bit first, 16-bit when it saturates (ssw.c:843-860)
    bool wide = false;
    Ends fwd = striped_pass_u8(r.data(), 0, ref_len, q.data(), query_len, 255, mask_len);
    if (fwd.score == 255) { fwd = striped_pass_i16(r.data(), 0, ref_len, q.data(), query_len, 0xffff, mask_len); wide = true; }
    e.score = fwd.score; e.ref_end = fwd.ref; e.query_end = fwd.read;
    if (mask_len >= 15) { e.score2 = fwd.score2; e.ref_end2 = fwd.ref2; } else { e.score2 = 0; e.ref_end2 = -1; }
    if (e.score == 0 || e.ref_end < 0) { e.ref_end = -1; return; }      // nothing aligned at all: see ssw_finish

    // reverse pass from the end point locates the beginning (ssw.c:874-891)
    std::vector<int8_t> q_rev(q.begin(), q.begin() + e.query_end + 1);
    std::reverse(q_rev.begin(), q_rev.end());
    const Ends rev = wide ? striped_pass_i16(r.data(), 1, e.ref_end + 1, q_rev.data(), e.query_end + 1, e.score, mask_len)
                          : striped_pass_u8(r.data(), 1, e.ref_end + 1, q_rev.data(), e.query_end + 1, e.score, mask_len);
    e.ref_begin = rev.ref;
    e.query_begin = e.query_end - rev.read;
    if (e.score > rev.score) e.flag = 2;
}

namespace {
// everything of an alignment after the passes; gpu_path: the path the GPU found (ssw_path.hip), null = search it here
void finish(const char *query, int query_len, const char *ref, int ref_len, const SswEnds &e, const SswPath *gpu_path, SswResult &out) {
    out.cigar.clear();              // the caller's buffer keeps its capacity from alignment to alignment
    out.mismatches = 0; out.skipped = false;
    out.score = e.score; out.ref_end = e.ref_end; out.query_end = e.query_end;
    out.score2 = e.score2; out.ref_end2 = e.ref_end2;
    out.ref_begin = e.ref_begin; out.query_begin = e.query_begin; out.flag = e.flag;
    if (e.score == 0 || e.ref_end < 0) {
        // nothing aligned at all.  The library would go on to read ref[-1] here (undefined behaviour);
        // defined instead as "no alignment": begin positions stay -1, the CIGAR is one soft clip.
        out.cigar = std::to_string(query_len) + "S";
        out.ref_end = -1; out.ref_begin = -1; out.query_begin = -1;
        return;
    }
    static thread_local std::vector<int8_t> q, r;
    q.resize((size_t)query_len); r.resize((size_t)std::max(ref_len, 1));
    for (int i = 0; i < query_len; ++i) q[(size_t)i] = translate(query[i]);
    for (int i = 0; i < ref_len; ++i) r[(size_t)i] = translate(ref[i]);

    static thread_local std::vector<Op> path;
    path.clear();
    const bool too_far = out.ref_end - out.ref_begin > DISTANCE_FILTER || out.query_end - out.query_begin > DISTANCE_FILTER;
    if (!too_far) {
        if (gpu_path) {
            if (gpu_path->failed) out.flag = 1;
            else for (int k = 0; k < gpu_path->n_ops; ++k) path.push_back({"MID"[gpu_path->ops[k] & 3u], (int)(gpu_path->ops[k] >> 2)});
        } else {
            const int rl = out.ref_end - out.ref_begin + 1, ql = out.query_end - out.query_begin + 1;
            if (!banded_path(r.data() + out.ref_begin, q.data() + out.query_begin, rl, ql, out.score, std::abs(rl - ql) + 1, path)) {
                out.flag = 1;
                path.clear();
            }
        }
    }

    // CIGAR with '=' / 'X' and soft clips, mismatch count (ssw_cpp.cpp:126-207)
    std::string &c = out.cigar;
    auto put = [&](int n, char op) {
        char buf[12];
        int at = 12;
        unsigned v = (unsigned)n;
        do { buf[--at] = (char)('0' + v % 10u); v /= 10u; } while (v);
        c.append(buf + at, (size_t)(12 - at));
        c += op;
    };
    if (out.query_begin > 0) put(out.query_begin, 'S');
    const int8_t *rp = r.data() + out.ref_begin, *qp = q.data() + out.query_begin;
    int run_eq = 0, run_x = 0;
    auto close_run = [&]() { if (run_eq) put(run_eq, '='); else if (run_x) put(run_x, 'X'); run_eq = run_x = 0; };
    for (const Op &o : path) {
        if (o.op == 'M') {
            for (int k = 0; k < o.len; ++k, ++rp, ++qp) {
                if (*rp != *qp) { ++out.mismatches; if (run_eq) put(run_eq, '='); run_eq = 0; ++run_x; }
                else { if (run_x) put(run_x, 'X'); run_x = 0; ++run_eq; }
            }
        } else if (o.op == 'I') { qp += o.len; out.mismatches += o.len; close_run(); put(o.len, 'I'); }
        else if (o.op == 'D') { rp += o.len; out.mismatches += o.len; close_run(); put(o.len, 'D'); }
    }
    close_run();
    const int tail = query_len - out.query_end - 1;
    if (tail > 0) put(tail, 'S');
}
}  // namespace

void ssw_finish(const char *query, int query_len, const char *ref, int ref_len, const SswEnds &e, SswResult &out) {
    finish(query, query_len, ref, ref_len, e, nullptr, out);
}

void ssw_finish_with_path(const char *query, int query_len, const char *ref, int ref_len, const SswEnds &e, const SswPath &path, SswResult &out) {
    finish(query, query_len, ref, ref_len, e, &path, out);
}

// The same result as ssw_finish_with_path for a reference that is `motif` (atom bases) repeated -- which every reference of
// ribbit's alignments is (the pseudo-perfect repeat, parse_seed.cpp:404) -- without the reference: with end points and path
// known, all that is left of ssw_cpp.cpp:126-207 is to split the path's M runs into '=' / 'X' and to count, and that reads the
// aligned windows only.  (finish() translates the whole query and the whole reference first, and its caller had spelt the
// reference out to the padded length: three passes over more bases than the alignment covers, ten million times a chromosome.)
void ssw_finish_with_path_periodic(const char *query, int query_len, const char *motif, int atom, const SswEnds &e, const SswPath &path, SswResult &out) {
    out.cigar.clear();
    out.mismatches = 0; out.skipped = false;
    out.score = e.score; out.ref_end = e.ref_end; out.query_end = e.query_end;
    out.score2 = e.score2; out.ref_end2 = e.ref_end2;
    out.ref_begin = e.ref_begin; out.query_begin = e.query_begin; out.flag = e.flag;
    if (e.score == 0 || e.ref_end < 0) {            // as finish(): "no alignment"
        out.cigar = std::to_string(query_len) + "S";
        out.ref_end = -1; out.ref_begin = -1; out.query_begin = -1;
        return;
    }
    std::string &c = out.cigar;
    auto put = [&](int n, char op) {
        char buf[12];
        int at = 12;
        unsigned v = (unsigned)n;
        do { buf[--at] = (char)('0' + v % 10u); v /= 10u; } while (v);
        c.append(buf + at, (size_t)(12 - at));
        c += op;
    };
    const bool too_far = out.ref_end - out.ref_begin > DISTANCE_FILTER || out.query_end - out.query_begin > DISTANCE_FILTER;
    if (path.failed && !too_far) out.flag = 1;
    if (out.query_begin > 0) put(out.query_begin, 'S');
    if (!too_far && !path.failed && atom > 0) {
        int8_t tm[1024];                              // the motif, translated (a motif is at most a few hundred bases)
        static thread_local std::vector<int8_t> tm_long;
        const int8_t *t = tm;
        if (atom > 1024) { tm_long.resize((size_t)atom); for (int k = 0; k < atom; ++k) tm_long[(size_t)k] = translate(motif[k]); t = tm_long.data(); }
        else for (int k = 0; k < atom; ++k) tm[k] = translate(motif[k]);
        const char *qp = query + out.query_begin;
        int rj = out.ref_begin % atom;                // reference position, modulo the motif
        int run_eq = 0, run_x = 0;
        auto close_run = [&]() { if (run_eq) put(run_eq, '='); else if (run_x) put(run_x, 'X'); run_eq = run_x = 0; };
        for (int k = 0; k < path.n_ops; ++k) {
            const int len = (int)(path.ops[k] >> 2), op = (int)(path.ops[k] & 3u);
            if (op == 0) {
                for (int i = 0; i < len; ++i, ++qp) {
                    if (t[rj] != translate(*qp)) { ++out.mismatches; if (run_eq) put(run_eq, '='); run_eq = 0; ++run_x; }
                    else { if (run_x) put(run_x, 'X'); run_x = 0; ++run_eq; }
                    if (++rj == atom) rj = 0;
                }
            } else if (op == 1) { qp += len; out.mismatches += len; close_run(); put(len, 'I'); }
            else if (op == 2) { rj = (int)(((long)rj + len) % atom); out.mismatches += len; close_run(); put(len, 'D'); }
        }
        close_run();
    }
    const int tail = query_len - out.query_end - 1;
    if (tail > 0) put(tail, 'S');
}

// Host twin of what the GPU hands refinement for an alignment -- end points, and the path as run-length operations (op 0 'M',
// 1 'I', 2 'D'; length << 2) -- finished by ssw_finish_with_path_periodic: the whole alignment against `motif` repeated, without
// a GPU, so that the CPU tests can hold the periodic finish against ssw_align on the spelt-out reference (tests/test_ssw.py).
void ssw_align_periodic(const char *query, int query_len, const char *motif, int atom, int ref_len, int mask_len, SswResult &out) {
    if (query_len <= 0 || atom <= 0) {
        out = SswResult{};
        out.ref_begin = -1; out.query_begin = -1;
        out.skipped = true;
        return;
    }
    std::string ref;
    while ((long)ref.size() <= (long)ref_len) ref.append(motif, (size_t)atom);
    SswEnds ends;
    ssw_passes(query, query_len, ref.data(), ref_len, mask_len, ends);
    std::vector<uint32_t> packed;
    SswPath path;
    if (ends.score != 0 && ends.ref_end >= 0) {
        const bool too_far = ends.ref_end - ends.ref_begin > DISTANCE_FILTER || ends.query_end - ends.query_begin > DISTANCE_FILTER;
        if (!too_far) {
            std::vector<int8_t> q((size_t)query_len), r((size_t)std::max(ref_len, 1));
            for (int i = 0; i < query_len; ++i) q[(size_t)i] = translate(query[i]);
            for (int i = 0; i < ref_len; ++i) r[(size_t)i] = translate(ref[(size_t)i]);
            const int rl = ends.ref_end - ends.ref_begin + 1, ql = ends.query_end - ends.query_begin + 1;
            std::vector<Op> ops;
            if (!banded_path(r.data() + ends.ref_begin, q.data() + ends.query_begin, rl, ql, ends.score, std::abs(rl - ql) + 1, ops)) path.failed = true;
            else for (const Op &o : ops) packed.push_back(((uint32_t)o.len << 2) | (o.op == 'M' ? 0u : o.op == 'I' ? 1u : 2u));
        }
    }
    path.ops = packed.empty() ? nullptr : packed.data();
    path.n_ops = (int32_t)packed.size();
    ssw_finish_with_path_periodic(query, query_len, motif, atom, ends, path, out);
}

void ssw_align(const char *query, int query_len, const char *ref, int ref_len, int mask_len, SswResult &out) {
    if (query_len <= 0) {       // Aligner::Align returns before touching `alignment`
        out = SswResult{};
        out.ref_begin = -1; out.query_begin = -1;
        out.skipped = true;
        return;
    }
    SswEnds ends;
    ssw_passes(query, query_len, ref, ref_len, mask_len, ends);
    ssw_finish(query, query_len, ref, ref_len, ends, out);
}

}  // namespace rb
