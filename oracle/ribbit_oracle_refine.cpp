/*
 * ribbit_oracle_refine.cpp -- CPU restatement of the per-seed refinement scans of ribbit that run
 * between seed dispatch and the Smith-Waterman alignment:
 * longestContinuousMatches, possibleMotifs, mostFrequentLongerMotif, calculateRepeatClass,
 * calculateAtomicity*, calculateMotif and the parts of processSeedMotifWise / processSeed that
 * build the alignment job (query range, motif, pseudo-perfect-repeat length).
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see ribbit_oracle.h).  C++ because the order of BED
 * rows inside one small-motif seed is the iteration order of a libstdc++
 * std::unordered_map<uint32_t,int> (parse_smallmotif_seed.cpp:177): the same container
 * with the same insertion sequence is the only faithful restatement of that.
 *
 * Defined divergences (undefined behaviour / crash in the reference):
 *   D3  parse_smallmotif_seed.cpp:221-226 / parse_seed.cpp:344-349 read N_bset beyond the record when a
 *       merge re-labelled a seed with a longer motif after the edge clamp; positions >= L are not N here.
 *   D4  a motif window may start before position 0 (wstart = j-(m-1) < 0 for a seed in the first m-1
 *       bases); the reference then calls string::substr with a huge offset and terminates.  The job is
 *       reported with the negative start as computed; consumers must clamp.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "ribbit_oracle.h"

namespace {

/* boost::multiprecision::uint256_t as used by the reference: unchecked, wraps mod 2^256 (Q11) */
struct u256 {
    uint64_t w[4] = {0, 0, 0, 0};
    void shl(unsigned k) {
        while (k >= 64) { w[3] = w[2]; w[2] = w[1]; w[1] = w[0]; w[0] = 0; k -= 64; }
        if (k) { w[3] = (w[3] << k) | (w[2] >> (64 - k)); w[2] = (w[2] << k) | (w[1] >> (64 - k));
                 w[1] = (w[1] << k) | (w[0] >> (64 - k)); w[0] <<= k; }
    }
    void shr(unsigned k) {
        while (k >= 64) { w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = 0; k -= 64; }
        if (k) { w[0] = (w[0] >> k) | (w[1] << (64 - k)); w[1] = (w[1] >> k) | (w[2] << (64 - k));
                 w[2] = (w[2] >> k) | (w[3] << (64 - k)); w[3] >>= k; }
    }
    bool eq(const u256 &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
    u256 band(const u256 &o) const { u256 r; for (int i = 0; i < 4; i++) r.w[i] = w[i] & o.w[i]; return r; }
    static u256 low_mask(int bits) {            /* `mask <<= 1; mask |= 1;` bits times */
        u256 m;
        for (int i = 0; i < bits; i++) { m.shl(1); m.w[0] |= 1; }
        return m;
    }
};

struct View {
    const uint8_t *code, *nmask;
    int L;
};

/* longestContinuousMatches, parse_seed.cpp:26-44 (on the bits of plane[start..end)) */
int longest_run(const uint8_t *plane, int start, int end) {
    int l = 0, best = 0;
    for (int j = start; j < end; j++) {
        if (plane[j] == 1) l += 1;
        else { if (l > best) best = l; l = 0; }
    }
    if (l > best) best = l;
    return best;
}

/* calculateRepeatClass, bitseq_utils.cpp:185-221: smallest of the m cyclic rotations of a 2m-bit word */
uint32_t repeat_class(uint32_t motif, int m) {
    const uint32_t mask = (m >= 16) ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    uint32_t best = motif;
    for (int i = 0; i < m - 1; i++) {
        const uint32_t cycle = ((motif >> (2 * (m - (i + 1)))) | (motif << (2 * (i + 1)))) & mask;
        if (cycle < best) best = cycle;
    }
    return best;
}

/* calculateAtomicity(uint32_t&, int&), bitseq_utils.cpp:139-183 (the memo only caches this value) */
int atomicity_small(uint32_t motif, int m) {
    for (int f = 1; f <= m / 2; f++) {
        if (m % f) continue;
        uint32_t mask = 0;
        for (int i = 0; i < 2 * (m - f); i++) { mask <<= 1; mask |= 1; }
        if ((motif >> (2 * f)) == (mask & motif)) return f;
    }
    return m;
}

/* calculateAtomicityLongMotif, bitseq_utils.cpp:116-137 */
int atomicity_long(const u256 &motif, int m) {
    for (int f = 1; f < m - m / 3; f++) {
        u256 shifted = motif; shifted.shr(2 * f);
        if (shifted.eq(u256::low_mask(2 * (m - f)).band(motif))) return f;
    }
    return m;
}

/* calculateMotif, bitseq_utils.cpp:14-38 */
std::string motif_string(const u256 &unit, int m) {
    std::string s;
    for (int i = 0; i < m; i++) {
        u256 v = unit; v.shr(2 * (m - 1 - i));
        s += "ACGT"[v.w[0] & 3];
    }
    return s;
}
std::string motif_string(uint32_t unit, int m) { u256 u; u.w[0] = unit; return motif_string(u, m); }

struct SmallMotif { uint32_t motif; int start, end; };

/* possibleMotifs, parse_smallmotif_seed.cpp:76-188 */
void possible_motifs(const View &v, int seed_start, int seed_sequence_length, int m, const rbo_refine_params_t &prm,
                     std::vector<SmallMotif> &out) {
    static std::vector<int> M_START(1 << 20), M_END(1 << 20), M_UNITS(1 << 20), M_GAPS(1 << 20), M_GAPSIZE(1 << 20);
    static std::vector<uint32_t> M_NEXT(1 << 20);
    std::unordered_map<uint32_t, int> new_motif_start;
    int seed_end = seed_start + seed_sequence_length;
    if (seed_end > v.L - 1) seed_end = v.L - 1;                                         /* :94 */
    const uint32_t wmask = (m >= 16) ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    const int min_len = prm.min_length[m], min_units = prm.perfect_units[m];
    uint32_t window = 0;
    for (int j = seed_start; j < seed_end; j++) {
        window = (window & ~3u) | v.code[j];                                            /* :99 window[0]=right, [1]=left */
        const uint32_t motif = repeat_class(window, m);
        const int wstart = j - (m - 1), wend = j + 1;
        const uint32_t next = ((window << 2) | (window >> ((m - 1) * 2))) & wmask;      /* :115 */
        if (j - seed_start >= (0.9 * m) - 1) {                                          /* :104 */
            auto it = new_motif_start.find(motif);
            if (it == new_motif_start.end()) {
                new_motif_start[motif] = wstart;
                M_START[motif] = wstart; M_END[motif] = wend; M_UNITS[motif] = 1;
                M_GAPS[motif] = 0; M_GAPSIZE[motif] = 0; M_NEXT[motif] = next;
            } else if (wstart - M_END[motif] > 3 * m) {                                 /* :120 */
                if (M_END[motif] - M_START[motif] >= min_len && M_UNITS[motif] >= min_units)
                    out.push_back({motif, M_START[motif], M_END[motif]});
                M_START[motif] = wstart; M_END[motif] = wend; M_UNITS[motif] = 1;
                M_GAPS[motif] = 0; M_GAPSIZE[motif] = 0; M_NEXT[motif] = next;
                new_motif_start[motif] = wstart;
            } else {
                if (M_END[motif] < j) {                                                 /* :143 */
                    if (j - M_END[motif] < m) { M_GAPS[motif] += 1; M_GAPSIZE[motif] += 1; }
                    else if ((j - M_END[motif]) % m > 0) { M_GAPS[motif] += ((j - M_END[motif]) / m) + 1; M_GAPSIZE[motif] += (j - M_END[motif]) + 1; }
                    else { M_GAPS[motif] += ((j - M_END[motif]) / m); M_GAPSIZE[motif] += (j - M_END[motif]); }
                } else if (M_END[motif] == j && M_NEXT[motif] != window) {
                    M_GAPS[motif] += 1; M_GAPSIZE[motif] += 1;
                }
                if (wstart - new_motif_start[motif] >= m) {                             /* :162 */
                    new_motif_start[motif] = wstart;
                    M_UNITS[motif] += 1;
                }
                M_END[motif] = wend;
                M_NEXT[motif] = next;
            }
        }
        window = (window << 2) & wmask;                                                 /* :173 */
    }
    for (auto &it : new_motif_start) {                                                  /* :177 unordered_map order (Q10) */
        const uint32_t motif = it.first;
        if (M_END[motif] - M_START[motif] >= min_len && M_UNITS[motif] >= min_units)
            out.push_back({motif, M_START[motif], M_END[motif]});
    }
}

/* mostFrequentLongerMotif, parse_seed.cpp:153-256.  (*MATRIX[r])[c] == 1  <=>  base c is not N and equals base r */
u256 most_frequent_longer_motif(const View &v, int seed_start, int seed_sequence_length, int m) {
    int seed_end = seed_start + seed_sequence_length;
    if (seed_end > v.L) seed_end = v.L;   /* D3 regime only: the reference would index MATRIX past the record */
    auto same = [&](int row, int col) { return !v.nmask[col] && v.code[col] == v.code[row]; };
    int best_row = 0, best_count = 0;
    for (int row_start = seed_start; row_start < seed_end - m + 1; row_start++) {
        int row_count = 0;
        int down = row_start + m;
        while (down < seed_end) {                                                       /* :181 */
            int best_x = -2, best_d = 0;
            for (int x = -2; x < 3; x++) {
                int d = 0;
                for (int i = 0; i < m; i++) {
                    if (down + x + i >= seed_end) break;
                    if (same(row_start + i, down + x + i)) d += 1;
                }
                if (d > best_d) { best_d = d; best_x = x; }
            }
            row_count += best_d;
            down += best_x;
            down += m;
        }
        int up = row_start - m;
        while (up > seed_start) {                                                       /* :201 */
            int best_x = -2, best_d = 0;
            for (int x = -2; x < 3; x++) {
                int d = 0;
                for (int i = 0; i < m; i++) {
                    if (up + x + i < 0) break;
                    if (same(row_start + i, up + x + i)) d += 1;
                }
                if (d > best_d) { best_d = d; best_x = x; }
            }
            row_count += best_d;
            up += best_x;
            up -= m;
        }
        if (up < seed_start && std::abs(up - seed_start) < m) {                         /* :219 */
            const int last_row = row_start + m - 1;
            const int pc = seed_start + ((m + (up - seed_start)) - 1);
            const int prefix_rows = m + (up - seed_start);
            int best_d = 0;
            for (int x = -2; x < 3; x++) {
                int d = 0;
                for (int i = 0; i < prefix_rows; i++) {
                    if (pc + x - i >= seed_end || pc + x - i < seed_start) break;
                    if (same(last_row - i, pc + x - i)) d += 1;
                }
                if (d > best_d) best_d = d;
            }
            row_count += best_d;
        }
        if (row_count > best_count) { best_count = row_count; best_row = row_start; }
    }
    if (best_count == 0) best_row = 0;   /* mmotif_index keeps its initial value 0 (:165) */
    u256 unit;
    for (int j = best_row; j < best_row + m; j++) {                                     /* :246-253 */
        unit.shl(1); if (v.code[j] >> 1) unit.w[0] |= 1;
        unit.shl(1); if (v.code[j] & 1) unit.w[0] |= 1;
    }
    return unit;
}

/* seed_sequence_length of parse_smallmotif_seed.cpp:219-226 / parse_seed.cpp:342-349 (D3 guard) */
int seed_sequence_length(const View &v, int seed_start, int seed_end, int m) {
    int len = (seed_end - seed_start) + m;
    for (int s = seed_start; s < seed_end + m; s++) {
        if (s < v.L && v.nmask[s] == 1) { len = s - seed_start; break; }
    }
    return len;
}

/* `int ppr_length = a + m + ((1-PURITY_THRESHOLD)*b);` -- int + int + float, truncated */
int ppr_length(int a, int m, int b, float purity) {
    const float f = (float)(a + m) + ((1 - purity) * (float)b);
    return (int)f;
}

struct Store {
    std::vector<rbo_job_t> jobs;
    std::string pool;
};

}  // namespace

extern "C" {

/* defaults of ribbit.cpp:151-174 plus the factor completion of :219-235 */
void rbo_refine_params_default(rbo_refine_params_t *p, int m_lo, int m_hi) {
    memset(p, 0, sizeof *p);
    p->purity_threshold = 0.85f;            /* global_variables.cpp:44; -p is ignored (Q1) */
    p->continuous_ones_threshold = 3;       /* ribbit.cpp:191 */
    std::vector<char> has_len(RBO_TABLE, 0);
    for (int k = m_lo; k <= m_hi && k < RBO_TABLE; k++) { p->min_length[k] = (12 < 2 * k) ? 2 * k : 12; has_len[k] = 1; }
    for (int m = 1; m <= m_hi && m < RBO_TABLE; m++) p->perfect_units[m] = (m == 1) ? 8 : (m == 2) ? 4 : (m == 3) ? 3 : 2;
    for (int m = m_lo; m <= m_hi && m < RBO_TABLE; m++)
        for (int f = 1; f <= m / 2; f++)
            if (m % f == 0 && !has_len[f]) { p->min_length[f] = p->min_length[m]; has_len[f] = 1; }
}

int64_t rbo_refine_jobs(rbo_ctx *c, const rbo_refine_params_t *prm, const rbo_job_t **jobs, const char **pool) {
    static Store store;
    store.jobs.clear(); store.pool.clear();
    const rbo_seed_t *seeds;
    const int64_t n = rbo_dispatch(c, &seeds);
    View v{rbo_codes(c), rbo_nmask(c), (int)rbo_length(c)};
    for (int64_t si = 0; si < n; si++) {
        const int start = seeds[si].start, end = seeds[si].end, m = seeds[si].mlen, type = seeds[si].type;
        const uint8_t *plane = rbo_plane(c, m);
        const int ssl = seed_sequence_length(v, start, end, m);
        if (m <= 10) {                                                       /* processSeedMotifWise */
            if (longest_run(plane, start, end) < prm->continuous_ones_threshold) continue;     /* :234-235 */
            std::vector<SmallMotif> found;
            possible_motifs(v, start, ssl, m, *prm, found);
            for (const SmallMotif &sm : found) {
                const int atom = atomicity_small(sm.motif, m);
                const std::string mot = motif_string(sm.motif, m).substr(0, atom);
                rbo_job_t j;
                j.seed_index = (int)si; j.seed_type = type; j.motif_length = m; j.atomicity = atom;
                j.query_start = sm.start; j.query_length = sm.end - sm.start;
                j.ppr_length = ppr_length(sm.end - sm.start, m, sm.end - sm.start, prm->purity_threshold);   /* :267 */
                j.small = 1; j.motif_offset = (int)store.pool.size();
                store.pool += mot;
                store.jobs.push_back(j);
            }
        } else {                                                             /* processSeed, first level */
            if (end - start < 0.9 * m) continue;                                                /* :360 */
            if (longest_run(plane, start, end) < prm->continuous_ones_threshold) continue;      /* :366-367 */
            const u256 unit = most_frequent_longer_motif(v, start, ssl, m);
            const int atom = atomicity_long(unit, m);
            if (m % atom != 0) continue;                                                        /* :392 */
            const std::string mot = motif_string(unit, m).substr(0, atom);
            rbo_job_t j;
            j.seed_index = (int)si; j.seed_type = type; j.motif_length = m; j.atomicity = atom;
            j.query_start = start;
            j.query_length = (start + ssl > v.L) ? v.L - start : ssl;                           /* substr clamps */
            j.ppr_length = ppr_length(ssl, m, ssl, prm->purity_threshold);                      /* :379 */
            j.small = 0; j.motif_offset = (int)store.pool.size();
            store.pool += mot;
            store.jobs.push_back(j);
        }
    }
    *jobs = store.jobs.data();
    *pool = store.pool.data();
    return (int64_t)store.jobs.size();
}

}  // extern "C"
