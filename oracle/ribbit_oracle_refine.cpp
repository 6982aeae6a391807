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
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
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
    /* D4: a position below 0 (reachable only after a motif window that starts before the record, where the
     * reference has already terminated in substr) reads as base A, not N */
    uint8_t code_at(int p) const { return p < 0 ? 0 : code[p]; }
};

/* one shift-XOR plane of the oracle (one bit per base, ribbit_oracle.c) read like the reference's bitset */
struct Plane {
    const uint64_t *w;
    int operator[](int p) const { return (int)((w[p >> 6] >> (p & 63)) & 1); }
};

/* longestContinuousMatches, parse_seed.cpp:26-44 (on the bits of plane[start..end)) */
int longest_run(const Plane &plane, int start, int end) {
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
        window = (window & ~3u) | v.code_at(j);                                         /* :99 window[0]=right, [1]=left */
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
        const Plane plane{rbo_plane_bits(c, m)};
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

/* ===================================================================================================
 * Rows f1 / f4: alignment, CIGAR processing and BED rows.  The alignment itself is NOT restated here:
 * the oracle calls the reference's own vendored SSW, compiled from /root/reference into
 * oracle/_ref/libssw_ref.so (entry point ref_ssw_align, oracle/ssw_ref_shim.cpp).
 * =================================================================================================== */
namespace {

struct ref_ssw_result {
    int32_t sw_score, sw_score_next_best, ref_begin, ref_end, query_begin, query_end, ref_end_next_best, mismatches;
    int32_t flag, cigar_len;
};
typedef int (*ref_ssw_fn)(const char *, const char *, int, int, ref_ssw_result *, char *, int);

ref_ssw_fn load_ref_ssw() {
    static ref_ssw_fn fn = nullptr;
    if (fn) return fn;
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void *)&load_ref_ssw, &info) && info.dli_fname) {
        dir = info.dli_fname;
        dir = dir.substr(0, dir.find_last_of('/'));
    }
    void *h = dlopen((dir + "/_ref/libssw_ref.so").c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) return nullptr;
    fn = (ref_ssw_fn)dlsym(h, "ref_ssw_align");
    return fn;
}

/* Aligner().Align(query, ref, ref_len, Filter(), &alignment, 15) -> alignment.cigar_string.
 * An empty query makes Align return before touching `alignment` (ssw_cpp.cpp:365): the caller then sees the
 * previous call's CIGAR; `stale` carries it. */
std::string align_cigar(const std::string &query, const std::string &ref, int ref_len, std::string &stale) {
    if (query.empty()) return stale;
    ref_ssw_fn fn = load_ref_ssw();
    if (!fn) { std::fprintf(stderr, "oracle: oracle/_ref/libssw_ref.so is missing\n"); std::abort(); }
    ref_ssw_result r;
    std::vector<char> buf(16 * (query.size() + ref.size()) + 64);
    fn(query.c_str(), ref.c_str(), ref_len, 15, &r, buf.data(), (int)buf.size());
    stale = buf.data();
    return stale;
}

/* cigarSplit, process_cigar.cpp:14-31 */
void cigar_split(const std::string &cigar, std::vector<int> &lens, std::vector<char> &ops) {
    std::string num;
    for (char ch : cigar) {
        if (ch >= '0' && ch <= '9') num += ch;
        else { lens.push_back(num.empty() ? 0 : std::stoi(num)); ops.push_back(ch); num.clear(); }
    }
}

/* calculateTrimEdges (float overload), process_cigar.cpp:34-86; purity and alignment_length are in/out.
 * D5: the reference's loop bound is computed in size_t and can wrap when every trim combination is
 * exhausted; signed arithmetic here (unreachable for CIGARs that start and end with a match block). */
std::pair<int, int> trim_edges(float threshold, float &purity, const std::vector<int> &cc, int &alignment_length, int min_len) {
    int trim = 0;
    std::pair<int, int> edges(0, 0);
    const long n = (long)cc.size();
    while (purity < threshold) {
        trim += 1;
        float max_purity = 0; int max_alen = 0;
        for (int i = 0; i <= trim; i++) {
            int pm = 0, pa = 0;
            for (long j = 2L * i; j <= (n - 1) - 2L * (trim - i); j++) {
                if (j % 2 == 0) pm += cc[(size_t)j];
                pa += cc[(size_t)j];
            }
            const float pp = float(pm) / float(pa);
            if (pp >= threshold && max_alen < pa) { max_purity = pp; max_alen = pa; edges = {i, trim - i}; }
        }
        if (max_purity > purity) { purity = max_purity; alignment_length = max_alen; }
        if (alignment_length < min_len) break;
        if (2L * trim > n + 2) break;     /* D5 guard */
    }
    return edges;
}

struct Processed { int repeat_start, repeat_end, alignment_length, match_units; std::string cigar; float purity; };

/* processCIGARMotifWise (process_cigar.cpp:254-336) and, with prune, processCIGARWithPruning (:126-251) */
Processed process_cigar(int seed_start, int seed_sequence_length, const std::string &cigar, int motif_length, bool prune,
                        const rbo_refine_params_t &prm) {
    std::vector<int> lens; std::vector<char> ops;
    cigar_split(cigar, lens, ops);
    int repeat_start = seed_start, repeat_end = seed_start + seed_sequence_length, alignment_length = 0;
    int matches = 0, match_units = 0, start_soft_clip = 0;
    std::vector<int> cc_idx, cc_len;
    bool mismatch_continue = false;
    std::string new_cigar;
    for (size_t c = 0; c < lens.size(); c++) {
        const int len = lens[c]; const char op = ops[c];
        switch (op) {
            case 'S':
                if (c == 0) { repeat_start += len; start_soft_clip = len; } else repeat_end -= len;
                break;
            case 'X': case 'I': case 'D':
                alignment_length += len;
                if (mismatch_continue) cc_len.back() += len; else cc_len.push_back(len);
                cc_idx.push_back((int)cc_len.size() - 1);
                mismatch_continue = true; new_cigar += std::to_string(len) + op;
                break;
            case '=': case 'M':
                alignment_length += len; matches += len; match_units += len / motif_length;
                cc_len.push_back(len); cc_idx.push_back((int)cc_len.size() - 1);
                mismatch_continue = false; new_cigar += std::to_string(len) + op;
                break;
            default: break;
        }
    }
    float purity = float(matches) / float(alignment_length);
    if (prune && purity < prm.purity_threshold) {
        float thr = prm.purity_threshold;
        const std::pair<int, int> te = trim_edges(thr, purity, cc_len, alignment_length, prm.min_length[motif_length]);
        new_cigar.clear(); matches = 0; match_units = 0;
        for (size_t i = 0; i < cc_idx.size(); i++) {
            const int ccidx = cc_idx[i];
            const size_t src = start_soft_clip ? i + 1 : i;
            const int len = lens[src]; const char op = ops[src];
            if (ccidx < 2 * te.first) {
                if (op != 'D') repeat_start += len;
            } else if (ccidx >= 2 * te.first && (long)ccidx <= (long)cc_len.size() - 1 - 2L * te.second) {
                new_cigar += std::to_string(len) + op;
                if (op == 'M' || op == '=') { matches += len; match_units += len / motif_length; }
            } else {
                if (op != 'D') repeat_end -= len;
            }
        }
    }
    return Processed{repeat_start, repeat_end, alignment_length, match_units, new_cigar, purity};
}

/* calculateMotifUnits, parse_smallmotif_seed.cpp:26-72 */
int motif_units_of(const View &v, int start, int length, int m, uint32_t unit) {
    std::unordered_map<uint32_t, int> pos, units;
    int seed_end = start + length;
    if (seed_end > v.L - 1) seed_end = v.L - 1;
    const uint32_t wmask = (m >= 16) ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    uint32_t window = 0;
    for (int j = start; j < seed_end; j++) {
        window = (window & ~3u) | v.code_at(j);
        if (j - start >= (0.9 * m) - 1) {
            const uint32_t motif = repeat_class(window, m);
            if (pos.find(motif) == pos.end()) { pos[motif] = j - (m - 1); units[motif] = 1; }
            else if ((j - (m - 1)) - pos[motif] >= m) { pos[motif] = j - (m - 1); units[motif] += 1; }
        }
        window = (window << 2) & wmask;
    }
    return units[unit];
}

struct BedOut {
    std::ostringstream os;
    std::string id;
    std::string stale_cigar;      /* Alignment object shared by all seeds of a record (fasta_utils.cpp:177) */
    void row(int start, int end, const std::string &motif, int atom, int m, float purity, int type, const std::string &cigar) {
        os << id << "\t" << start << "\t" << end << "\t" << motif << "\t" << atom << " | " << m << "\t" << end - start << "\t"
           << (end - start) / atom << "\t" << purity << "\t" << "+\tSEED-" << type << "\t" << cigar << "\n";
    }
};

std::string substr_clamped(const char *seq, int L, int start, int len) {     /* std::string::substr; D4: start < 0 clamps to 0 */
    if (start < 0) { len += start; start = 0; }
    if (start >= L || len <= 0) return std::string();
    if (start + len > L) len = L - start;
    return std::string(seq + start, (size_t)len);
}

std::string repeat_past(const std::string &motif, int ppr_len) {            /* while (ppr.length() <= ppr_length) ppr += motif; */
    std::string s;
    while ((long)s.size() <= (long)ppr_len) s += motif;
    return s;
}

/* processSeedMotifWise, parse_smallmotif_seed.cpp:190-288 */
void refine_small(const View &v, const char *seq, const Plane &plane, int start, int end, int m, int type,
                  const rbo_refine_params_t &prm, BedOut &out) {
    const int ssl = seed_sequence_length(v, start, end, m);
    if (longest_run(plane, start, end) < prm.continuous_ones_threshold) return;
    std::vector<SmallMotif> found;
    possible_motifs(v, start, ssl, m, prm, found);
    for (const SmallMotif &sm : found) {
        const int atom = atomicity_small(sm.motif, m);
        const std::string motif = motif_string(sm.motif, m).substr(0, atom);
        const uint32_t unit = sm.motif >> (2 * (m - atom));
        const std::string query = substr_clamped(seq, v.L, sm.start, sm.end - sm.start);
        const int ppr_len = ppr_length(sm.end - sm.start, m, sm.end - sm.start, prm.purity_threshold);
        const std::string cigar = align_cigar(query, repeat_past(motif, ppr_len), ppr_len, out.stale_cigar);
        const Processed p = process_cigar(sm.start, sm.end - sm.start, cigar, atom, false, prm);
        const int repeat_length = p.repeat_end - p.repeat_start;
        const int units = motif_units_of(v, p.repeat_start, repeat_length, atom, unit);
        if (units >= prm.perfect_units[atom] && repeat_length >= prm.min_length[atom])
            out.row(p.repeat_start, p.repeat_end, motif, atom, m, p.purity, type, p.cigar);
    }
}

/* processSeed, parse_seed.cpp:318-464 (recursive on the flanks of the aligned repeat) */
void refine_long(const View &v, const char *seq, const Plane &plane, int start, int end, int m, int type,
                 const rbo_refine_params_t &prm, BedOut &out, int depth) {
    if (depth > 10000) return;
    const int ssl = seed_sequence_length(v, start, end, m);
    if (end - start < 0.9 * m) return;
    if (longest_run(plane, start, end) < prm.continuous_ones_threshold) return;
    const int ppr_len = ppr_length(ssl, m, ssl, prm.purity_threshold);
    const u256 unit = most_frequent_longer_motif(v, start, ssl, m);
    const int atom = atomicity_long(unit, m);
    if (m % atom != 0) return;
    const std::string motif = motif_string(unit, m).substr(0, atom);
    const std::string query = substr_clamped(seq, v.L, start, ssl);
    const std::string cigar = align_cigar(query, repeat_past(motif, ppr_len), ppr_len, out.stale_cigar);
    const Processed p = process_cigar(start, ssl, cigar, atom, true, prm);
    if (p.alignment_length >= prm.min_length[atom]) {
        if (p.repeat_end - p.repeat_start >= prm.min_length[m])
            out.row(p.repeat_start, p.repeat_end, motif, atom, m, p.purity, type, p.cigar);
    }
    /* one locus (repeat_start, repeat_end - atomicity); flanks of at least MINIMUM_LENGTH[m] are re-processed (:443-463) */
    int locus_first = p.repeat_start;
    const int locus_second = p.repeat_end - atom;
    int flank_start = start;
    if (flank_start >= locus_first) {
        flank_start = locus_second;
    } else {
        if (locus_first - flank_start >= prm.min_length[m]) {
            if (locus_first > end) locus_first = end;
            if (!(flank_start == start && locus_first == end))
                refine_long(v, seq, plane, flank_start, locus_first, m, type, prm, out, depth + 1);
        }
        flank_start = locus_second;
    }
    if (end - flank_start >= prm.min_length[m]) {
        if (flank_start < start) flank_start = start;
        if (flank_start != start) refine_long(v, seq, plane, flank_start, end, m, type, prm, out, depth + 1);
    }
}

}  // namespace

extern "C" {

/* fasta_utils.cpp:187-242 + processSeedMotifWise / processSeed: BED text of one record.  Returns a pointer to
 * an internal buffer (valid until the next call) and its length.  Needs rbo_run_dispatch(). */
const char *rbo_refine_bed(rbo_ctx *c, const rbo_refine_params_t *prm, const char *seq, const char *seq_id, int64_t *len) {
    static std::string text;
    BedOut out;
    out.id = seq_id;
    const rbo_seed_t *seeds;
    const int64_t n = rbo_dispatch(c, &seeds);
    View v{rbo_codes(c), rbo_nmask(c), (int)rbo_length(c)};
    for (int64_t si = 0; si < n; si++) {
        const rbo_seed_t &s = seeds[si];
        const Plane plane{rbo_plane_bits(c, s.mlen)};
        if (s.mlen <= 10) refine_small(v, seq, plane, s.start, s.end, s.mlen, s.type, *prm, out);
        else refine_long(v, seq, plane, s.start, s.end, s.mlen, s.type, *prm, out, 0);
    }
    text = out.os.str();
    *len = (int64_t)text.size();
    return text.c_str();
}

}  // extern "C"
