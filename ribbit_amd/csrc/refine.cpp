#include "refine.h"

#include "ssw_exact.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <immintrin.h>
#include <sstream>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace rb {

namespace {

// 256-bit unsigned with wrap-around, the arithmetic the reference gets from an unchecked
// boost::multiprecision::uint256_t (motifs longer than 128 bases lose their head)
struct Wide {
    uint64_t limb[4] = {0, 0, 0, 0};
    void push_base(unsigned code) {           // unit = (unit << 2) | code
        limb[3] = (limb[3] << 2) | (limb[2] >> 62);
        limb[2] = (limb[2] << 2) | (limb[1] >> 62);
        limb[1] = (limb[1] << 2) | (limb[0] >> 62);
        limb[0] = (limb[0] << 2) | code;
    }
    unsigned base_at(int index_from_low) const {       // 2-bit field number index_from_low
        const int bit = 2 * index_from_low;
        return (bit < 0 || bit >= 256) ? 0u : (unsigned)(limb[bit >> 6] >> (bit & 63)) & 3u;
    }
};

// one byte per base, decoded once per record from the planes the GPU packed (HostPlanes::symbols): 0..3 = A C G T, 4 = N
struct Bases {
    int L;
    std::shared_ptr<const std::vector<uint8_t>> hold;
    const uint8_t *sym;
    const uint32_t *brk;      // the N mask as a bit plane (32 positions a word; bits past L are set: padding)
    explicit Bases(const HostPlanes &hp, unsigned threads = 0) : L((int)hp.length), hold(hp.symbols(threads)), sym(hold->data()), brk(hp.brk.data()) {}
    // N encodes as 00 (fasta_utils.cpp:111-113).  D4: a position below 0 -- reachable only after a motif window that
    // starts before the record, where the reference has already terminated in substr -- reads as base A, not N.
    unsigned code(int p) const { return p < 0 ? 0u : sym[(size_t)p] & 3u; }
    bool is_n(int p) const { return p >= 0 && p < L && sym[(size_t)p] == 4; }
};

// calculateRepeatClass (bitseq_utils.cpp:185-221): lexicographically smallest rotation of an m-base word
uint32_t smallest_rotation(uint32_t word, int m) {
    const uint32_t mask = m >= 16 ? 0xffffffffu : (1u << (2 * m)) - 1u;
    uint32_t best = word, rot = word;
    for (int i = 1; i < m; ++i) {
        rot = ((rot << 2) | (rot >> (2 * (m - 1)))) & mask;   // rotate left by one base
        best = std::min(best, rot);
    }
    return best;
}

// calculateAtomicity(uint32_t&) (bitseq_utils.cpp:139-183): smallest proper divisor period, else m
int small_atomicity(uint32_t word, int m) {
    for (int f = 1; 2 * f <= m; ++f) {
        if (m % f) continue;
        const uint32_t keep = 2 * (m - f) >= 32 ? 0xffffffffu : (1u << (2 * (m - f))) - 1u;
        if ((word >> (2 * f)) == (word & keep)) return f;
    }
    return m;
}

// calculateAtomicityLongMotif (bitseq_utils.cpp:116-137): smallest shift f < m - m/3 with
// unit >> 2f == low 2(m-f) bits of unit, i.e. base i == base i+f for every i with both inside the
// low m bases ... and every higher 2-bit field of `unit` (beyond base m-1) must be zero.  For m > 128 the
// reference's 256-bit integer has dropped the leading bases of the unit and the test degenerates accordingly
// (e.g. every f >= 128 compares zero with the masked low fields); the same arithmetic is reproduced here.
int long_atomicity(const Wide &unit, int m) {
    // uint256_t arithmetic of the reference, field by field: (unit >> 2f) has base i+f in field i and zeros from
    // field 128-f on (all zeros once f >= 128); (mask & unit) keeps the fields below m-f (all 128 once m-f >= 128)
    for (int f = 1; f < m - m / 3; ++f) {
        bool same = true;
        for (int i = 0; i < 128 && same; ++i) {
            const unsigned shifted = i + f < 128 ? unit.base_at(i + f) : 0u;
            const unsigned masked = i < m - f ? unit.base_at(i) : 0u;
            same = shifted == masked;
        }
        if (same) return f;
    }
    return m;
}

std::string spell(const Wide &unit, int m, int take) {       // calculateMotif(...).substr(0, take)
    std::string s;
    for (int i = 0; i < take; ++i) s.push_back("ACGT"[unit.base_at(m - 1 - i)]);
    return s;
}

// `int x = a + m + ((1 - PURITY_THRESHOLD) * b);` with float PURITY_THRESHOLD: float math, truncation
int padded_length(int a, int m, int b, float purity) { return (int)((float)(a + m) + (1 - purity) * (float)b); }

// the seed's sequence length: seed + one motif, cut at the first N (parse_seed.cpp:342-349)
// (The first N at or after `start`, below end + m, from the N mask's words: a walk over the bases, 45 of them for the average
// seed, was done for every dispatched seed by the job set-up, again by the workers and once more before the consensus-row
// scan.  Positions below 0 and from L on are never N for this test, and the plane's padding bits past L are set.)
int first_n_cut(const uint32_t *brk, int L, int start, int end, int m) {
    const int stop = std::min(end + m, L);
    for (int s = std::max(start, 0); s < stop;) {
        const uint32_t bits = brk[(size_t)(s >> 5)] >> (s & 31);
        if (bits) {
            const int p = s + __builtin_ctz(bits);
            return p < stop ? p - start : (end - start) + m;
        }
        s = ((s >> 5) + 1) << 5;
    }
    return (end - start) + m;
}
int usable_length(const Bases &b, int start, int end, int m) { return first_n_cut(b.brk, b.L, start, end, m); }

struct Tracked { int first, last_end, units, anchor; uint32_t expect; };

// possibleMotifs (parse_smallmotif_seed.cpp:76-188).  Per rotation class of the rolling m-base window
// the reference keeps (start, end, units) in global arrays and the latest unit start in an
// unordered_map; a class that reappears more than 3m past its end is reported (if long enough) and
// restarted; survivors are reported in the map's iteration order (Q10).
void discover_small_motifs(const Bases &b, int seed_start, int seq_len, int m, int min_len, int min_units,
                           std::vector<uint32_t> &classes, std::vector<int> &starts, std::vector<int> &ends) {
    std::unordered_map<uint32_t, int> unit_start;          // new_motif_start: drives the output order
    std::unordered_map<uint32_t, Tracked> track;           // MOTIF_START / MOTIF_END / MOTIF_UNITS / MOTIF_NEXT
    const int stop = std::min(seed_start + seq_len, b.L - 1);
    const uint32_t mask = m >= 16 ? 0xffffffffu : (1u << (2 * m)) - 1u;
    uint32_t window = 0;
    auto qualifies = [&](const Tracked &t) { return t.last_end - t.first >= min_len && t.units >= min_units; };
    for (int j = seed_start; j < stop; ++j) {
        window = ((window << 2) | b.code(j)) & mask;
        if (!(j - seed_start >= 0.9 * m - 1)) continue;
        const uint32_t cls = smallest_rotation(window, m);
        const int wstart = j - (m - 1), wend = j + 1;
        const uint32_t next = ((window << 2) | (window >> (2 * (m - 1)))) & mask;
        auto it = unit_start.find(cls);
        if (it == unit_start.end()) {
            unit_start[cls] = wstart;
            track[cls] = Tracked{wstart, wend, 1, wstart, next};
            continue;
        }
        Tracked &t = track[cls];
        if (wstart - t.last_end > 3 * m) {
            if (qualifies(t)) { classes.push_back(cls); starts.push_back(t.first); ends.push_back(t.last_end); }
            t = Tracked{wstart, wend, 1, wstart, next};
            it->second = wstart;
            continue;
        }
        if (wstart - it->second >= m) { it->second = wstart; t.units += 1; }
        t.last_end = wend;
        t.expect = next;
    }
    for (const auto &kv : unit_start) {
        const Tracked &t = track[kv.first];
        if (qualifies(t)) { classes.push_back(kv.first); starts.push_back(t.first); ends.push_back(t.last_end); }
    }
}

// The same list from the GPU's records of seed i (small_motifs.hip): the early reports as they are, then the classes
// that survive to the seed's end in the iteration order of the reference's unordered_map, which the keys' insertion
// order (= order of first appearance = record order) determines.  False: the seed has no device result.
// The records of seed i lie wherever the kernel's wavefronts happened to append them (an arena filled through one atomic
// counter): a cache miss per seed for a loop that walks the seeds in order, 13.6 M times a chromosome in the job set-up and
// again in the workers.  The head table IS in seed order, so the records of a seed a few places ahead can be asked for early.
inline void prefetch_small_records(const SmallMotifTable *table, size_t i, size_t n) {
    if (!table || !table->head || i >= n) return;
    const int32_t *hd = table->head + 4 * i;
    if (hd[3] == 0) __builtin_prefetch(table->records + 4 * (size_t)(uint32_t)hd[0], 0, 1);
}

bool small_motifs_from_table(const SmallMotifTable *table, size_t i, int min_len, int min_units,
                             std::vector<uint32_t> &classes, std::vector<int> &starts, std::vector<int> &ends) {
    if (!table || !table->head) return false;
    const int32_t *hd = table->head + 4 * i;
    if (hd[3] != 0) return false;
    const uint32_t *rec = table->records + 4 * (size_t)(uint32_t)hd[0];
    for (int k = 0; k < hd[1]; ++k, rec += 4) { classes.push_back(rec[0]); starts.push_back((int)rec[1]); ends.push_back((int)rec[2]); }
    auto qualifies = [&](const uint32_t *r) { return (int)r[2] - (int)r[1] >= min_len && (int)r[3] >= min_units; };
    auto report = [&](const uint32_t *r) { classes.push_back(r[0]); starts.push_back((int)r[1]); ends.push_back((int)r[2]); };
    // the order only matters between survivors that are reported: with fewer than two of them no map is needed
    int n_reported = 0, only = -1;
    for (int k = 0; k < hd[2]; ++k)
        if (qualifies(rec + 4 * k)) { ++n_reported; only = k; }
    if (n_reported == 0) return true;
    if (n_reported == 1) { report(rec + 4 * only); return true; }
    std::unordered_map<uint32_t, int> order;
    for (int k = 0; k < hd[2]; ++k) order[rec[4 * k]] = k;
    for (const auto &kv : order)
        if (qualifies(rec + 4 * kv.second)) report(rec + 4 * kv.second);
    return true;
}

// mostFrequentLongerMotif (parse_seed.cpp:153-256): every window row_start..row_start+m-1 of the seed is
// scored by walking down- and upstream in steps of m with a +-2 jitter, counting identical bases on the
// best-matching diagonal; the best row's bases form the motif.
Wide unit_at(const Bases &b, int row, int m) {                           // parse_seed.cpp:246-253
    Wide unit;
    for (int j = row; j < row + m; ++j) unit.push_base(b.code(j));
    return unit;
}

// The five jittered diagonals of one step of consensus_row, 64 / 32 symbols at a time where the host has AVX-512BW / AVX2
// (chosen once at run time; the plain 8-at-a-time form below is the fallback and the definition): at -M 500 this loop was
// 45 % of refinement's host time on a stream of reads (1209 of 2689 thread-seconds per Gbp).
// cnt[x]: symbols diagonal x compares (0: none), most: the largest of them.
__attribute__((target("avx512bw,avx512vl"))) void diagonals5_avx512(const uint8_t *sym, int row0, int col0, const int (&cnt)[5], int most, int (&matches)[5]) {
    const __m512i four = _mm512_set1_epi8(4);
    for (int i = 0; i < most; i += 64) {
        const __m512i rb = _mm512_loadu_si512((const void *)(sym + row0 + i));
        for (int x = 0; x < 5; ++x) {
            const int left = cnt[x] - i;
            if (left <= 0) continue;
            const __m512i a = _mm512_loadu_si512((const void *)(sym + (col0 + x - 2 + i)));
            uint64_t eq = _mm512_cmpeq_epi8_mask(a, rb) & _mm512_cmplt_epu8_mask(a, four);
            if (left < 64) eq &= (1ull << left) - 1ull;
            matches[x] += __builtin_popcountll(eq);
        }
    }
}
__attribute__((target("avx2"))) void diagonals5_avx2(const uint8_t *sym, int row0, int col0, const int (&cnt)[5], int most, int (&matches)[5]) {
    const __m256i four = _mm256_set1_epi8(4);
    for (int i = 0; i < most; i += 32) {
        const __m256i rb = _mm256_loadu_si256((const __m256i *)(sym + row0 + i));
        for (int x = 0; x < 5; ++x) {
            const int left = cnt[x] - i;
            if (left <= 0) continue;
            const __m256i a = _mm256_loadu_si256((const __m256i *)(sym + (col0 + x - 2 + i)));
            const __m256i ok = _mm256_and_si256(_mm256_cmpeq_epi8(a, rb), _mm256_cmpgt_epi8(four, a));      // symbols are 0..4: signed compare is fine
            uint32_t eq = (uint32_t)_mm256_movemask_epi8(ok);
            if (left < 32) eq &= (1u << left) - 1u;
            matches[x] += __builtin_popcount(eq);
        }
    }
}
int simd_level() {          // RIBBIT_HOST_SIMD=0/1/2 caps it (the tests compare the three forms)
    static const int level = [] {
        int have = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") ? 2 : __builtin_cpu_supports("avx2") ? 1 : 0;
        if (const char *env = std::getenv("RIBBIT_HOST_SIMD")) have = std::min(have, std::max(0, std::atoi(env)));
        return have;
    }();
    return level;
}

int consensus_row(const Bases &b, int seed_start, int seq_len, int m, int simd = -1) {
    if (simd < 0) simd = simd_level();
    const int seed_end = std::min(seed_start + seq_len, b.L);
    auto diagonal = [&](int row0, int col0, int lo, int hi, int n, int step) {
        // matches between rows row0, row0+step, ... and columns col0, col0+step, ...; stops at the first
        // column outside [lo, hi)
        int d = 0;
        for (int i = 0; i < n; ++i) {
            const int col = col0 + step * i, row = row0 + step * i;
            if (col >= hi || col < lo) break;
            d += b.sym[(size_t)col] == b.sym[(size_t)row] && b.sym[(size_t)col] < 4;
        }
        return d;
    };
    // The five jittered forward diagonals of one step, eight symbols at a time (as long_motif_rows_kernel does on the GPU: a
    // symbol is 0..4, so the bytes of (a ^ b) + 0x7f..7f have bit 7 clear exactly where the symbols are equal, and a column
    // symbol is a base where its bit 2 is clear).  The walk of `diagonal` ends at the first column outside [lo, hi): for an
    // increasing column, none if the first is below lo, else the first hi - col0.  Loads run up to 7 bytes past what is counted:
    // sym is padded by 16 entries (host_planes.cpp).  Flank seeds get their row here: 29 K of them per chromosome and pass.
    auto diagonals5 = [&](int row0, int col0, int lo, int hi, int n, int (&matches)[5]) {
        int cnt[5], most = 0;
        for (int x = 0; x < 5; ++x) {
            const int c0 = col0 + x - 2;
            cnt[x] = c0 < lo ? 0 : std::max(0, std::min(n, hi - c0));
            most = std::max(most, cnt[x]);
            matches[x] = 0;
        }
        if (simd == 2) { diagonals5_avx512(b.sym, row0, col0, cnt, most, matches); return; }
        if (simd == 1) { diagonals5_avx2(b.sym, row0, col0, cnt, most, matches); return; }
        for (int i = 0; i < most; i += 8) {
            uint64_t rb8;
            std::memcpy(&rb8, b.sym + row0 + i, 8);
            for (int x = 0; x < 5; ++x) {
                const int left = cnt[x] - i;
                if (left <= 0) continue;
                uint64_t a;
                std::memcpy(&a, b.sym + (col0 + x - 2 + i), 8);
                uint64_t eq = ~((a ^ rb8) + 0x7f7f7f7f7f7f7f7full) & ~(a << 5) & 0x8080808080808080ull;
                if (left < 8) eq &= (1ull << (8 * left)) - 1ull;
                matches[x] += __builtin_popcountll(eq);
            }
        }
    };
    int best_row = 0, best_score = 0;
    int d5[5];
    for (int row = seed_start; row < seed_end - m + 1; ++row) {
        int score = 0;
        for (int col = row + m; col < seed_end;) {                    // downstream copies (:181-198)
            int pick = -2, top = 0;
            diagonals5(row, col, INT32_MIN, seed_end, m, d5);
            for (int x = -2; x <= 2; ++x)
                if (d5[x + 2] > top) { top = d5[x + 2]; pick = x; }
            score += top;
            col += pick + m;
        }
        int col = row - m;
        for (; col > seed_start;) {                                   // upstream copies (:200-217)
            int pick = -2, top = 0;
            diagonals5(row, col, 0, INT32_MAX, m, d5);
            for (int x = -2; x <= 2; ++x)
                if (d5[x + 2] > top) { top = d5[x + 2]; pick = x; }
            score += top;
            col += pick - m;
        }
        if (col < seed_start && std::abs(col - seed_start) < m) {     // partial copy at the seed's head (:219-237)
            const int rows = m + (col - seed_start);
            int top = 0;
            for (int x = -2; x <= 2; ++x)
                top = std::max(top, diagonal(row + m - 1, seed_start + rows - 1 + x, seed_start, seed_end, rows, -1));
            score += top;
        }
        if (score > best_score) { best_score = score; best_row = row; }
    }
    return best_row;
}

}  // namespace

int longest_run_host(const HostPlanes &hp, int mlen, int start, int end) {
    static thread_local std::vector<uint32_t> w;
    hp.xa_slice(mlen, start, end, w);
    const int base = (start >> 5) << 5;
    int best = 0, run = 0;
    for (int p = start; p < end; ++p) {
        if ((w[(size_t)((p - base) >> 5)] >> (p & 31)) & 1u) { ++run; best = std::max(best, run); }
        else run = 0;
    }
    return best;
}

int usable_length_host(const HostPlanes &hp, int start, int end, int m) { return first_n_cut(hp.brk.data(), (int)hp.length, start, end, m); }

namespace {
thread_local double tl_build_join_ms = 0.0;      // profile: joining the chunks' jobs and pools
thread_local double tl_build_par_ms = 0.0;       // profile: the chunks' parallel region (threads started to threads joined)
void build_align_jobs_range(const Bases &b, const RibbitRefineParams &prm, const SeedVec &dispatch,
                            const int32_t *longest_runs, const int32_t *best_rows, size_t lo, size_t hi,
                            std::vector<RibbitAlignJob> &jobs, std::string &motif_pool, const SmallMotifTable *small) {
    std::vector<uint32_t> classes;
    std::vector<int> starts, ends;
    for (size_t i = lo; i < hi; ++i) {
        prefetch_small_records(small, i + 12, dispatch.size());
        const RibbitSeed &seed = dispatch[i];
        const int m = seed.mlen;
        if (m > 10 && seed.end - seed.start < 0.9 * m) continue;                    // parse_seed.cpp:360
        if (longest_runs[i] < prm.continuous_ones_threshold) continue;              // :366-367 / smallmotif :234-235
        const int seq_len = usable_length(b, seed.start, seed.end, m);
        RibbitAlignJob job{};
        job.seed_index = (int32_t)i;
        job.seed_type = seed.type;
        job.motif_length = m;
        if (m <= 10) {
            classes.clear(); starts.clear(); ends.clear();
            if (!small_motifs_from_table(small, i, prm.min_length[m], prm.perfect_units[m], classes, starts, ends))
                discover_small_motifs(b, seed.start, seq_len, m, prm.min_length[m], prm.perfect_units[m], classes, starts, ends);
            for (size_t k = 0; k < classes.size(); ++k) {
                Wide unit; unit.limb[0] = classes[k];
                job.atomicity = small_atomicity(classes[k], m);
                job.query_start = starts[k];
                job.query_length = ends[k] - starts[k];
                job.ppr_length = padded_length(job.query_length, m, job.query_length, prm.purity_threshold);
                job.small = 1;
                job.motif_offset = (int32_t)motif_pool.size();
                motif_pool += spell(unit, m, job.atomicity);
                jobs.push_back(job);
            }
        } else {
            const int row = (best_rows && best_rows[i] >= 0) ? best_rows[i] : consensus_row(b, seed.start, seq_len, m);
            const Wide unit = unit_at(b, row, m);
            job.atomicity = long_atomicity(unit, m);
            if (m % job.atomicity != 0) continue;                                    // parse_seed.cpp:392
            job.query_start = seed.start;
            job.query_length = std::min(seq_len, b.L - seed.start);
            job.ppr_length = padded_length(seq_len, m, seq_len, prm.purity_threshold);
            job.small = 0;
            job.motif_offset = (int32_t)motif_pool.size();
            motif_pool += spell(unit, m, job.atomicity);
            jobs.push_back(job);
        }
    }
}
}  // namespace

void build_align_jobs(const HostPlanes &hp, const RibbitRefineParams &prm, const SeedVec &dispatch,
                      const int32_t *longest_runs, const int32_t *best_rows, std::vector<RibbitAlignJob> &jobs,
                      std::string &motif_pool, unsigned host_threads, size_t seed_lo, size_t seed_hi, const SmallMotifTable *small) {
    jobs.clear();
    motif_pool.clear();
    const Bases b(hp, host_threads);
    seed_hi = std::min(seed_hi, dispatch.size());
    seed_lo = std::min(seed_lo, seed_hi);
    const size_t n = seed_hi - seed_lo;
    const unsigned threads = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads ? host_threads : 1, n / 1024 + 1));
    if (threads == 1) { build_align_jobs_range(b, prm, dispatch, longest_runs, best_rows, seed_lo, seed_hi, jobs, motif_pool, small); return; }
    // seeds are independent here: chunks on worker threads, concatenated in seed order with the motif offsets rebased
    const auto tp0 = std::chrono::steady_clock::now();
    const size_t chunk = 2048, nchunks = (n + chunk - 1) / chunk;
    std::vector<std::vector<RibbitAlignJob>> part_jobs(nchunks);
    std::vector<std::string> part_pool(nchunks);
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&]() {
            for (size_t c; (c = next.fetch_add(1)) < nchunks;)
                build_align_jobs_range(b, prm, dispatch, longest_runs, best_rows, seed_lo + c * chunk, seed_lo + std::min(n, (c + 1) * chunk), part_jobs[c], part_pool[c], small);
        });
    for (std::thread &th : pool) th.join();
    // the chunks' jobs and motif strings into place, on the threads again (one thread did this for a fifth of the set-up's time)
    const auto tj0 = std::chrono::steady_clock::now();
    tl_build_par_ms += std::chrono::duration<double, std::milli>(tj0 - tp0).count();
    std::vector<size_t> job_at(nchunks + 1, 0), pool_at(nchunks + 1, 0);
    for (size_t c = 0; c < nchunks; ++c) { job_at[c + 1] = job_at[c] + part_jobs[c].size(); pool_at[c + 1] = pool_at[c] + part_pool[c].size(); }
    jobs.resize(job_at[nchunks]);
    motif_pool.resize(pool_at[nchunks]);
    next = 0;
    auto place = [&]() {
        for (size_t c; (c = next.fetch_add(1)) < nchunks;) {
            RibbitAlignJob *out = jobs.data() + job_at[c];
            const int32_t base = (int32_t)pool_at[c];
            for (size_t k = 0; k < part_jobs[c].size(); ++k) { out[k] = part_jobs[c][k]; out[k].motif_offset += base; }
            if (!part_pool[c].empty()) std::memcpy(&motif_pool[pool_at[c]], part_pool[c].data(), part_pool[c].size());
        }
    };
    pool.clear();
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(place);
    place();
    for (std::thread &th : pool) th.join();
    tl_build_join_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj0).count();
}

// The jobs of several slices of the seed list in ONE parallel region: the chunks of all slices are handed out in order to the
// same threads, and the thread that finishes a slice's last chunk puts that slice's jobs and motif strings in place and hands
// them to `done` (called on that thread; slices are done roughly, not strictly, in order) while the others go on with the next
// slice's chunks.  Slice by slice through build_align_jobs this was two teams of threads and two barriers per slice: 512
// thread starts and a third of the set-up's time outside its parallel work at chromosome-1 size.
void build_align_jobs_slices(const HostPlanes &hp, const RibbitRefineParams &prm, const SeedVec &dispatch, const int32_t *longest_runs,
                             const int32_t *best_rows, unsigned host_threads, const SmallMotifTable *small,
                             const std::vector<std::pair<size_t, size_t>> &bounds,
                             const std::function<void(size_t, std::vector<RibbitAlignJob> &&, std::string &&)> &done) {
    const Bases b(hp, host_threads);
    constexpr size_t chunk = 2048;
    struct Slice {
        size_t lo = 0, hi = 0, first_chunk = 0, nchunks = 0;
        std::vector<std::vector<RibbitAlignJob>> jobs;
        std::vector<std::string> pool;
        std::atomic<size_t> left{0};
    };
    std::vector<Slice> sl(bounds.size());
    size_t total = 0;
    for (size_t c = 0; c < bounds.size(); ++c) {
        sl[c].lo = std::min(bounds[c].first, dispatch.size());
        sl[c].hi = std::max(sl[c].lo, std::min(bounds[c].second, dispatch.size()));
        sl[c].first_chunk = total;
        sl[c].nchunks = (sl[c].hi - sl[c].lo + chunk - 1) / chunk;
        sl[c].jobs.resize(sl[c].nchunks);
        sl[c].pool.resize(sl[c].nchunks);
        sl[c].left = sl[c].nchunks;
        total += sl[c].nchunks;
    }
    auto finish = [&](size_t c) {
        Slice &s = sl[c];
        std::vector<RibbitAlignJob> jobs;
        std::string pool;
        size_t nj = 0, np = 0;
        for (size_t k = 0; k < s.nchunks; ++k) { nj += s.jobs[k].size(); np += s.pool[k].size(); }
        jobs.reserve(nj);
        pool.reserve(np);
        for (size_t k = 0; k < s.nchunks; ++k) {
            const int32_t base = (int32_t)pool.size();
            for (RibbitAlignJob j : s.jobs[k]) { j.motif_offset += base; jobs.push_back(j); }
            pool += s.pool[k];
            std::vector<RibbitAlignJob>().swap(s.jobs[k]);
            std::string().swap(s.pool[k]);
        }
        done(c, std::move(jobs), std::move(pool));
    };
    for (size_t c = 0; c < sl.size(); ++c)
        if (sl[c].nchunks == 0) finish(c);            // an empty slice is done at once
    std::atomic<size_t> next{0};
    auto work = [&]() {
        size_t c = 0;
        for (size_t id; (id = next.fetch_add(1)) < total;) {
            while (c + 1 < sl.size() && id >= sl[c].first_chunk + sl[c].nchunks) ++c;      // ids only grow for one thread
            Slice &s = sl[c];
            const size_t k = id - s.first_chunk;
            build_align_jobs_range(b, prm, dispatch, longest_runs, best_rows, s.lo + k * chunk, std::min(s.hi, s.lo + (k + 1) * chunk), s.jobs[k], s.pool[k], small);
            if (s.left.fetch_sub(1) == 1) finish(c);
        }
    };
    const unsigned threads = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads ? host_threads : 1, total));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (std::thread &th : pool) th.join();
}

double build_align_jobs_join_ms(bool reset) { const double v = tl_build_join_ms; if (reset) tl_build_join_ms = 0.0; return v; }
double build_align_jobs_parallel_ms(bool reset) { const double v = tl_build_par_ms; if (reset) tl_build_par_ms = 0.0; return v; }

void build_align_jobs_of(const HostPlanes &hp, const RibbitRefineParams &prm, const SeedVec &dispatch, const int32_t *longest_runs,
                         const int32_t *best_rows, const std::vector<uint32_t> &which, std::vector<RibbitAlignJob> &jobs, std::string &motif_pool,
                         unsigned host_threads, const SmallMotifTable *small) {
    jobs.clear();
    motif_pool.clear();
    const Bases b(hp, host_threads);
    const size_t n = which.size();
    const size_t chunk = 256, nchunks = (n + chunk - 1) / chunk;
    const unsigned threads = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads ? host_threads : 1, nchunks));
    std::vector<std::vector<RibbitAlignJob>> part_jobs(nchunks);
    std::vector<std::string> part_pool(nchunks);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t c; (c = next.fetch_add(1)) < nchunks;)
            for (size_t k = c * chunk; k < std::min(n, (c + 1) * chunk); ++k)
                build_align_jobs_range(b, prm, dispatch, longest_runs, best_rows, which[k], (size_t)which[k] + 1, part_jobs[c], part_pool[c], small);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (std::thread &th : pool) th.join();
    for (size_t c = 0; c < nchunks; ++c) {
        const int32_t base = (int32_t)motif_pool.size();
        for (RibbitAlignJob &j : part_jobs[c]) { j.motif_offset += base; jobs.push_back(j); }
        motif_pool += part_pool[c];
    }
}

// ---------------------------------------------------------------------------------------------------
// Alignment post-processing and BED rows
namespace {

// RIBBIT_PROFILE=1: wall-clock split of the refinement stages on stderr (summed over the worker threads).
// Counters and times are gathered per thread and added to the process-wide totals once per chunk of seeds: a shared
// atomic bumped for every seed by sixteen threads is a cache line in permanent transit.
std::atomic<long long> g_t_align{0}, g_t_small{0}, g_t_long{0};       // nanoseconds
std::atomic<long> g_n_align{0}, g_n_known{0}, g_n_paths{0}, g_n_small_device{0}, g_n_small_host{0};
std::atomic<long> g_n_flank{0};
std::atomic<long long> g_t_flank{0}, g_t_whole_first{0};      // profile: flank-recursion alignments; whole first-level alignments on the host
std::atomic<long long> g_t_digest{0}, g_t_units{0}, g_t_row{0}, g_t_query{0}, g_t_atom{0}, g_t_small_all{0}, g_t_long_all{0}, g_t_range{0};
const bool g_profile = std::getenv("RIBBIT_PROFILE") != nullptr;
struct LocalCounters {
    long n_align = 0, n_known = 0, n_paths = 0, n_small_device = 0, n_small_host = 0, n_flank = 0;
    long long t_align = 0, t_small = 0, t_long = 0, t_flank = 0, t_whole_first = 0, t_digest = 0, t_units = 0, t_row = 0, t_query = 0, t_atom = 0,
              t_small_all = 0, t_long_all = 0, t_range = 0;
};
thread_local LocalCounters tl;
void flush_counters() {
    g_n_align += tl.n_align; g_n_known += tl.n_known; g_n_paths += tl.n_paths; g_n_small_device += tl.n_small_device;
    g_n_small_host += tl.n_small_host; g_n_flank += tl.n_flank;
    g_t_align += tl.t_align; g_t_small += tl.t_small; g_t_long += tl.t_long; g_t_flank += tl.t_flank; g_t_whole_first += tl.t_whole_first;
    g_t_digest += tl.t_digest; g_t_units += tl.t_units; g_t_row += tl.t_row; g_t_query += tl.t_query; g_t_atom += tl.t_atom;
    g_t_small_all += tl.t_small_all; g_t_long_all += tl.t_long_all; g_t_range += tl.t_range;
    tl = LocalCounters{};
}
struct Stopwatch {
    long long *acc;
    std::chrono::steady_clock::time_point t0;
    explicit Stopwatch(long long *a) : acc(g_profile ? a : nullptr) { if (acc) t0 = std::chrono::steady_clock::now(); }
    ~Stopwatch() { if (acc) *acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

struct CigarOp { int len; char op; };

// text += decimal(n) + op, without the temporary std::to_string makes
inline void put_op(std::string &text, int n, char op) {
    char buf[12];
    int at = 12;
    unsigned v = n < 0 ? 0u - (unsigned)n : (unsigned)n;
    do { buf[--at] = (char)('0' + v % 10u); v /= 10u; } while (v);
    if (n < 0) buf[--at] = '-';
    text.append(buf + at, (size_t)(12 - at));
    text.push_back(op);
}

void parse_cigar(const std::string &text, std::vector<CigarOp> &ops) {        // cigarSplit, process_cigar.cpp:14-31
    ops.clear();
    int n = 0;
    for (char ch : text) {
        if (ch >= '0' && ch <= '9') n = n * 10 + (ch - '0');
        else { ops.push_back({n, ch}); n = 0; }
    }
}

struct Repeat { int start, end, alignment_length, match_units; float purity; std::string cigar; };

// processCIGARMotifWise (process_cigar.cpp:254-336); with prune == true processCIGARWithPruning
// (:126-251) incl. calculateTrimEdges (:34-86).  The alignment is compressed into alternating blocks
// match, non-match, match, ...; trimming drops whole block pairs from either end until the purity
// reaches the threshold (keeping, per trim depth, the longest combination that reaches it).
Repeat digest_cigar(int seed_start, int seq_len, const std::string &text, int unit_len, bool prune, const RibbitRefineParams &prm) {
    // scratch that lives as long as the thread: millions of alignments per record, each a handful of operations
    static thread_local std::vector<CigarOp> ops;
    static thread_local std::vector<int> block_of;      // compressed-block index of every aligned (non-S) op
    static thread_local std::vector<int> block_len;     // lengths of the alternating blocks
    parse_cigar(text, ops);
    block_of.clear(); block_len.clear();
    Repeat r{seed_start, seed_start + seq_len, 0, 0, 0.f, std::string()};
    r.cigar.reserve(text.size() + 4);
    int matches = 0, lead_clip = 0;
    bool in_mismatch = false;
    for (size_t i = 0; i < ops.size(); ++i) {
        const CigarOp &o = ops[i];
        if (o.op == 'S') { if (i == 0) { r.start += o.len; lead_clip = o.len; } else r.end -= o.len; continue; }
        if (o.op == '=' || o.op == 'M') {
            matches += o.len; r.match_units += o.len / unit_len;
            block_len.push_back(o.len); in_mismatch = false;
        } else if (o.op == 'X' || o.op == 'I' || o.op == 'D') {
            if (in_mismatch) block_len.back() += o.len; else block_len.push_back(o.len);
            in_mismatch = true;
        } else continue;
        r.alignment_length += o.len;
        block_of.push_back((int)block_len.size() - 1);
        put_op(r.cigar, o.len, o.op);
    }
    r.purity = float(matches) / float(r.alignment_length);
    if (!prune || !(r.purity < prm.purity_threshold)) return r;

    // calculateTrimEdges: left/right counts of (match, mismatch) block pairs to drop
    const long nb = (long)block_len.size();
    int drop_left = 0, drop_right = 0;
    for (int depth = 1; r.purity < prm.purity_threshold; ++depth) {
        float best_purity = 0; int best_len = 0;
        for (int left = 0; left <= depth; ++left) {
            int m_sum = 0, a_sum = 0;
            for (long j = 2L * left; j <= (nb - 1) - 2L * (depth - left); ++j) {
                if ((j & 1) == 0) m_sum += block_len[(size_t)j];
                a_sum += block_len[(size_t)j];
            }
            const float p = float(m_sum) / float(a_sum);
            if (p >= prm.purity_threshold && best_len < a_sum) { best_purity = p; best_len = a_sum; drop_left = left; drop_right = depth - left; }
        }
        if (best_purity > r.purity) { r.purity = best_purity; r.alignment_length = best_len; }
        if (r.alignment_length < prm.min_length[unit_len]) break;
        if (2L * depth > nb + 2) break;       // D5: the reference's unsigned loop bound would wrap here
    }
    r.cigar.clear(); r.match_units = 0;
    const size_t first_aligned = lead_clip ? 1 : 0;      // ops[] index of the first aligned op (:216-217)
    for (size_t i = 0; i < block_of.size(); ++i) {
        const CigarOp &o = ops[first_aligned + i];
        const long b = block_of[i];
        if (b < 2L * drop_left) { if (o.op != 'D') r.start += o.len; }
        else if (b <= nb - 1 - 2L * drop_right) {
            put_op(r.cigar, o.len, o.op);
            if (o.op == 'M' || o.op == '=') r.match_units += o.len / unit_len;
        } else if (o.op != 'D') r.end -= o.len;
    }
    return r;
}

// calculateMotifUnits (parse_smallmotif_seed.cpp:26-72): non-overlapping exact occurrences of the unit's
// rotation class in [start, start+length)
int count_units(const Bases &b, int start, int length, int m, uint32_t unit) {
    const int stop = std::min(start + length, b.L - 1);
    const uint32_t mask = m >= 16 ? 0xffffffffu : (1u << (2 * m)) - 1u;
    // the reference keeps (last unit start, units) per rotation class in a map and reads the entry of `unit` at the end;
    // the classes do not interact, so only that one entry is kept here
    int last_at = 0, units = 0;
    uint32_t window = 0;
    for (int j = start; j < stop; ++j) {
        window = ((window << 2) | b.code(j)) & mask;
        if (!(j - start >= 0.9 * m - 1)) continue;
        if (smallest_rotation(window, m) != unit) continue;
        const int at = j - (m - 1);
        if (units == 0) { last_at = at; units = 1; }
        else if (at - last_at >= m) { last_at = at; units += 1; }
    }
    return units;
}

struct Writer {
    const Bases &b;
    const HostPlanes &hp;
    const char *sequence;
    const RibbitRefineParams &prm;
    const std::string &id;
    // the rows, as text.  A plain string with the three calls of a string stream that the callers use: a row through an
    // ostringstream (locale, sentry, virtual calls per field) was a seventh of the workers' time -- 2.6 M rows a chromosome
    struct Text {
        std::string s;
        size_t tellp() const { return s.size(); }
        std::string str() const { return s; }
        void str(std::string v) { s = std::move(v); }
    } os;
    std::string last_cigar;     // the Alignment object lives across seeds (fasta_utils.cpp:177): an empty query leaves it untouched
    Writer(const Bases &b_, const HostPlanes &hp_, const char *sequence_, const RibbitRefineParams &prm_, const std::string &id_)
        : b(b_), hp(hp_), sequence(sequence_), prm(prm_), id(id_) {}

    struct Span { const char *p; int n; };
    Span slice(int start, int len) const {                      // sequence.substr(start, len), in place; D4: negative start clamps
        if (start < 0) { len += start; start = 0; }
        if (start >= b.L || len <= 0) return Span{sequence, 0};
        return Span{sequence + start, std::min(len, b.L - start)};
    }
    bool saw_empty_query = false;   // the one order dependence between seeds: see refine_to_bed
    // ---- pieces and the recursion tree (refine.h: BedPiece, DeferredNode).  With a sink the rows go out as pieces: a piece is
    // closed wherever the next rows in printing order are not made here (a seed left out, a node put off).
    std::vector<BedPiece> *sink = nullptr;
    uint32_t seg_first = 0;         // label of the open piece
    std::string seg_path;
    const Deferral *tree = nullptr;
    uint32_t cur_root = 0;          // the seed being refined and the place of the node being refined in its tree
    std::string cur_path;
    bool deferred_in_seed = false;
    std::vector<DeferredNode> deferred;      // handed over by the caller when the writer is done
    void flush_piece(uint32_t next_first) {
        if (sink && !os.s.empty()) { sink->push_back(BedPiece{seg_first, std::move(os.s), seg_path}); os.s.clear(); }
        seg_first = next_first;
        seg_path.clear();
    }
    // first-level alignments whose striped passes the GPU has done already: the jobs of the current seed, in the
    // order in which small_seed / long_seed reach them (build_align_jobs follows the same control flow)
    const RibbitAlignJob *jobs = nullptr;
    const SswEnds *ends = nullptr;
    const SswPath *paths = nullptr;     // per job; ops == nullptr && !failed: path not found on the GPU, searched here
    size_t next_job = 0, last_job = 0;
    void begin_seed(size_t first, size_t last) { next_job = first; last_job = last; }
    std::string ref;                    // the pseudo-perfect repeat of the current alignment (capacity kept from seed to seed)
    SswResult res;
    // query_start < 0: not a first-level alignment (flank recursion), always aligned here.  The result is last_cigar.
    const std::string &align(Span query, const std::string &motif, int ppr_len, int query_start = -1) {
        const SswEnds *known = nullptr;
        const SswPath *known_path = nullptr;
        if (query_start >= 0 && jobs && next_job < last_job) {
            const RibbitAlignJob &jb = jobs[next_job];
            const SswEnds &e = ends[next_job];
            if (e.flag != -1 && jb.query_start == query_start && jb.ppr_length == ppr_len &&
                query.n == std::min(jb.query_length, b.L - jb.query_start) && jb.atomicity == (int)motif.size()) {
                known = &e;
                if (paths && (paths[next_job].ops || paths[next_job].failed)) known_path = &paths[next_job];
            }
            ++next_job;
        }
        if (query.n == 0) { saw_empty_query = true; return last_cigar; }
        // with end points AND path known nothing reads the reference as a string (ssw_finish_with_path_periodic)
        if (!(known && known_path)) { Stopwatch swq(&tl.t_query); ref.clear(); while ((long)ref.size() <= (long)ppr_len) ref += motif; }
        {
            Stopwatch sw(&tl.t_align);
            Stopwatch sw2(known ? nullptr : (query_start < 0 ? &tl.t_flank : &tl.t_whole_first));
            if (query_start < 0) ++tl.n_flank;
            ++tl.n_align;
            if (known && known_path) { ++tl.n_known; ++tl.n_paths; ssw_finish_with_path_periodic(query.p, query.n, motif.data(), (int)motif.size(), *known, *known_path, res); }
            else if (known) { ++tl.n_known; ssw_finish(query.p, query.n, ref.data(), ppr_len, *known, res); }
            else ssw_align(query.p, query.n, ref.data(), ppr_len, 15, res);
        }
        last_cigar.swap(res.cigar);
        return last_cigar;
    }
    void row(const Repeat &r, const std::string &motif, int atom, int m, int type) {   // parse_seed.cpp:434-436
        Stopwatch sw(&tl.t_row);
        // the first row after a node of this seed was put off opens a piece of its own, labelled with the place of its node
        if (deferred_in_seed && os.s.empty()) { seg_first = cur_root; seg_path = cur_path; }
        // id \t start \t end \t motif \t atom | m \t length \t units \t purity \t + \t SEED-type \t cigar \n.  The purity is a float
        // through an ostream in the reference: "%g" at the default precision of 6, which is what num_put hands to printf
        std::string &t = os.s;
        auto num = [&](long v) {
            char buf[24];
            int at = 24;
            unsigned long u = v < 0 ? 0ul - (unsigned long)v : (unsigned long)v;
            do { buf[--at] = (char)('0' + u % 10ul); u /= 10ul; } while (u);
            if (v < 0) buf[--at] = '-';
            t.append(buf + at, (size_t)(24 - at));
        };
        t += id; t += '\t'; num(r.start); t += '\t'; num(r.end); t += '\t'; t += motif; t += '\t'; num(atom); t += " | "; num(m); t += '\t';
        num(r.end - r.start); t += '\t'; num((r.end - r.start) / atom); t += '\t';
        char pbuf[40];
        const int pn = std::snprintf(pbuf, sizeof pbuf, "%g", (double)r.purity);
        t.append(pbuf, (size_t)std::max(0, std::min(pn, (int)sizeof pbuf - 1)));
        t += "\t+\tSEED-"; num(type); t += '\t'; t += r.cigar; t += '\n';
    }

    const SmallMotifTable *small = nullptr;     // possibleMotifs of the dispatched seeds from the GPU (optional)
    std::vector<uint32_t> classes; std::vector<int> cls_starts, cls_ends;      // of the current small-motif seed
    void small_seed(const RibbitSeed &seed, int longest, size_t index) {                // processSeedMotifWise
        const int m = seed.mlen;
        if (longest < prm.continuous_ones_threshold) return;
        classes.clear(); cls_starts.clear(); cls_ends.clear();
        {
            Stopwatch sw(&tl.t_small);
            if (small_motifs_from_table(small, index, prm.min_length[m], prm.perfect_units[m], classes, cls_starts, cls_ends)) tl.n_small_device += 1;
            else { tl.n_small_host += 1; discover_small_motifs(b, seed.start, usable_length(b, seed.start, seed.end, m), m, prm.min_length[m], prm.perfect_units[m], classes, cls_starts, cls_ends); }
        }
        for (size_t k = 0; k < classes.size(); ++k) {
            const int atom = small_atomicity(classes[k], m);
            Wide unit; unit.limb[0] = classes[k];
            const std::string motif = spell(unit, m, atom);
            const int qlen = cls_ends[k] - cls_starts[k];
            const std::string &cigar = align(slice(cls_starts[k], qlen), motif, padded_length(qlen, m, qlen, prm.purity_threshold), std::max(cls_starts[k], 0));
            Repeat r;
            { Stopwatch swd(&tl.t_digest); r = digest_cigar(cls_starts[k], qlen, cigar, atom, false, prm); }
            int units;
            { Stopwatch swu(&tl.t_units); units = count_units(b, r.start, r.end - r.start, atom, classes[k] >> (2 * (m - atom))); }
            if (units >= prm.perfect_units[atom] && r.end - r.start >= prm.min_length[atom]) row(r, motif, atom, m, seed.type);
        }
    }

    // root_call: the node is the one the caller asked for (a dispatched seed, or a node put off earlier): its alignment may be
    // among the jobs the GPU has done
    void long_seed(int start, int end, int m, int type, int longest, int known_row, int depth, bool root_call) {       // processSeed
        if (depth > 10000) return;
        if (end - start < 0.9 * m) return;
        if (longest < 0) longest = longest_run_host(hp, m, start, end);
        if (longest < prm.continuous_ones_threshold) return;
        const int seq_len = usable_length(b, start, end, m);
        const int ppr_len = padded_length(seq_len, m, seq_len, prm.purity_threshold);
        // (not the node the caller asked for when it is a node put off earlier, or has its alignment among the GPU's jobs: it is due now)
        if (tree && tree->out && sink && seq_len >= tree->min_length && !(root_call && (jobs || tree->nodes)) &&
            std::min(seq_len, b.L - start) <= tree->max_query && ppr_len <= tree->max_ref && start >= 0) {
            // worth a GPU batch: not done here.  What has been printed so far is a piece of its own; the node's rows will sort
            // behind it, and whatever this seed prints afterwards behind them
            flush_piece(cur_root);
            deferred_in_seed = true;
            deferred.push_back(DeferredNode{start, end, m, type, longest, known_row, cur_root, cur_path});
            return;
        }
        Wide unit;
        { Stopwatch sw(&tl.t_long); unit = unit_at(b, known_row >= 0 ? known_row : consensus_row(b, start, seq_len, m), m); }
        int atom;
        { Stopwatch swa(&tl.t_atom); atom = long_atomicity(unit, m); }
        if (m % atom != 0) return;
        const std::string motif = spell(unit, m, atom);
        const std::string &cigar = align(slice(start, seq_len), motif, ppr_len, root_call ? std::max(start, 0) : -1);
        Repeat r;
        { Stopwatch swd(&tl.t_digest); r = digest_cigar(start, seq_len, cigar, atom, true, prm); }
        if (r.alignment_length >= prm.min_length[atom] && r.end - r.start >= prm.min_length[m]) row(r, motif, atom, m, type);
        // flanks on either side of the aligned repeat, if at least MINIMUM_LENGTH[m] long (:443-463)
        const int right_from = r.end - atom;
        if (start < r.start) {
            const int left_to = std::min(r.start, end);
            if (r.start - start >= prm.min_length[m] && !(left_to == end)) {
                cur_path.push_back('1');
                long_seed(start, left_to, m, type, -1, -1, depth + 1, false);
                cur_path.pop_back();
            }
        }
        if (end - right_from >= prm.min_length[m]) {
            const int from = std::max(right_from, start);
            if (from != start) {
                cur_path.push_back('2');
                long_seed(from, end, m, type, -1, -1, depth + 1, false);
                cur_path.pop_back();
            }
        }
    }
};

}  // namespace

namespace { thread_local bool tl_met_empty_query = false; }
bool refine_met_empty_query(bool reset) { const bool v = tl_met_empty_query; if (reset) tl_met_empty_query = false; return v; }

void refine_to_bed(const HostPlanes &hp, const char *sequence, const RibbitRefineParams &prm,
                   const SeedVec &dispatch, const int32_t *longest_runs, const int32_t *best_rows,
                   const std::string &sequence_id, std::string &bed, unsigned host_threads,
                   const std::vector<RibbitAlignJob> *jobs, const std::vector<SswEnds> *ends, const std::vector<SswPath> *paths,
                   size_t seed_lo, size_t seed_hi, bool *order_dependent, const SmallMotifTable *small,
                   const uint32_t *job_first_all, const uint8_t *skip, std::vector<BedPiece> *pieces, const std::vector<uint32_t> *only,
                   size_t job_first_given_base, const Deferral *tree) {
    if (!pieces || !order_dependent) tree = nullptr;   // rows that go straight into `bed` cannot be put in order afterwards; and a call
                                                       // that redoes its range itself on an empty query would put its nodes off twice
    const DeferredNode *nodes = tree ? tree->nodes : nullptr;
    auto hand_over = [&](Writer &w) {        // the nodes a writer put off, to the caller's list
        if (w.deferred.empty()) return;
        std::lock_guard<std::mutex> lk(*static_cast<std::mutex *>(tree->lock));
        for (DeferredNode &nd : w.deferred) tree->out->push_back(std::move(nd));
        w.deferred.clear();
    };
    const auto wall0 = std::chrono::steady_clock::now();
    const Bases b(hp, host_threads);
    seed_hi = std::min(seed_hi, dispatch.size());
    seed_lo = std::min(seed_lo, seed_hi);
    if (only) { seed_lo = 0; seed_hi = dispatch.size(); }
    const size_t n_seeds = only ? only->size() : seed_hi - seed_lo;
    const bool have_jobs = jobs && ends && ends->size() == jobs->size();
    // first job of every seed of the range (jobs are in seed order)
    std::vector<uint32_t> job_first_own;
    const uint32_t *job_first = nullptr;          // indexed by seed - job_first_base
    size_t job_first_base = 0;
    if (have_jobs && job_first_all) { job_first = job_first_all; job_first_base = job_first_given_base; }
    else if (have_jobs) {
        const size_t span = seed_hi - seed_lo;
        job_first_own.assign(span + 1, (uint32_t)jobs->size());
        for (size_t j = jobs->size(); j-- > 0;) {
            const size_t si = (size_t)(*jobs)[j].seed_index;
            if (si >= seed_lo && si < seed_hi) job_first_own[si - seed_lo] = (uint32_t)j;
        }
        for (size_t i = span; i-- > 0;) job_first_own[i] = std::min(job_first_own[i], job_first_own[i + 1]);
        job_first = job_first_own.data();
        job_first_base = seed_lo;
    }
    auto prepare = [&](Writer &w) {
        if (have_jobs) { w.jobs = jobs->data(); w.ends = ends->data(); w.paths = (paths && paths->size() == jobs->size()) ? paths->data() : nullptr; }
        w.small = small;
        w.tree = tree;
    };
    auto one_seed = [&](size_t i, Writer &w) {
        const RibbitSeed &seed = dispatch[i];
        if (have_jobs) w.begin_seed(job_first[i - job_first_base], job_first[i - job_first_base + 1]);
        // where the seed's rows belong: its own index, or -- a node put off earlier -- its place in its seed's tree
        w.cur_root = nodes ? nodes[i].root : (uint32_t)i;
        if (nodes) w.cur_path = nodes[i].path; else w.cur_path.clear();
        w.deferred_in_seed = false;
        if (seed.mlen <= 10) { Stopwatch sws(&tl.t_small_all); w.small_seed(seed, longest_runs[i], i); }
        else { Stopwatch swl(&tl.t_long_all); w.long_seed(seed.start, seed.end, seed.mlen, seed.type, longest_runs[i], best_rows ? best_rows[i] : -1, (int)w.cur_path.size(), true); }
    };
    // seeds lo .. hi into `out`; with `segments`, into pieces cut at every seed left out and around every node put off
    auto run_range = [&](size_t lo, size_t hi, Writer &w, std::vector<BedPiece> *segments) {
        prepare(w);
        w.sink = segments;
        w.seg_first = (uint32_t)lo;
        for (size_t i = lo; i < hi; ++i) {
            prefetch_small_records(small, i + 12, dispatch.size());
            if (skip && skip[i]) { w.flush_piece((uint32_t)(i + 1)); continue; }
            one_seed(i, w);
            if (w.deferred_in_seed) { w.flush_piece((uint32_t)(i + 1)); w.deferred_in_seed = false; }
        }
        if (segments) w.flush_piece((uint32_t)hi);
        hand_over(w);
    };
    // Seeds are refined independently of each other, except that an alignment with an EMPTY query leaves the
    // reference's shared Alignment object untouched and so sees the previous seed's CIGAR.  Chunks of seeds
    // therefore run on host threads with their own writers, concatenated in seed order; if any chunk met an
    // empty query the record is redone sequentially.
    unsigned threads = host_threads ? host_threads : std::min(std::thread::hardware_concurrency(), 16u);   // one GPU's share of the host by default
    if (!host_threads)
        if (const char *env = std::getenv("RIBBIT_THREADS")) threads = (unsigned)std::max(1, std::atoi(env));
    threads = std::max(1u, std::min(threads, 256u));
    if (only) {
        // the seeds an earlier call left out, each a piece of its own (they are few and individually expensive)
        // (a seed is one piece, or several around the nodes it put off; a thread collects its own and adds them at the end)
        std::atomic<size_t> next{0};
        std::atomic<bool> empty_seen{false};
        std::mutex out_lock;
        auto work = [&]() {
            std::vector<BedPiece> mine;
            for (size_t k; (k = next.fetch_add(1)) < n_seeds;) {
                const size_t i = (*only)[k];
                Writer w(b, hp, sequence, prm, sequence_id);
                prepare(w);
                w.sink = &mine;
                w.seg_first = nodes ? nodes[i].root : (uint32_t)i;
                if (nodes) w.seg_path = nodes[i].path;
                one_seed(i, w);
                if (w.saw_empty_query) empty_seen = true;
                w.flush_piece(0);
                hand_over(w);
            }
            if (!mine.empty()) {
                std::lock_guard<std::mutex> lk(out_lock);
                for (BedPiece &pc : mine) pieces->push_back(std::move(pc));
            }
            flush_counters();
        };
        threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n_seeds));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
        work();
        for (std::thread &th : pool) th.join();
        if (empty_seen) tl_met_empty_query = true;
        if (empty_seen && order_dependent) *order_dependent = true;
        return;
    }
    if (n_seeds < 512 && !order_dependent) threads = 1;
    bool sequential = threads == 1 && !order_dependent;
    if (!sequential) {
        // ~8 chunks per thread (seed costs vary by orders of magnitude), 64..2048 seeds each
        const size_t chunk = std::min<size_t>(2048, std::max<size_t>(64, n_seeds / (threads * 8)));
        threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, (n_seeds + chunk - 1) / chunk));
        const size_t nchunks = (n_seeds + chunk - 1) / chunk;
        std::vector<std::string> parts(pieces ? 0 : nchunks);
        std::vector<std::vector<BedPiece>> part_pieces(pieces ? nchunks : 0);
        std::atomic<size_t> next{0};
        std::atomic<bool> empty_seen{false};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&]() {
                for (size_t c; (c = next.fetch_add(1)) < nchunks;) {
                    Stopwatch swr(&tl.t_range);
                    Writer w(b, hp, sequence, prm, sequence_id);
                    run_range(seed_lo + c * chunk, seed_lo + std::min(n_seeds, (c + 1) * chunk), w, pieces ? &part_pieces[c] : nullptr);
                    if (w.saw_empty_query) empty_seen = true;
                    if (!pieces) parts[c] = w.os.str();
                }
                flush_counters();
            });
        for (std::thread &th : pool) th.join();
        if (empty_seen) tl_met_empty_query = true;
        if (empty_seen && order_dependent) { *order_dependent = true; return; }
        if (empty_seen) sequential = true;
        else if (pieces) { for (auto &pp : part_pieces) for (BedPiece &pc : pp) pieces->push_back(std::move(pc)); }
        else for (const std::string &p : parts) bed += p;
    }
    if (sequential) {
        Writer w(b, hp, sequence, prm, sequence_id);
        std::vector<BedPiece> segs;
        run_range(seed_lo, seed_hi, w, pieces ? &segs : nullptr);
        if (w.saw_empty_query) tl_met_empty_query = true;
        if (pieces) for (BedPiece &pc : segs) pieces->push_back(std::move(pc));
        else bed += w.os.str();
        flush_counters();
    }
    static const bool wall_lines = std::getenv("RIBBIT_REFINE_WALL") != nullptr;      // one line per call, no per-seed timers (those perturb)
    if (wall_lines)
        std::fprintf(stderr, "[refine] call over %zu seeds on %u threads: %.1f ms wall\n", n_seeds, threads,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
    if (g_profile)
        std::fprintf(stderr, "[refine] seeds %zu  threads %u  alignments %ld (%ld with GPU passes, %ld with GPU paths)  small-motif seeds %ld from the GPU / %ld on the host  (summed over threads, cumulative) align %.2fs (of it %.2fs in %ld flank-recursion alignments, %.2fs in first-level alignments done whole on the host)  small-motif discovery %.2fs  long-motif consensus %.2fs  CIGAR digestion %.2fs  motif units %.2fs  row text %.2fs  reference strings %.2fs  long atomicity %.2fs | whole small seeds %.2fs  whole long seeds %.2fs  whole chunks %.2fs\n",
                     n_seeds, threads, g_n_align.load(), g_n_known.load(), g_n_paths.load(), g_n_small_device.load(), g_n_small_host.load(), g_t_align.load() * 1e-9, g_t_flank.load() * 1e-9, g_n_flank.load(), g_t_whole_first.load() * 1e-9, g_t_small.load() * 1e-9, g_t_long.load() * 1e-9,
                     g_t_digest.load() * 1e-9, g_t_units.load() * 1e-9, g_t_row.load() * 1e-9, g_t_query.load() * 1e-9, g_t_atom.load() * 1e-9,
                     g_t_small_all.load() * 1e-9, g_t_long_all.load() * 1e-9, g_t_range.load() * 1e-9);
}

void alignment_counters(long &all, long &gpu_passes, long &gpu_paths) { all = g_n_align.load(); gpu_passes = g_n_known.load(); gpu_paths = g_n_paths.load(); }

void small_motif_counters(long &from_device, long &on_host) { from_device = g_n_small_device.load(); on_host = g_n_small_host.load(); }

}  // namespace rb
