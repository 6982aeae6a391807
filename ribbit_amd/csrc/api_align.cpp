// api_align.cpp -- seed scans (a13-a15), alignment jobs, batched striped passes and path searches (f1); see api_internal.h for the map of the files behind include/ribbit_hip.h.
// There is no CPU fallback for any scan anywhere in this library.
#include "api_internal.h"

namespace rbapi {

// longestContinuousMatches of every dispatched seed, one GPU launch (a13)
int build_longest_runs(RibbitHandle *h) {
    if (h->longest_valid) return RIBBIT_OK;
    int rc = advance_to_anchored(h);
    if (rc) return rc;
    if ((rc = bind_device(h))) return rc;
    const size_t n = h->dispatch.size();
    h->longest_runs.resize(n);
    if (n) {
        if ((rc = h->d_seeds.ensure(n))) return rc;
        if ((rc = h->d_longest.ensure(n))) return rc;
        // Through page-locked memory, copied on the host threads: the list is a quarter of a gigabyte for a chromosome, and a copy
        // straight from (or to) a vector's pageable storage runs at a tenth of the link's speed on one thread.
        if ((rc = h->h_seed_stage.ensure(n)) || (rc = h->h_longest_stage.ensure(n))) return rc;
        unsigned nt = h->host_threads ? h->host_threads : std::min(std::thread::hardware_concurrency(), 16u);
        if (!h->host_threads)
            if (const char *env = std::getenv("RIBBIT_THREADS")) nt = (unsigned)std::max(1, std::atoi(env));
        nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n / 262144 + 1));
        auto on_threads = [&](auto fn) {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nt; ++t) pool.emplace_back(fn, n * t / nt, n * (t + 1) / nt);
            fn((size_t)0, n / nt);
            for (std::thread &th : pool) th.join();
        };
        on_threads([&](size_t lo, size_t hi) { std::memcpy(h->h_seed_stage.p + lo, h->dispatch.data() + lo, (hi - lo) * sizeof(RibbitSeed)); });
        HIP_TRY(hipMemcpyAsync(h->d_seeds.p, h->h_seed_stage.p, n * sizeof(RibbitSeed), hipMemcpyHostToDevice, h->stream));
        rb::launch_seed_longest_runs(h->d_xa.p, h->xa_stride, h->params.min_motif, h->d_seeds.p, (int64_t)n, h->d_longest.p, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->h_longest_stage.p, h->d_longest.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        on_threads([&](size_t lo, size_t hi) { std::memcpy(h->longest_runs.data() + lo, h->h_longest_stage.p + lo, (hi - lo) * sizeof(int32_t)); });
    }
    h->longest_valid = true;
    return RIBBIT_OK;
}

// mostFrequentLongerMotif's row selection for every dispatched seed with m > 10 that will reach it
// (parse_seed.cpp:360-386), one GPU launch (a15)
// best[i] (where it is -1 on entry and seed i reaches mostFrequentLongerMotif) = the row it selects, for any list of seeds of
// the loaded record: the dispatched seeds, or nodes of their recursion trees put off for a GPU batch (refine.h)
int best_rows_of(RibbitHandle *h, const RibbitRefineParams &prm, const rb::SeedVec &seeds, const int32_t *longest, int32_t *best) {
    int rc;
    const size_t n = seeds.size();
    rb::SeedVec jobs;          // reused as int4 {seed_start, seed_sequence_length, m, index}
    {
        // the usable length of every long-motif seed (a walk over its bases up to the first N) on the host threads: 0.8 M seeds of
        // a hundred bases per 64 Mbp were 80 ms on one
        unsigned nt = h->host_threads ? h->host_threads : std::min(std::thread::hardware_concurrency(), 16u);
        if (!h->host_threads)
            if (const char *env = std::getenv("RIBBIT_THREADS")) nt = (unsigned)std::max(1, std::atoi(env));
        nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n / 65536 + 1));
        std::vector<rb::SeedVec> part(nt);
        auto work = [&](unsigned t) {
            const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
            for (size_t i = lo; i < hi; ++i) {
                const RibbitSeed &s = seeds[i];
                if (best[i] >= 0 || s.mlen <= 10 || s.end - s.start < 0.9 * s.mlen || longest[i] < prm.continuous_ones_threshold) continue;
                part[t].push_back(RibbitSeed{s.start, rb::usable_length_host(h->host, s.start, s.end, s.mlen), s.mlen, (int32_t)i});
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
        for (const rb::SeedVec &p : part) jobs.insert(jobs.end(), p.begin(), p.end());
    }
    if (!jobs.empty()) {
        if ((rc = bind_device(h))) return rc;
        if ((rc = h->d_sym.ensure((size_t)h->length + 16))) return rc;
        if ((rc = h->d_seeds_small.ensure(jobs.size()))) return rc;      // (d_seeds holds the dispatch list: the small-motif scan beside this one reads it)
        if ((rc = h->d_best.ensure(jobs.size()))) return rc;
        if (!h->sym_valid) { rb::launch_sym(h->dev_ascii_src, h->length, h->d_sym.p, h->stream); HIP_TRY(hipGetLastError()); h->sym_valid = true; }
        HIP_TRY(hipMemcpyAsync(h->d_seeds_small.p, jobs.data(), jobs.size() * sizeof(RibbitSeed), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(h->d_best.p, 0, jobs.size() * sizeof(unsigned long long), h->stream));
        // 64-row slices of every seed: {job, first row}
        std::vector<int32_t> slices;
        for (size_t j = 0; j < jobs.size(); ++j) {
            const int64_t seed_end = std::min<int64_t>((int64_t)jobs[j].start + jobs[j].end, h->length);   // .end holds the length here
            const int64_t rows = seed_end - jobs[j].mlen + 1 - jobs[j].start;
            for (int64_t r = 0; r < rows; r += 64) { slices.push_back((int32_t)j); slices.push_back((int32_t)r); }
        }
        if ((rc = h->d_slices.ensure(std::max<size_t>(slices.size(), 2)))) return rc;
        if (!slices.empty())
            HIP_TRY(hipMemcpyAsync(h->d_slices.p, slices.data(), slices.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        rb::launch_long_motif_rows(h->d_sym.p, h->length, h->d_seeds_small.p, (int64_t)jobs.size(), h->d_slices.p,
                                   (int64_t)(slices.size() / 2), h->d_best.p, h->stream);
        HIP_TRY(hipGetLastError());
        std::vector<unsigned long long> got(jobs.size());
        HIP_TRY(hipMemcpyAsync(got.data(), h->d_best.p, jobs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (size_t j = 0; j < jobs.size(); ++j)      // no positive score: mmotif_index keeps its initial 0 (parse_seed.cpp:165)
            best[(size_t)jobs[j].type] = got[j] ? (int32_t)(0xffffffffu - (uint32_t)got[j]) : 0;
    }
    return RIBBIT_OK;
}

int build_best_rows(RibbitHandle *h, const RibbitRefineParams &prm) {
    if (h->best_rows_valid) return RIBBIT_OK;
    int rc = build_longest_runs(h);
    if (rc) return rc;
    {
        // (-1 everywhere, on the host threads: seventy megabytes for a chromosome were 10-15 ms of every refinement on one)
        const size_t n = h->dispatch.size();
        h->best_rows.resize(n);
        unsigned nt = h->host_threads ? h->host_threads : std::min(std::thread::hardware_concurrency(), 16u);
        if (!h->host_threads)
            if (const char *env = std::getenv("RIBBIT_THREADS")) nt = (unsigned)std::max(1, std::atoi(env));
        nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n / 1048576 + 1));
        auto fill = [&](size_t lo, size_t hi) { std::fill(h->best_rows.begin() + (std::ptrdiff_t)lo, h->best_rows.begin() + (std::ptrdiff_t)hi, -1); };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(fill, n * t / nt, n * (t + 1) / nt);
        fill(0, n / nt);
        for (std::thread &th : pool) th.join();
    }
    if ((rc = best_rows_of(h, prm, h->dispatch, h->longest_runs.data(), h->best_rows.data()))) return rc;
    h->best_rows_valid = true;
    return RIBBIT_OK;
}

// possibleMotifs of every dispatched seed with m <= 10 that reaches it (parse_smallmotif_seed.cpp:234-236), one GPU
// launch (a14 / f2); seeds the kernel flags (more than 64 classes) keep flags != 0 and the host twin runs for them
// `stream`: where its copies and its kernel go (default: the handle's).  With another stream it may run beside build_best_rows on
// another thread, PROVIDED the longest runs and the symbols are there already (scan_seeds_side_by_side sees to that).
int build_small_motifs(RibbitHandle *h, const RibbitRefineParams &prm, hipStream_t stream) {
    if (h->small_valid) return RIBBIT_OK;
    if (!stream) stream = h->stream;
    int rc = build_longest_runs(h);
    if (rc) return rc;
    const size_t n = h->dispatch.size();
    const double t0 = now_ms();
    if ((rc = h->small_head.ensure(std::max<size_t>(4 * n, 4)))) return rc;
    h->n_small_records = 0;
    // The kernel selects the seeds itself (m <= 10, a long enough run of matches) from the dispatch list and the longest runs that
    // build_longest_runs left on the device: until late in round 4 the host made a job list of them -- a walk over seventeen million
    // seeds on one thread, a quarter of a gigabyte of fresh memory and its copy up: 60 of this scan's 130-160 ms for a chromosome.
    if (n == 0) {
        // nothing to do
    } else {
        if ((rc = bind_device(h))) return rc;
        rb::SmallMotifLimits lim{};
        for (int m = 1; m <= 10; ++m) {
            int d = 0;
            while (!(d >= 0.9 * m - 1)) ++d;                 // the reference's test, in its own (double) arithmetic
            lim.first_window[m] = d;
            lim.min_length[m] = prm.min_length[m];
            lim.min_units[m] = prm.perfect_units[m];
        }
        // room for four records a seed (0.43 on average on the simulated 20-Mbp record: early reports, the one reported
        // survivor, and all classes only for the seeds with two or more) and a million more; a seed that finds the arena
        // full is left to the host
        const size_t cap = std::min<size_t>(4 * n + (1u << 20), 0x7fffffffu);
        if ((rc = h->d_sym.ensure((size_t)h->length + 16)) || (rc = h->d_small_head.ensure(4 * n)) ||
            (rc = h->d_small_records.ensure(4 * cap)) || (rc = h->d_small_count.ensure(4)))
            return rc;
        if (!h->sym_valid) { rb::launch_sym(h->dev_ascii_src, h->length, h->d_sym.p, stream); HIP_TRY(hipGetLastError()); h->sym_valid = true; }
        HIP_TRY(hipMemsetAsync(h->d_small_count.p, 0, 4 * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(h->d_small_head.p, 0xff, 4 * n * sizeof(int32_t), stream));      // flags -1: no device result
        rb::launch_small_motifs(h->d_sym.p, h->length, h->d_seeds.p, (int64_t)n, lim, h->d_small_records.p, (uint32_t)cap, h->d_small_count.p,
                                h->d_small_head.p, stream, h->d_longest.p, prm.continuous_ones_threshold);
        HIP_TRY(hipGetLastError());
        uint32_t used = 0;
        HIP_TRY(hipMemcpyAsync(h->small_head.p, h->d_small_head.p, 4 * n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&used, h->d_small_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        used = (uint32_t)std::min<size_t>(used, cap);
        if ((rc = h->small_records.ensure(std::max<size_t>(4 * (size_t)used, 4)))) return rc;
        if (used) {
            HIP_TRY(hipMemcpyAsync(h->small_records.p, h->d_small_records.p, 4 * (size_t)used * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        h->n_small_records = used;
    }
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    if (profile) std::fprintf(stderr, "[small motifs] %zu dispatched seeds looked at on the GPU, %zu records, %.1f ms incl. transfers\n", n, h->n_small_records, now_ms() - t0);
    h->small_valid = true;
    return RIBBIT_OK;
}

// The two scans of the dispatched seeds that refinement starts with -- consensus rows of the long-motif seeds, possibleMotifs of
// the small-motif ones -- side by side: each prepares its seeds on the host, copies, launches, copies back and post-processes
// (80-90 ms apiece at chromosome-1 size), and neither needs anything of the other.  The small-motif scan goes to a helper
// thread and the copy stream; what both read (longest runs, one symbol per base) is made first.
int scan_seeds_side_by_side(RibbitHandle *h, const RibbitRefineParams &prm) {
    int rc = build_longest_runs(h);
    if (rc) return rc;
    if (h->best_rows_valid || h->small_valid || h->dispatch.size() < 200000) {      // (a small record: not worth a thread)
        if ((rc = build_best_rows(h, prm))) return rc;
        return build_small_motifs(h, prm);
    }
    if ((rc = bind_device(h))) return rc;
    if ((rc = h->d_sym.ensure((size_t)h->length + 16))) return rc;
    if (!h->sym_valid) { rb::launch_sym(h->dev_ascii_src, h->length, h->d_sym.p, h->stream); HIP_TRY(hipGetLastError()); h->sym_valid = true; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    int small_rc = RIBBIT_OK;
    std::string small_error;
    std::thread side([&]() {
        try { small_rc = build_small_motifs(h, prm, h->copy_stream); if (small_rc) small_error = g_last_error; }
        catch (const std::bad_alloc &) { small_rc = RIBBIT_E_NOMEM; small_error = "out of host memory in the small-motif scan"; }
    });
    try { rc = build_best_rows(h, prm); }
    catch (...) { side.join(); throw; }
    side.join();
    if (rc) return rc;
    if (small_rc) { g_last_error = small_error; return small_rc; }
    return RIBBIT_OK;
}

void fill_refine_defaults(RibbitRefineParams *p, int min_motif, int max_motif) {
    std::memset(p, 0, sizeof *p);
    p->purity_threshold = 0.85f;             // global_variables.cpp:44 (the -p option is never read)
    p->continuous_ones_threshold = 3;        // ribbit.cpp:191
    std::vector<char> known(RIBBIT_TABLE, 0);
    for (int k = min_motif; k <= max_motif && k < RIBBIT_TABLE; ++k) { p->min_length[k] = std::max(12, 2 * k); known[k] = 1; }   // ribbit.cpp:153-159
    for (int m = 1; m <= max_motif && m < RIBBIT_TABLE; ++m) p->perfect_units[m] = m == 1 ? 8 : m == 2 ? 4 : m == 3 ? 3 : 2;     // :166-173
    for (int m = min_motif; m <= max_motif && m < RIBBIT_TABLE; ++m)                                                            // :219-235
        for (int f = 1; f <= m / 2; ++f)
            if (m % f == 0 && !known[f]) { p->min_length[f] = p->min_length[m]; known[f] = 1; }
}

// forward + reverse striped Smith-Waterman passes of every job in one (two) launches; ends[j].flag == -1 where the
// job is too large for the kernel's LDS budget (the host aligns those)
// size class of an alignment job on the GPU: 0 / 1 one DPP row resp. one wavefront with short tails, 2 / 3 / 4 the long classes
// (a workgroup of 4 / 8 / 16 wavefronts per alignment), -1 too large for the kernels (the host aligns it)
int ssw_class(const RibbitAlignJob &jb) {
    if (jb.query_length <= rb::SSW_SMALL_Q && jb.ppr_length <= rb::SSW_SMALL_R) return 0;
    if (jb.query_length <= rb::SSW_BIG_Q && jb.ppr_length <= rb::SSW_BIG_R) return 1;
    if (jb.query_length <= rb::SSW_HUGE_Q && jb.ppr_length <= rb::SSW_HUGE_R) return 2;
    if (jb.query_length <= rb::SSW_GIANT_Q && jb.ppr_length <= rb::SSW_GIANT_R) return 3;
    if (jb.query_length <= rb::SSW_COLOSSAL_Q && jb.ppr_length <= rb::SSW_COLOSSAL_R) return 4;
    return -1;
}

// classes: bit c set = jobs of size class c run here (the others keep flag -1).  pool_resident: the motif pool is on the
// device already (an earlier call of the same record uploaded it).
static std::atomic<int64_t> g_tm[16];
double feeder_phase_ms(int k) { return (double)g_tm[k].load() / 1e3; }
static inline void tm_add(int k, double t0) { g_tm[k] += (int64_t)((now_ms() - t0) * 1000.0); }
int run_ssw_passes(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const char *pool, size_t pool_len, int mask_len,
                          std::vector<rb::SswEnds> &ends, unsigned classes, bool pool_resident) {
    double tq = now_ms();
    static_assert(sizeof(RibbitAlignJob) == 9 * sizeof(int32_t), "job record layout");
    static_assert(sizeof(rb::SswEnds) == 8 * sizeof(int32_t), "ends record layout");
    ends.assign(n, rb::SswEnds{});
    if (n == 0) return RIBBIT_OK;
    if (n > 0x3fffffffu) return fail(RIBBIT_E_ARG, "too many alignment jobs");
    if (!h->dev_ascii_src) return fail(RIBBIT_E_STATE, "the record's bases are not resident on the device");
    int rc;
    if ((rc = bind_device(h))) return rc;
    // four size classes, each sorted by work (largest first) so that the alignments of a wavefront are alike.  The last one
    // (queries of 2049..4096 bases: a thousand jobs in a 64-Mbp record, and a third of all its alignment cells) runs on the
    // handle's copy stream beside the others: one such alignment occupies its wavefront for tens of milliseconds.
    // Order inside a class: by work (cells), largest first, so that the alignments of a wavefront are alike.  A bucket per
    // (power of two, next four bits) of the work instead of a comparison sort: the order only has to be roughly monotone, and
    // the sort was a tenth of a slice's time on the feeder thread (190 K jobs a slice, seven slices a record).
    constexpr int BUCKETS = 32 * 16;
    auto bucket_of = [](uint64_t work) {
        if (work < 16) return (int)work;
        const int top = 63 - __builtin_clzll(work);                    // >= 4
        return (top - 3) * 16 + (int)((work >> (top - 4)) & 15u);
    };
    std::vector<int32_t> cls_of(n, -1), bkt_of(n, 0);
    constexpr int NCLS = 5;
    std::vector<uint32_t> count(NCLS * BUCKETS + 1, 0);                // slot = class-major, buckets descending
    size_t class_count[NCLS] = {0, 0, 0, 0, 0};
    for (size_t j = 0; j < n; ++j) {
        const RibbitAlignJob &jb = jobs[j];
        const int cls = ssw_class(jb);
        if (cls < 0 || !((classes >> cls) & 1u)) { ends[j].flag = -1; continue; }
        const uint64_t work = (uint64_t)std::max(jb.query_length, 0) * (uint64_t)std::max(jb.ppr_length, 0);      // < 2^27
        cls_of[j] = cls;
        bkt_of[j] = std::min(bucket_of(work), BUCKETS - 1);
        ++class_count[cls];
        // order of the list: class 4 first ... class 0 last; inside a class the largest bucket first
        ++count[(size_t)(NCLS - 1 - cls) * BUCKETS + (size_t)(BUCKETS - 1 - bkt_of[j]) + 1];
    }
    for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
    std::vector<int32_t> order(count.back());
    for (size_t j = 0; j < n; ++j)
        if (cls_of[j] >= 0) order[count[(size_t)(NCLS - 1 - cls_of[j]) * BUCKETS + (size_t)(BUCKETS - 1 - bkt_of[j])]++] = (int32_t)j;
    const size_t n_colossal = class_count[4], n_giant = class_count[3], n_huge = class_count[2], n_big = class_count[1], n_small = class_count[0];
    if (order.empty()) return RIBBIT_OK;
    tm_add(0, tq); tq = now_ms();
    if ((rc = h->d_ssw_jobs.ensure(n * 9, true))) return rc;
    if ((rc = h->d_ssw_out.ensure(n * 8, true))) return rc;
    if ((rc = h->d_ssw_order.ensure(order.size(), true))) return rc;
    if ((rc = h->d_ssw_pool.ensure(std::max<size_t>(pool_len, 1), true))) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_ssw_jobs.p, jobs, n * sizeof(RibbitAlignJob), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_ssw_order.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    if (pool_len && !pool_resident) HIP_TRY(hipMemcpyAsync(h->d_ssw_pool.p, pool, pool_len, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->d_ssw_out.p, 0xff, n * 8 * sizeof(int32_t), h->stream));      // flag -1 unless a kernel writes the record
    tm_add(1, tq); tq = now_ms();
    // The long classes run a workgroup per alignment (ssw_group.hip: 4 / 8 / 16 wavefronts for the huge / giant / colossal class).
    // (Until round 4 RIBBIT_SSW_GROUP=0 selected the older one-wavefront-per-alignment kernel for them: it lost its measurement
    // in round 3 -- 295 against 110 ms for a 64-Mbp record's long batch -- and is gone from the product)
    const size_t n_apart = n_colossal + n_giant;      // on the copy stream beside the others
    if (n_apart) {
        HIP_TRY(hipEventRecord(h->ev_ssw, h->stream));
        HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->ev_ssw, 0));
        // queries of 4097..8192 bases: a workgroup of 16 wavefronts and 124 KB of LDS (left to the host if the device will not give that much)
        if (n_colossal && rb::ssw_group_fits(rb::SSW_COLOSSAL_Q, 16))
            HIP_TRY(rb::launch_ssw_passes_group(h->dev_ascii_src, h->length, h->d_ssw_pool.p, h->d_ssw_jobs.p, h->d_ssw_order.p, (int)n_colossal, mask_len,
                                                rb::SSW_COLOSSAL_Q, rb::SSW_COLOSSAL_R, 16, h->d_ssw_out.p, h->copy_stream));
        if (n_giant) {
            if (rb::ssw_group_fits(rb::SSW_GIANT_Q, 8))
                HIP_TRY(rb::launch_ssw_passes_group(h->dev_ascii_src, h->length, h->d_ssw_pool.p, h->d_ssw_jobs.p, h->d_ssw_order.p + n_colossal, (int)n_giant, mask_len,
                                                    rb::SSW_GIANT_Q, rb::SSW_GIANT_R, 8, h->d_ssw_out.p, h->copy_stream));
            else
                rb::launch_ssw_passes_wave(h->dev_ascii_src, h->length, h->d_ssw_pool.p, h->d_ssw_jobs.p, h->d_ssw_order.p + n_colossal, (int)n_giant, mask_len,
                                           rb::SSW_GIANT_Q, rb::SSW_GIANT_R, h->d_ssw_out.p, h->copy_stream);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(h->ev_ssw, h->copy_stream));
    }
    const int32_t *rest = h->d_ssw_order.p + n_apart;
    rb::launch_ssw_passes(h->dev_ascii_src, h->length, h->d_ssw_pool.p, h->d_ssw_jobs.p, rest + n_huge + n_big, (int)n_small,
                          rest + n_huge, (int)n_big, rest, (int)n_huge, mask_len, h->d_ssw_out.p, h->stream, 4);
    HIP_TRY(hipGetLastError());
    if (n_apart) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_ssw, 0));
    HIP_TRY(hipStreamSynchronize(h->stream));
    tm_add(2, tq); tq = now_ms();
    HIP_TRY(hipMemcpyAsync(ends.data(), h->d_ssw_out.p, n * sizeof(rb::SswEnds), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    tm_add(3, tq);
    return RIBBIT_OK;
}

// The banded path search of every job whose striped passes the GPU has done (run_ssw_passes left jobs, motif pool and end
// points on the device): rounds of one launch each, the band doubling for the alignments still open (ssw.c:603-728).
int run_ssw_paths(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const std::vector<rb::SswEnds> &ends, std::vector<rb::SswPath> &paths) {
    double tq = now_ms();
    paths.assign(n, rb::SswPath{});
    if (n == 0) return RIBBIT_OK;
    int rc;
    if ((rc = bind_device(h))) return rc;
    struct Open { int32_t job, band; };
    std::vector<Open> open;
    uint64_t worst_ops = 0;
    auto dims = [&](size_t j, int &rl, int &ql) { rl = ends[j].ref_end - ends[j].ref_begin + 1; ql = ends[j].query_end - ends[j].query_begin + 1; };
    for (size_t j = 0; j < n; ++j) {
        const rb::SswEnds &e = ends[j];
        if (e.flag == -1 || e.score == 0 || e.ref_end < 0 || e.ref_begin < 0 || e.query_begin < 0) continue;
        int rl, ql;
        dims(j, rl, ql);
        if (rl - 1 > 32767 || ql - 1 > 32767 || rl <= 0 || ql <= 0) continue;      // the distance filter: no path is searched at all
        const int band = std::abs(rl - ql) + 1;
        if (band > rb::SSW_PATH_MAX_BAND) continue;                                 // left to the host
        open.push_back({(int32_t)j, band});
        worst_ops += (uint64_t)(rl + ql + 2);
    }
    if (open.empty()) return RIBBIT_OK;
    tm_add(12, tq); tq = now_ms();
    const uint64_t path_cap = std::min<uint64_t>(worst_ops, 0xfffffff0u);
    if ((rc = h->d_path_ops.ensure((size_t)path_cap, true))) return rc;
    if ((rc = h->d_path_count.ensure(4))) return rc;
    if ((rc = h->d_path_result.ensure(4 * n, true))) return rc;
    tm_add(13, tq); tq = now_ms();
    HIP_TRY(hipMemsetAsync(h->d_path_count.p, 0, 4 * sizeof(uint32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_path_result.p, 0xff, 4 * n * sizeof(int32_t), h->stream));      // state -1: no path searched (the buffer is reused)
    tm_add(14, tq); tq = now_ms();
    std::vector<int32_t> items, result(4 * n, -1);
    std::vector<uint64_t> cell_off, ops_off;
    std::vector<Open> next;
    constexpr uint64_t ARENA = (uint64_t)6 << 30;            // cell bytes per launch
    // Alignments with a narrow band (nineteen in twenty) run four to a wavefront (ssw_path4_kernel): they come first among a
    // round's items, in seed order, the others behind them, in seed order too.
    tm_add(4, tq);
    while (!open.empty()) {
        tq = now_ms();
        // one launch per arena-full of items
        size_t n_narrow_all = 0;
        {
            next.clear();
            for (const Open &o : open) if (o.band <= rb::SSW_PATH_NARROW_BAND) next.push_back(o);
            n_narrow_all = next.size();
            for (const Open &o : open) if (o.band > rb::SSW_PATH_NARROW_BAND) next.push_back(o);
            open.swap(next);
        }
        size_t at = 0;
        next.clear();
        while (at < open.size()) {
            items.clear(); cell_off.clear(); ops_off.clear();
            uint64_t cells = 0, ops = 0;
            int max_band = 1;
            size_t first = at;
            for (; at < open.size(); ++at) {
                int rl, ql;
                dims((size_t)open[at].job, rl, ql);
                const uint64_t need = (uint64_t)(2 * open[at].band + 1) * (uint64_t)ql;
                if (cells + need > ARENA && at > first) break;
                items.push_back(open[at].job); items.push_back(open[at].band); items.push_back(0); items.push_back(0);
                cell_off.push_back(cells); ops_off.push_back(ops);
                cells += need; ops += (uint64_t)(rl + ql + 2);
                if (at >= n_narrow_all) max_band = std::max(max_band, open[at].band);        // (the LDS of the one-per-wavefront launch)
            }
            tm_add(5, tq); tq = now_ms();
            const size_t ni = at - first;
            const size_t n_narrow = first < n_narrow_all ? std::min(ni, n_narrow_all - first) : 0;
            if ((rc = h->d_path_items.ensure(items.size(), true)) || (rc = h->d_path_cell_off.ensure(ni, true)) || (rc = h->d_path_ops_off.ensure(ni, true)) ||
                (rc = h->d_path_cells.ensure((size_t)std::max<uint64_t>(cells, 16), true)) || (rc = h->d_path_scratch.ensure((size_t)ops, true)))
                return rc;
            HIP_TRY(hipMemcpyAsync(h->d_path_items.p, items.data(), items.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->d_path_cell_off.p, cell_off.data(), ni * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->d_path_ops_off.p, ops_off.data(), ni * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
            rb::launch_ssw_paths(h->dev_ascii_src, h->length, h->d_ssw_pool.p, h->d_ssw_jobs.p, h->d_ssw_out.p, h->d_path_items.p, h->d_path_cell_off.p,
                                 h->d_path_ops_off.p, (int)ni, max_band, h->d_path_cells.p, h->d_path_scratch.p, h->d_path_ops.p, (uint32_t)path_cap,
                                 h->d_path_count.p, h->d_path_result.p, h->stream, (int)n_narrow);
            HIP_TRY(hipGetLastError());
            tm_add(6, tq); tq = now_ms();
            HIP_TRY(hipStreamSynchronize(h->stream));       // the item arrays above are reused by the next launch
            tm_add(7, tq); tq = now_ms();
        }
        HIP_TRY(hipMemcpyAsync(result.data(), h->d_path_result.p, 4 * n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        tm_add(8, tq); tq = now_ms();
        for (const Open &o : open) {
            const int32_t *r = &result[4 * (size_t)o.job];
            if (r[0] == 2) {
                if (o.band * 2 <= rb::SSW_PATH_MAX_BAND) next.push_back({o.job, o.band * 2});
            } else if (r[0] == 1) {
                paths[(size_t)o.job].failed = true;
            } else if (r[0] == 0) {
                paths[(size_t)o.job].n_ops = r[3];
            }
        }
        open.swap(next);
        tm_add(9, tq);
    }
    tq = now_ms();
    uint32_t used = 0;
    HIP_TRY(hipMemcpyAsync(&used, h->d_path_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (used > path_cap) return fail(RIBBIT_E_INTERNAL, "path operations overflowed their arena");
    if ((rc = h->h_path_ops.ensure(std::max<size_t>(used, 1), true))) return rc;
    if (used) {
        HIP_TRY(hipMemcpyAsync(h->h_path_ops.p, h->d_path_ops.p, (size_t)used * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    tm_add(10, tq); tq = now_ms();
    for (size_t j = 0; j < n; ++j)
        if (result[4 * j] == 0 && !paths[j].failed) paths[j].ops = h->h_path_ops.p + (uint32_t)result[4 * j + 2];
    tm_add(11, tq);
    return RIBBIT_OK;
}

}  // namespace rbapi

extern "C" {

void ribbit_refine_params_default(RibbitRefineParams *p, int32_t min_motif, int32_t max_motif) {
    if (p) fill_refine_defaults(p, min_motif, max_motif);
}

int ribbit_hip_seed_longest_runs(RibbitHandle *h, const int32_t **out, size_t *n) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    int rc = build_longest_runs(h);
    if (rc) return rc;
    *out = h->longest_runs.data();
    *n = h->longest_runs.size();
    return RIBBIT_OK;
}

int ribbit_hip_ssw_passes(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const char *motif_pool, size_t pool_len,
                          int32_t mask_len, RibbitSswEnds *out) {
    if (!h || (n && (!jobs || !out || !motif_pool))) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    for (size_t j = 0; j < n; ++j)
        if (jobs[j].atomicity <= 0 || jobs[j].motif_offset < 0 || (size_t)jobs[j].motif_offset + (size_t)jobs[j].atomicity > pool_len)
            return fail(RIBBIT_E_ARG, "job %zu: motif outside the pool", j);
    std::vector<rb::SswEnds> ends;
    int rc = run_ssw_passes(h, jobs, n, motif_pool, pool_len, mask_len, ends);
    if (rc) return rc;
    static_assert(sizeof(RibbitSswEnds) == sizeof(rb::SswEnds), "ends record layout");
    if (n) std::memcpy(out, ends.data(), n * sizeof(RibbitSswEnds));
    return RIBBIT_OK;
}

int ribbit_hip_ssw_align_jobs(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const char *motif_pool, size_t pool_len, int32_t mask_len,
                              RibbitAlignment *out, char *cigars, size_t cap, int64_t *cigar_off, int32_t *on_gpu) {
    if (!h || (n && (!jobs || !out || !motif_pool || !cigars || !cigar_off))) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    for (size_t j = 0; j < n; ++j)
        if (jobs[j].atomicity <= 0 || jobs[j].motif_offset < 0 || (size_t)jobs[j].motif_offset + (size_t)jobs[j].atomicity > pool_len)
            return fail(RIBBIT_E_ARG, "job %zu: motif outside the pool", j);
    std::vector<rb::SswEnds> ends;
    std::vector<rb::SswPath> paths;
    int rc = run_ssw_passes(h, jobs, n, motif_pool, pool_len, mask_len, ends);
    if (rc) return rc;
    if ((rc = run_ssw_paths(h, jobs, n, ends, paths))) return rc;
    // the bases of the record for the host's share (CIGAR text; whole alignments the GPU left alone)
    std::string bases((size_t)h->length, 'N');
    if (h->length) {
        HIP_TRY(hipMemcpyAsync(&bases[0], h->dev_ascii_src, (size_t)h->length, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    size_t at = 0;
    for (size_t j = 0; j < n; ++j) {
        int qs = jobs[j].query_start, ql = jobs[j].query_length;
        if (qs < 0) { ql += qs; qs = 0; }
        if ((int64_t)qs + ql > h->length) ql = (int)(h->length - qs);
        std::string ref;
        while ((long)ref.size() <= (long)jobs[j].ppr_length) ref.append(motif_pool + jobs[j].motif_offset, (size_t)jobs[j].atomicity);
        rb::SswResult r;
        const bool gpu_ends = ends[j].flag != -1 && ql > 0;
        const bool gpu_path = gpu_ends && (paths[j].ops || paths[j].failed);
        if (ql <= 0) { r = rb::SswResult{}; r.ref_begin = r.query_begin = -1; }
        else if (gpu_path) rb::ssw_finish_with_path_periodic(bases.data() + qs, ql, motif_pool + jobs[j].motif_offset, jobs[j].atomicity, ends[j], paths[j], r);
        else if (gpu_ends) rb::ssw_finish(bases.data() + qs, ql, ref.data(), jobs[j].ppr_length, ends[j], r);
        else rb::ssw_align(bases.data() + qs, ql, ref.data(), jobs[j].ppr_length, mask_len, r);
        if (on_gpu) on_gpu[j] = gpu_path ? 2 : gpu_ends ? 1 : 0;
        out[j].sw_score = r.score; out[j].sw_score_next_best = r.score2;
        out[j].ref_begin = r.ref_begin; out[j].ref_end = r.ref_end;
        out[j].query_begin = r.query_begin; out[j].query_end = r.query_end;
        out[j].ref_end_next_best = r.ref_end2; out[j].mismatches = r.mismatches;
        out[j].flag = r.flag;
        out[j].cigar_len = (int32_t)r.cigar.size();
        cigar_off[j] = (int64_t)at;
        if (at + r.cigar.size() + 1 > cap) return fail(RIBBIT_E_OVERFLOW, "CIGAR buffer too small");
        std::memcpy(cigars + at, r.cigar.c_str(), r.cigar.size() + 1);
        at += r.cigar.size() + 1;
    }
    return RIBBIT_OK;
}

int ribbit_hip_refine_jobs(RibbitHandle *h, const RibbitRefineParams *prm, const RibbitAlignJob **jobs, size_t *n,
                           const char **motif_pool) {
    if (!h || !prm || !jobs || !n || !motif_pool) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    h->best_rows_valid = false;            // depends on prm's thresholds
    h->small_valid = false;
    int rc = build_best_rows(h, *prm);
    if (rc) return rc;
    if ((rc = build_small_motifs(h, *prm))) return rc;
    const rb::SmallMotifTable small{h->small_head.p, h->small_records.p};
    rb::build_align_jobs(h->host, *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(), h->jobs, h->motif_pool, 1, 0, (size_t)-1, &small);
    *jobs = h->jobs.data();
    *n = h->jobs.size();
    *motif_pool = h->motif_pool.c_str();
    return RIBBIT_OK;
}

int ribbit_host_refine_jobs(const RibbitScanParams *params, const RibbitRefineParams *prm, int64_t length,
                            const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                            const uint32_t *xa, size_t xa_stride, const RibbitSeed *dispatch, size_t n_dispatch,
                            RibbitAlignJob **jobs, size_t *n_jobs, char **motif_pool, size_t *pool_len) {
    if (!params || !prm || !jobs || !n_jobs || !motif_pool || !pool_len || (n_dispatch && !dispatch)) return fail(RIBBIT_E_ARG, "null argument");
    if (length > 0 && (!hi || !lo || !brk)) return fail(RIBBIT_E_ARG, "null plane");
    if (nwords < (size_t)(length / 32 + 1) || (xa && xa_stride < (size_t)(length / 32 + 1))) return fail(RIBBIT_E_ARG, "planes too short");
    if (!xa && nwords < (size_t)(length / 32 + 1) + (size_t)(params->max_motif + 2) / 32 + 2)
        return fail(RIBBIT_E_ARG, "planes too short to recompute the composed planes (zero padding past the record)");
    rb::HostPlanes hp;
    hp.resize(length, nwords);
    if (nwords) {
        std::memcpy(hp.hi.data(), hi, nwords * sizeof(uint32_t));
        std::memcpy(hp.lo.data(), lo, nwords * sizeof(uint32_t));
        std::memcpy(hp.brk.data(), brk, nwords * sizeof(uint32_t));
    }
    const size_t nm = (size_t)(params->max_motif - params->min_motif + 1);
    if (xa) hp.xa.assign(xa, xa + nm * xa_stride);        // else: recomputed slice by slice from the packed planes
    hp.xa_stride = xa ? (int64_t)xa_stride : 0;
    hp.xa_m_lo = params->min_motif;
    hp.xa_m_hi = params->max_motif;
    rb::SeedVec seeds(dispatch, dispatch + n_dispatch);
    std::vector<int32_t> longest(n_dispatch);
    for (size_t i = 0; i < n_dispatch; ++i) longest[i] = rb::longest_run_host(hp, seeds[i].mlen, seeds[i].start, seeds[i].end);
    std::vector<RibbitAlignJob> out;
    std::string pool;
    // Test hook RIBBIT_DEBUG_JOB_SLICES=n: the jobs through the GPU pipeline's builder instead (n slices of the seed list in one
    // parallel region, each handed over by the thread that finished it: refine.cpp) and put together in seed order -- they must
    // be the same jobs, and this entry point needs no GPU (tests/test_refine.py).
    const char *slices_env = std::getenv("RIBBIT_DEBUG_JOB_SLICES");
    const size_t n_slices = slices_env ? (size_t)std::max(1, std::atoi(slices_env)) : 0;
    if (n_slices == 0) {
        rb::build_align_jobs(hp, *prm, seeds, longest.data(), nullptr, out, pool);
    } else {
        std::vector<std::pair<size_t, size_t>> bounds(n_slices);
        for (size_t c = 0; c < n_slices; ++c) bounds[c] = {n_dispatch * c / n_slices, n_dispatch * (c + 1) / n_slices};
        std::vector<std::vector<RibbitAlignJob>> slice_jobs(n_slices);
        std::vector<std::string> slice_pool(n_slices);
        std::mutex mu;
        unsigned nt = std::min(std::thread::hardware_concurrency(), 16u);
        if (const char *env = std::getenv("RIBBIT_THREADS")) nt = (unsigned)std::max(1, std::atoi(env));
        rb::build_align_jobs_slices(hp, *prm, seeds, longest.data(), nullptr, nt, nullptr, bounds,
                                    [&](size_t c, std::vector<RibbitAlignJob> &&j, std::string &&p) {
                                        std::lock_guard<std::mutex> lk(mu);
                                        slice_jobs[c] = std::move(j);
                                        slice_pool[c] = std::move(p);
                                    });
        for (size_t c = 0; c < n_slices; ++c) {
            const int32_t base = (int32_t)pool.size();
            for (RibbitAlignJob j : slice_jobs[c]) { j.motif_offset += base; out.push_back(j); }
            pool += slice_pool[c];
        }
    }
    *n_jobs = out.size();
    *pool_len = pool.size();
    *jobs = (RibbitAlignJob *)std::malloc(std::max<size_t>(out.size(), 1) * sizeof(RibbitAlignJob));
    *motif_pool = (char *)std::malloc(pool.size() + 1);
    if (!*jobs || !*motif_pool) { std::free(*jobs); std::free(*motif_pool); return fail(RIBBIT_E_NOMEM, "out of host memory"); }
    if (!out.empty()) std::memcpy(*jobs, out.data(), out.size() * sizeof(RibbitAlignJob));
    std::memcpy(*motif_pool, pool.c_str(), pool.size() + 1);
    return RIBBIT_OK;
}

int ribbit_host_longest_runs(const RibbitScanParams *params, int64_t length, const uint32_t *hi, const uint32_t *lo,
                             const uint32_t *brk, size_t nwords, const RibbitSeed *seeds, size_t n, int32_t *out) {
    if (!params || (n && (!seeds || !out)) || (length > 0 && (!hi || !lo || !brk))) return fail(RIBBIT_E_ARG, "null argument");
    if (nwords < (size_t)(length / 32 + 1) + (size_t)(params->max_motif + 2) / 32 + 2)
        return fail(RIBBIT_E_ARG, "planes too short (zero padding past the record)");
    rb::HostPlanes hp;
    hp.resize(length, nwords);
    std::memcpy(hp.hi.data(), hi, nwords * sizeof(uint32_t));
    std::memcpy(hp.lo.data(), lo, nwords * sizeof(uint32_t));
    std::memcpy(hp.brk.data(), brk, nwords * sizeof(uint32_t));
    hp.xa_m_lo = params->min_motif;
    hp.xa_m_hi = params->max_motif;
    for (size_t i = 0; i < n; ++i) {
        if (seeds[i].start < 0 || seeds[i].end > length || seeds[i].start > seeds[i].end || seeds[i].mlen < 1 || seeds[i].mlen > params->max_motif + 2)
            return fail(RIBBIT_E_ARG, "seed %zu outside the record or the shift range", i);
        out[i] = rb::longest_run_host(hp, seeds[i].mlen, seeds[i].start, seeds[i].end);
    }
    return RIBBIT_OK;
}

void ribbit_refine_jobs_free(RibbitAlignJob *jobs, char *motif_pool) {
    std::free(jobs);
    std::free(motif_pool);
}

int ribbit_ssw_align(const char *query, int32_t query_len, const char *ref, int32_t ref_len, int32_t mask_len,
                     RibbitAlignment *out, char *cigar, size_t cap) {
    if (!query || !ref || !out || (cap && !cigar) || query_len < 0 || ref_len < 0) return fail(RIBBIT_E_ARG, "bad argument");
    rb::SswResult r;
    rb::ssw_align(query, query_len, ref, ref_len, mask_len, r);
    out->sw_score = r.score; out->sw_score_next_best = r.score2;
    out->ref_begin = r.ref_begin; out->ref_end = r.ref_end;
    out->query_begin = r.query_begin; out->query_end = r.query_end;
    out->ref_end_next_best = r.ref_end2; out->mismatches = r.mismatches;
    out->flag = r.flag;
    out->cigar_len = (int32_t)r.cigar.size();
    if (cap) {
        std::strncpy(cigar, r.cigar.c_str(), cap - 1);
        cigar[cap - 1] = 0;
    }
    return RIBBIT_OK;
}

int ribbit_debug_ssw_align_periodic(const char *query, int32_t query_len, const char *motif, int32_t atom, int32_t ref_len, int32_t mask_len,
                                    RibbitAlignment *out, char *cigar, size_t cap) {
    if (!query || !motif || !out || (cap && !cigar) || query_len < 0 || ref_len < 0 || atom <= 0) return fail(RIBBIT_E_ARG, "bad argument");
    rb::SswResult r;
    rb::ssw_align_periodic(query, query_len, motif, atom, ref_len, mask_len, r);
    out->sw_score = r.score; out->sw_score_next_best = r.score2;
    out->ref_begin = r.ref_begin; out->ref_end = r.ref_end;
    out->query_begin = r.query_begin; out->query_end = r.query_end;
    out->ref_end_next_best = r.ref_end2; out->mismatches = r.mismatches;
    out->flag = r.flag;
    out->cigar_len = (int32_t)r.cigar.size();
    if (cap) {
        std::strncpy(cigar, r.cigar.c_str(), cap - 1);
        cigar[cap - 1] = 0;
    }
    return RIBBIT_OK;
}

int ribbit_hip_small_motifs(RibbitHandle *h, const RibbitRefineParams *prm, const int32_t **head, size_t *n_seeds,
                            const uint32_t **records, size_t *n_records) {
    if (!h || !prm || !head || !n_seeds || !records || !n_records) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    h->small_valid = false;                // depends on prm's thresholds
    const int rc = build_small_motifs(h, *prm);
    if (rc) return rc;
    *head = h->small_head.p;
    *n_seeds = h->dispatch.size();
    *records = h->small_records.p;
    *n_records = h->n_small_records;
    return RIBBIT_OK;
}

void ribbit_debug_alignment_counters(int64_t out[3]) {
    long a = 0, b = 0, c = 0;
    rb::alignment_counters(a, b, c);
    out[0] = a; out[1] = b; out[2] = c;
}

void ribbit_debug_small_motif_counters(int64_t out[2]) {
    long a = 0, b = 0;
    rb::small_motif_counters(a, b);
    out[0] = a; out[1] = b;
}

}  // extern "C"
