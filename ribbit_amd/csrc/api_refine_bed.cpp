// api_refine_bed.cpp -- refinement to BED text (f1, f4); see api_internal.h for the map of the files behind include/ribbit_hip.h.
// There is no CPU fallback for any scan anywhere in this library.
#include "api_internal.h"

extern "C" {

static int refine_bed_impl(RibbitHandle *h, const RibbitRefineParams *prm, const char *sequence_id, const char **text, size_t *len);

int ribbit_hip_refine_bed(RibbitHandle *h, const RibbitRefineParams *prm, const char *sequence_id,
                          const char **text, size_t *len) {
    try {
        // (the helper threads' calls report an empty query through their order_dependent flags, which end in a call on THIS thread)
        rb::refine_met_empty_query(true);
        const int rc = refine_bed_impl(h, prm, sequence_id, text, len);
        device_free_pending();       // (buffers the alignment batches outgrew: released now that their streams are idle)
        if (h) h->refine_met_empty_query = rb::refine_met_empty_query(true);
        return rc;
    } catch (const std::bad_alloc &) {           // nothing may unwind through the C boundary
        return fail(RIBBIT_E_NOMEM, "out of host memory in refinement");
    }
}

// Nodes of long-motif seeds' recursion trees that were put off for the GPU (refine.h: DeferredNode): their usable length from
// this many bases on (RIBBIT_DEFER_MIN; 0 = nothing is put off, every node is done where it is met, as until round 4).
static int defer_min_length() {
    const char *env = std::getenv("RIBBIT_DEFER_MIN");      // (read per record: the tests change it)
    return env ? std::max(0, std::atoi(env)) : 700;
}

static rb::Deferral make_deferral(std::vector<rb::DeferredNode> *out, std::mutex *lock) {
    rb::Deferral d;
    d.out = out; d.lock = lock; d.min_length = defer_min_length();
    d.max_query = rb::SSW_COLOSSAL_Q; d.max_ref = rb::SSW_COLOSSAL_R;
    return d;
}

// The nodes put off, level by level, each level like a first level of its own: consensus rows (long_motif_rows_kernel), job
// set-up, striped passes and path search of all of them in one batch on the handle's streams, then the host threads finish
// the alignments, print the rows into pieces that sort into place and put off the next level's nodes (the flanks of this
// level's that are worth it; the others are done on the spot).  *order_dependent: an empty query was met (the caller redoes
// the record in call order, on the host).
//
// A level gets a batch of its own from LEVEL_OWN_BATCH nodes on (a chromosome's first levels: 46 K, 28 K, 15 K ... alignments
// at -M 500); a smaller level -- and with it everything below it -- is finished on the host threads by plain recursion.
// Measured (chromosome-1-sized record at -M 500): the trees are deep chains (128 levels, one flank trimmed at a time), and a
// level of a hundred alignments costs 40 ms of latency on the GPU -- a long alignment holds its workgroup that long however few
// there are -- where the host threads need 20: levels batched all the way down took 6.6 s, 3.5 of them below level 8.
// (Tried and dropped, round 4: the small levels of MANY short records in flight -- a stream of reads -- in batches shared across
// the records.  100 Mbp of 10-100 kb reads at -M 500: 22.8 s on the host threads alone, 104 s with 8 reads in flight sharing
// batches, 53 s with 32: a read's tree is 3-7 levels deep with 5-30 nodes a level, every level waits for a batch, and a batch
// lasts as long as its longest alignment.  What did pay for the reads is consensus_row with AVX-512, refine.cpp.)
constexpr size_t LEVEL_OWN_BATCH = 400;

static int refine_levels(RibbitHandle *h, const RibbitRefineParams &prm, const std::string &sequence_id, std::vector<rb::DeferredNode> &nodes,
                         std::vector<rb::BedPiece> &pieces, unsigned threads, bool *order_dependent, int64_t counts[3]) {
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    const char *own_env = std::getenv("RIBBIT_LEVEL_MIN");          // (test hook, read per record: levels from this many nodes on get a batch of their own)
    const size_t own_batch = own_env ? (size_t)std::max(1, std::atoi(own_env)) : LEVEL_OWN_BATCH;
    static const bool level_lines = profile && std::getenv("RIBBIT_PROFILE_LEVELS") != nullptr;
    std::vector<rb::DeferredNode> next;
    std::mutex lock;
    std::string unused;
    double t_rows = 0, t_passes = 0, t_paths = 0, t_finish = 0, t_host = 0;
    size_t n_nodes = 0, n_host = 0;
    int level = 1;
    for (; !nodes.empty(); ++level) {
        const double t0 = now_ms();
        const size_t n = nodes.size();
        rb::SeedVec seeds(n);
        std::vector<int32_t> longest(n), best(n);
        std::vector<uint32_t> all(n);
        for (size_t i = 0; i < n; ++i) {
            seeds[i] = RibbitSeed{nodes[i].start, nodes[i].end, nodes[i].mlen, nodes[i].type};
            longest[i] = nodes[i].longest; best[i] = nodes[i].known_row; all[i] = (uint32_t)i;
        }
        bool od = false;
        if (n < own_batch) {
            // too few for a batch of their own: here, by recursion, nothing put off any further
            rb::Deferral d;
            d.nodes = nodes.data();
            rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), prm, seeds, longest.data(), best.data(), sequence_id, unused, threads,
                              nullptr, nullptr, nullptr, 0, n, &od, nullptr, nullptr, nullptr, &pieces, &all, 0, &d);
            if (od) { *order_dependent = true; return RIBBIT_OK; }
            t_host += now_ms() - t0; n_host += n;
            nodes.clear();
            break;
        }
        int rc;
        if ((rc = best_rows_of(h, prm, seeds, longest.data(), best.data()))) return rc;
        const double t1 = now_ms();
        std::vector<RibbitAlignJob> jobs;
        std::string pool;
        rb::build_align_jobs(h->host, prm, seeds, longest.data(), best.data(), jobs, pool, threads, 0, n, nullptr);
        std::vector<rb::SswEnds> ends;
        std::vector<rb::SswPath> paths;
        if ((rc = run_ssw_passes(h, jobs.data(), jobs.size(), pool.data(), pool.size(), 15, ends, 0x1fu))) return rc;
        const double t2 = now_ms();
        if ((rc = run_ssw_paths(h, jobs.data(), jobs.size(), ends, paths))) return rc;
        const double t3 = now_ms();
        next.clear();
        rb::Deferral d = make_deferral(&next, &lock);
        d.nodes = nodes.data();
        rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), prm, seeds, longest.data(), best.data(), sequence_id, unused, threads,
                          &jobs, &ends, &paths, 0, n, &od, nullptr, nullptr, nullptr, &pieces, &all, 0, &d);
        if (od) { *order_dependent = true; return RIBBIT_OK; }
        counts[0] += 1; counts[1] += (int64_t)n; counts[2] += (int64_t)jobs.size();
        const double t4 = now_ms();
        t_rows += t1 - t0; t_passes += t2 - t1; t_paths += t3 - t2; t_finish += t4 - t3; n_nodes += n;
        if (level_lines)
            std::fprintf(stderr, "[refine levels] level %d: %zu nodes put off, %zu alignments: consensus rows %.1f ms, set-up + striped passes %.1f ms, path search %.1f ms, "
                                 "host finish %.1f ms; %zu nodes put off for the next level\n", level, n, jobs.size(), t1 - t0, t2 - t1, t3 - t2, t4 - t3, next.size());
        nodes.swap(next);
    }
    if (profile)
        std::fprintf(stderr, "[refine levels] %zu nodes in %d GPU levels: consensus rows %.1f ms, set-up + striped passes %.1f ms, path search %.1f ms, host finish %.1f ms; "
                             "%zu nodes of the last level finished on the host threads by recursion in %.1f ms\n", n_nodes, level - 1, t_rows, t_passes, t_paths, t_finish, n_host, t_host);
    return RIBBIT_OK;
}

// the pieces' text into h->bed, in printing order: by seed, and inside a seed's recursion tree by place (refine.h)
static void join_pieces(RibbitHandle *h, std::vector<rb::BedPiece> &pieces, unsigned threads) {
    std::sort(pieces.begin(), pieces.end(), [](const rb::BedPiece &x, const rb::BedPiece &y) {
        return x.first_seed != y.first_seed ? x.first_seed < y.first_seed : x.path < y.path; });
    std::vector<size_t> at(pieces.size() + 1, 0);
    for (size_t k = 0; k < pieces.size(); ++k) at[k + 1] = at[k] + pieces[k].text.size();
    // Plain storage, not h->bed: a string constructs (zeroes) what it grows by, on one thread -- 35-45 ms for a chromosome's 150 MB on a
    // handle's first record; here the threads that copy the pieces are also the first to touch the pages they copy into.
    const size_t total = at[pieces.size()];
    if (total + 1 > h->bed_raw_cap) {
        std::free(h->bed_raw);
        h->bed_raw_cap = total + total / 8 + 1;
        h->bed_raw = static_cast<char *>(std::malloc(h->bed_raw_cap));
        if (!h->bed_raw) { h->bed_raw_cap = 0; throw std::bad_alloc(); }
    }
    h->bed_raw[total] = '\0';
    h->bed_raw_len = total;
    h->bed_in_raw = true;
    h->bed.clear();
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, at[pieces.size()] / (4u << 20) + 1));
    std::atomic<size_t> next_piece{0};
    auto place = [&]() {
        for (size_t k; (k = next_piece.fetch_add(64)) < pieces.size();)
            for (size_t q = k; q < std::min(pieces.size(), k + 64); ++q)
                if (!pieces[q].text.empty()) std::memcpy(h->bed_raw + at[q], pieces[q].text.data(), pieces[q].text.size());
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(place);
    place();
    for (std::thread &th : pool) th.join();
}

static std::atomic<int64_t> g_level_counts[3];      // levels run, nodes put off, their alignments (process-wide, cumulative)

static int refine_bed_impl(RibbitHandle *h, const RibbitRefineParams *prm, const char *sequence_id, const char **text, size_t *len) {
    if (!h || !prm || !sequence_id || !text || !len) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    h->best_rows_valid = false;
    h->small_valid = false;
    h->bed_in_raw = false;
    // cumulative over every handle of the process (ribbit-hip runs up to 64 workers through here at once): microseconds in atomics
    const double t_begin = now_ms();
    static std::atomic<int64_t> t_rows_us{0}, t_text_us{0}, t_jobs_us{0};
    auto add_ms = [](std::atomic<int64_t> &acc, double ms) { acc.fetch_add((int64_t)(ms * 1000.0), std::memory_order_relaxed); };
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    double t0 = now_ms();
    int rc = scan_seeds_side_by_side(h, *prm);
    if (rc) return rc;
    const rb::SmallMotifTable small{h->small_head.p, h->small_records.p};
    add_ms(t_rows_us, now_ms() - t0);
    t0 = now_ms();
    if (!h->host_bases && !h->host_ascii_valid) {      // bases not on the host in memory we may keep reading: fetch them once
        h->host_ascii.resize((size_t)h->length);
        if (h->length) {
            HIP_TRY(hipMemcpyAsync(&h->host_ascii[0], h->dev_ascii_src, (size_t)h->length, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        h->host_ascii_valid = true;
    }
    h->bed.clear();
    // First-level alignments set up on host threads, the striped passes and the banded path search of all of them in GPU
    // batches, the host then only writes the CIGAR text (whole alignments for oversized jobs and the flank recursion).
    // Used from a record size on, by measurement (tools/refine_threshold_probe.sh; refinement of one record, host
    // threads only / this path): 1 Mbp 46 / 79 ms, 2 Mbp 71 / 108, 5 Mbp 148 / 145, 10 Mbp 280 / 173, 20 Mbp 484 / 182,
    // 40 Mbp 906 / 285 -- about 70 ms fixed (the long batch and a slice's launches, which a small record cannot hide behind
    // anything), then 5 ms per Mbp against 24 on the host; and 400 records of 50 kb with 8 in flight take 2.28 s instead of
    // 1.3 s with it.  The switch is the number of dispatched seeds (0.35 M at 5 Mbp); RIBBIT_GPU_SSW=0 / =1 forces it off / on.
    constexpr size_t GPU_SSW_MIN_SEEDS = 400000;
    static const char *const gpu_ssw_env = std::getenv("RIBBIT_GPU_SSW");
    const bool gpu_ssw = gpu_ssw_env ? std::atoi(gpu_ssw_env) != 0 : h->dispatch.size() >= GPU_SSW_MIN_SEEDS;
    unsigned threads = h->host_threads ? h->host_threads : std::min(std::thread::hardware_concurrency(), 16u);
    if (!h->host_threads)
        if (const char *env = std::getenv("RIBBIT_THREADS")) threads = (unsigned)std::max(1, std::atoi(env));
    bool done = false;
    if (gpu_ssw && !h->dispatch.empty()) {
        // The pipeline (DESIGN.md 7 has the measurements behind every step):
        //   * the seeds that can have a LONG job (queries beyond 512 bases) are set up first; their long jobs go to the GPU as ONE
        //     batch on a helper handle's streams (a workgroup per alignment, ssw_group.hip) and those seeds are set aside; a seed
        //     with a job no kernel takes (queries beyond 8192 bases) is refined on a few host threads from the start;
        //   * the rest goes through in slices of the seed list.  A slice owns its jobs, motif strings and results; the main thread
        //     sets the slices up one after the other on the host threads, a helper makes each slice's tables, and two feeder
        //     threads (the second on another helper handle) take alternating slices as they are set up: striped passes and path
        //     search on the GPU, the tails of one slice's launches behind the other's work;
        //   * when all slices are set up the host threads refine them in order as their batches land, each slice with the
        //     results of its own batch;
        //   * the seeds set aside are refined last, longest first, when the long batch has landed, and all rows are put in place.
        // (Round 2 ran the long classes inside every slice: ~150 ms of tail per slice, which is why two slices were the optimum
        // and the workers sat idle for the whole first one -- tools/refine_slices_probe.sh.)
        const size_t n_seeds = h->dispatch.size();
        const double t_setup0 = now_ms();
        // First the seeds that can have a long job at all -- a job's query is at most the seed plus one motif long, and its
        // reference 15 % more plus a motif: below 500 bases of seed + motif (and a motif of at most 400) both stay inside
        // the short classes -- so that the long batch is on the GPU while the other four million seeds are still being set up.
        std::vector<uint32_t> cand;
        {   // (on the threads, pieces joined in order: seventeen million seeds on one thread were a quarter of this step)
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n_seeds / 262144 + 1));
            std::vector<std::vector<uint32_t>> part(nt);
            auto scan = [&](unsigned t) {
                const size_t lo = n_seeds * t / nt, hi = n_seeds * (t + 1) / nt;
                for (size_t i = lo; i < hi; ++i) {
                    const RibbitSeed &sd = h->dispatch[i];
                    if ((int64_t)sd.end - sd.start + sd.mlen > 500 || sd.mlen > 400) part[t].push_back((uint32_t)i);
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nt; ++t) pool.emplace_back(scan, t);
            scan(0);
            for (std::thread &th : pool) th.join();
            for (const std::vector<uint32_t> &pt : part) cand.insert(cand.end(), pt.begin(), pt.end());
        }
        std::vector<RibbitAlignJob> cand_jobs;
        std::string cand_pool;
        rb::build_align_jobs_of(h->host, *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(), cand, cand_jobs, cand_pool, threads, &small);
        std::vector<uint8_t> set_aside(n_seeds, 0);          // 1: waits for the long batch; 2: has a job beyond the kernels' reach (host-aligned)
        for (const RibbitAlignJob &jb : cand_jobs) {
            const int cls = ssw_class(jb);
            if (cls >= 0 && cls < 2) continue;
            uint8_t &mark = set_aside[(size_t)jb.seed_index];
            mark = std::max<uint8_t>(mark, cls < 0 ? 2 : 1);
        }
        // the long batch takes the long jobs of the seeds that wait for it; a seed with a job no kernel takes is aligned on
        // the host threads as a whole, right away
        std::vector<RibbitAlignJob> long_jobs;
        std::vector<uint32_t> long_ordinal, later, giants;      // long_ordinal: which of its seed's jobs a long job is
        {
            int32_t seed = -1, ordinal = 0;
            for (const RibbitAlignJob &jb : cand_jobs) {
                if (jb.seed_index != seed) { seed = jb.seed_index; ordinal = 0; }
                if (ssw_class(jb) >= 2 && set_aside[(size_t)seed] == 1) { long_jobs.push_back(jb); long_ordinal.push_back((uint32_t)ordinal); }
                ++ordinal;
            }
        }
        for (uint32_t i : cand) {
            if (set_aside[i] == 1) later.push_back(i);
            else if (set_aside[i] == 2) giants.push_back(i);
        }
        const std::string &long_pool = cand_pool;
        const double t_setup_long = now_ms() - t_setup0;

        // ---- the long batch, on a helper handle (own streams and buffers, same device, same resident bases)
        std::vector<rb::SswEnds> long_ends;
        std::vector<rb::SswPath> long_paths;
        std::vector<uint32_t> long_ops;
        int long_rc = RIBBIT_OK;
        std::string long_error;
        double t_long = 0;
        // whatever happens on this thread from here on (the set-up or refine_to_bed may throw std::bad_alloc at chromosome size),
        // the helper threads are stopped and joined before the frame goes: a joinable std::thread's destructor ends the process
        std::mutex mu;
        std::condition_variable cv;
        std::atomic<bool> stop{false};
        // (a guard is declared AFTER everything its threads touch: locals die in reverse order, so the guard joins first.  Until
        // round 4 one guard up here held all three threads, and an exception after they had started freed the slices and
        // `later_pieces` under a feeder still running.)
        std::vector<rb::BedPiece> later_pieces;
        bool later_order_dependent = false, later_done = false;
        double t_later_thread = 0;
        // nodes of the seeds' recursion trees that are put off for a GPU batch of their own (refine.h), from every call below
        std::vector<rb::DeferredNode> put_off;
        std::mutex put_off_lock;
        const rb::Deferral tree = make_deferral(&put_off, &put_off_lock);
        const rb::Deferral *const treep = tree.min_length > 0 ? &tree : nullptr;
        const rb::Deferral *treep_now = treep;        // (the workers' and the set-aside seeds' calls: see the loop over the slices)
        const char *level_env = std::getenv("RIBBIT_LEVEL_MIN");
        const size_t level_batch = level_env ? (size_t)std::max(1, std::atoi(level_env)) : LEVEL_OWN_BATCH;
        if ((rc = bind_device(h))) return rc;       // before any helper thread exists: nothing to join on this way out
        std::thread long_thread, later_thread;
        struct JoinGuard {
            std::atomic<bool> &stop; std::condition_variable &cv; std::thread &a, &b;
            ~JoinGuard() { stop = true; cv.notify_all(); if (a.joinable()) a.join(); if (b.joinable()) b.join(); }
        } join_guard{stop, cv, later_thread, long_thread};
        static const bool fail_later_slices = std::getenv("RIBBIT_DEBUG_FAIL_BATCHES") != nullptr;      // test hook: see below
        if (!long_jobs.empty()) {
            if (!h->aux && (rc = ribbit_hip_open(&h->params, h->device, &h->aux))) return rc;
            RibbitHandle *aux = h->aux;
            aux->dev_ascii_src = h->dev_ascii_src; aux->length = h->length; aux->loaded = true;
            long_thread = std::thread([&, aux]() {
                const double tl0 = now_ms();
                try {
                    long_rc = run_ssw_passes(aux, long_jobs.data(), long_jobs.size(), long_pool.data(), long_pool.size(), 15, long_ends, 0x1cu);
                    if (!long_rc) long_rc = run_ssw_paths(aux, long_jobs.data(), long_jobs.size(), long_ends, long_paths);
                    if (!long_rc) {
                        size_t n_ops = 0;
                        for (const rb::SswPath &pt : long_paths) if (pt.ops) n_ops = std::max(n_ops, (size_t)(pt.ops - aux->h_path_ops.p) + (size_t)pt.n_ops);
                        long_ops.assign(aux->h_path_ops.p, aux->h_path_ops.p + n_ops);
                        for (rb::SswPath &pt : long_paths) if (pt.ops) pt.ops = long_ops.data() + (pt.ops - aux->h_path_ops.p);
                    } else long_error = g_last_error;
                } catch (const std::bad_alloc &) { long_rc = RIBBIT_E_NOMEM; long_error = "out of host memory in the long alignment batch"; }
                t_long = now_ms() - tl0;
            });
        }
        // the seeds set aside are refined as soon as the long batch AND the slice that holds their short jobs have landed, on a
        // few threads beside the workers (who are mostly waiting for the feeder): not after everything else, where their long
        // host-side tails (queries beyond the kernels' reach, flank recursion) were 120 of 715 ms at 64 Mbp
        // seeds with a job beyond the kernels' reach (queries over 8192 bases; over 4096 until the end of round 3: 5-60 ms of host
        // alignment each) need nothing from
        // the GPU: they are refined on a few host threads from the start, beside everything else, instead of as a tail
        if (!giants.empty())
            later_thread = std::thread([&]() {
                const double tl0 = now_ms();
                try {
                    bool od = false;
                    rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(),
                                      sequence_id, h->bed, std::max(2u, threads / 4), nullptr, nullptr, nullptr, 0, n_seeds, &od, &small, nullptr, nullptr,
                                      &later_pieces, &giants, 0, treep);
                    if (od) later_order_dependent = true;
                    else later_done = true;
                } catch (const std::bad_alloc &) { later_done = false; }
                t_later_thread = now_ms() - tl0;
            });

        // ---- slices of the seed list: one per ~600 K seeds, 2 .. 16 (RIBBIT_SSW_SLICES overrides).  Measured at 64 Mbp (4.5 M
        // seeds; tools/refine_slices_sweep.sh): 2 slices 788 ms, 6 .. 8 757, 17 787, 32 911, 48 1085 -- a slice costs ~10 ms of
        // launches, copies and synchronisation beyond its kernels.
        // A slice owns its first-level jobs, their motif strings and their results.  The jobs are set up slice by slice on
        // the host threads, and the feeder takes a slice as soon as it is set up: the GPU used to wait for the set-up of the
        // whole record (0.45 s at chromosome-1 size, a third of what the feeder then needs for all slices).
        size_t n_slices = std::max<size_t>(2, std::min<size_t>(16, n_seeds / 600000));
        if (const char *env = std::getenv("RIBBIT_SSW_SLICES")) n_slices = (size_t)std::max(1, std::atoi(env));
        n_slices = std::min(n_slices, n_seeds);
        struct Slice {
            size_t lo = 0, hi = 0;
            std::vector<RibbitAlignJob> jobs;       // in seed order
            std::string pool;
            std::vector<uint32_t> job_first;        // job_first[i - lo] = first job of dispatch seed i, for i = lo .. hi
            std::vector<rb::SswEnds> ends;
            std::vector<rb::SswPath> paths;
            std::vector<uint32_t> ops;
            int rc = RIBBIT_OK, table_rc = RIBBIT_OK;      // rc: the feeder's; table_rc: the table maker's (folded into rc by the feeder)
            std::string error;
            bool built = false, tabled = false, ready = false;
            double t_passes = 0, t_paths = 0, t_feed = 0;
        };
        std::vector<Slice> slices(n_slices);
        for (size_t c = 0; c < n_slices; ++c) { slices[c].lo = n_seeds * c / n_slices; slices[c].hi = n_seeds * (c + 1) / n_slices; }
        auto slice_of = [&](size_t seed) {
            size_t c = std::min(n_slices - 1, seed * n_slices / n_seeds);
            while (c > 0 && seed < slices[c].lo) --c;
            while (c + 1 < n_slices && seed >= slices[c].hi) ++c;
            return c;
        };
        // where a long job's results go: (slice of its seed, first job of the seed + which of the seed's jobs it is)
        std::vector<uint32_t> long_slice(long_jobs.size());
        for (size_t k = 0; k < long_jobs.size(); ++k) long_slice[k] = (uint32_t)slice_of((size_t)long_jobs[k].seed_index);
        // safety net: a job outside the short classes whose seed the candidate test above let through would be a bug in that
        // test's arithmetic, not in the result -- its seed is aligned on the host at the end
        std::vector<uint32_t> stragglers;
        size_t n_jobs = 0;
        // a slice's tables: first job of every seed, the results' places, the safety net.  Needed when its batch has landed, not
        // before: made by a helper thread of their own, neither between two slices' set-ups on the main thread (56 ms of the
        // set-up's 400 at chromosome-1 size) nor by the feeder (whose slices the workers then waited for)
        auto slice_tables = [&](Slice &sl) {
            const size_t nj = sl.jobs.size(), span = sl.hi - sl.lo;
            sl.job_first.assign(span + 1, (uint32_t)nj);
            for (size_t j = nj; j-- > 0;) sl.job_first[(size_t)sl.jobs[j].seed_index - sl.lo] = (uint32_t)j;
            for (size_t i = span; i-- > 0;) sl.job_first[i] = std::min(sl.job_first[i], sl.job_first[i + 1]);
            for (size_t j = 0; j < nj; ++j) {
                const int cls = ssw_class(sl.jobs[j]);
                uint8_t &mark = set_aside[(size_t)sl.jobs[j].seed_index];      // (no worker reads this slice's marks before it is ready)
                if ((cls < 0 || cls >= 2) && mark == 0) { mark = 3; stragglers.push_back((uint32_t)sl.jobs[j].seed_index); }
            }
            sl.ends.assign(nj, rb::SswEnds{});
            for (rb::SswEnds &e : sl.ends) e.flag = -1;
            sl.paths.assign(nj, rb::SswPath{});
        };
        auto feed = [&](size_t c, RibbitHandle *fh) {
            Slice &sl = slices[c];
            const double tf0 = now_ms();
            try {
                const size_t nj = sl.jobs.size();
                const double tp = now_ms();
                std::vector<rb::SswEnds> e;
                std::vector<rb::SswPath> pth;
                sl.rc = (fail_later_slices && c > 0) ? fail(RIBBIT_E_NOMEM, "forced by RIBBIT_DEBUG_FAIL_BATCHES")
                                                     : run_ssw_passes(fh, sl.jobs.data(), nj, sl.pool.data(), sl.pool.size(), 15, e, 0x3u);
                const double tq = now_ms();
                if (!sl.rc) sl.rc = run_ssw_paths(fh, sl.jobs.data(), nj, e, pth);
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return sl.tabled || stop.load(); }); if (!sl.tabled) return; }
                if (!sl.rc && sl.table_rc) { sl.rc = fail(sl.table_rc, "out of host memory while making a slice's tables"); }
                if (!sl.rc) {
                    // the paths point into the handle's pinned buffer, which the next slice overwrites
                    size_t n_ops = 0;
                    for (const rb::SswPath &pt : pth) if (pt.ops) n_ops = std::max(n_ops, (size_t)(pt.ops - fh->h_path_ops.p) + (size_t)pt.n_ops);
                    sl.ops.assign(fh->h_path_ops.p, fh->h_path_ops.p + n_ops);
                    for (size_t k = 0; k < nj; ++k) {
                        if (e[k].flag == -1) continue;          // not this batch's (a long job: the other thread owns its entries)
                        sl.ends[k] = e[k];
                        sl.paths[k] = pth[k];
                        if (pth[k].ops) sl.paths[k].ops = sl.ops.data() + (pth[k].ops - fh->h_path_ops.p);
                    }
                } else {
                    sl.error = g_last_error;
                }
                sl.t_passes = tq - tp; sl.t_paths = now_ms() - tq;
            } catch (const std::bad_alloc &) {
                sl.rc = RIBBIT_E_NOMEM;
                sl.error = "out of host memory while running a slice's alignment batches";
            }
            sl.t_feed = now_ms() - tf0;
        };
        std::thread tabler;
        struct TablerGuard { std::atomic<bool> &stop; std::condition_variable &cv; std::thread &t; ~TablerGuard() { stop = true; cv.notify_all(); if (t.joinable()) t.join(); } } tabler_guard{stop, cv, tabler};
        tabler = std::thread([&]() {
            for (size_t c = 0; c < n_slices; ++c) {
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return slices[c].built || stop.load(); }); if (!slices[c].built) return; }
                try { slice_tables(slices[c]); }
                catch (const std::bad_alloc &) { slices[c].table_rc = RIBBIT_E_NOMEM; }      // (published with `tabled`, under the mutex, below)
                { std::lock_guard<std::mutex> lk(mu); slices[c].tabled = true; }
                cv.notify_all();
            }
        });
        // Two feeders on alternating slices, the second on a helper handle of its own (streams, buffers): a launch of the path
        // search lasts as long as its longest alignment, and with one feeder the GPU idles through every such tail before the
        // next slice's passes start.
        // (Two, by measurement: with three and four of them a chromosome's refinement took 1092-1103 ms against 1105-1127 ms with
        // two, inside the run-to-run spread -- the GPU side is bound by the throughput of the passes and path searches, not by
        // gaps between one feeder's launches; round 4, tools/refine_timing.py.)
        constexpr size_t FEEDERS = 2;
        size_t n_feeders = std::min(FEEDERS, n_slices);
        for (size_t k = 1; k < n_feeders; ++k) {
            if (h->feed_aux.size() < k) {
                RibbitHandle *fa = nullptr;
                if (ribbit_hip_open(&h->params, h->device, &fa) != RIBBIT_OK) { n_feeders = k; break; }      // (no memory for it: fewer feeders)
                h->feed_aux.push_back(fa);
            }
            RibbitHandle *fa = h->feed_aux[k - 1];
            fa->dev_ascii_src = h->dev_ascii_src; fa->length = h->length; fa->loaded = true;
        }
        auto feeder_loop = [&](size_t k, RibbitHandle *fh) {
            for (size_t c = k; c < n_slices && !stop; c += n_feeders) {
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return slices[c].built || stop.load(); }); if (!slices[c].built) break; }
                feed(c, fh);
                { std::lock_guard<std::mutex> lk(mu); slices[c].ready = true; }
                cv.notify_all();
                if (slices[c].rc) break;
            }
        };
        std::vector<std::thread> feeders;
        struct FeederGuard {
            std::atomic<bool> &stop; std::condition_variable &cv; std::vector<std::thread> &all;
            ~FeederGuard() { stop = true; cv.notify_all(); for (std::thread &t : all) if (t.joinable()) t.join(); }
        } feeder_guard{stop, cv, feeders};
        feeders.reserve(n_feeders);
        for (size_t k = 0; k < n_feeders; ++k) {
            RibbitHandle *fh = k == 0 ? h : h->feed_aux[k - 1];
            feeders.emplace_back([&feeder_loop, k, fh]() { feeder_loop(k, fh); });
        }
        {
            // all slices in one parallel region (refine.cpp): a slice is handed over by the thread that finished its last chunk
            std::vector<std::pair<size_t, size_t>> bounds(n_slices);
            for (size_t c = 0; c < n_slices; ++c) bounds[c] = {slices[c].lo, slices[c].hi};
            rb::build_align_jobs_slices(h->host, *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(), threads, &small, bounds,
                                        [&](size_t c, std::vector<RibbitAlignJob> &&jobs, std::string &&pool) {
                                            slices[c].jobs = std::move(jobs);
                                            slices[c].pool = std::move(pool);
                                            { std::lock_guard<std::mutex> lk(mu); n_jobs += slices[c].jobs.size(); slices[c].built = true; }
                                            cv.notify_all();
                                        });
        }
        const double t_setup = now_ms() - t_setup0;
        bool order_dependent = false;
        double t_wait = 0, t_passes = 0, t_paths = 0, t_feed = 0, t_work = 0, t_later = 0, t_join = 0;
        std::vector<rb::BedPiece> pieces;
        for (size_t c = 0; c < n_slices; ++c) {
            Slice &sl = slices[c];
            const double tw = now_ms();
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return sl.ready; }); }
            t_wait += now_ms() - tw;
            if (sl.rc) { rc = sl.rc; g_last_error = sl.error; break; }
            t_passes += sl.t_passes; t_paths += sl.t_paths; t_feed += sl.t_feed;
            const double tk = now_ms();
            rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(),
                              sequence_id, h->bed, h->host_threads, &sl.jobs, &sl.ends, &sl.paths, sl.lo, sl.hi, &order_dependent, &small,
                              sl.job_first.data(), set_aside.data(), &pieces, nullptr, sl.lo, treep_now);
            t_work += now_ms() - tk;
            if (order_dependent) break;
            // Putting nodes off pays when a level fills a GPU batch of its own (refine_levels: LEVEL_OWN_BATCH nodes); nodes of smaller
            // levels are finished by plain recursion BEHIND everything else, where the workers would have done them beside the feeders
            // for nothing.  At -M 100 a chromosome puts 35 nodes off (80 ms at the end of 1.4 s), at -M 500 a hundred thousand: if the
            // slices so far do not promise a batch's worth over the whole record, the rest is done where it is met.
            if (treep_now) {
                size_t so_far;
                { std::lock_guard<std::mutex> lk(put_off_lock); so_far = put_off.size(); }
                if (so_far * n_slices < level_batch * (c + 1)) treep_now = nullptr;
            }
        }
        stop = true;
        cv.notify_all();
        for (std::thread &t : feeders) t.join();
        const double tw = now_ms();
        if (long_thread.joinable()) long_thread.join();
        if (later_thread.joinable()) later_thread.join();
        const double t_wait_long = now_ms() - tw;
        if (later_order_dependent) order_dependent = true;
        if (!rc && long_rc) { rc = long_rc; g_last_error = long_error; }
        bool batches_failed = false;
        if (rc == RIBBIT_E_NOMEM) {
            // the batches' buffers did not fit (several large records in flight on one GPU): the alignments of this record
            // run on the host threads instead, with the same result
            std::fprintf(stderr, "ribbit_hip_refine_bed: GPU alignment batches skipped for this record (%s)\n", g_last_error.c_str());
            rc = RIBBIT_OK;
            batches_failed = true;
        }
        if (rc) return rc;
        if (!order_dependent && !batches_failed) {
            // the seeds set aside: their long alignments come from the long batch, the others from the slices
            const double tl0 = now_ms();
            // the seeds set aside for the long batch: its results in place, then one call per slice over those of its seeds
            for (size_t k = 0; k < long_jobs.size(); ++k) {
                Slice &sl = slices[long_slice[k]];
                const size_t at = (size_t)sl.job_first[(size_t)long_jobs[k].seed_index - sl.lo] + long_ordinal[k];
                sl.ends[at] = long_ends[k]; sl.paths[at] = long_paths[k];
            }
            if (!later.empty()) {
                // one call over all of them, on all threads (they are few and individually expensive: a call per slice waited
                // for its slowest seed seven times over): their jobs and results gathered from the slices, in seed order
                std::vector<RibbitAlignJob> lj;
                std::vector<rb::SswEnds> le;
                std::vector<rb::SswPath> lp;
                std::vector<uint32_t> lfirst(n_seeds + 1, 0);       // only the entries of these seeds (and the one after each) are read
                for (uint32_t i : later) {
                    const Slice &sl = slices[slice_of(i)];
                    const size_t ja = sl.job_first[i - sl.lo], jb = sl.job_first[i - sl.lo + 1];
                    lfirst[i] = (uint32_t)lj.size();
                    lj.insert(lj.end(), sl.jobs.begin() + (long)ja, sl.jobs.begin() + (long)jb);
                    le.insert(le.end(), sl.ends.begin() + (long)ja, sl.ends.begin() + (long)jb);
                    lp.insert(lp.end(), sl.paths.begin() + (long)ja, sl.paths.begin() + (long)jb);
                    lfirst[i + 1] = (uint32_t)lj.size();
                }
                // the longest first: a seed of these costs anything from microseconds to tens of milliseconds (digestion of a long
                // alignment, flank recursion on the host), and the call ends with its last seed
                std::vector<uint32_t> by_cost(later);
                std::stable_sort(by_cost.begin(), by_cost.end(), [&](uint32_t x, uint32_t y) {
                    return h->dispatch[x].end - h->dispatch[x].start > h->dispatch[y].end - h->dispatch[y].start; });
                rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(),
                                  sequence_id, h->bed, h->host_threads, &lj, &le, &lp, 0, n_seeds, &order_dependent, &small, lfirst.data(), nullptr,
                                  &pieces, &by_cost, 0, treep_now);
            }
            if (!stragglers.empty())
                rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(),
                                  sequence_id, h->bed, h->host_threads, nullptr, nullptr, nullptr, 0, n_seeds, &order_dependent, &small, nullptr, nullptr,
                                  &pieces, &stragglers);
            if (!giants.empty()) {
                if (later_done) { for (rb::BedPiece &pc : later_pieces) pieces.push_back(std::move(pc)); }
                else              // (the thread ran out of memory)
                    rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(),
                                      sequence_id, h->bed, h->host_threads, nullptr, nullptr, nullptr, 0, n_seeds, &order_dependent, &small, nullptr, nullptr,
                                      &pieces, &giants);
            }
            t_later = now_ms() - tl0;
        }
        double t_levels = 0;
        if (!order_dependent && !batches_failed && !put_off.empty()) {
            // the nodes put off by all of the above, level by level on this handle's streams (nothing else runs on them now)
            const double tv0 = now_ms();
            int64_t counts[3] = {0, 0, 0};
            rc = refine_levels(h, *prm, sequence_id, put_off, pieces, threads, &order_dependent, counts);
            for (int k = 0; k < 3; ++k) g_level_counts[k] += counts[k];
            if (rc == RIBBIT_E_NOMEM) { rc = RIBBIT_OK; batches_failed = true; }
            if (rc) return rc;
            t_levels = now_ms() - tv0;
        }
        done = !order_dependent && !batches_failed;
        const double tj0 = now_ms();
        if (done) join_pieces(h, pieces, threads);     // the pieces' text into place on the threads (150 MB for a chromosome)
        else h->bed.clear();                           // an empty query somewhere (or no batches): the whole record in one call (below)
        t_join = now_ms() - tj0;
        if (profile) { std::fprintf(stderr, "[feeders' host phases, ms, cumulative] passes: order %.1f, H2D+memset enqueue %.1f, launches+kernel wait %.1f, ends D2H %.1f | paths: first loop %.1f, ensure %.1f, memsets %.1f, vectors %.1f, items %.1f, H2D+launch %.1f, kernel wait %.1f, result D2H %.1f, round loop %.1f, ops D2H %.1f, final loop %.1f\n",
            feeder_phase_ms(0), feeder_phase_ms(1), feeder_phase_ms(2), feeder_phase_ms(3), feeder_phase_ms(12), feeder_phase_ms(13), feeder_phase_ms(14), feeder_phase_ms(4), feeder_phase_ms(5), feeder_phase_ms(6), feeder_phase_ms(7), feeder_phase_ms(8), feeder_phase_ms(9), feeder_phase_ms(10), feeder_phase_ms(11)); }
        if (profile) std::fprintf(stderr, "[refine_bed] %zu alignment jobs (%zu long ones in their own batch: %.1f ms, set up first in %.1f ms; %zu seeds set aside), set-up in all %.1f ms; %zu slices: "
                                  "feeder %.1f ms in all (GPU striped passes incl. transfers %.1f ms, GPU path search %.1f ms); workers: %.1f ms in their calls, waited %.1f ms "
                                  "for slices, %.1f ms for the long batch; seeds set aside for it %.1f ms; %zu seeds with jobs beyond the kernels' reach refined on the host beside all that in %.1f ms; "
                                  "nodes put off, level by level %.1f ms; rows put together %.1f ms; since the call began %.1f ms\n",
                                  n_jobs, long_jobs.size(), t_long, t_setup_long, later.size(), t_setup, n_slices, t_feed, t_passes, t_paths, t_work, t_wait, t_wait_long, t_later, giants.size(), t_later_thread, t_levels, t_join,
                                  now_ms() - t_begin);
        add_ms(t_jobs_us, t_wait + t_wait_long);
    }
    if (!done && !gpu_ssw && defer_min_length() > 0 && h->dev_ascii_src && !h->dispatch.empty()) {
        // A record below the size of the GPU alignment pipeline: its seeds are refined on the host threads, but the expensive
        // nodes of its long-motif seeds -- first level or flanks -- are put off, and where they are many (a few megabases at
        // -M 500: thousands) they get GPU batches of their own, level by level; where they are few (a read: 5-30) they are
        // finished on the host threads right after.
        t0 = now_ms();
        std::vector<rb::DeferredNode> put_off;
        std::mutex put_off_lock;
        const rb::Deferral tree = make_deferral(&put_off, &put_off_lock);
        std::vector<rb::BedPiece> pieces;
        bool order_dependent = false;
        rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(), sequence_id, h->bed,
                          h->host_threads, nullptr, nullptr, nullptr, 0, (size_t)-1, &order_dependent, &small, nullptr, nullptr, &pieces, nullptr, 0, &tree);
        bool no_room = false;
        if (!order_dependent && !put_off.empty()) {
            int64_t counts[3] = {0, 0, 0};
            rc = refine_levels(h, *prm, sequence_id, put_off, pieces, threads, &order_dependent, counts);
            for (int k = 0; k < 3; ++k) g_level_counts[k] += counts[k];
            if (rc == RIBBIT_E_NOMEM) { rc = RIBBIT_OK; no_room = true; }
            if (rc) return rc;
        }
        if (!order_dependent && !no_room) { join_pieces(h, pieces, threads); done = true; }
        else h->bed.clear();
        add_ms(t_text_us, now_ms() - t0);
    }
    if (!done) {
        t0 = now_ms();
        rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(), sequence_id, h->bed,
                          h->host_threads, nullptr, nullptr, nullptr, 0, (size_t)-1, nullptr, &small);
        add_ms(t_text_us, now_ms() - t0);
    }
    if (profile) std::fprintf(stderr, "[refine_bed] cumulative: GPU scans of the seeds %.1f ms, alignment set-up + GPU striped passes %.1f ms, host refinement + BED %.1f ms; nodes put off for GPU batches: %lld in %lld levels (%lld alignments)\n",
                              t_rows_us.load() / 1000.0, t_jobs_us.load() / 1000.0, t_text_us.load() / 1000.0,
                              (long long)g_level_counts[1].load(), (long long)g_level_counts[0].load(), (long long)g_level_counts[2].load());
    *text = h->bed_in_raw ? h->bed_raw : h->bed.c_str();
    *len = h->bed_in_raw ? h->bed_raw_len : h->bed.size();
    return RIBBIT_OK;
}

int ribbit_host_refine_bed(const RibbitScanParams *params, const RibbitRefineParams *prm, const char *sequence, int64_t length,
                           const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                           const uint32_t *xa, size_t xa_stride, const RibbitSeed *dispatch, size_t n_dispatch,
                           const char *sequence_id, char **text, size_t *len) {
    if (!params || !prm || !text || !len || !sequence_id || (n_dispatch && !dispatch)) return fail(RIBBIT_E_ARG, "null argument");
    if (length > 0 && (!sequence || !hi || !lo || !brk)) return fail(RIBBIT_E_ARG, "null plane");
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
    std::string bed;
    bool in_pieces = false;
    if (std::getenv("RIBBIT_HOST_DEFER") != nullptr && defer_min_length() > 0 && n_dispatch) {
        // Test hook (no GPU needed): the recursion of the long-motif seeds cut into nodes, pieces and levels exactly as the GPU
        // path cuts it (refine.h: DeferredNode) -- nodes from RIBBIT_DEFER_MIN bases on are put off, every level is refined by
        // this function's own host code (which may put off the next one), and the pieces are sorted into place.  The BED must
        // not depend on any of it (tests/test_refine.py).
        std::vector<rb::DeferredNode> nodes, next;
        std::mutex lock;
        std::vector<rb::BedPiece> pieces;
        bool order_dependent = false;
        rb::Deferral tree = make_deferral(&nodes, &lock);
        tree.max_query = INT32_MAX; tree.max_ref = INT32_MAX;        // (no kernel to fit here)
        rb::refine_to_bed(hp, sequence, *prm, seeds, longest.data(), nullptr, sequence_id, bed, 0, nullptr, nullptr, nullptr, 0, (size_t)-1, &order_dependent,
                          nullptr, nullptr, nullptr, &pieces, nullptr, 0, &tree);
        int64_t levels = 0, put_off = 0;
        while (!order_dependent && !nodes.empty()) {
            const size_t n = nodes.size();
            rb::SeedVec ns(n);
            std::vector<int32_t> nl(n), nb(n);
            std::vector<uint32_t> all(n);
            for (size_t i = 0; i < n; ++i) { ns[i] = RibbitSeed{nodes[i].start, nodes[i].end, nodes[i].mlen, nodes[i].type}; nl[i] = nodes[i].longest; nb[i] = nodes[i].known_row; all[i] = (uint32_t)i; }
            next.clear();
            rb::Deferral d = make_deferral(&next, &lock);
            d.max_query = INT32_MAX; d.max_ref = INT32_MAX;
            d.nodes = nodes.data();
            // (no jobs: a level's nodes are aligned here; a node may put its own flanks off again -- not itself: it is this call's)
            d.min_length = levels % 3 == 2 ? INT32_MAX : d.min_length;      // every third level finishes its subtrees by recursion, as the GPU path's last level does
            std::string unused;
            rb::refine_to_bed(hp, sequence, *prm, ns, nl.data(), nb.data(), sequence_id, unused, 0, nullptr, nullptr, nullptr, 0, n, &order_dependent,
                              nullptr, nullptr, nullptr, &pieces, &all, 0, &d);
            ++levels; put_off += (int64_t)n;
            nodes.swap(next);
        }
        if (!order_dependent) {
            std::sort(pieces.begin(), pieces.end(), [](const rb::BedPiece &x, const rb::BedPiece &y) {
                return x.first_seed != y.first_seed ? x.first_seed < y.first_seed : x.path < y.path; });
            for (const rb::BedPiece &pc : pieces) bed += pc.text;
            in_pieces = true;
            g_level_counts[0] += levels; g_level_counts[1] += put_off;
        } else bed.clear();
    }
    if (!in_pieces) rb::refine_to_bed(hp, sequence, *prm, seeds, longest.data(), nullptr, sequence_id, bed);
    *len = bed.size();
    *text = (char *)std::malloc(bed.size() + 1);
    if (!*text) return fail(RIBBIT_E_NOMEM, "out of host memory");
    std::memcpy(*text, bed.c_str(), bed.size() + 1);
    return RIBBIT_OK;
}

void ribbit_text_free(char *text) { std::free(text); }

void ribbit_debug_level_counters(int64_t out[3]) {
    for (int k = 0; k < 3; ++k) out[k] = g_level_counts[k].load();
}

int ribbit_hip_adopt_dispatch(RibbitHandle *h, const RibbitSeed *seeds, size_t n) {
    if (!h || (n && !seeds)) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (h->pair_pending || h->copy_pending) return fail(RIBBIT_E_STATE, "a perfect scan is in flight on this handle");
    int rc;
    if ((rc = bind_device(h))) return rc;
    if ((rc = ensure_host_planes(h))) return rc;
    // the composed planes on THIS device (the scans of the seeds read them there): the planes kernel alone, no window scan
    if (!h->xa_on_device) {
        if ((rc = prepare_anchored(h))) return rc;
        rb::PerfectLaunch pp;
        pp.m_lo = h->params.min_motif;
        pp.m_hi = h->params.max_motif;
        pp.ev_cap = 0;
        rb::launch_scan_anchored(h->planes(), pp, h->d_xa.p, h->xa_stride, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->xa_on_device = true;
    }
    // No host copy of the composed planes is made: refinement reads them on the host only for the flank nodes' longest runs, and
    // HostPlanes recomputes the slice a query covers from the packed planes when it has no copy (0.6 us a query).
    if (h->stage_done < STAGE_ANCHORED) {
        if (h->xa_copy_pending) { HIP_TRY(hipEventSynchronize(h->ev_xa)); h->xa_copy_pending = false; }
        h->host.xa.clear(); h->host.xa_view = nullptr; h->host.xa_stride = 0;
        h->host.xa_m_lo = h->params.min_motif; h->host.xa_m_hi = h->params.max_motif;      // plane m IS the composed plane (fasta_utils.cpp:159), recomputed on request
    }
    // `seeds` may point into this handle's own dispatch list (a slice of it)
    rb::SeedVec taken(seeds, seeds + n);
    h->dispatch.swap(taken);
    h->longest_valid = false; h->best_rows_valid = false; h->small_valid = false;
    h->stage_done = STAGE_ANCHORED;
    return RIBBIT_OK;
}

int ribbit_hip_refine_met_empty_query(const RibbitHandle *h) { return h && h->refine_met_empty_query ? 1 : 0; }

}  // extern "C"
