// api_merge.cpp -- the anchored stage's merge as device work: what rb::AnchoredDevicePass::run (parallel_merge.h) does on a
// handle.  The ranges, their cuts and starting cursors come from the host's preparation (parallel_merge.cpp); the kept calls
// and the composed planes are in device memory already (window_stage_device, scan_anchored_kernel); the perfect and
// substitution lists travel up (0.1 GB for a chromosome), the ranges' parts of the anchored list and their logs travel down.
#include "api_internal.h"

namespace rbapi {

namespace {

// HIP_TRY returns an int status; inside run() a failure means "the host pass runs"
#define MERGE_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)fail(RIBBIT_E_INTERNAL, "%s: %s", #expr, hipGetErrorString(e_)); return false; } } while (0)

bool run_device_pass(RibbitHandle *h, RibbitCall *d_calls, const int32_t *d_pend, const rb::SeedLists &lists, const rb::KeptCalls &kc,
                     const std::vector<size_t> &first, const std::vector<int> &cut_pos, const std::vector<rb::Cursor2> &start_cursor,
                     const std::vector<uint32_t> &order, size_t device_limit, const std::function<void(uint32_t *, const uint32_t *)> &meanwhile,
                     std::vector<rb::AnchoredDevicePass::RangeResult> &ranges, std::vector<rb::AnchoredDevicePass::LogEntry> &undo,
                     std::vector<rb::AnchoredDevicePass::LogEntry> &reads, std::vector<rb::AnchoredDevicePass::HeadEntry> &heads) {
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    const double t0 = now_ms();
    RibbitHandle::MergeBufs &mg = h->mg;
    const size_t nr = cut_pos.size(), nP = lists.perfect.size(), nS = lists.subst.size(), n = kc.n;
    if (nr < 2 || first.size() != nr + 1 || start_cursor.size() != nr || first[nr] != n || order.size() != nr) return false;
    if (nP >= (1u << 30) || nS >= (1u << 30) || n + nr >= (1ull << 30) || nr >= (1u << 28)) return false;      // (the packing of candidates and log entries)
    if (!h->xa_on_device || !h->d_xa.p) return false;
    if (bind_device(h)) return false;
    const uint32_t head_cap = 1u << 16;
    const size_t log_cap = nP + nS + n / 4 + 65536;
    if (mg.d_perfect.ensure(std::max<size_t>(nP, 1)) || mg.d_subst.ensure(std::max<size_t>(nS, 1)) || mg.d_type0.ensure(nP + nS + 1) ||
        mg.d_own.ensure(n + nr) || mg.d_first.ensure(2 * nr + 1) || mg.d_cuts.ensure(3 * nr) || mg.d_range_out.ensure(nr * rb::AM_RANGE_OUT_WORDS) ||
        mg.d_log.ensure(4 * log_cap) || mg.d_head_log.ensure(8 * (size_t)head_cap) || mg.d_counts.ensure(4) ||
        mg.d_scratch.ensure(64 * (size_t)rb::AM_RESIDENT_WAVES * (size_t)rb::AM_SCRATCH_WORDS) || mg.h_own.ensure(n + nr) || mg.h_range_out.ensure(nr * rb::AM_RANGE_OUT_WORDS) ||
        mg.h_counts.ensure(4) || mg.h_head_log.ensure(8 * (size_t)head_cap))
        return false;
    if (!mg.h_sync.p) {
        if (mg.h_sync.ensure(16)) return false;
        MERGE_TRY(hipHostGetDevicePointer((void **)&mg.h_sync_dev, mg.h_sync.p, 0));
    }
    mg.h_sync.p[0] = (uint32_t)std::min(device_limit, order.size()); mg.h_sync.p[1] = 0;
    // ranges, cuts, cursors
    std::vector<uint32_t> first32(nr + 1);
    for (size_t k = 0; k <= nr; ++k) first32[k] = (uint32_t)first[k];
    std::vector<int32_t> cuts(3 * nr);
    for (size_t k = 0; k < nr; ++k) { cuts[k] = cut_pos[k]; cuts[nr + 2 * k] = start_cursor[k].perfect; cuts[nr + 2 * k + 1] = start_cursor[k].subst; }
    hipStream_t st = h->stream;
    MERGE_TRY(hipMemcpyAsync(mg.d_first.p, first32.data(), (nr + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    MERGE_TRY(hipMemcpyAsync(mg.d_first.p + nr + 1, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    MERGE_TRY(hipMemsetAsync(mg.d_range_out.p, 0xff, nr * rb::AM_RANGE_OUT_WORDS * sizeof(uint32_t), st));      // (status of a range nobody merged: all ones)
    MERGE_TRY(hipMemcpyAsync(mg.d_cuts.p, cuts.data(), 3 * nr * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (nP) MERGE_TRY(hipMemcpyAsync(mg.d_perfect.p, lists.perfect.data(), nP * sizeof(RibbitSeed), hipMemcpyHostToDevice, st));
    if (nS) MERGE_TRY(hipMemcpyAsync(mg.d_subst.p, lists.subst.data(), nS * sizeof(RibbitSeed), hipMemcpyHostToDevice, st));
    // (the host threads retire seeds in these lists while the kernel runs: the copies must have left them by then)
    MERGE_TRY(hipStreamSynchronize(st));
    rb::launch_seed_types(mg.d_perfect.p, (uint32_t)nP, mg.d_type0.p, st);
    rb::launch_seed_types(mg.d_subst.p, (uint32_t)nS, mg.d_type0.p + nP, st);
    MERGE_TRY(hipMemsetAsync(mg.d_counts.p, 0, 4 * sizeof(uint32_t), st));
    rb::AnchoredMergeArgs a{};
    a.P = mg.d_perfect.p; a.S = mg.d_subst.p; a.nP = (uint32_t)nP; a.nS = (uint32_t)nS;
    a.P_type0 = mg.d_type0.p; a.S_type0 = mg.d_type0.p + nP;
    a.calls = d_calls; a.pend = d_pend;
    a.first = mg.d_first.p; a.cut_pos = mg.d_cuts.p; a.cur0 = mg.d_cuts.p + nr; a.nr = (uint32_t)nr;
    a.xa = h->d_xa.p; a.xa_stride = h->xa_stride; a.length = lists.length; a.m_lo = h->params.min_motif; a.m_hi = h->params.max_motif;
    a.own = mg.d_own.p; a.range_out = mg.d_range_out.p;
    a.log = mg.d_log.p; a.log_count = mg.d_counts.p; a.log_cap = (uint32_t)std::min<size_t>(log_cap, 0xffffffffu);
    a.head_log = mg.d_head_log.p; a.head_count = mg.d_counts.p + 1; a.head_cap = head_cap;
    a.scratch = mg.d_scratch.p; a.next_range = mg.d_counts.p + 2;
    a.order = mg.d_first.p + nr + 1; a.n_order = (uint32_t)order.size(); a.sync = mg.h_sync_dev;
    const double t1 = now_ms();
    a.max_passes = rb::AM_MAX_PASSES;
    rb::launch_anchored_merge(a, (uint32_t)n, (uint32_t)rb::AM_RESIDENT_WAVES, st);
    MERGE_TRY(hipGetLastError());
    meanwhile(mg.h_sync.p, mg.h_sync.p + 1);         // the host threads' share of the ranges, while the kernel runs
    const double t1b = now_ms();
    MERGE_TRY(hipMemcpyAsync(mg.h_counts.p, mg.d_counts.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MERGE_TRY(hipMemcpyAsync(mg.h_range_out.p, mg.d_range_out.p, nr * rb::AM_RANGE_OUT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MERGE_TRY(hipMemcpyAsync(mg.h_own.p, mg.d_own.p, (n + nr) * sizeof(RibbitSeed), hipMemcpyDeviceToHost, st));
    MERGE_TRY(hipStreamSynchronize(st));
    const double t2 = now_ms();
    const uint32_t n_log = mg.h_counts.p[0], n_head = mg.h_counts.p[1];
    if (n_log > a.log_cap || n_head > head_cap) return false;            // (every range that met the full log says so too; nothing was applied to the host lists)
    if (mg.h_log.ensure(4 * std::max<size_t>(n_log, 1))) return false;
    if (n_log) MERGE_TRY(hipMemcpyAsync(mg.h_log.p, mg.d_log.p, 4 * (size_t)n_log * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (n_head) MERGE_TRY(hipMemcpyAsync(mg.h_head_log.p, mg.d_head_log.p, 8 * (size_t)n_head * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MERGE_TRY(hipStreamSynchronize(st));
    ranges.resize(nr);
    for (size_t k = 0; k < nr; ++k) {
        const uint32_t *o = mg.h_range_out.p + k * rb::AM_RANGE_OUT_WORDS;
        rb::AnchoredDevicePass::RangeResult &r = ranges[k];
        r.own = mg.h_own.p + first[k] + k;
        r.own_n = o[0]; r.status = o[1];
        if (!r.status && r.own_n > first[k + 1] - first[k] + 1) r.status |= 0x80000000u;
        r.guard_hits = (int64_t)(((uint64_t)o[3] << 32) | o[2]);
        r.cursor.perfect = (int)o[4]; r.cursor.subst = (int)o[5];
        r.head_reads[0] = ((uint64_t)o[7] << 32) | o[6];
        r.head_reads[1] = ((uint64_t)o[9] << 32) | o[8];
    }
    undo.clear(); reads.clear(); heads.clear();
    for (uint32_t i = 0; i < n_log; ++i) {
        const uint32_t *e = mg.h_log.p + 4 * (size_t)i;
        const uint32_t kind = e[0] >> 28, range = e[0] & 0x0fffffffu;
        if (range >= nr) return false;
        const rb::AnchoredDevicePass::LogEntry le{range, e[1] >> 31, e[1] & 0x7fffffffu, (int32_t)e[2]};
        if (le.index >= (le.list ? nS : nP)) return false;
        if (kind == rb::AM_LOG_UNDO) undo.push_back(le);
        else if (kind == rb::AM_LOG_READ) reads.push_back(le);
        else return false;
    }
    for (uint32_t i = 0; i < n_head; ++i) {
        const uint32_t *e = mg.h_head_log.p + 8 * (size_t)i;
        if (e[0] >= nr || e[1] > 1 || e[2] >= (e[1] ? nS : nP)) return false;
        heads.push_back({e[0], e[1], e[2], RibbitSeed{(int32_t)e[3], (int32_t)e[4], (int32_t)e[5], (int32_t)e[6]}, e[7] != 0});
    }
    if (profile) {
        size_t slow_k = 0, merged = 0, too_slow = 0, too_many = 0, calls_on_device = 0;
        uint64_t ticks_sum = 0, passes_sum = 0; uint32_t ticks_max = 0;
        for (size_t k = 0; k < nr; ++k) {
            const uint32_t *o = mg.h_range_out.p + k * rb::AM_RANGE_OUT_WORDS;
            if (o[1] == 0xffffffffu) continue;                     // not the device's
            too_slow += (o[1] & rb::AM_TOO_SLOW) != 0; too_many += (o[1] & rb::AM_SCRATCH_FULL) != 0;
            if (o[1]) continue;
            ++merged; calls_on_device += first[k + 1] - first[k];
            ticks_sum += o[11]; passes_sum += o[10];
            if (o[11] > ticks_max) { ticks_max = o[11]; slow_k = k; }
        }
        std::fprintf(stderr, "[anchored merge on the device] %zu calls in %zu ranges, %zu of them within a lane's reach: lists up and set-up %.1f ms, the host threads' share while the kernel ran %.1f ms, "
                     "%.1f ms more for the kernel and its results, logs %.1f ms (%u entries: %zu retirements, %zu reads left of a range; %u list-head writes).  Merged on the device: %zu ranges, %zu calls; "
                     "left to the host: %zu ranges (%zu over their budget of passes, %zu with more candidates than the LDS holds).  A range took %.2f ms on average (%.0f passes of its lane's loop), the slowest %.2f ms (%zu calls)\n",
                     n, nr, std::min(device_limit, order.size()), t1 - t0, t1b - t1, t2 - t1b, now_ms() - t2, n_log, undo.size(), reads.size(), n_head, merged, calls_on_device, too_slow + too_many, too_slow, too_many,
                     merged ? (double)ticks_sum * 1e-5 / (double)merged : 0.0, merged ? (double)passes_sum / (double)merged : 0.0, ticks_max * 1e-5, first[slow_k + 1] - first[slow_k]);
    }
    return true;
}

}  // namespace

rb::AnchoredDevicePass anchored_device_pass(RibbitHandle *h, RibbitCall *d_calls, const int32_t *d_pend) {
    rb::AnchoredDevicePass p;
    if (const char *e = std::getenv("RIBBIT_DEVICE_MERGE_MIN")) p.min_calls = (size_t)std::strtoull(e, nullptr, 10);        // tuning / tests
    if (const char *e = std::getenv("RIBBIT_DEVICE_MERGE_RANGE")) p.calls_per_range = std::max<size_t>(1, (size_t)std::strtoull(e, nullptr, 10));
    p.run = [h, d_calls, d_pend](const rb::SeedLists &lists, const rb::KeptCalls &kc, const std::vector<size_t> &first, const std::vector<int> &cut_pos,
                                 const std::vector<rb::Cursor2> &start_cursor, const std::vector<uint32_t> &order, size_t device_limit, const std::function<void(uint32_t *, const uint32_t *)> &meanwhile,
                                 std::vector<rb::AnchoredDevicePass::RangeResult> &ranges,
                                 std::vector<rb::AnchoredDevicePass::LogEntry> &undo, std::vector<rb::AnchoredDevicePass::LogEntry> &reads,
                                 std::vector<rb::AnchoredDevicePass::HeadEntry> &heads) {
        return run_device_pass(h, d_calls, d_pend, lists, kc, first, cut_pos, start_cursor, order, device_limit, meanwhile, ranges, undo, reads, heads);
    };
    return p;
}

}  // namespace rbapi
