// api_perfect.cpp -- the perfect stage (a3, a4) and its chunk form; see api_internal.h for the map of the files behind include/ribbit_hip.h.
// There is no CPU fallback for any scan anywhere in this library.
#include "api_internal.h"

namespace rbapi {

// Launch one scan kernel, compact its sharded event regions, copy the events back and index the
// (motif, tile) chunks.  which: 0 perfect run scan (the only stage whose events ever travel: ribbit_hip_perfect_runs_partial).
int collect_events(RibbitHandle *h, int which) {
    int rc;
    if ((rc = bind_device(h))) return rc;
    if ((rc = h->d_counters.ensure(rb::EV_COUNTER_WORDS))) return rc;
    if ((rc = h->h_counters.ensure(rb::EV_COUNTER_WORDS))) return rc;
    // capacity in events, split evenly over EV_SHARDS regions; grows on overflow
    // typical event densities on repeat-rich sequence: 0.07 per base (perfect), 0.25 (1-mismatch windows), 3.7 (anchored
    // windows at 99 motif sizes); a too small first guess costs a second launch and a second round of allocations
    const size_t per_base_x4 = which == 0 ? 1 : which == 1 ? 2 : (size_t)std::max(16, (h->params.max_motif - h->params.min_motif + 1) / 6);
    size_t cap = std::max<size_t>((size_t)1 << 20, (size_t)h->length * per_base_x4 / 4);
    cap = std::max(cap, h->d_events.cap);
    if (h->debug_first_cap) cap = h->debug_first_cap;
    const rb::DevicePlanes pl = h->planes();
    uint32_t produced = 0;
    for (int attempt = 0;; ++attempt) {
        cap = std::min<size_t>((cap + rb::EV_SHARDS - 1) / rb::EV_SHARDS * rb::EV_SHARDS, 0xffffff00u);
        if ((rc = h->d_events.ensure(cap))) return rc;
        if ((rc = h->d_dense.ensure(cap))) return rc;
        HIP_TRY(hipEventRecord(h->ev[4], h->stream));
        HIP_TRY(hipMemsetAsync(h->d_counters.p, 0, rb::EV_COUNTER_WORDS * sizeof(uint32_t), h->stream));
        h->counters_clean = false;
        rb::PerfectLaunch pp;
        pp.m_lo = h->params.min_motif;
        pp.m_hi = h->params.max_motif;
        pp.ev_cap = (uint32_t)cap;
        HIP_TRY(hipEventRecord(h->ev[2], h->stream));
        if (which != 0) return fail(RIBBIT_E_INTERNAL, "collect_events: the window stages' events stay on the device (window_stage.hip)");
        rb::launch_scan_perfect(pl, pp, h->d_events.p, h->d_counters.p, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(h->ev[3], h->stream));
        rb::launch_compact_events(h->d_events.p, pp.ev_cap, h->d_counters.p, h->d_dense.p, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->h_counters.p, h->d_counters.p, rb::EV_COUNTER_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        produced = h->h_counters.p[rb::EV_SUMMARY];
        if (!h->h_counters.p[rb::EV_SUMMARY + 1]) break;
        // some region overflowed: size every region for the fullest one and retry
        uint32_t worst = 0;
        for (int t = 0; t < rb::EV_SHARDS; ++t) worst = std::max(worst, h->h_counters.p[t * rb::EV_COUNTER_STRIDE]);
        if (attempt == 2 || (size_t)worst * rb::EV_SHARDS > 0xffffff00u)
            return fail(RIBBIT_E_OVERFLOW, "event buffer overflow: fullest region needs %u events", worst);
        cap = ((size_t)worst + 1024) * rb::EV_SHARDS;
    }
    h->last_event_count = produced;
    h->produced = produced;
    if ((rc = h->h_events.ensure(std::max<size_t>(produced, 1)))) return rc;
    // Events arrive as position-ordered chunks, exactly one per (motif, tile) that has any event.  A kernel indexes
    // them in a direct-address table keyed (motif, tile) -- every event looks at its neighbours -- so the host
    // neither sorts nor walks the events to find the chunks.
    const uint32_t m_lo = (uint32_t)h->params.min_motif;
    const size_t nm = (size_t)(h->params.max_motif - h->params.min_motif + 1);
    const uint32_t tile_bases = which == 2 ? (uint32_t)rb::anchored_tile_words(rb::anchored_halo_lanes(h->params.max_motif)) * 32u
                                           : (uint32_t)rb::TILE_BASES;
    const size_t ntile = (size_t)(h->length / tile_bases + 1);
    if (nm * ntile > 0xfffffff0u) return fail(RIBBIT_E_ARG, "record too long for %zu motif sizes", nm);
    if ((rc = h->d_pair_table.ensure(nm * ntile))) return rc;
    if ((rc = h->d_pair_status.ensure(rb::PAIR_STATUS_WORDS))) return rc;
    HIP_TRY(rb::launch_chunk_table(h->d_dense.p, h->d_counters.p, m_lo, (uint32_t)nm, (uint32_t)ntile, tile_bases, h->d_pair_table.p, h->d_pair_status.p, h->stream));
    h->chunk_table.resize(nm * ntile);
    h->table_ntile = ntile;
    uint32_t table_status = 0;
    HIP_TRY(hipMemcpyAsync(h->chunk_table.data(), h->d_pair_table.p, nm * ntile * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&table_status, h->d_pair_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    if (produced) {
        HIP_TRY(hipMemcpyAsync(h->h_events.p, h->d_dense.p, (size_t)produced * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipEventRecord(h->ev[5], h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_timing[1] = h->have_timing[2] = true;
    if (table_status & 1u) return fail(RIBBIT_E_INTERNAL, "malformed event (motif or tile outside the launch)");
    if (table_status & 2u) return fail(RIBBIT_E_INTERNAL, "duplicate event chunk");
    // {first + 1, end}  ->  {offset, count}
    struct Chunk { uint32_t off, n; };
    static_assert(sizeof(Chunk) == sizeof(uint64_t), "chunk table entry is one 64-bit word");
    Chunk *table = reinterpret_cast<Chunk *>(h->chunk_table.data());
    for (size_t k = 0; k < nm * ntile; ++k) {
        const uint32_t first1 = table[k].off, end = table[k].n;
        table[k] = first1 ? Chunk{first1 - 1u, end - (first1 - 1u)} : Chunk{0, 0};
    }
    return RIBBIT_OK;
}

rb::EventSource event_source(const RibbitHandle *h) {
    rb::EventSource src;
    src.ev = h->h_events.p;
    src.segs = reinterpret_cast<const rb::Seg *>(h->chunk_table.data());
    src.segs_per_motif = h->table_ntile;
    src.nm = (size_t)(h->params.max_motif - h->params.min_motif + 1);
    src.m_lo = h->params.min_motif;
    return src;
}

// Perfect stage on the device end to end: scan kernel -> START/END events (left in their regions, never
// copied to the host) -> pairing kernels -> RibbitRun records ordered by (motif, start) -> one D2H copy
// into pinned memory.  The host only checks the counters and the pairing status.
int perfect_wait(RibbitHandle *h) {
    if (!h->copy_pending) return RIBBIT_OK;
    h->copy_pending = false;
    int rc;
    if ((rc = bind_device(h))) return rc;
    HIP_TRY(hipStreamSynchronize(h->copy_stream));
    return RIBBIT_OK;
}

// The perfect stage in two halves, so that a caller with several handles can keep one record's kernels running
// while another record's results travel to the host (each handle has its own stream):
//   perfect_enqueue: memset + scan + pairing kernels + D2H of counters and status, no synchronisation;
//   perfect_finish:  waits for those, grows the event buffer and repeats on overflow, then copies the run records.
// own_lo/own_hi/pos_offset: see rb::PairLaunch (a whole record is 0, INT64_MAX, 0).
int perfect_enqueue(RibbitHandle *h, size_t cap) {
    int rc;
    if ((rc = bind_device(h))) return rc;
    rb::PairLaunch &pr = h->pair;
    cap = std::min<size_t>((cap + rb::EV_SHARDS - 1) / rb::EV_SHARDS * rb::EV_SHARDS, 0xffffff00u);
    if ((rc = h->d_events.ensure(cap))) return rc;
    if ((rc = h->d_dense.ensure(cap))) return rc;        // cap/2 runs of 16 bytes
    if (h->timing) HIP_TRY(hipEventRecord(h->ev[4], h->stream));
    if (!h->counters_clean) HIP_TRY(hipMemsetAsync(h->d_counters.p, 0, rb::EV_COUNTER_WORDS * sizeof(uint32_t), h->stream));
    h->counters_clean = false;
    rb::PerfectLaunch pp;
    pp.m_lo = h->params.min_motif;
    pp.m_hi = h->params.max_motif;
    pp.ev_cap = (uint32_t)cap;
    pr.region_cap = pp.ev_cap / (uint32_t)rb::EV_SHARDS;
    if (h->timing) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    rb::launch_scan_perfect(h->planes(), pp, h->d_events.p, h->d_counters.p, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    // Everything after the scan (nine small, latency-bound launches, later the result copy) runs on the handle's
    // own post stream: on a compute stream shared by several handles the next record's pack and scan start right
    // behind this scan instead of waiting out the pairing chain's launch gaps.
    HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->ev[3], 0));
    HIP_TRY(rb::launch_pair_runs(h->d_events.p, h->d_counters.p, pr, h->d_pair_table.p, h->d_run_base.p, h->d_pair_partial.p,
                                 h->d_dense.p, (uint32_t)(cap / 2), h->d_halves.p, (uint32_t)(2 * (size_t)pr.nm), h->d_pair_status.p, h->copy_stream));
    rb::launch_pair_publish(h->d_counters.p, h->d_pair_status.p, h->h_pub_dev, h->copy_stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_ready, h->copy_stream));
    return RIBBIT_OK;
}

int perfect_begin(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset) {
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (h->copy_pending) { int rcw = perfect_wait(h); if (rcw) return rcw; }
    h->runs_valid = h->calls_valid = false;
    h->pair_pending = false;
    int rc;
    if ((rc = h->d_counters.ensure(rb::EV_COUNTER_WORDS))) return rc;
    if ((rc = h->h_counters.ensure(rb::EV_COUNTER_WORDS))) return rc;
    if ((rc = h->d_pair_status.ensure(rb::PAIR_STATUS_WORDS))) return rc;
    if (!h->h_pub.p) {
        if ((rc = h->h_pub.ensure(rb::EV_SHARDS + rb::PAIR_STATUS_WORDS))) return rc;
        HIP_TRY(hipHostGetDevicePointer((void **)&h->h_pub_dev, h->h_pub.p, 0));
    }
    rb::PairLaunch &pr = h->pair;
    pr.m_lo = (uint32_t)h->params.min_motif;
    pr.nm = (uint32_t)(h->params.max_motif - h->params.min_motif + 1);
    pr.ntile = (uint32_t)(h->length / rb::TILE_BASES + 1);
    pr.tile_bases = (uint32_t)rb::TILE_BASES;
    pr.own_lo = own_lo; pr.own_hi = own_hi; pr.pos_offset = pos_offset;
    const size_t half_cap = 2 * (size_t)pr.nm;
    if ((rc = h->d_halves.ensure(half_cap))) return rc;
    if ((rc = h->h_halves.ensure(half_cap))) return rc;
    const size_t entries = (size_t)pr.nm * pr.ntile;
    if (entries > 0xfffffff0u) return fail(RIBBIT_E_ARG, "record too long for %u motif sizes", pr.nm);
    if ((rc = h->d_pair_table.ensure(entries))) return rc;
    if ((rc = h->d_run_base.ensure(entries))) return rc;
    if ((rc = h->d_pair_partial.ensure(entries / 1024 + 2))) return rc;
    const size_t cap = h->debug_first_cap ? h->debug_first_cap
                                          : std::max(std::max<size_t>((size_t)1 << 20, (size_t)(h->length / 4)), h->d_events.cap);
    if ((rc = perfect_enqueue(h, cap))) return rc;
    h->pair_pending = true;
    return RIBBIT_OK;
}

// the scan in flight is complete on the device: counts known, overflow handled (the scan is run again with more room),
// pairing checked.  The run records are in d_dense, the cut ones in d_halves.
int perfect_collect(RibbitHandle *h) {
    if (!h->pair_pending) return fail(RIBBIT_E_STATE, "no perfect scan in flight on this handle");
    h->pair_pending = false;
    int rc;
    if ((rc = bind_device(h))) return rc;
    const rb::PairLaunch &pr = h->pair;
    uint64_t produced = 0;
    for (int attempt = 0;; ++attempt) {
        // wait for THIS record's kernels only: the stream may be shared with other handles whose kernels come later
        HIP_TRY(hipEventSynchronize(h->ev_ready));
        uint32_t worst = 0;
        produced = 0;
        for (int t = 0; t < rb::EV_SHARDS; ++t) {
            const uint32_t c = h->h_pub.p[t];
            worst = std::max(worst, c);
            produced += c;
        }
        if (worst <= pr.region_cap) break;
        // some region overflowed: size every region for the fullest one and run again
        if (attempt == 2 || (size_t)worst * rb::EV_SHARDS > 0xffffff00u)
            return fail(RIBBIT_E_OVERFLOW, "event buffer overflow: fullest region needs %u events", worst);
        if ((rc = perfect_enqueue(h, ((size_t)worst + 1024) * rb::EV_SHARDS))) return rc;
    }
    h->last_event_count = (int64_t)produced;
    const uint32_t flags = h->h_pub.p[rb::EV_SHARDS + rb::PAIR_FLAGS];
    if (flags) {
        return fail(RIBBIT_E_INTERNAL, "run pairing failed (flags 0x%x):%s%s%s%s%s", flags,
                    flags & rb::PAIR_BAD_EVENT ? " malformed event;" : "", flags & rb::PAIR_DUP_CHUNK ? " duplicate event chunk;" : "",
                    flags & rb::PAIR_NOT_ALTERNATING ? " run starts and ends do not alternate;" : "",
                    flags & rb::PAIR_UNTERMINATED ? " unterminated run;" : "", flags & rb::PAIR_NO_ROOM ? " run buffer too small;" : "");
    }
    h->n_runs = h->h_pub.p[rb::EV_SHARDS + rb::PAIR_TOTAL];
    if (h->n_runs * 2 != produced) return fail(RIBBIT_E_INTERNAL, "%llu events but %zu runs", (unsigned long long)produced, h->n_runs);
    h->n_halves = h->h_pub.p[rb::EV_SHARDS + rb::PAIR_HALVES];
    return RIBBIT_OK;
}

int perfect_finish(RibbitHandle *h, RibbitRun *dst, size_t dst_cap, RibbitRun *half_dst, size_t half_dst_cap, bool wait) {
    int rc = perfect_collect(h);
    if (rc) return rc;
    const rb::PairLaunch &pr = h->pair;
    // everything that can fail is checked before the first copy is enqueued: an error return must not leave a DMA in flight
    // into a buffer the caller may free
    if (half_dst && h->n_halves > half_dst_cap) return fail(RIBBIT_E_OVERFLOW, "%zu half records do not fit the caller's buffer of %zu", h->n_halves, half_dst_cap);
    if (dst && h->n_runs > dst_cap) return fail(RIBBIT_E_OVERFLOW, "%zu run records do not fit the caller's buffer of %zu", h->n_runs, dst_cap);
    const bool whole = pr.own_lo == 0 && pr.own_hi == INT64_MAX && pr.pos_offset == 0 && !dst;
    if (!dst) {
        if ((rc = h->h_runs.ensure(std::max<size_t>(h->n_runs, 1)))) return rc;
        dst = h->h_runs.p;
    }
    if (!half_dst) half_dst = h->h_halves.p;
    if (h->n_halves)
        HIP_TRY(hipMemcpyAsync(half_dst, h->d_halves.p, h->n_halves * sizeof(RibbitRun), hipMemcpyDeviceToHost, h->copy_stream));
    h->copy_pending = h->n_halves != 0;      // from here on a failure leaves the wait to the next call on the handle
    if (h->n_runs)
        HIP_TRY(hipMemcpyAsync(dst, h->d_dense.p, h->n_runs * sizeof(RibbitRun), hipMemcpyDeviceToHost, h->copy_stream));
    if (h->timing) HIP_TRY(hipEventRecord(h->ev[5], h->copy_stream));
    h->have_timing[1] = h->have_timing[2] = h->timing;
    h->host_ms = 0.0;
    h->runs_valid = whole;
    h->copy_pending = true;
    return wait ? perfect_wait(h) : RIBBIT_OK;
}

int run_perfect_scan_range(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset, RibbitRun *dst, size_t dst_cap,
                           RibbitRun *half_dst, size_t half_dst_cap) {
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    const bool whole = own_lo == 0 && own_hi == INT64_MAX && pos_offset == 0 && !dst;
    if (whole && h->runs_valid) return RIBBIT_OK;
    int rc = perfect_begin(h, own_lo, own_hi, pos_offset);
    if (rc) return rc;
    return perfect_finish(h, dst, dst_cap, half_dst, half_dst_cap);
}

int run_perfect_scan(RibbitHandle *h) { return run_perfect_scan_range(h, 0, INT64_MAX, 0, nullptr, 0); }

int build_perfect_calls(RibbitHandle *h) {
    if (h->calls_valid) return RIBBIT_OK;
    int rc = run_perfect_scan(h);
    if (rc) return rc;
    rb::perfect_calls_from_runs(h->h_runs.p, h->n_runs, h->length, h->min_shift, h->perfect_calls);
    h->calls_valid = true;
    return RIBBIT_OK;
}

int advance_to_perfect(RibbitHandle *h) {
    if (h->stage_done >= STAGE_PERFECT) return RIBBIT_OK;
    const double t0 = now_ms();
    int rc = build_perfect_calls(h);
    if (rc) return rc;
    const double t1 = now_ms();
    h->lists.perfect.clear();
    for (const RibbitCall &c : h->perfect_calls) rb::perfect_add(h->lists, c.start, c.end, c.mlen);
    h->stage_done = STAGE_PERFECT;
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    if (profile) std::fprintf(stderr, "[perfect stage] scan, pairing, runs and planes to the host, calls %.1f ms; merge of %zu calls into %zu seeds on one thread %.1f ms\n",
                              t1 - t0, h->perfect_calls.size(), h->lists.perfect.size(), now_ms() - t1);
    return RIBBIT_OK;
}

}  // namespace rbapi

extern "C" {

int ribbit_hip_scan_perfect_runs(RibbitHandle *h, const RibbitRun **out, size_t *n) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    h->runs_valid = false;   // an explicit scan call always relaunches the kernel
    h->calls_valid = false;
    int rc = run_perfect_scan(h);
    if (rc) return rc;
    *out = h->h_runs.p;
    *n = h->n_runs;
    return RIBBIT_OK;
}

int ribbit_hip_perfect_calls(RibbitHandle *h, const RibbitCall **out, size_t *n) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    int rc = build_perfect_calls(h);
    if (rc) return rc;
    *out = h->perfect_calls.data();
    *n = h->perfect_calls.size();
    return RIBBIT_OK;
}

int ribbit_hip_seeds_perfect(RibbitHandle *h, const RibbitSeed **out, size_t *n) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (h->stage_done > STAGE_PERFECT) return fail(RIBBIT_E_STATE, "a later stage already re-typed the perfect list; reload the record");
    int rc = advance_to_perfect(h);
    if (rc) return rc;
    *out = h->lists.perfect.data();
    *n = h->lists.perfect.size();
    return RIBBIT_OK;
}

int ribbit_hip_perfect_runs_partial(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset,
                                    const RibbitRun **runs, size_t *n_runs, const uint64_t **halves, size_t *n_halves) {
    if (!h || !runs || !n_runs || !halves || !n_halves) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    int rc = collect_events(h, 0);
    if (rc) return rc;
    h->runs_valid = h->calls_valid = false;
    const double t0 = now_ms();
    std::string why;
    if (!rb::pair_perfect_runs_partial(event_source(h), own_lo, own_hi, pos_offset, h->runs, h->export_events, &why))
        return fail(RIBBIT_E_INTERNAL, "%s", why.c_str());
    h->host_ms = now_ms() - t0;
    *runs = h->runs.data();
    *n_runs = h->runs.size();
    *halves = h->export_events.data();
    *n_halves = h->export_events.size();
    return RIBBIT_OK;
}

int ribbit_hip_scan_perfect_chunk(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset, RibbitRun *dst, size_t dst_cap,
                                  RibbitRun *half_dst, size_t half_dst_cap, const RibbitRun **out, size_t *n,
                                  const RibbitRun **halves, size_t *n_halves) {
    if (!h || !out || !n || !halves || !n_halves) return fail(RIBBIT_E_ARG, "null argument");
    if (own_lo < 0 || own_hi < own_lo) return fail(RIBBIT_E_ARG, "bad own range");
    int rc = run_perfect_scan_range(h, own_lo, own_hi, pos_offset, dst, dst_cap, half_dst, half_dst_cap);
    if (rc) return rc;
    *out = dst ? dst : h->h_runs.p;
    *n = h->n_runs;
    *halves = half_dst ? half_dst : h->h_halves.p;
    *n_halves = h->n_halves;
    return RIBBIT_OK;
}

int ribbit_hip_scan_perfect_begin(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset) {
    if (!h) return fail(RIBBIT_E_ARG, "null argument");
    if (own_lo < 0 || own_hi < own_lo) return fail(RIBBIT_E_ARG, "bad own range");
    return perfect_begin(h, own_lo, own_hi, pos_offset);
}

int ribbit_hip_scan_perfect_end_device(RibbitHandle *h, const void **dev_runs, size_t *n, const void **dev_halves, size_t *n_halves) {
    if (!h || !dev_runs || !n || !dev_halves || !n_halves) return fail(RIBBIT_E_ARG, "null argument");
    int rc = perfect_collect(h);
    if (rc) return rc;
    h->runs_valid = false;
    h->have_timing[1] = h->timing;
    h->have_timing[2] = false;
    *dev_runs = h->d_dense.p;
    *n = h->n_runs;
    *dev_halves = h->d_halves.p;
    *n_halves = h->n_halves;
    return RIBBIT_OK;
}

int ribbit_hip_scan_perfect_wait(RibbitHandle *h) {
    if (!h) return fail(RIBBIT_E_ARG, "null argument");
    return perfect_wait(h);
}

int ribbit_hip_scan_perfect_end(RibbitHandle *h, RibbitRun *dst, size_t dst_cap, RibbitRun *half_dst, size_t half_dst_cap,
                                int wait, const RibbitRun **out, size_t *n, const RibbitRun **halves, size_t *n_halves) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    int rc = perfect_finish(h, dst, dst_cap, half_dst, half_dst_cap, wait != 0);
    if (rc) return rc;
    *out = dst ? dst : h->h_runs.p;
    *n = h->n_runs;
    if (halves) *halves = half_dst ? half_dst : h->h_halves.p;
    if (n_halves) *n_halves = h->n_halves;
    return RIBBIT_OK;
}

int ribbit_hip_debug_pair_events(RibbitHandle *h, const uint64_t *events, size_t n, int64_t length, RibbitRun *runs, size_t runs_cap,
                                 size_t *n_runs, uint32_t *flags) {
    if (!h || (n && !events) || !n_runs || !flags || (runs_cap && !runs)) return fail(RIBBIT_E_ARG, "null argument");
    if (length < 0 || n > ((size_t)1 << 24)) return fail(RIBBIT_E_ARG, "bad size");
    int rc;
    if ((rc = bind_device(h))) return rc;
    rb::PairLaunch pr{};
    pr.m_lo = (uint32_t)h->params.min_motif;
    pr.nm = (uint32_t)(h->params.max_motif - h->params.min_motif + 1);
    pr.ntile = (uint32_t)(length / rb::TILE_BASES + 1);
    pr.tile_bases = (uint32_t)rb::TILE_BASES;
    pr.own_lo = 0; pr.own_hi = INT64_MAX; pr.pos_offset = 0;
    pr.region_cap = (uint32_t)std::max<size_t>(n, 1);                  // the whole stream sits in region 0
    const size_t entries = (size_t)pr.nm * pr.ntile, cap = (size_t)pr.region_cap * rb::EV_SHARDS;
    DevBuf<uint64_t> d_ev, d_table, d_runs;
    DevBuf<uint32_t> d_cnt, d_base, d_part, d_status;
    DevBuf<RibbitRun> d_half;
    if ((rc = d_ev.ensure(cap)) || (rc = d_table.ensure(entries)) || (rc = d_runs.ensure(cap)) || (rc = d_cnt.ensure(rb::EV_COUNTER_WORDS)) ||
        (rc = d_base.ensure(entries)) || (rc = d_part.ensure(entries / 1024 + 2)) || (rc = d_status.ensure(rb::PAIR_STATUS_WORDS)) ||
        (rc = d_half.ensure(2 * (size_t)pr.nm)))
        return rc;
    std::vector<uint32_t> counters(rb::EV_COUNTER_WORDS, 0), status(rb::PAIR_STATUS_WORDS, 0);
    counters[0] = (uint32_t)n;
    int ret = RIBBIT_OK;
    do {
        if (n && hipMemcpy(d_ev.p, events, n * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) { ret = fail(RIBBIT_E_DEVICE, "copy failed"); break; }
        if (hipMemcpy(d_cnt.p, counters.data(), counters.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) { ret = fail(RIBBIT_E_DEVICE, "copy failed"); break; }
        if (rb::launch_pair_runs(d_ev.p, d_cnt.p, pr, d_table.p, d_base.p, d_part.p, d_runs.p, (uint32_t)(cap / 2), d_half.p, 2 * pr.nm, d_status.p, h->stream) != hipSuccess) {
            ret = fail(RIBBIT_E_DEVICE, "pairing kernels could not be launched");
            break;
        }
        if (hipStreamSynchronize(h->stream) != hipSuccess || hipMemcpy(status.data(), d_status.p, status.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            ret = fail(RIBBIT_E_DEVICE, "pairing kernels failed");
            break;
        }
        *flags = status[rb::PAIR_FLAGS];
        *n_runs = status[rb::PAIR_TOTAL];
        const size_t take = std::min(*n_runs, runs_cap);
        if (take && hipMemcpy(runs, d_runs.p, take * sizeof(RibbitRun), hipMemcpyDeviceToHost) != hipSuccess) ret = fail(RIBBIT_E_DEVICE, "copy failed");
    } while (false);
    d_ev.release(); d_table.release(); d_runs.release(); d_cnt.release(); d_base.release(); d_part.release(); d_status.release(); d_half.release();
    return ret;
}

int ribbit_host_perfect_runs_from_events(const RibbitScanParams *params, size_t nparts, const uint64_t *events,
                                         const uint64_t *counts, RibbitRun **runs, size_t *n) {
    if (!params || !counts || !runs || !n) return fail(RIBBIT_E_ARG, "null argument");
    const size_t nm = (size_t)(params->max_motif - params->min_motif + 1);
    std::vector<rb::Seg> segs(nm * nparts, rb::Seg{0, 0});
    uint64_t off = 0;
    for (size_t p = 0; p < nparts; ++p)
        for (size_t mi = 0; mi < nm; ++mi) { segs[mi * nparts + p] = rb::Seg{(uint32_t)off, (uint32_t)counts[p * nm + mi]}; off += counts[p * nm + mi]; }
    rb::EventSource src;
    src.ev = events; src.segs = segs.data(); src.segs_per_motif = nparts; src.nm = nm; src.m_lo = params->min_motif;
    std::vector<RibbitRun> out;
    std::string why;
    if (!rb::pair_perfect_runs(src, out, &why)) return fail(RIBBIT_E_INTERNAL, "perfect events: %s", why.c_str());
    *n = out.size();
    *runs = (RibbitRun *)std::malloc(std::max<size_t>(out.size(), 1) * sizeof(RibbitRun));
    if (!*runs) return fail(RIBBIT_E_NOMEM, "out of host memory");
    if (!out.empty()) std::memcpy(*runs, out.data(), out.size() * sizeof(RibbitRun));
    return RIBBIT_OK;
}

void ribbit_runs_free(RibbitRun *runs) { std::free(runs); }

}  // extern "C"
