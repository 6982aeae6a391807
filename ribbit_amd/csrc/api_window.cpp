// api_window.cpp -- the substitution and anchored stages (a6-a12); see api_internal.h for the map of the files behind include/ribbit_hip.h.
// There is no CPU fallback for any scan anywhere in this library.
#include "api_internal.h"

namespace rbapi {

// ---- window stages on the device ---------------------------------------------------------------------
// scan kernel -> pass-streak START / END events (left in their regions) -> pairing kernels -> one 16-byte record per
// streak, motif-major by start, in d_dense.  which: 1 window scan (1 mismatch), 2 fused anchored scan.
// filter (anchored scan only): groups of pass-streaks whose call cannot pass min_span leave no events (kernels.hip, "group
// filter"); their ends are left in h->d_dropmap for window_stage_device.
int scan_and_pair_streaks(RibbitHandle *h, int which, uint32_t *n_streaks, int (*filter_min_span)(int)) {
    int rc;
    if ((rc = bind_device(h))) return rc;
    if (h->copy_pending && (rc = perfect_wait(h))) return rc;      // d_dense / d_events are shared with the perfect stage
    if ((rc = h->d_counters.ensure(rb::EV_COUNTER_WORDS))) return rc;
    if ((rc = h->d_pair_status.ensure(rb::PAIR_STATUS_WORDS))) return rc;
    if (!h->h_pub.p) {
        if ((rc = h->h_pub.ensure(rb::EV_SHARDS + rb::PAIR_STATUS_WORDS))) return rc;
        HIP_TRY(hipHostGetDevicePointer((void **)&h->h_pub_dev, h->h_pub.p, 0));
    }
    rb::PairLaunch pr{};
    pr.m_lo = (uint32_t)h->params.min_motif;
    pr.nm = (uint32_t)(h->params.max_motif - h->params.min_motif + 1);
    // the anchored stage runs as two kernels: the planes (anchors + composition), then the window scan of the planes (kernels.hip).
    // (The fused form of rounds 1-2 -- 256 VGPRs, two waves per SIMD, 3.3 against 2.3 ms per 100 Mbp -- is gone from the product
    // since round 4; DESIGN.md 4 has its measurements.)
    pr.tile_bases = (uint32_t)rb::TILE_BASES;
    pr.ntile = (uint32_t)(h->length / pr.tile_bases + 1);
    pr.own_lo = 0; pr.own_hi = INT64_MAX; pr.pos_offset = 0;
    const size_t entries = (size_t)pr.nm * pr.ntile;
    if (entries > 0xfffffff0u) return fail(RIBBIT_E_ARG, "record too long for %u motif sizes", pr.nm);
    if ((rc = h->d_pair_table.ensure(entries))) return rc;
    if ((rc = h->d_run_base.ensure(entries))) return rc;
    if ((rc = h->d_pair_partial.ensure(entries / 1024 + 2))) return rc;
    if ((rc = h->d_halves.ensure(2 * (size_t)pr.nm))) return rc;
    const bool filter = which == 2 && filter_min_span != nullptr && std::getenv("RIBBIT_NO_GROUP_FILTER") == nullptr;
    // typical event densities on repeat-rich sequence: 0.25 per base (1-mismatch windows); anchored windows 3.7 at 99 motif
    // sizes without the group filter (growing with the motif sizes), and 1.3 (99 sizes) .. 1.6 (499) with it: the filter drops
    // nearly everything the large motifs add.  A too small first guess costs a second launch of the window scan; a too large
    // one costs memory -- until round 4 the filtered scan was sized like the unfiltered one, which at -M 500 meant 2 x 34 GB for
    // 3.3 GB of events, and on a box whose memory had just been used two seconds of the driver clearing it.
    // (1-mismatch windows at 499 motif sizes: 0.49 per base, and the profile of round 4 showed the scan launched twice)
    const size_t nm_all = (size_t)(h->params.max_motif - h->params.min_motif + 1);
    const size_t per_base_x4 = which == 1 ? std::max<size_t>(2, nm_all / 100) : filter ? 10 : std::max<size_t>(16, nm_all / 6);
    size_t cap = std::max<size_t>((size_t)1 << 20, (size_t)h->length * per_base_x4 / 4);
    cap = std::max(cap, h->d_events.cap);
    if (h->debug_first_cap) cap = h->debug_first_cap;
    const rb::DevicePlanes pl = h->planes();
    uint64_t produced = 0;
    bool first_attempt_fit = false;
    const size_t drop_words = (size_t)(h->length / 32 + 1) + 1024;
    h->dropmap_valid = false;
    if (filter) {
        // positions a group must span for its call to be able to pass: the call's length is the group's span + 7
        std::vector<int32_t> tj(pr.nm);
        for (uint32_t mi = 0; mi < pr.nm; ++mi) {
            const int t = std::min(filter_min_span((int)(pr.m_lo + mi)), rb::GROUP_FILTER_MAX + 7) - 7;
            tj[mi] = t > 1 ? t : 0;
        }
        if ((rc = h->d_tj.ensure(pr.nm)) || (rc = h->d_dropmap.ensure(drop_words))) return rc;
        HIP_TRY(hipMemcpyAsync(h->d_tj.p, tj.data(), pr.nm * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));       // tj is a local
    }
    for (int attempt = 0;; ++attempt) {
        cap = std::min<size_t>((cap + rb::EV_SHARDS - 1) / rb::EV_SHARDS * rb::EV_SHARDS, 0xffffff00u);
        if ((rc = h->d_events.ensure(cap))) return rc;
        if ((rc = h->d_dense.ensure(cap))) return rc;          // cap / 2 streak records of 16 bytes
        if (filter) HIP_TRY(hipMemsetAsync(h->d_dropmap.p, 0, drop_words * sizeof(uint32_t), h->stream));
        HIP_TRY(hipEventRecord(h->ev[4], h->stream));
        if (!h->counters_clean) HIP_TRY(hipMemsetAsync(h->d_counters.p, 0, rb::EV_COUNTER_WORDS * sizeof(uint32_t), h->stream));
        h->counters_clean = false;
        rb::PerfectLaunch pp;
        pp.m_lo = h->params.min_motif;
        pp.m_hi = h->params.max_motif;
        pp.ev_cap = (uint32_t)cap;
        pr.region_cap = pp.ev_cap / (uint32_t)rb::EV_SHARDS;
        HIP_TRY(hipEventRecord(h->ev[2], h->stream));
        // (two-kernel anchored stage: the planes kernel runs on the first attempt only, so the stage's start mark stays where it
        // was put then -- timer 7 is "both kernels", also when the window scan had to run again with more room)
        if (attempt == 0 || which == 1) HIP_TRY(hipEventRecord(h->ev_stage[which - 1][0], h->stream));
        if (which == 1) rb::launch_scan_window(pl, pp, 1, h->d_events.p, h->d_counters.p, h->stream);
        else {
            if (attempt == 0) {          // the planes do not depend on the event capacity: once
                rb::launch_scan_anchored(pl, pp, h->d_xa.p, h->xa_stride, h->stream);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(h->ev_planes, h->stream));
            }
            rb::launch_scan_xa_window(pl, pp, h->d_xa.p, h->xa_stride, h->d_events.p, h->d_counters.p, filter ? h->d_tj.p : nullptr,
                                      filter ? h->d_dropmap.p : nullptr, h->stream);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(h->ev[3], h->stream));
        HIP_TRY(hipEventRecord(h->ev_stage[which - 1][1], h->stream));
        h->have_stage_timing[which - 1] = true;
        HIP_TRY(rb::launch_pair_runs(h->d_events.p, h->d_counters.p, pr, h->d_pair_table.p, h->d_run_base.p, h->d_pair_partial.p,
                                     h->d_dense.p, (uint32_t)(cap / 2), h->d_halves.p, (uint32_t)(2 * (size_t)pr.nm), h->d_pair_status.p, h->stream));
        rb::launch_pair_publish(h->d_counters.p, h->d_pair_status.p, h->h_pub_dev, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
        uint32_t worst = 0;
        produced = 0;
        for (int t = 0; t < rb::EV_SHARDS; ++t) { worst = std::max(worst, h->h_pub.p[t]); produced += h->h_pub.p[t]; }
        if (worst <= pr.region_cap) { first_attempt_fit = attempt == 0; break; }
        if (attempt == 2 || (size_t)worst * rb::EV_SHARDS > 0xffffff00u)
            return fail(RIBBIT_E_OVERFLOW, "event buffer overflow: fullest region needs %u events", worst);
        cap = ((size_t)worst + 1024) * rb::EV_SHARDS;
    }
    h->have_timing[1] = true;
    h->last_event_count = (int64_t)produced;
    const uint32_t flags = h->h_pub.p[rb::EV_SHARDS + rb::PAIR_FLAGS];
    if (flags)
        return fail(RIBBIT_E_INTERNAL, "streak pairing failed (flags 0x%x):%s%s%s%s%s", flags,
                    flags & rb::PAIR_BAD_EVENT ? " malformed event;" : "", flags & rb::PAIR_DUP_CHUNK ? " duplicate event chunk;" : "",
                    flags & rb::PAIR_NOT_ALTERNATING ? " streak starts and ends do not alternate;" : "",
                    flags & rb::PAIR_UNTERMINATED ? " unterminated streak;" : "", flags & rb::PAIR_NO_ROOM ? " streak buffer too small;" : "");
    const uint32_t n = h->h_pub.p[rb::EV_SHARDS + rb::PAIR_TOTAL];
    if ((uint64_t)n * 2 != produced) return fail(RIBBIT_E_INTERNAL, "%llu events but %u streaks", (unsigned long long)produced, n);
    if (h->h_pub.p[rb::EV_SHARDS + rb::PAIR_HALVES]) return fail(RIBBIT_E_INTERNAL, "streak cut by the own range of a whole record");
    *n_streaks = n;
    h->last_streaks = n;
    h->dropmap_valid = filter;
    if (which == 2) h->planes_timing_valid = first_attempt_fit;
    return RIBBIT_OK;
}

// The whole window stage on the device (window_stage.hip).  full: every call, unfiltered (the call-list entry points
// and the parity tests); otherwise only the calls that pass min_span, with their cursor bounds.  cw: chunk mode.
int window_stage_device(RibbitHandle *h, int which, bool full, int (*min_span)(int), DeviceCalls *out, ChunkWindow *cw) {
    int rc;
    uint32_t n = 0;
    PinnedBuf<RibbitCall> &h_calls = h->h_calls_[which - 1], &h_flush = h->h_flush_[which - 1];
    PinnedBuf<int32_t> &h_pend = h->h_pend_[which - 1];
    PinnedBuf<uint32_t> &h_ws = h->h_ws_[which - 1];
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    const double t_scan = now_ms();
    if ((rc = scan_and_pair_streaks(h, which, &n, full ? nullptr : min_span))) return rc;
    const double t0 = now_ms();
    const uint32_t nm = (uint32_t)(h->params.max_motif - h->params.min_motif + 1);
    const uint32_t n_words = (uint32_t)(h->length / 32 + 1);
    int key_bits = 10;
    while (((int64_t)1 << (key_bits - 10)) <= h->length + 8) ++key_bits;
    size_t scratch = rb::window_stage_scratch_bytes(n, n_words, n, std::max<size_t>(n / 16, (size_t)1 << 16), key_bits);
    if ((rc = h->d_scratch.ensure(scratch))) return rc;
    if ((rc = h->d_word_tmp.ensure(n_words + 1))) return rc;
    if (!h->eval_valid) {
        if ((rc = h->d_eval.ensure(n_words + 1))) return rc;
        if ((rc = h->d_first_rev.ensure(n_words + 1))) return rc;
        HIP_TRY(rb::launch_eval_planes(h->d_brk.p + rb::LEAD_WORDS, n_words, h->d_eval.p, h->d_first_rev.p, h->d_word_tmp.p, h->d_scratch.p, h->d_scratch.cap, h->stream));
        HIP_TRY(hipGetLastError());
        h->eval_valid = true;
    }
    if ((rc = h->d_group.ensure(std::max<size_t>(n, 1)))) return rc;
    const RibbitRun *runs = reinterpret_cast<const RibbitRun *>(h->d_dense.p);
    HIP_TRY(rb::launch_group_starts(runs, n, (uint32_t)h->params.min_motif, h->d_group.p, h->d_scratch.p, h->d_scratch.cap, h->stream));
    HIP_TRY(hipGetLastError());
    // length filter of the stage (seedlen_cutoffs), per motif
    std::vector<int32_t> spans(nm, 0);
    if (!full) for (uint32_t mi = 0; mi < nm; ++mi) spans[mi] = min_span(h->params.min_motif + (int)mi);
    if ((rc = h->d_min_span.ensure(nm))) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_min_span.p, spans.data(), nm * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    if ((rc = h->d_flush.ensure(nm))) return rc;
    if ((rc = h->d_ws_counters.ensure(rb::WS_WORDS))) return rc;
    if ((rc = h->d_bitmap.ensure(n_words + 1))) return rc;
    if ((rc = h_ws.ensure(rb::WS_WORDS))) return rc;
    if ((rc = h_flush.ensure(nm))) return rc;
    size_t edge_cap = std::max<size_t>(h->d_edge_keys.cap, std::max<size_t>((size_t)1 << 16, n / 16));
    // the events are spent: their buffer (2 x 8 bytes per streak at least) receives the unsorted calls
    uint64_t *keys = h->d_events.p, *vals = h->d_events.p + h->d_events.cap / 2;
    uint32_t n_main = 0, n_edge = 0;
    for (int attempt = 0;; ++attempt) {
        if (!full) {
            if ((rc = h->d_edge_keys.ensure(edge_cap)) || (rc = h->d_edge_vals.ensure(edge_cap))) return rc;
            HIP_TRY(hipMemsetAsync(h->d_bitmap.p, 0, ((size_t)n_words + 1) * sizeof(uint32_t), h->stream));
        }
        HIP_TRY(hipMemsetAsync(h->d_flush.p, 0, nm * sizeof(RibbitCall), h->stream));
        HIP_TRY(hipMemsetAsync(h->d_ws_counters.p, 0, rb::WS_WORDS * sizeof(uint32_t), h->stream));
        rb::WindowCallsLaunch w{};
        w.runs = runs; w.n_streaks = n; w.group = h->d_group.p;
        w.eval = h->d_eval.p; w.first_rev = h->d_first_rev.p; w.brk = h->d_brk.p + rb::LEAD_WORDS; w.n_words = n_words;
        w.length = h->length; w.m_lo = (uint32_t)h->params.min_motif; w.nm = nm; w.min_span = h->d_min_span.p; w.full = full ? 1 : 0;
        w.keys = keys; w.vals = vals; w.cap = (uint32_t)(h->d_events.cap / 2);
        w.edge_keys = h->d_edge_keys.p; w.edge_vals = h->d_edge_vals.p; w.edge_cap = full ? 0u : (uint32_t)edge_cap;
        w.flush = h->d_flush.p; w.bitmap = h->d_bitmap.p; w.counters = h->d_ws_counters.p;
        if (cw) { w.own_lo = cw->own_lo; w.own_hi = cw->own_hi; w.z_lo = cw->z_lo; w.keep_flush = cw->keep_flush ? 1 : 0; }
        rb::launch_window_calls(w, h->stream);
        HIP_TRY(hipGetLastError());
        if (h->dropmap_valid && !full) {
            rb::launch_merge_dropmap(h->d_dropmap.p, (uint32_t)((size_t)(h->length / 32 + 1) + 1024), n_words, w.own_lo, w.own_hi, h->d_bitmap.p,
                                     h->d_ws_counters.p, h->stream);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(h_ws.p, h->d_ws_counters.p, rb::WS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(h_flush.p, h->d_flush.p, nm * sizeof(RibbitCall), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        n_main = h_ws.p[rb::WS_N_MAIN];
        n_edge = h_ws.p[rb::WS_N_EDGE];
        if (n_edge <= edge_cap || full) break;
        if (attempt == 1) return fail(RIBBIT_E_INTERNAL, "edge-call list overflow");
        edge_cap = (size_t)n_edge + 1024;
    }
    const double t_calls = now_ms();
    uint32_t wflags = h_ws.p[rb::WS_FLAGS];
    if (wflags) return fail(RIBBIT_E_INTERNAL, "window state machine on the device failed (flags 0x%x)", wflags);
    if (cw) cw->inexact = h_ws.p[rb::WS_INEXACT] != 0;
    const int32_t pos_offset = cw ? cw->pos_offset : 0;
    if (n_main > h->d_events.cap / 2) return fail(RIBBIT_E_INTERNAL, "more calls than streaks");
    // call order: scan position major, motif minor
    scratch = rb::window_stage_scratch_bytes(0, n_words, n_main, n_edge, key_bits);
    if ((rc = h->d_scratch.ensure(scratch))) return rc;
    if ((rc = h->d_sort_keys.ensure(std::max<size_t>(n_main, 1))) || (rc = h->d_sort_vals.ensure(std::max<size_t>(n_main, 1)))) return rc;
    HIP_TRY(rb::launch_sort_calls(keys, vals, h->d_sort_keys.p, h->d_sort_vals.p, n_main, key_bits, h->d_scratch.p, h->d_scratch.cap, h->stream));
    HIP_TRY(hipGetLastError());
    const bool bounds = !full && n_edge > 0 && n_main > 0;
    if (bounds) {
        if ((rc = h->d_edge_keys2.ensure(n_edge)) || (rc = h->d_edge_vals2.ensure(n_edge))) return rc;
        if ((rc = h->d_edge_tmp.ensure(n_edge)) || (rc = h->d_edge_end1.ensure(n_edge))) return rc;
        if ((rc = h->d_last_word.ensure(n_words + 1))) return rc;
        if ((rc = h->d_pend.ensure(n_main))) return rc;
        HIP_TRY(rb::launch_sort_calls(h->d_edge_keys.p, h->d_edge_vals.p, h->d_edge_keys2.p, h->d_edge_vals2.p, n_edge, key_bits, h->d_scratch.p, h->d_scratch.cap, h->stream));
        HIP_TRY(hipMemsetAsync(h->d_pend.p, 0xff, (size_t)n_main * sizeof(int32_t), h->stream));
        HIP_TRY(rb::launch_edge_bounds(h->d_edge_keys2.p, h->d_edge_vals2.p, n_edge, h->d_edge_tmp.p, h->d_edge_end1.p, h->d_bitmap.p, h->d_word_tmp.p,
                                       h->d_last_word.p, n_words, h->d_sort_keys.p, n_main, h->d_pend.p, h->d_ws_counters.p, pos_offset, h->d_scratch.p, h->d_scratch.cap, h->stream));
        HIP_TRY(hipGetLastError());
    }
    // 16-byte call records for the host; the streak records are spent, their buffer takes them
    RibbitCall *d_calls = reinterpret_cast<RibbitCall *>(h->d_dense.p);
    rb::launch_assemble_calls(h->d_sort_keys.p, h->d_sort_vals.p, n_main, d_calls, pos_offset, h->stream);
    HIP_TRY(hipGetLastError());
    if ((rc = h_calls.ensure(std::max<size_t>(n_main, 1)))) return rc;
    if (n_main) HIP_TRY(hipMemcpyAsync(h_calls.p, d_calls, (size_t)n_main * sizeof(RibbitCall), hipMemcpyDeviceToHost, h->stream));
    if (bounds) {
        if ((rc = h_pend.ensure(n_main))) return rc;
        HIP_TRY(hipMemcpyAsync(h_pend.p, h->d_pend.p, (size_t)n_main * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(h_ws.p, h->d_ws_counters.p, rb::WS_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipEventRecord(h->ev[5], h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_timing[2] = true;
    wflags = h_ws.p[rb::WS_FLAGS];
    if (wflags) return fail(RIBBIT_E_INTERNAL, "cursor bounds of the edge calls failed (flags 0x%x)", wflags);
    // end-of-sequence calls: at most one per motif, already in motif order; close the gaps
    size_t nf = 0;
    for (uint32_t mi = 0; mi < nm; ++mi)
        if (h_flush.p[mi].mlen != 0) {
            RibbitCall c = h_flush.p[mi];
            c.pos += pos_offset; c.start += pos_offset; c.end += pos_offset;
            h_flush.p[nf++] = c;
        }
    out->calls = h_calls.p;
    out->n = n_main;
    out->pend = bounds ? h_pend.p : nullptr;
    out->tail_pend = h_ws.p[rb::WS_MAX_END] ? (int32_t)h_ws.p[rb::WS_MAX_END] - 1 + pos_offset : -1;
    out->flush = h_flush.p;
    out->n_flush = nf;
    h->last_calls = n_main;
    h->last_edge_calls = n_edge;
    h->host_ms = now_ms() - t0;
    if (profile)
        std::fprintf(stderr, "[window stage %d%s] scan + pairing %.1f ms (%u streaks), group scan + calls kernel %.1f ms, sort + bounds + read-back %.1f ms: "
                     "%u calls, %u edge calls, %zu flush calls\n", which, full ? " full" : "", t0 - t_scan, n, t_calls - t0, now_ms() - t_calls, n_main, n_edge, nf);
    return RIBBIT_OK;
}

void full_calls_from_device(const DeviceCalls &dc, rb::CallVec &calls) {
    calls.resize(dc.n + dc.n_flush);
    if (dc.n) std::memcpy(calls.data(), dc.calls, dc.n * sizeof(RibbitCall));
    if (dc.n_flush) std::memcpy(calls.data() + dc.n, dc.flush, dc.n_flush * sizeof(RibbitCall));
}

// window scan (1 mismatch) + per-motif state machine, both on the device -> the addSeed call list of
// processShiftXORswithSubstitutions (parse_substitute_shiftxor.cpp:430-574)
int build_subst_calls(RibbitHandle *h) {
    if (h->subst_calls_valid) return RIBBIT_OK;
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    DeviceCalls dc;
    int rc = window_stage_device(h, 1, true, nullptr, &dc);
    if (rc) return rc;
    full_calls_from_device(dc, h->subst_calls);
    h->subst_calls_valid = true;
    return RIBBIT_OK;
}

// host half of the substitution stage: the merges of parse_substitute_shiftxor.cpp:18-388 over the stage's calls
void subst_merge(RibbitHandle *h, const DeviceCalls *dc) {
    const rb::HostPlanes *hp = &h->host;
    h->lists.range_count = [hp](int shift, int start, int end) { return hp->range_count(shift, start, end); };
    h->lists.subst.clear();
    const double t0 = now_ms();
    const unsigned threads = rb::merge_threads(h->host_threads);
    rb::MergeStats st;
    if (dc) rb::merge_subst_stage(h->lists, *dc, threads, &st);
    else rb::merge_subst_stage_full(h->lists, h->subst_calls.data(), h->subst_calls.size(), threads, &st);
    h->merge_ms = now_ms() - t0;
    h->stage_done = STAGE_SUBST;
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    if (profile)
        std::fprintf(stderr, "[subst merge] %zu seeds: %u ranges on %u threads%s, preparation %.1f ms, merges %.1f ms\n", h->lists.subst.size(), st.ranges,
                     st.threads, st.redone_in_order ? " (REDONE IN ORDER)" : "", st.prepare_ms, st.merge_ms);
}

int advance_to_subst(RibbitHandle *h) {
    if (h->stage_done >= STAGE_SUBST) return RIBBIT_OK;
    int rc = advance_to_perfect(h);
    if (rc) return rc;
    if ((rc = ensure_host_planes(h))) return rc;
    DeviceCalls dc;
    const bool full = h->subst_calls_valid;      // the full call list has been asked for (ribbit_hip_subst_calls): replay that
    if (!full && (rc = window_stage_device(h, 1, false, rb::subst_seedlen_cutoff, &dc))) return rc;
    subst_merge(h, full ? nullptr : &dc);
    return RIBBIT_OK;
}

// The anchored stage's kernel writes the composed planes XA_m (fasta_utils.cpp:143-161) to HBM: the device-side
// refinement scans read them there, and the host merges' range reads (retainNestedSeedAnchored,
// parse_anchored_shiftxor.cpp:59-84: ~200 K per Mbp, each steering the next decision) read a host copy.  Recomputing
// the slice of a query from the packed planes instead (HostPlanes::xa_slice, what the host-only entry points do when
// they are not given the planes) costs ~0.6 us per query -- 1.4 s per 20 Mbp against 5 ms for the copy, which
// moreover runs behind the substitution stage's merge.
int prepare_anchored(RibbitHandle *h) {
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (h->params.max_motif > rb::ANCHORED_MAX_MOTIF)
        return fail(RIBBIT_E_ARG, "the anchored stage of this build supports max_motif <= %d (got %d)", rb::ANCHORED_MAX_MOTIF, h->params.max_motif);
    const size_t nm = (size_t)(h->params.max_motif - h->params.min_motif + 1);
    // whole tiles of the window kernel that reads the planes back (scan_xa_window_kernel), plus its two words of look-ahead
    h->xa_stride = (h->length / 32 + 1 + rb::TILE_WORDS - 1) / rb::TILE_WORDS * rb::TILE_WORDS + 16;
    return h->d_xa.ensure(nm * (size_t)h->xa_stride);
}

// enqueue the copy of the composed planes on the handle's copy stream (behind everything enqueued on the compute
// stream so far); xa_wait_host() makes them readable
int xa_copy_begin(RibbitHandle *h) {
    int rc;
    const size_t nm = (size_t)(h->params.max_motif - h->params.min_motif + 1);
    if ((rc = h->h_xa.ensure(nm * (size_t)h->xa_stride))) return rc;      // page-locked: the copy runs at link speed
    HIP_TRY(hipEventRecord(h->ev_xa, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->ev_xa, 0));
    HIP_TRY(hipMemcpyAsync(h->h_xa.p, h->d_xa.p, nm * (size_t)h->xa_stride * sizeof(uint32_t), hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(hipEventRecord(h->ev_xa, h->copy_stream));
    h->xa_copy_pending = true;
    return RIBBIT_OK;
}

int xa_wait_host(RibbitHandle *h) {
    if (h->xa_copy_pending) {
        HIP_TRY(hipEventSynchronize(h->ev_xa));
        h->xa_copy_pending = false;
    }
    h->host.xa.clear();
    h->host.xa_view = h->h_xa.p;
    h->host.xa_stride = h->xa_stride;
    h->host.xa_m_lo = h->params.min_motif;
    h->host.xa_m_hi = h->params.max_motif;
    return RIBBIT_OK;
}

// fused anchored kernel (anchor planes + composition + 6-of-8 window scan) + state machine, on the device ->
// the addSeed call list of processShiftXORsAnchored (parse_anchored_shiftxor.cpp:580-723)
int build_anchored_calls(RibbitHandle *h) {
    if (h->anchored_calls_valid) return RIBBIT_OK;
    int rc = prepare_anchored(h);
    if (rc) return rc;
    DeviceCalls dc;
    if ((rc = window_stage_device(h, 2, true, nullptr, &dc))) return rc;
    full_calls_from_device(dc, h->anchored_calls);
    h->xa_on_device = true;
    h->anchored_calls_valid = true;
    return RIBBIT_OK;
}

// processShiftXORswithSubstitutions + processShiftXORsAnchored.  All GPU work of both stages is enqueued before
// either host merge starts, so the copies (kept calls, composed planes) travel while the host merges.
// RIBBIT_PROFILE line of the anchored stage's merge (GPU path and host replay alike)
void print_anchored_merge_profile(size_t seeds, const rb::MergeStats &st, double dispatch_ms, unsigned dispatch_ranges) {
    std::fprintf(stderr, "[anchored merge] %zu seeds: %u ranges on %u threads%s, %u passes, %lld changing head writes, %u ranges done again, preparation %.1f ms (cuts %.1f, cursors %.1f, type snapshots %.1f), merges %.1f ms "
                 "(parallel passes %.1f ms over %u range runs [device pass %.1f ms for %u ranges, the host threads' share meanwhile: %u ranges in %.1f ms; results into the ranges' states %.1f ms; then %u ranges the device left]: the ranges' own times sum to %.1f ms = %.1f ms per thread, longest range %.1f ms; in-order walk %.1f ms; joining the ranges' lists %.1f ms; before the first pass %.1f ms; end-of-sequence calls %.1f ms), dispatch order %.1f ms in %u ranges\n",
                 seeds, st.ranges, st.threads, st.redone_in_order ? " (REDONE IN ORDER)" : (st.head_writes ? " (list-head writes: ranges done again, see passes)" : ""), st.passes, st.head_writes,
                 st.ranges_redone, st.prepare_ms, st.prep_parts[0], st.prep_parts[1] - st.prep_parts[0], st.prep_parts[2] - st.prep_parts[1], st.merge_ms, st.pass_ms, st.ranges_run, st.device_ms, st.device_ranges, st.device_host_share, st.device_meanwhile_ms, st.device_apply_ms, st.device_bailed, st.range_ms_sum, st.range_ms_sum / std::max(1u, st.threads), st.range_ms_max, st.walk_ms, st.concat_ms, st.before_passes_ms, st.flush_ms, dispatch_ms, dispatch_ranges);
}

int advance_to_anchored(RibbitHandle *h) {
    if (h->stage_done >= STAGE_ANCHORED) return RIBBIT_OK;
    int rc = advance_to_perfect(h);
    if (rc) return rc;
    if ((rc = ensure_host_planes(h))) return rc;
    DeviceCalls dcs, dca;
    const bool subst_todo = h->stage_done < STAGE_SUBST;
    const bool subst_full = h->subst_calls_valid;
    // the full call lists only when they have been asked for (ribbit_hip_*_calls); otherwise the compact form:
    // nine anchored calls in ten fail the length filter and never leave the device
    if (subst_todo && !subst_full && (rc = window_stage_device(h, 1, false, rb::subst_seedlen_cutoff, &dcs))) return rc;
    const bool full = h->anchored_calls_valid;
    if (!full) {
        if ((rc = prepare_anchored(h))) return rc;
        if ((rc = window_stage_device(h, 2, false, rb::anchored_seedlen_cutoff, &dca))) return rc;
        h->xa_on_device = true;
    }
    if ((rc = bind_device(h))) return rc;
    const double tx0 = now_ms();
    if ((rc = xa_copy_begin(h))) return rc;
    const double tx1 = now_ms();
    if (subst_todo) subst_merge(h, subst_full ? nullptr : &dcs);
    const double merge_s = h->merge_ms;
    const double tx2 = now_ms();
    if ((rc = xa_wait_host(h))) return rc;
    static const bool profile_xa = std::getenv("RIBBIT_PROFILE") != nullptr;
    if (profile_xa)
        std::fprintf(stderr, "[composed planes] %.2f GB to the host for the merges' range reads: page-locked room and enqueue %.1f ms, waited %.1f ms for the copy after the substitution merge\n",
                     (double)(h->params.max_motif - h->params.min_motif + 1) * (double)h->xa_stride * 4e-9, tx1 - tx0, now_ms() - tx2);
    // from here on "plane m" means the composed plane XA_m (fasta_utils.cpp:159)
    const rb::HostPlanes *hp = &h->host;
    h->lists.range_count = [hp](int shift, int start, int end) {
        return hp->has_xa(shift) ? hp->range_count_xa(shift, start, end) : hp->range_count(shift, start, end);
    };
    if (hp->xa_stored()) { h->lists.plane_words = hp->xa_words(); h->lists.plane_stride = hp->xa_stride; h->lists.plane_lo = hp->xa_m_lo; h->lists.plane_hi = hp->xa_m_hi; }
    // (lists.anchored is not cleared here: every path of the stage sets it, and the join reuses what it holds, parallel_merge.cpp)
    const double t0 = now_ms();
    const unsigned threads = rb::merge_threads(h->host_threads);
    rb::MergeStats st;
    if (full) rb::merge_anchored_stage_full(h->lists, h->anchored_calls.data(), h->anchored_calls.size(), threads, &st);
    else {
        // the first parallel pass as device work for stages of a million calls and more (api_merge.cpp): the kept calls are still
        // where window_stage_device put them (d_dense as 16-byte call records, d_pend)
        const rb::AnchoredDevicePass on_device = anchored_device_pass(h, reinterpret_cast<RibbitCall *>(h->d_dense.p), dca.pend ? h->d_pend.p : nullptr);
        rb::merge_anchored_stage(h->lists, dca, threads, &st, &on_device);
    }
    const double t1 = now_ms();
    const unsigned dispatch_ranges = rb::dispatch_order_ranges(h->lists, st.cut_pos, threads, h->dispatch);
    h->merge_ms = now_ms() - t0;
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    if (profile) print_anchored_merge_profile(h->lists.anchored.size(), st, now_ms() - t1, dispatch_ranges);
    h->subst_merge_ms = subst_todo ? merge_s : 0.0;
    h->stage_done = STAGE_ANCHORED;
    return RIBBIT_OK;
}

}  // namespace rbapi

extern "C" {

int ribbit_hip_subst_calls(RibbitHandle *h, const RibbitCall **out, size_t *n) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    int rc = build_subst_calls(h);
    if (rc) return rc;
    *out = h->subst_calls.data();
    *n = h->subst_calls.size();
    return RIBBIT_OK;
}

int ribbit_hip_seeds_substitutions(RibbitHandle *h, const RibbitSeed **perfect, size_t *n_perfect,
                                   const RibbitSeed **subst, size_t *n_subst) {
    if (!h || !perfect || !n_perfect || !subst || !n_subst) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (h->stage_done > STAGE_SUBST) return fail(RIBBIT_E_STATE, "a later stage already re-typed the lists; reload the record");
    int rc = advance_to_subst(h);
    if (rc) return rc;
    *perfect = h->lists.perfect.data();
    *n_perfect = h->lists.perfect.size();
    *subst = h->lists.subst.data();
    *n_subst = h->lists.subst.size();
    return RIBBIT_OK;
}

int ribbit_hip_anchored_calls(RibbitHandle *h, const RibbitCall **out, size_t *n) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    int rc = build_anchored_calls(h);
    if (rc) return rc;
    *out = h->anchored_calls.data();
    *n = h->anchored_calls.size();
    return RIBBIT_OK;
}

int ribbit_hip_seeds_anchored(RibbitHandle *h, const RibbitSeed **perfect, size_t *n_perfect,
                              const RibbitSeed **subst, size_t *n_subst,
                              const RibbitSeed **anchored, size_t *n_anchored) {
    if (!h || !perfect || !n_perfect || !subst || !n_subst || !anchored || !n_anchored) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    int rc = advance_to_anchored(h);
    if (rc) return rc;
    *perfect = h->lists.perfect.data();   *n_perfect = h->lists.perfect.size();
    *subst = h->lists.subst.data();       *n_subst = h->lists.subst.size();
    *anchored = h->lists.anchored.data(); *n_anchored = h->lists.anchored.size();
    return RIBBIT_OK;
}

int ribbit_hip_dispatch_seeds(RibbitHandle *h, const RibbitSeed **out, size_t *n) {
    if (!h || !out || !n) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    int rc = advance_to_anchored(h);
    if (rc) return rc;
    *out = h->dispatch.data();
    *n = h->dispatch.size();
    return RIBBIT_OK;
}

void ribbit_debug_set_merge_min_range(size_t calls) { rb::set_merge_min_range(calls); }

int32_t ribbit_debug_last_dispatch_ranges(void) { return (int32_t)rb::last_dispatch_ranges(); }

void ribbit_debug_last_merge(int stage, int32_t out[5]) {
    const rb::MergeStats st = rb::last_merge_stats(stage);
    out[0] = (int32_t)st.ranges; out[1] = (int32_t)std::min(st.ranges_redone, 0xffffu) | (int32_t)(std::min(st.stale_by_sight, 0x7fffu) << 16); out[2] = (st.redone_in_order ? 1 : 0) | (int32_t)(st.ranges_run << 1);
    out[3] = (int32_t)std::min<long long>(st.head_writes, INT32_MAX); out[4] = (st.first_range_empty ? 1 : 0) | (int32_t)(st.passes << 8);
}

void ribbit_debug_last_device_merge(int32_t out[5]) {
    const rb::MergeStats st = rb::last_merge_stats(1);
    out[0] = (int32_t)st.device_ranges; out[1] = (int32_t)st.device_bailed; out[2] = (int32_t)st.device_host_share; out[3] = (int32_t)st.ranges; out[4] = (int32_t)st.ranges_redone;
}

int ribbit_host_replay_calls(const RibbitScanParams *params, int64_t length,
                             const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                             const uint32_t *xa, size_t xa_stride,
                             const RibbitCall *perfect_calls, size_t n_perfect_calls,
                             const RibbitCall *subst_calls, size_t n_subst_calls,
                             const RibbitCall *anchored_calls, size_t n_anchored_calls,
                             RibbitSeedLists *out) {
    if (!params || !out || (length > 0 && (!hi || !lo || !brk))) return fail(RIBBIT_E_ARG, "null argument");
    if ((n_perfect_calls && !perfect_calls) || (n_subst_calls && !subst_calls) || (n_anchored_calls && !anchored_calls))
        return fail(RIBBIT_E_ARG, "null call list");
    const size_t need = (size_t)(length / 32 + 1) + (size_t)(params->max_motif + 2) / 32 + 2;
    if (nwords < need) return fail(RIBBIT_E_ARG, "planes too short: %zu words, need %zu (zero padding past the record)", nwords, need);
    if (xa && xa_stride < (size_t)(length / 32 + 1)) return fail(RIBBIT_E_ARG, "composed planes (xa) too short");
    std::memset(out, 0, sizeof *out);
    rb::HostPlanes hp;
    hp.resize(length, nwords);
    std::memcpy(hp.hi.data(), hi, nwords * sizeof(uint32_t));
    std::memcpy(hp.lo.data(), lo, nwords * sizeof(uint32_t));
    std::memcpy(hp.brk.data(), brk, nwords * sizeof(uint32_t));
    rb::SeedLists sl;
    sl.length = length;
    sl.min_motif = params->min_motif;
    sl.max_motif = params->max_motif;
    sl.min_shift = (params->min_motif > 2) ? params->min_motif - 2 : 1;
    sl.range_count = [&hp](int shift, int start, int end) { return hp.range_count(shift, start, end); };
    for (size_t i = 0; i < n_perfect_calls; ++i) rb::perfect_add(sl, perfect_calls[i].start, perfect_calls[i].end, perfect_calls[i].mlen);
    rb::merge_subst_stage_full(sl, subst_calls, n_subst_calls, rb::merge_threads(0));
    // the anchored stage runs when there are anchored calls or composed planes are given; anchored_calls non-null with
    // n == 0 also asks for it (a record whose anchored scan made no call still gets its dispatch list)
    const bool anchored_stage = n_anchored_calls || xa || anchored_calls;
    if (anchored_stage) {
        const size_t nm = (size_t)(params->max_motif - params->min_motif + 1);
        if (xa) hp.xa.assign(xa, xa + nm * xa_stride);        // else: recomputed slice by slice from the packed planes
        hp.xa_stride = xa ? (int64_t)xa_stride : 0;
        hp.xa_m_lo = params->min_motif;
        hp.xa_m_hi = params->max_motif;
        sl.range_count = [&hp](int shift, int start, int end) {
            return hp.has_xa(shift) ? hp.range_count_xa(shift, start, end) : hp.range_count(shift, start, end);
        };
        if (hp.xa_stored()) { sl.plane_words = hp.xa_words(); sl.plane_stride = hp.xa_stride; sl.plane_lo = hp.xa_m_lo; sl.plane_hi = hp.xa_m_hi; }
    }
    rb::SeedVec dispatch;
    if (anchored_stage) {
        rb::MergeStats st;
        // test hook RIBBIT_MERGE_DEVICE_RANGES=<calls per range>: the stage cut as for the GPU's pass of the anchored merge (hundreds of
        // thousands of ranges, prepared on the host threads) with a device that cannot run -- the host threads then merge those ranges
        rb::AnchoredDevicePass no_device;
        const char *fine = std::getenv("RIBBIT_MERGE_DEVICE_RANGES");
        if (fine) {
            no_device.min_calls = 0; no_device.calls_per_range = (size_t)std::max(1, std::atoi(fine));
            // (the "device" takes the lighter half of the ranges and merges none of them)
            no_device.run = [](const rb::SeedLists &, const rb::KeptCalls &, const std::vector<size_t> &, const std::vector<int> &, const std::vector<rb::Cursor2> &, const std::vector<uint32_t> &order, size_t,
                               const std::function<void(uint32_t *, const uint32_t *)> &meanwhile, std::vector<rb::AnchoredDevicePass::RangeResult> &, std::vector<rb::AnchoredDevicePass::LogEntry> &,
                               std::vector<rb::AnchoredDevicePass::LogEntry> &, std::vector<rb::AnchoredDevicePass::HeadEntry> &) {
                uint32_t from_back = (uint32_t)order.size(), from_front = (uint32_t)order.size() / 2;
                meanwhile(&from_back, &from_front);
                return false;
            };
        }
        rb::merge_anchored_stage_full(sl, anchored_calls, n_anchored_calls, rb::merge_threads(0), &st, fine ? &no_device : nullptr);
        const double td = now_ms();
        const unsigned dispatch_ranges = rb::dispatch_order_ranges(sl, st.cut_pos, rb::merge_threads(0), dispatch);
        if (std::getenv("RIBBIT_PROFILE")) print_anchored_merge_profile(sl.anchored.size(), st, now_ms() - td, dispatch_ranges);
    }
    auto give = [](const rb::SeedVec &v, RibbitSeed **p, size_t *n) {
        *n = v.size();
        *p = (RibbitSeed *)std::malloc(std::max<size_t>(v.size(), 1) * sizeof(RibbitSeed));
        if (*p && !v.empty()) std::memcpy(*p, v.data(), v.size() * sizeof(RibbitSeed));
        return *p != nullptr;
    };
    if (!give(sl.perfect, &out->perfect, &out->n_perfect) || !give(sl.subst, &out->subst, &out->n_subst) ||
        !give(sl.anchored, &out->anchored, &out->n_anchored) || !give(dispatch, &out->dispatch, &out->n_dispatch)) {
        ribbit_seed_lists_free(out);
        return fail(RIBBIT_E_NOMEM, "out of host memory");
    }
    out->guard_hits = sl.guard_hits;
    return RIBBIT_OK;
}

void ribbit_seed_lists_free(RibbitSeedLists *lists) {
    if (!lists) return;
    std::free(lists->perfect); std::free(lists->subst); std::free(lists->anchored); std::free(lists->dispatch);
    std::memset(lists, 0, sizeof *lists);
}

}  // extern "C"
