// api_chunks.cpp -- one chunk of a longer record, and the merging rank; see api_internal.h for the map of the files behind include/ribbit_hip.h.
// There is no CPU fallback for any scan anywhere in this library.
#include "api_internal.h"

extern "C" {

int ribbit_hip_xa_words(RibbitHandle *h, int64_t word_lo, int64_t word_hi, uint32_t *out) {
    if (!h || !out) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded || !h->d_xa.p || h->xa_stride == 0) return fail(RIBBIT_E_STATE, "the anchored kernel has not run on this record");
    if (word_lo < 0 || word_hi < word_lo || word_hi > h->xa_stride) return fail(RIBBIT_E_ARG, "word range outside the planes");
    int rc;
    if ((rc = bind_device(h))) return rc;
    const size_t nm = (size_t)(h->params.max_motif - h->params.min_motif + 1);
    const size_t w = (size_t)(word_hi - word_lo);
    if (w == 0) return RIBBIT_OK;
    HIP_TRY(hipMemcpy2DAsync(out, w * sizeof(uint32_t), h->d_xa.p + word_lo, (size_t)h->xa_stride * sizeof(uint32_t),
                             w * sizeof(uint32_t), nm, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RIBBIT_OK;
}

int ribbit_hip_xa_words_strided(RibbitHandle *h, int64_t word_lo, int64_t word_hi, uint32_t *out, int64_t out_stride) {
    if (!h || !out) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded || !h->d_xa.p || h->xa_stride == 0) return fail(RIBBIT_E_STATE, "the anchored kernel has not run on this record");
    if (word_lo < 0 || word_hi < word_lo || word_hi > h->xa_stride || out_stride < word_hi - word_lo) return fail(RIBBIT_E_ARG, "word range outside the planes");
    int rc;
    if ((rc = bind_device(h))) return rc;
    const size_t nm = (size_t)(h->params.max_motif - h->params.min_motif + 1);
    const size_t w = (size_t)(word_hi - word_lo);
    if (w == 0) return RIBBIT_OK;
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)out_stride * sizeof(uint32_t), h->d_xa.p + word_lo, (size_t)h->xa_stride * sizeof(uint32_t),
                             w * sizeof(uint32_t), nm, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RIBBIT_OK;
}

// One window stage of one chunk of a longer record, on the device end to end (include/ribbit_hip.h).
int ribbit_hip_stage_calls_chunk(RibbitHandle *h, int stage, int64_t own_lo, int64_t own_hi, int64_t pos_offset, int64_t record_length,
                                 RibbitChunkCalls *out) {
    if (!h || !out) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (stage != RIBBIT_STAGE_SUBST && stage != RIBBIT_STAGE_ANCHORED) return fail(RIBBIT_E_ARG, "stage must be RIBBIT_STAGE_SUBST or RIBBIT_STAGE_ANCHORED");
    if (own_lo < 0 || own_hi < own_lo || pos_offset < 0 || pos_offset + h->length > record_length || record_length > INT32_MAX)
        return fail(RIBBIT_E_ARG, "bad chunk geometry: own [%lld, %lld), piece of %lld bases at %lld of a record of %lld", (long long)own_lo, (long long)own_hi,
                    (long long)h->length, (long long)pos_offset, (long long)record_length);
    const int64_t s = h->params.max_motif + 2;
    const bool first = pos_offset == 0, last = pos_offset + h->length == record_length;
    const int64_t reach_left = 2 * s + 8, reach_right = 4 * s + 16;       // how far a streak event depends on the sequence (the anchored kernel's are the larger)
    if (!last && own_hi > h->length - reach_right)
        return fail(RIBBIT_E_ARG, "the piece must reach %lld bases beyond the chunk's own range (own_hi %lld, piece %lld)", (long long)reach_right, (long long)own_hi, (long long)h->length);
    // (+ GROUP_FILTER_MAX: a group the anchored scan's group filter drops never reaches window_calls_kernel, where condition
    // (b) is tested; a group cut by the piece's artificial left end can only be dropped wrongly if it also ends within the
    // filter's span of the first exact position -- with this margin its call cannot be an owned one)
    const int64_t min_left = reach_left + 32 + rb::GROUP_FILTER_MAX;
    if (!first && own_lo < min_left) return fail(RIBBIT_E_ARG, "the piece must start at least %lld bases before the chunk's own range", (long long)min_left);
    std::memset(out, 0, sizeof *out);
    out->tail_pend = -1;
    int rc;
    ChunkWindow cw;
    cw.own_lo = (uint32_t)own_lo;
    cw.own_hi = (uint32_t)std::min<int64_t>(own_hi, 0xffffffffll);
    cw.z_lo = first ? 0u : (uint32_t)reach_left;
    cw.keep_flush = last && own_hi > h->length;      // the end-of-sequence calls are made at position record_length: its owner's
    cw.pos_offset = (int32_t)pos_offset;
    bool inexact = false;
    if (!first) {
        // (a) of the left-halo condition: an evaluated window between the first exact position and the own range, so that
        // every group reported at an owned position ends inside the exact part of the piece
        if ((rc = ensure_host_planes(h))) return rc;
        const int64_t q0 = h->host.first_evaluated(reach_left + 16);
        if (q0 < 0 || q0 + 7 >= own_lo) inexact = true;
    }
    DeviceCalls dc;
    const int which = stage == RIBBIT_STAGE_SUBST ? 1 : 2;
    if (which == 2 && (rc = prepare_anchored(h))) return rc;
    if ((rc = window_stage_device(h, which, false, which == 1 ? rb::subst_seedlen_cutoff : rb::anchored_seedlen_cutoff, &dc, &cw))) return rc;
    if (which == 2) h->xa_on_device = true;
    out->calls = dc.calls; out->n = dc.n;
    out->pend = dc.pend;
    out->tail_pend = dc.tail_pend;
    out->flush = dc.flush; out->n_flush = dc.n_flush;
    out->inexact = (inexact || cw.inexact) ? 1 : 0;
    out->dev_calls = h->d_dense.p;
    out->dev_pend = dc.pend ? h->d_pend.p : nullptr;
    out->streaks = h->last_streaks;
    return RIBBIT_OK;
}

// The merging rank's half of the chunk-sharded path: the three seed-list merges and the dispatch merge over what the
// chunks kept (include/ribbit_hip.h).
int ribbit_host_merge_chunks(const RibbitScanParams *params, int64_t length,
                             const uint32_t *hi, const uint32_t *lo, const uint32_t *brk, size_t nwords,
                             const uint32_t *xa, size_t xa_stride, const RibbitChunkPart *parts, size_t nparts,
                             RibbitSeedLists *out) {
    if (!params || !out || (nparts && !parts) || (length > 0 && (!hi || !lo || !brk))) return fail(RIBBIT_E_ARG, "null argument");
    const size_t need = (size_t)(length / 32 + 1) + (size_t)(params->max_motif + 2) / 32 + 2;
    if (nwords < need) return fail(RIBBIT_E_ARG, "planes too short: %zu words, need %zu (zero padding past the record)", nwords, need);
    if (xa && xa_stride < (size_t)(length / 32 + 1)) return fail(RIBBIT_E_ARG, "composed planes (xa) too short");
    for (size_t p = 0; p < nparts; ++p) {
        const RibbitChunkPart &pt = parts[p];
        if ((pt.n_runs && !pt.runs) || (pt.n_halves && !pt.halves) || (pt.subst.n && !pt.subst.calls) || (pt.anchored.n && !pt.anchored.calls) ||
            (pt.subst.n_flush && !pt.subst.flush) || (pt.anchored.n_flush && !pt.anchored.flush))
            return fail(RIBBIT_E_ARG, "chunk %zu: null array", p);
        if (pt.subst.inexact || pt.anchored.inexact) return fail(RIBBIT_E_ARG, "chunk %zu is marked inexact (left halo too short): scan it again with a longer halo", p);
    }
    std::memset(out, 0, sizeof *out);
    try {
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
        const unsigned threads = rb::merge_threads(0);

        // ---- perfect stage: the chunks' complete runs, and the runs a chunk edge cut, paired across chunks
        {
            std::vector<RibbitRun> runs, starts, ends;
            for (size_t p = 0; p < nparts; ++p) {
                for (size_t i = 0; i < parts[p].n_runs; ++i)
                    if (parts[p].runs[i].term >= 0) runs.push_back(parts[p].runs[i]);
                for (size_t i = 0; i < parts[p].n_halves; ++i) {
                    const RibbitRun &hv = parts[p].halves[i];
                    if (hv.term == RIBBIT_RUN_HALF_START) starts.push_back(hv);
                    else if (hv.term >= RIBBIT_RUN_HALF_END) ends.push_back(hv);
                    else return fail(RIBBIT_E_ARG, "chunk %zu: half record with term %d", p, hv.term);
                }
            }
            if (starts.size() != ends.size()) return fail(RIBBIT_E_ARG, "%zu open run starts but %zu orphan run ends across the chunks", starts.size(), ends.size());
            std::sort(starts.begin(), starts.end(), [](const RibbitRun &a, const RibbitRun &b) { return a.mlen != b.mlen ? a.mlen < b.mlen : a.start < b.start; });
            std::sort(ends.begin(), ends.end(), [](const RibbitRun &a, const RibbitRun &b) { return a.mlen != b.mlen ? a.mlen < b.mlen : a.end < b.end; });
            for (size_t i = 0; i < starts.size(); ++i) {
                if (starts[i].mlen != ends[i].mlen || ends[i].end <= starts[i].start ||
                    (i + 1 < starts.size() && starts[i + 1].mlen == starts[i].mlen && starts[i + 1].start <= ends[i].end))
                    return fail(RIBBIT_E_ARG, "the chunks' run halves do not pair up");
                runs.push_back(RibbitRun{starts[i].start, ends[i].end, starts[i].mlen, ends[i].term - RIBBIT_RUN_HALF_END});
            }
            rb::CallVec calls;
            rb::perfect_calls_from_runs(runs.data(), runs.size(), length, sl.min_shift, calls);
            for (const RibbitCall &c : calls) rb::perfect_add(sl, c.start, c.end, c.mlen);
        }

        // ---- a window stage's calls of all chunks as one list.  The chunks own disjoint, increasing ranges of scan
        // positions, so their lists back to back are in call order; an edge call's cursor bound also covers every call of the
        // chunks before it (their tail_pend); the end-of-sequence calls are the last chunk's.
        struct Joined { rb::CallVec calls; std::vector<int32_t> pend; rb::KeptCalls kc; };
        auto join = [&](bool anchored, Joined &j) -> int {
            size_t total = 0;
            bool any_pend = false, contiguous = true;
            for (size_t p = 0; p < nparts; ++p) {
                const RibbitChunkCalls &c = anchored ? parts[p].anchored : parts[p].subst;
                total += c.n;
                any_pend = any_pend || (c.pend && c.n);
            }
            // where the transport has put the chunks' lists back to back already (one gather buffer), they are read in place
            const RibbitCall *first_calls = nullptr;
            for (size_t p = 0; p < nparts && !first_calls; ++p) { const RibbitChunkCalls &c = anchored ? parts[p].anchored : parts[p].subst; if (c.n) first_calls = c.calls; }
            {
                size_t at = 0;
                for (size_t p = 0; p < nparts; ++p) {
                    const RibbitChunkCalls &c = anchored ? parts[p].anchored : parts[p].subst;
                    if (c.n && c.calls != first_calls + at) contiguous = false;
                    at += c.n;
                }
            }
            if (!contiguous) {
                j.calls.resize(total);
                size_t at = 0;
                for (size_t p = 0; p < nparts; ++p) {
                    const RibbitChunkCalls &c = anchored ? parts[p].anchored : parts[p].subst;
                    if (c.n) std::memcpy(j.calls.data() + at, c.calls, c.n * sizeof(RibbitCall));
                    at += c.n;
                }
            }
            // A call's cursor bound is the largest end of ANY call before it: this chunk's share (pend[i], which the device
            // works out for the calls that need one) and every call of the chunks before (their tail_pend).  For an ordinary
            // call the second part is moot like the first (no earlier call ends beyond pos - 8 = its own end), so it is
            // folded into every entry: which calls are made at an N cannot be seen from here.
            int32_t before = -1, tail = -1;          // largest end of any call of the chunks before this one / of all chunks
            any_pend = any_pend || nparts > 1;
            if (any_pend) j.pend.assign(total, -1);
            size_t at = 0;
            const RibbitCall *flush = nullptr; size_t n_flush = 0;
            int32_t last_pos = -1;
            for (size_t p = 0; p < nparts; ++p) {
                const RibbitChunkCalls &c = anchored ? parts[p].anchored : parts[p].subst;
                if (c.n) {
                    if (c.calls[0].pos < last_pos) return fail(RIBBIT_E_ARG, "chunk %zu's calls start before the previous chunk's end: the own ranges must increase", p);
                    last_pos = c.calls[c.n - 1].pos;
                }
                if (any_pend)
                    for (size_t i = 0; i < c.n; ++i) j.pend[at + i] = std::max(c.pend ? c.pend[i] : -1, before);
                at += c.n;
                tail = std::max(tail, c.tail_pend);
                before = tail;
                if (c.n_flush) {
                    if (flush) return fail(RIBBIT_E_ARG, "end-of-sequence calls from more than one chunk");
                    flush = c.flush; n_flush = c.n_flush;
                }
            }
            j.kc.calls = contiguous ? first_calls : j.calls.data();
            j.kc.n = total;
            j.kc.pend = any_pend ? j.pend.data() : nullptr;
            j.kc.tail_pend = tail;
            j.kc.flush = flush; j.kc.n_flush = n_flush;
            return RIBBIT_OK;
        };
        int rc;
        {
            Joined j;
            if ((rc = join(false, j))) return rc;
            rb::merge_subst_stage(sl, j.kc, threads);
        }
        hp.xa.clear();
        hp.xa_view = xa;                                     // null: recomputed slice by slice from the packed planes
        hp.xa_stride = xa ? (int64_t)xa_stride : 0;
        hp.xa_m_lo = params->min_motif;
        hp.xa_m_hi = params->max_motif;
        sl.range_count = [&hp](int shift, int start, int end) {
            return hp.has_xa(shift) ? hp.range_count_xa(shift, start, end) : hp.range_count(shift, start, end);
        };
        if (hp.xa_stored()) { sl.plane_words = hp.xa_words(); sl.plane_stride = hp.xa_stride; sl.plane_lo = hp.xa_m_lo; sl.plane_hi = hp.xa_m_hi; }
        rb::SeedVec dispatch;
        {
            Joined j;
            if ((rc = join(true, j))) return rc;
            rb::MergeStats st;
            rb::merge_anchored_stage(sl, j.kc, threads, &st);
            rb::dispatch_order_ranges(sl, st.cut_pos, threads, dispatch);
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
    } catch (const std::bad_alloc &) {
        ribbit_seed_lists_free(out);
        return fail(RIBBIT_E_NOMEM, "out of host memory in the merge of the chunks");
    }
    return RIBBIT_OK;
}

}  // extern "C"
