// api.cpp -- extern "C" boundary of libribbit_hip.so (see include/ribbit_hip.h).
// Device memory, streams and HIP-event timing live here; kernels are in kernels.hip, the
// per-motif window state machines in window_stage.hip (device) and the sequential seed-list merges in
// seed_lists.cpp / parallel_merge.cpp.  There is no CPU fallback for any scan anywhere in this library.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "device_planes.h"
#include "event_stream.h"
#include "host_planes.h"
#include "kernels.h"
#include "parallel_merge.h"
#include "refine.h"
#include "ssw_exact.h"
#include "ribbit_hip.h"
#include "seed_lists.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RIBBIT_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return RIBBIT_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return fail(RIBBIT_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e)); }
        cap = n;
        return RIBBIT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return RIBBIT_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc((void **)&p, n * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail(RIBBIT_E_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e)); }
        cap = n;
        return RIBBIT_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

enum Stage { STAGE_NONE = 0, STAGE_PERFECT = 1, STAGE_SUBST = 2, STAGE_ANCHORED = 3 };

}  // namespace

struct RibbitHandle {
    RibbitScanParams params{};
    int device = 0;
    int min_shift = 1, max_shift = 102;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // post stream of the perfect scan: pairing kernels and result copies, so that they overlap
                                         // the next record's kernels when several handles share `stream`
    hipEvent_t ev_ready = nullptr;       // pairing done, counters and status on the host
    hipEvent_t ev[6] = {};        // 0/1 pack, 2/3 scan kernel, 4/5 whole GPU side of the last scan
    bool have_timing[3] = {false, false, false};
    bool timing = true;           // record the HIP events behind ribbit_hip_last_timing_ms (each costs a barrier packet on the stream)
    double host_ms = 0.0;         // post-processing of the last scan after its pairing (device state machine, sort, read-back), wall clock
    double merge_ms = 0.0;        // sequential host merge of the last window stage, wall clock
    double subst_merge_ms = 0.0;  // ... of the substitution stage when ribbit_hip_seeds_anchored ran both
    bool xa_on_device = false;    // the anchored kernel has written the composed planes of the loaded record
    unsigned host_threads = 0;    // worker threads of the host stages (0 = RIBBIT_THREADS or min(cores, 16))

    bool loaded = false;
    int64_t length = 0;
    int64_t ntiles = 0, total_words = 0, tail_words = 0;
    DevBuf<uint8_t> d_ascii;
    DevBuf<uint32_t> d_hi, d_lo, d_brk;
    DevBuf<uint64_t> d_events, d_dense;
    DevBuf<uint32_t> d_counters;
    DevBuf<uint32_t> d_query;
    DevBuf<uint32_t> d_xa;             // composed planes XA_m, motif-major
    int64_t xa_stride = 0;
    PinnedBuf<uint64_t> h_events;
    PinnedBuf<uint32_t> h_counters;
    PinnedBuf<uint32_t> h_query;

    // host copy of the packed planes: answers the sparse, latency-bound range reads of the
    // sequential merges (retainNestedSeed & co) without a GPU round trip per query
    rb::HostPlanes host;
    bool host_planes_valid = false;

    // ordered view of the last event collection
    int64_t last_event_count = 0;
    uint32_t produced = 0;
    std::vector<uint64_t> chunk_table;   // (offset, count) per (motif, tile)
    size_t table_ntile = 0;

    bool runs_valid = false, calls_valid = false;
    std::vector<RibbitRun> runs;          // chunk-local pairing (multi-GPU path)
    // device-side pairing of the perfect scan: scratch + the pinned run list it lands in
    DevBuf<uint64_t> d_pair_table;
    DevBuf<uint32_t> d_run_base, d_pair_partial, d_pair_status;
    PinnedBuf<uint32_t> h_pub;             // region counters + pairing status, written by the GPU (pair_publish_kernel)
    uint32_t *h_pub_dev = nullptr;         // the same memory as the device sees it
    PinnedBuf<RibbitRun> h_runs, h_halves;
    rb::PairLaunch pair{};                // the perfect scan in flight (perfect_begin .. perfect_finish)
    bool pair_pending = false;
    size_t debug_first_cap = 0;           // ribbit_hip_debug_set_event_capacity: first guess of the event capacity (tests of the overflow path)
    bool counters_clean = false;          // d_counters zeroed by the pack kernel and not used since
    bool copy_pending = false;            // result copies enqueued but not yet waited for (ribbit_hip_scan_perfect_end with wait = 0)
    DevBuf<RibbitRun> d_halves;
    size_t n_runs = 0, n_halves = 0;
    // window stages on the device (window_stage.hip): scratch of the streak -> call pipeline and its pinned results
    DevBuf<uint32_t> d_eval, d_first_rev, d_word_tmp, d_last_word, d_bitmap, d_edge_tmp, d_edge_end1, d_ws_counters;
    bool eval_valid = false;              // d_eval / d_first_rev belong to the loaded record
    DevBuf<uint64_t> d_group, d_sort_keys, d_sort_vals, d_edge_keys, d_edge_vals, d_edge_keys2, d_edge_vals2;
    DevBuf<int32_t> d_min_span, d_pend, d_tj;
    DevBuf<uint32_t> d_dropmap;           // group filter of the anchored scan: ends of the groups it dropped
    bool dropmap_valid = false;           // the last anchored scan ran with the filter
    DevBuf<RibbitCall> d_flush;
    DevBuf<uint8_t> d_scratch;
    // results of the substitution [0] and anchored [1] stage, page-locked: both stages' kernels run before either merge
    PinnedBuf<RibbitCall> h_calls_[2], h_flush_[2];
    PinnedBuf<int32_t> h_pend_[2];
    PinnedBuf<uint32_t> h_ws_[2];
    PinnedBuf<uint32_t> h_xa;          // host copy of the composed planes (rb::HostPlanes::xa_view points here)
    hipEvent_t ev_xa = nullptr;        // the copy of the composed planes has landed
    hipEvent_t ev_ssw = nullptr;       // orders the longest alignment class (on the copy stream) against the compute stream
    bool xa_copy_pending = false;
    int64_t last_streaks = 0, last_calls = 0, last_edge_calls = 0;
    rb::CallVec perfect_calls;
    bool subst_calls_valid = false;
    rb::CallVec subst_calls;
    bool anchored_calls_valid = false;
    rb::CallVec anchored_calls;
    rb::SeedVec dispatch;
    bool longest_valid = false;
    std::vector<int32_t> longest_runs;
    DevBuf<RibbitSeed> d_seeds;
    DevBuf<RibbitSeed> d_seeds_small;     // the small-motif scan's own, so that it can run beside the consensus-row scan
    DevBuf<int32_t> d_longest;
    DevBuf<uint8_t> d_sym;
    bool sym_valid = false;                                 // d_sym holds the loaded record
    DevBuf<uint32_t> d_small_records, d_small_count;        // possibleMotifs of the dispatched seeds (small_motifs.hip)
    DevBuf<int32_t> d_small_head;
    PinnedBuf<int32_t> small_head;                          // 4 per dispatched seed; flags (4 i + 3) != 0: no device result
    PinnedBuf<uint32_t> small_records;
    size_t n_small_records = 0;
    bool small_valid = false;
    DevBuf<unsigned long long> d_best;
    DevBuf<int32_t> d_slices;
    DevBuf<int32_t> d_ssw_jobs, d_ssw_order, d_ssw_out;   // batched striped passes (ssw_kernels.hip)
    DevBuf<uint8_t> d_ssw_pool;
    DevBuf<int32_t> d_path_items, d_path_result;
    DevBuf<uint64_t> d_path_cell_off, d_path_ops_off;
    DevBuf<uint8_t> d_path_cells;
    DevBuf<uint32_t> d_path_scratch, d_path_ops, d_path_count;
    PinnedBuf<uint32_t> h_path_ops;
    std::vector<rb::SswPath> ssw_paths;                    // per job of h->jobs: the path the GPU found (ops == null: none)
    std::vector<rb::SswEnds> ssw_ends;                     // per job of h->jobs; flag -1 = not computed on the GPU          // {job, first row} per 64-row slice of the long-motif seeds
    bool best_rows_valid = false;
    std::vector<int32_t> best_rows;       // per dispatch seed: mostFrequentLongerMotif's window start, or -1
    std::vector<RibbitAlignJob> jobs;
    std::string motif_pool;
    std::vector<uint64_t> export_events, export_counts;
    std::string host_ascii;   // the record's bases on the host when they had to be fetched back (refinement slices them for the aligner)
    bool host_ascii_valid = false;
    const char *host_bases = nullptr;     // where refinement reads the bases: the caller's page-locked buffer (load_record_pinned) or host_ascii
    hipStream_t up_stream = nullptr;      // uploads: the next record's bases travel while this record's kernels run
    hipEvent_t ev_up = nullptr, ev_busy = nullptr;
    hipEvent_t ev_stage[2][2] = {};       // scan kernel of the substitution [0] / anchored [1] stage
    hipEvent_t ev_planes = nullptr;       // between the two kernels of the anchored stage (planes | window scan)
    bool planes_timing_valid = false;     // ev_stage[1][0] .. ev_planes .. ev_stage[1][1] bracket the two kernels of one run
    bool have_stage_timing[2] = {false, false};
    const uint8_t *dev_ascii_src = nullptr;
    std::string bed;
    int stage_done = STAGE_NONE;          // how far the seed lists have been advanced
    rb::SeedLists lists;
    RibbitHandle *aux = nullptr;          // helper handle of ribbit_hip_refine_bed: streams and buffers of the long alignment batch
    RibbitHandle *aux2 = nullptr;         // ... and of its second feeder (every other slice of the short alignments)
    RibbitAlignBatcher *batcher = nullptr;    // shared alignment batches of the records in flight (ribbit_hip_set_batcher)

    rb::DevicePlanes planes() const {
        rb::DevicePlanes pl;
        pl.hi = d_hi.p + rb::LEAD_WORDS;
        pl.lo = d_lo.p + rb::LEAD_WORDS;
        pl.brk = d_brk.p + rb::LEAD_WORDS;
        pl.length = length;
        pl.ntiles = ntiles;
        pl.tail_words = tail_words;
        return pl;
    }
};

namespace {

int bind_device(const RibbitHandle *h) {
    HIP_TRY(hipSetDevice(h->device));
    return RIBBIT_OK;
}

int is_gfx950(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int perfect_wait(RibbitHandle *h);

int pack_loaded_ascii(RibbitHandle *h, const uint8_t *dev_ascii, int64_t length) {
    // a perfect scan enqueued with ribbit_hip_scan_perfect_begin still reads the planes and counters the pack kernel
    // is about to rewrite
    if (h->pair_pending) return fail(RIBBIT_E_STATE, "a perfect scan is in flight on this handle: call ribbit_hip_scan_perfect_end first");
    if (h->copy_pending) { const int rcw = perfect_wait(h); if (rcw) return rcw; }
    h->loaded = false;
    h->dev_ascii_src = dev_ascii;
    h->runs_valid = h->calls_valid = h->subst_calls_valid = h->anchored_calls_valid = false;
    h->longest_valid = false;
    h->best_rows_valid = false;
    h->small_valid = false;
    h->sym_valid = false;
    h->host_planes_valid = false;
    h->eval_valid = false;
    h->xa_on_device = false;
    if (h->xa_copy_pending) { (void)hipEventSynchronize(h->ev_xa); h->xa_copy_pending = false; }
    h->stage_done = STAGE_NONE;
    h->length = length;
    const int64_t nwords = length / 32 + 1;   // word holding position L is included
    h->ntiles = (nwords + rb::TILE_WORDS - 1) / rb::TILE_WORDS;
    h->tail_words = h->max_shift / 32 + rb::TAIL_SLACK_WORDS;
    h->total_words = rb::LEAD_WORDS + h->ntiles * rb::TILE_WORDS + h->tail_words;
    int rc;
    if ((rc = h->d_hi.ensure((size_t)h->total_words))) return rc;
    if ((rc = h->d_lo.ensure((size_t)h->total_words))) return rc;
    if ((rc = h->d_brk.ensure((size_t)h->total_words))) return rc;
    if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    if ((rc = h->d_counters.ensure(rb::EV_COUNTER_WORDS))) return rc;
    rb::launch_pack(dev_ascii, length, h->d_hi.p, h->d_lo.p, h->d_brk.p, h->total_words, h->d_counters.p, rb::EV_COUNTER_WORDS, h->stream);
    h->counters_clean = true;      // until a scan kernel runs
    HIP_TRY(hipGetLastError());
    if (h->timing) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    h->have_timing[0] = h->timing;
    // the previous record's seed lists are emptied, not freed: giving half a gigabyte back to the system took 98 ms after a
    // chromosome (munmap walks every page), and the next record's merges then faulted the same pages in again
    h->lists.perfect.clear(); h->lists.subst.clear(); h->lists.anchored.clear();
    h->lists.range_count = nullptr;
    h->lists.guard_hits = 0;
    h->lists.plane_words = nullptr; h->lists.plane_stride = 0; h->lists.plane_lo = 0; h->lists.plane_hi = -1;
    h->lists.length = length;
    h->lists.min_motif = h->params.min_motif;
    h->lists.max_motif = h->params.max_motif;
    h->lists.min_shift = h->min_shift;
    h->loaded = true;
    return RIBBIT_OK;
}

// D2H of the packed planes (3 bits per base), once per record, for the host-side sparse reads
int ensure_host_planes(RibbitHandle *h) {
    if (h->host_planes_valid) return RIBBIT_OK;
    int rc;
    if ((rc = bind_device(h))) return rc;
    const size_t n = (size_t)(h->ntiles * rb::TILE_WORDS + h->tail_words);
    h->host.resize(h->length, n);
    HIP_TRY(hipMemcpyAsync(h->host.hi.data(), h->d_hi.p + rb::LEAD_WORDS, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->host.lo.data(), h->d_lo.p + rb::LEAD_WORDS, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->host.brk.data(), h->d_brk.p + rb::LEAD_WORDS, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->host.index_breaks();
    h->host_planes_valid = true;
    return RIBBIT_OK;
}

// Launch one scan kernel, compact its sharded event regions, copy the events back and index the
// (motif, tile) chunks.  which: 0 perfect run scan, 1 window scan (1 mismatch), 2 fused anchored scan.
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
        if (which == 0) rb::launch_scan_perfect(pl, pp, h->d_events.p, h->d_counters.p, h->stream);
        else if (which == 1) rb::launch_scan_window(pl, pp, 1, h->d_events.p, h->d_counters.p, h->stream);
        else rb::launch_scan_anchored(pl, pp, h->d_xa.p, h->xa_stride, h->d_events.p, h->d_counters.p, nullptr, nullptr, h->stream);
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

int perfect_finish(RibbitHandle *h, RibbitRun *dst, size_t dst_cap, RibbitRun *half_dst, size_t half_dst_cap, bool wait = true) {
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
                           RibbitRun *half_dst = nullptr, size_t half_dst_cap = 0) {
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

// ---- window stages on the device ---------------------------------------------------------------------
// scan kernel -> pass-streak START / END events (left in their regions) -> pairing kernels -> one 16-byte record per
// streak, motif-major by start, in d_dense.  which: 1 window scan (1 mismatch), 2 fused anchored scan.
// filter (anchored scan only): groups of pass-streaks whose call cannot pass min_span leave no events (kernels.hip, "group
// filter"); their ends are left in h->d_dropmap for window_stage_device.
int scan_and_pair_streaks(RibbitHandle *h, int which, uint32_t *n_streaks, int (*filter_min_span)(int) = nullptr) {
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
    // the anchored stage runs as two kernels (planes, then the window scan of the planes: kernels.hip) unless
    // RIBBIT_FUSED_ANCHORED=1 asks for the fused one (a measurement knob)
    static const bool fused_anchored = std::getenv("RIBBIT_FUSED_ANCHORED") && std::atoi(std::getenv("RIBBIT_FUSED_ANCHORED")) != 0;
    pr.tile_bases = (which == 2 && fused_anchored) ? (uint32_t)rb::anchored_tile_words(rb::anchored_halo_lanes(h->params.max_motif)) * 32u : (uint32_t)rb::TILE_BASES;
    pr.ntile = (uint32_t)(h->length / pr.tile_bases + 1);
    pr.own_lo = 0; pr.own_hi = INT64_MAX; pr.pos_offset = 0;
    const size_t entries = (size_t)pr.nm * pr.ntile;
    if (entries > 0xfffffff0u) return fail(RIBBIT_E_ARG, "record too long for %u motif sizes", pr.nm);
    if ((rc = h->d_pair_table.ensure(entries))) return rc;
    if ((rc = h->d_run_base.ensure(entries))) return rc;
    if ((rc = h->d_pair_partial.ensure(entries / 1024 + 2))) return rc;
    if ((rc = h->d_halves.ensure(2 * (size_t)pr.nm))) return rc;
    // typical event densities on repeat-rich sequence: 0.25 per base (1-mismatch windows), 3.7 (anchored windows at 99
    // motif sizes); a too small first guess costs a second launch
    const size_t per_base_x4 = which == 1 ? 2 : (size_t)std::max(16, (h->params.max_motif - h->params.min_motif + 1) / 6);
    size_t cap = std::max<size_t>((size_t)1 << 20, (size_t)h->length * per_base_x4 / 4);
    cap = std::max(cap, h->d_events.cap);
    if (h->debug_first_cap) cap = h->debug_first_cap;
    const rb::DevicePlanes pl = h->planes();
    uint64_t produced = 0;
    const bool filter = which == 2 && filter_min_span != nullptr && std::getenv("RIBBIT_NO_GROUP_FILTER") == nullptr;
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
        if (attempt == 0 || which == 1 || fused_anchored) HIP_TRY(hipEventRecord(h->ev_stage[which - 1][0], h->stream));
        if (which == 1) rb::launch_scan_window(pl, pp, 1, h->d_events.p, h->d_counters.p, h->stream);
        else if (fused_anchored) rb::launch_scan_anchored(pl, pp, h->d_xa.p, h->xa_stride, h->d_events.p, h->d_counters.p, filter ? h->d_tj.p : nullptr,
                                                          filter ? h->d_dropmap.p : nullptr, h->stream);
        else {
            if (attempt == 0) {          // the planes do not depend on the event capacity: once
                rb::launch_scan_anchored(pl, pp, h->d_xa.p, h->xa_stride, nullptr, nullptr, nullptr, nullptr, h->stream);
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
    if (which == 2) h->planes_timing_valid = !fused_anchored && first_attempt_fit;
    return RIBBIT_OK;
}

using DeviceCalls = rb::KeptCalls;      // views of handle-owned page-locked memory

// The loaded record as one chunk (plus halos) of a longer record: which scan positions are this chunk's (piece
// coordinates), from where on the piece's streak events are exact (0: the piece starts where the record starts), whether
// the piece ends where the record ends, and what to add to piece coordinates to get record coordinates.
struct ChunkWindow {
    uint32_t own_lo = 0, own_hi = 0xffffffffu, z_lo = 0;
    bool keep_flush = true;
    int32_t pos_offset = 0;
    bool inexact = false;          // out: an owned call's group reaches the piece's artificial left end
};

// The whole window stage on the device (window_stage.hip).  full: every call, unfiltered (the call-list entry points
// and the parity tests); otherwise only the calls that pass min_span, with their cursor bounds.  cw: chunk mode.
int window_stage_device(RibbitHandle *h, int which, bool full, int (*min_span)(int), DeviceCalls *out, ChunkWindow *cw = nullptr) {
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
    std::fprintf(stderr, "[anchored merge] %zu seeds: %u ranges on %u threads%s, %u passes, %lld changing head writes, %u ranges done again, preparation %.1f ms, merges %.1f ms "
                 "(parallel passes %.1f ms over %u range runs: the ranges' own times sum to %.1f ms = %.1f ms per thread, longest range %.1f ms; in-order walk %.1f ms; joining the ranges' lists %.1f ms; before the first pass %.1f ms; end-of-sequence calls %.1f ms), dispatch order %.1f ms in %u ranges\n",
                 seeds, st.ranges, st.threads, st.redone_in_order ? " (REDONE IN ORDER)" : (st.head_writes ? " (list-head writes: ranges done again, see passes)" : ""), st.passes, st.head_writes,
                 st.ranges_redone, st.prepare_ms, st.merge_ms, st.pass_ms, st.ranges_run, st.range_ms_sum, st.range_ms_sum / std::max(1u, st.threads), st.range_ms_max, st.walk_ms, st.concat_ms, st.before_passes_ms, st.flush_ms, dispatch_ms, dispatch_ranges);
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
    else rb::merge_anchored_stage(h->lists, dca, threads, &st);
    const double t1 = now_ms();
    const unsigned dispatch_ranges = rb::dispatch_order_ranges(h->lists, st.cut_pos, threads, h->dispatch);
    h->merge_ms = now_ms() - t0;
    static const bool profile = std::getenv("RIBBIT_PROFILE") != nullptr;
    if (profile) print_anchored_merge_profile(h->lists.anchored.size(), st, now_ms() - t1, dispatch_ranges);
    h->subst_merge_ms = subst_todo ? merge_s : 0.0;
    h->stage_done = STAGE_ANCHORED;
    return RIBBIT_OK;
}

// longestContinuousMatches of every dispatched seed, one GPU launch (a13)
int build_longest_runs(RibbitHandle *h) {
    if (h->longest_valid) return RIBBIT_OK;
    int rc = advance_to_anchored(h);
    if (rc) return rc;
    if ((rc = bind_device(h))) return rc;
    const size_t n = h->dispatch.size();
    h->longest_runs.assign(n, 0);
    if (n) {
        if ((rc = h->d_seeds.ensure(n))) return rc;
        if ((rc = h->d_longest.ensure(n))) return rc;
        HIP_TRY(hipMemcpyAsync(h->d_seeds.p, h->dispatch.data(), n * sizeof(RibbitSeed), hipMemcpyHostToDevice, h->stream));
        rb::launch_seed_longest_runs(h->d_xa.p, h->xa_stride, h->params.min_motif, h->d_seeds.p, (int64_t)n, h->d_longest.p, h->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->longest_runs.data(), h->d_longest.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
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
        if ((rc = h->d_seeds.ensure(jobs.size()))) return rc;
        if ((rc = h->d_best.ensure(jobs.size()))) return rc;
        if (!h->sym_valid) { rb::launch_sym(h->dev_ascii_src, h->length, h->d_sym.p, h->stream); HIP_TRY(hipGetLastError()); h->sym_valid = true; }
        HIP_TRY(hipMemcpyAsync(h->d_seeds.p, jobs.data(), jobs.size() * sizeof(RibbitSeed), hipMemcpyHostToDevice, h->stream));
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
        rb::launch_long_motif_rows(h->d_sym.p, h->length, h->d_seeds.p, (int64_t)jobs.size(), h->d_slices.p,
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
    h->best_rows.assign(h->dispatch.size(), -1);
    if ((rc = best_rows_of(h, prm, h->dispatch, h->longest_runs.data(), h->best_rows.data()))) return rc;
    h->best_rows_valid = true;
    return RIBBIT_OK;
}

// possibleMotifs of every dispatched seed with m <= 10 that reaches it (parse_smallmotif_seed.cpp:234-236), one GPU
// launch (a14 / f2); seeds the kernel flags (more than 64 classes) keep flags != 0 and the host twin runs for them
// `stream`: where its copies and its kernel go (default: the handle's).  With another stream it may run beside build_best_rows on
// another thread, PROVIDED the longest runs and the symbols are there already (scan_seeds_side_by_side sees to that).
int build_small_motifs(RibbitHandle *h, const RibbitRefineParams &prm, hipStream_t stream = nullptr) {
    if (h->small_valid) return RIBBIT_OK;
    if (!stream) stream = h->stream;
    int rc = build_longest_runs(h);
    if (rc) return rc;
    const size_t n = h->dispatch.size();
    const double t0 = now_ms();
    if ((rc = h->small_head.ensure(std::max<size_t>(4 * n, 4)))) return rc;
    h->n_small_records = 0;
    rb::SeedVec jobs;          // reused as int4 {seed start, seed end, m, dispatch index}
    jobs.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const RibbitSeed &s = h->dispatch[i];
        if (s.mlen > 10 || s.mlen < 1 || h->longest_runs[i] < prm.continuous_ones_threshold) continue;
        jobs.push_back(RibbitSeed{s.start, s.end, s.mlen, (int32_t)i});
    }
    if (jobs.empty()) {
        for (size_t i = 0; i < n; ++i) h->small_head.p[4 * i + 3] = -1;
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
        const size_t cap = std::min<size_t>(4 * jobs.size() + (1u << 20), 0x7fffffffu);
        if ((rc = h->d_sym.ensure((size_t)h->length + 16)) || (rc = h->d_seeds_small.ensure(jobs.size())) || (rc = h->d_small_head.ensure(4 * n)) ||
            (rc = h->d_small_records.ensure(4 * cap)) || (rc = h->d_small_count.ensure(4)))
            return rc;
        if (!h->sym_valid) { rb::launch_sym(h->dev_ascii_src, h->length, h->d_sym.p, stream); HIP_TRY(hipGetLastError()); h->sym_valid = true; }
        HIP_TRY(hipMemcpyAsync(h->d_seeds_small.p, jobs.data(), jobs.size() * sizeof(RibbitSeed), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(h->d_small_count.p, 0, 4 * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(h->d_small_head.p, 0xff, 4 * n * sizeof(int32_t), stream));      // flags -1: no device result
        rb::launch_small_motifs(h->d_sym.p, h->length, h->d_seeds_small.p, (int64_t)jobs.size(), lim, h->d_small_records.p, (uint32_t)cap, h->d_small_count.p,
                                h->d_small_head.p, stream);
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
    if (profile) std::fprintf(stderr, "[small motifs] %zu seeds on the GPU, %zu records, %.1f ms incl. transfers\n", jobs.size(), h->n_small_records, now_ms() - t0);
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

}  // namespace

extern "C" {

void ribbit_scan_params_default(RibbitScanParams *p, int32_t min_motif, int32_t max_motif) {
    if (!p) return;
    p->min_motif = min_motif;
    p->max_motif = max_motif;
    p->window_length = 8;
    p->subst_threshold = 7;
    p->anchor_threshold = 6;
    p->anchor_length = 3;
}

const char *ribbit_hip_last_error(void) { return g_last_error.c_str(); }
int ribbit_hip_abi_version(void) { return RIBBIT_ABI_VERSION; }

int ribbit_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) ok += is_gfx950(d);
    return ok;
}

int ribbit_hip_open(const RibbitScanParams *params, int device, RibbitHandle **out) {
    if (!params || !out) return fail(RIBBIT_E_ARG, "null argument");
    *out = nullptr;
    if (params->min_motif < 1 || params->max_motif < params->min_motif || params->max_motif > 990)
        return fail(RIBBIT_E_ARG, "motif range [%d,%d] not supported (1 <= m <= M <= 990)", params->min_motif, params->max_motif);
    if (params->window_length != 8) return fail(RIBBIT_E_ARG, "window_length must be 8");
    if (params->subst_threshold != 7 || params->anchor_threshold != 6 || params->anchor_length != 3)
        return fail(RIBBIT_E_ARG, "only the reference's fixed thresholds are supported (7, 6, anchor 3: ribbit.cpp:191, fasta_utils.cpp:165)");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RIBBIT_E_DEVICE, "no HIP device available (%s); ribbit_amd has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RIBBIT_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    if (!is_gfx950(device)) return fail(RIBBIT_E_DEVICE, "device %d is not gfx950 (MI355X); kernels are built for gfx950 only", device);
    RibbitHandle *h = new (std::nothrow) RibbitHandle();
    if (!h) return fail(RIBBIT_E_NOMEM, "out of host memory");
    h->params = *params;
    h->device = device;
    h->min_shift = (params->min_motif > 2) ? params->min_motif - 2 : 1;   // ribbit.cpp:241
    h->max_shift = params->max_motif + 2;                                   // ribbit.cpp:242
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&h->ev_xa, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&h->ev_ssw, hipEventDisableTiming);
    // The upload stream is created by the first upload that uses it.  Measured (bench.py, three handles on one shared
    // compute stream, same box, alternating runs): with an unused upload stream per handle a step takes 0.267 ms, without
    // 0.245 ms (round 1: 0.243), while the scan kernel's own time is unchanged.  Presumably the extra streams change which
    // hardware queue the handles' post streams share, so that the pairing chain no longer overlaps the next scan; that
    // part is inferred, not observed.
    if (err == hipSuccess) err = hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&h->ev_busy, hipEventDisableTiming);
    for (int i = 0; i < 4 && err == hipSuccess; ++i) err = hipEventCreate(&h->ev_stage[i / 2][i % 2]);
    if (err == hipSuccess) err = hipEventCreate(&h->ev_planes);
    for (int i = 0; i < 6 && err == hipSuccess; ++i) err = hipEventCreate(&h->ev[i]);
    if (err != hipSuccess) {
        delete h;
        return fail(RIBBIT_E_DEVICE, "device setup failed: %s", hipGetErrorString(err));
    }
    h->stream = h->own_stream;
    *out = h;
    return RIBBIT_OK;
}

int ribbit_hip_close(RibbitHandle *h) {
    if (!h) return RIBBIT_OK;
    if (h->aux) { (void)ribbit_hip_close(h->aux); h->aux = nullptr; }
    if (h->aux2) { (void)ribbit_hip_close(h->aux2); h->aux2 = nullptr; }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    h->d_ascii.release(); h->d_hi.release(); h->d_lo.release(); h->d_brk.release();
    h->d_events.release(); h->d_dense.release(); h->d_counters.release(); h->d_query.release(); h->d_xa.release(); h->d_seeds.release(); h->d_seeds_small.release(); h->d_longest.release(); h->d_sym.release(); h->d_best.release(); h->d_slices.release();
    h->d_ssw_jobs.release(); h->d_ssw_order.release(); h->d_ssw_out.release(); h->d_ssw_pool.release();
    h->d_path_items.release(); h->d_path_result.release(); h->d_path_cell_off.release(); h->d_path_ops_off.release(); h->d_path_cells.release();
    h->d_path_scratch.release(); h->d_path_ops.release(); h->d_path_count.release(); h->h_path_ops.release();
    h->d_small_head.release(); h->d_small_records.release(); h->d_small_count.release(); h->small_head.release(); h->small_records.release();
    h->h_events.release(); h->h_counters.release(); h->h_query.release();
    h->d_pair_table.release(); h->d_run_base.release(); h->d_pair_partial.release(); h->d_pair_status.release();
    h->d_tj.release(); h->d_dropmap.release();
    h->h_pub.release(); h->h_runs.release(); h->h_halves.release(); h->d_halves.release();
    h->d_eval.release(); h->d_first_rev.release(); h->d_word_tmp.release(); h->d_last_word.release(); h->d_bitmap.release();
    h->d_edge_tmp.release(); h->d_edge_end1.release(); h->d_ws_counters.release(); h->d_group.release(); h->d_sort_keys.release();
    h->d_sort_vals.release(); h->d_edge_keys.release(); h->d_edge_vals.release(); h->d_edge_keys2.release(); h->d_edge_vals2.release();
    h->d_min_span.release(); h->d_pend.release(); h->d_flush.release(); h->d_scratch.release();
    for (int k = 0; k < 2; ++k) { h->h_calls_[k].release(); h->h_flush_[k].release(); h->h_pend_[k].release(); h->h_ws_[k].release(); }
    h->h_xa.release();
    if (h->ev_xa) (void)hipEventDestroy(h->ev_xa);
    if (h->ev_ssw) (void)hipEventDestroy(h->ev_ssw);
    if (h->ev_up) (void)hipEventDestroy(h->ev_up);
    if (h->ev_busy) (void)hipEventDestroy(h->ev_busy);
    for (int i = 0; i < 4; ++i) if (h->ev_stage[i / 2][i % 2]) (void)hipEventDestroy(h->ev_stage[i / 2][i % 2]);
    if (h->ev_planes) (void)hipEventDestroy(h->ev_planes);
    if (h->up_stream) { (void)hipStreamSynchronize(h->up_stream); (void)hipStreamDestroy(h->up_stream); }
    for (int i = 0; i < 6; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
    delete h;
    return RIBBIT_OK;
}

int ribbit_hip_set_stream(RibbitHandle *h, void *hip_stream) {
    if (!h) return fail(RIBBIT_E_ARG, "null handle");
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return RIBBIT_OK;
}

// H2D of the bases on the upload stream (so that it overlaps kernels of other handles on a shared compute stream),
// then the pack kernel on the compute stream
static int upload_and_pack(RibbitHandle *h, const char *ascii, int64_t length) {
    int rc;
    if ((rc = bind_device(h))) return rc;
    if ((rc = h->d_ascii.ensure((size_t)std::max<int64_t>(length, 16)))) return rc;
    if (length) {
        // the previous record's kernels may still read d_ascii
        HIP_TRY(hipEventRecord(h->ev_busy, h->stream));
        if (!h->up_stream) HIP_TRY(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamWaitEvent(h->up_stream, h->ev_busy, 0));
        HIP_TRY(hipMemcpyAsync(h->d_ascii.p, ascii, (size_t)length, hipMemcpyHostToDevice, h->up_stream));
        HIP_TRY(hipEventRecord(h->ev_up, h->up_stream));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_up, 0));
    }
    return pack_loaded_ascii(h, h->d_ascii.p, length);
}

int ribbit_hip_load_record(RibbitHandle *h, const char *ascii, int64_t length) {
    if (!h || (!ascii && length > 0)) return fail(RIBBIT_E_ARG, "null argument");
    if (length < 0 || length >= ((int64_t)1 << 31) - 64) return fail(RIBBIT_E_ARG, "record length %lld not supported (positions are int32, fasta_utils.cpp:78)", (long long)length);
    h->host_ascii_valid = false;      // not duplicated: refinement fetches the bases back from the device if it runs
    h->host_bases = nullptr;
    return upload_and_pack(h, ascii, length);
}

int ribbit_hip_load_record_pinned(RibbitHandle *h, const char *pinned_ascii, int64_t length) {
    if (!h || (!pinned_ascii && length > 0)) return fail(RIBBIT_E_ARG, "null argument");
    if (length < 0 || length >= ((int64_t)1 << 31) - 64) return fail(RIBBIT_E_ARG, "record length %lld not supported", (long long)length);
    h->host_ascii_valid = false;
    h->host_bases = pinned_ascii;     // stays the caller's; read again by refinement
    return upload_and_pack(h, pinned_ascii, length);
}

int ribbit_hip_load_record_device(RibbitHandle *h, const void *dev_ascii, int64_t length) {
    if (!h || (!dev_ascii && length > 0)) return fail(RIBBIT_E_ARG, "null argument");
    if (length < 0 || length >= ((int64_t)1 << 31) - 64) return fail(RIBBIT_E_ARG, "record length %lld not supported", (long long)length);
    int rc;
    if ((rc = bind_device(h))) return rc;
    h->host_ascii_valid = false;
    h->host_bases = nullptr;
    return pack_loaded_ascii(h, (const uint8_t *)dev_ascii, length);
}

int ribbit_hip_host_alloc(size_t bytes, void **out) {
    if (!out || !bytes) return fail(RIBBIT_E_ARG, "bad argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(RIBBIT_E_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
    return RIBBIT_OK;
}

int ribbit_hip_host_free(void *p) {
    if (!p) return RIBBIT_OK;
    HIP_TRY(hipHostFree(p));
    return RIBBIT_OK;
}

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

// forward + reverse striped Smith-Waterman passes of every job in one (two) launches; ends[j].flag == -1 where the
// job is too large for the kernel's LDS budget (the host aligns those)
// size class of an alignment job on the GPU: 0 / 1 one DPP row resp. one wavefront with short tails, 2 / 3 / 4 the long classes
// (a workgroup of 4 / 8 / 16 wavefronts per alignment), -1 too large for the kernels (the host aligns it)
static int ssw_class(const RibbitAlignJob &jb) {
    if (jb.query_length <= rb::SSW_SMALL_Q && jb.ppr_length <= rb::SSW_SMALL_R) return 0;
    if (jb.query_length <= rb::SSW_BIG_Q && jb.ppr_length <= rb::SSW_BIG_R) return 1;
    if (jb.query_length <= rb::SSW_HUGE_Q && jb.ppr_length <= rb::SSW_HUGE_R) return 2;
    if (jb.query_length <= rb::SSW_GIANT_Q && jb.ppr_length <= rb::SSW_GIANT_R) return 3;
    if (jb.query_length <= rb::SSW_COLOSSAL_Q && jb.ppr_length <= rb::SSW_COLOSSAL_R) return 4;
    return -1;
}

// classes: bit c set = jobs of size class c run here (the others keep flag -1).  pool_resident: the motif pool is on the
// device already (an earlier call of the same record uploaded it).
static int run_ssw_passes(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const char *pool, size_t pool_len, int mask_len,
                          std::vector<rb::SswEnds> &ends, unsigned classes = 0x1fu, bool pool_resident = false) {
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
    if ((rc = h->d_ssw_jobs.ensure(n * 9))) return rc;
    if ((rc = h->d_ssw_out.ensure(n * 8))) return rc;
    if ((rc = h->d_ssw_order.ensure(order.size()))) return rc;
    if ((rc = h->d_ssw_pool.ensure(std::max<size_t>(pool_len, 1)))) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_ssw_jobs.p, jobs, n * sizeof(RibbitAlignJob), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_ssw_order.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    if (pool_len && !pool_resident) HIP_TRY(hipMemcpyAsync(h->d_ssw_pool.p, pool, pool_len, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->d_ssw_out.p, 0xff, n * 8 * sizeof(int32_t), h->stream));      // flag -1 unless a kernel writes the record
    // The two long classes run a workgroup per alignment (ssw_group.hip: 8 wavefronts for the giant class, 4 for the huge one)
    // unless RIBBIT_SSW_GROUP=0 asks for the older one-wavefront-per-alignment kernel (kept for comparison: same results).
    static const bool group_kernels = !(std::getenv("RIBBIT_SSW_GROUP") && std::atoi(std::getenv("RIBBIT_SSW_GROUP")) == 0);
    const size_t n_apart = n_colossal + n_giant;      // on the copy stream beside the others
    if (n_apart) {
        HIP_TRY(hipEventRecord(h->ev_ssw, h->stream));
        HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->ev_ssw, 0));
        // queries of 4097..8192 bases: only as a workgroup (16 wavefronts, 124 KB of LDS); with the group kernels off they are
        // the host's, as until the end of round 3
        if (n_colossal && group_kernels && rb::ssw_group_fits(rb::SSW_COLOSSAL_Q, 16))
            HIP_TRY(rb::launch_ssw_passes_group(h->dev_ascii_src, h->length, h->d_ssw_pool.p, h->d_ssw_jobs.p, h->d_ssw_order.p, (int)n_colossal, mask_len,
                                                rb::SSW_COLOSSAL_Q, rb::SSW_COLOSSAL_R, 16, h->d_ssw_out.p, h->copy_stream));
        if (n_giant) {
            if (group_kernels && rb::ssw_group_fits(rb::SSW_GIANT_Q, 8))
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
                          rest + n_huge, (int)n_big, rest, (int)n_huge, mask_len, h->d_ssw_out.p, h->stream, group_kernels ? 4 : 0);
    HIP_TRY(hipGetLastError());
    if (n_apart) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_ssw, 0));
    HIP_TRY(hipMemcpyAsync(ends.data(), h->d_ssw_out.p, n * sizeof(rb::SswEnds), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RIBBIT_OK;
}

// The banded path search of every job whose striped passes the GPU has done (run_ssw_passes left jobs, motif pool and end
// points on the device): rounds of one launch each, the band doubling for the alignments still open (ssw.c:603-728).
static int run_ssw_paths(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const std::vector<rb::SswEnds> &ends, std::vector<rb::SswPath> &paths) {
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
    const uint64_t path_cap = std::min<uint64_t>(worst_ops, 0xfffffff0u);
    if ((rc = h->d_path_ops.ensure((size_t)path_cap))) return rc;
    if ((rc = h->d_path_count.ensure(4))) return rc;
    if ((rc = h->d_path_result.ensure(4 * n))) return rc;
    HIP_TRY(hipMemsetAsync(h->d_path_count.p, 0, 4 * sizeof(uint32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_path_result.p, 0xff, 4 * n * sizeof(int32_t), h->stream));      // state -1: no path searched (the buffer is reused)
    std::vector<int32_t> items, result(4 * n, -1);
    std::vector<uint64_t> cell_off, ops_off;
    std::vector<Open> next;
    constexpr uint64_t ARENA = (uint64_t)6 << 30;            // cell bytes per launch
    // Alignments with a narrow band (nineteen in twenty) run four to a wavefront (ssw_path4_kernel): they come first among a
    // round's items, in seed order, the others behind them, in seed order too; RIBBIT_PATH_PACKED=0: all on the one-per-wavefront
    // kernel, as until the end of round 3.
    static const bool packed_paths = !(std::getenv("RIBBIT_PATH_PACKED") && std::atoi(std::getenv("RIBBIT_PATH_PACKED")) == 0);
    while (!open.empty()) {
        // one launch per arena-full of items
        size_t n_narrow_all = 0;
        if (packed_paths) {
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
            const size_t ni = at - first;
            const size_t n_narrow = first < n_narrow_all ? std::min(ni, n_narrow_all - first) : 0;
            if ((rc = h->d_path_items.ensure(items.size())) || (rc = h->d_path_cell_off.ensure(ni)) || (rc = h->d_path_ops_off.ensure(ni)) ||
                (rc = h->d_path_cells.ensure((size_t)std::max<uint64_t>(cells, 16))) || (rc = h->d_path_scratch.ensure((size_t)ops)))
                return rc;
            HIP_TRY(hipMemcpyAsync(h->d_path_items.p, items.data(), items.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->d_path_cell_off.p, cell_off.data(), ni * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(h->d_path_ops_off.p, ops_off.data(), ni * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
            rb::launch_ssw_paths(h->dev_ascii_src, h->length, h->d_ssw_pool.p, h->d_ssw_jobs.p, h->d_ssw_out.p, h->d_path_items.p, h->d_path_cell_off.p,
                                 h->d_path_ops_off.p, (int)ni, max_band, h->d_path_cells.p, h->d_path_scratch.p, h->d_path_ops.p, (uint32_t)path_cap,
                                 h->d_path_count.p, h->d_path_result.p, h->stream, (int)n_narrow);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(h->stream));       // the item arrays above are reused by the next launch
        }
        HIP_TRY(hipMemcpyAsync(result.data(), h->d_path_result.p, 4 * n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
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
    }
    uint32_t used = 0;
    HIP_TRY(hipMemcpyAsync(&used, h->d_path_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (used > path_cap) return fail(RIBBIT_E_INTERNAL, "path operations overflowed their arena");
    if ((rc = h->h_path_ops.ensure(std::max<size_t>(used, 1)))) return rc;
    if (used) {
        HIP_TRY(hipMemcpyAsync(h->h_path_ops.p, h->d_path_ops.p, (size_t)used * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    for (size_t j = 0; j < n; ++j)
        if (result[4 * j] == 0 && !paths[j].failed) paths[j].ops = h->h_path_ops.p + (uint32_t)result[4 * j + 2];
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

// ---- alignment batches across the records in flight (include/ribbit_hip.h) ----------------------------------------
struct AlignSubmission {
    const RibbitAlignJob *jobs = nullptr; size_t n = 0;
    const char *pool = nullptr; size_t pool_len = 0;
    const uint8_t *dev_bases = nullptr; int64_t length = 0;        // the record's bases on the batcher's device
    std::vector<rb::SswEnds> ends; std::vector<rb::SswPath> paths; std::vector<uint32_t> ops;      // results
    unsigned classes = 0x3u;              // size classes of the submission's jobs that are to run (run_ssw_passes)
    int rc = RIBBIT_OK; std::string error;
    bool done = false;
};

struct RibbitAlignBatcher {
    RibbitHandle *bh = nullptr;           // streams and buffers of the batches
    int device = 0;
    int clients = 1;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv_submit, cv_done;
    std::deque<AlignSubmission *> queue;
    bool closing = false;
    DevBuf<uint8_t> stage;
    std::atomic<int64_t> n_batches{0}, n_subs{0}, n_jobs{0}, n_gpu{0};

    void run_batch(std::vector<AlignSubmission *> &subs) {
        constexpr int64_t PAD = 256;          // between records: a kernel never reads one record's bases for another's query
        std::vector<RibbitAlignJob> jobs;
        std::string pool;
        std::vector<int64_t> base_off(subs.size());
        std::vector<size_t> first(subs.size() + 1, 0);
        int64_t total = PAD;
        for (size_t k = 0; k < subs.size(); ++k) { base_off[k] = total; total += subs[k]->length + PAD; first[k + 1] = first[k] + subs[k]->n; }
        int rc = RIBBIT_OK;
        std::string error;
        std::vector<rb::SswEnds> ends;
        std::vector<rb::SswPath> paths;
        try {
            if (total > INT32_MAX) rc = fail(RIBBIT_E_ARG, "alignment batch of %lld staged bases", (long long)total);
            jobs.reserve(first.back());
            for (size_t k = 0; k < subs.size() && !rc; ++k) {
                const AlignSubmission &sb = *subs[k];
                const int32_t pool_base = (int32_t)pool.size();
                pool.append(sb.pool, sb.pool_len);
                for (size_t j = 0; j < sb.n; ++j) {
                    RibbitAlignJob jb = sb.jobs[j];
                    // the kernels' own clipping (a negative start clamps, the end clamps to the record), done here against
                    // the job's OWN record; then into the staging buffer's coordinates
                    if (jb.query_start < 0) { jb.query_length += jb.query_start; jb.query_start = 0; }
                    if ((int64_t)jb.query_start + jb.query_length > sb.length) jb.query_length = (int32_t)(sb.length - jb.query_start);
                    if (jb.query_length < 0) jb.query_length = 0;
                    jb.query_start += (int32_t)base_off[k];
                    jb.motif_offset += pool_base;
                    jobs.push_back(jb);
                }
            }
            if (!rc) rc = bind_device(bh);
            if (!rc) rc = stage.ensure((size_t)total);
            if (!rc) {
                hipError_t e = hipMemsetAsync(stage.p, 'N', (size_t)total, bh->stream);
                for (size_t k = 0; k < subs.size() && e == hipSuccess; ++k)
                    if (subs[k]->length)
                        e = hipMemcpyAsync(stage.p + base_off[k], subs[k]->dev_bases, (size_t)subs[k]->length, hipMemcpyDeviceToDevice, bh->stream);
                if (e != hipSuccess) rc = fail(RIBBIT_E_DEVICE, "staging the records' bases failed: %s", hipGetErrorString(e));
            }
            if (!rc) {
                bh->dev_ascii_src = stage.p; bh->length = total; bh->loaded = true;
                unsigned classes = 0;       // (a job of a class its own submission did not ask for is aligned all the same: same result)
                for (const AlignSubmission *sb : subs) classes |= sb->classes;
                rc = run_ssw_passes(bh, jobs.data(), jobs.size(), pool.data(), pool.size(), 15, ends, classes);
            }
            // RIBBIT_BATCH_PATHS=0: striped passes only; the banded path search of these short alignments stays on the records'
            // host threads (a batch then is one launch and one synchronisation instead of a round per band width)
            static const bool gpu_paths = !(std::getenv("RIBBIT_BATCH_PATHS") && std::atoi(std::getenv("RIBBIT_BATCH_PATHS")) == 0);
            if (!rc && gpu_paths) rc = run_ssw_paths(bh, jobs.data(), jobs.size(), ends, paths);
            else if (!rc) paths.assign(jobs.size(), rb::SswPath{});
            if (rc) error = g_last_error;
        } catch (const std::bad_alloc &) { rc = RIBBIT_E_NOMEM; error = "out of host memory in a shared alignment batch"; }
        int64_t on_gpu = 0;
        for (size_t k = 0; k < subs.size(); ++k) {
            AlignSubmission &sb = *subs[k];
            sb.rc = rc; sb.error = error;
            if (!rc) {
                try {
                    sb.ends.assign(ends.begin() + (std::ptrdiff_t)first[k], ends.begin() + (std::ptrdiff_t)first[k + 1]);
                    sb.paths.assign(paths.begin() + (std::ptrdiff_t)first[k], paths.begin() + (std::ptrdiff_t)first[k + 1]);
                    size_t n_ops = 0;
                    for (const rb::SswPath &pt : sb.paths) if (pt.ops) n_ops += (size_t)pt.n_ops;
                    sb.ops.resize(n_ops);
                    size_t at = 0;
                    for (rb::SswPath &pt : sb.paths)
                        if (pt.ops) { std::memcpy(sb.ops.data() + at, pt.ops, (size_t)pt.n_ops * sizeof(uint32_t)); pt.ops = sb.ops.data() + at; at += (size_t)pt.n_ops; }
                    for (const rb::SswEnds &e : sb.ends) on_gpu += e.flag != -1;
                } catch (const std::bad_alloc &) { sb.rc = RIBBIT_E_NOMEM; sb.error = "out of host memory in a shared alignment batch"; }
            }
        }
        n_batches += 1; n_subs += (int64_t)subs.size(); n_jobs += (int64_t)first.back(); n_gpu += on_gpu;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (AlignSubmission *sb : subs) sb->done = true;
        }
        cv_done.notify_all();
    }

    void loop() {
        std::vector<AlignSubmission *> subs;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_submit.wait(lk, [&]() { return closing || !queue.empty(); });
                if (queue.empty()) return;       // closing
                // a short window for the other records in flight to get here: a batch of everybody's jobs costs little more
                // than a batch of one record's
                static const int window_us = std::getenv("RIBBIT_BATCH_WINDOW_US") ? std::max(0, std::atoi(std::getenv("RIBBIT_BATCH_WINDOW_US"))) : 400;
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
                cv_submit.wait_until(lk, deadline, [&]() { return closing || (int)queue.size() >= clients; });
                subs.assign(queue.begin(), queue.end());
                queue.clear();
            }
            run_batch(subs);
        }
    }

    // blocks until the submission's results are in place
    void submit(AlignSubmission &sb) {
        {
            std::lock_guard<std::mutex> lk(mu);
            queue.push_back(&sb);
        }
        cv_submit.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&]() { return sb.done; });
    }
};

int ribbit_hip_batcher_open(const RibbitScanParams *params, int device, int32_t expected_clients, RibbitAlignBatcher **out) {
    if (!params || !out) return fail(RIBBIT_E_ARG, "null argument");
    RibbitAlignBatcher *b = new (std::nothrow) RibbitAlignBatcher();
    if (!b) return fail(RIBBIT_E_NOMEM, "out of host memory");
    const int rc = ribbit_hip_open(params, device, &b->bh);
    if (rc) { delete b; return rc; }
    b->device = device;
    b->clients = std::max(1, (int)expected_clients);
    b->worker = std::thread([b]() { b->loop(); });
    *out = b;
    return RIBBIT_OK;
}

int ribbit_hip_batcher_close(RibbitAlignBatcher *b) {
    if (!b) return RIBBIT_OK;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->closing = true;
    }
    b->cv_submit.notify_all();
    if (b->worker.joinable()) b->worker.join();
    (void)hipSetDevice(b->device);
    b->stage.release();
    b->bh->dev_ascii_src = nullptr; b->bh->loaded = false;
    (void)ribbit_hip_close(b->bh);
    delete b;
    return RIBBIT_OK;
}

int ribbit_hip_set_batcher(RibbitHandle *h, RibbitAlignBatcher *b) {
    if (!h) return fail(RIBBIT_E_ARG, "null argument");
    if (b && b->device != h->device) return fail(RIBBIT_E_ARG, "the batcher runs on device %d, the handle on device %d", b->device, h->device);
    h->batcher = b;
    return RIBBIT_OK;
}

void ribbit_hip_batcher_stats(const RibbitAlignBatcher *b, int64_t out[4]) {
    if (!b || !out) return;
    out[0] = b->n_batches.load(); out[1] = b->n_subs.load(); out[2] = b->n_jobs.load(); out[3] = b->n_gpu.load();
}

static int refine_bed_impl(RibbitHandle *h, const RibbitRefineParams *prm, const char *sequence_id, const char **text, size_t *len);

int ribbit_hip_refine_bed(RibbitHandle *h, const RibbitRefineParams *prm, const char *sequence_id,
                          const char **text, size_t *len) {
    try {
        return refine_bed_impl(h, prm, sequence_id, text, len);
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
// A level's batch goes where it is large enough to pay: on this handle's own streams from LEVEL_OWN_BATCH nodes on (a
// chromosome's first levels: 46 K, 28 K, 15 K ... alignments at -M 500); into the GPU's SHARED batches if the handle has a
// batcher (a read among many in flight: its levels hold 5-30 alignments, and a batch of one record's costs as much as a batch
// of everybody's); else the level -- and with it everything below it -- is finished on the host threads by plain recursion.
// Measured (chromosome-1-sized record at -M 500): the trees are deep chains (128 levels, one flank trimmed at a time), and a
// level of a hundred alignments costs 40 ms of latency on the GPU -- a long alignment holds its workgroup that long however few
// there are -- where the host threads need 20: levels own-batched all the way down took 6.6 s, 3.5 of them below level 8.
constexpr size_t LEVEL_OWN_BATCH = 400;
static int refine_levels(RibbitHandle *h, const RibbitRefineParams &prm, const std::string &sequence_id, std::vector<rb::DeferredNode> &nodes,
                         std::vector<rb::BedPiece> &pieces, unsigned threads, bool *order_dependent, int64_t counts[3], bool may_share) {
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
        const bool shared = may_share && h->batcher != nullptr && n < 8 * own_batch;
        bool od = false;
        if (!shared && n < own_batch) {
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
        AlignSubmission sb;
        double t2, t3;
        if (shared) {
            sb.jobs = jobs.data(); sb.n = jobs.size(); sb.pool = pool.data(); sb.pool_len = pool.size();
            sb.dev_bases = h->dev_ascii_src; sb.length = h->length; sb.classes = 0x1fu;
            if (!jobs.empty()) h->batcher->submit(sb);
            if (sb.rc) { g_last_error = sb.error; return sb.rc; }
            if (jobs.empty()) { sb.ends.clear(); sb.paths.clear(); }
            ends.swap(sb.ends); paths.swap(sb.paths);       // (the paths' operations live in sb.ops)
            t2 = t3 = now_ms();
        } else {
            if ((rc = run_ssw_passes(h, jobs.data(), jobs.size(), pool.data(), pool.size(), 15, ends, 0x1fu))) return rc;
            t2 = now_ms();
            if ((rc = run_ssw_paths(h, jobs.data(), jobs.size(), ends, paths))) return rc;
            t3 = now_ms();
        }
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
            std::fprintf(stderr, "[refine levels] level %d: %zu nodes put off, %zu alignments%s: consensus rows %.1f ms, set-up + striped passes %.1f ms, path search %.1f ms, "
                                 "host finish %.1f ms; %zu nodes put off for the next level\n", level, n, jobs.size(), shared ? " (shared batch)" : "", t1 - t0, t2 - t1, t3 - t2, t4 - t3, next.size());
        nodes.swap(next);
    }
    if (profile)
        std::fprintf(stderr, "[refine levels] %zu nodes in %d GPU levels: consensus rows %.1f ms, set-up + striped passes (or shared batches) %.1f ms, path search %.1f ms, host finish %.1f ms; "
                             "%zu nodes of the last level finished on the host threads by recursion in %.1f ms\n", n_nodes, level - 1, t_rows, t_passes, t_paths, t_finish, n_host, t_host);
    return RIBBIT_OK;
}

// the pieces' text into h->bed, in printing order: by seed, and inside a seed's recursion tree by place (refine.h)
static void join_pieces(RibbitHandle *h, std::vector<rb::BedPiece> &pieces, unsigned threads) {
    std::sort(pieces.begin(), pieces.end(), [](const rb::BedPiece &x, const rb::BedPiece &y) {
        return x.first_seed != y.first_seed ? x.first_seed < y.first_seed : x.path < y.path; });
    std::vector<size_t> at(pieces.size() + 1, 0);
    for (size_t k = 0; k < pieces.size(); ++k) at[k + 1] = at[k] + pieces[k].text.size();
    h->bed.resize(at[pieces.size()]);
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, at[pieces.size()] / (4u << 20) + 1));
    std::atomic<size_t> next_piece{0};
    auto place = [&]() {
        for (size_t k; (k = next_piece.fetch_add(64)) < pieces.size();)
            for (size_t q = k; q < std::min(pieces.size(), k + 64); ++q)
                if (!pieces[q].text.empty()) std::memcpy(&h->bed[at[q]], pieces[q].text.data(), pieces[q].text.size());
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
        static const char *const large_env = std::getenv("RIBBIT_SSW_LARGE");
        const bool large_class = large_env ? std::atoi(large_env) != 0 : true;        // 0: the long jobs stay on the host threads (a measurement knob)
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
            mark = std::max<uint8_t>(mark, (cls < 0 || !large_class) ? 2 : 1);
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
        // next slice's passes start (RIBBIT_SSW_FEEDERS=1: one feeder, as until the end of round 3).
        static const char *const feeders_env = std::getenv("RIBBIT_SSW_FEEDERS");
        size_t n_feeders = feeders_env ? (size_t)std::max(1, std::min(2, std::atoi(feeders_env))) : 2;
        if (n_slices < 2) n_feeders = 1;
        if (n_feeders == 2) {
            if (!h->aux2 && ribbit_hip_open(&h->params, h->device, &h->aux2) != RIBBIT_OK) n_feeders = 1;      // (no memory for it: one feeder)
            else { h->aux2->dev_ascii_src = h->dev_ascii_src; h->aux2->length = h->length; h->aux2->loaded = true; }
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
        std::thread feeder, feeder2;
        struct FeederGuard {
            std::atomic<bool> &stop; std::condition_variable &cv; std::thread &a, &b;
            ~FeederGuard() { stop = true; cv.notify_all(); if (a.joinable()) a.join(); if (b.joinable()) b.join(); }
        } feeder_guard{stop, cv, feeder, feeder2};
        feeder = std::thread([&]() { feeder_loop(0, h); });
        if (n_feeders == 2) feeder2 = std::thread([&]() { feeder_loop(1, h->aux2); });
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
                              sl.job_first.data(), set_aside.data(), &pieces, nullptr, sl.lo, treep);
            t_work += now_ms() - tk;
            if (order_dependent) break;
        }
        stop = true;
        cv.notify_all();
        feeder.join();
        if (feeder2.joinable()) feeder2.join();
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
                                  &pieces, &by_cost, 0, treep);
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
            rc = refine_levels(h, *prm, sequence_id, put_off, pieces, threads, &order_dependent, counts, false);
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
        if (profile) std::fprintf(stderr, "[refine_bed] %zu alignment jobs (%zu long ones in their own batch: %.1f ms, set up first in %.1f ms; %zu seeds set aside), set-up in all %.1f ms; %zu slices: "
                                  "feeder %.1f ms in all (GPU striped passes incl. transfers %.1f ms, GPU path search %.1f ms); workers: %.1f ms in their calls, waited %.1f ms "
                                  "for slices, %.1f ms for the long batch; seeds set aside for it %.1f ms; %zu seeds with jobs beyond the kernels' reach refined on the host beside all that in %.1f ms; "
                                  "nodes put off, level by level %.1f ms; rows put together %.1f ms; since the call began %.1f ms\n",
                                  n_jobs, long_jobs.size(), t_long, t_setup_long, later.size(), t_setup, n_slices, t_feed, t_passes, t_paths, t_work, t_wait, t_wait_long, t_later, giants.size(), t_later_thread, t_levels, t_join,
                                  now_ms() - t_begin);
        add_ms(t_jobs_us, t_wait + t_wait_long);
    }
    static const char *const shared_env = std::getenv("RIBBIT_SHARED_SSW");
    static const size_t shared_max_seeds = std::getenv("RIBBIT_SHARED_MAX_SEEDS") ? (size_t)std::atoll(std::getenv("RIBBIT_SHARED_MAX_SEEDS")) : 100000;
    if (!done && !gpu_ssw && h->batcher && !h->dispatch.empty() && h->dispatch.size() <= shared_max_seeds && h->dev_ascii_src &&
        shared_env && std::atoi(shared_env) != 0) {      // (the batcher is there for the levels below; first-level jobs join it only on request)
        // a short record among several in flight: its alignment jobs join the shared batch of this GPU's batcher
        std::vector<RibbitAlignJob> jobs;
        std::string pool;
        rb::build_align_jobs(h->host, *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(), jobs, pool, threads, 0, h->dispatch.size(), &small);
        AlignSubmission sb;
        sb.jobs = jobs.data(); sb.n = jobs.size(); sb.pool = pool.data(); sb.pool_len = pool.size();
        sb.dev_bases = h->dev_ascii_src; sb.length = h->length;
        if (!jobs.empty()) h->batcher->submit(sb);
        if (sb.rc == RIBBIT_OK) {
            bool order_dependent = false;
            if (jobs.empty()) { sb.ends.clear(); sb.paths.clear(); }
            rb::refine_to_bed(h->host, h->host_bases ? h->host_bases : h->host_ascii.data(), *prm, h->dispatch, h->longest_runs.data(), h->best_rows.data(),
                              sequence_id, h->bed, h->host_threads, &jobs, &sb.ends, &sb.paths, 0, (size_t)-1, &order_dependent, &small);
            done = !order_dependent;
            if (!done) h->bed.clear();
        } else if (sb.rc != RIBBIT_E_NOMEM) {
            g_last_error = sb.error;
            return sb.rc;
        }
    }
    const char *defer_reads_env = std::getenv("RIBBIT_DEFER_READS");
    if (!done && !gpu_ssw && defer_reads_env && std::atoi(defer_reads_env) != 0 && defer_min_length() > 0 && h->dev_ascii_src && !h->dispatch.empty()) {
        // A short record (a read among many in flight): its seeds are refined on the host threads; with RIBBIT_DEFER_READS=1 the
        // expensive nodes of its long-motif seeds -- first level or flanks -- are put off and done in GPU batches, level by level
        // (the GPU's shared batches when the handle has a batcher).  OFF by default, by measurement (100 Mbp of 10-100 kb reads
        // at -M 500, tools/m500_probe.py): 22.8 s on the host threads alone, 104 s with 8 reads in flight sharing batches, 53 s
        // with 32 -- a read's tree is 3-7 levels deep with 5-30 nodes a level, every level waits for a batch, and a batch lasts
        // as long as its longest alignment (tens of milliseconds on one workgroup where a host thread needs ten).  What did pay
        // for the reads is the consensus rows with AVX-512 (refine.cpp).
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
            rc = refine_levels(h, *prm, sequence_id, put_off, pieces, threads, &order_dependent, counts, true);
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
    *text = h->bed.c_str();
    *len = h->bed.size();
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
    rb::refine_to_bed(hp, sequence, *prm, seeds, longest.data(), nullptr, sequence_id, bed);
    *len = bed.size();
    *text = (char *)std::malloc(bed.size() + 1);
    if (!*text) return fail(RIBBIT_E_NOMEM, "out of host memory");
    std::memcpy(*text, bed.c_str(), bed.size() + 1);
    return RIBBIT_OK;
}

void ribbit_text_free(char *text) { std::free(text); }

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

int64_t ribbit_hip_guard_hits(const RibbitHandle *h) { return h ? h->lists.guard_hits : 0; }

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

int ribbit_hip_debug_set_event_capacity(RibbitHandle *h, size_t events) {
    if (!h) return fail(RIBBIT_E_ARG, "null argument");
    h->debug_first_cap = events;
    return RIBBIT_OK;
}

void ribbit_debug_set_merge_min_range(size_t calls) { rb::set_merge_min_range(calls); }

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

void ribbit_debug_level_counters(int64_t out[3]) {
    for (int k = 0; k < 3; ++k) out[k] = g_level_counts[k].load();
}

void ribbit_debug_small_motif_counters(int64_t out[2]) {
    long a = 0, b = 0;
    rb::small_motif_counters(a, b);
    out[0] = a; out[1] = b;
}

int32_t ribbit_debug_last_dispatch_ranges(void) { return (int32_t)rb::last_dispatch_ranges(); }

void ribbit_debug_last_merge(int stage, int32_t out[5]) {
    const rb::MergeStats st = rb::last_merge_stats(stage);
    out[0] = (int32_t)st.ranges; out[1] = (int32_t)std::min(st.ranges_redone, 0xffffu) | (int32_t)(std::min(st.stale_by_sight, 0x7fffu) << 16); out[2] = (st.redone_in_order ? 1 : 0) | (int32_t)(st.ranges_run << 1);
    out[3] = (int32_t)std::min<long long>(st.head_writes, INT32_MAX); out[4] = (st.first_range_empty ? 1 : 0) | (int32_t)(st.passes << 8);
}

int ribbit_hip_set_timing(RibbitHandle *h, int32_t enabled) {
    if (!h) return fail(RIBBIT_E_ARG, "null argument");
    h->timing = enabled != 0;
    if (!h->timing) h->have_timing[0] = h->have_timing[1] = h->have_timing[2] = false;
    return RIBBIT_OK;
}

int ribbit_hip_set_host_threads(RibbitHandle *h, int32_t threads) {
    if (!h || threads < 0) return fail(RIBBIT_E_ARG, "bad argument");
    h->host_threads = (unsigned)threads;
    return RIBBIT_OK;
}

int ribbit_hip_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return fail(RIBBIT_E_ARG, "null argument");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return RIBBIT_OK;
}

int ribbit_hip_host_unregister(void *p) {
    if (!p) return fail(RIBBIT_E_ARG, "null argument");
    HIP_TRY(hipHostUnregister(p));
    return RIBBIT_OK;
}

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
        rb::merge_anchored_stage_full(sl, anchored_calls, n_anchored_calls, rb::merge_threads(0), &st);
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

static int query_plane(RibbitHandle *h, int32_t shift, int64_t start, int64_t end, bool want_words, uint32_t *count_out) {
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (shift < h->min_shift || shift > h->max_shift) return fail(RIBBIT_E_ARG, "shift %d outside [%d,%d]", shift, h->min_shift, h->max_shift);
    if (start < 0 || end > h->length || start > end) return fail(RIBBIT_E_ARG, "range [%lld,%lld) outside the record", (long long)start, (long long)end);
    int rc;
    if ((rc = bind_device(h))) return rc;
    const int64_t w0 = start / 32, w1 = (end + 31) / 32;
    const int64_t nw = w1 - w0;
    if (count_out) *count_out = 0;
    if (nw <= 0) return RIBBIT_OK;
    if ((rc = h->d_query.ensure((size_t)nw + 16))) return rc;
    if ((rc = h->h_query.ensure((size_t)nw + 16))) return rc;
    uint32_t *d_count = h->d_query.p + nw;
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), h->stream));
    rb::launch_plane_words(h->planes(), shift, w0, nw, want_words ? h->d_query.p : nullptr, start, end, d_count, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->h_query.p, h->d_query.p, ((size_t)nw + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (count_out) *count_out = h->h_query.p[nw];
    return RIBBIT_OK;
}

int ribbit_hip_plane_bits(RibbitHandle *h, int32_t shift, int64_t start, int64_t end, uint8_t *out) {
    if (!h || (!out && end > start)) return fail(RIBBIT_E_ARG, "null argument");
    if (h->loaded && h->xa_on_device && shift >= h->params.min_motif && shift <= h->params.max_motif) {
        // composed plane (fasta_utils.cpp:159): written by the anchored kernel, resident in HBM
        if (start < 0 || end > h->length || start > end) return fail(RIBBIT_E_ARG, "range [%lld,%lld) outside the record", (long long)start, (long long)end);
        if (end == start) return RIBBIT_OK;
        int rcx;
        if ((rcx = bind_device(h))) return rcx;
        const int64_t w0 = start / 32, nw = (end + 31) / 32 - w0;
        if ((rcx = h->h_query.ensure((size_t)nw + 16))) return rcx;
        HIP_TRY(hipMemcpyAsync(h->h_query.p, h->d_xa.p + (int64_t)(shift - h->params.min_motif) * h->xa_stride + w0, (size_t)nw * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int64_t p = start; p < end; ++p) out[p - start] = (h->h_query.p[p / 32 - w0] >> (p & 31)) & 1u;
        return RIBBIT_OK;
    }
    int rc = query_plane(h, shift, start, end, true, nullptr);
    if (rc) return rc;
    const int64_t w0 = start / 32;
    for (int64_t p = start; p < end; ++p) out[p - start] = (h->h_query.p[p / 32 - w0] >> (p & 31)) & 1u;
    return RIBBIT_OK;
}

int ribbit_hip_range_popcount(RibbitHandle *h, int32_t shift, int64_t start, int64_t end, int32_t *count) {
    if (!h || !count) return fail(RIBBIT_E_ARG, "null argument");
    uint32_t c = 0;
    int rc = query_plane(h, shift, start, end, false, &c);
    if (rc) return rc;
    *count = (int32_t)c;
    return RIBBIT_OK;
}

int64_t ribbit_hip_plane_words(const RibbitHandle *h) { return h && h->loaded ? h->length / 32 + 1 : 0; }

int ribbit_hip_packed_plane(RibbitHandle *h, int which, uint32_t *out_words) {
    if (!h || !out_words) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    if (which < 0 || which > 2) return fail(RIBBIT_E_ARG, "which must be 0, 1 or 2");
    int rc;
    if ((rc = bind_device(h))) return rc;
    const uint32_t *src = (which == 0 ? h->d_hi.p : which == 1 ? h->d_lo.p : h->d_brk.p) + rb::LEAD_WORDS;
    const size_t n = (size_t)(h->length / 32 + 1);
    HIP_TRY(hipMemcpyAsync(out_words, src, n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RIBBIT_OK;
}

int ribbit_hip_last_timing_ms(const RibbitHandle *h, int what, double *ms) {
    if (!h || !ms) return fail(RIBBIT_E_ARG, "null argument");
    if (what == 3) { *ms = h->host_ms; return RIBBIT_OK; }
    if (what == 4) { *ms = h->merge_ms; return RIBBIT_OK; }
    if (what == 5) { *ms = h->subst_merge_ms; return RIBBIT_OK; }
    if (what == 6 || what == 7) {
        if (!h->have_stage_timing[what - 6]) return fail(RIBBIT_E_STATE, "that stage's kernel has not run on this handle");
        float f = 0.f;
        HIP_TRY(hipEventSynchronize(h->ev_stage[what - 6][1]));
        HIP_TRY(hipEventElapsedTime(&f, h->ev_stage[what - 6][0], h->ev_stage[what - 6][1]));
        *ms = f;
        return RIBBIT_OK;
    }
    if (what == 8 || what == 9) {      // the anchored stage's two kernels: 8 planes (anchors + composition), 9 window scan of the planes
        if (!h->have_stage_timing[1] || !h->planes_timing_valid) return fail(RIBBIT_E_STATE, "the anchored stage has not run as two kernels on this handle");
        float f = 0.f;
        HIP_TRY(hipEventSynchronize(h->ev_stage[1][1]));
        if (what == 8) HIP_TRY(hipEventElapsedTime(&f, h->ev_stage[1][0], h->ev_planes));
        else HIP_TRY(hipEventElapsedTime(&f, h->ev_planes, h->ev_stage[1][1]));
        *ms = f;
        return RIBBIT_OK;
    }
    if (what < 0 || what > 9) return fail(RIBBIT_E_ARG, "what must be 0..9");
    if (!h->have_timing[what]) return fail(RIBBIT_E_STATE, "no timing recorded yet");
    float f = 0.f;
    HIP_TRY(hipEventSynchronize(h->ev[2 * what + 1]));
    HIP_TRY(hipEventElapsedTime(&f, h->ev[2 * what], h->ev[2 * what + 1]));
    *ms = f;
    return RIBBIT_OK;
}

int ribbit_hip_debug_stream_read(RibbitHandle *h, int64_t nbytes, int64_t *bytes_read) {
    if (!h || !bytes_read) return fail(RIBBIT_E_ARG, "null argument");
    if (!h->loaded) return fail(RIBBIT_E_STATE, "no record loaded");
    int rc;
    if ((rc = bind_device(h))) return rc;
    if ((rc = h->d_counters.ensure(rb::EV_COUNTER_WORDS))) return rc;
    // the event buffer is the largest resident allocation; fall back to the hi plane
    const uint32_t *src = h->d_events.p ? (const uint32_t *)h->d_events.p : h->d_hi.p;
    const int64_t avail = h->d_events.p ? (int64_t)h->d_events.cap * 8 : h->total_words * 4;
    const int64_t n = std::max<int64_t>(0, std::min(nbytes, avail)) / 4;
    rb::launch_calib_stream_read(src, n, h->d_counters.p + rb::EV_SUMMARY + 8, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    *bytes_read = n * 4;
    return RIBBIT_OK;
}

int64_t ribbit_hip_last_event_count(const RibbitHandle *h) { return h ? h->last_event_count : 0; }

}  // extern "C"
