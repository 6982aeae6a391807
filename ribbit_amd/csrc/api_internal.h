// api_internal.h -- shared by the api_*.cpp files behind the extern "C" boundary of libribbit_hip.so (include/ribbit_hip.h):
// the handle, its device / page-locked buffers, error reporting, and the internal steps that more than one of those files
// takes.  Until round 4 all of this was one 3,200-line api.cpp; it is now
//   api_core.cpp        handles, streams, loading a record (pack), timers, plane queries
//   api_perfect.cpp     the perfect stage: scan, device-side pairing, runs, calls, seeds; its chunk form
//   api_window.cpp      the substitution and anchored stages: scans, streak pairing, window state machines, merges, dispatch order
//   api_chunks.cpp      one chunk of a longer record (window stages) and the merging rank's half
//   api_merge.cpp       the anchored stage's merge on the device: what parallel_merge.h's AnchoredDevicePass does on a handle
//   api_align.cpp       the scans of the dispatched seeds, alignment jobs, batched striped passes and path searches
//   api_refine_bed.cpp  refinement to BED text: the GPU alignment pipeline, the recursion's levels, the host-only form
// Not part of the ABI; nothing outside ribbit_amd/csrc includes it.
#pragma once
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


namespace rbapi {

extern thread_local std::string g_last_error;
int fail(int code, const char *fmt, ...);
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RIBBIT_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// A device buffer that is being replaced by a larger one is released at the next record load or handle close, not on the spot:
// hipFree synchronises with every stream of the device (api_core.cpp).
void device_free_later(void *p);
void pinned_free_later(void *p);
void device_free_pending();      // (both kinds)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    // grows: a buffer that is sized again and again within one call (the alignment batches', slice by slice) asks for half as much
    // again as it has, at least, and a quarter of headroom: it is then released and allocated a few times per record instead of at
    // every slice that is a little larger than the one before.  Everything else is sized exactly: callers derive region sizes
    // from `cap` (a quarter more event capacity is a third off the perfect scan's throughput).
    int ensure(size_t n, bool grows = false) {
        if (n <= cap) return RIBBIT_OK;
        if (grows) n = std::max(n + std::min(n / 4, ((size_t)256 << 20) / sizeof(T)), cap + std::min(cap / 2, ((size_t)1 << 30) / sizeof(T)));
        if (p) { device_free_later(p); p = nullptr; cap = 0; }      // (hipFree waits for the whole device: not while another feeder's kernels run)
        // RIBBIT_PROFILE_MEMORY=<MB>: one line per device allocation of at least that size (which buffers are large, and when)
        static const size_t trace_from = std::getenv("RIBBIT_PROFILE_MEMORY") ? (size_t)std::max(1, std::atoi(std::getenv("RIBBIT_PROFILE_MEMORY"))) << 20 : 0;
        if (trace_from && n * sizeof(T) >= trace_from)
            std::fprintf(stderr, "[device memory] %.2f GB (%zu elements of %zu bytes), called from %p\n", (double)(n * sizeof(T)) * 1e-9, n, sizeof(T), __builtin_return_address(0));
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e != hipSuccess) { device_free_pending(); e = hipMalloc((void **)&p, n * sizeof(T)); }      // (what was put aside may be the room that is missing)
        if (e != hipSuccess) { p = nullptr; return fail(RIBBIT_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e)); }
        cap = n;
        return RIBBIT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Page-locked host memory (api_core.cpp).  From PINNED_HUGE_FROM bytes on: anonymous memory advised into 2-MB pages, touched on
// several threads, then registered with the runtime -- measured on the GPU box (tools/probes/pin_probe.hip, 8 GB): 0.05 s against
// 1.27 s for hipHostMalloc, the copies into it at the same 53 GB/s.  The composed planes' host copy is 3 GB for a chromosome at
// -M 100 and 15.5 GB at -M 500 (2.5 s of a 15-s run before).  Falls back to hipHostMalloc wherever a step fails.
int pinned_alloc(size_t bytes, void **out);      // RIBBIT_OK / RIBBIT_E_NOMEM (fail() has the message)
void pinned_free(void *p);

template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n, bool grows = false) {
        if (n <= cap) return RIBBIT_OK;
        if (grows) n = std::max(n + std::min(n / 4, ((size_t)256 << 20) / sizeof(T)), cap + std::min(cap / 2, ((size_t)1 << 30) / sizeof(T)));      // (as DevBuf)
        if (p) { pinned_free_later(p); p = nullptr; cap = 0; }
        void *q = nullptr;
        const int rc = pinned_alloc(n * sizeof(T), &q);
        if (rc) return rc;
        p = static_cast<T *>(q);
        cap = n;
        return RIBBIT_OK;
    }
    void release() { if (p) pinned_free(p); p = nullptr; cap = 0; }
};

inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

enum Stage { STAGE_NONE = 0, STAGE_PERFECT = 1, STAGE_SUBST = 2, STAGE_ANCHORED = 3 };

}  // namespace rbapi

using namespace rbapi;

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
    // the anchored stage's merge as device work (api_merge.cpp, anchored_merge.hip): the lists, the ranges and what they leave
    struct MergeBufs {
        DevBuf<RibbitSeed> d_perfect, d_subst, d_own;
        DevBuf<int32_t> d_type0, d_cuts;                // types before the stage (perfect then substitution list); cut_pos | cur0
        DevBuf<uint32_t> d_first, d_range_out, d_log, d_head_log, d_counts, d_scratch;
        PinnedBuf<RibbitSeed> h_own;
        PinnedBuf<uint32_t> h_range_out, h_log, h_head_log, h_counts, h_sync;      // h_sync: read and written by the lanes while the kernel runs
        uint32_t *h_sync_dev = nullptr;
        void release() {
            d_perfect.release(); d_subst.release(); d_own.release(); d_type0.release(); d_cuts.release(); d_first.release(); d_range_out.release();
            d_log.release(); d_head_log.release(); d_counts.release(); d_scratch.release();
            h_own.release(); h_range_out.release(); h_log.release(); h_head_log.release(); h_counts.release(); h_sync.release(); h_sync_dev = nullptr;
        }
    } mg;
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
    DevBuf<RibbitSeed> d_seeds_small;     // the consensus-row scan's jobs (the small-motif scan beside it reads d_seeds: the dispatch list as build_longest_runs left it)
    PinnedBuf<RibbitSeed> h_seed_stage;   // the dispatch list on its way up
    PinnedBuf<int32_t> h_longest_stage;   // ... and the longest runs on their way down
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
    char *bed_raw = nullptr;              // the text of the last refinement when its pieces were joined (join_pieces: storage the copying threads touch first)
    size_t bed_raw_len = 0, bed_raw_cap = 0;
    bool bed_in_raw = false;              // the last ribbit_hip_refine_bed returned bed_raw, not bed
    int stage_done = STAGE_NONE;          // how far the seed lists have been advanced
    rb::SeedLists lists;
    bool refine_met_empty_query = false;  // the last ribbit_hip_refine_bed on this handle met an alignment with an empty query (ribbit_hip_refine_met_empty_query)
    RibbitHandle *aux = nullptr;          // helper handle of ribbit_hip_refine_bed: streams and buffers of the long alignment batch
    std::vector<RibbitHandle *> feed_aux; // ... and of its further feeders (each takes every n-th slice of the short alignments)

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

namespace rbapi {


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

// ---- steps shared between the files (each is defined, and described, in the file the list above names)
int bind_device(const RibbitHandle *h);
int is_gfx950(int device);
int pack_loaded_ascii(RibbitHandle *h, const uint8_t *dev_ascii, int64_t length);
int ensure_host_planes(RibbitHandle *h);
int collect_events(RibbitHandle *h, int which);
rb::EventSource event_source(const RibbitHandle *h);
int perfect_wait(RibbitHandle *h);
int perfect_enqueue(RibbitHandle *h, size_t cap);
int perfect_begin(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset);
int perfect_collect(RibbitHandle *h);
int perfect_finish(RibbitHandle *h, RibbitRun *dst, size_t dst_cap, RibbitRun *half_dst, size_t half_dst_cap, bool wait = true);
int run_perfect_scan_range(RibbitHandle *h, int64_t own_lo, int64_t own_hi, int64_t pos_offset, RibbitRun *dst, size_t dst_cap,
                           RibbitRun *half_dst = nullptr, size_t half_dst_cap = 0);
int run_perfect_scan(RibbitHandle *h);
int build_perfect_calls(RibbitHandle *h);
int advance_to_perfect(RibbitHandle *h);
int scan_and_pair_streaks(RibbitHandle *h, int which, uint32_t *n_streaks, int (*filter_min_span)(int) = nullptr);
int window_stage_device(RibbitHandle *h, int which, bool full, int (*min_span)(int), DeviceCalls *out, ChunkWindow *cw = nullptr);
void full_calls_from_device(const DeviceCalls &dc, rb::CallVec &calls);
int build_subst_calls(RibbitHandle *h);
void subst_merge(RibbitHandle *h, const DeviceCalls *dc);
double feeder_phase_ms(int k);      // (profiling: host phases of run_ssw_passes / run_ssw_paths, cumulative)
// api_merge.cpp: the first parallel pass of the anchored stage's merge on the device (d_calls / d_pend: the stage's kept calls as
// window_stage_device left them in device memory, d_pend null when kc.pend is)
rb::AnchoredDevicePass anchored_device_pass(RibbitHandle *h, RibbitCall *d_calls, const int32_t *d_pend);
int advance_to_subst(RibbitHandle *h);
int prepare_anchored(RibbitHandle *h);
int xa_copy_begin(RibbitHandle *h);
int xa_wait_host(RibbitHandle *h);
int build_anchored_calls(RibbitHandle *h);
void print_anchored_merge_profile(size_t seeds, const rb::MergeStats &st, double dispatch_ms, unsigned dispatch_ranges);
int advance_to_anchored(RibbitHandle *h);
int build_longest_runs(RibbitHandle *h);
int best_rows_of(RibbitHandle *h, const RibbitRefineParams &prm, const rb::SeedVec &seeds, const int32_t *longest, int32_t *best);
int build_best_rows(RibbitHandle *h, const RibbitRefineParams &prm);
int build_small_motifs(RibbitHandle *h, const RibbitRefineParams &prm, hipStream_t stream = nullptr);
int scan_seeds_side_by_side(RibbitHandle *h, const RibbitRefineParams &prm);
void fill_refine_defaults(RibbitRefineParams *p, int min_motif, int max_motif);
int ssw_class(const RibbitAlignJob &jb);
int run_ssw_passes(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const char *pool, size_t pool_len, int mask_len,
                          std::vector<rb::SswEnds> &ends, unsigned classes = 0x1fu, bool pool_resident = false);
int run_ssw_paths(RibbitHandle *h, const RibbitAlignJob *jobs, size_t n, const std::vector<rb::SswEnds> &ends, std::vector<rb::SswPath> &paths);

}  // namespace rbapi
