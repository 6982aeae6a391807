// api_core.cpp -- handles, streams, loading a record, timers, plane queries; see api_internal.h for the map of the files behind include/ribbit_hip.h.
// There is no CPU fallback for any scan anywhere in this library.
#include "api_internal.h"

#include <sys/mman.h>

namespace rbapi {

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


// ---- page-locked host memory (api_internal.h)
namespace {
constexpr size_t PINNED_HUGE_FROM = (size_t)64 << 20;
std::mutex g_mapped_mu;
std::vector<std::pair<void *, size_t>> g_mapped;      // regions made by mmap + hipHostRegister (a handful per process)
}  // namespace

int pinned_alloc(size_t bytes, void **out) {
    *out = nullptr;
    if (bytes >= PINNED_HUGE_FROM) {
        const size_t len = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p != MAP_FAILED) {
            (void)madvise(p, len, MADV_HUGEPAGE);
            // fault the pages in on several threads (one touch per 4 KB covers both page sizes): registration alone would do it on one
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(std::thread::hardware_concurrency(), 16u), len >> 26));
            auto touch = [p, len, nt](unsigned t) {
                volatile char *c = static_cast<volatile char *>(p);
                for (size_t i = len / nt * t, e = t + 1 == nt ? len : len / nt * (t + 1); i < e; i += 4096) c[i] = 0;
            };
            std::vector<std::thread> pool;
            unsigned started = 1;          // part 0 is this thread's
            try { for (; started < nt; ++started) pool.emplace_back(touch, started); } catch (...) {}
            touch(0);
            for (unsigned t = started; t < nt; ++t) touch(t);      // (parts whose thread could not start)
            for (std::thread &th : pool) th.join();
            if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> lk(g_mapped_mu);
                g_mapped.emplace_back(p, len);
                *out = p;
                return RIBBIT_OK;
            }
            (void)hipGetLastError();
            munmap(p, len);
        }
    }
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(RIBBIT_E_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
    return RIBBIT_OK;
}

namespace {
std::mutex g_free_mu;
std::vector<void *> g_free_later, g_unpin_later;
}
// A buffer that is being replaced by a larger one is released later (at the end of refinement, at a handle's close, or when an
// allocation fails): hipFree and the unpinning calls wait for every stream of the device -- 20 ms at a time beside another feeder's
// kernels, 240 ms of a chromosome's first refinement until round 4.
void device_free_pending();
namespace {
// (a process that only ever scans never reaches the end of a refinement: what it has put aside is released when it has become much)
void put_aside(std::vector<void *> &list, void *p) {
    size_t waiting;
    { std::lock_guard<std::mutex> lk(g_free_mu); list.push_back(p); waiting = g_free_later.size() + g_unpin_later.size(); }
    if (waiting > 256) device_free_pending();
}
}
void device_free_later(void *p) { put_aside(g_free_later, p); }
void pinned_free_later(void *p) { put_aside(g_unpin_later, p); }
void device_free_pending() {
    std::vector<void *> mine, pinned;
    { std::lock_guard<std::mutex> lk(g_free_mu); mine.swap(g_free_later); pinned.swap(g_unpin_later); }
    for (void *p : mine) (void)hipFree(p);
    for (void *p : pinned) pinned_free(p);
}

void pinned_free(void *p) {
    if (!p) return;
    size_t len = 0;
    {
        std::lock_guard<std::mutex> lk(g_mapped_mu);
        for (size_t i = 0; i < g_mapped.size(); ++i)
            if (g_mapped[i].first == p) { len = g_mapped[i].second; g_mapped.erase(g_mapped.begin() + (std::ptrdiff_t)i); break; }
    }
    if (len) { (void)hipHostUnregister(p); munmap(p, len); }
    else (void)hipHostFree(p);
}

}  // namespace rbapi

namespace rbapi {

int bind_device(const RibbitHandle *h) {
    HIP_TRY(hipSetDevice(h->device));
    return RIBBIT_OK;
}

int is_gfx950(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

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

}  // namespace rbapi

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

int ribbit_hip_device_pci_bus_id(int device, char *out, size_t cap) {
    if (!out || cap < 16) return fail(RIBBIT_E_ARG, "bad argument");
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)cap, device));
    return RIBBIT_OK;
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
    for (RibbitHandle *fa : h->feed_aux) (void)ribbit_hip_close(fa);
    h->feed_aux.clear();
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    device_free_pending();
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
    h->mg.release(); h->h_seed_stage.release(); h->h_longest_stage.release();
    std::free(h->bed_raw); h->bed_raw = nullptr;
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
    return pinned_alloc(bytes, out);
}

int ribbit_hip_host_free(void *p) {
    pinned_free(p);
    return RIBBIT_OK;
}

int64_t ribbit_hip_guard_hits(const RibbitHandle *h) { return h ? h->lists.guard_hits : 0; }

int ribbit_hip_debug_set_event_capacity(RibbitHandle *h, size_t events) {
    if (!h) return fail(RIBBIT_E_ARG, "null argument");
    h->debug_first_cap = events;
    return RIBBIT_OK;
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
