// window_stage.hip -- the per-motif window state machine of processShiftXORswithSubstitutions
// (parse_substitute_shiftxor.cpp:430-574) and processShiftXORsAnchored (parse_anchored_shiftxor.cpp:580-723)
// on the GPU: from the pass-streaks the window-scan kernels found to the addSeedToSeedPositions* calls those
// loops make, length-filtered and in the reference's call order.  Nothing but the calls that reach a merge
// leaves the chip.
//
// The reference walks every motif's windows in position order with three variables (pending group start / end,
// current streak start).  Restated over streaks i = [s_i, e_i) (first passing window, first non-passing window
// after it; closed by a failing window, an N or the end of the record) the machine is local:
//   * streak i+1 JOINS the group streak i belongs to iff streak i was closed by a failing window and
//     e_i + 7 >= s_{i+1}  (`last_ends[m] < window_start` is false at :477);
//   * a streak closed by an N is dropped (:433-458), one still open at the end of the record is flushed (:534-574),
//     so both always end their group;
//   * every group makes exactly one call (start of its first streak, e_b + 7 of its last streak b that was closed
//     by a failing window).  It is made at scan position q + 7, q = the first evaluated window start > e_b + 7
//     (:515-527 or :477-491), or -- when an N closes a streak that joined the group and the group's end lies
//     before that streak's last passing window -- at the N (:441-452).  Without an evaluated window left, the call
//     belongs to the end-of-sequence flush.
// "Start of the group" is a prefix maximum over the motif's streaks (the start of the latest streak that did not
// join), so the whole machine is one scan plus one independent decision per streak.
//
// What the host merges need (event_stream.h: CompactCalls) is the calls that pass the stage's length filter in
// call order -- scan position major, motif minor -- each with the largest end among ALL calls before it (those
// only moved the merge's cursors).  A call made at position x has end <= x - 8, with equality unless an N is
// involved, so for a call with end == x - 8 no earlier call can reach beyond it and the bound is moot.  The others
// ("edge" calls: x is an N, or the window before the reporting window was not evaluated) are few; for them the
// bound is the larger of (a) the ends of the edge calls before them and (b) y - 8 for the last position y < x at
// which any ordinary call was made -- kept in a bitmap over positions that every ordinary call sets.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "kernels.h"
#include "ribbit_hip.h"

namespace rb {

namespace {

constexpr uint32_t NONE32 = 0xffffffffu;

// ------------------------------------------------------------------------------- evaluated windows
// E bit q = the window starting at q holds no N and lies inside the record (`valid_position >= window_length`,
// parse_substitute_shiftxor.cpp:469): no break bit in [q, q+7].  first_word[w] (stored reversed, see
// launch_eval_planes) = the first word >= w with an evaluated window.
__global__ __launch_bounds__(256) void eval_plane_kernel(const uint32_t *__restrict__ brk, uint32_t nwords,
                                                         uint32_t *__restrict__ eval, uint32_t *__restrict__ first_rev) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= nwords) return;
    uint64_t b = (uint64_t)brk[w] | ((uint64_t)brk[w + 1] << 32);
    b |= b >> 1;
    b |= b >> 2;
    b |= b >> 4;
    const uint32_t e = ~(uint32_t)b;
    eval[w] = e;
    first_rev[nwords - 1u - w] = e ? w : NONE32;
}

struct MinOp {
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; }
};
struct MaxOp64 {
    __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; }
};
struct MaxOp32 {
    __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

struct EvalView {
    const uint32_t *eval;
    const uint32_t *first_rev;
    const uint32_t *brk;
    uint32_t nwords;
    // smallest evaluated window start >= x, NONE32 if there is none
    __device__ uint32_t first_evaluated(uint32_t x) const {
        const uint32_t w = x >> 5;
        if (w >= nwords) return NONE32;
        const uint32_t m = eval[w] & (0xffffffffu << (x & 31u));
        if (m) return (w << 5) + (uint32_t)__builtin_ctz(m);
        if (w + 1u >= nwords) return NONE32;
        const uint32_t w2 = first_rev[nwords - 2u - w];       // first word >= w + 1 with an evaluated window
        if (w2 == NONE32) return NONE32;
        return (w2 << 5) + (uint32_t)__builtin_ctz(eval[w2]);
    }
    __device__ bool evaluated(uint32_t q) const { return (eval[q >> 5] >> (q & 31u)) & 1u; }
    __device__ bool is_break(uint32_t p) const { return (brk[p >> 5] >> (p & 31u)) & 1u; }
};

// ---------------------------------------------------------------------------------- group starts
// value of streak i for the prefix maximum: (motif index, start + 1) if the streak opens a group, (motif index, 0)
// if it joins its predecessor's.  Motif-major order makes the motif index a segment barrier.
struct GroupHead {
    const RibbitRun *runs;
    uint32_t m_lo;
    __device__ uint64_t operator()(uint32_t i) const {
        const RibbitRun r = runs[i];
        bool joins = false;
        if (i > 0) {
            const RibbitRun p = runs[i - 1];
            joins = p.mlen == r.mlen && p.term == RIBBIT_TERM_ZERO && p.end + 7 >= r.start;
        }
        return ((uint64_t)((uint32_t)r.mlen - m_lo) << 32) | (joins ? 0u : (uint32_t)r.start + 1u);
    }
};

// ------------------------------------------------------------------------------------ the calls
constexpr int EMIT_ITEMS = 8;
constexpr int EMIT_TILE = 256 * EMIT_ITEMS;

struct EmitArgs {
    const RibbitRun *runs;
    uint32_t n;
    const uint64_t *group;        // inclusive prefix maximum of GroupHead
    EvalView ev;
    int32_t length;
    uint32_t m_lo, nm;
    const int32_t *min_span;      // [nm]: smallest end - start that passes the stage's length filter (seedlen_cutoffs)
    int full;                     // 1: every call goes to the main list (no filter, no bound bookkeeping)
    uint64_t *keys, *vals;        // main list
    uint32_t cap;
    uint64_t *edge_keys, *edge_vals;
    uint32_t edge_cap;
    RibbitCall *flush;            // [nm], mlen == 0: none
    uint32_t *bitmap;             // positions at which an ordinary call was made
    uint32_t *counters;           // WS_* below
    // The loaded record may be one PIECE (chunk + halos) of a longer record that several GPUs scan: a call belongs to the
    // chunk that owns its scan position, own_lo <= pos < own_hi (a whole record: 0, 0xffffffff).  Calls of the halos are
    // the neighbours' and leave no trace here (no list entry, no bitmap bit, no bound).  z_lo != 0: streak events before
    // position z_lo are artefacts of the piece's artificial left end, so an owned call whose group starts within 8
    // positions of it may be a longer group cut short: WS_INEXACT tells the caller to load a longer left halo.
    // keep_flush: the piece ends where the record ends, so the end-of-sequence calls are this chunk's.
    uint32_t own_lo, own_hi, z_lo;
    int keep_flush;
};

__device__ __forceinline__ uint64_t call_key(uint32_t pos, uint32_t mlen) { return ((uint64_t)pos << 10) | mlen; }
__device__ __forceinline__ uint64_t call_val(uint32_t start, uint32_t end) { return ((uint64_t)start << 32) | end; }

__global__ __launch_bounds__(256) void window_calls_kernel(EmitArgs a) {
    __shared__ uint64_t s_keys[EMIT_TILE];
    __shared__ uint64_t s_vals[EMIT_TILE];
    __shared__ uint32_t s_n, s_base, s_max_end;
    if (threadIdx.x == 0) { s_n = 0; s_max_end = 0; }
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * (uint32_t)EMIT_TILE;
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int k = 0; k < EMIT_ITEMS; ++k) {
        const uint32_t i = tile0 + (uint32_t)k * 256u + threadIdx.x;
        bool in_loop = false, edge = false, kept = false;
        uint32_t pos = 0, start = 0, end = 0, mlen = 0;
        if (i < a.n) {
            const RibbitRun r = a.runs[i];
            mlen = (uint32_t)r.mlen;
            bool continues = false;
            if (r.term == RIBBIT_TERM_ZERO && i + 1u < a.n) {
                const RibbitRun nx = a.runs[i + 1];
                continues = nx.mlen == r.mlen && r.end + 7 >= nx.start;
            }
            if (!continues) {
                bool joined = false;
                RibbitRun p{};
                if (i > 0) {
                    p = a.runs[i - 1];
                    joined = p.mlen == r.mlen && p.term == RIBBIT_TERM_ZERO && p.end + 7 >= r.start;
                }
                const uint32_t gstart = (uint32_t)a.group[i] - 1u;
                bool have = false, to_flush = false;
                if (r.term == RIBBIT_TERM_ZERO) {
                    have = true; start = gstart; end = (uint32_t)r.end + 7u;
                    const uint32_t q = a.ev.first_evaluated(end + 1u);
                    if (q == NONE32) to_flush = true; else pos = q + 7u;
                } else if (r.term == RIBBIT_TERM_N) {
                    if (joined) {
                        have = true; start = gstart; end = (uint32_t)p.end + 7u;
                        if ((int32_t)end < r.end) pos = (uint32_t)r.end + 7u;           // reported at the N (:441-452)
                        else {
                            const uint32_t q = a.ev.first_evaluated(end + 1u);
                            if (q == NONE32) to_flush = true; else pos = q + 7u;
                        }
                    }
                } else {                                                                // open at the end of the record
                    have = true; to_flush = true;
                    start = joined ? gstart : (uint32_t)r.start;
                    end = (uint32_t)a.length;
                }
                if (have && to_flush) {
                    const uint32_t mi = mlen - a.m_lo;
                    if (!a.keep_flush) {
                        // no evaluated window left in a piece that ends before the record does: the call is made beyond the piece
                    } else if (mi < a.nm) {
                        // at most one per motif: an open streak and an unreported group cannot both be left over
                        if (atomicExch(&a.flush[mi].mlen, (int32_t)mlen) != 0) atomicOr(&a.counters[WS_FLAGS], (uint32_t)WS_TWO_FLUSH);
                        a.flush[mi].pos = a.length; a.flush[mi].start = (int32_t)start; a.flush[mi].end = (int32_t)end;
                        if (a.z_lo && start < a.z_lo + 8u) atomicOr(&a.counters[WS_INEXACT], 1u);
                    } else atomicOr(&a.counters[WS_FLAGS], (uint32_t)WS_BAD_MOTIF);
                } else if (have && pos >= a.own_lo && pos < a.own_hi) {
                    in_loop = true;
                    if (a.z_lo && start < a.z_lo + 8u) atomicOr(&a.counters[WS_INEXACT], 1u);
                    const uint32_t mi = mlen - a.m_lo;
                    kept = a.full || (mi < a.nm && (int32_t)(end - start) >= a.min_span[mi]);
                    if (!a.full) {
                        edge = a.ev.is_break(pos) || pos < 8u || !a.ev.evaluated(pos - 8u);
                        if (!edge) {
                            if (end + 8u != pos) atomicOr(&a.counters[WS_FLAGS], (uint32_t)WS_NOT_PROMPT);
                            atomicOr(&a.bitmap[pos >> 5], 1u << (pos & 31u));
                        }
                        atomicMax(&s_max_end, end + 1u);
                    }
                }
            }
        }
        // main list: staged in LDS, one global atomic per workgroup
        const bool to_main = in_loop && kept;
        const unsigned long long mm = __ballot(to_main);
        if (mm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(mm));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (to_main) {
                const uint32_t at = base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                s_keys[at] = call_key(pos, mlen);
                s_vals[at] = call_val(start, end);
            }
        }
        // edge list (rare): one global atomic per wave that has any
        const bool to_edge = in_loop && edge;
        const unsigned long long em = __ballot(to_edge);
        if (em) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&a.counters[WS_N_EDGE], (uint32_t)__popcll(em));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (to_edge) {
                const uint32_t at = base + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
                if (at < a.edge_cap) {
                    a.edge_keys[at] = call_key(pos, mlen);
                    a.edge_vals[at] = call_val(start, end) | (kept ? (1ull << 63) : 0ull);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base = s_n ? atomicAdd(&a.counters[WS_N_MAIN], s_n) : 0u;
        if (s_max_end) atomicMax(&a.counters[WS_MAX_END], s_max_end);
    }
    __syncthreads();
    const uint32_t n = s_n, base = s_base;
    for (uint32_t j = threadIdx.x; j < n; j += 256u) {
        if (base + j < a.cap) { a.keys[base + j] = s_keys[j]; a.vals[base + j] = s_vals[j]; }
    }
}

// -------------------------------------------------------------- bounds of the edge calls (compact mode)
// last_set_word[w] = the last bitmap word <= w that is not empty (NONE32 -> 0 after the max-scan of (word+1))
__global__ __launch_bounds__(256) void bitmap_words_kernel(const uint32_t *__restrict__ bitmap, uint32_t nwords,
                                                           uint32_t *__restrict__ word1) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w < nwords) word1[w] = bitmap[w] ? w + 1u : 0u;
}

__global__ __launch_bounds__(256) void edge_ends_kernel(const uint64_t *__restrict__ edge_vals, uint32_t n, uint32_t *__restrict__ end1) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < n) end1[j] = (uint32_t)edge_vals[j] + 1u;
}

// For every kept edge call j (edge list sorted by key): bound = max(end of an edge call before it, y - 8 for the
// last position y < pos_j with an ordinary call), written to pend[index of the call in the sorted main list].
__global__ __launch_bounds__(256) void edge_bounds_kernel(const uint64_t *__restrict__ edge_keys, const uint64_t *__restrict__ edge_vals,
                                                          const uint32_t *__restrict__ end1_before, uint32_t n_edge,
                                                          const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ last_word1,
                                                          const uint64_t *__restrict__ main_keys, uint32_t n_main,
                                                          int32_t *__restrict__ pend, uint32_t *__restrict__ counters, int32_t pos_offset) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n_edge) return;
    const uint64_t val = edge_vals[j];
    if (!(val >> 63)) return;
    const uint64_t key = edge_keys[j];
    const uint32_t pos = (uint32_t)(key >> 10);
    int32_t bound = (int32_t)end1_before[j] - 1;                     // exclusive prefix maximum of end + 1
    // last ordinary call strictly before pos
    const uint32_t w = pos >> 5;
    uint32_t m = bitmap[w] & ((1u << (pos & 31u)) - 1u);
    int64_t y = -1;
    if (m) y = (int64_t)(w << 5) + 31 - __builtin_clz(m);
    else if (w > 0) {
        const uint32_t w1 = last_word1[w - 1u];
        if (w1) y = (int64_t)((w1 - 1u) << 5) + 31 - __builtin_clz(bitmap[w1 - 1u]);
    }
    if (y >= 8 && (int32_t)(y - 8) > bound) bound = (int32_t)(y - 8);
    // the call's place in the sorted main list (keys are unique)
    uint32_t lo = 0, hi = n_main;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (main_keys[mid] < key) lo = mid + 1u; else hi = mid;
    }
    if (lo >= n_main || main_keys[lo] != key) { atomicOr(&counters[WS_FLAGS], (uint32_t)WS_EDGE_LOST); return; }
    pend[lo] = bound < 0 ? bound : bound + pos_offset;      // piece coordinates -> record coordinates
}

// The anchored scan's group filter (kernels.hip) drops the groups whose call cannot pass the length filter before they
// become events; what such a call would have left here is a bit in the map of ordinary call positions (it is made at
// end-of-group + 15: its window e + 8 is evaluated, else the group had been kept) and a candidate for the largest end
// (end-of-group + 7).  dropmap bit e = a dropped group ends at e.  Calls outside the own range are a neighbour's.
__global__ __launch_bounds__(256) void merge_dropmap_kernel(const uint32_t *__restrict__ dropmap, uint32_t drop_words, uint32_t n_words,
                                                            uint32_t own_lo, uint32_t own_hi, uint32_t *__restrict__ bitmap,
                                                            uint32_t *__restrict__ counters) {
    __shared__ uint32_t s_top;
    if (threadIdx.x == 0) s_top = 0;
    __syncthreads();
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    uint32_t top = 0;
    if (w <= n_words) {
        const uint32_t hi = w < drop_words ? dropmap[w] : 0u, lo = (w > 0 && w - 1u < drop_words) ? dropmap[w - 1u] : 0u;
        uint32_t bits = (hi << 15) | (lo >> 17);                      // bit p <- a group ending at p - 15
        const uint64_t base = (uint64_t)w << 5;
        if (base + 32u <= own_lo || base >= own_hi) bits = 0;
        else {
            if (base < own_lo) bits &= 0xffffffffu << (own_lo - (uint32_t)base);
            if (base + 32u > own_hi) bits &= 0xffffffffu >> ((uint32_t)(base + 32u) - own_hi);
        }
        if (bits) {
            bitmap[w] |= bits;
            top = (w << 5) + 31u - (uint32_t)__builtin_clz(bits) - 7u;      // end + 1 of the call made at that position
        }
    }
    if (top) atomicMax(&s_top, top);
    __syncthreads();
    if (threadIdx.x == 0 && s_top) atomicMax(&counters[WS_MAX_END], s_top);
}

__global__ __launch_bounds__(256) void assemble_calls_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ vals,
                                                             uint32_t n, RibbitCall *__restrict__ out, int32_t pos_offset) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i], v = vals[i];
    out[i] = RibbitCall{(int32_t)(k >> 10) + pos_offset, (int32_t)(k & 1023u), (int32_t)(v >> 32) + pos_offset, (int32_t)(uint32_t)v + pos_offset};
}

}  // namespace

size_t window_stage_scratch_bytes(size_t n_streaks, size_t n_words, size_t n_calls, size_t n_edge, int key_bits) {
    size_t need = 0, b = 0;
    (void)rocprim::inclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, n_words, MinOp());
    need = std::max(need, b);
    (void)rocprim::inclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, n_words, MaxOp32());
    need = std::max(need, b);
    {
        auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), GroupHead{nullptr, 0u});
        (void)rocprim::inclusive_scan(nullptr, b, in, (uint64_t *)nullptr, n_streaks, MaxOp64());
        need = std::max(need, b);
    }
    (void)rocprim::radix_sort_pairs(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                    std::max(n_calls, n_edge), 0u, (unsigned)key_bits);
    need = std::max(need, b);
    (void)rocprim::exclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n_edge, MaxOp32());
    need = std::max(need, b);
    return need + 256;
}

hipError_t launch_eval_planes(const uint32_t *brk, uint32_t nwords, uint32_t *eval, uint32_t *first_rev, uint32_t *word_tmp,
                              void *scratch, size_t scratch_bytes, hipStream_t stream) {
    if (nwords == 0) return hipSuccess;
    hipLaunchKernelGGL(eval_plane_kernel, dim3((nwords + 255u) / 256u), dim3(256), 0, stream, brk, nwords, eval, word_tmp);
    // suffix minimum of "first word with an evaluated window" = prefix minimum of the reversed array
    return rocprim::inclusive_scan(scratch, scratch_bytes, word_tmp, first_rev, (size_t)nwords, MinOp(), stream);
}

hipError_t launch_group_starts(const RibbitRun *runs, uint32_t n, uint32_t m_lo, uint64_t *group, void *scratch,
                               size_t scratch_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0u), GroupHead{runs, m_lo});
    return rocprim::inclusive_scan(scratch, scratch_bytes, in, group, (size_t)n, MaxOp64(), stream);
}

void launch_window_calls(const WindowCallsLaunch &w, hipStream_t stream) {
    if (w.n_streaks == 0) return;
    EmitArgs a;
    a.runs = w.runs; a.n = w.n_streaks; a.group = w.group;
    a.ev = EvalView{w.eval, w.first_rev, w.brk, w.n_words};
    a.length = (int32_t)w.length; a.m_lo = w.m_lo; a.nm = w.nm; a.min_span = w.min_span; a.full = w.full;
    a.keys = w.keys; a.vals = w.vals; a.cap = w.cap;
    a.edge_keys = w.edge_keys; a.edge_vals = w.edge_vals; a.edge_cap = w.edge_cap;
    a.flush = w.flush; a.bitmap = w.bitmap; a.counters = w.counters;
    a.own_lo = w.own_lo; a.own_hi = w.own_hi; a.z_lo = w.z_lo; a.keep_flush = w.keep_flush;
    hipLaunchKernelGGL(window_calls_kernel, dim3((w.n_streaks + (uint32_t)EMIT_TILE - 1u) / (uint32_t)EMIT_TILE), dim3(256), 0, stream, a);
}

void launch_merge_dropmap(const uint32_t *dropmap, uint32_t drop_words, uint32_t n_words, uint32_t own_lo, uint32_t own_hi, uint32_t *bitmap,
                          uint32_t *counters, hipStream_t stream) {
    hipLaunchKernelGGL(merge_dropmap_kernel, dim3((n_words + 256u) / 256u), dim3(256), 0, stream, dropmap, drop_words, n_words, own_lo, own_hi,
                       bitmap, counters);
}

hipError_t launch_sort_calls(uint64_t *keys_in, uint64_t *vals_in, uint64_t *keys_out, uint64_t *vals_out, uint32_t n, int key_bits,
                             void *scratch, size_t scratch_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(scratch, scratch_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)key_bits, stream);
}

hipError_t launch_edge_bounds(const uint64_t *edge_keys, const uint64_t *edge_vals, uint32_t n_edge, uint32_t *edge_tmp, uint32_t *edge_end1,
                              const uint32_t *bitmap, uint32_t *word_tmp, uint32_t *last_word1, uint32_t n_words, const uint64_t *main_keys,
                              uint32_t n_main, int32_t *pend, uint32_t *counters, int32_t pos_offset, void *scratch, size_t scratch_bytes,
                              hipStream_t stream) {
    if (n_edge == 0) return hipSuccess;
    hipLaunchKernelGGL(edge_ends_kernel, dim3((n_edge + 255u) / 256u), dim3(256), 0, stream, edge_vals, n_edge, edge_tmp);
    hipError_t e = rocprim::exclusive_scan(scratch, scratch_bytes, edge_tmp, edge_end1, 0u, (size_t)n_edge, MaxOp32(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bitmap_words_kernel, dim3((n_words + 255u) / 256u), dim3(256), 0, stream, bitmap, n_words, word_tmp);
    e = rocprim::inclusive_scan(scratch, scratch_bytes, word_tmp, last_word1, (size_t)n_words, MaxOp32(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(edge_bounds_kernel, dim3((n_edge + 255u) / 256u), dim3(256), 0, stream, edge_keys, edge_vals, edge_end1, n_edge,
                       bitmap, last_word1, main_keys, n_main, pend, counters, pos_offset);
    return hipSuccess;
}

void launch_assemble_calls(const uint64_t *keys, const uint64_t *vals, uint32_t n, RibbitCall *out, int32_t pos_offset, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(assemble_calls_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, keys, vals, n, out, pos_offset);
}

}  // namespace rb
