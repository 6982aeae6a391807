// small_motifs.hip -- possibleMotifs (parse_smallmotif_seed.cpp:76-188) for every dispatched seed with m <= 10 of a
// record at once.  Host twin: discover_small_motifs in refine.cpp, which spells out the
// reference's bookkeeping (rotation class of the rolling m-base window; per class the first start, the last end, the
// number of units and the start of the latest unit; a class that comes back more than 3m past its end is reported --
// if long enough -- and starts over; the survivors are reported when the seed ends).
//
// The walk along a seed is sequential (every step reads what the step before left in the class it hits), the seeds
// are independent and there are hundreds of thousands of them: one wavefront per seed.  The wavefront's lanes ARE the
// class table -- lane i holds the i-th class the seed has shown, in order of first appearance -- so a step is
//   * the rolling window (wave-uniform, in scalar registers),
//   * its smallest rotation: lane r < m rotates by r bases, a four-step DPP minimum over the first row of sixteen lanes,
//   * one compare of every lane's class with it and a ballot: the lane that owns the class updates its registers,
//     or the next free lane takes it.
// The bases come in through one coalesced load per 64 positions and v_readlane.  A seed with more than 64 classes,
// or more than 64 reports before its end, is flagged and left to the host twin (2 in 1000 of the seeds of the
// simulated records: long impure ones; a class table in registers is the point of the design).
//
// What the host still does with the result: the reference reports the survivors in the iteration order of its
// std::unordered_map (Q10), which depends on the keys' insertion order; the kernel hands back all classes in order of
// first appearance and refine.cpp replays just those insertions (a few per seed instead of one look-up per base).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "kernels.h"

namespace rb {

namespace {

__device__ __forceinline__ uint32_t row_min(uint32_t v) {      // minimum over the 16 lanes of a DPP row, in each of them
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

}  // namespace

// jobs[s] = {seed start, seed end, m, dispatch index}; head[dispatch index] = {first record, early reports, classes, flags}
// (flags: 1 = more than 64 classes or early reports, 2 = the record arena is full: the host computes the seed itself);
// records: {class, first start, last end, units}: the early reports (only those long enough, in the order the reference
// pushes them), then the classes with their final state: all of them, in order of first appearance, when two or more will
// be reported at the seed's end (the host needs every key to replay the map's order); otherwise just the one that will be,
// or none.
__global__ __launch_bounds__(64) void small_motifs_kernel(const uint8_t *__restrict__ sym, int64_t length, const int4 *__restrict__ jobs,
                                                          int64_t njobs, SmallMotifLimits lim, uint4 *__restrict__ records,
                                                          uint32_t record_cap, uint32_t *__restrict__ record_count, int4 *__restrict__ head,
                                                          const int32_t *__restrict__ longest, int32_t longest_threshold) {
    __shared__ uint4 early[64];
    const int64_t s = blockIdx.x;
    if (s >= njobs) return;
    const int lane = (int)threadIdx.x;
    int4 jb = jobs[s];
    // longest != null: `jobs` is the dispatch list itself {start, end, m, type} and this the selection the host used to make
    // (parse_smallmotif_seed.cpp:234-236: m <= 10 and a long enough run of matches); the seed's index is its place in the list
    if (longest) {
        if (jb.z > 10 || jb.z < 1 || longest[s] < longest_threshold) return;
        jb.w = (int)s;
    }
    const int start = jb.x, end = jb.y, m = jb.z;
    const int L = (int)length;
    // the seed's sequence length: seed + one motif, cut at the first N (parse_smallmotif_seed.cpp:214-221)
    int seq_len = (end - start) + m;
    for (int p0 = start; p0 < end + m; p0 += 64) {
        const int p = p0 + lane;
        const bool is_n = p < end + m && p >= 0 && p < L && sym[p] == 4;
        const unsigned long long hit = __ballot(is_n);
        if (hit) { seq_len = p0 + (int)__builtin_ctzll(hit) - start; break; }
    }
    const int stop = min(start + seq_len, L - 1);
    const uint32_t mask = (1u << (2 * m)) - 1u;                  // m <= 10
    const int warm = lim.first_window[m], min_len = lim.min_length[m], min_units = lim.min_units[m];
    uint32_t my_class = 0;
    int my_first = 0, my_last = 0, my_units = 0, my_unit_start = 0;
    int count = 0, n_early = 0;
    bool overflow = false;
    uint32_t window = 0;
    for (int j0 = start; j0 < stop && !overflow; j0 += 64) {
        const int p = j0 + lane;
        const int my_code = (p < stop && p >= 0) ? (int)(sym[p] & 3u) : 0;      // N reads as A (fasta_utils.cpp:111-113)
        const int n = min(64, stop - j0);
        for (int t = 0; t < n; ++t) {
            const uint32_t code = (uint32_t)__builtin_amdgcn_readlane(my_code, t);
            window = ((window << 2) | code) & mask;
            const int j = j0 + t;
            if (j - start < warm) continue;
            const uint32_t rot = lane == 0 ? window : (((window << (2 * lane)) | (window >> (2 * (m - lane)))) & mask);
            const uint32_t cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_min(lane < m ? rot : 0xffffffffu));
            const int wstart = j - (m - 1), wend = j + 1;
            const bool mine = lane < count && my_class == cls;
            if (__ballot(mine) == 0ull) {
                if (count == 64) { overflow = true; break; }
                if (lane == count) { my_class = cls; my_first = wstart; my_last = wend; my_units = 1; my_unit_start = wstart; }
                ++count;
                continue;
            }
            bool report = false;
            if (mine) {
                if (wstart - my_last > 3 * m) {
                    report = my_last - my_first >= min_len && my_units >= min_units;
                    if (report && n_early < 64) early[n_early] = make_uint4(my_class, (uint32_t)my_first, (uint32_t)my_last, (uint32_t)my_units);
                    my_first = wstart; my_last = wend; my_units = 1; my_unit_start = wstart;
                } else {
                    if (wstart - my_unit_start >= m) { my_unit_start = wstart; ++my_units; }
                    my_last = wend;
                }
            }
            if (__ballot(report) != 0ull) {
                if (n_early == 64) { overflow = true; break; }
                ++n_early;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    int flags = overflow ? 1 : 0;
    // The survivors' order only matters between those that are reported.  With fewer than two of them (nearly every seed)
    // the classes that are not reported need not leave the device: one record or none instead of eight on average.
    const bool reported = lane < count && my_last - my_first >= min_len && my_units >= min_units;
    const unsigned long long rmask = __ballot(reported);
    const int n_reported = __popcll(rmask);
    const bool all_classes = n_reported >= 2;
    const int n_final = all_classes ? count : n_reported;
    uint32_t base = 0;
    const uint32_t total = overflow ? 0u : (uint32_t)(n_early + n_final);
    if (total) {
        if (lane == 0) base = atomicAdd(record_count, total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint64_t)base + total > record_cap) flags |= 2;
        else {
            if (lane < n_early) records[base + (uint32_t)lane] = early[lane];
            const uint4 mine = make_uint4(my_class, (uint32_t)my_first, (uint32_t)my_last, (uint32_t)my_units);
            if (all_classes) { if (lane < count) records[base + (uint32_t)n_early + (uint32_t)lane] = mine; }
            else if (reported) records[base + (uint32_t)n_early] = mine;
        }
    }
    if (lane == 0) head[jb.w] = make_int4((int)base, n_early, overflow ? 0 : n_final, flags);
}

void launch_small_motifs(const uint8_t *sym, int64_t length, const void *jobs, int64_t njobs, const SmallMotifLimits &lim, void *records,
                         uint32_t record_cap, uint32_t *record_count, void *head, hipStream_t stream, const int32_t *longest, int32_t longest_threshold) {
    if (njobs <= 0) return;
    hipLaunchKernelGGL(small_motifs_kernel, dim3((unsigned)njobs), dim3(64), 0, stream, sym, length, (const int4 *)jobs, njobs, lim,
                       (uint4 *)records, record_cap, record_count, (int4 *)head, longest, longest_threshold);
}

}  // namespace rb
