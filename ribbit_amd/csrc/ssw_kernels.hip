// ssw_kernels.hip -- the two striped Smith-Waterman passes of every first-level alignment of a record, batched
// on the GPU (gfx950).  Reference: the vendored SSW library v1.2.5 as ribbit calls it through Aligner::Align
// (parse_seed.cpp:404, parse_smallmotif_seed.cpp:270): sw_sse2_byte (ssw.c:197-386), sw_sse2_word (:412-588) and
// their orchestration in ssw_align (:843-891).  Same evaluation order as ssw_exact.cpp (vector j, lane l <-> query
// position j + l*segLen), so the scores, end points and tie-breaks are the library's, bit for bit; the host then
// only runs the banded traceback between the end points (ssw_exact.cpp: ssw_finish).  Pinned against the
// reference library itself in tests/test_ssw.py (GPU leg in tests/test_ssw_gpu.py).
//
// Mapping.  One SSE2 register of the library = one DPP row of 16 lanes: an alignment owns a row, a wavefront
// carries four alignments.  _mm_slli_si128(v, 1 element) is DPP row_shr:1 with zero fill, the horizontal maximum
// four row_ror steps; the 16-bit pass uses lanes 0..7 of its row.  H/E columns live in LDS ([stripe][lane], 16-bit
// cells); saturating byte / word arithmetic is done in 32-bit registers with explicit clamps.  Alignments of one
// wavefront diverge (their loops have different trip counts), so jobs are launched sorted by size.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "kernels.h"

namespace rb {

namespace {

constexpr int SSW_BIAS = 2;          // |mismatch|: the byte profile is stored with this bias (ssw.c:117-150)
constexpr int SSW_GAP_O = 3, SSW_GAP_E = 1;

__device__ __forceinline__ int row_shr1(int v) {      // lane l of a row receives lane l-1, lane 0 receives 0
    return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
}
__device__ __forceinline__ int row_max(int v) {       // maximum over the 16 lanes of a row, in every lane
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false));   // row_ror:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}
__device__ __forceinline__ int row_min(int v) { return -row_max(-v); }
__device__ __forceinline__ bool row_any(bool pred, int row) {
    const unsigned long long m = __ballot(pred);
    return ((m >> (16 * row)) & 0xffffull) != 0ull;
}

// kBaseTranslation (ssw_cpp.cpp:12-27)
__device__ __forceinline__ int ssw_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

struct PassOut { int score, ref, read, score2, ref2; };

// One striped pass of one alignment on one row.  W = 16: sw_sse2_byte, W = 8: sw_sse2_word.
// ref(i): code of reference position i;  rd(q): code of read position q (q < read_len).
// LDS (per row): hA, hB, E, hbest: [seg][16] uint16;  colmax: [ref_len] uint16.
template <int W, typename RefAt, typename ReadAt>
__device__ __forceinline__ PassOut striped_pass(RefAt ref, int dir, int ref_len, ReadAt rd, int read_len, int terminate, int mask_len,
                                                uint16_t *hA, uint16_t *hB, uint16_t *E, uint16_t *hbest, uint16_t *colmax,
                                                int lr, int row) {
    const bool use = lr < W;                          // lanes 8..15 idle in the 16-bit pass
    const int seg = (read_len + W - 1) / W;
    const int hi_clamp = W == 16 ? 255 : 32767;
    for (int j = 0; j < seg; ++j) { hA[j * 16 + lr] = 0; hB[j * 16 + lr] = 0; E[j * 16 + lr] = 0; hbest[j * 16 + lr] = 0; }
    for (int i = lr; i < ref_len; i += 16) colmax[i] = 0;
    __builtin_amdgcn_wave_barrier();
    uint16_t *h_store = hA, *h_load = hB;
    int run_max = 0, run_mark = 0, best = 0, end_ref = W == 16 ? -1 : 0;
    bool overflow = false;
    const int begin = dir ? ref_len - 1 : 0, stop = dir ? -1 : ref_len, step = dir ? -1 : 1;
    for (int i = begin; i != stop; i += step) {
        const int rc = ref(i);
        int F = 0, cmax = 0;
        int H = row_shr1((int)h_store[(seg - 1) * 16 + lr]);
        if (!use) H = 0;
        { uint16_t *t = h_store; h_store = h_load; h_load = t; }
        for (int j = 0; j < seg; ++j) {
            const int q = j + lr * seg;
            int P;
            if (W == 16) P = (q >= read_len) ? SSW_BIAS : (((rd(q) == rc) && rc < 4) ? 2 + SSW_BIAS : 0);
            else P = (!use || q >= read_len) ? 0 : (((rd(q) == rc) && rc < 4) ? 2 : -2);
            if (W == 16) H = max(min(H + P, 255) - SSW_BIAS, 0);
            else H = min(H + P, 32767);
            const int e = (int)E[j * 16 + lr];
            H = max(max(H, e), F);
            cmax = max(cmax, H);
            h_store[j * 16 + lr] = (uint16_t)H;
            H = max(H - SSW_GAP_O, 0);
            E[j * 16 + lr] = (uint16_t)max(max(e - SSW_GAP_E, 0), H);
            F = max(max(F - SSW_GAP_E, 0), H);
            H = (int)h_load[j * 16 + lr];
        }
        // lazy F: carry F across the lane boundary until it can no longer raise any H (E is not refreshed)
        bool settled = false;
        for (int k = 0; k < W && !settled; ++k) {
            F = row_shr1(F);
            if (!use) F = 0;
            for (int j = 0; j < seg; ++j) {
                int h = max((int)h_store[j * 16 + lr], F);
                cmax = max(cmax, h);
                h_store[j * 16 + lr] = (uint16_t)h;
                h = max(h - SSW_GAP_O, 0);
                F = max(F - SSW_GAP_E, 0);
                if (!row_any(F > h, row)) { settled = true; break; }
            }
        }
        (void)hi_clamp;
        run_max = max(run_max, cmax);
        if (row_any(run_mark != run_max, row)) {
            run_mark = run_max;
            const int top = row_max(run_max);
            if (top > best) {
                best = top;
                if (W == 16 && best + SSW_BIAS >= 255) { overflow = true; break; }
                end_ref = i;
                for (int j = 0; j < seg; ++j) hbest[j * 16 + lr] = h_store[j * 16 + lr];
            }
        }
        const int cm = row_max(cmax);
        if (lr == 0) colmax[i] = (uint16_t)cm;
        if (cm == terminate) break;
    }
    __builtin_amdgcn_wave_barrier();
    // smallest read position whose best-column cell holds the best score (ssw.c:345-351)
    int end_read = read_len - 1;
    if (use)
        for (int j = 0; j < seg; ++j)
            if ((int)hbest[j * 16 + lr] == best) end_read = min(end_read, j + lr * seg);
    end_read = row_min(end_read);
    PassOut r{(W == 16 && (overflow || best + SSW_BIAS >= 255)) ? 255 : best, end_ref, end_read, 0, 0};
    // second best score outside the mask window around the end (ssw.c:353-378); first strictly larger wins
    // = largest value, smallest index.  The byte pass skips the column at `edge`, the word pass does not.
    int s2 = 0, r2 = 0x7fffffff;
    const int left = max(end_ref - mask_len, 0);
    const int right = min(end_ref + mask_len, ref_len) + (W == 16 ? 1 : 0);
    for (int i = lr; i < ref_len; i += 16) {
        if (i >= left && i < right) continue;
        const int v = (int)colmax[i];
        if (v > s2) { s2 = v; r2 = i; }
    }
    const int top2 = row_max(s2);
    r2 = row_min((s2 == top2 && top2 > 0) ? r2 : 0x7fffffff);
    r.score2 = top2;
    r.ref2 = top2 > 0 ? r2 : 0;
    return r;
}

// LDS per row for jobs of one size class
template <int QCAP, int RCAP>
struct RowMem {
    static constexpr int SEG = QCAP / 8;              // stripes of the 16-bit pass (the larger of the two)
    uint16_t hA[SEG * 16], hB[SEG * 16], E[SEG * 16], hbest[SEG * 16];
    uint16_t colmax[RCAP];
    uint8_t read[QCAP];                                // query codes
    uint8_t refc[RCAP];                                // reference codes: the motif repeated
};

// ROWS alignments per workgroup (16 lanes each): four share a wavefront.  Used for queries of up to 128 bases (a handful of
// stripes per column); longer ones go to ssw_wave.hip.
template <int QCAP, int RCAP, int ROWS>
__global__ __launch_bounds__(16 * ROWS) void ssw_passes_kernel(const uint8_t *__restrict__ ascii, int64_t length,
                                                        const uint8_t *__restrict__ motif_pool,
                                                        const int32_t *__restrict__ jobs /* 9 ints each */,
                                                        const int32_t *__restrict__ order, int n, int mask_len,
                                                        int32_t *__restrict__ out /* 8 ints per job */) {
    __shared__ RowMem<QCAP, RCAP> mem[ROWS];
    const int lane = threadIdx.x & 63, row = lane >> 4, lr = lane & 15;
    const int slot = (int)blockIdx.x * ROWS + row;
    if (slot >= n) return;
    const int job = order[slot];
    const int32_t *jb = jobs + 9 * (int64_t)job;      // RibbitAlignJob: seed_index, seed_type, motif_length, atomicity,
    const int atom = jb[3];                           //   query_start, query_length, ppr_length, small, motif_offset
    int qstart = jb[4], qlen = jb[5];
    const int rlen = jb[6];
    const uint8_t *motif = motif_pool + jb[8];
    // the host's slice(): a negative start clamps, the end clamps to the record
    if (qstart < 0) { qlen += qstart; qstart = 0; }
    if ((int64_t)qstart + qlen > length) qlen = (int)(length - qstart);
    RowMem<QCAP, RCAP> &m = mem[row];
    int32_t *o = out + 8 * (int64_t)job;
    if (qlen <= 0 || qlen > QCAP || rlen > RCAP || rlen <= 0 || atom <= 0) {      // not for this kernel: the host aligns it
        if (lr == 0) o[7] = -1;
        return;
    }
    for (int q = lr; q < qlen; q += 16) m.read[q] = (uint8_t)ssw_code(ascii[qstart + q]);
    for (int i = lr; i < rlen; i += 16) m.refc[i] = (uint8_t)ssw_code(motif[i % atom]);
    __builtin_amdgcn_wave_barrier();
    auto ref_at = [&](int i) { return (int)m.refc[i]; };
    auto read_fwd = [&](int q) { return (int)m.read[q]; };

    bool wide = false;
    PassOut fwd = striped_pass<16>(ref_at, 0, rlen, read_fwd, qlen, 255, mask_len, m.hA, m.hB, m.E, m.hbest, m.colmax, lr, row);
    if (fwd.score == 255) {
        fwd = striped_pass<8>(ref_at, 0, rlen, read_fwd, qlen, 0xffff, mask_len, m.hA, m.hB, m.E, m.hbest, m.colmax, lr, row);
        wide = true;
    }
    int score = fwd.score, ref_end = fwd.ref, query_end = fwd.read;
    int score2 = mask_len >= 15 ? fwd.score2 : 0, ref_end2 = mask_len >= 15 ? fwd.ref2 : -1;
    int ref_begin = -1, query_begin = -1, flag = 0;
    if (score == 0 || ref_end < 0) {
        ref_end = -1;
    } else {
        const int rq = query_end + 1;
        auto read_rev = [&](int q) { return (int)m.read[query_end - q]; };
        const PassOut rev = wide ? striped_pass<8>(ref_at, 1, ref_end + 1, read_rev, rq, score, mask_len, m.hA, m.hB, m.E, m.hbest, m.colmax, lr, row)
                                 : striped_pass<16>(ref_at, 1, ref_end + 1, read_rev, rq, score, mask_len, m.hA, m.hB, m.E, m.hbest, m.colmax, lr, row);
        ref_begin = rev.ref;
        query_begin = query_end - rev.read;
        if (score > rev.score) flag = 2;
    }
    if (lr == 0) {
        o[0] = score; o[1] = ref_end; o[2] = query_end; o[3] = score2; o[4] = ref_end2; o[5] = ref_begin; o[6] = query_begin; o[7] = flag;
    }
}

}  // namespace

void launch_ssw_passes(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs,
                       const int32_t *order_small, int n_small, const int32_t *order_big, int n_big, const int32_t *order_huge, int n_huge,
                       int mask_len, int32_t *out, hipStream_t stream, int huge_group_waves) {
    // queries of more than 128 bases: one wavefront per alignment, stripes spread over its lanes (ssw_wave.hip) -- or, for
    // the huge class, a workgroup per alignment (ssw_group.hip); the longest first
    if (huge_group_waves && ssw_group_fits(SSW_HUGE_Q, huge_group_waves))
        (void)launch_ssw_passes_group(ascii, length, motif_pool, jobs, order_huge, n_huge, mask_len, SSW_HUGE_Q, SSW_HUGE_R, huge_group_waves, out, stream);      // (the caller checks hipGetLastError)
    else
        launch_ssw_passes_wave(ascii, length, motif_pool, jobs, order_huge, n_huge, mask_len, SSW_HUGE_Q, SSW_HUGE_R, out, stream);
    launch_ssw_passes_wave(ascii, length, motif_pool, jobs, order_big, n_big, mask_len, SSW_BIG_Q, SSW_BIG_R, out, stream);
    if (n_small > 0)
        hipLaunchKernelGGL((ssw_passes_kernel<SSW_SMALL_Q, SSW_SMALL_R, 4>), dim3((unsigned)((n_small + 3) / 4)), dim3(64), 0, stream,
                           ascii, length, motif_pool, jobs, order_small, n_small, mask_len, out);
}

}  // namespace rb
