// ssw_wave.hip -- the two striped Smith-Waterman passes of a LONG alignment on a whole wavefront (gfx950).
// Reference: sw_sse2_byte (ssw.c:197-386), sw_sse2_word (:412-588), their orchestration (:843-891); same results as
// ssw_kernels.hip / ssw_exact.cpp (the library's stripe order), pinned against the reference library in
// tests/test_ssw_gpu.py.
//
// (Queries of 129..4096 bases, in three classes by LDS need.  Since the end of round 3 the product uses this kernel for the
// first class only, 129..512 bases; longer queries run a workgroup per alignment, ssw_group.hip -- and this kernel only if the
// device will not give a workgroup the LDS it asks for.)
// ssw_kernels.hip gives an alignment one DPP row of 16 lanes (= one SSE2 register) and walks the stripes j = 0..segLen-1
// one after the other: right for queries of a few dozen bases, but a 1000-base query has 125 stripes per column and
// every step waits for LDS.  Here the 64 lanes are W register lanes x G = 64/W consecutive stripes (lane = l*G + jj), and
// both inner loops of a column are evaluated G stripes at a time from closed forms that are exact in integers:
//   main loop.   H_j = max(g_j, F_j) with g_j = max(sat(Hdiag_j + P_j), E_j) known from the previous column, and
//                F_{j+1} = max(F_j - e, H_j - o, 0) = max(F_j - e, g_j - o, 0) because o >= e.  So
//                F_j = max(0, max_{s<j} (a_s + (s+1) e) - j e),  a_s = max(g_s - o, 0): a prefix maximum along j
//                (three DPP steps inside a group of G lanes, one carried maximum per register lane across groups).
//   lazy F.      With F shifted in from the lane below, step j sees F0 - j e (clamped at 0) whatever happened before, so
//                every step's "can F still raise any H?" test is known at once; the loop's exit is the first stripe,
//                in order, where no register lane says yes (a ballot folded over the register lanes, then ctz), and the
//                stripes up to and including it are updated.
// Nothing else in a column depends on the order of the stripes.  Saturation is the library's: the byte pass clamps at
// 255 and carries the bias, the word pass clamps at 32767.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "kernels.h"

namespace rb {

namespace {

constexpr int BIAS = 2, GAP_O = 3, GAP_E = 1;
constexpr int NEG = -(1 << 28);

__device__ __forceinline__ int wcode(uint8_t c) {      // kBaseTranslation (ssw_cpp.cpp:12-27)
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

// maximum over the G lanes of a group (G = 4: a quad; G = 8: half a DPP row), in each of them
template <int G>
__device__ __forceinline__ int group_max(int v) {
    v = max(v, dpp<0xB1>(v));                       // quad_perm [1,0,3,2]
    v = max(v, dpp<0x4E>(v));                       // quad_perm [2,3,0,1]
    if (G == 8) v = max(v, dpp<0x141>(v));          // row_half_mirror: the other quad of the half row
    return v;
}
// exclusive prefix maximum along the G lanes of a group (lane jj receives the maximum of lanes 0..jj-1, NEG for jj = 0)
template <int G>
__device__ __forceinline__ int group_exclusive_max(int v, int jj) {
    int x = v;
    int y = dpp<0x111>(x); if (jj >= 1) x = max(x, y);      // row_shr:1
    y = dpp<0x112>(x);     if (jj >= 2) x = max(x, y);      // row_shr:2
    if (G == 8) { y = dpp<0x114>(x); if (jj >= 4) x = max(x, y); }
    y = dpp<0x111>(x);
    return jj >= 1 ? y : NEG;
}
__device__ __forceinline__ int wave_max(int v) {           // uniform
    v = max(v, dpp<0x128>(v)); v = max(v, dpp<0x124>(v)); v = max(v, dpp<0x122>(v)); v = max(v, dpp<0x121>(v));   // row_ror 8,4,2,1
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_min(int v) { return -wave_max(-v); }

struct WaveOut { int score, ref, read, score2, ref2; };

// One striped pass.  W = 16: sw_sse2_byte, W = 8: sw_sse2_word.  LDS: hA, hB, E, hbest [seg][W] uint16, colmax [ref_len]
// uint16, fs [16] int.  Called by all 64 lanes; control flow is wave-uniform.
template <int W, typename RefAt, typename ReadAt>
__device__ __forceinline__ WaveOut wave_pass(RefAt ref, int dir, int ref_len, ReadAt rd, int read_len, int terminate, int mask_len,
                                             uint16_t *hA, uint16_t *hB, uint16_t *E, uint16_t *hbest, uint16_t *colmax, int *fs, int lane) {
    constexpr int G = 64 / W;
    const int l = lane / G, jj = lane % G;
    const int seg = (read_len + W - 1) / W;
    const int nchunk = (seg + G - 1) / G;
    for (int k = lane; k < seg * W; k += 64) { hA[k] = 0; hB[k] = 0; E[k] = 0; hbest[k] = 0; }
    for (int i = lane; i < ref_len; i += 64) colmax[i] = 0;
    __builtin_amdgcn_wave_barrier();
    uint16_t *h_store = hA, *h_load = hB;
    int run_max = 0, run_mark = 0, best = 0, end_ref = W == 16 ? -1 : 0;
    bool overflow = false;
    const int begin = dir ? ref_len - 1 : 0, stop = dir ? -1 : ref_len, step = dir ? -1 : 1;
    for (int i = begin; i != stop; i += step) {
        const int rc = ref(i);
        { uint16_t *t = h_store; h_store = h_load; h_load = t; }       // h_load: the previous column
        int T = NEG, cm = 0;
        for (int c = 0; c < nchunk; ++c) {
            const int j = c * G + jj;
            const bool valid = j < seg;
            int g = 0, e = 0, b = NEG;
            if (valid) {
                // the cell diagonally above: stripe j-1 of the same register lane, or the last stripe of the lane below
                // (_mm_slli_si128 of the previous column's last vector; zero for lane 0)
                const int hd = j > 0 ? (int)h_load[(j - 1) * W + l] : (l > 0 ? (int)h_load[(seg - 1) * W + (l - 1)] : 0);
                const int q = j + l * seg;
                int hin;
                if (W == 16) {
                    const int P = (q >= read_len) ? BIAS : (((rd(q) == rc) && rc < 4) ? 2 + BIAS : 0);
                    hin = max(min(hd + P, 255) - BIAS, 0);
                } else {
                    const int P = (q >= read_len) ? 0 : (((rd(q) == rc) && rc < 4) ? 2 : -2);
                    hin = min(hd + P, 32767);
                }
                e = (int)E[j * W + l];
                g = max(hin, e);
                b = max(g - GAP_O, 0) + (j + 1) * GAP_E;
            }
            const int before = max(T, group_exclusive_max<G>(b, jj));
            const int F = max(before - j * GAP_E, 0);
            const int H = max(g, F);
            if (valid) {
                cm = max(cm, H);
                h_store[j * W + l] = (uint16_t)H;
                E[j * W + l] = (uint16_t)max(max(e - GAP_E, 0), max(H - GAP_O, 0));
            }
            T = max(T, group_max<G>(b));
        }
        int F_end = max(T - seg * GAP_E, 0);             // F after the last stripe, per register lane
        __builtin_amdgcn_wave_barrier();
        // lazy F (ssw.c:283-301 / :499-514): at most W shifts, left as soon as F cannot raise any H
        bool settled = false;
        for (int k = 0; k < W && !settled; ++k) {
            if (jj == 0) fs[l] = F_end;
            __builtin_amdgcn_wave_barrier();
            const int F0 = l > 0 ? fs[l - 1] : 0;
            __builtin_amdgcn_wave_barrier();
            for (int c = 0; c < nchunk; ++c) {
                const int j = c * G + jj;
                const bool valid = j < seg;
                int h = 0;
                bool more = false;
                if (valid) {
                    h = max((int)h_store[j * W + l], max(F0 - j * GAP_E, 0));
                    more = max(F0 - (j + 1) * GAP_E, 0) > max(h - GAP_O, 0);
                }
                unsigned long long m = __ballot(more);
                if (G == 8) { m |= m >> 32; m |= m >> 16; m |= m >> 8; }
                else { m |= m >> 32; m |= m >> 16; m |= m >> 8; m |= m >> 4; }
                const int in_chunk = min(G, seg - c * G);
                const unsigned live = (1u << in_chunk) - 1u;
                const unsigned quiet = ~(unsigned)m & live;                  // stripes (in order) where no lane can go on
                const int last = quiet ? (int)__builtin_ctz(quiet) : G;      // the loop leaves after this stripe
                if (valid && jj <= last) { cm = max(cm, h); h_store[j * W + l] = (uint16_t)h; }
                if (quiet) { settled = true; break; }
            }
            F_end = max(F0 - seg * GAP_E, 0);
            __builtin_amdgcn_wave_barrier();
        }
        const int cml = group_max<G>(cm);                 // vMaxColumn, per register lane
        run_max = max(run_max, cml);
        if (__ballot(run_mark != run_max) != 0ull) {
            run_mark = run_max;
            const int top = wave_max(run_max);
            if (top > best) {
                best = top;
                if (W == 16 && best + BIAS >= 255) { overflow = true; break; }
                end_ref = i;
                for (int k = lane; k < seg * W; k += 64) hbest[k] = h_store[k];
            }
        }
        const int cmw = wave_max(cml);
        if (lane == 0) colmax[i] = (uint16_t)cmw;
        if (cmw == terminate) break;
    }
    __builtin_amdgcn_wave_barrier();
    // smallest read position whose best-column cell holds the best score (ssw.c:345-351)
    int end_read = read_len - 1;
    for (int k = lane; k < seg * W; k += 64)
        if ((int)hbest[k] == best) end_read = min(end_read, (k / W) + (k % W) * seg);
    end_read = wave_min(end_read);
    WaveOut r{(W == 16 && (overflow || best + BIAS >= 255)) ? 255 : best, end_ref, end_read, 0, 0};
    // second best outside the mask window (ssw.c:353-378): largest value, smallest index; the byte pass skips the column
    // at `edge`, the word pass does not
    int s2 = 0, r2 = 0x7fffffff;
    const int left = max(end_ref - mask_len, 0);
    const int right = min(end_ref + mask_len, ref_len) + (W == 16 ? 1 : 0);
    for (int i = lane; i < ref_len; i += 64) {
        if (i >= left && i < right) continue;
        const int v = (int)colmax[i];
        if (v > s2) { s2 = v; r2 = i; }
    }
    const int top2 = wave_max(s2);
    r2 = wave_min((s2 == top2 && top2 > 0) ? r2 : 0x7fffffff);
    r.score2 = top2;
    r.ref2 = top2 > 0 ? r2 : 0;
    return r;
}

}  // namespace

// One wavefront per alignment.  Dynamic LDS, sized by the launch for its class: 4 x (qcap + 16) + rcap uint16, 16 int,
// qcap + rcap bytes.
__global__ __launch_bounds__(64) void ssw_passes_wave_kernel(const uint8_t *__restrict__ ascii, int64_t length, const uint8_t *__restrict__ motif_pool,
                                                             const int32_t *__restrict__ jobs /* 9 ints each */, const int32_t *__restrict__ order,
                                                             int n, int mask_len, int qcap, int rcap, int32_t *__restrict__ out /* 8 ints per job */) {
    extern __shared__ uint16_t lds16[];
    const int slot = (int)blockIdx.x;
    if (slot >= n) return;
    const int lane = (int)threadIdx.x;
    const int job = order[slot];
    const int32_t *jb = jobs + 9 * (int64_t)job;
    const int atom = jb[3];
    int qstart = jb[4], qlen = jb[5];
    const int rlen = jb[6];
    const uint8_t *motif = motif_pool + jb[8];
    if (qstart < 0) { qlen += qstart; qstart = 0; }                       // the host's slice(): a negative start clamps,
    if ((int64_t)qstart + qlen > length) qlen = (int)(length - qstart);   // the end clamps to the record
    int32_t *o = out + 8 * (int64_t)job;
    if (qlen <= 0 || qlen > qcap || rlen > rcap || rlen <= 0 || atom <= 0) {      // not for this launch: the host aligns it
        if (lane == 0) o[7] = -1;
        return;
    }
    const int cells = qcap + 16;
    uint16_t *hA = lds16, *hB = hA + cells, *E = hB + cells, *hbest = E + cells, *colmax = hbest + cells;
    int *fs = (int *)(colmax + ((rcap + 1) & ~1));
    uint8_t *read = (uint8_t *)(fs + 16), *refc = read + qcap;
    for (int q = lane; q < qlen; q += 64) read[q] = (uint8_t)wcode(ascii[qstart + q]);
    for (int i = lane; i < rlen; i += 64) refc[i] = (uint8_t)wcode(motif[i % atom]);
    __builtin_amdgcn_wave_barrier();
    auto ref_at = [&](int i) { return (int)refc[i]; };
    auto read_fwd = [&](int q) { return (int)read[q]; };

    bool wide = false;
    WaveOut fwd = wave_pass<16>(ref_at, 0, rlen, read_fwd, qlen, 255, mask_len, hA, hB, E, hbest, colmax, fs, lane);
    if (fwd.score == 255) {
        fwd = wave_pass<8>(ref_at, 0, rlen, read_fwd, qlen, 0xffff, mask_len, hA, hB, E, hbest, colmax, fs, lane);
        wide = true;
    }
    int score = fwd.score, ref_end = fwd.ref, query_end = fwd.read;
    int score2 = mask_len >= 15 ? fwd.score2 : 0, ref_end2 = mask_len >= 15 ? fwd.ref2 : -1;
    int ref_begin = -1, query_begin = -1, flag = 0;
    if (score == 0 || ref_end < 0) {
        ref_end = -1;
    } else {
        const int rq = query_end + 1;
        auto read_rev = [&](int q) { return (int)read[query_end - q]; };
        const WaveOut rev = wide ? wave_pass<8>(ref_at, 1, ref_end + 1, read_rev, rq, score, mask_len, hA, hB, E, hbest, colmax, fs, lane)
                                 : wave_pass<16>(ref_at, 1, ref_end + 1, read_rev, rq, score, mask_len, hA, hB, E, hbest, colmax, fs, lane);
        ref_begin = rev.ref;
        query_begin = query_end - rev.read;
        if (score > rev.score) flag = 2;
    }
    if (lane == 0) {
        o[0] = score; o[1] = ref_end; o[2] = query_end; o[3] = score2; o[4] = ref_end2; o[5] = ref_begin; o[6] = query_begin; o[7] = flag;
    }
}

size_t ssw_wave_lds_bytes(int qcap, int rcap) {
    return (size_t)(4 * (qcap + 16) + ((rcap + 1) & ~1)) * sizeof(uint16_t) + 16 * sizeof(int) + (size_t)qcap + (size_t)rcap;
}

void launch_ssw_passes_wave(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs, const int32_t *order, int n,
                            int mask_len, int qcap, int rcap, int32_t *out, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(ssw_passes_wave_kernel, dim3((unsigned)n), dim3(64), ssw_wave_lds_bytes(qcap, rcap), stream, ascii, length, motif_pool, jobs,
                       order, n, mask_len, qcap, rcap, out);
}

}  // namespace rb
