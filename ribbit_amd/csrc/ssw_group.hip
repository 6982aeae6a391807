// ssw_group.hip -- the two striped Smith-Waterman passes of a LONG alignment on a whole WORKGROUP of T wavefronts (gfx950).
// Reference: sw_sse2_byte (ssw.c:197-386), sw_sse2_word (:412-588), their orchestration (:843-891); same results as
// ssw_wave.hip / ssw_kernels.hip / ssw_exact.cpp (the library's stripe order), pinned against the reference library in
// tests/test_ssw_gpu.py.
//
// ssw_wave.hip gives an alignment one wavefront: 64 lanes = W register lanes x G = 64/W consecutive stripes, a column's
// stripes walked G at a time ("chunks").  A query of 4096 bases has 64 chunks per column, each a round trip through LDS, and
// one such alignment keeps its wavefront for 40-160 ms while its 50-60 KB of LDS keep all but three wavefronts off the CU: a
// record's long alignments (a few thousand) were the longest single step of its refinement.  Here the chunks of a column are
// dealt to the T wavefronts of a workgroup (T = 4, 8 or 16 by query length; wavefront w: chunks w*C .. w*C+C-1, C = ceil(chunks / T) <= 8, unrolled, so that a
// wavefront's LDS reads are in flight together), on the same LDS footprint:
//   1. every wavefront forms g, E and the prefix term b of its stripes (ssw_wave.hip's closed form of the main loop) and the
//      prefix maximum INSIDE its own chunks; its total per register lane goes to LDS;                          [barrier]
//   2. the totals of the wavefronts before it are its carry: F, H, E of its stripes are final and stored;          [barrier]
//   3. the lazy-F loop (ssw.c:283-301 / :499-514) is the first wavefront's alone, exactly as in ssw_wave.hip -- it leaves at
//      the first stripe where F can raise nothing, which is in the first chunk for almost every column;            [barrier]
//   4. the column maximum is the maximum of the wavefronts' partial maxima.  The library keeps a running maximum per register
//      lane and looks at its largest lane when any lane rose; that largest lane IS the running maximum of the column maxima,
//      so "this column's maximum exceeds the best so far" is the same test.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "kernels.h"

namespace rb {

namespace {

constexpr int BIAS = 2, GAP_O = 3, GAP_E = 1;
constexpr int NEG = -(1 << 28);
constexpr int MAXC = 8;           // chunks of a column per wavefront

__device__ __forceinline__ int gcode(uint8_t c) {      // kBaseTranslation (ssw_cpp.cpp:12-27)
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

template <int CTRL>
__device__ __forceinline__ int gdpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

template <int G>
__device__ __forceinline__ int lanes_max(int v) {           // over the G lanes of a group, in each of them
    v = max(v, gdpp<0xB1>(v));
    v = max(v, gdpp<0x4E>(v));
    if (G == 8) v = max(v, gdpp<0x141>(v));
    return v;
}
template <int G>
__device__ __forceinline__ int lanes_exclusive_max(int v, int jj) {      // lane jj: maximum of lanes 0..jj-1 of its group
    int x = v;
    int y = gdpp<0x111>(x); if (jj >= 1) x = max(x, y);
    y = gdpp<0x112>(x);     if (jj >= 2) x = max(x, y);
    if (G == 8) { y = gdpp<0x114>(x); if (jj >= 4) x = max(x, y); }
    y = gdpp<0x111>(x);
    return jj >= 1 ? y : NEG;
}
__device__ __forceinline__ int wavefront_max(int v) {       // uniform
    v = max(v, gdpp<0x128>(v)); v = max(v, gdpp<0x124>(v)); v = max(v, gdpp<0x122>(v)); v = max(v, gdpp<0x121>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

struct GroupOut { int score, ref, read, score2, ref2; };

// shared integer scratch of a workgroup: fs (lazy F), the wavefronts' carries and partial maxima, four reduction rows
template <int T>
struct Scratch {
    int fs[16];
    int carry[T * 16];
    int part[T];
    int red[4][T];
};

// maximum over the workgroup of a per-thread value; `row` must not be in use by another reduction still being read
template <int T>
__device__ __forceinline__ int group_wide_max(int v, int *row, int wv, int lane) {
    const int m = wavefront_max(v);
    if (lane == 0) row[wv] = m;
    __syncthreads();
    int r = row[0];
#pragma unroll
    for (int w = 1; w < T; ++w) r = max(r, row[w]);
    return r;
}

// One striped pass.  W = 16: sw_sse2_byte, W = 8: sw_sse2_word.  LDS as in ssw_wave.hip.  Called by all threads of the
// workgroup; control flow is uniform over the workgroup except where a wavefront index is tested.
template <int W, int T, typename RefAt, typename ReadAt>
__device__ __forceinline__ GroupOut group_pass(RefAt ref, int dir, int ref_len, ReadAt rd, int read_len, int terminate, int mask_len,
                                               uint16_t *hA, uint16_t *hB, uint16_t *E, uint16_t *hbest, uint16_t *colmax, Scratch<T> *sc,
                                               int wv, int lane) {
    constexpr int G = 64 / W;
    const int tid = wv * 64 + lane;
    const int l = lane / G, jj = lane % G;
    const int seg = (read_len + W - 1) / W;
    const int nchunk = (seg + G - 1) / G;
    const int cpw = (nchunk + T - 1) / T;                 // <= MAXC by the launch's size class
    for (int k = tid; k < seg * W; k += 64 * T) { hA[k] = 0; hB[k] = 0; E[k] = 0; hbest[k] = 0; }
    for (int i = tid; i < ref_len; i += 64 * T) colmax[i] = 0;
    __syncthreads();
    uint16_t *h_store = hA, *h_load = hB;
    int best = 0, end_ref = W == 16 ? -1 : 0;
    bool overflow = false;
    const int begin = dir ? ref_len - 1 : 0, stop = dir ? -1 : ref_len, step = dir ? -1 : 1;
    for (int i = begin; i != stop; i += step) {
        const int rc = ref(i);
        { uint16_t *t = h_store; h_store = h_load; h_load = t; }       // h_load: the previous column
        // ---- 1: this wavefront's stripes, prefix maximum inside its own chunks
        int g[MAXC], e[MAXC], before[MAXC];
        int T_own = NEG;
#pragma unroll
        for (int cc = 0; cc < MAXC; ++cc) {
            g[cc] = 0; e[cc] = 0; before[cc] = NEG;
            if (cc < cpw) {
                const int j = (wv * cpw + cc) * G + jj;
                int b = NEG;
                if (j < seg) {
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
                    e[cc] = (int)E[j * W + l];
                    g[cc] = max(hin, e[cc]);
                    b = max(g[cc] - GAP_O, 0) + (j + 1) * GAP_E;
                }
                before[cc] = max(T_own, lanes_exclusive_max<G>(b, jj));
                T_own = max(T_own, lanes_max<G>(b));
            }
        }
        if (jj == 0) sc->carry[wv * 16 + l] = T_own;
        __syncthreads();
        // ---- 2: the carry of the wavefronts before this one; F, H, E final
        int T_in = NEG, T_all = NEG;
#pragma unroll
        for (int w = 0; w < T; ++w) {
            const int t = sc->carry[w * 16 + l];
            if (w < wv) T_in = max(T_in, t);
            T_all = max(T_all, t);
        }
        int cm = 0;
#pragma unroll
        for (int cc = 0; cc < MAXC; ++cc) {
            if (cc < cpw) {
                const int j = (wv * cpw + cc) * G + jj;
                if (j < seg) {
                    const int F = max(max(T_in, before[cc]) - j * GAP_E, 0);
                    const int H = max(g[cc], F);
                    cm = max(cm, H);
                    h_store[j * W + l] = (uint16_t)H;
                    E[j * W + l] = (uint16_t)max(max(e[cc] - GAP_E, 0), max(H - GAP_O, 0));
                }
            }
        }
        __syncthreads();
        // ---- 3: lazy F, the first wavefront's (as ssw_wave.hip: at most W shifts, left as soon as F cannot raise any H)
        if (wv == 0) {
            int F_end = max(T_all - seg * GAP_E, 0);             // F after the last stripe, per register lane
            bool settled = false;
            for (int k = 0; k < W && !settled; ++k) {
                if (jj == 0) sc->fs[l] = F_end;
                __builtin_amdgcn_wave_barrier();
                const int F0 = l > 0 ? sc->fs[l - 1] : 0;
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
        }
        {
            const int m = wavefront_max(cm);
            if (lane == 0) sc->part[wv] = m;
        }
        __syncthreads();
        // ---- 4: the column's maximum (vMaxColumn's largest lane), the best column so far
        int cmw = sc->part[0];
#pragma unroll
        for (int w = 1; w < T; ++w) cmw = max(cmw, sc->part[w]);
        if (cmw > best) {
            best = cmw;
            if (W == 16 && best + BIAS >= 255) { overflow = true; break; }
            end_ref = i;
            for (int k = tid; k < seg * W; k += 64 * T) hbest[k] = h_store[k];
        }
        if (tid == 0) colmax[i] = (uint16_t)cmw;
        if (cmw == terminate) break;
    }
    __syncthreads();
    // smallest read position whose best-column cell holds the best score (ssw.c:345-351)
    int end_read = read_len - 1;
    for (int k = tid; k < seg * W; k += 64 * T)
        if ((int)hbest[k] == best) end_read = min(end_read, (k / W) + (k % W) * seg);
    end_read = -group_wide_max<T>(-end_read, sc->red[0], wv, lane);
    GroupOut r{(W == 16 && (overflow || best + BIAS >= 255)) ? 255 : best, end_ref, end_read, 0, 0};
    // second best outside the mask window (ssw.c:353-378): largest value, smallest index; the byte pass skips the column
    // at `edge`, the word pass does not
    int s2 = 0, r2 = 0x7fffffff;
    const int left = max(end_ref - mask_len, 0);
    const int right = min(end_ref + mask_len, ref_len) + (W == 16 ? 1 : 0);
    for (int i = tid; i < ref_len; i += 64 * T) {
        if (i >= left && i < right) continue;
        const int v = (int)colmax[i];
        if (v > s2) { s2 = v; r2 = i; }
    }
    const int top2 = group_wide_max<T>(s2, sc->red[1], wv, lane);
    r2 = -group_wide_max<T>(-((s2 == top2 && top2 > 0) ? r2 : 0x7fffffff), sc->red[2], wv, lane);
    r.score2 = top2;
    r.ref2 = top2 > 0 ? r2 : 0;
    __syncthreads();            // the next pass clears the arrays these reductions read
    return r;
}

// One workgroup of T wavefronts per alignment.  Dynamic LDS, sized by the launch for its class: as ssw_wave.hip, with the
// workgroup's integer scratch in place of its sixteen ints.
template <int T>
__global__ __launch_bounds__(64 * T) void ssw_passes_group_kernel(const uint8_t *__restrict__ ascii, int64_t length, const uint8_t *__restrict__ motif_pool,
                                                                  const int32_t *__restrict__ jobs /* 9 ints each */, const int32_t *__restrict__ order,
                                                                  int n, int mask_len, int qcap, int rcap, int32_t *__restrict__ out /* 8 ints per job */) {
    extern __shared__ uint16_t glds16[];
    const int slot = (int)blockIdx.x;
    if (slot >= n) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
        if (tid == 0) o[7] = -1;
        return;
    }
    const int cells = qcap + 16;
    uint16_t *hA = glds16, *hB = hA + cells, *E = hB + cells, *hbest = E + cells, *colmax = hbest + cells;
    Scratch<T> *sc = (Scratch<T> *)(colmax + ((rcap + 1) & ~1));
    uint8_t *read = (uint8_t *)(sc + 1), *refc = read + qcap;
    for (int q = tid; q < qlen; q += 64 * T) read[q] = (uint8_t)gcode(ascii[qstart + q]);
    for (int i = tid; i < rlen; i += 64 * T) refc[i] = (uint8_t)gcode(motif[i % atom]);
    __syncthreads();
    auto ref_at = [&](int i) { return (int)refc[i]; };
    auto read_fwd = [&](int q) { return (int)read[q]; };

    bool wide = false;
    GroupOut fwd = group_pass<16, T>(ref_at, 0, rlen, read_fwd, qlen, 255, mask_len, hA, hB, E, hbest, colmax, sc, wv, lane);
    if (fwd.score == 255) {
        fwd = group_pass<8, T>(ref_at, 0, rlen, read_fwd, qlen, 0xffff, mask_len, hA, hB, E, hbest, colmax, sc, wv, lane);
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
        const GroupOut rev = wide ? group_pass<8, T>(ref_at, 1, ref_end + 1, read_rev, rq, score, mask_len, hA, hB, E, hbest, colmax, sc, wv, lane)
                                  : group_pass<16, T>(ref_at, 1, ref_end + 1, read_rev, rq, score, mask_len, hA, hB, E, hbest, colmax, sc, wv, lane);
        ref_begin = rev.ref;
        query_begin = query_end - rev.read;
        if (score > rev.score) flag = 2;
    }
    if (tid == 0) {
        o[0] = score; o[1] = ref_end; o[2] = query_end; o[3] = score2; o[4] = ref_end2; o[5] = ref_begin; o[6] = query_begin; o[7] = flag;
    }
}

template <int T>
size_t group_lds_bytes(int qcap, int rcap) {
    return (size_t)(4 * (qcap + 16) + ((rcap + 1) & ~1)) * sizeof(uint16_t) + sizeof(Scratch<T>) + (size_t)qcap + (size_t)rcap;
}

}  // namespace

// the chunks of the longest query of a class must fit the workgroup: ceil(ceil(q / 8) / 8) <= T * MAXC covers both passes
// (byte: q/16 stripes in chunks of 4; word: q/8 stripes in chunks of 8)
bool ssw_group_fits(int qcap, int waves) { return ((qcap + 7) / 8 + 7) / 8 <= waves * MAXC; }

hipError_t launch_ssw_passes_group(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs, const int32_t *order, int n,
                                   int mask_len, int qcap, int rcap, int waves, int32_t *out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (waves == 16) {
        // the largest class needs more LDS than a workgroup gets without asking (64 KB); a CU has 160 KB
        const size_t lds = group_lds_bytes<16>(qcap, rcap);
        const hipError_t e = hipFuncSetAttribute((const void *)ssw_passes_group_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ssw_passes_group_kernel<16>, dim3((unsigned)n), dim3(1024), lds, stream, ascii, length, motif_pool, jobs,
                           order, n, mask_len, qcap, rcap, out);
    } else if (waves == 8)
        hipLaunchKernelGGL(ssw_passes_group_kernel<8>, dim3((unsigned)n), dim3(512), group_lds_bytes<8>(qcap, rcap), stream, ascii, length, motif_pool, jobs,
                           order, n, mask_len, qcap, rcap, out);
    else
        hipLaunchKernelGGL(ssw_passes_group_kernel<4>, dim3((unsigned)n), dim3(256), group_lds_bytes<4>(qcap, rcap), stream, ascii, length, motif_pool, jobs,
                           order, n, mask_len, qcap, rcap, out);
    return hipGetLastError();
}

}  // namespace rb
