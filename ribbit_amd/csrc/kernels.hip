// kernels.hip -- hand-written CDNA4 (gfx950) kernels for ribbit's shift-XOR scan.
//
// Integer/bit work only: no MFMA.  The binding resource is 32-bit VALU issue, so the design
// minimises VALU ops per (32-base word, motif):
//   * a wavefront owns a tile of 64 lanes x WORDS_PER_LANE consecutive words; every lane keeps
//     its words of the hi/lo/brk planes in registers for the whole motif loop;
//   * the shifted operand for shift s = 32q + r is one v_alignbit per word from a second
//     register set holding the words at offset q, reloaded from LDS only when q changes
//     (once per 32 shifts), so the sweep X_s = ~(H ^ H>>s) & ~(L ^ L>>s) of
//     fasta_utils.cpp:117-122 costs 5 VALU ops per word and no memory traffic;
//   * runs are found with OR-doubling funnel shifts inside a lane (neighbour words are in the
//     same lane; one halo word each side), and only the rare qualified run STARTs and ENDs are
//     written out, compacted in position order with a wave prefix sum.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <type_traits>

#include "kernels.h"

namespace rb {

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh) {
    // ({hi,lo} >> (sh & 31))[31:0]  -> v_alignbit_b32
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// v_bitop3_b32: any boolean function of three words in one full-rate instruction.  TT = the function applied to
// the constants 0xF0, 0xCC, 0xAA in place of a, b, c (e.g. (a ^ b) | c  ->  (0xF0 ^ 0xCC) | 0xAA = 0xBE).
template <unsigned TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}

// ---------------------------------------------------------------------------------- pack
// fasta_utils.cpp:90-115: A/a 00, C/c 01, G/g 10, T/t 11, anything else -> N (code 00).
// One thread produces one 32-base word of each plane.
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ ascii, int64_t length,
                                                   uint32_t *__restrict__ hi, uint32_t *__restrict__ lo,
                                                   uint32_t *__restrict__ brk, int64_t total_words,
                                                   uint32_t *__restrict__ zero_words, int n_zero) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the event counters of the scan that follows, zeroed here to save that scan a fill launch
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0u;
    if (t >= total_words) return;
    const int64_t p0 = (t - LEAD_WORDS) * 32;
    uint32_t h = 0, l = 0, b = 0xffffffffu;
    if (p0 >= 0 && p0 < length) {
        uint32_t w[8];
        const int64_t left = length - p0;
        const uint8_t *src = ascii + p0;
        if (left >= 32 && (((uintptr_t)src) & 15) == 0) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(src);
            const uint4 v1 = *reinterpret_cast<const uint4 *>(src + 16);
            w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
            w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t x = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int idx = 4 * i + k;
                    const uint32_t c = (idx < left) ? (uint32_t)src[idx] : (uint32_t)'N';
                    x |= c << (8 * k);
                }
                w[i] = x;
            }
        }
        // Four bases per 32-bit word, all byte lanes at once (fasta_utils.cpp:94-114: A 00, C 01, G 10, T 11, any
        // case; everything else is N).  With b1, b2 = bits 1 and 2 of the character: code = (b2, b1 ^ b2).  The
        // character is valid iff, lower-cased, it equals 'a' + 2*(lo|hi) + 4*hi + 13*(lo&hi) -- 'a','c','g','t'.
        // v_dot4_u32_u8 with weights 1,2,4,8 then gathers the four flag bytes of a word into a nibble.
        b = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t x = w[i];
            const uint32_t b2 = (x >> 2) & 0x01010101u;
            const uint32_t lo1 = ((x >> 1) & 0x01010101u) ^ b2;
            const uint32_t both = lo1 & b2, either = lo1 | b2;
            uint32_t expect = (either << 1) + 0x61616161u;
            expect = (b2 << 2) + expect;
            expect = (both << 3) + expect;
            expect = (both << 2) + expect;
            expect = both + expect;
            const uint32_t diff = (x | 0x20202020u) ^ expect;
            // bit 7 of every byte of nz = that byte of diff is non-zero (no carries between bytes)
            const uint32_t nz = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;
            const uint32_t bad = nz >> 7;                       // 0 / 1 per byte
            const uint32_t ok = bad ^ 0x01010101u;
            const uint32_t weights = 0x08040201u;
            h |= __builtin_amdgcn_udot4(b2 & ok, weights, 0u, false) << (4 * i);
            l |= __builtin_amdgcn_udot4(lo1 & ok, weights, 0u, false) << (4 * i);
            b |= __builtin_amdgcn_udot4(bad, weights, 0u, false) << (4 * i);
        }
        if (left < 32) b |= 0xffffffffu << (uint32_t)left;   // positions >= L break every run
    }
    hi[t] = h; lo[t] = l; brk[t] = b;
}

void launch_pack(const uint8_t *dev_ascii, int64_t length, uint32_t *hi, uint32_t *lo, uint32_t *brk,
                 int64_t total_words, uint32_t *zero_words, int n_zero, hipStream_t stream) {
    const int threads = 256;
    const int64_t blocks = (total_words + threads - 1) / threads;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, dev_ascii, length, hi, lo, brk,
                       total_words, zero_words, n_zero);
}

// ---------------------------------------------------------------------- event staging
// Events are compacted in position order inside a wave (DPP prefix sum, no LDS traffic), staged
// in a per-wave LDS buffer across the motif loop and flushed with ONE global atomic per flush.
// The atomic goes to one of EV_SHARDS counters (each on its own 128-byte line, each owning a
// fixed region of the event buffer): a single counter sustains only ~88 returning atomics/us
// (MI355X_MICROARCH.md "dequeue"), which bounded the first version of this kernel at 4.8 ms.
constexpr int EV_STAGE = 256;        // events staged per wave in LDS (2 KiB)

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    // Hillis-Steele inside each row of 16 lanes, then two row broadcasts (GFX9 DPP)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

struct EventSink {
    uint64_t *events;        // global buffer, EV_SHARDS regions of region_cap events
    uint32_t *counters;      // EV_SHARDS counters, EV_COUNTER_STRIDE words apart
    uint32_t region_cap;
    uint32_t shard;
};

__device__ __forceinline__ void sink_flush(const EventSink &sk, volatile uint64_t *stage, int &staged, int lane) {
    if (staged == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&sk.counters[sk.shard * EV_COUNTER_STRIDE], (uint32_t)staged);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    uint64_t *dst = sk.events + (size_t)sk.shard * sk.region_cap;
    for (int i = lane; i < staged; i += 64) {
        const uint32_t idx = base + (uint32_t)i;
        if (idx < sk.region_cap) dst[idx] = stage[i];
    }
    __builtin_amdgcn_wave_barrier();
    staged = 0;
}

// Compact one motif's START (S) / END (E) bitmaps of this wave, in position order, into the
// wave's LDS stage (or straight to global memory when there are more than EV_STAGE of them).
// end_kind(k, bit, pos) classifies an END event.  Called wave-uniformly.
template <typename EndKind>
__device__ __forceinline__ void stage_events(const uint32_t (&S)[WORDS_PER_LANE], const uint32_t (&E)[WORDS_PER_LANE],
                                             uint32_t word0, uint32_t mlen, const EventSink &sink,
                                             volatile uint64_t *stage, int &staged, int lane, EndKind end_kind) {
    // The caller reaches this on a rarely taken, wave-uniform branch.  hipcc otherwise speculates the
    // popcounts and the DPP scan below into the caller's hot loop (45 of its 217 VALU ops); reading
    // the bitmaps through an empty asm makes them opaque at this point and keeps the work here.
    uint32_t Sv[WORDS_PER_LANE], Ev[WORDS_PER_LANE];
#pragma unroll
    for (int k = 0; k < WORDS_PER_LANE; k++) {
        Sv[k] = S[k]; Ev[k] = E[k];
        asm volatile("" : "+v"(Sv[k]), "+v"(Ev[k]));
    }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < WORDS_PER_LANE; k++) cnt += __popc(Sv[k]) + __popc(Ev[k]);
    const int incl = wave_inclusive_scan(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (staged + total > EV_STAGE) sink_flush(sink, stage, staged, lane);
    const bool direct = total > EV_STAGE;
    uint32_t idx = (uint32_t)(incl - cnt);
    uint64_t *gdst = nullptr;
    if (direct) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&sink.counters[sink.shard * EV_COUNTER_STRIDE], (uint32_t)total);
        idx += (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        gdst = sink.events + (size_t)sink.shard * sink.region_cap;
    } else {
        idx += (uint32_t)staged;
    }
#pragma unroll
    for (int k = 0; k < WORDS_PER_LANE; k++) {
        uint32_t both = Sv[k] | Ev[k];
        while (both) {
            const uint32_t b = (uint32_t)__builtin_ctz(both);
            both &= both - 1u;
            const uint32_t pos = ((word0 + (uint32_t)k) << 5) + b;
            const uint32_t kind = ((Sv[k] >> b) & 1u) ? (uint32_t)EV_START : end_kind(k, b, pos);
            const uint64_t e = ev_pack(pos, mlen, kind);
            if (direct) { if (idx < sink.region_cap) gdst[idx] = e; }
            else stage[idx] = e;
            idx++;
        }
    }
    if (!direct) staged += total;
    __builtin_amdgcn_wave_barrier();
}

constexpr int K = WORDS_PER_LANE;

// The same compaction when events are SPARSE (the perfect scan: where a (tile, motif) pair or a flush of the candidate queue
// has events at all, a dozen lanes hold one or two each).  stage_events walks the set bits of a lane's words IN that lane --
// eight serialised loops with two or three lanes alive in each: a quarter of the perfect kernel's instructions and 29 % of its
// time (rounds 2-3).  Here the lanes that hold events park their words in LDS, COOP_SRC lanes a round, and all 64
// lanes expand them: lane (r, j) takes words 2j and 2j + 1 of parked lane r, so the two short loops below run with most lanes
// that have anything to do doing it at once.  Where an event goes is settled before any is written -- one scan over the
// lanes' event counts, as in stage_events -- so a pair's events stay in one piece whatever the number of rounds.
// word0 / mlen / brk_at may differ from lane to lane (candidates of several motifs, see scan_perfect_kernel): first own word,
// motif, and the LDS index in s_brk of the lane's word k = 0 (what closed a run: an N, the end of the record, a mismatch).
// above RB_COOP_MAX_LANES lanes with events the lanes expand their own words (stage_events): nothing is gained by parking most
// of a wave; at or below COOP_SRC one round does it and the parked words need not stay in registers
#ifndef RB_COOP_MAX_LANES
#define RB_COOP_MAX_LANES 16
#endif
constexpr int COOP_SRC = 16;
struct CoopScratch { uint32_t w[COOP_SRC][20]; };      // per parked lane: S[0..7], E[0..7], word0, mlen, brk_at, first event's place

__device__ __forceinline__ void stage_events_coop(const uint32_t (&S)[WORDS_PER_LANE], const uint32_t (&E)[WORDS_PER_LANE], uint32_t word0, uint32_t mlen,
                                                  uint32_t brk_at, const uint32_t *s_brk, uint32_t length, CoopScratch &sc, const EventSink &sink,
                                                  volatile uint64_t *stage, int &staged, int lane) {
    uint32_t Sv[WORDS_PER_LANE], Ev[WORDS_PER_LANE];
#pragma unroll
    for (int k = 0; k < WORDS_PER_LANE; k++) {
        Sv[k] = S[k]; Ev[k] = E[k];
        asm volatile("" : "+v"(Sv[k]), "+v"(Ev[k]));      // (keeps this block out of the caller's hot loop, as in stage_events)
    }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < WORDS_PER_LANE; k++) cnt += __popc(Sv[k] | Ev[k]);      // a position is a START or an END, never both
    const int incl = wave_inclusive_scan(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (staged + total > EV_STAGE) sink_flush(sink, stage, staged, lane);
    const bool direct = total > EV_STAGE;
    uint32_t first = (uint32_t)(incl - cnt);
    uint64_t *gdst = nullptr;
    if (direct) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&sink.counters[sink.shard * EV_COUNTER_STRIDE], (uint32_t)total);
        first += (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        gdst = sink.events + (size_t)sink.shard * sink.region_cap;
    } else {
        first += (uint32_t)staged;
    }
    unsigned long long todo = __ballot(cnt != 0);
    const int r = lane >> 2, j = lane & 3;
    do {                                                    // rounds of COOP_SRC parked lanes (wave-uniform; one round if the caller's limit is COOP_SRC)
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
        const bool mine = ((todo >> lane) & 1ull) != 0ull && rank < COOP_SRC;
        if (mine) {
            uint32_t *row = sc.w[rank];
#pragma unroll
            for (int k = 0; k < WORDS_PER_LANE; k++) { row[k] = Sv[k]; row[WORDS_PER_LANE + k] = Ev[k]; }
            row[16] = word0; row[17] = mlen; row[18] = brk_at; row[19] = first;
        }
        const unsigned long long handled = __ballot(mine);
        const int nsrc = __popcll(handled);
        todo &= ~handled;
        __builtin_amdgcn_wave_barrier();
        uint32_t s0 = 0, s1 = 0, e0 = 0, e1 = 0;
        if (r < nsrc) { s0 = sc.w[r][2 * j]; s1 = sc.w[r][2 * j + 1]; e0 = sc.w[r][WORDS_PER_LANE + 2 * j]; e1 = sc.w[r][WORDS_PER_LANE + 2 * j + 1]; }
        const int c = __popc(s0 | e0) + __popc(s1 | e1);
        // events of the parked lane's earlier words: the counts of the quad's lower lanes
        const int q0 = __builtin_amdgcn_update_dpp(0, c, 0x00, 0xf, 0xf, true);      // quad_perm:[0,0,0,0]
        const int q1 = __builtin_amdgcn_update_dpp(0, c, 0x55, 0xf, 0xf, true);      // quad_perm:[1,1,1,1]
        const int q2 = __builtin_amdgcn_update_dpp(0, c, 0xaa, 0xf, 0xf, true);      // quad_perm:[2,2,2,2]
        if (c != 0) {
            const uint32_t w0 = sc.w[r][16], ml = sc.w[r][17], bb = sc.w[r][18];
            uint32_t idx = sc.w[r][19] + (uint32_t)((j > 0 ? q0 : 0) + (j > 1 ? q1 : 0) + (j > 2 ? q2 : 0));
            auto expand = [&](uint32_t sw, uint32_t ew, uint32_t k) {
                uint32_t both = sw | ew;
                while (both) {
                    const uint32_t b = (uint32_t)__builtin_ctz(both);
                    both &= both - 1u;
                    const uint32_t pos = ((w0 + k) << 5) + b;
                    uint32_t kind = (uint32_t)EV_START;
                    if (!((sw >> b) & 1u))
                        kind = pos >= length ? (uint32_t)EV_END_EOS : (((s_brk[bb + k] >> b) & 1u) ? (uint32_t)EV_END_N : (uint32_t)EV_END_ZERO);
                    const uint64_t e = ev_pack(pos, ml, kind);
                    if (direct) { if (idx < sink.region_cap) gdst[idx] = e; }
                    else stage[idx] = e;
                    idx++;
                }
            };
            expand(s0, e0, 2u * (uint32_t)j);
            expand(s1, e1, 2u * (uint32_t)j + 1u);
        }
        __builtin_amdgcn_wave_barrier();
    } while (RB_COOP_MAX_LANES > COOP_SRC && todo != 0ull);
    if (!direct) staged += total;
}
// ------------------------------------------------------------------------- group filter
// See scan_anchored_kernel.  PASS / EVAL: words k = -1 .. K of the pass bits and of the evaluated-window mask (word K of PASS
// need only be exact in bits 0..23).  tj (1 < tj <= GROUP_FILTER_MAX, wave-uniform): positions a group must span.
// SV (words k = -1 .. K-1; of word -1 only bit 31 is meaningful): bit q = the group position q belongs to is kept.
// DR (own words): END positions of the groups that are dropped.  Called wave-uniformly.
__device__ __forceinline__ void group_filter(const uint32_t (&PASS)[K + 2], const uint32_t (&EVAL)[K + 2], int tj, uint32_t (&SV)[K + 1],
                                             uint32_t (&DR)[K]) {
    uint32_t J[K + 2], W[K + 2];
    // closing over gaps <= 7: dilate towards higher positions by 7, erode back
#pragma unroll
    for (int j = K + 1; j >= 1; j--) J[j] = PASS[j] | funnel(PASS[j], PASS[j - 1], 31);
    J[0] = PASS[0] | (PASS[0] << 1);
#pragma unroll
    for (int j = K + 1; j >= 1; j--) J[j] = J[j] | funnel(J[j], J[j - 1], 30);
    J[0] = J[0] | (J[0] << 2);
#pragma unroll
    for (int j = K + 1; j >= 1; j--) J[j] = J[j] | funnel(J[j], J[j - 1], 28);
    J[0] = J[0] | (J[0] << 4);
#pragma unroll
    for (int j = 0; j < K + 1; j++) J[j] = J[j] & funnel(J[j + 1], J[j], 1);
    J[K + 1] = J[K + 1] & (J[K + 1] >> 1);
#pragma unroll
    for (int j = 0; j < K + 1; j++) J[j] = J[j] & funnel(J[j + 1], J[j], 2);
    J[K + 1] = J[K + 1] & (J[K + 1] >> 2);
#pragma unroll
    for (int j = 0; j < K + 1; j++) J[j] = J[j] & funnel(J[j + 1], J[j], 4);
    J[K + 1] = J[K + 1] & (J[K + 1] >> 4);
    // the closing only ever adds bits between pass bits; where it ran out of neighbours (top of word K, bottom of
    // word -1) it lost some: put the pass bits back so that J covers PASS everywhere
#pragma unroll
    for (int j = 0; j < K + 2; j++) J[j] |= PASS[j];
    // opening with span tj (<= 16): W = erosion anchored at the low end, then dilated back
#pragma unroll
    for (int j = 0; j < K + 2; j++) W[j] = J[j];
    {
        int span = 1;
#pragma unroll
        for (int step = 0; step < 4; step++) {
            const int sh = min(span, tj - span);          // wave-uniform
            if (sh > 0) {
#pragma unroll
                for (int j = 0; j < K + 1; j++) W[j] = W[j] & funnel(W[j + 1], W[j], (uint32_t)sh);
                W[K + 1] = W[K + 1] & (W[K + 1] >> sh);
                span += sh;
            }
        }
        span = 1;
#pragma unroll
        for (int step = 0; step < 4; step++) {
            const int sh = min(span, tj - span);
            if (sh > 0) {
#pragma unroll
                for (int j = K + 1; j >= 1; j--) W[j] = W[j] | funnel(W[j], W[j - 1], (uint32_t)(32 - sh));
                W[0] = W[0] | (W[0] << sh);
                span += sh;
            }
        }
    }
    // groups kept regardless of their length: the group's end E (first position after its J run) is not an evaluated
    // window (an N or the end of the record closes its last streak), or the window E + 8 is not (its call is made out
    // of turn).  Short ones among them (< tj <= 16 positions) are found by a flood from E - 1 down the run.
    auto special_ends = [&](int j) {          // special group ends in word j - 1
        const uint32_t jprev = j > 0 ? funnel(J[j], J[j - 1], 31) : (J[0] << 1);            // bit b = J at b - 1
        const uint32_t ev8 = j < K + 1 ? funnel(EVAL[j + 1], EVAL[j], 8) : (EVAL[K + 1] >> 8);      // bit b = EVAL at b + 8
        const uint32_t g = ~J[j] & jprev & ~(EVAL[j] & ev8);
        // a group that reaches an own position ends at or beyond the first own position (j = 0: none); ends more than 16
        // positions beyond the last one cannot belong to a group shorter than tj that reaches it (and J is not exact there)
        return j == 0 ? 0u : (j == K + 1 ? (g & 0x0000ffffu) : g);
    };
    uint32_t special = 0;
#pragma unroll
    for (int j = 1; j < K + 2; j++) special |= special_ends(j);
    if (__ballot(special != 0) != 0ull) {
        // G := positions of J runs within 16 below a special end
        uint32_t G[K + 2], JL[K + 2];
#pragma unroll
        for (int j = 0; j < K + 2; j++) { JL[j] = J[j]; G[j] = special_ends(j); }
#pragma unroll
        for (int j = 0; j < K + 1; j++) G[j] = funnel(G[j + 1], G[j], 1) & J[j];        // E - 1, inside the run
        G[K + 1] = (G[K + 1] >> 1) & J[K + 1];
#pragma unroll
        for (int sh = 1; sh <= 8; sh <<= 1) {
            // G |= (G >> sh) & JL, JL[q] = J all ones over q .. q + sh - 1 (... up to the source bit)
#pragma unroll
            for (int j = 0; j < K + 1; j++) G[j] |= funnel(G[j + 1], G[j], (uint32_t)sh) & JL[j];
            G[K + 1] |= (G[K + 1] >> sh) & JL[K + 1];
#pragma unroll
            for (int j = 0; j < K + 1; j++) JL[j] = JL[j] & funnel(JL[j + 1], JL[j], (uint32_t)sh);
            JL[K + 1] = JL[K + 1] & (JL[K + 1] >> sh);
        }
#pragma unroll
        for (int j = 0; j < K + 1; j++) W[j] |= G[j];
    }
#pragma unroll
    for (int j = 0; j < K + 1; j++) SV[j] = W[j];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t jprev = funnel(J[k + 1], J[k], 31);
        const uint32_t svprev = funnel(W[k + 1], W[k], 31);
        DR[k] = ~J[k + 1] & jprev & ~svprev;
    }
}

// -------------------------------------------------------------------------- perfect scan
constexpr int LDS_EXTRA = 40;   // halo + shifted-operand words (supports shifts < 1024)

// Hot loop of processShiftXORsPerfect (parse_perfect_shiftxor.cpp:173-223).
// For motif m the reference walks maximal runs of X_m ones over non-N bases.  With
// Z = mismatch | brk those are maximal zero runs of Z.  The run-length cut-offs are
// c1 = (m<=6 ? 12-m : m) for a run closed by a mismatch (:193) and >= c1 otherwise (:179),
// so only runs of at least sp = min(c1, 32) zeros can matter; the host applies the exact
// cut-off.  Per lane and motif:
//   D[p]  = OR of Z[p .. p+sp-1]          (3-5 OR-doubling funnel steps)
//   START = Z[p-1] & ~D[p]                 run of >= sp zeros begins at p
//   END   = Z[p]  & ~D[p-sp]               run of >= sp zeros ends just before p
// Every qualifying run yields exactly one START and one END (position L for an open run,
// because brk is 1 from L on), so the host pairs them without any device-side walk.
// D = OR of Z over [p, p+sp) by funnel doubling (spans 2, 4, 4+t3, 4+t3+t4, 4+t3+t4+t5 = sp), then
// START = Z[p-1] & ~D[p] and END = Z[p] & ~D[p-sp] on the own words.  The shift amounts may differ from lane to
// lane (queued candidates of several motifs, see scan_perfect_kernel); a zero amount makes its step a no-op.
template <bool UNIFORM>
__device__ __forceinline__ uint32_t perfect_edges(const uint32_t (&Z)[K + 2], uint32_t t3, uint32_t t4, uint32_t t5,
                                                  uint32_t back, uint32_t (&SQ)[K], uint32_t (&EQ)[K]) {
    uint32_t D[K + 2];
#pragma unroll
    for (int j = 0; j < K + 1; j++) D[j] = Z[j] | funnel(Z[j + 1], Z[j], 1);
    D[K + 1] = Z[K + 1] | (Z[K + 1] >> 1);
#pragma unroll
    for (int j = 0; j < K + 1; j++) D[j] = D[j] | funnel(D[j + 1], D[j], 2);
    D[K + 1] = D[K + 1] | (D[K + 1] >> 2);
#pragma unroll
    for (int j = 0; j < K + 1; j++) D[j] = D[j] | funnel(D[j + 1], D[j], t3);
    D[K + 1] = D[K + 1] | (D[K + 1] >> t3);
#pragma unroll
    for (int j = 0; j < K + 1; j++) D[j] = D[j] | funnel(D[j + 1], D[j], t4);
    if (!UNIFORM || t5) {   // uniform case: only motifs with a cut-off above 16 need the fifth step
        D[K + 1] = D[K + 1] | (D[K + 1] >> t4);
#pragma unroll
        for (int j = 0; j < K + 1; j++) D[j] = D[j] | funnel(D[j + 1], D[j], t5);
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = k + 1;
        const uint32_t prev_z = funnel(Z[j], Z[j - 1], 31);      // bit b = Z at position b-1
        const uint32_t d_back = funnel(D[j], D[j - 1], back);    // bit b = D at position b-sp
        SQ[k] = prev_z & ~D[j];
        EQ[k] = Z[j] & ~d_back;
        any = bitop3<0xFE>(SQ[k], EQ[k], any);
    }
    return any;
}

// per-wave queue of candidate lanes (LDS): the Z window of a lane whose prefilter fired, and which motif it is of
constexpr int CAND_CAP = 64;          // one wavefront's worth
#ifndef RB_CAND_DIRECT
#define RB_CAND_DIRECT 32
#endif
constexpr int CAND_DIRECT = RB_CAND_DIRECT;   // a (tile, motif) pair with more candidate lanes than this is scanned in place
struct CandQueue {
    uint32_t z[K + 2][CAND_CAP];      // word-major: lane i of a flush reads z[w][i] (conflict free)
    uint32_t tag[CAND_CAP];           // source lane | motif << 8
};

#ifndef RB_PERFECT_WAVES
#define RB_PERFECT_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RB_PERFECT_WAVES, RB_PERFECT_WAVES))) void scan_perfect_kernel(DevicePlanes pl, PerfectLaunch pp, int motifs_per_block,
                                                           uint64_t *__restrict__ events,
                                                           uint32_t *__restrict__ counters) {
    __shared__ uint32_t s_hi[TILE_WORDS + LDS_EXTRA];
    __shared__ uint32_t s_lo[TILE_WORDS + LDS_EXTRA];
    __shared__ uint32_t s_brk[TILE_WORDS + 8];
    __shared__ uint64_t s_stage[4][EV_STAGE];
    __shared__ CandQueue s_cand[4];
    __shared__ CoopScratch s_coop[4];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform on purpose: keeps the motif loop scalar
    const int64_t tile_base = (int64_t)blockIdx.x * TILE_WORDS;   // first word owned by this block

    const int bm_lo = pp.m_lo + (int)blockIdx.y * motifs_per_block;
    const int bm_hi = min(pp.m_hi, bm_lo + motifs_per_block - 1);
    const int q_hi = bm_hi >> 5;

    // stage words [tile_base-1, tile_base+TILE_WORDS+3+q_hi) : coalesced dword loads, shared by 4 waves
    const int n_ext = TILE_WORDS + 4 + q_hi;
    for (int i = threadIdx.x; i < n_ext; i += 256) {
        s_hi[i] = pl.hi[tile_base - 1 + i];
        s_lo[i] = pl.lo[tile_base - 1 + i];
    }
    for (int i = threadIdx.x; i < TILE_WORDS + 2; i += 256) s_brk[i] = pl.brk[tile_base - 1 + i];
    __syncthreads();

    // The block's motifs are dealt round-robin to its 4 waves (one wavefront per (tile, motif) pair at a time).
    // Small motifs produce most of the runs and large ones mostly leave through the prefilter below, so
    // contiguous quarters would make wave 0 the block's critical path (measured: 38 % VALU utilisation).
    const int wm_lo = bm_lo + wave;
    const int wm_hi = bm_hi;
    if (wm_lo > wm_hi) return;

    const int lb = lane * K;   // LDS index of word k = -1 of this lane
    uint32_t H[K + 2], Lo[K + 2], Bk[K + 2];
#pragma unroll
    for (int j = 0; j < K + 2; j++) { H[j] = s_hi[lb + j]; Lo[j] = s_lo[lb + j]; Bk[j] = s_brk[lb + j]; }

    uint32_t Hq[K + 3], Lq[K + 3];
    int cur_q = -1;
    const uint32_t length = (uint32_t)pl.length;
    const uint32_t word0 = (uint32_t)(tile_base + lb);   // global index of own word k = 0

    EventSink sink;
    sink.events = events;
    sink.counters = counters;
    sink.region_cap = pp.ev_cap / EV_SHARDS;
    sink.shard = (blockIdx.x * 4u + (uint32_t)wave + blockIdx.y) % EV_SHARDS;
    volatile uint64_t *stage = s_stage[wave];
    int staged = 0;   // wave-uniform

    // Candidate queue.  The doubling chain below is ~2/3 of the VALU work, yet where the prefilter fires it does
    // so in 2-3 of the 64 lanes (a repeat locus is a few hundred bases of a 16-kb tile; 49 % of the large-motif
    // (tile, motif) pairs of the synthetic FASTA have such a lane).  Instead of running the chain on the whole
    // wave for them, the few candidate lanes park their Z window in LDS; once a wavefront's worth has collected
    // -- over several motifs -- the chain runs once, every lane on a different (source lane, motif) with its own
    // shift amounts.  A pair's candidates are never split over two flushes, so its events still form one chunk.
    CandQueue &cq = s_cand[wave];
    int queued = 0;   // wave-uniform
    auto flush_candidates = [&]() {
        if (queued == 0) return;
        const bool live = lane < queued;
        const uint32_t tag = live ? cq.tag[lane] : 0u;
        const uint32_t src = tag & 0xffu, qm = tag >> 8;
        uint32_t Zq[K + 2];
#pragma unroll
        for (int j = 0; j < K + 2; j++) Zq[j] = live ? cq.z[j][lane] : 0xffffffffu;
        const int c1 = (qm <= 6u) ? 12 - (int)qm : (int)qm;
        const int sp = live ? min(c1, 32) : 32;
        const uint32_t t3 = (uint32_t)min(4, sp - 4);
        const uint32_t t4 = (uint32_t)min(8, sp - 4 - (int)t3);
        const uint32_t t5 = (uint32_t)(sp - 4) - t3 - t4;
        uint32_t SQ[K], EQ[K];
        uint32_t any = perfect_edges<false>(Zq, t3, t4, t5, 32u - (uint32_t)sp, SQ, EQ);
        if (!live) {
            any = 0;
#pragma unroll
            for (int k = 0; k < K; k++) { SQ[k] = 0; EQ[k] = 0; }
        }
#ifdef RB_ABLATE_STAGING        // tools/build_variant.sh: timing without the event staging
        const unsigned long long with_events = __ballot(any == 0xdeadbeefu);
#else
        const unsigned long long with_events = __ballot(any != 0);
#endif
        if (with_events != 0ull) {
            const uint32_t src_word0 = (uint32_t)tile_base + src * (uint32_t)K;
            if (__popcll(with_events) <= RB_COOP_MAX_LANES)
                stage_events_coop(SQ, EQ, src_word0, qm, src * (uint32_t)K + 1u, s_brk, length, s_coop[wave], sink, stage, staged, lane);
            else
                stage_events(SQ, EQ, src_word0, qm, sink, stage, staged, lane, [&](int k, uint32_t b, uint32_t pos) {
                    if (pos >= length) return (uint32_t)EV_END_EOS;
                    return ((s_brk[src * (uint32_t)K + (uint32_t)k + 1u] >> b) & 1u) ? (uint32_t)EV_END_N : (uint32_t)EV_END_ZERO;
                });
        }
        queued = 0;
        __builtin_amdgcn_wave_barrier();
    };

    for (int m = wm_lo; m <= wm_hi; m += 4) {
        const int q = m >> 5;
        const uint32_t r = (uint32_t)m & 31u;
        if (q != cur_q) {
            cur_q = q;
#pragma unroll
            for (int j = 0; j < K + 3; j++) { Hq[j] = s_hi[lb + j + q]; Lq[j] = s_lo[lb + j + q]; }
        }
        const int c1 = (m <= 6) ? 12 - m : m;
        const int sp = min(c1, 32);
        const uint32_t t3 = (uint32_t)min(4, sp - 4);
        const uint32_t t4 = (uint32_t)min(8, sp - 4 - (int)t3);
        const uint32_t t5 = (uint32_t)(sp - 4) - t3 - t4;

        // Z = mismatch | break, words k = -1 .. K
        uint32_t Z[K + 2];
#pragma unroll
        for (int j = 0; j < K + 2; j++) {
            const uint32_t hs = funnel(Hq[j + 1], Hq[j], r);
            const uint32_t ls = funnel(Lq[j + 1], Lq[j], r);
            Z[j] = bitop3<0xF6>(bitop3<0xBE>(H[j], hs, Bk[j]), Lo[j], ls);      // ((H ^ hs) | Bk) | (Lo ^ ls)
        }
        // Cheap necessary condition for a lane to hold a START or END of this motif: the first / last sp positions
        // of a run lie inside the lane's 10-word window, and sp >= 7 consecutive zeros cover an aligned nibble,
        // >= 15 an aligned byte, >= 31 an aligned halfword.  (On random DNA 96 % of the (tile, motif) pairs with
        // sp >= 15 have no such lane at all.)  sp = 6 (motif 6 only) has no such test and is scanned in place.
        if (sp >= 7) {
            bool cand;
            if (sp >= 31) {
                // an aligned zero halfword somewhere in the window  <=>  the packed 16-bit minimum over the ten words has a
                // zero half: nine v_pk_min_u16 and one zero-halfword test instead of ten of the latter
                typedef unsigned short half2 __attribute__((ext_vector_type(2)));
                auto pk_min = [](uint32_t a, uint32_t b) {
                    const half2 m = __builtin_elementwise_min(__builtin_bit_cast(half2, a), __builtin_bit_cast(half2, b));
                    return __builtin_bit_cast(uint32_t, m);
                };
                uint32_t lowest = Z[0];
#pragma unroll
                for (int j = 1; j < K + 2; j++) lowest = pk_min(lowest, Z[j]);
                cand = (((lowest - 0x00010001u) & ~lowest) & 0x80008000u) != 0;
            } else {
                const uint32_t ones = sp >= 15 ? 0x01010101u : 0x11111111u;
                const uint32_t tops = sp >= 15 ? 0x80808080u : 0x88888888u;
                uint32_t hit = 0;
#pragma unroll
                for (int j = 0; j < K + 2; j++) hit = bitop3<0xBA>(Z[j] - ones, Z[j], hit);      // ((Z - ones) & ~Z) | hit
                cand = (hit & tops) != 0;
            }
#ifdef RB_ABLATE_CHAIN          // ... and without the doubling chain / candidate queue: what remains is Z + prefilter
            const unsigned long long mask = __ballot(cand && Z[0] == 0xdeadbeefu);
#else
            const unsigned long long mask = __ballot(cand);
#endif
            if (mask == 0ull) continue;
            const int n = __popcll(mask);
            if (n <= CAND_DIRECT) {
                if (queued + n > CAND_CAP) flush_candidates();
                if (cand) {
                    const int slot = queued + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
#pragma unroll
                    for (int j = 0; j < K + 2; j++) cq.z[j][slot] = Z[j];
                    cq.tag[slot] = (uint32_t)lane | ((uint32_t)m << 8);
                }
                queued += n;
                __builtin_amdgcn_wave_barrier();
                continue;
            }
        }
        // dense pair: scan the whole wave in place (its events are staged now, those of parked candidates later:
        // either way a (tile, motif) pair's events stay in one piece)
        uint32_t SQ[K], EQ[K];
        const uint32_t any = perfect_edges<true>(Z, t3, t4, t5, 32u - (uint32_t)sp, SQ, EQ);
#ifdef RB_ABLATE_STAGING
        const unsigned long long with_events = __ballot(any == 0xdeadbeefu);
#else
        const unsigned long long with_events = __ballot(any != 0);
#endif
        if (with_events != 0ull) {
            if (__popcll(with_events) <= RB_COOP_MAX_LANES)
                stage_events_coop(SQ, EQ, word0, (uint32_t)m, (uint32_t)lb + 1u, s_brk, length, s_coop[wave], sink, stage, staged, lane);
            else
                stage_events(SQ, EQ, word0, (uint32_t)m, sink, stage, staged, lane, [&](int k, uint32_t b, uint32_t pos) {
                    if (pos >= length) return (uint32_t)EV_END_EOS;
                    return ((Bk[k + 1] >> b) & 1u) ? (uint32_t)EV_END_N : (uint32_t)EV_END_ZERO;
                });
        }
    }
    flush_candidates();
    sink_flush(sink, stage, staged, lane);
}

void launch_scan_perfect(const DevicePlanes &pl, const PerfectLaunch &pp, uint64_t *events, uint32_t *counters,
                         hipStream_t stream) {
    const int nm = pp.m_hi - pp.m_lo + 1;
    if (nm <= 0 || pl.ntiles <= 0) return;
    // fill the chip: >= ~2048 blocks when the record is short, by splitting the motif range
    int64_t want_y = (2048 + pl.ntiles - 1) / pl.ntiles;
    int max_y = (nm + 3) / 4;              // at least one motif per wave
    int gy = (int)(want_y < 1 ? 1 : (want_y > max_y ? max_y : want_y));
    int motifs_per_block = (nm + gy - 1) / gy;
    gy = (nm + motifs_per_block - 1) / motifs_per_block;
    dim3 grid((unsigned)pl.ntiles, (unsigned)gy);
    hipLaunchKernelGGL(scan_perfect_kernel, grid, dim3(256), 0, stream, pl, pp, motifs_per_block, events, counters);
}

// --------------------------------------------------------------------------- window scan
// Hot loop of processShiftXORswithSubstitutions (parse_substitute_shiftxor.cpp:430-532) and, with
// ALLOWED = 2, of processShiftXORsAnchored (parse_anchored_shiftxor.cpp:580-679).
// The reference slides an 8-bit window over X_m, skipping N: at scan position p (window start
// q = p-7) it "passes" when popcount(X_m[q..q+7]) >= 8-ALLOWED and no N lies in [q, q+7]
// (`valid_position >= window_length`).  Here, per 32-base word and motif:
//   ge2 / ge3 = "at least 2 / 3 mismatches in [q, q+7]"   bit-sliced counter, 3 doubling levels
//   pass      = evaluated & ~(ALLOWED == 1 ? ge2 : ge3)
// and only the transitions of `pass` leave the chip: START at the first passing q of a streak,
// END at the first non-passing q after it, classified FAIL (evaluated, failed -> the reference's
// "pending end" is q+7), N (position q+7 is an N: the streak is discarded, :433-458) or EOS
// (q+7 == L: streak still open at the end-of-sequence flush, :534-574).  The per-motif finite
// state machine that turns streaks into addSeed calls is replayed on the host from these events.
template <int ALLOWED>
__global__ __launch_bounds__(256) void scan_window_kernel(DevicePlanes pl, PerfectLaunch pp, int motifs_per_block,
                                                          uint64_t *__restrict__ events,
                                                          uint32_t *__restrict__ counters) {
    __shared__ uint32_t s_hi[TILE_WORDS + LDS_EXTRA];
    __shared__ uint32_t s_lo[TILE_WORDS + LDS_EXTRA];
    __shared__ uint32_t s_brk[TILE_WORDS + 8];
    __shared__ uint64_t s_stage[4][EV_STAGE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform on purpose: keeps the motif loop scalar
    const int64_t tile_base = (int64_t)blockIdx.x * TILE_WORDS;

    const int bm_lo = pp.m_lo + (int)blockIdx.y * motifs_per_block;
    const int bm_hi = min(pp.m_hi, bm_lo + motifs_per_block - 1);
    const int q_hi = bm_hi >> 5;
    const int n_ext = TILE_WORDS + 4 + q_hi;
    for (int i = threadIdx.x; i < n_ext; i += 256) {
        s_hi[i] = pl.hi[tile_base - 1 + i];
        s_lo[i] = pl.lo[tile_base - 1 + i];
    }
    for (int i = threadIdx.x; i < TILE_WORDS + 2; i += 256) s_brk[i] = pl.brk[tile_base - 1 + i];
    __syncthreads();

    // motifs dealt round-robin to the 4 waves (see scan_perfect_kernel)
    const int wm_lo = bm_lo + wave;
    const int wm_hi = bm_hi;
    if (wm_lo > wm_hi) return;

    const int lb = lane * K;
    uint32_t H[K + 2], Lo[K + 2];
    uint32_t EVAL[K + 1];   // words k = -1 .. K-1: window starting here holds no break
    {
        uint32_t B[K + 2];
#pragma unroll
        for (int j = 0; j < K + 2; j++) { H[j] = s_hi[lb + j]; Lo[j] = s_lo[lb + j]; B[j] = s_brk[lb + j]; }
#pragma unroll
        for (int j = 0; j < K + 1; j++) B[j] |= funnel(B[j + 1], B[j], 1);
        B[K + 1] |= B[K + 1] >> 1;
#pragma unroll
        for (int j = 0; j < K + 1; j++) B[j] |= funnel(B[j + 1], B[j], 2);
        B[K + 1] |= B[K + 1] >> 2;
#pragma unroll
        for (int j = 0; j < K + 1; j++) EVAL[j] = ~(B[j] | funnel(B[j + 1], B[j], 4));
    }

    uint32_t Hq[K + 3], Lq[K + 3];
    int cur_q = -1;
    const uint32_t length = (uint32_t)pl.length;
    const uint32_t word0 = (uint32_t)(tile_base + lb);

    EventSink sink;
    sink.events = events;
    sink.counters = counters;
    sink.region_cap = pp.ev_cap / EV_SHARDS;
    sink.shard = (blockIdx.x * 4u + (uint32_t)wave + blockIdx.y) % EV_SHARDS;
    volatile uint64_t *stage = s_stage[wave];
    int staged = 0;

    for (int m = wm_lo; m <= wm_hi; m += 4) {
        const int q = m >> 5;
        const uint32_t r = (uint32_t)m & 31u;
        if (q != cur_q) {
            cur_q = q;
#pragma unroll
            for (int j = 0; j < K + 3; j++) { Hq[j] = s_hi[lb + j + q]; Lq[j] = s_lo[lb + j + q]; }
        }
        // A1 = any mismatch in a span, B1 = at least two, C1 = at least three (spans 1 -> 2 -> 4 -> 8)
        uint32_t A1[K + 2], B1[K + 2], C1[K + 2];
#pragma unroll
        for (int j = 0; j < K + 2; j++) {
            const uint32_t hs = funnel(Hq[j + 1], Hq[j], r);
            const uint32_t ls = funnel(Lq[j + 1], Lq[j], r);
            A1[j] = (H[j] ^ hs) | (Lo[j] ^ ls);     // mismatch word (N compares as A, as in the reference)
        }
#pragma unroll
        for (int j = 0; j < K + 2; j++) {          // span 2
            const uint32_t sa = (j <= K) ? funnel(A1[j + 1 <= K + 1 ? j + 1 : j], A1[j], 1) : (A1[j] >> 1);
            B1[j] = A1[j] & sa;
            A1[j] = A1[j] | sa;
        }
#pragma unroll
        for (int j = 0; j < K + 2; j++) {          // span 4
            const uint32_t sa = (j <= K) ? funnel(A1[j + 1 <= K + 1 ? j + 1 : j], A1[j], 2) : (A1[j] >> 2);
            const uint32_t sb = (j <= K) ? funnel(B1[j + 1 <= K + 1 ? j + 1 : j], B1[j], 2) : (B1[j] >> 2);
            if (ALLOWED == 2) C1[j] = (B1[j] & sa) | (A1[j] & sb);
            B1[j] = B1[j] | sb | (A1[j] & sa);
            A1[j] = A1[j] | sa;
        }
        uint32_t PASS[K + 1];
#pragma unroll
        for (int j = 0; j < K + 1; j++) {          // span 8, words k = -1 .. K-1 only
            const uint32_t sa = funnel(A1[j + 1], A1[j], 4);
            const uint32_t sb = funnel(B1[j + 1], B1[j], 4);
            uint32_t bad;
            if (ALLOWED == 2) {
                const uint32_t sc = funnel(C1[j + 1], C1[j], 4);
                bad = C1[j] | sc | (B1[j] & sa) | (A1[j] & sb);
            } else {
                bad = B1[j] | sb | (A1[j] & sa);
            }
            PASS[j] = EVAL[j] & ~bad;
        }
        uint32_t ST[K], EN[K];
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t prev = funnel(PASS[k + 1], PASS[k], 31);   // bit b = pass at q-1
            ST[k] = PASS[k + 1] & ~prev;
            EN[k] = ~PASS[k + 1] & prev;
            any |= ST[k] | EN[k];
        }
        if (__ballot(any != 0) != 0ull) {
            stage_events(ST, EN, word0, (uint32_t)m, sink, stage, staged, lane, [&](int k, uint32_t b, uint32_t pos) {
                if ((EVAL[k + 1] >> b) & 1u) return (uint32_t)EV_END_ZERO;          // evaluated and failed
                return (pos + 7u >= length) ? (uint32_t)EV_END_EOS : (uint32_t)EV_END_N;
            });
        }
    }
    sink_flush(sink, stage, staged, lane);
}

void launch_scan_window(const DevicePlanes &pl, const PerfectLaunch &pp, int allowed_mismatches, uint64_t *events,
                        uint32_t *counters, hipStream_t stream) {
    const int nm = pp.m_hi - pp.m_lo + 1;
    if (nm <= 0 || pl.ntiles <= 0) return;
    int64_t want_y = (2048 + pl.ntiles - 1) / pl.ntiles;
    int max_y = (nm + 3) / 4;
    int gy = (int)(want_y < 1 ? 1 : (want_y > max_y ? max_y : want_y));
    int motifs_per_block = (nm + gy - 1) / gy;
    gy = (nm + motifs_per_block - 1) / motifs_per_block;
    dim3 grid((unsigned)pl.ntiles, (unsigned)gy);
    if (allowed_mismatches == 1)
        hipLaunchKernelGGL(scan_window_kernel<1>, grid, dim3(256), 0, stream, pl, pp, motifs_per_block, events, counters);
    else
        hipLaunchKernelGGL(scan_window_kernel<2>, grid, dim3(256), 0, stream, pl, pp, motifs_per_block, events, counters);
}

// ------------------------------------------------------------------------- anchored scan
// Fused generateAnchoredShiftXORs (parse_anchored_shiftxor.cpp:20-56), plane composition
// (fasta_utils.cpp:143-161) and window scan of processShiftXORsAnchored (:580-679, threshold 6).
//
// anchor_s keeps the bits of every maximal run of X_s ones that (a) is closed by a zero at a
// position p <= L-1-s (the reference only walks p = 0..L-1-s, so a run still open there is dropped)
// and (b) has 3 <= length < 2s.  N is ignored.  To make (a) a length test, X_s is forced to 1 from
// p = L-s on: such a run becomes unbounded and fails "< 2s".  Run lengths are classified per lane:
//   * a lane owns 8 consecutive words (256 bases); per word it knows its leading / trailing ones;
//   * a sequential carry over the 8 words gives, for every word, the length of the run entering
//     from the left (CL) and from the right (CR) -- seeded with the trailing / leading ones of the
//     lanes to the left / right, combined over `hl` lanes by a segmented Kogge-Stone scan across the
//     wave (256 per all-ones lane; hl*256 >= 2s + 32, so longer runs need not be told apart);
//   * runs interior to a word are handled bit-parallel (>= 3 by AND of shifts; >= 2s, only
//     possible when 2s <= 30, by AND-/OR-doubling).
// The first and last hl lanes are halo lanes (what lies beyond them is unknown): they compute but emit
// nothing (device_planes.h: anchored_halo_lanes).
// The composed plane XA_m = X_m | anchor_{m-2} | anchor_{m-1} | anchor_{m+1} | anchor_{m+2}
// needs five consecutive shifts, so every wave walks s = m_lo-2 .. m_hi+2 of its motif group with a
// register ring of the last five anchor words and the last three mismatch words.
constexpr int RUN_SAT = 1 << 20;

//
// Group filter (round 3).  The window state machine downstream turns every GROUP of pass-streaks (streaks separated by at
// most 7 failing windows, window_stage.hip) into one addSeed call of length (group end + 7 - group start), and nine such
// calls in ten fail the stage's length filter (parse_anchored_shiftxor.cpp:572-573) -- yet every streak of every group used
// to leave the kernel as two 8-byte events (29 of the 42 bytes per base this kernel wrote, two thirds of its time).  Groups
// are runs of J = closing of the pass bits over gaps <= 7 (6 funnel steps), and a group's length is that of its J run, so
// the filter is an opening of J with the span `tj` the call must reach (min(cut-off, 23) - 7 <= 16 positions: 8 more steps):
// streaks of groups that cannot pass leave no event.  Kept regardless are groups that end at an N, at the end of the
// record, or right before a blocked stretch (the calls of those are made out of turn, "edge" calls downstream): rare, found
// by a flood from their end marks on a wave-uniform branch.  What a dropped group would have contributed besides its call --
// a bit in the map of positions at which ordinary calls are made, the largest end of any call -- is a function of where
// it ends, so its end bit goes to `dropmap` (one OR per lane word and wave at the end of the motif loop).
// Every decision is a function of absolute positions inside a lane's view (its own words, one word to the left, 24 bits of
// the right neighbour's first word), so all lanes and tiles agree on every group.
// This kernel makes the planes only (anchor classification + composition, XA_m written to HBM); the window scan of those planes,
// with the group filter described above, is scan_xa_window_kernel's.  Rounds 1-2 did both in one fused kernel, which held the
// shifted operands, the five-shift anchor ring AND the window counters' words at once (256 VGPRs, two waves per SIMD, VALUBusy
// 58 %, 3.3 ms per 100 Mbp); apart, the planes kernel and the window kernel each keep half of that and run at twice the
// occupancy (0.68 + 1.60 ms) for one extra read of the planes (max_motif / 8 bytes per base at HBM speed).  The fused form was
// kept behind a switch through round 3 and is gone since round 4 (DESIGN.md 4 has its measurements).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void scan_anchored_kernel(DevicePlanes pl, PerfectLaunch pp, int motifs_per_block, int hl,
                                                            uint32_t *__restrict__ xa, int64_t xa_stride) {
    __shared__ uint32_t s_hi[64 * K + LDS_EXTRA];
    __shared__ uint32_t s_lo[64 * K + LDS_EXTRA];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform on purpose: keeps the motif loop scalar
    const int64_t tile_base = (int64_t)blockIdx.x * anchored_tile_words(hl);   // first OWN word (lane hl, k = 0)
    const int64_t first = tile_base - (int64_t)hl * K - 1;                     // word held at LDS index 0

    const int bm_lo = pp.m_lo + (int)blockIdx.y * motifs_per_block;
    const int bm_hi = min(pp.m_hi, bm_lo + motifs_per_block - 1);
    const int q_hi = (bm_hi + 2) >> 5;
    const int n_ext = 64 * K + 4 + q_hi;
    for (int i = threadIdx.x; i < n_ext; i += 256) {
        s_hi[i] = pl.hi[first + i];
        s_lo[i] = pl.lo[first + i];
    }
    __syncthreads();

    const int nmb = bm_hi - bm_lo + 1;
    const int per = (nmb + 3) >> 2;
    const int wm_lo = bm_lo + wave * per;
    const int wm_hi = min(bm_hi, wm_lo + per - 1);
    if (wm_lo > wm_hi) return;

    const int lb = lane * K;
    const bool own_lane = lane >= hl && lane < 64 - hl;
    uint32_t H[K + 2], Lo[K + 2];
#pragma unroll
    for (int j = 0; j < K + 2; j++) { H[j] = s_hi[lb + j]; Lo[j] = s_lo[lb + j]; }

    const int64_t w_own0 = tile_base + (int64_t)(lane - hl) * K;    // global index of this lane's word k = 0
    const int64_t length = pl.length;
    // masks are only needed where the lane touches p < 0 or p >= L - (largest shift)
    const int64_t lane_lo = (w_own0 - 1) * 32, lane_hi = (w_own0 + K + 1) * 32;
    const bool edge_lane = lane_lo < 0 || lane_hi > length - (int64_t)(wm_hi + 2);
    const bool edge_wave = __ballot(edge_lane) != 0ull;

    uint32_t Hq[K + 3], Lq[K + 3];
    int cur_q = -1;

    // Ring over the last five shifts, slot = (s - s_first) mod 5: anchor words (index j: word k = j-1).  The shift loop
    // is unrolled five-fold so that every slot index is a compile-time constant: the ring lives in registers and
    // nothing is copied when the window of shifts moves on (a rotating ring cost 70 v_mov per shift).  The mismatch
    // words of motif m = s - 2 are not kept across two shifts but recomputed when the motif is composed (4 ops per
    // word against 30 registers).
    uint32_t AN[5][K + 2];
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int j = 0; j < K + 2; j++) AN[a][j] = 0;

    const int s_first = max(1, wm_lo - 2);
    const int s_last = wm_hi + 2;
    auto step = [&](auto slot_c, int s) {
        constexpr int R = decltype(slot_c)::value;             // ring slot of shift s; shift s - d lives in slot (R - d) mod 5
        constexpr int R1 = (R + 4) % 5, R3 = (R + 2) % 5, R4 = (R + 1) % 5;
        const int q = s >> 5;
        const uint32_t r = (uint32_t)s & 31u;
        if (q != cur_q) {
            cur_q = q;
#pragma unroll
            for (int j = 0; j < K + 3; j++) { Hq[j] = s_hi[lb + j + q]; Lq[j] = s_lo[lb + j + q]; }
        }
        uint32_t X[K];   // X'_s on the own words k = 0..K-1
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t hs = funnel(Hq[k + 2], Hq[k + 1], r);
            const uint32_t ls = funnel(Lq[k + 2], Lq[k + 1], r);
            X[k] = ~((H[k + 1] ^ hs) | (Lo[k + 1] ^ ls));
        }
        if (edge_wave) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int64_t bp = (w_own0 + k) * 32;
                const int64_t f = (length - s) - bp;          // first forced bit of this word
                const uint32_t force = f <= 0 ? 0xffffffffu : (f >= 32 ? 0u : (0xffffffffu << (uint32_t)f));
                X[k] = (bp < 0) ? 0u : (X[k] | force);
            }
        }
        // ---- anchors of shift s on the own words
        const int two_s = 2 * s;
        int lead1[K], trail1[K];
        int span_lead = 0, span_trail = 0;
        bool alive = true;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t nx = ~X[k];
            lead1[k] = nx ? __builtin_ctz(nx) : 32;
            trail1[k] = nx ? __builtin_clz(nx) : 32;
            span_lead += alive ? lead1[k] : 0;
            alive = alive && (nx == 0u);
        }
        const bool span_full = alive;
        alive = true;
#pragma unroll
        for (int k = K - 1; k >= 0; k--) {
            span_trail += alive ? trail1[k] : 0;
            alive = alive && (X[k] == 0xffffffffu);
        }
        // ones ending at this lane's right edge (accL) / starting at its left edge (accR), over a growing window
        // of lanes: after the loop the windows span >= hl lanes, which is as far as a length below 2s can reach
        int accL = span_trail, accR = span_lead, fullL = span_full ? 1 : 0, fullR = fullL;
        for (int d = 1; d < hl; d <<= 1) {          // wave-uniform; no iteration up to max_motif 110
            int a2 = __shfl_up(accL, d), f2 = __shfl_up(fullL, d);
            if (lane < d) { a2 = 0; f2 = 0; }
            accL = min(accL + (fullL ? a2 : 0), RUN_SAT);
            fullL &= f2;
            a2 = __shfl_down(accR, d); f2 = __shfl_down(fullR, d);
            if (lane + d > 63) { a2 = 0; f2 = 0; }
            accR = min(accR + (fullR ? a2 : 0), RUN_SAT);
            fullR &= f2;
        }
        int left_in = __shfl_up(accL, 1);
        int right_in = __shfl_down(accR, 1);
        if (lane == 0) left_in = 0;
        if (lane == 63) right_in = 0;
        int CL[K], CR[K];
        {
            int c = left_in;
#pragma unroll
            for (int k = 0; k < K; k++) { CL[k] = c; c = (X[k] == 0xffffffffu) ? min(c + 32, RUN_SAT) : trail1[k]; }
            c = right_in;
#pragma unroll
            for (int k = K - 1; k >= 0; k--) { CR[k] = c; c = (X[k] == 0xffffffffu) ? min(c + 32, RUN_SAT) : lead1[k]; }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t x = X[k];
            const bool isfull = x == 0xffffffffu;
            const uint32_t mask_low = isfull ? 0xffffffffu : ((1u << lead1[k]) - 1u);
            const uint32_t mask_high = (isfull || trail1[k] == 0) ? 0u : (0xffffffffu << (32 - trail1[k]));
            const int low_len = CL[k] + lead1[k] + (isfull ? CR[k] : 0);
            const int high_len = trail1[k] + CR[k];
            const uint32_t inter = x & ~mask_low & ~mask_high;
            const uint32_t r3 = inter & (inter >> 1) & (inter >> 2);
            uint32_t keep = r3 | (r3 << 1) | (r3 << 2);
            if (two_s <= 30) {   // wave-uniform: interior runs can reach 2s only for small shifts
                uint32_t rr = inter;
                for (int span = 1; span < two_s;) { const int sh = min(span, two_s - span); rr &= rr >> sh; span += sh; }
                for (int span = 1; span < two_s;) { const int sh = min(span, two_s - span); rr |= rr << sh; span += sh; }
                keep &= ~rr;
            }
            if (lead1[k] > 0 && low_len >= 3 && low_len < two_s) keep |= mask_low;
            if (high_len >= 3 && high_len < two_s) keep |= mask_high;
            AN[R][k + 1] = keep;
        }

        const int m = s - 2;
        if (m < wm_lo) return;
        // ---- composed mismatch of motif m:  ~XA_m = mismatch_m & ~(anchor_{m-2,m-1,m+1,m+2})
        uint32_t A1[K + 2];
        {
            // mismatch words of shift m (words k = -1 .. K): the shifted operands usually sit in the registers loaded for
            // shift s = m + 2; only when m and s straddle a multiple of 32 they are read from LDS
            const int qm = m >> 5;
            const uint32_t rm = (uint32_t)m & 31u;
            constexpr int J0 = 1, J1 = K + 1;      // the planes need the own words only
            if (qm == cur_q) {
#pragma unroll
                for (int j = J0; j < J1; j++) {
                    const uint32_t hs = funnel(Hq[j + 1], Hq[j], rm);
                    const uint32_t ls = funnel(Lq[j + 1], Lq[j], rm);
                    A1[j] = ((H[j] ^ hs) | (Lo[j] ^ ls)) & ~(AN[R4][j] | AN[R3][j] | AN[R1][j] | AN[R][j]);
                }
            } else {
                uint32_t ph = s_hi[lb + J0 + qm], pl = s_lo[lb + J0 + qm];
#pragma unroll
                for (int j = J0; j < J1; j++) {
                    const uint32_t nh = s_hi[lb + j + 1 + qm], nl = s_lo[lb + j + 1 + qm];
                    A1[j] = ((H[j] ^ funnel(nh, ph, rm)) | (Lo[j] ^ funnel(nl, pl, rm))) & ~(AN[R4][j] | AN[R3][j] | AN[R1][j] | AN[R][j]);
                    ph = nh; pl = nl;
                }
            }
        }
        if (xa != nullptr && own_lane) {
            uint32_t *dst = xa + (int64_t)(m - pp.m_lo) * xa_stride + w_own0;
            if (w_own0 + K <= xa_stride) {          // the lane's 8 words: two 16-byte stores, contiguous across lanes
                static_assert(K == 8, "two uint4 stores cover a lane's words");
                reinterpret_cast<uint4 *>(dst)[0] = make_uint4(~A1[1], ~A1[2], ~A1[3], ~A1[4]);
                reinterpret_cast<uint4 *>(dst)[1] = make_uint4(~A1[5], ~A1[6], ~A1[7], ~A1[8]);
            } else {
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (w_own0 + k < xa_stride) dst[k] = ~A1[k + 1];
            }
        }
    };
    for (int s = s_first; s <= s_last; s += 5) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 <= s_last) step(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 <= s_last) step(std::integral_constant<int, 2>{}, s + 2);
        if (s + 3 <= s_last) step(std::integral_constant<int, 3>{}, s + 3);
        if (s + 4 <= s_last) step(std::integral_constant<int, 4>{}, s + 4);
    }
}

void launch_scan_anchored(const DevicePlanes &pl, const PerfectLaunch &pp, uint32_t *xa, int64_t xa_stride, hipStream_t stream) {
    const int nm = pp.m_hi - pp.m_lo + 1;
    const int64_t nwords = pl.length / 32 + 1;
    const int hl = anchored_halo_lanes(pp.m_hi);
    const int64_t ntiles = (nwords + anchored_tile_words(hl) - 1) / anchored_tile_words(hl);
    if (nm <= 0 || ntiles <= 0 || pp.m_hi > ANCHORED_MAX_MOTIF) return;
    // each wave recomputes 4 extra shifts around its motif group, so keep the groups large
    int64_t want_y = (1024 + ntiles - 1) / ntiles;
    int max_y = (nm + 31) / 32;
    int gy = (int)(want_y < 1 ? 1 : (want_y > max_y ? max_y : want_y));
    int motifs_per_block = (nm + gy - 1) / gy;
    gy = (nm + motifs_per_block - 1) / motifs_per_block;
    dim3 grid((unsigned)ntiles, (unsigned)gy);
    hipLaunchKernelGGL(scan_anchored_kernel, grid, dim3(256), 0, stream, pl, pp, motifs_per_block, hl, xa, xa_stride);
}

// ------------------------------------------------------------- window scan of composed planes
// The 6-of-8 window scan of processShiftXORsAnchored (parse_anchored_shiftxor.cpp:580-679) on the composed planes XA_m that
// scan_anchored_kernel<false> wrote: per (tile, motif) a lane loads its eight words of the plane and three neighbours, counts
// mismatches (the complement of the plane) over every 8-window bit-sliced, and emits the START / END events of the pass-streaks
// the group filter keeps (see scan_anchored_kernel).  No shifted operands, no anchor ring: ~110 VGPRs, four to five waves per
// SIMD.  Same event conventions as the other scan kernels, tiles of TILE_WORDS words.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan_xa_window_kernel(DevicePlanes pl, PerfectLaunch pp, int motifs_per_block,
                                                             const uint32_t *__restrict__ xa, int64_t xa_stride,
                                                             uint64_t *__restrict__ events, uint32_t *__restrict__ counters,
                                                             const int32_t *__restrict__ tj_table, uint32_t *__restrict__ dropmap) {
    __shared__ uint64_t s_stage[4][EV_STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile_base = (int64_t)blockIdx.x * TILE_WORDS;
    const int bm_lo = pp.m_lo + (int)blockIdx.y * motifs_per_block;
    const int bm_hi = min(pp.m_hi, bm_lo + motifs_per_block - 1);
    const int wm_lo = bm_lo + wave;          // motifs dealt round-robin to the 4 waves
    const int wm_hi = bm_hi;
    if (wm_lo > wm_hi) return;

    const int64_t w0 = tile_base + (int64_t)lane * K;      // global index of this lane's word k = 0
    uint32_t EVAL[K + 2];   // words k = -1 .. K: window starting here holds no break
    {
        uint32_t B[K + 3];
#pragma unroll
        for (int j = 0; j < K + 3; j++) B[j] = pl.brk[w0 - 1 + j];
#pragma unroll
        for (int j = 0; j < K + 2; j++) B[j] |= funnel(B[j + 1], B[j], 1);
        B[K + 2] |= B[K + 2] >> 1;
#pragma unroll
        for (int j = 0; j < K + 2; j++) B[j] |= funnel(B[j + 1], B[j], 2);
        B[K + 2] |= B[K + 2] >> 2;
#pragma unroll
        for (int j = 0; j < K + 2; j++) EVAL[j] = ~(B[j] | funnel(B[j + 1], B[j], 4));
    }
    uint32_t DROP[K];
#pragma unroll
    for (int k = 0; k < K; k++) DROP[k] = 0;
    const int64_t length = pl.length;
    const uint32_t word0 = (uint32_t)w0;

    EventSink sink;
    sink.events = events;
    sink.counters = counters;
    sink.region_cap = pp.ev_cap / EV_SHARDS;
    sink.shard = (blockIdx.x * 4u + (uint32_t)wave + blockIdx.y) % EV_SHARDS;
    volatile uint64_t *stage = s_stage[wave];
    int staged = 0;

    for (int m = wm_lo; m <= wm_hi; m += 4) {
        // mismatch words k = -1 .. K+1: the complement of the composed plane.  The planes have no lead padding: the word before
        // the record reads as all mismatches (no window is evaluated there anyway)
        const uint32_t *plane = xa + (int64_t)(m - pp.m_lo) * xa_stride;
        uint32_t A1[K + 3], B1[K + 3], C1[K + 3];
        A1[0] = w0 > 0 ? ~plane[w0 - 1] : 0xffffffffu;
        {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(plane + w0), v1 = *reinterpret_cast<const uint4 *>(plane + w0 + 4);
            A1[1] = ~v0.x; A1[2] = ~v0.y; A1[3] = ~v0.z; A1[4] = ~v0.w;
            A1[5] = ~v1.x; A1[6] = ~v1.y; A1[7] = ~v1.z; A1[8] = ~v1.w;
        }
        A1[K + 1] = ~plane[w0 + K];
        A1[K + 2] = ~plane[w0 + K + 1];
#pragma unroll
        for (int j = 0; j < K + 3; j++) {          // span 2
            const uint32_t sa = j < K + 2 ? funnel(A1[j + 1], A1[j], 1) : (A1[j] >> 1);
            B1[j] = A1[j] & sa;
            A1[j] = A1[j] | sa;
        }
#pragma unroll
        for (int j = 0; j < K + 3; j++) {          // span 4
            const uint32_t sa = j < K + 2 ? funnel(A1[j + 1], A1[j], 2) : (A1[j] >> 2);
            const uint32_t sb = j < K + 2 ? funnel(B1[j + 1], B1[j], 2) : (B1[j] >> 2);
            C1[j] = (B1[j] & sa) | (A1[j] & sb);
            B1[j] = B1[j] | sb | (A1[j] & sa);
            A1[j] = A1[j] | sa;
        }
        uint32_t PASS[K + 2];   // words k = -1 .. K
#pragma unroll
        for (int j = 0; j < K + 2; j++) {          // span 8: at least three mismatches fail the window
            const uint32_t sa = funnel(A1[j + 1], A1[j], 4);
            const uint32_t sb = funnel(B1[j + 1], B1[j], 4);
            const uint32_t sc = funnel(C1[j + 1], C1[j], 4);
            const uint32_t bad = C1[j] | sc | (B1[j] & sa) | (A1[j] & sb);
            PASS[j] = EVAL[j] & ~bad;
        }
        const int tj = tj_table ? tj_table[m - pp.m_lo] : 0;          // wave-uniform; 0: every group is kept
        uint32_t SV[K + 1];
        if (tj > 0) {
            uint32_t DR[K];
            group_filter(PASS, EVAL, tj, SV, DR);
#pragma unroll
            for (int k = 0; k < K; k++) DROP[k] |= DR[k];
        } else {
#pragma unroll
            for (int j = 0; j < K + 1; j++) SV[j] = 0xffffffffu;
        }
        uint32_t ST[K], EN[K];
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t prev = funnel(PASS[k + 1], PASS[k], 31);
            const uint32_t svprev = funnel(SV[k + 1], SV[k], 31);
            ST[k] = PASS[k + 1] & ~prev & SV[k + 1];
            EN[k] = ~PASS[k + 1] & prev & svprev;
            any |= ST[k] | EN[k];
        }
        if (__ballot(any != 0) != 0ull) {
            stage_events(ST, EN, word0, (uint32_t)m, sink, stage, staged, lane, [&](int k, uint32_t b, uint32_t pos) {
                if ((EVAL[k + 1] >> b) & 1u) return (uint32_t)EV_END_ZERO;
                return ((int64_t)pos + 7 >= length) ? (uint32_t)EV_END_EOS : (uint32_t)EV_END_N;
            });
        }
    }
    sink_flush(sink, stage, staged, lane);
    if (dropmap != nullptr) {
#pragma unroll
        for (int k = 0; k < K; k++)
            if (DROP[k]) atomicOr(&dropmap[w0 + k], DROP[k]);
    }
}

void launch_scan_xa_window(const DevicePlanes &pl, const PerfectLaunch &pp, const uint32_t *xa, int64_t xa_stride, uint64_t *events,
                           uint32_t *counters, const int32_t *tj_table, uint32_t *dropmap, hipStream_t stream) {
    const int nm = pp.m_hi - pp.m_lo + 1;
    if (nm <= 0 || pl.ntiles <= 0) return;
    int64_t want_y = (2048 + pl.ntiles - 1) / pl.ntiles;
    int max_y = (nm + 3) / 4;
    int gy = (int)(want_y < 1 ? 1 : (want_y > max_y ? max_y : want_y));
    int motifs_per_block = (nm + gy - 1) / gy;
    gy = (nm + motifs_per_block - 1) / motifs_per_block;
    hipLaunchKernelGGL(scan_xa_window_kernel, dim3((unsigned)pl.ntiles, (unsigned)gy), dim3(256), 0, stream, pl, pp, motifs_per_block, xa, xa_stride,
                       events, counters, tj_table, dropmap);
}

// ------------------------------------------------------------------------ event compaction
__global__ __launch_bounds__(256) void compact_events_kernel(const uint64_t *__restrict__ events, uint32_t region_cap,
                                                             uint32_t *__restrict__ counters,
                                                             uint64_t *__restrict__ dense) {
    const uint32_t shard = blockIdx.x;
    uint32_t offset = 0, total = 0, overflow = 0, mine = 0;
    for (uint32_t t = 0; t < (uint32_t)EV_SHARDS; ++t) {
        const uint32_t raw = counters[t * EV_COUNTER_STRIDE];
        const uint32_t c = raw < region_cap ? raw : region_cap;
        overflow |= raw > region_cap;
        if (t < shard) offset += c;
        if (t == shard) mine = c;
        total += c;
    }
    const uint64_t *src = events + (size_t)shard * region_cap;
    for (uint32_t i = blockIdx.y * 256u + threadIdx.x; i < mine; i += gridDim.y * 256u) dense[offset + i] = src[i];
    if (shard == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        counters[EV_SUMMARY] = total;
        counters[EV_SUMMARY + 1] = overflow;
    }
}

// (motif, tile) chunk table of the compacted event buffer of a window scan, for the host replay: entry
// (motif - m_lo) * ntile + tile = {index of the chunk's first event, index past its last}.  The host used to find
// the chunk boundaries by walking all events (0.9 G of them for a 250-Mbp record); here every event looks at its
// neighbours instead.  status: bit 0 malformed event, bit 1 a (motif, tile) pair with two chunks.
__global__ __launch_bounds__(256) void chunk_table_kernel(const uint64_t *__restrict__ dense, const uint32_t *__restrict__ counters,
                                                          uint32_t m_lo, uint32_t nm, uint32_t ntile, uint32_t tile_bases,
                                                          uint2 *__restrict__ table, uint32_t *__restrict__ status) {
    const uint32_t n = counters[EV_SUMMARY];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint64_t e = dense[i];
        const uint32_t mi = ev_mlen(e) - m_lo, tile = ev_pos(e) / tile_bases;
        if (mi >= nm || tile >= ntile) { atomicOr(status, 1u); continue; }
        const uint32_t key = mi * ntile + tile;
        bool first = i == 0, last = i + 1 == n;
        if (!first) { const uint64_t p = dense[i - 1]; first = ev_mlen(p) - m_lo != mi || ev_pos(p) / tile_bases != tile; }
        if (!last) { const uint64_t q = dense[i + 1]; last = ev_mlen(q) - m_lo != mi || ev_pos(q) / tile_bases != tile; }
        if (first && atomicExch(&table[key].x, i + 1u) != 0u) atomicOr(status, 2u);      // stored + 1: 0 = no chunk
        if (last) table[key].y = i + 1u;
    }
}

hipError_t launch_chunk_table(const uint64_t *dense, const uint32_t *counters, uint32_t m_lo, uint32_t nm, uint32_t ntile,
                              uint32_t tile_bases, void *table, uint32_t *status, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(table, 0, (size_t)nm * ntile * sizeof(uint2), stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(status, 0, sizeof(uint32_t), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(chunk_table_kernel, dim3(4096), dim3(256), 0, stream, dense, counters, m_lo, nm, ntile, tile_bases,
                       (uint2 *)table, status);
    return hipGetLastError();
}

void launch_compact_events(const uint64_t *events, uint32_t ev_cap, uint32_t *counters, uint64_t *dense,
                           hipStream_t stream) {
    hipLaunchKernelGGL(compact_events_kernel, dim3(EV_SHARDS, 16), dim3(256), 0, stream, events,
                       ev_cap / (uint32_t)EV_SHARDS, counters, dense);
}

// ------------------------------------------------------------- device-side run pairing (a5)
// The perfect scan leaves its events as position-ordered chunks, exactly one per (motif, tile) that has
// any event, scattered over EV_SHARDS regions in arrival order.  These kernels turn them into the run
// records of the perfect stage, ordered by (motif, start), without a sort and without the host:
//   1. chunk_bounds: every event looks at its neighbours; the first/last event of a chunk records the
//      chunk's [begin, end) in a direct-address table keyed (motif, tile);
//   2. chunk_starts + scan: START events per chunk, exclusive prefix sum in (motif, tile) order;
//   3. pair_runs: every START event finds its END (the next event of the chunk, or the first event of
//      the next non-empty tile of the same motif) and writes its run at prefix + index-in-chunk.
// Malformed streams (which would mean a kernel bug) raise PAIR_* flags instead of producing garbage.
struct ChunkEntry { uint32_t begin1, end1; };    // global event index + 1; 0 = no chunk

__device__ __forceinline__ bool chunk_key_of(uint64_t e, const PairLaunch &pl, uint32_t &key) {
    const uint32_t mi = ev_mlen(e) - pl.m_lo;
    const uint32_t tile = ev_pos(e) / pl.tile_bases;
    key = mi * pl.ntile + tile;
    return mi < pl.nm && tile < pl.ntile;
}

__device__ __forceinline__ uint32_t region_count(const uint32_t *counters, uint32_t shard, uint32_t region_cap) {
    const uint32_t raw = counters[shard * EV_COUNTER_STRIDE];
    return raw < region_cap ? raw : region_cap;
}

__global__ __launch_bounds__(256) void pair_chunk_bounds_kernel(const uint64_t *__restrict__ events,
                                                                const uint32_t *__restrict__ counters, PairLaunch pl,
                                                                ChunkEntry *__restrict__ table,
                                                                uint32_t *__restrict__ status) {
    const uint32_t shard = blockIdx.y;
    const uint32_t n = region_count(counters, shard, pl.region_cap);
    const uint64_t *ev = events + (size_t)shard * pl.region_cap;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint64_t e = ev[i];
        uint32_t key, other;
        if (!chunk_key_of(e, pl, key)) { atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_BAD_EVENT); continue; }
        const uint32_t g = shard * pl.region_cap + i;
        const bool first = i == 0 || !chunk_key_of(ev[i - 1], pl, other) || other != key;
        const bool last = i + 1 == n || !chunk_key_of(ev[i + 1], pl, other) || other != key;
        if (first && atomicExch(&table[key].begin1, g + 1u) != 0u) atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_DUP_CHUNK);
        if (last) table[key].end1 = g + 2u;
        // inside a chunk START and END alternate
        if (!first && (ev_kind(ev[i - 1]) == EV_START) == (ev_kind(e) == EV_START))
            atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_NOT_ALTERNATING);
    }
}

// START events of chunk `key` (0 for an empty entry)
__device__ __forceinline__ uint32_t chunk_start_count(const ChunkEntry *table, const uint64_t *events, uint32_t key) {
    const ChunkEntry c = table[key];
    if (c.begin1 == 0u) return 0u;
    const uint32_t cnt = c.end1 - c.begin1;
    return (cnt + (ev_kind(events[c.begin1 - 1u]) == EV_START ? 1u : 0u)) >> 1;
}

constexpr int SCAN_ITEMS = 4;                        // table entries per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t &block_total) {
    __shared__ uint32_t wave_sum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)v);
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { if (w < wave) before += wave_sum[w]; total += wave_sum[w]; }
    __syncthreads();
    block_total = total;
    return before + incl - v;
}

// pass A: per-entry START counts -> run_base[], per-block sums -> partial[]
__global__ __launch_bounds__(256) void pair_chunk_starts_kernel(const ChunkEntry *__restrict__ table,
                                                                const uint64_t *__restrict__ events, uint32_t entries,
                                                                uint32_t *__restrict__ run_base,
                                                                uint32_t *__restrict__ partial) {
    const uint32_t base = blockIdx.x * (uint32_t)SCAN_BLOCK + threadIdx.x * (uint32_t)SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t key = base + (uint32_t)k;
        const uint32_t c = key < entries ? chunk_start_count(table, events, key) : 0u;
        if (key < entries) run_base[key] = c;
        sum += c;
    }
    uint32_t total;
    (void)block_exclusive_scan(sum, total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// pass B: exclusive scan of partial[] by one block; the grand total goes to status[PAIR_TOTAL]
__global__ __launch_bounds__(256) void pair_scan_partials_kernel(uint32_t *__restrict__ partial, uint32_t nblocks,
                                                                 uint32_t *__restrict__ status) {
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? partial[i] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(v, total);
        if (i < nblocks) partial[i] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) status[PAIR_TOTAL] = carry;
}

// pass C: run_base[] counts -> exclusive prefix in (motif, tile) order
__global__ __launch_bounds__(256) void pair_run_base_kernel(uint32_t entries, const uint32_t *__restrict__ partial,
                                                            uint32_t *__restrict__ run_base) {
    const uint32_t base = blockIdx.x * (uint32_t)SCAN_BLOCK + threadIdx.x * (uint32_t)SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        c[k] = base + (uint32_t)k < entries ? run_base[base + (uint32_t)k] : 0u;
        sum += c[k];
    }
    uint32_t total;
    uint32_t at = partial[blockIdx.x] + block_exclusive_scan(sum, total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + (uint32_t)k < entries) run_base[base + (uint32_t)k] = at;
        at += c[k];
    }
}

struct RunRecord { int32_t start, end, mlen, term; };    // == RibbitRun (ribbit_hip.h)

__global__ __launch_bounds__(256) void pair_runs_kernel(const uint64_t *__restrict__ events,
                                                        const uint32_t *__restrict__ counters, PairLaunch pl,
                                                        const ChunkEntry *__restrict__ table,
                                                        const uint32_t *__restrict__ run_base,
                                                        RunRecord *__restrict__ runs, uint32_t run_cap,
                                                        RunRecord *__restrict__ halves, uint32_t half_cap,
                                                        uint32_t *__restrict__ status) {
    const uint32_t shard = blockIdx.y;
    const uint32_t n = region_count(counters, shard, pl.region_cap);
    const uint64_t *ev = events + (size_t)shard * pl.region_cap;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint64_t e = ev[i];
        uint32_t key;
        if (!chunk_key_of(e, pl, key)) continue;                        // flagged by chunk_bounds
        const ChunkEntry c = table[key];
        const uint32_t g1 = shard * pl.region_cap + i + 1u;             // this event's index + 1
        if (c.begin1 == 0u || g1 < c.begin1 || g1 >= c.end1) continue;  // duplicate chunk, flagged by chunk_bounds
        const uint32_t mi = key / pl.ntile, tile = key - mi * pl.ntile;
        const bool chunk_opens_with_end = ev_kind(events[c.begin1 - 1u]) != EV_START;
        if (ev_kind(e) != EV_START) {
            // the END that opens a chunk closes a run begun in an earlier tile: there must be one
            if (g1 == c.begin1) {
                bool ok = false;
                for (uint32_t t = tile; t-- > 0u;) {
                    const ChunkEntry p = table[mi * pl.ntile + t];
                    if (p.begin1 == 0u) continue;
                    ok = ev_kind(events[p.end1 - 2u]) == EV_START;
                    break;
                }
                if (!ok) atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_NOT_ALTERNATING);
            }
            continue;
        }
        // partner: the next event of this chunk, or the first event of the next non-empty tile of this motif
        uint64_t closer = 0;
        bool found = false;
        if (g1 + 1u < c.end1) { closer = ev[i + 1]; found = true; }
        else {
            for (uint32_t t = tile + 1u; t < pl.ntile; ++t) {
                const ChunkEntry nx = table[key + (t - tile)];
                if (nx.begin1 == 0u) continue;
                closer = events[nx.begin1 - 1u];
                found = true;
                break;
            }
        }
        if (!found) { atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_UNTERMINATED); continue; }
        const uint32_t kind = ev_kind(closer);
        if (kind == EV_START || ev_pos(closer) <= ev_pos(e)) { atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_NOT_ALTERNATING); continue; }
        const uint32_t in_chunk = (g1 - c.begin1 - (chunk_opens_with_end ? 1u : 0u)) >> 1;
        const uint32_t slot = run_base[key] + in_chunk;
        if (slot >= run_cap) { atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_NO_ROOM); continue; }
        // RIBBIT_TERM_ZERO / _N / _EOS = EV_END_ZERO / _N / _EOS - 1.  When the loaded piece is one chunk of a longer
        // record (own range narrower than the piece), a run is reported by the chunk that owns its START; if its END
        // lies beyond the own range the record is an open half, and the chunk that owns the END reports the other half.
        const int64_t rs = (int64_t)ev_pos(e), re = (int64_t)ev_pos(closer);
        const int32_t mlen = (int32_t)(mi + pl.m_lo), term = (int32_t)kind - 1;
        const bool start_owned = rs >= pl.own_lo && rs < pl.own_hi, end_owned = re >= pl.own_lo && re < pl.own_hi;
        if (start_owned && re < pl.own_hi) {
            runs[slot] = RunRecord{(int32_t)(rs + pl.pos_offset), (int32_t)(re + pl.pos_offset), mlen, term};
            continue;
        }
        runs[slot] = RunRecord{0, 0, 0, RUN_NOT_OWNED};
        if (!start_owned && !end_owned) continue;
        // at most two halves per motif and chunk: a plain atomic append is enough
        const uint32_t at = atomicAdd(&status[PAIR_HALVES], 1u);
        if (at >= half_cap) { atomicOr(&status[PAIR_FLAGS], (uint32_t)PAIR_NO_ROOM); continue; }
        halves[at] = start_owned ? RunRecord{(int32_t)(rs + pl.pos_offset), -1, mlen, RUN_HALF_START}
                                 : RunRecord{-1, (int32_t)(re + pl.pos_offset), mlen, RUN_HALF_END + term};
    }
}

// Region counters and pairing status -> page-locked host memory, written by the GPU itself: a copy command for
// these 300 bytes would queue behind another handle's multi-megabyte result copy on the same DMA engine and
// hold up the compute stream.
__global__ __launch_bounds__(128) void pair_publish_kernel(const uint32_t *__restrict__ counters,
                                                           const uint32_t *__restrict__ status,
                                                           uint32_t *__restrict__ host_words) {
    const uint32_t t = threadIdx.x;
    if (t < (uint32_t)EV_SHARDS) host_words[t] = counters[t * EV_COUNTER_STRIDE];
    else if (t < (uint32_t)EV_SHARDS + PAIR_STATUS_WORDS) host_words[t] = status[t - (uint32_t)EV_SHARDS];
    __threadfence_system();
}

void launch_pair_publish(const uint32_t *counters, const uint32_t *status, uint32_t *host_words, hipStream_t stream) {
    hipLaunchKernelGGL(pair_publish_kernel, dim3(1), dim3(128), 0, stream, counters, status, host_words);
}

hipError_t launch_pair_runs(const uint64_t *events, const uint32_t *counters, const PairLaunch &pl, void *table,
                            uint32_t *run_base, uint32_t *partial, void *runs, uint32_t run_cap, void *halves,
                            uint32_t half_cap, uint32_t *status, hipStream_t stream) {
    const uint32_t entries = pl.nm * pl.ntile;
    const uint32_t nblocks = (entries + (uint32_t)SCAN_BLOCK - 1u) / (uint32_t)SCAN_BLOCK;
    const uint32_t per_region = std::min<uint32_t>(std::max<uint32_t>((pl.region_cap + 255u) / 256u, 1u), 64u);
    hipError_t e = hipMemsetAsync(table, 0, (size_t)entries * sizeof(ChunkEntry), stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(status, 0, PAIR_STATUS_WORDS * sizeof(uint32_t), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(pair_chunk_bounds_kernel, dim3(per_region, EV_SHARDS), dim3(256), 0, stream, events, counters, pl,
                       (ChunkEntry *)table, status);
    hipLaunchKernelGGL(pair_chunk_starts_kernel, dim3(nblocks), dim3(256), 0, stream, (const ChunkEntry *)table, events,
                       entries, run_base, partial);
    hipLaunchKernelGGL(pair_scan_partials_kernel, dim3(1), dim3(256), 0, stream, partial, nblocks, status);
    hipLaunchKernelGGL(pair_run_base_kernel, dim3(nblocks), dim3(256), 0, stream, entries, (const uint32_t *)partial, run_base);
    hipLaunchKernelGGL(pair_runs_kernel, dim3(per_region, EV_SHARDS), dim3(256), 0, stream, events, counters, pl,
                       (const ChunkEntry *)table, (const uint32_t *)run_base, (RunRecord *)runs, run_cap,
                       (RunRecord *)halves, half_cap, status);
    return hipGetLastError();
}

// ------------------------------------------------------------------- plane query (a5, a13)
// X_shift words for host-side range reads: retainNestedSeed / retainIdenticalSeeds
// (parse_perfect_shiftxor.cpp:18-43) and the seed bit extraction of fasta_utils.cpp:220-222.
__global__ __launch_bounds__(256) void plane_words_kernel(DevicePlanes pl, int shift, int64_t w0, int64_t nw,
                                                          uint32_t *__restrict__ out_words, int64_t p0, int64_t p1,
                                                          uint32_t *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bits = 0;
    if (i < nw) {
        const int64_t w = w0 + i;
        const int q = shift >> 5;
        const uint32_t r = (uint32_t)shift & 31u;
        const uint32_t hs = funnel(pl.hi[w + q + 1], pl.hi[w + q], r);
        const uint32_t ls = funnel(pl.lo[w + q + 1], pl.lo[w + q], r);
        const uint32_t x = ~((pl.hi[w] ^ hs) | (pl.lo[w] ^ ls));
        if (out_words) out_words[i] = x;
        if (count) {
            // mask to [p0, p1)
            const int64_t lo_p = w * 32, hi_p = lo_p + 32;
            uint32_t mask = 0xffffffffu;
            if (p0 > lo_p) mask &= (p0 >= hi_p) ? 0u : (0xffffffffu << (uint32_t)(p0 - lo_p));
            if (p1 < hi_p) mask &= (p1 <= lo_p) ? 0u : (0xffffffffu >> (uint32_t)(hi_p - p1));
            bits = (uint32_t)__popc(x & mask);
        }
    }
    if (count) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) bits += __shfl_down(bits, d);
        if ((threadIdx.x & 63) == 0 && bits) atomicAdd(count, bits);
    }
}

void launch_plane_words(const DevicePlanes &pl, int shift, int64_t w0, int64_t nw, uint32_t *out_words, int64_t p0,
                        int64_t p1, uint32_t *count, hipStream_t stream) {
    if (nw <= 0) return;
    const int64_t blocks = (nw + 255) / 256;
    hipLaunchKernelGGL(plane_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pl, shift, w0, nw, out_words,
                       p0, p1, count);
}

// ------------------------------------------------------- longestContinuousMatches, batched (a13)
// parse_seed.cpp:26-44 for every dispatched seed at once: longest run of ones of the composed plane
// XA_mlen over [start, end).  One thread per seed (seeds are a few words long); word-parallel:
// leading / trailing ones by ctz / clz, the longest interior run by the y &= y << 1 peel.
__global__ __launch_bounds__(256) void seed_longest_run_kernel(const uint32_t *__restrict__ xa, int64_t xa_stride, int m_lo,
                                                               const int4 *__restrict__ seeds, int64_t n,
                                                               int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 seed = seeds[i];                       // start, end, mlen, type
    const int start = seed.x, end = seed.y;
    int best = 0;
    if (end > start) {
        const uint32_t *w = xa + (int64_t)(seed.z - m_lo) * xa_stride;
        const int w0 = start >> 5, w1 = (end - 1) >> 5;
        int run = 0;
        for (int k = w0; k <= w1; ++k) {
            uint32_t x = w[k];
            if (k == w0) x &= 0xffffffffu << (start & 31);
            if (k == w1) { const int top = ((end - 1) & 31) + 1; if (top < 32) x &= (1u << top) - 1u; }
            if (x == 0xffffffffu) { run += 32; best = max(best, run); continue; }
            best = max(best, run + (int)__builtin_ctz(~x));
            int inner = 0;
            for (uint32_t y = x; y; y &= y << 1) ++inner;
            best = max(best, inner);
            run = __builtin_clz(~x);
        }
    }
    out[i] = best;
}

void launch_seed_longest_runs(const uint32_t *xa, int64_t xa_stride, int m_lo, const void *seeds, int64_t n, int32_t *out,
                              hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(seed_longest_run_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, xa, xa_stride, m_lo,
                       (const int4 *)seeds, n, out);
}

// ------------------------------------------------- mostFrequentLongerMotif, batched (a15)
// parse_seed.cpp:153-256 for every dispatched seed with m > 10 at once.  Every window
// row_start .. row_start+m-1 of a seed is scored independently (walk down- and upstream in steps of
// m with a +-2 jitter, count identical bases on the best diagonal), so the rows are the parallel
// axis: one thread per row, 64 rows per wavefront-sized workgroup, a long seed spread over as many
// workgroups as it has 64-row slices (blocks[] maps a workgroup to its seed and first row; a row costs
// ~5 x seed length byte compares, so one workgroup per whole seed made the longest seed of a record the
// critical path of the launch).  The seed's best (score, smallest row) is kept with a 64-bit atomicMax.
// sym = one byte per base: 0..3 = A C G T, 4 = N.
__global__ __launch_bounds__(256) void sym_kernel(const uint8_t *__restrict__ ascii, int64_t length, uint8_t *__restrict__ sym) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= length) return;
    const uint32_t c = ascii[i], u = c | 0x20u;
    const bool valid = (u == 'a') | (u == 'c') | (u == 'g') | (u == 't');
    sym[i] = valid ? (uint8_t)(((c >> 1) & 3u) ^ ((c >> 2) & 1u)) : (uint8_t)4;
}

__device__ __forceinline__ int diag_matches(const uint8_t *__restrict__ sym, int row0, int col0, int lo, int hi, int n, int step) {
    int d = 0;
    for (int i = 0; i < n; ++i) {
        const int col = col0 + step * i;
        if (col >= hi || col < lo) break;
        const uint8_t a = sym[col];
        d += (a == sym[row0 + step * i]) & (a < 4);
    }
    return d;
}

// The five jittered diagonals of one step, eight symbols at a time: matches[x + 2] = diag_matches(sym, row0, col0 + x, lo, hi,
// n, 1) for x = -2 .. 2.  A symbol is 0..4, so the bytes of (a ^ b) + 0x7f..7f have bit 7 clear exactly where the symbols are
// equal (no carry between bytes), and a column symbol is a base where its bit 2 is clear.  The walk of the scalar form ends
// at the first column outside [lo, hi): for an increasing column that is "none at all if the first is below lo, else the
// first hi - col0".  Loads run up to 7 bytes past what is counted: sym is padded by 16 bytes, and whatever is there is masked
// (a carry out of such a byte only reaches bytes further up, which are masked too).
__device__ __forceinline__ void diag_matches5(const uint8_t *__restrict__ sym, int row0, int col0, int lo, int hi, int n, int (&matches)[5]) {
    int cnt[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) {
        const int c0 = col0 + x - 2;
        cnt[x] = c0 < lo ? 0 : max(0, min(n, hi - c0));
        matches[x] = 0;
    }
    const int most = max(max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])), cnt[4]);
    for (int i = 0; i < most; i += 8) {
        uint64_t b;
        __builtin_memcpy(&b, sym + row0 + i, 8);
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            const int left = cnt[x] - i;
            if (left <= 0) continue;
            uint64_t a;
            __builtin_memcpy(&a, sym + col0 + x - 2 + i, 8);
            uint64_t eq = ~((a ^ b) + 0x7f7f7f7f7f7f7f7full) & ~(a << 5) & 0x8080808080808080ull;
            if (left < 8) eq &= (1ull << (8 * left)) - 1ull;
            matches[x] += __popcll(eq);
        }
    }
}

__global__ __launch_bounds__(64) void long_motif_rows_kernel(const uint8_t *__restrict__ sym, int64_t length,
                                                             const int4 *__restrict__ jobs, int64_t njobs,
                                                             const int2 *__restrict__ blocks,
                                                             unsigned long long *__restrict__ best) {
    const int2 blk = blocks[blockIdx.x];       // job, first row of this slice (relative to seed_start)
    const int64_t job = blk.x;
    if (job < 0 || job >= njobs) return;
    const int4 jb = jobs[job];                 // seed_start, seed_sequence_length, m, unused
    const int seed_start = jb.x, m = jb.z;
    int seed_end = jb.x + jb.y;
    if (seed_end > (int)length) seed_end = (int)length;
    unsigned long long mine = 0;
    const int row = seed_start + blk.y + (int)threadIdx.x;
    if (row < seed_end - m + 1) {
        int score = 0;
        int d5[5];
        for (int col = row + m; col < seed_end;) {
            int pick = -2, top = 0;
            diag_matches5(sym, row, col, INT_MIN, seed_end, m, d5);
#pragma unroll
            for (int x = -2; x <= 2; ++x)
                if (d5[x + 2] > top) { top = d5[x + 2]; pick = x; }
            score += top;
            col += pick + m;
        }
        int col = row - m;
        for (; col > seed_start;) {
            int pick = -2, top = 0;
            diag_matches5(sym, row, col, 0, INT_MAX, m, d5);
#pragma unroll
            for (int x = -2; x <= 2; ++x)
                if (d5[x + 2] > top) { top = d5[x + 2]; pick = x; }
            score += top;
            col += pick - m;
        }
        if (col < seed_start && abs(col - seed_start) < m) {
            const int rows = m + (col - seed_start);
            int top = 0;
            for (int x = -2; x <= 2; ++x)
                top = max(top, diag_matches(sym, row + m - 1, seed_start + rows - 1 + x, seed_start, seed_end, rows, -1));
            score += top;
        }
        // larger score wins; among equal scores the smallest row (the reference keeps the first strict maximum)
        const unsigned long long key = ((unsigned long long)(unsigned)score << 32) | (0xffffffffu - (unsigned)row);
        if (score > 0 && key > mine) mine = key;
    }
    // one atomic per workgroup: reduce over the wave first
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long other = __shfl_down(mine, d);
        if (other > mine) mine = other;
    }
    if (threadIdx.x == 0 && mine) atomicMax(&best[job], mine);
}

void launch_sym(const uint8_t *ascii, int64_t length, uint8_t *sym, hipStream_t stream) {
    if (length <= 0) return;
    hipLaunchKernelGGL(sym_kernel, dim3((unsigned)((length + 255) / 256)), dim3(256), 0, stream, ascii, length, sym);
}

void launch_long_motif_rows(const uint8_t *sym, int64_t length, const void *jobs, int64_t njobs, const void *blocks,
                            int64_t nblocks, unsigned long long *best, hipStream_t stream) {
    if (njobs <= 0 || nblocks <= 0) return;
    hipLaunchKernelGGL(long_motif_rows_kernel, dim3((unsigned)nblocks), dim3(64), 0, stream, sym, length, (const int4 *)jobs, njobs,
                       (const int2 *)blocks, best);
}

// ------------------------------------------------------------------- PMC calibration
// Streams `nwords` dwords with the same access shape as the scan kernels' staging loads (one
// coalesced dword per lane).  rocprofv3's FETCH_SIZE is only calibrated for 16-byte-per-lane
// streams on gfx950 (MI355X_MICROARCH.md, HBM); running this on a known byte count in the same
// profiling pass gives the correction factor for our access width.
__global__ __launch_bounds__(256) void calib_stream_read_kernel(const uint32_t *__restrict__ src, int64_t nwords,
                                                                uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * 256) acc ^= src[i];
    if (acc == 0x9e3779b9u) sink[0] = acc;   // practically never true; keeps the loads alive
}

void launch_calib_stream_read(const uint32_t *src, int64_t nwords, uint32_t *sink, hipStream_t stream) {
    hipLaunchKernelGGL(calib_stream_read_kernel, dim3(8192), dim3(256), 0, stream, src, nwords, sink);
}

}  // namespace rb
