// anchored_merge.hip -- the anchored stage's seed-list merge (addSeedToSeedPositionsAnchored,
// parse_anchored_shiftxor.cpp:113-534, with mergeAllLists, merge_types.cpp:11-189, and retainNestedSeedAnchored,
// parse_anchored_shiftxor.cpp:59-84) as per-range GPU work: one LANE per independent position range of the stage's kept
// calls (parallel_merge.h says what makes ranges independent and what crosses a cut), ~10^5 ranges of a few hundred
// calls for a chromosome.  It is the device twin of rb::anchored_add / merge_all_lists / AnchoredReplay (seed_lists.cpp,
// parallel_merge.cpp): the host versions stay the definition (and the path of small records, of ranges done again and of
// every later pass), the tests hold both against the oracle's lists.
//
// A merge call is a chain of dependent reads and branches with nothing to vectorise inside it; what the GPU has is 10^5
// such chains, the lists and the composed planes XA_m (12 bytes per base, already in HBM).  Written as the host writes it
// -- nested loops per call -- a wavefront pays, for every call, the LONGEST candidate walk among its 64 lanes: the walks
// have a heavy tail (3 candidates on average, hundreds at a dense locus), and a first version of this kernel took as long
// as sixteen host threads.  So every lane runs a STATE MACHINE instead: one pass of the wavefront's loop is one small
// step of each lane's own call (one cursor step, one step of a candidate walk, one candidate judged), whatever state the
// other lanes are in, and a lane whose call is done fetches its next call at once.  A wavefront's time is then the
// largest SUM of steps among its lanes, which averages out, not the sum of the largest.  Every state reads at most two
// list entries, so one pass issues two 16-byte loads for all lanes together, wherever each lane's entries are; the
// candidate lists live in LDS.
//
// What a range leaves behind is what the host's validation walk needs (parallel_merge.cpp: RangeState): its part of the
// anchored list, the retirements it made in the shared lists (undo log), the types it read of seeds left of its range
// (foreign reads), Q8's writes to list heads (logged, not made) and by-counter reads, its guard count and final cursors.
// A range whose candidate lists overflow their LDS (a hundred candidates for one call) reports that instead: the host
// merges it.  Types are read with plain loads: a stale type of a seed LEFT of the range is exactly what the log and the
// validation walk are for, and nobody else writes the seeds inside a range.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "kernels.h"
#include "ribbit_hip.h"

namespace rb {

namespace {

constexpr int RN = RIBBIT_RANK_N, RA = RIBBIT_RANK_A, RC = RIBBIT_RANK_C, RS = RIBBIT_RANK_S, RQ = RIBBIT_RANK_Q, RP = RIBBIT_RANK_P;

enum : uint32_t { FROM_P = 0u, FROM_S = 1u, FROM_A = 2u };
__device__ __forceinline__ uint32_t cand(uint32_t src, uint32_t idx) { return (src << 30) | idx; }
__device__ __forceinline__ uint32_t cand_src(uint32_t c) { return c >> 30; }
__device__ __forceinline__ uint32_t cand_idx(uint32_t c) { return c & 0x3fffffffu; }

// seedlen_cutoffs of processShiftXORsAnchored (parse_anchored_shiftxor.cpp:572-573; seed_lists.h)
__device__ __forceinline__ int anchored_cutoff(int mlen) { return mlen >= 10 ? (int)(0.9 * (double)mlen) : (mlen > 6 ? mlen : 10); }

constexpr int AM_WALK = 8;      // entries of the own list per pass of a lane that is walking back over it
enum : int { ST_FETCH = 0, ST_CATCHUP, ST_CURSOR, ST_PHASE1, ST_PHASE2, ST_CANDS, ST_TAIL, ST_NEXT, ST_DONE };

// x % d == 0 for 0 < d and 0 <= x < 2^20 (motif sizes): the candidate loop asks a dozen times per candidate, and an integer
// remainder is forty instructions.  q is the rounded quotient from a 1-ulp reciprocal -- exact whenever d divides x -- and the
// test itself is in integers.
__device__ __forceinline__ bool divides(int d, int x) {
    const int q = (int)((float)x * __frcp_rn((float)d) + 0.5f);
    return q * d == x;
}

// a 16-byte list entry (or call record) from global memory, whichever list it is in
typedef int Int4V __attribute__((ext_vector_type(4)));
typedef const Int4V __attribute__((address_space(1))) *GlobalInt4;
__device__ __forceinline__ RibbitSeed load_entry(const void *p) {
    const Int4V v = *(GlobalInt4)(uintptr_t)p;
    return RibbitSeed{v.x, v.y, v.z, v.w};
}

// entry d of four, field by field (a select between whole structs would put them in scratch memory)
__device__ __forceinline__ RibbitSeed pick4(int d, RibbitSeed e0, RibbitSeed e1, RibbitSeed e2, RibbitSeed e3) {
    RibbitSeed r;
    r.start = d == 0 ? e0.start : d == 1 ? e1.start : d == 2 ? e2.start : e3.start;
    r.end = d == 0 ? e0.end : d == 1 ? e1.end : d == 2 ? e2.end : e3.end;
    r.mlen = d == 0 ? e0.mlen : d == 1 ? e1.mlen : d == 2 ? e2.mlen : e3.mlen;
    r.type = d == 0 ? e0.type : d == 1 ? e1.type : d == 2 ? e2.type : e3.type;
    return r;
}

// One wavefront = 64 ranges.  See the head of the file for the shape of the loop; the states follow rb::anchored_add
// (seed_lists.cpp) and AnchoredReplay (parallel_merge.cpp) block by block, with their line references.
__global__ __launch_bounds__(64) void anchored_merge_kernel(AnchoredMergeArgs a) {
    __shared__ uint32_t ps[AM_PS_CAP * 64], cands[AM_CAND_CAP * 64];      // entry j of this lane at [j * 64 + lane]
    const uint32_t lane = threadIdx.x;
    // the range this lane is merging: taken from a counter, so that a lane whose range is done takes the next one (ranges differ
    // several-fold in work: a dense locus costs a hundred steps per call) -- a fixed number of wavefronts stays resident
    uint32_t kk = 0, first = 0, last = 0;
    unsigned long long clock0 = 0;
    RibbitSeed *const P = a.P, *const S = a.S;
    const uint64_t nP = a.nP, nS = a.nS;
    const int32_t *const P_type0 = a.P_type0, *const S_type0 = a.S_type0;
    const uint32_t *const xa = a.xa;
    const int64_t xa_stride = a.xa_stride;
    const int m_lo = a.m_lo, m_hi = a.m_hi;
    uint32_t *const log = a.log, *const log_count = a.log_count;
    const uint32_t log_cap = a.log_cap;
    int range_lo = 0, range_hi = 0;
    RibbitSeed *A = a.own;                                                // this range's part of the anchored list
    uint32_t *const fch = a.scratch + (size_t)(blockIdx.x * 64u + lane) * AM_SCRATCH_WORDS, *const nfch = fch + AM_CHILD_CAP;      // factor / non-factor children
    int32_t *const cov = reinterpret_cast<int32_t *>(nfch + AM_CHILD_CAP);      // coverage table: AM_COV_CAP x {motif size, previous start, coverage}
    uint32_t *const ps_spill = nfch + AM_CHILD_CAP + 3 * AM_COV_CAP, *const cand_spill = ps_spill + AM_PS_SPILL;
    uint32_t nA = 0, n_ps = 0, n_cands = 0, n_f = 0, n_nf = 0, status = 0;
    long long guard_hits = 0;
    unsigned long long head_reads0 = 0, head_reads1 = 0;

    auto log4 = [&](uint32_t kind, uint32_t x, uint32_t y) __attribute__((always_inline)) {
        const uint32_t at = atomicAdd(log_count, 1u);
        if (at < log_cap) {
            uint4 e; e.x = (kind << 28) | kk; e.y = x; e.z = y; e.w = 0u;
            reinterpret_cast<uint4 *>(log)[at] = e;
        } else status |= AM_LOG_FULL;
    };
    auto seed_ptr = [&](uint32_t c) __attribute__((always_inline)) -> const RibbitSeed * {
        const uint32_t src = cand_src(c), idx = cand_idx(c);
        return src == FROM_P ? P + idx : src == FROM_S ? S + idx : A + idx;
    };
    // retire_shared / live_shared of seed_lists.cpp
    auto retire_shared = [&](uint32_t list, uint32_t idx) __attribute__((always_inline)) {
        RibbitSeed *s = (list ? S : P) + idx;
        const int32_t was = s->type;
        if (was != RN) log4(AM_LOG_UNDO, (list << 31) | idx, (uint32_t)was);
        s->type = RN;
    };
    auto live_shared = [&](uint32_t list, uint32_t idx, const RibbitSeed &s) __attribute__((always_inline)) -> bool {      // s: the entry as loaded
        if (s.start > range_hi) return (list ? S_type0 : P_type0)[idx] != RN;
        const bool live = s.type != RN;
        if (s.end < range_lo) log4(AM_LOG_READ, (list << 31) | idx, live ? 1u : 0u);
        return live;
    };
    // popcount of XA_mlen over [start, end)  (HostPlanes::range_count_xa)
    auto range_count = [&](int m, int start, int end) __attribute__((always_inline)) -> int {
        if (end <= start) return 0;
        if (m < m_lo || m > m_hi || start < 0) { status |= AM_BAD_PLANE; return 0; }
        const uint32_t *w = xa + (int64_t)(m - m_lo) * xa_stride;
        const int64_t w0 = start >> 5, w1 = (end - 1) >> 5;
        int total = 0;
        for (int64_t q = w0; q <= w1; ++q) {
            uint32_t x = w[q];
            if (q == w0) x &= 0xffffffffu << (start & 31);
            if (q == w1) { const int hi_bits = ((end - 1) & 31) + 1; if (hi_bits < 32) x &= (1u << hi_bits) - 1u; }
            total += __popc(x);
        }
        return total;
    };
    // retainNestedSeedAnchored (parse_anchored_shiftxor.cpp:59-84): keep the nested seed unless the parent plane has strictly more matches
    auto keep_nested = [&](int start, int end, int nested_mlen, int parent_mlen) __attribute__((always_inline)) -> bool {
        return !(range_count(nested_mlen, start, end) < range_count(parent_mlen, start, end));
    };
    // the candidate lists of a call: the first AM_PS_CAP / AM_CAND_CAP entries in LDS, what a dense locus has beyond in global memory
    auto push_ps = [&](uint32_t c) __attribute__((always_inline)) {
        if (n_ps < (uint32_t)AM_PS_CAP) ps[64u * n_ps++ + lane] = c;
        else if (n_ps < (uint32_t)(AM_PS_CAP + AM_PS_SPILL)) ps_spill[n_ps++ - AM_PS_CAP] = c;
        else status |= AM_SCRATCH_FULL;
    };
    auto push_cand = [&](uint32_t c) __attribute__((always_inline)) {
        if (n_cands < (uint32_t)AM_CAND_CAP) cands[64u * n_cands++ + lane] = c;
        else if (n_cands < (uint32_t)(AM_CAND_CAP + AM_CAND_SPILL)) cand_spill[n_cands++ - AM_CAND_CAP] = c;
        else status |= AM_SCRATCH_FULL;
    };
    auto ps_at = [&](uint32_t j) __attribute__((always_inline)) -> uint32_t { return j < (uint32_t)AM_PS_CAP ? ps[64u * j + lane] : ps_spill[j - AM_PS_CAP]; };
    auto cand_at = [&](uint32_t j) __attribute__((always_inline)) -> uint32_t { return j < (uint32_t)AM_CAND_CAP ? cands[64u * j + lane] : cand_spill[j - AM_CAND_CAP]; };
    // the coverage table of :471-526 (the reference's two unordered_maps; a missing key reads as 0)
    auto cov_slot = [&](uint32_t &n_cov, int key, int prev_if_new) __attribute__((always_inline)) -> int {
        for (uint32_t q = 0; q < n_cov; ++q) if (cov[3 * q] == key) return (int)q;
        if (n_cov >= (uint32_t)AM_COV_CAP) { status |= AM_SCRATCH_FULL; return 0; }
        cov[3 * n_cov] = key; cov[3 * n_cov + 1] = prev_if_new; cov[3 * n_cov + 2] = 0;
        return (int)n_cov++;
    };

    // AnchoredReplay (parallel_merge.cpp): the cursor pair between calls; calls that only moved the cursors are folded into one advance
    uint32_t i = 0;
    int cur_p = 0, cur_s = 0;
    int pending_end = -1;
    // the call being added (anchored_add's arguments as its tail recursions change them) and the cursors of its current round
    int seed_start = 0, seed_end = 0, mlen = 0, seed_type = 0, from_p = 0, from_s = 0, cp = 0, cs = 0;
    uint32_t rounds = 0;
    int o_start = 0, o_end = 0, o_rend = 0, o_mlen = 0, o_type = 0;      // function-scope state the coverage code reads (Q8)
    long pi = 0, si = 0, ci = 0, ai = 0;
    bool p_done = false, s_done = false, c_done = false, a_done = false, restart = false;
    uint32_t ic = 0;
    int state = ST_NEXT;
    uint32_t passes = 0;

    // (transitions shared by several states)
    auto begin_round = [&]() __attribute__((always_inline)) {                                                               // :133-151 both cursors restart from the caller's values
        if (++rounds > (uint32_t)AM_MAX_ROUNDS) { status |= AM_RUNAWAY; return; }
        cp = from_p; cs = from_s;
        state = ST_CURSOR;
    };
    auto finish_call = [&]() __attribute__((always_inline)) { cur_p = cp; cur_s = cs; state = ST_FETCH; };
    auto begin_cands = [&]() __attribute__((always_inline)) {
        n_f = 0; n_nf = 0; ic = 0; restart = false;
        state = n_cands ? ST_CANDS : ST_TAIL;
    };
    auto begin_phase2 = [&]() __attribute__((always_inline)) {
        n_cands = 0;
        if (nA == 0) { for (uint32_t j = 0; j < n_ps; ++j) push_cand(ps_at(j)); begin_cands(); return; }     // :103-106
        ai = (long)nA - 1;
        if (n_ps == 0) { c_done = true; a_done = false; }                                  // :107-122 (the walk of :143-157 alone)
        else { ci = (long)n_ps - 1; c_done = false; a_done = false; }
        state = ST_PHASE2;
    };

    // what the range leaves for the host (kernels.h: AnchoredMergeArgs::range_out)
    auto range_done = [&]() __attribute__((always_inline)) {
        uint32_t *o = a.range_out + (size_t)kk * AM_RANGE_OUT_WORDS;
        o[0] = nA; o[1] = status;
        o[2] = (uint32_t)(guard_hits & 0xffffffffll); o[3] = (uint32_t)((unsigned long long)guard_hits >> 32);
        o[4] = (uint32_t)cur_p; o[5] = (uint32_t)cur_s;
        o[6] = (uint32_t)head_reads0; o[7] = (uint32_t)(head_reads0 >> 32);
        o[8] = (uint32_t)head_reads1; o[9] = (uint32_t)(head_reads1 >> 32);
        o[10] = passes;                                       // (profiling: passes of the loop below, and ticks of the constant 100-MHz counter)
        o[11] = (uint32_t)(wall_clock64() - clock0);
        state = ST_NEXT;
    };
    while (__ballot(state != ST_DONE) != 0ull) {
        if (++passes > a.max_passes && state != ST_DONE && state != ST_NEXT) status |= AM_TOO_SLOW;      // a dense locus: quicker on a host thread than on a lane
        if (status && state != ST_DONE && state != ST_NEXT) range_done();       // (the range is the host's: on to the next one)
        // ---- the two list entries this lane's state looks at, loaded for all lanes together
        const RibbitSeed *pa = nullptr, *pb = nullptr;
        uint32_t cand_now = 0;
        switch (state) {
            case ST_FETCH: if (i < last) pa = reinterpret_cast<const RibbitSeed *>(a.calls + i); break;
            case ST_CATCHUP: if ((uint64_t)cur_p < nP) pa = P + cur_p; if ((uint64_t)cur_s < nS) pb = S + cur_s; break;
            case ST_CURSOR: if ((uint64_t)cp < nP) pa = P + cp; if ((uint64_t)cs < nS) pb = S + cs; break;
            case ST_PHASE1: if (!p_done) pa = P + pi; if (!s_done) pb = S + si; break;
            case ST_PHASE2: if (!c_done) { cand_now = ps_at((uint32_t)ci); pa = seed_ptr(cand_now); } if (!a_done) pb = A + ai; break;
            case ST_CANDS: cand_now = cand_at(ic); pa = seed_ptr(cand_now); break;
            default: break;
        }
        RibbitSeed ea{0, 0, 0, 0}, eb{0, 0, 0, 0};
        if (pa) ea = load_entry(pa);
        if (pb) eb = load_entry(pb);
        // the walk back over the range's own list (:143-157) meets every entry that reaches the call, retired or not -- dozens at a
        // dense locus, and most passes of this loop if taken one by one: AM_WALK entries per pass, their loads in flight together
        RibbitSeed ew[AM_WALK - 1];
        const bool own_walk = state == ST_PHASE2 && c_done && !a_done;
#pragma unroll
        for (int t = 1; t < AM_WALK; ++t) {
            ew[t - 1] = RibbitSeed{0, 0, 0, 0};
            if (own_walk && ai - t >= 0) ew[t - 1] = load_entry(A + (ai - t));
        }
        // ... and the walks back over the perfect and substitution lists (phase 1), which meet every retired seed too: four steps per pass
        RibbitSeed pw[3], sw[3];
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            pw[t - 1] = RibbitSeed{0, 0, 0, 0}; sw[t - 1] = RibbitSeed{0, 0, 0, 0};
            if (state == ST_PHASE1 && !p_done && pi - t >= 0) pw[t - 1] = load_entry(P + (pi - t));
            if (state == ST_PHASE1 && !s_done && si - t >= 0) sw[t - 1] = load_entry(S + (si - t));
        }

        switch (state) {
        case ST_FETCH: {
            if (i >= last) { range_done(); break; }
            // (a RibbitCall through the seed-shaped load: pos | mlen | start | end; pos holds the call's cursor bound, anchored_merge_prepare_kernel)
            const int c_pend = ea.start, c_mlen = ea.end, c_start = ea.mlen, c_end = ea.type;
            ++i;
            if (c_pend > pending_end) pending_end = c_pend;
            if (c_end - c_start < anchored_cutoff(c_mlen)) { if (c_end > pending_end) pending_end = c_end; break; }
            seed_start = c_start; seed_end = c_end; mlen = c_mlen; seed_type = RA; rounds = 0;
            o_start = 0; o_end = 0; o_rend = 0; o_mlen = 0; o_type = 0;
            if (pending_end >= 0) state = ST_CATCHUP;
            else { from_p = cur_p; from_s = cur_s; begin_round(); }
            break;
        }
        case ST_CATCHUP: {                                                                   // AnchoredReplay::catch_up
            bool moved = false;
            if ((uint64_t)cur_p < nP && ea.start <= pending_end && (uint64_t)cur_p != nP - 1) { ++cur_p; moved = true; }
            if ((uint64_t)cur_s < nS && eb.start <= pending_end && (uint64_t)cur_s != nS - 1) { ++cur_s; moved = true; }
            if (!moved) { pending_end = -1; from_p = cur_p; from_s = cur_s; begin_round(); }
            break;
        }
        case ST_CURSOR: {                                                                    // :133-151
            bool moved = false;
            if ((uint64_t)cp < nP && ea.start <= seed_end && (uint64_t)cp != nP - 1) { ++cp; moved = true; }
            if ((uint64_t)cs < nS && eb.start <= seed_end && (uint64_t)cs != nS - 1) { ++cs; moved = true; }
            if (moved) break;
            if (seed_end - seed_start < anchored_cutoff(mlen)) { finish_call(); break; }     // :153
            // mergeAllLists (merge_types.cpp:11-189), as rb::merge_all_lists: phase 1, perfect and substitution seeds by descending end
            n_ps = 0;
            p_done = nP == 0; s_done = false;
            if (nS == 0) { s_done = true; ++guard_hits; }
            pi = cp; si = cs;
            if (p_done && s_done) begin_phase2(); else state = ST_PHASE1;
            break;
        }
        case ST_PHASE1: {
            int dp = 0, ds = 0;                              // entries of this pass's loads taken so far
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                if ((p_done && s_done) || status) break;
                const RibbitSeed pe = pick4(dp, ea, pw[0], pw[1], pw[2]);
                const RibbitSeed se = pick4(ds, eb, sw[0], sw[1], sw[2]);
                if (s_done) {                                                                // :30-45
                    const int end = pe.end;
                    if (end >= seed_start) { if (live_shared(0, (uint32_t)pi, pe)) push_ps(cand(FROM_P, (uint32_t)pi)); --pi; ++dp; }
                    if (pi < 0 || end < seed_start) p_done = true;
                } else if (p_done) {                                                         // :47-62
                    const int end = se.end;
                    if (end >= seed_start) { if (live_shared(1, (uint32_t)si, se)) push_ps(cand(FROM_S, (uint32_t)si)); --si; ++ds; }
                    if (si < 0 || end < seed_start) s_done = true;
                } else {                                                                     // :64-93
                    const int p_end = pe.end, s_end = se.end;
                    if (s_end > p_end) { if (live_shared(1, (uint32_t)si, se)) push_ps(cand(FROM_S, (uint32_t)si)); --si; ++ds; }
                    else { if (live_shared(0, (uint32_t)pi, pe)) push_ps(cand(FROM_P, (uint32_t)pi)); --pi; ++dp; }
                    if (pi < 0 || p_end < seed_start) p_done = true;
                    if (si < 0 || s_end < seed_start) s_done = true;
                }
            }
            if (p_done && s_done) begin_phase2();
            break;
        }
        case ST_PHASE2: {                                                                    // :124-187
            if (a_done) {                                                                    // :127-140
                const int end = ea.end;
                if (end >= seed_start) { push_cand(cand_now); --ci; }
                if (ci < 0 || end < seed_start) c_done = true;
            } else if (c_done) {                                                             // :143-157 (and :107-122)
#pragma unroll
                for (int t = 0; t < AM_WALK; ++t) {
                    if (a_done) break;
                    RibbitSeed e = eb;
                    if (t > 0) e = ew[t > 0 ? t - 1 : 0];      // (t is a constant of the unrolled loop)
                    const int end = e.end;
                    if (end >= seed_start) { if (e.type != RN) push_cand(cand(FROM_A, (uint32_t)ai)); --ai; }
                    if (ai < 0 || end < seed_start) a_done = true;
                }
            } else {                                                                         // :160-185
                const int c_end = ea.end, a_end = eb.end;
                if (a_end > c_end) { push_cand(cand(FROM_A, (uint32_t)ai)); --ai; }        // no RANK_N test here (:168-171)
                else { push_cand(cand_now); --ci; }
                if (ci < 0 || c_end < seed_start) c_done = true;
                if (ai < 0 || a_end < seed_start) a_done = true;
            }
            if (c_done && a_done) begin_cands();
            break;
        }
        case ST_CANDS: {
            // one candidate of the loop at parse_anchored_shiftxor.cpp:196-438.  leave: 0 next candidate, 1 the loop ends (break), 2 the call returns
            const uint32_t idx = cand_idx(cand_now);
            const int seed_rend = seed_end + mlen, seed_len = seed_end - seed_start, seed_rlen = seed_len + mlen;
            int leave = 0;
#define RB_AGAIN(s_, e_, m_, t_) { const int s2_ = (s_), e2_ = (e_), m2_ = (m_), t2_ = (t_); seed_start = s2_; seed_end = e2_; mlen = m2_; seed_type = t2_; restart = true; leave = 1; }
#define RB_RETIRE_BY_TYPE(type_, idx_) { if ((type_) == RP) retire_shared(0, (idx_)); else if ((type_) == RS || (type_) == RQ) retire_shared(1, (idx_)); }
            o_start = ea.start; o_mlen = ea.mlen; o_end = ea.end; o_rend = o_end + o_mlen; o_type = ea.type;
            do {
                if (o_end < seed_start) { leave = 1; break; }                                // :203
                if (o_type == RN) break;                                                     // :205
                if (seed_end < o_start) break;                                               // :208
                const int o_len = o_end - o_start, o_rlen = o_rend - o_start;
                const bool same_rank = (seed_type == RA && o_type == RA) || (seed_type == RC && o_type == RC);

                if (seed_start == o_start && seed_end == o_end) {                            // :215 identical
                    if (seed_type == RA && o_type > RA) { leave = 2; break; }
                    if (seed_type == RC && o_type == RA) A[idx].type = RN;
                } else if (o_start <= seed_start && seed_end <= o_end) {                     // :231 new inside old
                    if (o_type > seed_type) { leave = 2; break; }
                    if (seed_type == RC && o_type == RA) break;
                    if (same_rank) {
                        if (divides(o_mlen, mlen) && mlen != 4) { leave = 2; break; }           // :241
                        if (divides(mlen, o_mlen) && o_mlen != 4) {                             // :246
                            if (seed_rlen >= o_mlen - 1 || seed_rlen >= o_len) {
                                A[idx].type = RN;
                                RB_AGAIN(o_start, o_end, mlen, seed_type);
                            }
                            break;
                        }
                        if (!keep_nested(seed_start, seed_end, mlen, o_mlen)) { leave = 2; break; }   // :257
                        break;
                    }
                } else if (seed_start <= o_start && o_end <= seed_end) {                     // :265 old inside new
                    if (o_type > seed_type) {
                        if (divides(o_mlen, mlen)) {                                            // :268
                            if (o_rlen >= mlen - 2 || o_rlen >= seed_len - 2) {
                                RB_RETIRE_BY_TYPE(o_type, idx);
                                RB_AGAIN(seed_start, seed_end, o_mlen, RC);
                                break;
                            }
                            if (n_f < (uint32_t)AM_CHILD_CAP) fch[n_f++] = (uint32_t)o_mlen | ((uint32_t)(o_type & 0xff) << 16); else status |= AM_SCRATCH_FULL;
                        } else if (divides(mlen, o_mlen) || o_mlen > mlen) {                    // :285 / :301 (same action)
                            if (o_mlen >= 4 * mlen || o_len >= 4 * mlen) {
                                RB_RETIRE_BY_TYPE(o_type, idx);
                                RB_AGAIN(seed_start, seed_end, mlen, RC);
                                break;
                            }
                        } else {                                                             // :312
                            if (n_nf < (uint32_t)AM_CHILD_CAP) nfch[n_nf++] = (uint32_t)o_mlen | ((uint32_t)(o_type & 0xff) << 16); else status |= AM_SCRATCH_FULL;
                        }
                    } else if (seed_type == RC && o_type == RA) {                            // :319
                        A[idx].type = RN;
                    } else if (same_rank) {                                                  // :323
                        if (o_mlen == mlen || !keep_nested(o_start, o_end, o_mlen, mlen)) {
                            A[idx].type = RN;
                        } else if (divides(o_mlen, mlen)) {                                     // :332
                            if (o_rlen >= mlen - 2 || o_rlen >= seed_len - 2) {
                                A[idx].type = RN;
                                RB_AGAIN(seed_start, seed_end, o_mlen, seed_type);
                                break;
                            }
                        }
                    }
                } else {                                                                     // :351 partial overlap
                    int overlap, ms, me;
                    if (o_start < seed_start) {
                        const int reach = (o_mlen <= mlen) ? o_rend : o_end;
                        overlap = (seed_end <= reach ? seed_end : reach) - seed_start;
                        ms = o_start; me = seed_end;
                    } else {
                        const int reach = (mlen <= o_mlen) ? seed_rend : seed_end;
                        overlap = (o_end <= reach ? o_end : reach) - o_start;
                        ms = seed_start; me = o_end;
                    }
                    if (seed_type == RA && o_type > RC) {                                    // :376
                        if (mlen == o_mlen && overlap >= 4 * mlen) {
                            RB_RETIRE_BY_TYPE(o_type, idx);
                            RB_AGAIN(ms, me, mlen, RC);
                            break;
                        }
                        if (!(divides(o_mlen, mlen) || divides(mlen, o_mlen)) && (overlap >= mlen - 1 || overlap >= seed_len - 1)) { leave = 2; break; }
                    } else if ((seed_type == RA || seed_type == RC) && (o_type == RA || o_type == RC)) {   // :398
                        if (mlen == o_mlen) {
                            bool merge;
                            if (o_len >= seed_len)
                                merge = (seed_len >= 3 * mlen) ? (overlap >= 3 * mlen - 1 || overlap >= seed_len - 1)
                                                               : (overlap >= mlen - 1 || overlap >= seed_len - 1);
                            else
                                merge = (o_len >= 3 * o_mlen && (overlap >= 3 * o_mlen - 1 || overlap >= o_len - 1)) ||
                                        (!(o_len >= 3 * o_mlen && (overlap >= 3 * o_mlen - 1 || overlap >= o_len - 1)) &&
                                         seed_len < 3 * o_mlen && (overlap >= o_mlen - 1 || overlap >= o_len - 1));
                            if (merge) {
                                A[idx].type = RN;
                                RB_AGAIN(ms, me, o_mlen, seed_type);
                                break;
                            }
                        }
                    }
                }
            } while (false);
#undef RB_AGAIN
#undef RB_RETIRE_BY_TYPE
            if (leave == 2) { finish_call(); break; }
            if (leave == 0 && ++ic < n_cands) break;
            if (restart) begin_round(); else state = ST_TAIL;
            break;
        }
        case ST_TAIL: {
            bool returned = false;
            // :441-468 coverage by non-factor children; Q8: the lists are indexed with the LOOP COUNTER j
            if (n_nf) {
                int coverage = 0;
                uint32_t prev_start = 0xffffffffu;
                for (uint32_t j = 0; j < n_nf; ++j) {
                    const int t = (int)(nfch[j] >> 16);
                    const int which = t == RP ? 0 : t == RS ? 1 : -1;
                    if (which >= 0) {
                        const RibbitSeed *src = which ? S : P;
                        if ((uint64_t)j < (which ? nS : nP)) {
                            { const unsigned long long bit = 1ull << (j < 63u ? j : 63u); if (which) head_reads1 |= bit; else head_reads0 |= bit; }
                            o_start = src[j].start; o_mlen = src[j].mlen; o_end = src[j].end; o_rend = o_end + o_mlen;
                        } else ++guard_hits;
                    }
                    if ((uint32_t)o_rend >= prev_start) coverage = (int)((uint32_t)coverage + (prev_start - (uint32_t)o_start));
                    else if (o_rend < seed_end) coverage += o_rend - o_start;
                    else coverage += seed_end - o_start;
                    prev_start = (uint32_t)o_start;
                }
                if ((double)coverage > 0.5 * (double)(seed_end - seed_start)) returned = true;   // :467
            }
            // :471-526 coverage by factor children, per child motif size
            if (!returned && n_f) {
                const int seed_len = seed_end - seed_start;
                uint32_t n_cov = 0;
                for (uint32_t j = 0; j < n_f; ++j) {
                    const int s = cov_slot(n_cov, (int)(fch[j] & 0xffffu), -1);
                    cov[3 * s + 1] = -1; cov[3 * s + 2] = 0;
                }
                for (uint32_t j = 0; j < n_f && !status; ++j) {
                    const int t = (int)(fch[j] >> 16);
                    const int which = t == RP ? 0 : t == RS ? 1 : -1;
                    if (which >= 0) {
                        const RibbitSeed *src = which ? S : P;
                        if ((uint64_t)j < (which ? nS : nP)) {
                            { const unsigned long long bit = 1ull << (j < 63u ? j : 63u); if (which) head_reads1 |= bit; else head_reads0 |= bit; }
                            o_start = src[j].start; o_mlen = src[j].mlen; o_end = src[j].end; o_rend = o_end + o_mlen;
                        } else ++guard_hits;
                    }
                    const int s = cov_slot(n_cov, o_mlen, 0);
                    const uint32_t prev_start = (uint32_t)cov[3 * s + 1];
                    int cv = cov[3 * s + 2];
                    if ((uint32_t)o_rend >= prev_start) cv = (int)((uint32_t)cv + (prev_start - (uint32_t)o_start));
                    else if (o_rend < seed_end) cv += o_rend - o_start;
                    else cv += seed_end - o_start;
                    cov[3 * s + 2] = cv;
                    cov[3 * s + 1] = o_start;
                }
                // ascending factor sizes (:504-507): the smallest size whose coverage reaches 0.8 of the seed
                int f = 0x7fffffff;
                for (uint32_t q = 0; q < n_cov; ++q)
                    if ((double)cov[3 * q + 2] >= 0.8 * (double)seed_len && cov[3 * q] < f) f = cov[3 * q];
                if (f != 0x7fffffff && !status) {
                    mlen = f; seed_type = RC;                                                // :509
                    for (uint32_t j = 0; j < n_f; ++j) {                                   // :511-522, stale start/end written back
                        const int t = (int)(fch[j] >> 16);
                        const int which = t == RP ? 0 : t == RS ? 1 : -1;
                        if (which < 0) continue;
                        if ((uint64_t)j >= (which ? nS : nP)) { ++guard_hits; continue; }
                        { const unsigned long long bit = 1ull << (j < 63u ? j : 63u); if (which) head_reads1 |= bit; else head_reads0 |= bit; }
                        const RibbitSeed now = (which ? S : P)[j];
                        o_mlen = now.mlen;
                        if (o_mlen == f) {                                                   // logged, not made (parallel pass)
                            // e[7]: the write would change the entry as it is NOW, and the entry lies in this range (ListRefs::HeadWrite::changed_then)
                            const bool mine = now.end >= range_lo && now.start <= range_hi;
                            const bool changes = now.start != o_start || now.end != o_end || now.type != RN;
                            const uint32_t at = atomicAdd(a.head_count, 1u);
                            if (at < a.head_cap) {
                                uint32_t *e = a.head_log + 8 * (size_t)at;
                                e[0] = kk; e[1] = (uint32_t)which; e[2] = j; e[3] = (uint32_t)o_start; e[4] = (uint32_t)o_end; e[5] = (uint32_t)o_mlen; e[6] = (uint32_t)RN;
                                e[7] = mine && changes ? 1u : 0u;
                            } else status |= AM_LOG_FULL;
                        }
                    }
                }
            }
            if (!returned && !status) {
                const int limit = (int)a.length - mlen;                                      // :529-531
                if (seed_end > limit) seed_end = limit;
                A[nA++] = RibbitSeed{seed_start, seed_end, mlen, seed_type};
            }
            finish_call();
            break;
        }
        case ST_NEXT: {
            // (the host threads take ranges from the back of the same list meanwhile: a.sync[0] = entries still left to the lanes,
            // a.sync[1] = entries the lanes have taken, both in page-locked host memory; a range both sides take is the host's)
            const uint32_t q = atomicAdd(a.next_range, 1u);
            if (q >= a.n_order || q >= __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { state = ST_DONE; break; }
            __hip_atomic_store(a.sync + 1, q + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (lanes' stores may land out of order: the host may then see less than was taken, and take a range twice)
            const uint32_t k = a.order[q];
            kk = k; first = a.first[k]; last = a.first[k + 1];
            range_lo = a.cut_pos[k]; range_hi = k + 1u < a.nr ? a.cut_pos[k + 1] : 0x7fffffff;
            A = a.own + (size_t)first + k;
            nA = 0;
            if (k > 0) A[nA++] = RibbitSeed{-1, -1, 0, RN};      // stands for everything the earlier ranges appended (parallel_merge.cpp: SENTINEL)
            status = 0; guard_hits = 0; head_reads0 = 0; head_reads1 = 0; passes = 0;
            i = first; cur_p = a.cur0[2 * k]; cur_s = a.cur0[2 * k + 1]; pending_end = -1;
            clock0 = wall_clock64();
            state = ST_FETCH;
            break;
        }
        default: break;
        }
    }
}

// the cursor bound of every kept call (KeptCalls::pend; -1 without one) into the call record's `pos`, which the merge does
// not read: a lane then fetches a call with one 16-byte load
__global__ __launch_bounds__(256) void anchored_merge_prepare_kernel(RibbitCall *__restrict__ calls, const int32_t *__restrict__ pend, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) calls[i].pos = pend ? pend[i] : -1;
}

__global__ __launch_bounds__(256) void seed_types_kernel(const RibbitSeed *__restrict__ seeds, uint32_t n, int32_t *__restrict__ types) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) types[i] = seeds[i].type;
}

}  // namespace

void launch_seed_types(const RibbitSeed *seeds, uint32_t n, int32_t *types, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(seed_types_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, seeds, n, types);
}

void launch_anchored_merge(const AnchoredMergeArgs &a, uint32_t n_calls, uint32_t resident_waves, hipStream_t stream) {
    if (a.nr == 0 || a.n_order == 0) return;
    if (n_calls) hipLaunchKernelGGL(anchored_merge_prepare_kernel, dim3((n_calls + 255u) / 256u), dim3(256), 0, stream, a.calls, a.pend, n_calls);
    // as many wavefronts as the chip holds at once (the candidate lists' LDS: five per CU), each lane taking ranges until none is left
    const uint32_t waves = (a.n_order + 63u) / 64u;
    hipLaunchKernelGGL(anchored_merge_kernel, dim3(waves < resident_waves ? waves : resident_waves), dim3(64), 0, stream, a);
}

}  // namespace rb
