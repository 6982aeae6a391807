/*
 * ribbit_oracle.c -- CPU restatement of ribbit's shift-XOR tandem-repeat scan.
 *
 * TEST INFRASTRUCTURE ONLY (see ribbit_oracle.h).  PARITY UNPINNED: no reference
 * golden vectors exist and the reference cannot be built in this image.
 *
 * Every function cites the reference lines (SowpatiLab/ribbit @ 2024_10_08) it
 * restates.  The code is deliberately naive: one loop iteration per (base, motif),
 * the same shape as the reference, so that it can be audited against it, not so
 * that it is fast.  Codes and the N mask are one byte per base; the shift-XOR
 * planes are one BIT per base behind pl_get / pl_set (round 4: they were bytes,
 * 204 of them per base at -M 100 -- 62 GB for a chromosome-1-sized record and a
 * kilobyte per base at -M 500, which kept full-size checks off the build box).
 * rbo_plane() still hands the tests a byte-per-base view, made on request.
 *
 * Defined divergences from the reference (both are undefined behaviour there):
 *   D1  merge_types.cpp:47-68 reads seed_positions_substut[idx] even when that
 *       list is empty.  Here an empty list is treated as already exhausted.
 *   D2  parse_anchored_shiftxor.cpp:449-458,484-494,515-520 index the seed
 *       lists with a loop counter j that can exceed the list size.  Here an
 *       out-of-range j leaves the stale values in place / skips the write.
 * rbo_guard_hits() counts how often either guard fired, so that parity fixtures
 * can assert they stay clear of both regimes.
 */
#include "ribbit_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vectors */

typedef struct { rbo_seed_t *a; int64_t n, cap; } seedvec;
typedef struct { rbo_call_t *a; int64_t n, cap; } callvec;
typedef struct { int *a; int64_t n, cap; } intvec;

static void sv_push(seedvec *v, int s, int e, int m, int t) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 64;
        v->a = (rbo_seed_t *)realloc(v->a, (size_t)v->cap * sizeof(rbo_seed_t));
    }
    v->a[v->n].start = s; v->a[v->n].end = e; v->a[v->n].mlen = m; v->a[v->n].type = t;
    v->n++;
}
static void sv_erase(seedvec *v, int64_t i) {
    memmove(v->a + i, v->a + i + 1, (size_t)(v->n - i - 1) * sizeof(rbo_seed_t));
    v->n--;
}
static void cv_push(callvec *v, int pos, int m, int s, int e) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 64;
        v->a = (rbo_call_t *)realloc(v->a, (size_t)v->cap * sizeof(rbo_call_t));
    }
    v->a[v->n].pos = pos; v->a[v->n].mlen = m; v->a[v->n].start = s; v->a[v->n].end = e;
    v->n++;
}
static void iv_push(intvec *v, int x) {
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 16;
        v->a = (int *)realloc(v->a, (size_t)v->cap * sizeof(int));
    }
    v->a[v->n++] = x;
}
static void iv_free(intvec *v) { free(v->a); v->a = NULL; v->n = v->cap = 0; }

/* ------------------------------------------------------------------ context */

struct rbo_ctx {
    int64_t L;
    int m_lo, m_hi, nmotifs;            /* MINIMUM_MLEN, MAXIMUM_MLEN, NMOTIFS */
    int min_shift, max_shift, nshifts;  /* ribbit.cpp:240-243 */
    uint8_t *code, *nmask;
    uint64_t **plane;                   /* [nshifts] lshift_xor_bsets, bit p & 63 of word p >> 6 */
    uint64_t **anchor;                  /* [nshifts] lsxor_anchor_bsets */
    uint8_t **view[2];                  /* byte-per-base copies handed out by rbo_plane / rbo_anchor_plane */
    seedvec lists[3];
    callvec calls[3];
    seedvec dispatch;
    int64_t guard_hits;
    int64_t range_queries;      /* how often the merges asked for a range popcount (statistics only) */
};

static inline int pl_get(const uint64_t *x, int64_t p) { return (int)((x[p >> 6] >> (p & 63)) & 1); }
static inline void pl_set(uint64_t *x, int64_t p) { x[p >> 6] |= (uint64_t)1 << (p & 63); }
static size_t pl_words(int64_t len) { return (size_t)(len / 64 + 1); }

/* byte-per-base copy of a bit plane, kept until the planes change or the context is closed */
static const uint8_t *byte_view(rbo_ctx *c, int which, int idx, const uint64_t *bits) {
    if (!c->view[which]) c->view[which] = (uint8_t **)calloc((size_t)c->nshifts, sizeof(uint8_t *));
    if (!c->view[which][idx]) {
        uint8_t *v = (uint8_t *)malloc((size_t)c->L + 1);
        for (int64_t p = 0; p < c->L; p++) v[p] = (uint8_t)pl_get(bits, p);
        c->view[which][idx] = v;
    }
    return c->view[which][idx];
}
static void drop_views(rbo_ctx *c) {
    for (int w = 0; w < 2; w++) {
        if (!c->view[w]) continue;
        for (int i = 0; i < c->nshifts; i++) free(c->view[w][i]);
        free(c->view[w]); c->view[w] = NULL;
    }
}

int rbo_min_shift(const rbo_ctx *c) { return c->min_shift; }
int rbo_max_shift(const rbo_ctx *c) { return c->max_shift; }
int64_t rbo_length(const rbo_ctx *c) { return c->L; }
const uint8_t *rbo_nmask(const rbo_ctx *c) { return c->nmask; }
const uint8_t *rbo_codes(const rbo_ctx *c) { return c->code; }
const uint8_t *rbo_plane(rbo_ctx *c, int shift) {
    if (shift < c->min_shift || shift > c->max_shift) return NULL;
    return byte_view(c, 0, shift - c->min_shift, c->plane[shift - c->min_shift]);
}
const uint8_t *rbo_anchor_plane(rbo_ctx *c, int shift) {
    if (!c->anchor || shift < c->min_shift || shift > c->max_shift) return NULL;
    return byte_view(c, 1, shift - c->min_shift, c->anchor[shift - c->min_shift]);
}
const uint64_t *rbo_plane_bits(const rbo_ctx *c, int shift) {
    if (shift < c->min_shift || shift > c->max_shift) return NULL;
    return c->plane[shift - c->min_shift];
}
int64_t rbo_seeds(const rbo_ctx *c, int which, const rbo_seed_t **out) {
    *out = c->lists[which].a; return c->lists[which].n;
}
int64_t rbo_calls(const rbo_ctx *c, int which, const rbo_call_t **out) {
    *out = c->calls[which].a; return c->calls[which].n;
}
int64_t rbo_dispatch(const rbo_ctx *c, const rbo_seed_t **out) {
    *out = c->dispatch.a; return c->dispatch.n;
}
int64_t rbo_guard_hits(const rbo_ctx *c) { return c->guard_hits; }
int64_t rbo_range_queries(const rbo_ctx *c) { return c->range_queries; }

/* ------------------------------------------------------- encode + sweep (a1, a2) */

/*
 * fasta_utils.cpp:78-115 (2-bit encode: A/a 00, C/c 01, G/g 10, T/t 11, anything else sets the
 * N mask and encodes as 00) and fasta_utils.cpp:117-122 (shift-XOR sweep).  The reference computes
 * ~(left ^ (left<<s)) & ~(right ^ (right<<s)) on bitsets stored reversed, so bit p of plane s is
 * code[p]==code[p+s]; Boost's << zero-fills, so for the last s positions the comparison partner
 * is 00.  Shift range: ribbit.cpp:240-243.
 */
rbo_ctx *rbo_open(const char *seq, int64_t len, int m_lo, int m_hi) {
    rbo_ctx *c = (rbo_ctx *)calloc(1, sizeof(rbo_ctx));
    c->L = len; c->m_lo = m_lo; c->m_hi = m_hi;
    c->nmotifs = m_hi - m_lo + 1;
    c->min_shift = (m_lo > 2) ? m_lo - 2 : 1;
    c->max_shift = m_hi + 2;
    c->nshifts = c->max_shift - c->min_shift + 1;
    c->code = (uint8_t *)malloc((size_t)len + 1);
    c->nmask = (uint8_t *)malloc((size_t)len + 1);
    for (int64_t p = 0; p < len; p++) {
        uint8_t code = 0, n = 0;
        switch (seq[p]) {
            case 'A': case 'a': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': code = 3; break;
            default: n = 1; break;
        }
        c->code[p] = code; c->nmask[p] = n;
    }
    c->plane = (uint64_t **)calloc((size_t)c->nshifts, sizeof(uint64_t *));
    for (int i = 0; i < c->nshifts; i++) {
        int s = c->min_shift + i;
        uint64_t *x = (uint64_t *)calloc(pl_words(len), sizeof(uint64_t));
        for (int64_t p = 0; p < len; p++) {
            uint8_t partner = (p + s <= len - 1) ? c->code[p + s] : 0;
            if (c->code[p] == partner) pl_set(x, p);
        }
        c->plane[i] = x;
    }
    return c;
}

void rbo_close(rbo_ctx *c) {
    if (!c) return;
    for (int i = 0; i < c->nshifts; i++) {
        free(c->plane[i]);
        if (c->anchor) free(c->anchor[i]);
    }
    drop_views(c);
    free(c->plane); free(c->anchor); free(c->code); free(c->nmask);
    for (int k = 0; k < 3; k++) { free(c->lists[k].a); free(c->calls[k].a); }
    free(c->dispatch.a);
    free(c);
}

/* parse_perfect_shiftxor.cpp:18-29 / parse_anchored_shiftxor.cpp:59-70: the counting loop only */
int rbo_range_count(const rbo_ctx *c, int shift, int start, int end) {
    const uint64_t *x = c->plane[shift - c->min_shift];
    int n = 0;
    for (int i = start; i < end; i++) if (pl_get(x, i) == 1) n++;
    return n;
}

/* retainNestedSeed, parse_perfect_shiftxor.cpp:18-29 (== retainNestedSeedAnchored :59-70) */
static int retain_nested(rbo_ctx *c, int start, int end, int nested_mlen, int parent_mlen) {
    c->range_queries += 2;
    int nested = rbo_range_count(c, nested_mlen, start, end);
    int parent = rbo_range_count(c, parent_mlen, start, end);
    return !(nested < parent);
}
/* retainIdenticalSeeds, parse_perfect_shiftxor.cpp:31-43: tie goes to the smaller plane index */
static int retain_identical(rbo_ctx *c, int start, int end, int nested_mlen, int parent_mlen) {
    c->range_queries += 2;
    int nested = rbo_range_count(c, nested_mlen, start, end);
    int parent = rbo_range_count(c, parent_mlen, start, end);
    if (nested < parent) return 0;
    if (nested == parent) return nested_mlen < parent_mlen;
    return 1;
}

/* ---------------------------------------------------------- perfect stage (a3, a4) */

/* addSeedToSeedPositionsPerfect, parse_perfect_shiftxor.cpp:47-142 */
static void perfect_add(rbo_ctx *c, int seed_start, int seed_end, int mlen) {
    seedvec *v = &c->lists[RBO_LIST_PERFECT];
    const int bset_size = (int)c->L;
    const int seed_rlen = seed_end - seed_start + mlen;
    intvec doomed = {0};

    for (int64_t i = v->n - 1; i >= 0; i--) {                                   /* :60 */
        const int old_start = v->a[i].start, old_end = v->a[i].end, old_mlen = v->a[i].mlen;
        const int old_rlen = (old_end - old_start) + old_mlen;
        int overlap;

        if (old_end < seed_start) break;                                        /* :70 */

        if (old_start == seed_start && old_end == seed_end) {                   /* :73 identical */
            if (old_mlen < mlen) { iv_free(&doomed); return; }
            iv_push(&doomed, (int)i);
        } else if (old_start <= seed_start && old_end >= seed_end) {            /* :79 nested */
            if (seed_rlen < old_mlen / 3) continue;
            iv_free(&doomed); return;
        } else if (seed_start <= old_start && seed_end >= old_end) {            /* :85 parent */
            if (old_rlen < mlen / 3) continue;
            iv_push(&doomed, (int)i);
        } else {                                                                /* :91 overlap */
            int merge_start, merge_end;
            if (old_start < seed_start) {
                overlap = old_end - seed_start + old_mlen; merge_start = old_start; merge_end = seed_end;
            } else {
                overlap = seed_end - old_start + mlen; merge_start = seed_start; merge_end = old_end;
            }
            if (old_mlen == mlen) {                                             /* :96 */
                iv_free(&doomed);
                perfect_add(c, merge_start, merge_end, old_mlen);
                return;
            } else if (old_mlen < mlen) {                                       /* :102 */
                if (mlen - overlap <= 1 && seed_rlen / mlen < 3) {
                    iv_free(&doomed);
                    perfect_add(c, merge_start, merge_end, old_mlen);
                    return;
                } else if (seed_rlen - mlen - overlap <= old_mlen) {
                    iv_free(&doomed); return;
                }
            } else {                                                            /* :115 mlen < old_mlen */
                if (old_mlen - overlap <= 1 && old_rlen / old_mlen < 3) {
                    iv_free(&doomed);
                    perfect_add(c, merge_start, merge_end, old_mlen);
                    return;
                } else if (old_rlen - old_mlen - overlap <= mlen) {
                    iv_push(&doomed, (int)i);
                }
            }
        }
    }

    for (int64_t k = 0; k < doomed.n; k++) sv_erase(v, doomed.a[k]);             /* :129-134 */
    iv_free(&doomed);

    if (seed_end > bset_size - mlen) seed_end = bset_size - mlen;               /* :137-139 */
    sv_push(v, seed_start, seed_end, mlen, RBO_RANK_P);
}

/* processShiftXORsPerfect, parse_perfect_shiftxor.cpp:146-226 */
int rbo_run_perfect(rbo_ctx *c) {
    const int L = (int)c->L, nm = c->nmotifs;
    const int min_idx = c->m_lo - c->min_shift;
    int *open_start = (int *)calloc((size_t)nm, sizeof(int));
    open_start[0] = -1;           /* `int last_starts[NMOTIFS] = {-1};` -> {-1, 0, 0, ...}  (:161) */
    callvec *log = &c->calls[RBO_LIST_PERFECT];
    int pos = 0;

    for (int p = 0; p < L; p++) {                                               /* :173 */
        if (c->nmask[p]) {                                                      /* :175 */
            for (int d = 0; d < nm; d++) {
                const int midx = min_idx + d, mlen = c->min_shift + midx;
                const int cutoff = (mlen <= 6) ? 12 - mlen : mlen + midx;       /* :179 */
                if (open_start[d] != -1) {
                    if (pos - open_start[d] >= cutoff) {
                        cv_push(log, p, mlen, open_start[d], pos);
                        perfect_add(c, open_start[d], pos, mlen);
                    }
                    open_start[d] = -1;
                }
            }
        } else {
            for (int d = 0; d < nm; d++) {                                      /* :191 */
                const int midx = min_idx + d, mlen = c->min_shift + midx;
                const int cutoff = (mlen <= 6) ? 12 - mlen : mlen;              /* :193 */
                if (pl_get(c->plane[midx], p)) {
                    if (open_start[d] == -1) open_start[d] = pos;
                } else {
                    if (open_start[d] != -1) {
                        if (pos - open_start[d] >= cutoff) {
                            cv_push(log, p, mlen, open_start[d], pos);
                            perfect_add(c, open_start[d], pos, mlen);
                        }
                    }
                    open_start[d] = -1;
                }
            }
        }
        pos += 1;
    }

    pos -= 1;                                                                   /* :213 end = L-1 */
    for (int d = 0; d < nm; d++) {
        const int midx = min_idx + d, mlen = c->min_shift + midx;
        const int cutoff = (mlen <= 6) ? 12 - mlen : mlen;
        if (open_start[d] != -1) {
            if (pos - open_start[d] >= cutoff) {
                cv_push(log, L, mlen, open_start[d], pos);
                perfect_add(c, open_start[d], pos, mlen);
            }
            open_start[d] = -1;
        }
    }
    free(open_start);
    return (int)c->lists[RBO_LIST_PERFECT].n;
}

/* ------------------------------------------------------ substitution stage (a6, a7) */

#define TAG_N(v, i) ((v)->a[(i)].type = RBO_RANK_N)   /* `list[i] = {start, end, mlen, RANK_N}` */

/* addSeedToSeedPositionsSubstitutions, parse_substitute_shiftxor.cpp:18-388 */
static int subst_add(rbo_ctx *c, int seed_start, int seed_end, int mlen, const int *seedlen_cutoff,
                     int from_index, int seed_type) {
    seedvec *perf = &c->lists[RBO_LIST_PERFECT], *sub = &c->lists[RBO_LIST_SUBST];
    const int bset_size = (int)c->L;
    int old_start = 0, old_end = 0, old_rend = 0, old_mlen = 0, old_len, old_rlen, old_type = 0;

    /* :34-42 advance the cursor over the perfect list */
    for (int64_t i = from_index; i < perf->n; i++) {
        old_start = perf->a[i].start;
        if (old_start > seed_end) break;
        else if ((int64_t)from_index == perf->n - 1) break;
        else from_index += 1;
    }

    if (seed_end - seed_start < seedlen_cutoff[mlen - c->m_lo]) return from_index;   /* :44 */

    /* :48-116 candidates from both lists, larger end first */
    intvec cand_type = {0}, cand_idx = {0};
    int more_perf = perf->n != 0, more_sub = sub->n != 0;
    int64_t pi = from_index, si = sub->n - 1;
    int p_end, s_end, p_type, s_type;

    while (more_perf || more_sub) {
        if (!more_sub) {                                                        /* :60 */
            while (more_perf) {
                p_end = perf->a[pi].end; p_type = perf->a[pi].type;
                if (p_end >= seed_start) {
                    if (p_type != RBO_RANK_N) { iv_push(&cand_type, RBO_RANK_P); iv_push(&cand_idx, (int)pi); }
                    pi -= 1;
                }
                if (pi < 0 || p_end < seed_start) more_perf = 0;
            }
        } else if (!more_perf) {                                                /* :76 */
            while (more_sub) {
                s_end = sub->a[si].end; s_type = sub->a[si].type;
                if (s_end >= seed_start) {
                    if (s_type != RBO_RANK_N) { iv_push(&cand_type, RBO_RANK_S); iv_push(&cand_idx, (int)si); }
                    si -= 1;
                }
                if (si < 0 || s_end < seed_start) more_sub = 0;
            }
        } else {                                                                /* :92 */
            p_end = perf->a[pi].end; p_type = perf->a[pi].type;
            s_end = sub->a[si].end;  s_type = sub->a[si].type;
            if (s_end > p_end) {
                if (s_type != RBO_RANK_N) { iv_push(&cand_type, RBO_RANK_S); iv_push(&cand_idx, (int)si); }
                si -= 1;
            } else {
                if (p_type != RBO_RANK_N) { iv_push(&cand_type, RBO_RANK_P); iv_push(&cand_idx, (int)pi); }
                pi -= 1;
            }
            if (pi < 0 || p_end < seed_start) more_perf = 0;
            if (si < 0 || s_end < seed_start) more_sub = 0;
        }
    }

    const int seed_rend = seed_end + mlen;                                      /* :118-120 */
    const int seed_len = seed_end - seed_start;
    const int seed_rlen = seed_len + mlen;
    int merge_start = 0, merge_end = 0, overlap = 0, new_type;

#define SUB_RETURN(x) do { int r_ = (x); iv_free(&cand_type); iv_free(&cand_idx); return r_; } while (0)

    for (int64_t q = 0; q < cand_idx.n; q++) {                                  /* :127 */
        const int i = cand_idx.a[q];
        const seedvec *src = (cand_type.a[q] == RBO_RANK_P) ? perf : sub;
        old_start = src->a[i].start; old_mlen = src->a[i].mlen; old_end = src->a[i].end;
        old_rend = old_end + old_mlen; old_type = src->a[i].type;
        old_len = old_end - old_start;
        old_rlen = old_rend - old_start;

        if (old_end < seed_start) break;                                        /* :150 */
        if (old_type == RBO_RANK_N) continue;                                   /* :152 */
        if (seed_end < old_start) continue;                                     /* :155 */

        if (seed_start == old_start && seed_end == old_end) {                   /* :158 identical */
            if (seed_type == RBO_RANK_S && (old_type == RBO_RANK_P || old_type == RBO_RANK_Q)) SUB_RETURN(from_index);
            else if (seed_type == RBO_RANK_Q && old_type == RBO_RANK_P) SUB_RETURN(from_index);
            else if (seed_type == RBO_RANK_Q && old_type == RBO_RANK_S) TAG_N(sub, i);
            else if ((seed_type == RBO_RANK_Q && old_type == RBO_RANK_Q) || (seed_type == RBO_RANK_S && old_type == RBO_RANK_S)) {
                if (mlen % old_mlen == 0) SUB_RETURN(from_index);               /* :173 */
                else if (old_mlen % mlen == 0) {                                /* :176 */
                    TAG_N(sub, i);
                    SUB_RETURN(subst_add(c, seed_start, seed_end, mlen, seedlen_cutoff, from_index, seed_type));
                } else {                                                        /* :185 */
                    if (!retain_identical(c, seed_start, seed_end, mlen, old_mlen)) SUB_RETURN(from_index);
                    TAG_N(sub, i); break;
                }
            }
        }

        else if (old_start <= seed_start && seed_end <= old_end) {              /* :194 nested */
            if (seed_type == RBO_RANK_S && (old_type == RBO_RANK_P || old_type == RBO_RANK_Q)) SUB_RETURN(from_index);
            else if (seed_type == RBO_RANK_Q && old_type == RBO_RANK_P) SUB_RETURN(from_index);
            else if ((seed_type == RBO_RANK_Q && old_type == RBO_RANK_S) || (seed_type == RBO_RANK_Q && old_type == RBO_RANK_Q) ||
                     (seed_type == RBO_RANK_S && old_type == RBO_RANK_S)) {
                new_type = (seed_type == RBO_RANK_S && old_type == RBO_RANK_S) ? RBO_RANK_S : RBO_RANK_Q;   /* :203 */
                if (mlen == old_mlen) {                                         /* :206 */
                    sub->a[i].mlen = mlen; sub->a[i].type = new_type;
                    SUB_RETURN(from_index);
                } else if (mlen % old_mlen == 0) SUB_RETURN(from_index);        /* :213 */
                else if (old_mlen % mlen == 0 || old_mlen < mlen) {             /* :216 */
                    if (seed_rlen >= old_mlen - 1 || seed_rlen >= old_len - 1) {
                        sub->a[i].mlen = mlen; sub->a[i].type = new_type;
                        SUB_RETURN(from_index);
                    }
                } else {                                                        /* :227 */
                    if (!retain_nested(c, seed_start, seed_end, mlen, old_mlen)) SUB_RETURN(from_index);
                }
            }
        }

        else if (seed_start <= old_start && old_end <= seed_end) {              /* :235 parent */
            if ((seed_type == RBO_RANK_S && (old_type == RBO_RANK_P || old_type == RBO_RANK_Q)) ||
                (seed_type == RBO_RANK_Q && old_type == RBO_RANK_P)) {
                if (old_mlen % mlen == 0) {                                     /* :239 */
                    if (old_type == RBO_RANK_P) TAG_N(perf, i); else TAG_N(sub, i);
                    SUB_RETURN(subst_add(c, seed_start, seed_end, mlen, seedlen_cutoff, from_index, RBO_RANK_Q));
                } else if (mlen % old_mlen == 0 || old_mlen < mlen) {           /* :249 */
                    if (seed_len / mlen > 3 && old_rlen >= (3 * mlen) - 1) {
                        if (old_type != RBO_RANK_P) TAG_N(sub, i);
                        SUB_RETURN(subst_add(c, seed_start, seed_end, old_mlen, seedlen_cutoff, from_index, RBO_RANK_Q));
                    } else if ((seed_len / mlen <= 3) && ((old_rlen >= mlen - 1) || (old_rlen >= seed_len - 1))) {
                        if (old_type != RBO_RANK_P) TAG_N(sub, i);
                        SUB_RETURN(subst_add(c, seed_start, seed_end, old_mlen, seedlen_cutoff, from_index, RBO_RANK_Q));
                    }
                }
                /* :269 mlen < old_mlen: keep both */
            } else if (seed_type == RBO_RANK_Q && old_type == RBO_RANK_S) {     /* :275 */
                TAG_N(sub, i); break;
            } else if ((seed_type == RBO_RANK_Q && old_type == RBO_RANK_Q) || (seed_type == RBO_RANK_S && old_type == RBO_RANK_S)) {
                if (old_mlen % mlen == 0) {                                     /* :283 */
                    TAG_N(sub, i);
                } else if ((mlen % old_mlen == 0) || (mlen > old_mlen)) {       /* :288 */
                    if (old_rlen >= mlen - 1 || old_rlen >= seed_len - 1) {
                        TAG_N(sub, i);
                        SUB_RETURN(subst_add(c, seed_start, seed_end, old_mlen, seedlen_cutoff, from_index, seed_type));
                    } else {
                        if (retain_nested(c, old_start, old_end, old_mlen, mlen)) continue;
                        TAG_N(sub, i);
                    }
                } else if (old_mlen > mlen) {                                   /* :303 */
                    if (retain_nested(c, old_start, old_end, old_mlen, mlen)) continue;
                    TAG_N(sub, i);
                    SUB_RETURN(subst_add(c, seed_start, seed_end, mlen, seedlen_cutoff, from_index, seed_type));
                }
            }
        }

        else {                                                                  /* :318 overlap */
            if (old_start < seed_start) {
                if (old_mlen <= mlen) overlap = (seed_end <= old_rend) ? seed_end - seed_start : old_rend - seed_start;
                else                  overlap = (seed_end <= old_end)  ? seed_end - seed_start : old_end - seed_start;
                merge_start = old_start; merge_end = seed_end;
            } else {
                if (mlen <= old_mlen) overlap = (old_end <= seed_rend) ? old_end - old_start : seed_rend - old_start;
                else                  overlap = (old_end <= seed_end)  ? old_end - old_start : seed_end - old_start;
                merge_start = seed_start; merge_end = old_end;
            }

            if ((old_mlen % mlen == 0) || old_mlen > mlen) {                    /* :343 */
                if (old_len / old_mlen > 3 && overlap >= (3 * old_mlen) - 1) {
                    if (old_type == RBO_RANK_P) TAG_N(perf, i); else TAG_N(sub, i);
                    SUB_RETURN(subst_add(c, merge_start, merge_end, mlen, seedlen_cutoff, from_index, RBO_RANK_Q));
                } else if ((old_len / old_mlen <= 3) && ((overlap >= old_mlen - 1) || (overlap >= old_len - 1))) {
                    if (old_type == RBO_RANK_P) TAG_N(perf, i); else TAG_N(sub, i);
                    SUB_RETURN(subst_add(c, merge_start, merge_end, mlen, seedlen_cutoff, from_index, RBO_RANK_Q));
                }
            } else if ((mlen % old_mlen == 0) || mlen > old_mlen) {             /* :362 */
                if (seed_len / mlen > 3 && overlap >= (3 * mlen) - 1) {
                    if (old_type != RBO_RANK_P) TAG_N(sub, i);
                    SUB_RETURN(subst_add(c, merge_start, merge_end, old_mlen, seedlen_cutoff, from_index, RBO_RANK_Q));
                } else if ((seed_len / mlen <= 3) && ((overlap >= mlen - 1) || (overlap >= seed_len - 1))) {
                    if (old_type != RBO_RANK_P) TAG_N(sub, i);
                    SUB_RETURN(subst_add(c, merge_start, merge_end, old_mlen, seedlen_cutoff, from_index, RBO_RANK_Q));
                }
            }
        }
    }

    if (seed_end > bset_size - mlen) seed_end = bset_size - mlen;               /* :382-384 */
    sv_push(sub, seed_start, seed_end, mlen, seed_type);
    SUB_RETURN(from_index);
#undef SUB_RETURN
}

/*
 * The window finite-state machine shared by processShiftXORswithSubstitutions
 * (parse_substitute_shiftxor.cpp:391-577) and processShiftXORsAnchored
 * (parse_anchored_shiftxor.cpp:538-726).  The two reference functions are the same loop apart
 * from the threshold, the cut-off table, the add function and which end-of-sequence calls keep
 * the returned cursor; `anchored` selects between them.
 */
typedef struct { int perfect, subst; } cursor2;
static cursor2 anchored_add(rbo_ctx *c, int seed_start, int seed_end, int mlen, const int *seedlen_cutoffs,
                            cursor2 from, int seed_type);

static void window_scan(rbo_ctx *c, int anchored, int window_length, int threshold) {
    const int L = (int)c->L, nm = c->nmotifs;
    const int min_idx = c->m_lo - c->min_shift;
    int *pend_start = (int *)malloc((size_t)nm * sizeof(int));   /* last_starts */
    int *pend_end = (int *)malloc((size_t)nm * sizeof(int));     /* last_ends */
    int *cur_start = (int *)malloc((size_t)nm * sizeof(int));    /* current_starts */
    int *cutoffs = (int *)malloc((size_t)nm * sizeof(int));      /* seedlen_cutoffs */
    unsigned *window = (unsigned *)calloc((size_t)nm, sizeof(unsigned));
    const unsigned wmask = (window_length >= 32) ? 0xffffffffu : ((1u << window_length) - 1u);
    callvec *log = &c->calls[anchored ? RBO_LIST_ANCHORED : RBO_LIST_SUBST];
    int from_index = 0;
    cursor2 from = {0, 0};
    int valid = 0;

    for (int d = 0; d < nm; d++) {
        const int mlen = d + c->m_lo;
        pend_start[d] = pend_end[d] = cur_start[d] = -1;
        if (!anchored) {
            cutoffs[d] = (mlen > 30) ? mlen / 3 : 10;                     /* subst :423 */
        } else {
            cutoffs[d] = (mlen > 6) ? mlen : 10;                          /* anchored :572 */
            if (mlen >= 10) cutoffs[d] = (int)(0.9 * mlen);               /* anchored :573 */
        }
    }

#define EMIT(p_, m_, s_, e_) do { cv_push(log, (p_), (m_), (s_), (e_)); \
        if (!anchored) from_index = subst_add(c, (s_), (e_), (m_), cutoffs, from_index, RBO_RANK_S); \
        else from = anchored_add(c, (s_), (e_), (m_), cutoffs, from, RBO_RANK_A); } while (0)

    int wpos = -window_length;                                             /* :429 */
    for (int p = 0; p < L; p++) {
        wpos += 1;
        if (c->nmask[p]) {                                                 /* :433 */
            for (int d = 0; d < nm; d++) {
                const int mlen = c->min_shift + min_idx + d;
                if (cur_start[d] != -1) {
                    cur_start[d] = wpos;
                    if (pend_end[d] != -1 && pend_end[d] < cur_start[d]) {
                        EMIT(p, mlen, pend_start[d], pend_end[d]);
                        pend_start[d] = -1; pend_end[d] = -1;
                    }
                }
                window[d] = 0;                                             /* <<= window_length */
                cur_start[d] = -1;
            }
            valid = 0;
        } else {
            valid += 1;
            for (int d = 0; d < nm; d++)                                   /* :463-467 */
                window[d] = ((window[d] << 1) | (unsigned)pl_get(c->plane[min_idx + d], p)) & wmask;

            if (valid >= window_length) {                                  /* :469 */
                for (int d = 0; d < nm; d++) {
                    const int mlen = c->min_shift + min_idx + d;
                    const int bits = __builtin_popcount(window[d]);
                    if (bits >= threshold) {                               /* :474 */
                        if (cur_start[d] == -1) {
                            cur_start[d] = wpos;
                            if (pend_end[d] != -1 && pend_end[d] < cur_start[d]) {
                                EMIT(p, mlen, pend_start[d], pend_end[d]);
                                pend_start[d] = -1; pend_end[d] = -1;
                            }
                        }
                    } else if (cur_start[d] != -1) {                       /* :497 */
                        if (pend_start[d] == -1) pend_start[d] = cur_start[d];
                        pend_end[d] = wpos + window_length - 1;
                        cur_start[d] = -1;
                    } else {                                               /* :515 */
                        if (pend_end[d] != -1 && pend_end[d] < wpos) {
                            EMIT(p, mlen, pend_start[d], pend_end[d]);
                            pend_start[d] = -1; pend_end[d] = -1;
                        }
                    }
                }
            }
        }
    }

    /* end of sequence: subst :534-574, anchored :681-723 (xor_idx == -1, so the flush end is L) */
    for (int d = 0; d < nm; d++) {
        const int mlen = c->min_shift + min_idx + d;
        if (pend_end[d] == -1) {
            if (cur_start[d] != -1) {
                if (!anchored) EMIT(L, mlen, cur_start[d], L);
                else { cv_push(log, L, mlen, cur_start[d], L);                       /* :688 result dropped */
                       (void)anchored_add(c, cur_start[d], L, mlen, cutoffs, from, RBO_RANK_A); }
            }
        } else if (cur_start[d] == -1) {
            if (!anchored) EMIT(L, mlen, pend_start[d], pend_end[d]);
            else { cv_push(log, L, mlen, pend_start[d], pend_end[d]);                /* :697 result dropped */
                   (void)anchored_add(c, pend_start[d], pend_end[d], mlen, cutoffs, from, RBO_RANK_A); }
        } else if (pend_end[d] >= cur_start[d] - mlen) {
            pend_end[d] = L;
            if (!anchored) EMIT(L, mlen, pend_start[d], pend_end[d]);
            else { cv_push(log, L, mlen, pend_start[d], pend_end[d]);                /* :706 result dropped */
                   (void)anchored_add(c, pend_start[d], pend_end[d], mlen, cutoffs, from, RBO_RANK_A); }
        } else {
            EMIT(L, mlen, pend_start[d], pend_end[d]);                               /* :713 result kept */
            if (!anchored) EMIT(L, mlen, cur_start[d], L);
            else { cv_push(log, L, mlen, cur_start[d], L);                           /* :717 result dropped */
                   (void)anchored_add(c, cur_start[d], L, mlen, cutoffs, from, RBO_RANK_A); }
        }
    }
#undef EMIT
    free(pend_start); free(pend_end); free(cur_start); free(cutoffs); free(window);
}

/* processShiftXORswithSubstitutions as called at fasta_utils.cpp:136 (window 8, threshold 7: ribbit.cpp:191) */
int rbo_run_subst(rbo_ctx *c) {
    window_scan(c, 0, 8, 7);
    return (int)c->lists[RBO_LIST_SUBST].n;
}

/* ---------------------------------------------------- anchor planes (a8, a9) */

/*
 * generateAnchoredShiftXORs, parse_anchored_shiftxor.cpp:20-56, then the in-place composition of
 * fasta_utils.cpp:143-161.  The reference walks bit indices L-1 down to shift, i.e. p = 0..L-1-shift.
 */
int rbo_run_anchor_planes(rbo_ctx *c) {
    const int L = (int)c->L;
    const int anchor_size = 3;                                      /* ribbit.cpp:191 */
    drop_views(c);                                                  /* plane m becomes XA_m below */
    c->anchor = (uint64_t **)calloc((size_t)c->nshifts, sizeof(uint64_t *));
    for (int i = 0; i < c->nshifts; i++) {
        const int shift = c->min_shift + i;
        uint64_t *a = (uint64_t *)calloc(pl_words(L), sizeof(uint64_t));
        int run_start = -1;                                         /* anchor_start, in p space */
        for (int p = 0; p <= L - 1 - shift; p++) {                  /* :37 */
            if (pl_get(c->plane[i], p) == 1) {
                if (run_start == -1) run_start = p;
            } else {
                /* :44 `anchor_start - xor_idx` is the run length; with anchor_start == -1 it is negative */
                const int run_len = (run_start == -1) ? -1 : p - run_start;
                if (run_len >= anchor_size && run_len < 2 * shift)
                    for (int q = run_start; q < p; q++) pl_set(a, q);      /* set(pos, len, true), :46 */
                run_start = -1;
            }
        }
        c->anchor[i] = a;
    }

    /* fasta_utils.cpp:146-160: XA_m = X_m | anchor_i for i in [max(1,m-2) .. m+2], i != m */
    const size_t nw = pl_words(L);
    uint64_t *acc = (uint64_t *)malloc(nw * sizeof(uint64_t));
    for (int mlen = c->m_lo; mlen <= c->m_hi; mlen++) {
        memset(acc, 0, nw * sizeof(uint64_t));
        for (int i = (mlen > 2) ? mlen - 2 : 1; i <= mlen + 2; i++) {
            const uint64_t *src = (i == mlen) ? c->plane[i - c->min_shift] : c->anchor[i - c->min_shift];
            for (size_t w = 0; w < nw; w++) acc[w] |= src[w];          /* the bitsets' |=, :152-158 */
        }
        memcpy(c->plane[mlen - c->min_shift], acc, nw * sizeof(uint64_t));
    }
    free(acc);
    return 0;
}

/* ------------------------------------------------------- anchored stage (a10, a11) */

/* mergeAllLists, merge_types.cpp:11-189 (guard D1 for empty lists) */
static void merge_all_lists(rbo_ctx *c, int from_perfect, int from_subst, intvec *out_types, intvec *out_idx,
                            int seed_start) {
    seedvec *perf = &c->lists[RBO_LIST_PERFECT], *sub = &c->lists[RBO_LIST_SUBST], *anc = &c->lists[RBO_LIST_ANCHORED];
    intvec sp_types = {0}, sp_idx = {0};
    int perf_done = 0, sub_done = 0;
    int64_t pi = from_perfect, si = from_subst;
    int p_end = 0, s_end = 0, p_type, s_type;

    if (perf->n == 0) perf_done = 1;                                            /* :24 */
    if (sub->n == 0) { sub_done = 1; c->guard_hits++; }                         /* D1 */

    while (!(perf_done && sub_done)) {                                          /* :28 */
        if (sub_done) {
            while (pi >= 0 || !perf_done) {
                p_end = perf->a[pi].end; p_type = perf->a[pi].type;
                if (p_end >= seed_start) {
                    if (p_type != RBO_RANK_N) { iv_push(&sp_types, RBO_RANK_P); iv_push(&sp_idx, (int)pi); }
                    pi -= 1;
                }
                if (pi < 0 || p_end < seed_start) { perf_done = 1; break; }
            }
        } else if (perf_done) {                                                 /* :47 */
            for (;;) {                                      /* `while (substut_end >= 0 || !substut_start_bool)`: */
                s_end = sub->a[si].end; s_type = sub->a[si].type;   /* !substut_start_bool holds until the break */
                if (s_end >= seed_start) {
                    if (s_type != RBO_RANK_N) { iv_push(&sp_types, RBO_RANK_S); iv_push(&sp_idx, (int)si); }
                    si -= 1;
                }
                if (si < 0 || s_end < seed_start) { sub_done = 1; break; }
            }
        } else {                                                                /* :64 */
            p_end = perf->a[pi].end; s_end = sub->a[si].end;
            p_type = perf->a[pi].type; s_type = sub->a[si].type;
            if (s_end > p_end) {
                if (s_type != RBO_RANK_N) { iv_push(&sp_types, RBO_RANK_S); iv_push(&sp_idx, (int)si); }
                si -= 1;
            } else {
                if (p_type != RBO_RANK_N) { iv_push(&sp_types, RBO_RANK_P); iv_push(&sp_idx, (int)pi); }
                pi -= 1;
            }
            if (pi < 0 || p_end < seed_start) perf_done = 1;
            if (si < 0 || s_end < seed_start) sub_done = 1;
        }
    }

    /* :98-188 second merge: the sub+perfect candidates (walked from their LAST entry) against the anchored list */
    int sp_done = 0, anc_done = 0;
    int64_t spi = sp_idx.n - 1, ai = anc->n - 1;
    int sp_end = 0, a_end = 0, a_type, sp_type, idx;

    if (anc->n == 0) {                                                          /* :103 */
        for (int64_t k = 0; k < sp_idx.n; k++) iv_push(out_idx, sp_idx.a[k]);
        for (int64_t k = 0; k < sp_types.n; k++) iv_push(out_types, sp_types.a[k]);
    } else if (sp_idx.n == 0) {                                                 /* :107 */
        for (;;) {
            a_end = anc->a[ai].end; a_type = anc->a[ai].type;
            if (a_end >= seed_start) {
                if (a_type != RBO_RANK_N) { iv_push(out_types, RBO_RANK_A); iv_push(out_idx, (int)ai); }
                ai -= 1;
            }
            if (ai < 0 || a_end < seed_start) { anc_done = 1; break; }
        }
    } else {
        while (!(sp_done && anc_done)) {                                        /* :124 */
            if (anc_done) {
                while (spi >= 0 || !sp_done) {
                    sp_type = sp_types.a[spi]; idx = sp_idx.a[spi];
                    if (sp_type == RBO_RANK_P) sp_end = perf->a[idx].end;
                    else if (sp_type == RBO_RANK_S) sp_end = sub->a[idx].end;
                    if (sp_end >= seed_start) {
                        iv_push(out_types, sp_type); iv_push(out_idx, idx);
                        spi -= 1;
                    }
                    if (spi < 0 || sp_end < seed_start) { sp_done = 1; break; }
                }
            } else if (sp_done) {                                               /* :143 */
                for (;;) {
                    a_end = anc->a[ai].end; a_type = anc->a[ai].type;
                    if (a_end >= seed_start) {
                        if (a_type != RBO_RANK_N) { iv_push(out_types, RBO_RANK_A); iv_push(out_idx, (int)ai); }
                        ai -= 1;
                    }
                    if (ai < 0 || a_end < seed_start) { anc_done = 1; break; }
                }
            } else {                                                            /* :160 */
                sp_type = sp_types.a[spi]; idx = sp_idx.a[spi];
                if (sp_type == RBO_RANK_P) sp_end = perf->a[idx].end;
                else if (sp_type == RBO_RANK_S) sp_end = sub->a[idx].end;
                a_end = anc->a[ai].end;
                if (a_end > sp_end) {
                    iv_push(out_types, RBO_RANK_A); iv_push(out_idx, (int)ai);
                    ai -= 1;
                } else {
                    iv_push(out_types, sp_type); iv_push(out_idx, idx);
                    spi -= 1;
                }
                if (spi < 0 || sp_end < seed_start) sp_done = 1;
                if (ai < 0 || a_end < seed_start) anc_done = 1;
            }
        }
    }
    iv_free(&sp_types); iv_free(&sp_idx);
}

/* addSeedToSeedPositionsAnchored, parse_anchored_shiftxor.cpp:113-534 */
static cursor2 anchored_add(rbo_ctx *c, int seed_start, int seed_end, int mlen, const int *seedlen_cutoffs,
                            cursor2 from, int seed_type) {
    seedvec *perf = &c->lists[RBO_LIST_PERFECT], *sub = &c->lists[RBO_LIST_SUBST], *anc = &c->lists[RBO_LIST_ANCHORED];
    const int bset_size = (int)c->L;
    int old_start = 0, old_end = 0, old_rend = 0, old_mlen = 0, old_len = 0, old_rlen = 0, old_type = 0;
    int from_perfect = from.perfect, from_subst = from.subst;

    for (int64_t i = from_perfect; i < perf->n; i++) {                          /* :133-141 */
        old_start = perf->a[i].start;
        if (old_start > seed_end) break;
        else if ((int64_t)from_perfect == perf->n - 1) break;
        else from_perfect += 1;
    }
    for (int64_t i = from_subst; i < sub->n; i++) {                             /* :143-151 */
        old_start = sub->a[i].start;
        if (old_start > seed_end) break;
        else if ((int64_t)from_subst == sub->n - 1) break;
        else from_subst += 1;
    }
    const cursor2 advanced = {from_perfect, from_subst};

    if (seed_end - seed_start < seedlen_cutoffs[mlen - c->m_lo]) return advanced;   /* :153 */

    intvec cand_types = {0}, cand_idx = {0};
    merge_all_lists(c, from_perfect, from_subst, &cand_types, &cand_idx, seed_start);   /* :156 */

    const int seed_rend = seed_end + mlen;
    const int seed_len = seed_end - seed_start;
    const int seed_rlen = seed_len + mlen;
    int merge_start = 0, merge_end = 0, overlap = 0;

    /* :168-172; only the vectors that are read later are kept */
    intvec nonfactor = {0}, nonfactor_types = {0};
    intvec factor = {0}, factor_sizes = {0}, factor_types = {0};

#define ANC_FREE() do { iv_free(&cand_types); iv_free(&cand_idx); iv_free(&nonfactor); iv_free(&nonfactor_types); \
                        iv_free(&factor); iv_free(&factor_sizes); iv_free(&factor_types); } while (0)
#define ANC_RETURN(x) do { cursor2 r_ = (x); ANC_FREE(); return r_; } while (0)
#define TAG_BY_TYPE(t_, i_) do { if ((t_) == RBO_RANK_P) TAG_N(perf, (i_)); \
                                 else if ((t_) == RBO_RANK_S || (t_) == RBO_RANK_Q) TAG_N(sub, (i_)); } while (0)

    for (int64_t q = 0; q < cand_idx.n; q++) {                                  /* :175 */
        const int i = cand_idx.a[q];
        const seedvec *src = (cand_types.a[q] == RBO_RANK_P) ? perf : (cand_types.a[q] == RBO_RANK_S) ? sub : anc;
        old_start = src->a[i].start; old_mlen = src->a[i].mlen; old_end = src->a[i].end;
        old_rend = old_end + old_mlen; old_type = src->a[i].type;

        if (old_end < seed_start) break;                                        /* :203 */
        if (old_type == RBO_RANK_N) continue;                                   /* :205 */
        if (seed_end < old_start) continue;                                     /* :208 */

        old_len = old_end - old_start;
        old_rlen = old_rend - old_start;

        if (seed_start == old_start && seed_end == old_end) {                   /* :215 identical */
            if (seed_type == RBO_RANK_A && old_type > RBO_RANK_A) ANC_RETURN(advanced);
            else if (seed_type == RBO_RANK_C && old_type == RBO_RANK_A) TAG_N(anc, i);
            /* else: `identical.push_back(i)` -- never read again */
        }

        else if (old_start <= seed_start && seed_end <= old_end) {              /* :231 nested */
            if (old_type > seed_type) ANC_RETURN(advanced);
            else if (seed_type == RBO_RANK_C && old_type == RBO_RANK_A) { }
            else if ((seed_type == RBO_RANK_A && old_type == RBO_RANK_A) || (seed_type == RBO_RANK_C && old_type == RBO_RANK_C)) {
                if (mlen % old_mlen == 0 && (mlen != 4)) ANC_RETURN(advanced);             /* :241 */
                else if (old_mlen % mlen == 0 && (old_mlen != 4)) {                        /* :246 */
                    if (seed_rlen >= old_mlen - 1 || seed_rlen >= old_len) {
                        TAG_N(anc, i);
                        ANC_RETURN(anchored_add(c, old_start, old_end, mlen, seedlen_cutoffs, from, seed_type));
                    } else continue;
                } else {                                                                    /* :256 */
                    if (!retain_nested(c, seed_start, seed_end, mlen, old_mlen)) ANC_RETURN(advanced);
                    else continue;
                }
            }
        }

        else if (seed_start <= old_start && old_end <= seed_end) {              /* :265 parent */
            if (old_type > seed_type) {
                if (mlen % old_mlen == 0) {                                     /* :268 */
                    if ((old_rlen >= mlen - 2) || (old_rlen >= seed_len - 2)) {
                        TAG_BY_TYPE(old_type, i);
                        ANC_RETURN(anchored_add(c, seed_start, seed_end, old_mlen, seedlen_cutoffs, from, RBO_RANK_C));
                    } else {
                        iv_push(&factor, i); iv_push(&factor_sizes, old_mlen); iv_push(&factor_types, old_type);
                    }
                } else if (old_mlen % mlen == 0) {                              /* :285 */
                    if (old_mlen >= 4 * mlen || old_len >= 4 * mlen) {
                        TAG_BY_TYPE(old_type, i);
                        ANC_RETURN(anchored_add(c, seed_start, seed_end, mlen, seedlen_cutoffs, from, RBO_RANK_C));
                    }
                    /* else parentof_subperf_multiple: never read again */
                } else if (old_mlen > mlen) {                                   /* :301 */
                    if (old_mlen >= 4 * mlen || old_len >= 4 * mlen) {
                        TAG_BY_TYPE(old_type, i);
                        ANC_RETURN(anchored_add(c, seed_start, seed_end, mlen, seedlen_cutoffs, from, RBO_RANK_C));
                    }
                } else {                                                        /* :312 */
                    iv_push(&nonfactor, i); iv_push(&nonfactor_types, old_type);
                }
            } else if (seed_type == RBO_RANK_C && old_type == RBO_RANK_A) {     /* :319 */
                TAG_N(anc, i);
            } else if ((seed_type == RBO_RANK_A && old_type == RBO_RANK_A) || (seed_type == RBO_RANK_C && old_type == RBO_RANK_C)) {
                if (old_mlen == mlen) TAG_N(anc, i);                            /* :324 */
                else {
                    if (!retain_nested(c, old_start, old_end, old_mlen, mlen)) TAG_N(anc, i);
                    else {
                        if (mlen % old_mlen == 0) {                             /* :332 */
                            if ((old_rlen >= mlen - 2) || (old_rlen >= seed_len - 2)) {
                                TAG_N(anc, i);
                                ANC_RETURN(anchored_add(c, seed_start, seed_end, old_mlen, seedlen_cutoffs, from, seed_type));
                            }
                            /* else parentof_anchored_factor: never read again */
                        } else if (old_mlen % mlen == 0) continue;
                        /* else parentof_anchored_nonfactor: never read again */
                    }
                }
            }
        }

        else {                                                                  /* :351 overlap */
            if (old_start < seed_start) {
                if (old_mlen <= mlen) overlap = (seed_end <= old_rend) ? seed_end - seed_start : old_rend - seed_start;
                else                  overlap = (seed_end <= old_end)  ? seed_end - seed_start : old_end - seed_start;
                merge_start = old_start; merge_end = seed_end;
            } else {
                if (mlen <= old_mlen) overlap = (old_end <= seed_rend) ? old_end - old_start : seed_rend - old_start;
                else                  overlap = (old_end <= seed_end)  ? old_end - old_start : seed_end - old_start;
                merge_start = seed_start; merge_end = old_end;
            }

            if (seed_type == RBO_RANK_A && old_type > RBO_RANK_C) {             /* :376 */
                if (mlen == old_mlen) {
                    if (overlap >= 4 * mlen) {
                        TAG_BY_TYPE(old_type, i);
                        ANC_RETURN(anchored_add(c, merge_start, merge_end, mlen, seedlen_cutoffs, from, RBO_RANK_C));
                    }
                }
                if ((mlen % old_mlen == 0) || (old_mlen % mlen == 0)) { }       /* :389 */
                else if ((overlap >= mlen - 1) || (overlap >= seed_len - 1)) ANC_RETURN(advanced);
            }

            else if ((seed_type == RBO_RANK_A && old_type == RBO_RANK_A) || (seed_type == RBO_RANK_C && old_type == RBO_RANK_C) ||
                     (seed_type == RBO_RANK_A && old_type == RBO_RANK_C) || (seed_type == RBO_RANK_C && old_type == RBO_RANK_A)) {
                if (mlen == old_mlen) {                                         /* :399 */
                    /* the four `seed_type == ... ? RANK_C : RANK_A;` statements (:402,410,420,428) are no-ops (Q8) */
                    if (old_len >= seed_len) {
                        if ((seed_len >= 3 * mlen) && ((overlap >= 3 * mlen - 1) || (overlap >= seed_len - 1))) {
                            TAG_N(anc, i);
                            ANC_RETURN(anchored_add(c, merge_start, merge_end, old_mlen, seedlen_cutoffs, from, seed_type));
                        } else if ((seed_len < 3 * mlen) && ((overlap >= mlen - 1) || (overlap >= seed_len - 1))) {
                            TAG_N(anc, i);
                            ANC_RETURN(anchored_add(c, merge_start, merge_end, old_mlen, seedlen_cutoffs, from, seed_type));
                        }
                    } else {
                        if ((old_len >= 3 * old_mlen) && ((overlap >= 3 * old_mlen - 1) || (overlap >= old_len - 1))) {
                            TAG_N(anc, i);
                            ANC_RETURN(anchored_add(c, merge_start, merge_end, old_mlen, seedlen_cutoffs, from, seed_type));
                        } else if ((seed_len < 3 * old_mlen) && ((overlap >= old_mlen - 1) || (overlap >= old_len - 1))) {
                            TAG_N(anc, i);
                            ANC_RETURN(anchored_add(c, merge_start, merge_end, old_mlen, seedlen_cutoffs, from, seed_type));
                        }
                    }
                }
            }
        }
    }

    /* :441-468 coverage by non-factor perfect/substitution children.  Quirk Q8: the lists are indexed
     * with the loop counter j, not the stored index; types other than P/S leave stale values. */
    int nonfactor_cov = 0;
    uint32_t prev_start = (uint32_t)-1;
    if (nonfactor.n > 0) {
        for (int64_t j = 0; j < nonfactor.n; j++) {
            const int ktype = nonfactor_types.a[j];
            const seedvec *src = (ktype == RBO_RANK_P) ? perf : (ktype == RBO_RANK_S) ? sub : NULL;
            if (src) {
                if (j < src->n) {
                    old_start = src->a[j].start; old_mlen = src->a[j].mlen; old_end = src->a[j].end;
                    old_rend = old_end + old_mlen;
                } else c->guard_hits++;                                         /* D2 */
            }
            if ((uint32_t)old_rend >= prev_start) nonfactor_cov = (int)((uint32_t)nonfactor_cov + (prev_start - (uint32_t)old_start));
            else if (old_rend < seed_end) nonfactor_cov += old_rend - old_start;
            else nonfactor_cov += seed_end - old_start;
            prev_start = (uint32_t)old_start;
        }
        if (nonfactor_cov > 0.5 * seed_len) ANC_RETURN(advanced);               /* :467 */
    }

    /* :471-526 coverage by factor children, per motif size; maps restated as dense tables */
    if (factor.n > 0) {
        const int tbl = c->max_shift + 8;
        int *prev_starts = (int *)calloc((size_t)tbl, sizeof(int));
        int *coverage = (int *)calloc((size_t)tbl, sizeof(int));
        uint8_t *has_cov = (uint8_t *)calloc((size_t)tbl, 1);
        for (int64_t k = 0; k < factor_sizes.n; k++) {
            prev_starts[factor_sizes.a[k]] = -1;
            coverage[factor_sizes.a[k]] = 0; has_cov[factor_sizes.a[k]] = 1;
        }
        for (int64_t j = 0; j < factor.n; j++) {                                /* :480 */
            const int ktype = factor_types.a[j];
            const seedvec *src = (ktype == RBO_RANK_P) ? perf : (ktype == RBO_RANK_S) ? sub : NULL;
            if (src) {
                if (j < src->n) {
                    old_start = src->a[j].start; old_mlen = src->a[j].mlen; old_end = src->a[j].end;
                    old_rend = old_end + old_mlen;
                } else c->guard_hits++;                                         /* D2 */
            }
            prev_start = (uint32_t)prev_starts[old_mlen];       /* operator[] default-inserts 0 */
            has_cov[old_mlen] = 1;
            if ((uint32_t)old_rend >= prev_start) coverage[old_mlen] = (int)((uint32_t)coverage[old_mlen] + (prev_start - (uint32_t)old_start));
            else if (old_rend < seed_end) coverage[old_mlen] += old_rend - old_start;
            else coverage[old_mlen] += seed_end - old_start;
            prev_starts[old_mlen] = old_start;
        }
        for (int f = 0; f < tbl; f++) {                                         /* :504-507 ascending factors */
            if (!has_cov[f]) continue;
            if (coverage[f] >= 0.8 * seed_len) {                                /* :508 */
                mlen = f; seed_type = RBO_RANK_C;
                for (int64_t j = 0; j < factor.n; j++) {                        /* :511-522, stale start/end */
                    const int ktype = factor_types.a[j];
                    seedvec *dst = (ktype == RBO_RANK_P) ? perf : (ktype == RBO_RANK_S) ? sub : NULL;
                    if (!dst) continue;
                    if (j >= dst->n) { c->guard_hits++; continue; }             /* D2 */
                    old_mlen = dst->a[j].mlen;
                    if (old_mlen == f) {
                        dst->a[j].start = old_start; dst->a[j].end = old_end;
                        dst->a[j].mlen = old_mlen; dst->a[j].type = RBO_RANK_N;
                    }
                }
                break;
            }
        }
        free(prev_starts); free(coverage); free(has_cov);
    }

    if (seed_end > bset_size - mlen) seed_end = bset_size - mlen;               /* :529-531 */
    sv_push(anc, seed_start, seed_end, mlen, seed_type);
    ANC_RETURN(advanced);
#undef ANC_RETURN
#undef ANC_FREE
#undef TAG_BY_TYPE
}

/* processShiftXORsAnchored as called at fasta_utils.cpp:165-167 (window 8, threshold 6) */
int rbo_run_anchored(rbo_ctx *c) {
    window_scan(c, 1, 8, 6);
    return (int)c->lists[RBO_LIST_ANCHORED].n;
}

/* ------------------------------------------------------------- dispatch (a12) */

/* fasta_utils.cpp:187-224: 3-way merge by start, RANK_N skipped, length >= 0.9*m required */
int rbo_run_dispatch(rbo_ctx *c) {
    seedvec *perf = &c->lists[RBO_LIST_PERFECT], *sub = &c->lists[RBO_LIST_SUBST], *anc = &c->lists[RBO_LIST_ANCHORED];
    int64_t ip = 0, is = 0, ia = 0;
    int pick = -1;                                                              /* smallest_type: persists (:183) */
    c->dispatch.n = 0;
    while (ip < perf->n || is < sub->n || ia < anc->n) {
        uint64_t smallest = (uint64_t)-1;
        if (ip < perf->n && smallest > (uint64_t)(int64_t)perf->a[ip].start) { smallest = (uint64_t)(int64_t)perf->a[ip].start; pick = RBO_RANK_P; }
        if (is < sub->n  && smallest > (uint64_t)(int64_t)sub->a[is].start)  { smallest = (uint64_t)(int64_t)sub->a[is].start;  pick = RBO_RANK_S; }
        if (ia < anc->n  && smallest > (uint64_t)(int64_t)anc->a[ia].start)  { smallest = (uint64_t)(int64_t)anc->a[ia].start;  pick = RBO_RANK_A; }
        rbo_seed_t seed;
        if (pick == RBO_RANK_P) seed = perf->a[ip++];
        else if (pick == RBO_RANK_S) seed = sub->a[is++];
        else seed = anc->a[ia++];
        if (seed.type == -1) continue;                                          /* :213 */
        if (seed.end - seed.start >= 0.9 * seed.mlen)                           /* :224 */
            sv_push(&c->dispatch, seed.start, seed.end, seed.mlen, seed.type);
    }
    return (int)c->dispatch.n;
}
