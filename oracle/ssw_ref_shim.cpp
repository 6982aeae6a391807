/*
 * ssw_ref_shim.cpp -- TEST-ONLY.  A C entry point around the reference's own
 * StripedSmithWaterman::Aligner (ssw_cpp.cpp / ssw.c, compiled from /root/reference by oracle/Makefile
 * into oracle/_ref/) so that tests can drive the real thing exactly the way ribbit does
 * (parse_seed.cpp:404, parse_smallmotif_seed.cpp:270): default Aligner (match 2, mismatch 2, gap open 3,
 * gap extend 1), default Filter, Align(query, ref, ref_len, filter, &alignment, 15).
 * This file contains no reference code; it only calls the reference's public API.
 */
#include <cstdint>
#include <cstring>

#include "ssw_cpp.h"

extern "C" {

struct ref_ssw_result {
    int32_t sw_score, sw_score_next_best, ref_begin, ref_end, query_begin, query_end, ref_end_next_best, mismatches;
    int32_t flag;        /* return value of Align */
    int32_t cigar_len;   /* strlen of the cigar string (may exceed cap: then it is truncated) */
};

int ref_ssw_align(const char *query, const char *ref, int ref_len, int mask_len, ref_ssw_result *out, char *cigar, int cap) {
    static StripedSmithWaterman::Aligner aligner;
    StripedSmithWaterman::Filter filter;
    StripedSmithWaterman::Alignment al;
    al.Clear();
    const uint16_t flag = aligner.Align(query, ref, ref_len, filter, &al, mask_len);
    out->sw_score = al.sw_score; out->sw_score_next_best = al.sw_score_next_best;
    out->ref_begin = al.ref_begin; out->ref_end = al.ref_end;
    out->query_begin = al.query_begin; out->query_end = al.query_end;
    out->ref_end_next_best = al.ref_end_next_best; out->mismatches = al.mismatches;
    out->flag = flag;
    out->cigar_len = (int32_t)al.cigar_string.size();
    if (cap > 0) {
        std::strncpy(cigar, al.cigar_string.c_str(), (size_t)cap - 1);
        cigar[cap - 1] = 0;
    }
    return 0;
}

}
