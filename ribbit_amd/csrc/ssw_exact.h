// ssw_exact.h -- striped Smith-Waterman with the exact results of the library ribbit links
// (see ssw_exact.cpp).  Host code; the query/reference pairs come from refine.cpp's alignment jobs.
#pragma once
#include <string>

namespace rb {

struct SswResult {
    int score = 0, score2 = 0;            // sw_score, sw_score_next_best
    int ref_begin = -1, ref_end = 0;      // 0-based, inclusive
    int query_begin = -1, query_end = 0;
    int ref_end2 = 0;                     // ref_end_next_best
    int mismatches = 0;
    int flag = 0;                         // Aligner::Align's return value: 0 ok, 1 traceback failed, 2 path may miss a part
    bool skipped = false;                 // empty query: the library returns without touching its output
    std::string cigar;                    // "<n>S<n>=<n>X<n>I<n>D...": Alignment::cigar_string
};

// Aligner().Align(query, ref, ref_len, Filter(), &alignment, mask_len) with the default scores
// (match 2, mismatch 2, gap open 3, gap extend 1; ssw_cpp.cpp:230-242).
void ssw_align(const char *query, int query_len, const char *ref, int ref_len, int mask_len, SswResult &out);

}  // namespace rb
