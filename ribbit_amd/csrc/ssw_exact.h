// ssw_exact.h -- striped Smith-Waterman with the exact results of the library ribbit links
// (see ssw_exact.cpp).  Host code; the query/reference pairs come from refine.cpp's alignment jobs.
#pragma once
#include <stdint.h>

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

// What the two striped passes of an alignment (ssw.c:843-891) determine -- everything except the path.  The GPU
// computes these for all first-level alignments of a record at once (kernels.hip: ssw_passes_kernel).
struct SswEnds {
    int32_t score = 0, ref_end = -1, query_end = 0;       // forward pass
    int32_t score2 = 0, ref_end2 = -1;                    // second best outside the mask window (forward pass)
    int32_t ref_begin = -1, query_begin = -1;             // reverse pass from the end point
    int32_t flag = 0;                                     // 2 when the reverse pass scored less than the forward pass
};
// forward + reverse striped passes on the host (8-bit, 16-bit when it saturates); query_len > 0
void ssw_passes(const char *query, int query_len, const char *ref, int ref_len, int mask_len, SswEnds &ends);
// banded traceback between the end points and CIGAR / mismatch count (ssw.c:893-927, ssw_cpp.cpp:126-207)
void ssw_finish(const char *query, int query_len, const char *ref, int ref_len, const SswEnds &ends, SswResult &out);

// The path between the end points as the GPU found it (ssw_path.hip): run-length operations, length << 2 | {0 'M', 1 'I',
// 2 'D'}, in path order; failed = the library's "traceback error" (flag 1).
struct SswPath {
    const uint32_t *ops = nullptr;
    int32_t n_ops = 0;
    bool failed = false;
};
// ssw_finish with the path already known
void ssw_finish_with_path(const char *query, int query_len, const char *ref, int ref_len, const SswEnds &ends, const SswPath &path, SswResult &out);
// ... for a reference that is `motif` (atom bases) repeated, without spelling it out: same result, reads the aligned windows only
void ssw_finish_with_path_periodic(const char *query, int query_len, const char *motif, int atom, const SswEnds &ends, const SswPath &path, SswResult &out);
// the whole alignment against `motif` repeated, host only: passes, banded path, then the periodic finish (a test twin of the GPU path)
void ssw_align_periodic(const char *query, int query_len, const char *motif, int atom, int ref_len, int mask_len, SswResult &out);

// Aligner().Align(query, ref, ref_len, Filter(), &alignment, mask_len) with the default scores
// (match 2, mismatch 2, gap open 3, gap extend 1; ssw_cpp.cpp:230-242).
void ssw_align(const char *query, int query_len, const char *ref, int ref_len, int mask_len, SswResult &out);

}  // namespace rb
