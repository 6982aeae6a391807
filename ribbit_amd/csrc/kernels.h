// kernels.h -- host-callable launchers of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_planes.h"
#include "ribbit_hip.h"

namespace rb {

// fasta_utils.cpp:78-115 -> packed planes.  total_words counts the LEAD padding too; the three
// output pointers are the raw allocations (NOT advanced by LEAD_WORDS).
// zero_words (may be null): n_zero words the kernel also clears (the event counters of the scan that follows).
void launch_pack(const uint8_t *dev_ascii, int64_t length, uint32_t *hi, uint32_t *lo, uint32_t *brk,
                 int64_t total_words, uint32_t *zero_words, int n_zero, hipStream_t stream);

struct PerfectLaunch {
    int m_lo, m_hi;        // motif (== shift) range scanned
    uint32_t ev_cap;       // capacity of the event buffer, in events
};
// parse_perfect_shiftxor.cpp:173-223 hot loop -> run START / END events.
// Events land in EV_SHARDS regions of ev_cap/EV_SHARDS events; counters[] (EV_COUNTER_WORDS words,
// zeroed by the caller) holds one count per region.  A count above the region size = overflow.
void launch_scan_perfect(const DevicePlanes &pl, const PerfectLaunch &pp, uint64_t *events, uint32_t *counters,
                         hipStream_t stream);

// Window scan (parse_substitute_shiftxor.cpp:430-532 with allowed_mismatches = 1, i.e. threshold 7;
// parse_anchored_shiftxor.cpp:580-679 with 2, threshold 6) -> pass-streak START / END events at
// window-start positions.  Same event buffer conventions as launch_scan_perfect.
void launch_scan_window(const DevicePlanes &pl, const PerfectLaunch &pp, int allowed_mismatches, uint64_t *events,
                        uint32_t *counters, hipStream_t stream);

// generateAnchoredShiftXORs (parse_anchored_shiftxor.cpp:20-56) + the composition of fasta_utils.cpp:143-161: xa receives the
// composed planes XA_m, motif-major, xa_stride words per motif.  Tiles are anchored_tile_words(anchored_halo_lanes(pp.m_hi))
// wide.  Requires pp.m_hi <= ANCHORED_MAX_MOTIF.  launch_scan_xa_window scans the planes.
void launch_scan_anchored(const DevicePlanes &pl, const PerfectLaunch &pp, uint32_t *xa, int64_t xa_stride, hipStream_t stream);
// Group filter of the window scan below.  tj_table (may be null: no filter): per motif of the launch, the number of positions a
// group of pass-streaks must span for its call to be able to pass the stage's length filter (<= GROUP_FILTER_MAX; 0: keep every
// group); groups that cannot, and are not kept for another reason (see the kernel), emit no events and leave their end bit in
// dropmap (words 0 .. L/32, zeroed by the caller) instead.
constexpr int GROUP_FILTER_MAX = 16;
// The window scan of processShiftXORsAnchored on composed planes already in HBM (xa, xa_stride words per motif, readable up to
// word ntiles * TILE_WORDS + 2 of every plane): same events and filter as the fused kernel, tiles of TILE_WORDS words.
void launch_scan_xa_window(const DevicePlanes &pl, const PerfectLaunch &pp, const uint32_t *xa, int64_t xa_stride, uint64_t *events,
                           uint32_t *counters, const int32_t *tj_table, uint32_t *dropmap, hipStream_t stream);

// Gathers the used part of every region into `dense` (same capacity) in shard order and writes
// counters[EV_SUMMARY] = total events, counters[EV_SUMMARY+1] = 1 if any region overflowed.
void launch_compact_events(const uint64_t *events, uint32_t ev_cap, uint32_t *counters, uint64_t *dense,
                           hipStream_t stream);

// Device-side pairing of the perfect scan's events (still in their EV_SHARDS regions, counters[] as the scan
// left them) into RibbitRun records ordered by (motif, start): parse_perfect_shiftxor.cpp:173-223's run
// bookkeeping.  table: nm*ntile 8-byte entries, run_base: nm*ntile words, partial: nm*ntile/1024+1 words,
// status: PAIR_STATUS_WORDS words (all device scratch, initialised here).  runs holds run_cap records.
struct PairLaunch {
    uint32_t m_lo, nm;        // first motif, number of motifs
    uint32_t ntile;           // tiles of tile_bases positions covering 0..L
    uint32_t tile_bases;      // positions per scan-kernel tile: TILE_BASES, or the anchored kernel's narrower tile
    uint32_t region_cap;      // events per region
    // chunk of a longer record: only runs whose START lies in [own_lo, own_hi) (positions of the loaded piece) are
    // reported, with pos_offset added; a whole record is own_lo = 0, own_hi = INT64_MAX, pos_offset = 0
    int64_t own_lo, own_hi, pos_offset;
};
// halves (half_cap >= 2*nm records) receives the runs cut by the own range, status[PAIR_HALVES] their number.
hipError_t launch_pair_runs(const uint64_t *events, const uint32_t *counters, const PairLaunch &pl, void *table,
                            uint32_t *run_base, uint32_t *partial, void *runs, uint32_t run_cap, void *halves,
                            uint32_t half_cap, uint32_t *status, hipStream_t stream);

// host_words (page-locked, device-visible): [0, EV_SHARDS) the region counters, then PAIR_STATUS_WORDS status words
void launch_pair_publish(const uint32_t *counters, const uint32_t *status, uint32_t *host_words, hipStream_t stream);

// After launch_compact_events: table[(motif - m_lo) * ntile + tile] = {first event index + 1, index past the last event}
// of that (motif, tile) chunk of `dense` (both 0: no chunk); *status != 0: malformed stream.
hipError_t launch_chunk_table(const uint64_t *dense, const uint32_t *counters, uint32_t m_lo, uint32_t nm, uint32_t ntile,
                              uint32_t tile_bases, void *table, uint32_t *status, hipStream_t stream);

// ---- window stages on the device (window_stage.hip): streaks -> addSeed calls -------------------------------
// counters of launch_window_calls (WS_WORDS words, zeroed by the caller)
enum : uint32_t { WS_N_MAIN = 0, WS_N_EDGE = 1, WS_FLAGS = 2, WS_MAX_END = 3 /* largest end + 1 of an in-loop call */,
                  WS_INEXACT = 4 /* != 0: an owned call's group reaches the piece's artificial left end (chunk mode) */, WS_WORDS = 8 };
enum : uint32_t {
    WS_TWO_FLUSH = 1,    // two end-of-sequence calls for one motif
    WS_BAD_MOTIF = 2,    // streak with a motif outside the launch
    WS_NOT_PROMPT = 4,   // an ordinary call whose end is not its position - 8 (would break the bound argument)
    WS_EDGE_LOST = 8,    // a kept edge call is missing from the main list
};
// E plane (bit q: window [q, q+7] is evaluated) and, reversed, the first word >= w that has an evaluated window;
// n_words = L/32 + 1 words each (brk must be readable one word further); word_tmp: n_words words of scratch
hipError_t launch_eval_planes(const uint32_t *brk, uint32_t n_words, uint32_t *eval, uint32_t *first_rev, uint32_t *word_tmp,
                              void *scratch, size_t scratch_bytes, hipStream_t stream);
// group[i] = (motif index << 32) | (start of the group streak i belongs to) + 1, streaks = the run records the
// pairing kernels made of a window scan's events (motif-major, by start)
hipError_t launch_group_starts(const RibbitRun *runs, uint32_t n, uint32_t m_lo, uint64_t *group, void *scratch,
                               size_t scratch_bytes, hipStream_t stream);
struct WindowCallsLaunch {
    const RibbitRun *runs; uint32_t n_streaks;
    const uint64_t *group;
    const uint32_t *eval, *first_rev, *brk; uint32_t n_words;
    int64_t length;
    uint32_t m_lo, nm;
    const int32_t *min_span;           // device, [nm]
    int full;                          // 1: all calls, unfiltered; 0: compact (filtered + bounds)
    uint64_t *keys, *vals; uint32_t cap;              // main list: key = pos << 10 | motif, val = start << 32 | end
    uint64_t *edge_keys, *edge_vals; uint32_t edge_cap;   // compact mode; val bit 63 = passes the filter
    RibbitCall *flush;                 // [nm], zeroed by the caller: the end-of-sequence call of each motif
    uint32_t *bitmap;                  // [n_words + 1], zeroed by the caller
    uint32_t *counters;                // [WS_WORDS], zeroed by the caller
    // chunk mode (the loaded record is a chunk of a longer record plus halos): only calls made at scan positions
    // own_lo <= pos < own_hi are this chunk's; z_lo: first position whose streak events are exact (0: the piece starts
    // where the record starts); keep_flush: the piece ends where the record ends.  Whole record: 0, 0xffffffff, 0, 1.
    uint32_t own_lo = 0, own_hi = 0xffffffffu, z_lo = 0;
    int keep_flush = 1;
};
void launch_window_calls(const WindowCallsLaunch &w, hipStream_t stream);
// after launch_window_calls (compact mode): folds the calls of the groups the anchored scan's filter dropped (dropmap: bit e =
// such a group ends at e; drop_words words) into bitmap[0 .. n_words] and counters[WS_MAX_END], own range only
void launch_merge_dropmap(const uint32_t *dropmap, uint32_t drop_words, uint32_t n_words, uint32_t own_lo, uint32_t own_hi, uint32_t *bitmap,
                          uint32_t *counters, hipStream_t stream);
hipError_t launch_sort_calls(uint64_t *keys_in, uint64_t *vals_in, uint64_t *keys_out, uint64_t *vals_out, uint32_t n, int key_bits,
                             void *scratch, size_t scratch_bytes, hipStream_t stream);
// bounds of the kept edge calls (edge list sorted by key) -> pend[index in the sorted main list]
hipError_t launch_edge_bounds(const uint64_t *edge_keys, const uint64_t *edge_vals, uint32_t n_edge, uint32_t *edge_tmp, uint32_t *edge_end1,
                              const uint32_t *bitmap, uint32_t *word_tmp, uint32_t *last_word1, uint32_t n_words, const uint64_t *main_keys,
                              uint32_t n_main, int32_t *pend, uint32_t *counters, int32_t pos_offset, void *scratch, size_t scratch_bytes,
                              hipStream_t stream);
// pos_offset: added to every coordinate (a chunk's piece coordinates -> record coordinates)
void launch_assemble_calls(const uint64_t *keys, const uint64_t *vals, uint32_t n, RibbitCall *out, int32_t pos_offset, hipStream_t stream);
// bytes of scratch the rocPRIM scans / sorts above need for these sizes
size_t window_stage_scratch_bytes(size_t n_streaks, size_t n_words, size_t n_calls, size_t n_edge, int key_bits);

// X_shift words [w0, w0+nw) -> out_words (device); if count != nullptr also adds the popcount of
// bits in [p0, p1) to *count.
void launch_plane_words(const DevicePlanes &pl, int shift, int64_t w0, int64_t nw, uint32_t *out_words,
                        int64_t p0, int64_t p1, uint32_t *count, hipStream_t stream);

// longestContinuousMatches (parse_seed.cpp:26-44) of n seeds {start,end,mlen,type} on the composed planes
void launch_seed_longest_runs(const uint32_t *xa, int64_t xa_stride, int m_lo, const void *seeds, int64_t n, int32_t *out,
                              hipStream_t stream);

// one byte per base (0..3 = A C G T, 4 = N) from the resident ASCII record
void launch_sym(const uint8_t *ascii, int64_t length, uint8_t *sym, hipStream_t stream);

// small_motifs.hip: possibleMotifs of every dispatched seed with m <= 10 (one wavefront per seed)
struct SmallMotifLimits {
    int32_t first_window[11];      // smallest j - seed_start with j - seed_start >= 0.9 m - 1 (parse_smallmotif_seed.cpp:96)
    int32_t min_length[11];        // MINIMUM_LENGTH[m]
    int32_t min_units[11];         // PERFECT_UNITS[m]
};
// longest != null: `jobs` is the dispatch list on the device ({start, end, m, type}) and longest[i] its seeds' longest runs; the
// kernel then selects the seeds itself (m <= 10, longest[i] >= longest_threshold) and a seed's index is its place in the list
void launch_small_motifs(const uint8_t *sym, int64_t length, const void *jobs, int64_t njobs, const SmallMotifLimits &lim, void *records,
                         uint32_t record_cap, uint32_t *record_count, void *head, hipStream_t stream, const int32_t *longest = nullptr,
                         int32_t longest_threshold = 0);
// mostFrequentLongerMotif's row scores (parse_seed.cpp:165-243) for njobs seeds {seed_start, seed_sequence_length, m, -}:
// best[job] = (best score << 32) | (0xffffffff - first row with that score), 0 when every row scores 0.
// best[] must be zeroed by the caller.  blocks[nblocks] = {job, first row of a 64-row slice of that seed}: every
// seed is covered by ceil(rows / 64) slices, rows = seed_sequence_length - m + 1 (clipped at the record end).
void launch_long_motif_rows(const uint8_t *sym, int64_t length, const void *jobs, int64_t njobs, const void *blocks,
                            int64_t nblocks, unsigned long long *best, hipStream_t stream);

// The two striped Smith-Waterman passes (ssw.c:843-891) of njobs alignment jobs (RibbitAlignJob records, 9 ints each):
// query = record[query_start, +query_length), reference = the job's motif repeated to ppr_length.  Jobs are launched in
// three size classes (order_small / order_big / order_huge: job indices, each list sorted by size); out[8*job .. +8) = score,
// ref_end, query_end, score2, ref_end2, ref_begin, query_begin, flag -- flag -1: too large for its class, not computed.
constexpr int SSW_SMALL_Q = 128, SSW_SMALL_R = 256, SSW_BIG_Q = 512, SSW_BIG_R = 1024, SSW_HUGE_Q = 2048, SSW_HUGE_R = 4096,
              SSW_GIANT_Q = 4096, SSW_GIANT_R = 8192,      // 61.6 KB of LDS per alignment: the most one workgroup gets by default
              SSW_COLOSSAL_Q = 8192, SSW_COLOSSAL_R = 16384;    // 124 KB: asked for with hipFuncAttributeMaxDynamicSharedMemorySize (a
                                                                // workgroup may have all 160 KB of a CU: tools/probes/lds_probe.hip)
// ssw_wave.hip: the same two passes, one wavefront per alignment (queries of 129..qcap bases, reference up to rcap)
void launch_ssw_passes_wave(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs, const int32_t *order, int n,
                            int mask_len, int qcap, int rcap, int32_t *out, hipStream_t stream);
// ssw_group.hip: the same two passes, one WORKGROUP of `waves` (4, 8 or 16) wavefronts per alignment: a column's stripes dealt to
// the wavefronts, for the long classes.  ssw_group_fits: the class's longest query fits that many wavefronts.
bool ssw_group_fits(int qcap, int waves);
hipError_t launch_ssw_passes_group(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs, const int32_t *order, int n,
                                   int mask_len, int qcap, int rcap, int waves, int32_t *out, hipStream_t stream);
void launch_ssw_passes(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs,
                       const int32_t *order_small, int n_small, const int32_t *order_big, int n_big, const int32_t *order_huge, int n_huge,
                       int mask_len, int32_t *out, hipStream_t stream, int huge_group_waves = 0);
// huge_group_waves: 0 = the huge class on one wavefront per alignment (ssw_wave.hip), 4 / 8 = on a workgroup of that many

// The banded path search (ssw.c:590-775) of n_items alignments whose end points are known, one wavefront each (ssw_path.hip).
// items[4*t] = job index, items[4*t+1] = band of this round; cell_off[t] / ops_off[t]: where item t's cell bytes
// ((2*band+1) * read_len) and scratch operations (ref_len + read_len + 2 entries) go; finished paths are appended to
// path_ops (path_cap entries, *path_count their number so far); result[4*job..] = {state, band, first operation in
// path_ops, operations}: state 0 path found, 1 walk failed, 2 band too narrow (run again with twice the band).
// max_band: the largest band among the items (sizes the LDS).
constexpr int SSW_PATH_MAX_BAND = 2048;
constexpr int SSW_PATH_CODE_TABLE = 1024;      // bytes of LDS for the motif's base codes (a longer motif is read from global memory)
void launch_ssw_paths(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs, const int32_t *ends,
                      const int32_t *items, const uint64_t *cell_off, const uint64_t *ops_off, int n_items, int max_band,
                      uint8_t *cells, uint32_t *ops, uint32_t *path_ops, uint32_t path_cap, uint32_t *path_count, int32_t *result,
                      hipStream_t stream, int n_narrow = 0);
// n_narrow: the first n_narrow items have a band of at most SSW_PATH_NARROW_BAND and run four to a wavefront (ssw_path4_kernel);
// max_band then is the largest band among the OTHERS.
constexpr int SSW_PATH_NARROW_BAND = 7;

// anchored_merge.hip: the anchored stage's seed-list merge (parse_anchored_shiftxor.cpp:113-534, merge_types.cpp:11-189), one lane
// per independent range of the stage's kept calls (parallel_merge.h).  P / S: the perfect and substitution lists on the device (their
// `type` fields change: retirements); *_type0: the types before the stage; first[nr + 1], cut_pos[nr], cur0[2 nr]: the ranges' first
// calls, left cuts and starting cursors; own: n_calls + nr entries, range k's part of the anchored list at first[k] + k (entry 0 of
// every range but the first is the sentinel); range_out[AM_RANGE_OUT_WORDS k ..]: entries in own, status (AM_* bits: the host merges
// the range itself), guard count (2 words), final cursors (2), by-counter head reads (2 x 2), calls taken, the lane's time; log: 4 words per entry
// {kind << 28 | range, list << 31 | index, old type or "was live", 0}; head_log: 8 words per logged list-head write
// {range, list, index, start, end, mlen, type, 0}.
constexpr int AM_PS_CAP = 12, AM_CAND_CAP = 40;      // candidate lists of a call: this much of them in LDS (208 bytes per lane, 13 KB per wavefront) ...
constexpr int AM_CHILD_CAP = 128, AM_COV_CAP = 48, AM_MAX_ROUNDS = 1 << 20;
constexpr int AM_PS_SPILL = 244, AM_CAND_SPILL = 984;      // ... and what a dense locus has beyond, in global memory (256 / 1024 entries in all)
constexpr int AM_SCRATCH_WORDS = 2 * AM_CHILD_CAP + 3 * AM_COV_CAP + AM_PS_SPILL + AM_CAND_SPILL;
constexpr int AM_RANGE_OUT_WORDS = 12;
constexpr int AM_RESIDENT_WAVES = 12 * 256;      // what the chip holds at once: twelve wavefronts per CU (LDS, and three per SIMD by registers)
constexpr uint32_t AM_MAX_PASSES = 12000;        // a range that takes its lane more passes than this (80 ms) is given up: AM_TOO_SLOW
enum : uint32_t { AM_SCRATCH_FULL = 1u, AM_LOG_FULL = 2u, AM_BAD_PLANE = 4u, AM_RUNAWAY = 8u, AM_TOO_SLOW = 16u };
enum : uint32_t { AM_LOG_UNDO = 1u, AM_LOG_READ = 2u };
struct AnchoredMergeArgs {
    RibbitSeed *P, *S;
    uint32_t nP, nS;
    const int32_t *P_type0, *S_type0;
    RibbitCall *calls;                // (their `pos` fields are overwritten with the calls' cursor bounds)
    const int32_t *pend;              // may be null
    const uint32_t *first;
    const int32_t *cut_pos, *cur0;
    uint32_t nr;
    const uint32_t *xa;
    int64_t xa_stride, length;
    int32_t m_lo, m_hi;
    RibbitSeed *own;
    uint32_t *range_out;
    uint32_t *log, *log_count;
    uint32_t log_cap;
    uint32_t *head_log, *head_count;
    uint32_t head_cap;
    uint32_t *scratch;                // AM_SCRATCH_WORDS per LANE of the launch (64 AM_RESIDENT_WAVES at most)
    uint32_t *next_range;             // zero at launch: the lanes take ranges from it, in the order of ...
    const uint32_t *order;            // ... the ranges to merge (n_order of the nr; the others are not touched)
    uint32_t n_order;
    uint32_t *sync;                   // page-locked host memory: [0] entries of `order` left to the lanes (the host threads take from the back), [1] entries the lanes have taken
    uint32_t max_passes;              // a range that needs more passes of its lane's loop than this is left to the host (AM_TOO_SLOW)
};
void launch_anchored_merge(const AnchoredMergeArgs &a, uint32_t n_calls, uint32_t resident_waves /* <= AM_RESIDENT_WAVES */, hipStream_t stream);
void launch_seed_types(const RibbitSeed *seeds, uint32_t n, int32_t *types, hipStream_t stream);      // types[i] = seeds[i].type

// profiling aid: reads nwords dwords of src with one coalesced dword per lane (known byte count)
void launch_calib_stream_read(const uint32_t *src, int64_t nwords, uint32_t *sink, hipStream_t stream);

}  // namespace rb
