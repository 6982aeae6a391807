// kernels.h -- host-callable launchers of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_planes.h"

namespace rb {

// fasta_utils.cpp:78-115 -> packed planes.  total_words counts the LEAD padding too; the three
// output pointers are the raw allocations (NOT advanced by LEAD_WORDS).
void launch_pack(const uint8_t *dev_ascii, int64_t length, uint32_t *hi, uint32_t *lo, uint32_t *brk,
                 int64_t total_words, hipStream_t stream);

struct PerfectLaunch {
    int m_lo, m_hi;        // motif (== shift) range scanned
    uint32_t ev_cap;       // capacity of the event buffer, in events
};
// parse_perfect_shiftxor.cpp:173-223 hot loop -> run START / END events.
// counters[0] receives the number of events produced (may exceed ev_cap: overflow).
void launch_scan_perfect(const DevicePlanes &pl, const PerfectLaunch &pp, uint64_t *events, uint32_t *counters,
                         hipStream_t stream);

// X_shift words [w0, w0+nw) -> out_words (device); if count != nullptr also adds the popcount of
// bits in [p0, p1) to *count.
void launch_plane_words(const DevicePlanes &pl, int shift, int64_t w0, int64_t nw, uint32_t *out_words,
                        int64_t p0, int64_t p1, uint32_t *count, hipStream_t stream);

}  // namespace rb
