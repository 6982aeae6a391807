// device_planes.h -- HBM layout shared by all ribbit_amd kernels (gfx950 only).
//
// A record of L bases is held as three bit planes of 32-bit words, LSB first: base p is
// bit (p & 31) of word (p >> 5).
//   hi  = left bit of the 2-bit code   (fasta_utils.cpp:94-114: A 00, C 01, G 10, T 11)
//   lo  = right bit
//   brk = "break" mask: 1 for N (any non-ACGT byte), for every position >= L and for the
//         lead padding (positions < 0).  hi/lo are 0 wherever brk is 1, which reproduces both
//         the reference's N-encodes-as-A rule and Boost's zero fill of `bitset << s`.
// Each plane is allocated as LEAD_WORDS + nwords_padded words; kernels receive pointers that
// are already advanced by LEAD_WORDS, so index -1 is valid.  Tiles: one wavefront owns
// 64 lanes x WORDS_PER_LANE consecutive words (lane-contiguous, so cross-word funnel shifts
// stay inside a lane).
#pragma once
#include <stdint.h>

namespace rb {

constexpr int WORDS_PER_LANE = 8;
constexpr int TILE_WORDS = 64 * WORDS_PER_LANE;   // 512 words = 16384 bases per wave tile
constexpr int TILE_BASES = TILE_WORDS * 32;
constexpr int LEAD_WORDS = 72;          // >= 8 halo lanes x WORDS_PER_LANE + 1 (the anchored kernel's left halo at the largest motif)
constexpr int TAIL_SLACK_WORDS = 544;    // >= TILE_WORDS + halo lane + shifted-operand words

struct DevicePlanes {
    const uint32_t *hi;
    const uint32_t *lo;
    const uint32_t *brk;
    int64_t length;       // L
    int64_t ntiles;       // tiles covering words 0 .. L/32 (position L included)
    int64_t tail_words;   // words readable past ntiles*TILE_WORDS
};

// The anchored kernel gives up `hl` lanes at either end of every wave as halo lanes: whether a run of X_s ones is
// an anchor depends on its true length up to 2s, i.e. on up to 2*(max_motif+2) bases beyond a lane.  The last
// word of the last halo lane must still be classified exactly, so the halo has to span 2s + 32 bases:
//   hl = ceil((2*(max_motif+2) + 32) / 256)      1 up to max_motif 110, 5 at 500, 8 at 990
constexpr int ANCHORED_MAX_MOTIF = 990;
__host__ __device__ inline int anchored_halo_lanes(int max_motif) { return (2 * (max_motif + 2) + 32 + 255) / 256; }
__host__ __device__ inline int anchored_tile_words(int hl) { return (64 - 2 * hl) * WORDS_PER_LANE; }

// Event buffer sharding: EV_SHARDS regions, one counter each (own 128-byte line).
constexpr int EV_SHARDS = 64;
constexpr int EV_COUNTER_STRIDE = 32;                          // uint32 words between counters
constexpr int EV_SUMMARY = EV_SHARDS * EV_COUNTER_STRIDE;      // [0] total events, [1] overflow flag
constexpr int EV_COUNTER_WORDS = EV_SUMMARY + EV_COUNTER_STRIDE;

// raw device event: one transition of a per-motif bitmap
//   bits  0..31 position, 32..47 motif length, 48..51 kind
enum : uint32_t { EV_START = 0, EV_END_ZERO = 1, EV_END_N = 2, EV_END_EOS = 3 };

__host__ __device__ inline uint64_t ev_pack(uint32_t pos, uint32_t mlen, uint32_t kind) {
    return (uint64_t)pos | ((uint64_t)mlen << 32) | ((uint64_t)kind << 48);
}
__host__ __device__ inline uint32_t ev_pos(uint64_t e) { return (uint32_t)e; }
__host__ __device__ inline uint32_t ev_mlen(uint64_t e) { return (uint32_t)(e >> 32) & 0xffffu; }
__host__ __device__ inline uint32_t ev_kind(uint64_t e) { return (uint32_t)(e >> 48) & 0xfu; }

// `term` values of a run record beyond RIBBIT_TERM_* (ribbit_hip.h), used when the loaded piece is one chunk of a
// longer record: the run's START or END lies in the chunk's own range but its partner does not.
enum : int32_t { RUN_NOT_OWNED = -1, RUN_HALF_START = 3, RUN_HALF_END = 4 /* + RIBBIT_TERM_* of the END */ };

// Device-side pairing of the perfect scan's events into runs (pair_runs_* kernels): status words
//   [PAIR_FLAGS] error bits (0 = clean), [PAIR_TOTAL] number of run records, [PAIR_HALVES] number of half records
enum : uint32_t { PAIR_FLAGS = 0, PAIR_TOTAL = 1, PAIR_HALVES = 2, PAIR_STATUS_WORDS = 4 };
enum : uint32_t {
    PAIR_BAD_EVENT = 1,        // motif or tile outside the launch
    PAIR_DUP_CHUNK = 2,        // two chunks for one (motif, tile)
    PAIR_NOT_ALTERNATING = 4,  // START/END do not alternate inside a chunk or across tiles
    PAIR_UNTERMINATED = 8,     // START without a later END
    PAIR_NO_ROOM = 16,         // more runs than the output buffer holds
};

}  // namespace rb
