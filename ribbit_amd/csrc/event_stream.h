// event_stream.h -- position-ordered, per-motif view of scan-kernel events, and the host passes that
// consume it.  An event source is a buffer of packed events (device_planes.h: ev_pack) plus, per
// motif, a list of segments {offset, count}; inside a segment events are in increasing position, and
// a motif's segments follow each other in increasing position.  Two producers share this shape:
//   * one GPU: the (motif, tile) chunks the scan kernel wrote (segments per motif = tiles);
//   * several ranks scanning chunks of one record: every rank's own-range events, motif-major
//     (segments per motif = ranks), gathered over RCCL.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "device_planes.h"
#include "host_planes.h"
#include "call_vec.h"
#include "ribbit_hip.h"

namespace rb {

struct Seg { uint32_t off, n; };
static_assert(sizeof(Seg) == sizeof(uint64_t), "segment table entry is one 64-bit word");

struct EventSource {
    const uint64_t *ev = nullptr;
    const Seg *segs = nullptr;        // [nm][segs_per_motif]
    size_t segs_per_motif = 0;
    size_t nm = 0;
    int32_t m_lo = 0;
};

class MotifCursor {
  public:
    MotifCursor(const EventSource &src, size_t mi) : ev_(src.ev), seg_(src.segs + mi * src.segs_per_motif), end_(seg_ + src.segs_per_motif) { skip(); }
    bool done() const { return seg_ == end_; }
    uint64_t peek() const { return ev_[seg_->off + i_]; }
    void next() { if (++i_ >= seg_->n) { ++seg_; i_ = 0; skip(); } }

  private:
    void skip() { while (seg_ != end_ && seg_->n == 0) ++seg_; }
    const uint64_t *ev_;
    const Seg *seg_, *end_;
    uint32_t i_ = 0;
};

// Pair the START / END events of the perfect run scan into runs, motif by motif (sorted by motif, start).
// Returns false and sets *why on a malformed stream.
bool pair_perfect_runs(const EventSource &src, std::vector<RibbitRun> &runs, std::string *why);

// Same for ONE rank's own events of a chunk-sharded record (positions in [own_lo, own_hi), shifted by
// pos_offset): a motif's first event may be an END and its last a START -- the run continues in a
// neighbouring chunk.  Those unmatched events are returned in `halves` (packed events) for the cross-rank
// pairing; everything else becomes runs locally, so the pairing work scales with the chunk, not the record.
bool pair_perfect_runs_partial(const EventSource &src, int64_t own_lo, int64_t own_hi, int64_t pos_offset,
                               std::vector<RibbitRun> &runs, std::vector<uint64_t> &halves, std::string *why);

// parse_perfect_shiftxor.cpp:175-223: runs -> the addSeed calls the perfect scanner makes, in its order
void perfect_calls_from_runs(const RibbitRun *runs, size_t n_runs, int64_t length, int min_shift, CallVec &calls);

}  // namespace rb
