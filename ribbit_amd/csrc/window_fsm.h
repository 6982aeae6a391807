// window_fsm.h -- host replay of the per-motif window state machine of
// processShiftXORswithSubstitutions (parse_substitute_shiftxor.cpp:430-574) and
// processShiftXORsAnchored (parse_anchored_shiftxor.cpp:580-723) from the pass-streak events the
// window-scan kernel produces.  Output: the addSeed calls that machine would make for ONE motif,
// each stamped with the scan position at which the reference makes it (the caller merges the
// motifs by (position, motif) to recover the reference's global call order).
#pragma once
#include <stdint.h>

#include <vector>

#include "device_planes.h"
#include "host_planes.h"
#include "ribbit_hip.h"

namespace rb {

class WindowFsm {
  public:
    WindowFsm(const HostPlanes &planes, int32_t mlen, std::vector<RibbitCall> &out)
        : hp_(planes), mlen_(mlen), out_(out) {}

    // feed one kernel event (window-start position, EV_* kind) in position order; false = malformed
    bool event(int64_t q, uint32_t kind);
    // end of the motif's event stream: in-loop leftovers + the end-of-sequence flush
    bool finish();

  private:
    void emit(int64_t pos, int64_t start, int64_t end) {
        out_.push_back(RibbitCall{(int32_t)pos, mlen_, (int32_t)start, (int32_t)end});
    }
    // a pending group that no later streak merged with is reported at the first evaluated window
    // strictly beyond its end (the `else` branch at parse_substitute_shiftxor.cpp:515-527, or the
    // streak-start branch :477-491 when that window passes)
    void settle_pending_before(int64_t limit_q);

    const HostPlanes &hp_;
    int32_t mlen_;
    std::vector<RibbitCall> &out_;
    int64_t pend_start_ = -1, pend_end_ = -1;   // last_starts / last_ends
    int64_t cur_ = -1;                          // current_starts
    bool open_streak_ = false;                  // a START without its END yet
};

}  // namespace rb
