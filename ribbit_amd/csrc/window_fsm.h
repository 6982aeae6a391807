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
#include "call_vec.h"
#include "ribbit_hip.h"

namespace rb {

class WindowFsm {
  public:
    WindowFsm(const HostPlanes &planes, int32_t mlen) : hp_(&planes), mlen_(mlen) {}

    // calls are appended to *out (may be re-pointed between calls: the caller batches per tile)
    void set_output(CallVec *out) { out_ = out; }

    // feed one kernel event (window-start position, EV_* kind) in position order; false = malformed
    bool event(int64_t q, uint32_t kind);
    // all events with window start < limit_q have been fed: report a pending group now if the
    // window at which the reference reports it lies before limit_q (keeps call generation in
    // position order tile by tile instead of deferring to the motif's next streak)
    void settle_up_to(int64_t limit_q) {
        if (!open_streak_ && cur_ == -1 && pend_end_ != -1) settle_pending_before(limit_q - 1);
    }
    // end of the motif's event stream: in-loop leftovers + the end-of-sequence flush
    bool finish();

  private:
    void emit(int64_t pos, int64_t start, int64_t end) {
        out_->push_back(RibbitCall{(int32_t)pos, mlen_, (int32_t)start, (int32_t)end});
    }
    // a pending group that no later streak merged with is reported at the first evaluated window
    // strictly beyond its end (the `else` branch at parse_substitute_shiftxor.cpp:515-527, or the
    // streak-start branch :477-491 when that window passes)
    void settle_pending_before(int64_t limit_q);

    const HostPlanes *hp_;
    int32_t mlen_;
    CallVec *out_ = nullptr;
    int64_t pend_start_ = -1, pend_end_ = -1;   // last_starts / last_ends
    int64_t cur_ = -1;                          // current_starts
    bool open_streak_ = false;                  // a START without its END yet
};

inline void WindowFsm::settle_pending_before(int64_t limit_q) {
    // pending group (pend_start_, pend_end_) with no current streak: the reference emits it at the
    // first evaluated window start q with pend_end_ < q.  Only called when that q is <= limit_q.
    if (pend_end_ == -1) return;
    const int64_t q = hp_->first_evaluated(pend_end_ + 1);
    if (q == -1 || q > limit_q) return;
    emit(q + 7, pend_start_, pend_end_);
    pend_start_ = pend_end_ = -1;
}

inline bool WindowFsm::event(int64_t q, uint32_t kind) {
    if (kind == EV_START) {
        if (open_streak_) return false;
        open_streak_ = true;
        // windows between the previous END and this START were evaluated-and-failed or skipped
        // (N); a pending group that this streak does not merge with is reported at the first
        // evaluated one past its end -- at the latest q itself (:477-491)
        if (pend_end_ != -1 && pend_end_ < q) settle_pending_before(q);
        cur_ = q;
        return true;
    }
    if (!open_streak_) return false;
    open_streak_ = false;
    if (kind == EV_END_ZERO) {
        // first failing window after the streak (:497-513): the streak joins / becomes the pending group
        if (pend_start_ == -1) pend_start_ = cur_;
        pend_end_ = q + 7;
        cur_ = -1;
    } else if (kind == EV_END_N) {
        // N at scan position q+7 (:433-458): a pending group that ends before the window is reported
        // there; the current streak is dropped without being recorded
        if (pend_end_ != -1 && pend_end_ < q) {
            emit(q + 7, pend_start_, pend_end_);
            pend_start_ = pend_end_ = -1;
        }
        cur_ = -1;
    } else if (kind == EV_END_EOS) {
        // streak still open when the sequence ends: cur_ stays set for the flush
    } else {
        return false;
    }
    return true;
}

inline bool WindowFsm::finish() {
    if (open_streak_) return false;
    const int64_t L = hp_->length;
    if (cur_ == -1) settle_pending_before(L);   // leftover pending group, if any window is still evaluated
    // end-of-sequence flush (:534-574; anchored :681-723)
    if (pend_end_ == -1) {
        if (cur_ != -1) emit(L, cur_, L);
    } else if (cur_ == -1) {
        emit(L, pend_start_, pend_end_);
    } else if (pend_end_ >= cur_ - mlen_) {
        emit(L, pend_start_, L);
    } else {
        emit(L, pend_start_, pend_end_);
        emit(L, cur_, L);
    }
    return true;
}

}  // namespace rb
