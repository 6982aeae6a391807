#include "window_fsm.h"

namespace rb {

void WindowFsm::settle_pending_before(int64_t limit_q) {
    // pending group (pend_start_, pend_end_) with no current streak: the reference emits it at the
    // first evaluated window start q with pend_end_ < q.  Only called when that q is <= limit_q.
    if (pend_end_ == -1) return;
    const int64_t q = hp_.first_evaluated(pend_end_ + 1);
    if (q == -1 || q > limit_q) return;
    emit(q + 7, pend_start_, pend_end_);
    pend_start_ = pend_end_ = -1;
}

bool WindowFsm::event(int64_t q, uint32_t kind) {
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

bool WindowFsm::finish() {
    if (open_streak_) return false;
    const int64_t L = hp_.length;
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
