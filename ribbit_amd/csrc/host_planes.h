// host_planes.h -- host copy of the packed device planes (3 bits per base).
//
// The sequential seed-list merges (addSeedToSeedPositions*) ask for a handful of plane bits at a
// time -- retainNestedSeed / retainIdenticalSeeds (parse_perfect_shiftxor.cpp:18-43) several
// thousand times per Mbp -- and each answer steers the very next decision, so a GPU round trip per
// query would cost tens of microseconds each.  The planes the GPU packed are therefore copied back
// once per record and those sparse reads are answered here, word-parallel.  Nothing in this file
// scans the sequence per (base, motif): that work exists only as HIP kernels.
#pragma once
#include <stdint.h>

#include <memory>
#include <mutex>
#include <utility>
#include <vector>

namespace rb {

struct HostPlanes {
    int64_t length = 0;
    std::vector<uint32_t> hi, lo, brk;   // word 0 = bases 0..31, padded past L as on the device

    // maximal intervals [first, last] of window starts q that are NOT evaluated by the window scans
    // because an N lies in [q, q+7] (`valid_position < window_length`, parse_substitute_shiftxor.cpp:469)
    std::vector<std::pair<int64_t, int64_t>> blocked;

    // Composed planes XA_m of the anchored stage (fasta_utils.cpp:143-161) for m = xa_m_lo..xa_m_hi:
    // XA_m = X_m | anchor_{m-2} | anchor_{m-1} | anchor_{m+1} | anchor_{m+2}, anchor_s = the runs of X_s ones with
    // 3 <= length < 2s that a zero closes at a position <= L-1-s (parse_anchored_shiftxor.cpp:20-56).  The GPU keeps
    // them in HBM for its own refinement scans; the host merges read a few dozen bits at a time, so the slice a
    // query needs is recomputed here from the packed planes (a run of length < 2s is visible within 2s bases of
    // the slice).  Callers of the host-only entry points may instead supply the planes (xa / xa_view, motif-major,
    // xa_stride words per motif).
    std::vector<uint32_t> xa;
    const uint32_t *xa_view = nullptr;     // when set: the planes live in memory the caller owns
    const uint32_t *xa_words() const { return xa_view ? xa_view : xa.data(); }
    bool xa_stored() const { return xa_stride > 0 && (xa_view != nullptr || !xa.empty()); }
    int64_t xa_stride = 0;
    int xa_m_lo = 0, xa_m_hi = -1;
    // words of XA_mlen covering [start, end): out[0] holds positions (start & ~31) .. +31; bits outside [start, end) are 0
    void xa_slice(int mlen, int start, int end, std::vector<uint32_t> &out) const;
    // adds the bits of anchor_shift over [start, end) to out (same layout)
    void anchor_slice(int shift, int start, int end, std::vector<uint32_t> &out) const;

    void resize(int64_t len, size_t nwords) {
        length = len;
        hi.assign(nwords, 0); lo.assign(nwords, 0); brk.assign(nwords, 0);
        blocked.clear();
        xa.clear(); xa_view = nullptr; xa_stride = 0; xa_m_lo = 0; xa_m_hi = -1;
        sym_cache_.reset();
    }
    // One byte per base (0..3 = A C G T, 4 = N; 16 pad entries = 4 from position L on), decoded from the planes on first use
    // (host threads) and kept until the record changes: the refinement stages read bases by the hundred million, and
    // every one of their calls used to decode the whole record again on one thread (a second per chromosome and call).
    // Thread-safe; the planes must not change while a caller holds the result.
    std::shared_ptr<const std::vector<uint8_t>> symbols(unsigned threads = 0) const;

  private:
    mutable std::shared_ptr<const std::vector<uint8_t>> sym_cache_;
    mutable std::mutex sym_mu_;

  public:
    bool has_xa(int mlen) const { return mlen >= xa_m_lo && mlen <= xa_m_hi; }
    // popcount of XA_mlen over [start, end)
    int range_count_xa(int mlen, int start, int end) const;
    void index_breaks();

    // X_shift word w: bit b = (code[p] == code[p+shift]), p = 32w+b  (fasta_utils.cpp:120-122)
    uint32_t x_word(int shift, int64_t w) const {
        const int64_t q = shift >> 5;
        const unsigned r = (unsigned)shift & 31u;
        const uint64_t h2 = ((uint64_t)hi[w + q + 1] << 32) | hi[w + q];
        const uint64_t l2 = ((uint64_t)lo[w + q + 1] << 32) | lo[w + q];
        return ~((hi[w] ^ (uint32_t)(h2 >> r)) | (lo[w] ^ (uint32_t)(l2 >> r)));
    }
    // popcount of X_shift over [start, end)
    int range_count(int shift, int start, int end) const;

    // smallest window start q >= from that the window scans evaluate (0 <= q <= L-8, no N in
    // [q, q+7]); -1 if there is none
    int64_t first_evaluated(int64_t from) const;
};

}  // namespace rb
