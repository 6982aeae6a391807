// ssw_path.hip -- the banded path search of every first-level alignment of a record, batched on the GPU (gfx950).
// Reference: banded_sw of the vendored SSW library (ssw.c:590-775) as ribbit reaches it through Aligner::Align; host
// twin: banded_path in ssw_exact.cpp, which is pinned to the library in tests/test_ssw.py and spells out which parts
// of the arithmetic are the library's by necessity (recurrences, tie-breaks, zero-valued out-of-band neighbours and
// the cleared `edge` slot, the walk's tail).  This file computes the same cells with the lanes of a wavefront running
// ALONG a row of the band:
//   * E (gap that consumes query) and the diagonal depend on the row above only: independent per cell;
//   * F (gap that consumes reference) runs along the row: f[k+1] = max(h[k] - open, f[k] - extend), and since
//     h[k] = max(g[k], f[k]) with g[k] >= 0 the part of a cell that does not depend on F, and open >= extend,
//     f[k+1] = max(g[k] - open, f[k] - extend): a max-plus prefix scan over the lanes (exact in integers);
//   * rows follow each other (read_len steps); bands wider than the wavefront are walked in chunks of 64 cells with the
//     scan's carry.
// One wavefront per alignment; the cell bytes (same encoding as the host twin) go to global memory, the row state
// (h / e of the row above, h of the row) lives in LDS.  Lane 0 then walks the path back and writes it as run-length
// operations.  The band doubles from |ref_len - read_len| + 1 until the banded score reaches the striped score: every
// round is a launch over the alignments still open (the host sizes the cell arena per round).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "kernels.h"

namespace rb {

namespace {

constexpr int GAP_O = 3, GAP_E = 1;
enum : uint8_t { FROM_DIAG = 0, FROM_E = 1, FROM_F = 2, SRC_MASK = 3, E_OPENS = 4, F_OPENS = 8, NO_CELL = 0xff };

__device__ __forceinline__ int path_code(uint8_t c) {      // kBaseTranslation (ssw_cpp.cpp:12-27)
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

template <int CTRL>
__device__ __forceinline__ int pdpp(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }

// exclusive prefix maximum over the 64 lanes (lane 0 receives `identity`).  Inside a row of 16 lanes by DPP row shifts (lanes a
// shift does not reach keep their value), across the four rows through the rows' last lanes -- register operations, where six
// __shfl_up steps were six dependent trips through the LDS crossbar in every row of every alignment.
__device__ __forceinline__ int wave_exclusive_max(int v, int identity, int lane) {
    int x = v;
    x = max(x, pdpp<0x111>(x, x));      // row_shr:1
    x = max(x, pdpp<0x112>(x, x));      // row_shr:2
    x = max(x, pdpp<0x114>(x, x));      // row_shr:4
    x = max(x, pdpp<0x118>(x, x));      // row_shr:8: inclusive maximum inside the row
    const int r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
    const int row = lane >> 4;
    const int before_row = row == 0 ? INT32_MIN : row == 1 ? r0 : row == 2 ? max(r0, r1) : max(max(r0, r1), r2);
    x = max(x, before_row);                                 // inclusive over the wavefront
    const int prev = pdpp<0x138>(x, identity);              // wave_shr:1 (lane 0 keeps `identity`)
    return lane == 0 ? identity : prev;
}
// the value of the lane below (lane 0 receives `first`)
__device__ __forceinline__ int from_lane_below(int v, int first, int lane) {
    const int prev = pdpp<0x138>(v, first);                 // wave_shr:1
    return lane == 0 ? first : prev;
}

// The walk back from the end-point corner (ssw.c:731-775; same walker as the host twin), by ONE lane: the path as run-length
// operations, written backwards into the item's scratch `out` (ref_len + read_len + 2 entries) and appended to path_ops.
__device__ __forceinline__ void walk_back(const uint8_t *cell, int row_cells, int band, int ref_len, int read_len, uint32_t *out,
                                          uint32_t *__restrict__ path_ops, uint32_t path_cap, uint32_t *__restrict__ path_count, int32_t *res) {
    const int cap = ref_len + read_len + 2;
    int at_out = cap;                          // operations are written from the end backwards
    const long n_cells = (long)row_cells * read_len;
    int i = read_len - 1, j = ref_len - 1, count = 0;
    int state = 2;                             // 0 in E, 1 in F, 2 in H
    int op = 0, prev = 0;                      // 0 'M', 1 'I', 2 'D'
    bool ok = true;
    auto push = [&](int o, int n) { if (at_out > 0) out[--at_out] = ((uint32_t)n << 2) | (uint32_t)o; else ok = false; };
    while (i >= 0 && j > 0) {
        const long at = (long)row_cells * i + (j - max(0, i - band));
        if (at < 0 || at >= n_cells) { ok = false; break; }
        const uint8_t code = cell[at];
        if (code == NO_CELL) { ok = false; break; }
        const int via = state != 2 ? state : ((code & SRC_MASK) == FROM_DIAG ? 2 : (code & SRC_MASK) == FROM_E ? 0 : 1);
        if (via == 2) { --i; --j; state = 2; op = 0; }
        else if (via == 0) { --i; state = (code & E_OPENS) ? 2 : 0; op = 1; }
        else { --j; state = (code & F_OPENS) ? 2 : 1; op = 2; }
        if (op == prev) ++count;
        else { push(prev, count); prev = op; count = 1; }
    }
    if (ok) {
        // the library's tail: the operation in progress is closed, and a path that ends in a gap gets one more 'M'
        if (op == 0) push(0, count + 1);
        else { push(op, count); push(0, 1); }
    }
    uint32_t base = 0;
    const uint32_t n_ops = ok ? (uint32_t)(cap - at_out) : 0u;
    if (n_ops) {
        base = atomicAdd(path_count, n_ops);
        if ((uint64_t)base + n_ops <= path_cap)
            for (uint32_t k = 0; k < n_ops; ++k) path_ops[base + k] = out[at_out + (int)k];
    }
    res[0] = ok ? 0 : 1;
    res[1] = band;
    res[2] = (int32_t)base;
    res[3] = (int32_t)n_ops;
}

}  // namespace

// items[4*t .. +4) = {job, band, unused, unused}; cell_off[t] (bytes into `cells`), ops_off[t] (entries into the round's
// scratch `ops`, ref_len + read_len + 2 each).  A finished path is appended to `path_ops` (one atomic on path_count).
// result[4*job .. +4) = {state, band, first operation in path_ops, operations}: state 0 path written, 1 walk failed (flag 1
// of the library), 2 the band was too narrow (best < score): run again with 2 * band.
__global__ __launch_bounds__(64) void ssw_path_kernel(const uint8_t *__restrict__ ascii, int64_t length, const uint8_t *__restrict__ motif_pool,
                                                      const int32_t *__restrict__ jobs /* 9 ints each */, const int32_t *__restrict__ ends /* 8 ints each */,
                                                      const int32_t *__restrict__ items, const uint64_t *__restrict__ cell_off,
                                                      const uint64_t *__restrict__ ops_off, int n_items, uint8_t *__restrict__ cells,
                                                      uint32_t *__restrict__ ops, uint32_t *__restrict__ path_ops, uint32_t path_cap,
                                                      uint32_t *__restrict__ path_count, int32_t *__restrict__ result) {
    extern __shared__ int32_t lds[];          // h_above[slots + 1], e_above[slots + 1], h_row[slots + 1]
    const int t = (int)blockIdx.x;
    if (t >= n_items) return;
    const int lane = (int)threadIdx.x;
    const int job = items[4 * t], band = items[4 * t + 1];
    const int32_t *jb = jobs + 9 * (int64_t)job;
    const int32_t *en = ends + 8 * (int64_t)job;
    const int atom = jb[3];
    int qstart = jb[4];
    if (qstart < 0) qstart = 0;               // the host's slice(): a negative start clamps
    const uint8_t *motif = motif_pool + jb[8];
    const int score = en[0], ref_end = en[1], query_end = en[2], ref_begin = en[5], query_begin = en[6];
    const int ref_len = ref_end - ref_begin + 1, read_len = query_end - query_begin + 1;
    const int row_cells = 2 * band + 1, slots = row_cells + 2;
    int32_t *h_above = lds, *e_above = lds + (slots + 1), *h_row = lds + 2 * (slots + 1);
    for (int i = lane; i < 3 * (slots + 1); i += 64) lds[i] = 0;
    __builtin_amdgcn_wave_barrier();
    uint8_t *cell = cells + cell_off[t];
    // the reference base of column j is motif[(ref_begin + j) % atom]: one modulo per lane here instead of one per cell (a lane's
    // column is first + k0 + lane, and first and k0 are uniform
    // ... and the uniform part is carried from row to row: `first` grows by one per row once the row index passes the band)
    const int lane_mod = lane % atom, chunk_mod = 64 % atom;
    int row_mod = ref_begin % atom;            // (ref_begin + first) % atom of the current row
    const uint8_t *q_at = ascii + qstart + query_begin;      // query position i of the rectangle
    // the motif's bases as codes in LDS (a cell's reference base was a byte from global memory and a ten-way switch, per cell and
    // row), the query's sixty-four rows at a time in a register (lane r holds row i0 + r; a row reads its own with v_readlane)
    uint8_t *mcode = reinterpret_cast<uint8_t *>(lds + 3 * (slots + 1));
    const bool table = atom <= SSW_PATH_CODE_TABLE;
    if (table) for (int a = lane; a < atom; a += 64) mcode[a] = (uint8_t)path_code(motif[a]);
    __builtin_amdgcn_wave_barrier();
    int qcodes = 4;
    int best = 0;
    if (row_cells <= 64) {
        // ---- a row fits the wavefront (band <= 31: all but a few alignments of this kernel): the row above lives in REGISTERS.  Cell
        // (i, j) is lane j - first(i); the row above is shifted by s = first(i) - first(i - 1) lanes (1 once the band has left the
        // first column), so "above" is a wave_shl by s and "diagonal" a wave_shr by 1 - s; lanes beyond a row's cells hold 0, which is
        // what the out-of-band neighbours and the cleared edge slot of the LDS form below read as.  No LDS, no barrier in the loop.
        int h_prev = 0, e_prev = 0;
        for (int i = 0; i < read_len; ++i) {
            if ((i & 63) == 0) qcodes = i + lane < read_len ? path_code(q_at[i + lane]) : 4;
            const int qc = __builtin_amdgcn_readlane(qcodes, i & 63);
            const int first = max(0, i - band), last = min(ref_len - 1, i + band);
            const int n_in_row = last - first + 1;
            const bool shifted = i > band;
            if (shifted) { if (++row_mod == atom) row_mod = 0; }
            const bool live = lane < n_in_row;
            const int h_next = __builtin_amdgcn_update_dpp(0, h_prev, 0x130, 0xf, 0xf, false);      // wave_shl:1 (lane 63 receives 0)
            const int e_next = __builtin_amdgcn_update_dpp(0, e_prev, 0x130, 0xf, 0xf, false);
            const int h_below = from_lane_below(h_prev, 0, lane);
            const int up_h = shifted ? h_next : h_prev, up_e = shifted ? e_next : e_prev, dg_h = shifted ? h_prev : h_below;
            int e = 0, g = 0, diag = 0;
            uint8_t code = 0;
            if (live) {
                const int e_open = i == 0 ? -GAP_O : up_h - GAP_O;
                const int e_ext = i == 0 ? -GAP_E : up_e - GAP_E;
                e = max(e_open, e_ext);
                if (e_open > e_ext) code |= E_OPENS;
                int jm = row_mod + lane_mod;
                if (jm >= atom) jm -= atom;
                const int rc = table ? (int)mcode[jm] : path_code(motif[jm]);
                diag = dg_h + ((rc == qc && rc < 4) ? 2 : -2);
                g = max(max(e, 0), diag);
            }
            const int f_first = max(0 - GAP_O, 0 - GAP_E);           // the cell left of the row is out of band (h = f = 0)
            const int inject = live ? g - GAP_O + (lane + 1) * GAP_E : INT32_MIN / 2;
            const int before = wave_exclusive_max(inject, INT32_MIN / 2, lane);
            const int f = max(f_first, before) - lane * GAP_E;
            const int h = max(g, f);
            const int h_left_cell = from_lane_below(h, 0, lane), f_left_cell = from_lane_below(f, 0, lane);
            uint8_t *row = cell + (size_t)row_cells * i;
            if (live) {
                if (h_left_cell - GAP_O > f_left_cell - GAP_E) code |= F_OPENS;
                const int e0 = max(e, 0), f0 = max(f, 0);
                const int gap = max(e0, f0);
                code |= gap <= diag ? FROM_DIAG : (e0 > f0 ? FROM_E : FROM_F);
                row[lane] = code;
                best = max(best, h);
            } else if (lane < row_cells) row[lane] = NO_CELL;         // columns past the reference
            h_prev = live ? h : 0;
            e_prev = live ? e : 0;
        }
    } else
    for (int i = 0; i < read_len; ++i) {
        if ((i & 63) == 0) qcodes = i + lane < read_len ? path_code(q_at[i + lane]) : 4;
        const int first = max(0, i - band), last = min(ref_len - 1, i + band);
        const int first_above = max(0, i - 1 - band);
        const int edge = min(last + 1, slots - 1);
        if (lane == 0) { h_above[0] = 0; e_above[0] = 0; h_above[edge] = 0; e_above[edge] = 0; h_row[0] = 0; }
        __builtin_amdgcn_wave_barrier();
        const int qc = __builtin_amdgcn_readlane(qcodes, i & 63);
        uint8_t *row = cell + (size_t)row_cells * i;
        int h_left = 0, f_left = 0;            // h and f of the cell left of the chunk (out of band: 0, 0)
        const int n_in_row = last - first + 1;
        if (i > band) { if (++row_mod == atom) row_mod = 0; }   // first = i - band is one more than in the row above
        int base_mod = row_mod;                                 // (ref_begin + first + k0) % atom, uniform
        for (int k0 = 0; k0 < n_in_row; k0 += 64, base_mod = base_mod + chunk_mod >= atom ? base_mod + chunk_mod - atom : base_mod + chunk_mod) {
            const int k = k0 + lane;
            const bool live = k < n_in_row;
            const int j = first + k, u = k + 1;
            const int above = j - first_above + 1;
            int e = 0, g = 0, diag = 0;
            uint8_t code = 0;
            if (live) {
                const int e_open = i == 0 ? -GAP_O : h_above[above] - GAP_O;
                const int e_ext = i == 0 ? -GAP_E : e_above[above] - GAP_E;
                e = max(e_open, e_ext);
                if (e_open > e_ext) code |= E_OPENS;
                int jm = base_mod + lane_mod;
                if (jm >= atom) jm -= atom;
                const int rc = table ? (int)mcode[jm] : path_code(motif[jm]);
                diag = h_above[above - 1] + ((rc == qc && rc < 4) ? 2 : -2);
                g = max(max(e, 0), diag);
            }
            // F along the row: f[k] = max(f_first, max_{s<k} (g[s] - open + (s + 1) * extend)) - k * extend within the chunk,
            // f_first = the chunk's first cell's f from the carried (h, f) of the cell to its left
            const int f_first = max(h_left - GAP_O, f_left - GAP_E);
            const int inject = live ? g - GAP_O + (lane + 1) * GAP_E : INT32_MIN / 2;
            const int before = wave_exclusive_max(inject, INT32_MIN / 2, lane);
            const int f = max(f_first, before) - lane * GAP_E;
            const int h = max(g, f);
            // neighbours for the F direction bit
            const int h_prev = from_lane_below(h, h_left, lane), f_prev = from_lane_below(f, f_left, lane);
            if (live) {
                if (h_prev - GAP_O > f_prev - GAP_E) code |= F_OPENS;
                const int e0 = max(e, 0), f0 = max(f, 0);
                const int gap = max(e0, f0);
                code |= gap <= diag ? FROM_DIAG : (e0 > f0 ? FROM_E : FROM_F);
                row[k] = code;
                e_above[u] = e;
                h_row[u] = h;
                best = max(best, h);
            }
            // carry to the next chunk: the last live lane of this one
            const int last_lane = __builtin_amdgcn_readfirstlane(min(63, n_in_row - 1 - k0));
            h_left = __builtin_amdgcn_readlane(h, last_lane);
            f_left = __builtin_amdgcn_readlane(f, last_lane);
        }
        for (int k = n_in_row + lane; k < row_cells; k += 64) row[k] = NO_CELL;      // columns past the reference
        __builtin_amdgcn_wave_barrier();
        for (int k = 1 + lane; k <= n_in_row; k += 64) h_above[k] = h_row[k];
        __builtin_amdgcn_wave_barrier();
    }
    // wave maximum of best
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d));
    const int longest = max(ref_len, read_len);
    int32_t *res = result + 4 * (int64_t)job;
    if (best < score && band * 2 <= longest) {
        if (lane == 0) { res[0] = 2; res[1] = band; res[2] = 0; res[3] = 0; }
        return;
    }
    __threadfence();                           // the cells this wave wrote, visible to its own loads below
    if (lane != 0) return;
    walk_back(cell, row_cells, band, ref_len, read_len, ops + ops_off[t], path_ops, path_cap, path_count, res);
}

// The same search for alignments with a NARROW band (at most SSW_PATH_NARROW_BAND cells either side: a row of at most 15 cells),
// four to a wavefront: an alignment has one DPP row of 16 lanes, so the scan along its row is four row shifts with nothing to
// carry between rows or chunks, and four lanes walk their paths back at once.  Nineteen alignments in twenty are of this kind,
// and with a wavefront each they used fewer than a quarter of its lanes (the kernel above: bound by the throughput of a slice's
// first launch).  Items as above; the four of a wavefront run for as many rows as the longest of them has.
__global__ __launch_bounds__(64) void ssw_path4_kernel(const uint8_t *__restrict__ ascii, int64_t length, const uint8_t *__restrict__ motif_pool,
                                                       const int32_t *__restrict__ jobs /* 9 ints each */, const int32_t *__restrict__ ends /* 8 ints each */,
                                                       const int32_t *__restrict__ items, const uint64_t *__restrict__ cell_off,
                                                       const uint64_t *__restrict__ ops_off, int n_items, uint8_t *__restrict__ cells,
                                                       uint32_t *__restrict__ ops, uint32_t *__restrict__ path_ops, uint32_t path_cap,
                                                       uint32_t *__restrict__ path_count, int32_t *__restrict__ result) {
    const int lane = (int)threadIdx.x, grp = lane >> 4, gl = lane & 15;
    const int t = 4 * (int)blockIdx.x + grp;
    const bool have = t < n_items;
    int job = 0, band = 1, atom = 1, qstart = 0, score = 0, ref_begin = 0, query_begin = 0, ref_len = 0, read_len = 0;
    const uint8_t *motif = motif_pool;
    if (have) {
        job = items[4 * t]; band = items[4 * t + 1];
        const int32_t *jb = jobs + 9 * (int64_t)job;
        const int32_t *en = ends + 8 * (int64_t)job;
        atom = jb[3];
        qstart = jb[4] < 0 ? 0 : jb[4];            // the host's slice(): a negative start clamps
        motif = motif_pool + jb[8];
        score = en[0]; ref_begin = en[5]; query_begin = en[6];
        ref_len = en[1] - en[5] + 1; read_len = en[2] - en[6] + 1;
    }
    const int row_cells = 2 * band + 1;
    uint8_t *cell = have ? cells + cell_off[t] : cells;
    const uint8_t *q_at = ascii + qstart + query_begin;
    const int lane_mod = gl % max(atom, 1);
    int row_mod = ref_begin % max(atom, 1);
    const int my_rows = have ? read_len : 0;
    const int most_rows = max(max(__builtin_amdgcn_readlane(my_rows, 0), __builtin_amdgcn_readlane(my_rows, 16)),
                              max(__builtin_amdgcn_readlane(my_rows, 32), __builtin_amdgcn_readlane(my_rows, 48)));
    int best = 0;
    // the row above in registers, as in the kernel above: an alignment's 16 lanes shift it with row_shl / row_shr (lanes without a
    // source receive 0: bound_ctrl), no LDS and no barrier in the loop
    int h_prev = 0, e_prev = 0;
    for (int i = 0; i < most_rows; ++i) {
        const bool act = i < my_rows;                  // uniform over the 16 lanes of an alignment
        const int first = max(0, i - band), last = min(ref_len - 1, i + band);
        const bool shifted = i > band;
        if (act && shifted) { if (++row_mod == atom) row_mod = 0; }
        const int n_in_row = last - first + 1;         // <= row_cells <= 15
        const bool live = act && gl < n_in_row;
        const int h_next = __builtin_amdgcn_update_dpp(0, h_prev, 0x101, 0xf, 0xf, true);      // row_shl:1
        const int e_next = __builtin_amdgcn_update_dpp(0, e_prev, 0x101, 0xf, 0xf, true);
        const int h_below = __builtin_amdgcn_update_dpp(0, h_prev, 0x111, 0xf, 0xf, true);     // row_shr:1
        const int up_h = shifted ? h_next : h_prev, up_e = shifted ? e_next : e_prev, dg_h = shifted ? h_prev : h_below;
        int e = 0, g = 0, diag = 0;
        uint8_t code = 0;
        if (live) {
            const int qc = path_code(q_at[i]);
            const int e_open = i == 0 ? -GAP_O : up_h - GAP_O;
            const int e_ext = i == 0 ? -GAP_E : up_e - GAP_E;
            e = max(e_open, e_ext);
            if (e_open > e_ext) code |= E_OPENS;
            int jm = row_mod + lane_mod;
            if (jm >= atom) jm -= atom;
            const int rc = path_code(motif[jm]);
            diag = dg_h + ((rc == qc && rc < 4) ? 2 : -2);
            g = max(max(e, 0), diag);
        }
        // F along the row, inside the alignment's 16 lanes: the cell left of the row is out of band (h = f = 0)
        const int f_first = max(0 - GAP_O, 0 - GAP_E);
        int x = live ? g - GAP_O + (gl + 1) * GAP_E : INT32_MIN / 2;
        x = max(x, pdpp<0x111>(x, x));
        x = max(x, pdpp<0x112>(x, x));
        x = max(x, pdpp<0x114>(x, x));
        x = max(x, pdpp<0x118>(x, x));
        const int before = pdpp<0x111>(x, INT32_MIN / 2);          // row_shr:1: the first lane of a row keeps the identity
        const int f = max(f_first, before) - gl * GAP_E;
        const int h = max(g, f);
        const int h_left_cell = pdpp<0x111>(h, 0), f_left_cell = pdpp<0x111>(f, 0);      // left of the first cell: 0, 0
        if (live) {
            if (h_left_cell - GAP_O > f_left_cell - GAP_E) code |= F_OPENS;
            const int e0 = max(e, 0), f0 = max(f, 0);
            const int gap = max(e0, f0);
            code |= gap <= diag ? FROM_DIAG : (e0 > f0 ? FROM_E : FROM_F);
            uint8_t *row = cell + (size_t)row_cells * i;
            row[gl] = code;
            best = max(best, h);
        }
        if (act && gl >= n_in_row && gl < row_cells) cell[(size_t)row_cells * i + gl] = NO_CELL;      // columns past the reference
        if (act) { h_prev = live ? h : 0; e_prev = live ? e : 0; }
    }
    // the alignment's maximum of best: over its 16 lanes
    best = max(best, pdpp<0x128>(best, best));       // row_ror:8
    best = max(best, pdpp<0x124>(best, best));       // row_ror:4
    best = max(best, pdpp<0x122>(best, best));       // row_ror:2
    best = max(best, pdpp<0x121>(best, best));       // row_ror:1
    __threadfence();                                 // the cells this wavefront wrote, visible to its own loads below
    if (!have || gl != 0) return;
    const int longest = max(ref_len, read_len);
    int32_t *res = result + 4 * (int64_t)job;
    if (best < score && band * 2 <= longest) { res[0] = 2; res[1] = band; res[2] = 0; res[3] = 0; return; }
    walk_back(cell, row_cells, band, ref_len, read_len, ops + ops_off[t], path_ops, path_cap, path_count, res);
}

// n_narrow: the first n_narrow items have a band of at most SSW_PATH_NARROW_BAND (four to a wavefront); max_band: of the others
void launch_ssw_paths(const uint8_t *ascii, int64_t length, const uint8_t *motif_pool, const int32_t *jobs, const int32_t *ends,
                      const int32_t *items, const uint64_t *cell_off, const uint64_t *ops_off, int n_items, int max_band,
                      uint8_t *cells, uint32_t *ops, uint32_t *path_ops, uint32_t path_cap, uint32_t *path_count, int32_t *result,
                      hipStream_t stream, int n_narrow) {
    if (n_items <= 0) return;
    n_narrow = max(0, min(n_narrow, n_items));
    if (n_narrow > 0)
        hipLaunchKernelGGL(ssw_path4_kernel, dim3((unsigned)((n_narrow + 3) / 4)), dim3(64), 0, stream, ascii, length, motif_pool, jobs, ends, items, cell_off,
                           ops_off, n_narrow, cells, ops, path_ops, path_cap, path_count, result);
    if (n_items > n_narrow) {
        const size_t lds = 3 * (size_t)(2 * max_band + 1 + 2 + 1) * sizeof(int32_t) + (size_t)SSW_PATH_CODE_TABLE;      // row state + the motif's codes
        hipLaunchKernelGGL(ssw_path_kernel, dim3((unsigned)(n_items - n_narrow)), dim3(64), lds, stream, ascii, length, motif_pool, jobs, ends,
                           items + 4 * (size_t)n_narrow, cell_off + n_narrow, ops_off + n_narrow, n_items - n_narrow, cells, ops, path_ops, path_cap, path_count, result);
    }
}

}  // namespace rb
