// fasta_stream.cpp -- streaming FASTA reader behind ribbit_fasta_* (include/ribbit_hip.h).
//
// Replaces the reader loop of ribbit.cpp:269-280 (std::getline into a std::string per line, `sequence += line` into one
// pageable std::string per record): the file is read in large blocks and the line bodies are copied ONCE, straight into
// a page-locked buffer that ribbit_hip_load_record_pinned uploads from asynchronously and refinement later reads in
// place.  Buffers are recycled, so a run over thousands of reads page-locks memory a handful of times.
// What a record is follows the reference loop exactly: a line starting with '>' ends the previous record
// IF that record has any bases, and names the next one (text up to the first space); every other line is appended
// without its '\n' (a '\r' stays and later reads as N); the last record is handed out even when it is empty.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ribbit_hip.h"

namespace {

constexpr size_t BLOCK = (size_t)16 << 20;           // file block
constexpr size_t MIN_BUFFER = (size_t)1 << 20;       // smallest record buffer

struct Buffer { char *p = nullptr; size_t cap = 0; };

}  // namespace

struct RibbitFastaReader {
    FILE *f = nullptr;
    bool pinned = false;
    std::vector<char> block;          // raw file bytes
    size_t at = 0, have = 0;          // unread part of block
    bool eof = false, at_line_start = true, done = false;
    std::string next_name;            // name of the record being filled
    std::string out_name;             // name handed out with the last record
    bool in_header = false;           // a '>' line continues into the next block
    std::string header;
    Buffer cur;                       // record being filled
    size_t cur_len = 0;
    std::mutex mu;                    // guards `spare` (release may come from worker threads)
    std::vector<Buffer> spare;
    std::vector<Buffer> lent;
    std::string error;

    int alloc(size_t cap, Buffer &b) {
        b.cap = cap;
        if (pinned) {
            void *p = nullptr;
            if (ribbit_hip_host_alloc(cap, &p) != RIBBIT_OK) { error = "page-locked allocation failed"; return RIBBIT_E_NOMEM; }      // (huge pages + registration from 64 MB on, api_core.cpp)
            b.p = (char *)p;
        } else {
            b.p = (char *)std::malloc(cap);
            if (!b.p) { error = "out of host memory"; return RIBBIT_E_NOMEM; }
        }
        return RIBBIT_OK;
    }
    void dealloc(Buffer &b) {
        if (!b.p) return;
        if (pinned) (void)ribbit_hip_host_free(b.p); else std::free(b.p);
        b = Buffer{};
    }
    // a buffer of at least `cap` bytes: the smallest spare one that fits, else a new one
    int take(size_t cap, Buffer &b) {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = spare.size();
            for (size_t i = 0; i < spare.size(); ++i)
                if (spare[i].cap >= cap && (best == spare.size() || spare[i].cap < spare[best].cap)) best = i;
            if (best != spare.size()) { b = spare[best]; spare.erase(spare.begin() + (long)best); return RIBBIT_OK; }
        }
        return alloc(std::max(cap, MIN_BUFFER), b);
    }
    int append(const char *src, size_t n) {
        if (n == 0) return RIBBIT_OK;
        if (cur_len + n > cur.cap) {
            Buffer bigger;
            const int rc = take(std::max(cur_len + n, cur.cap * 2), bigger);
            if (rc) return rc;
            if (cur_len) std::memcpy(bigger.p, cur.p, cur_len);
            if (cur.p) { std::lock_guard<std::mutex> lk(mu); spare.push_back(cur); }
            cur = bigger;
        }
        std::memcpy(cur.p + cur_len, src, n);
        cur_len += n;
        return RIBBIT_OK;
    }
};

namespace {
thread_local std::string g_fasta_error;
int fail(int code, const std::string &msg) { g_fasta_error = msg; return code; }
}  // namespace

extern "C" {

const char *ribbit_fasta_last_error(void) { return g_fasta_error.c_str(); }

int ribbit_fasta_open(const char *path, int pinned, RibbitFastaReader **out) {
    if (!path || !out) return fail(RIBBIT_E_ARG, "null argument");
    *out = nullptr;
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(RIBBIT_E_ARG, std::string("cannot open ") + path);
    RibbitFastaReader *r = new (std::nothrow) RibbitFastaReader();
    if (!r) { std::fclose(f); return fail(RIBBIT_E_NOMEM, "out of host memory"); }
    r->f = f;
    r->pinned = pinned != 0;
    r->block.resize(BLOCK);
    *out = r;
    return RIBBIT_OK;
}

int ribbit_fasta_next(RibbitFastaReader *r, const char **name, const char **bases, int64_t *length, int *is_last) {
    if (!r || !name || !bases || !length || !is_last) return fail(RIBBIT_E_ARG, "null argument");
    if (r->done) return 0;
    for (;;) {
        if (r->at == r->have) {
            if (r->eof) break;
            r->have = std::fread(r->block.data(), 1, r->block.size(), r->f);
            r->at = 0;
            if (r->have < r->block.size()) r->eof = true;
            if (r->have == 0) break;
        }
        const char *p = r->block.data() + r->at, *end = r->block.data() + r->have;
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *stop = nl ? nl : end;
        if (r->at_line_start && !r->in_header && *p == '>') { r->in_header = true; r->header.clear(); ++p; }
        if (r->in_header) {
            r->header.append(p, (size_t)(stop - p));
        } else {
            const int rc = r->append(p, (size_t)(stop - p));
            if (rc) return fail(rc, r->error);
        }
        r->at = (size_t)((nl ? nl + 1 : end) - r->block.data());
        r->at_line_start = nl != nullptr;
        if (!nl) continue;                       // line continues in the next block
        if (r->in_header) {
            // ribbit.cpp:271-276: a header line ends the previous record if it has any bases
            r->in_header = false;
            const size_t sp = r->header.find(' ');
            std::string fresh = r->header.substr(0, sp);
            if (r->cur_len > 0) {
                r->out_name = r->next_name;
                r->next_name = fresh;
                *name = r->out_name.c_str();
                *bases = r->cur.p;
                *length = (int64_t)r->cur_len;
                *is_last = 0;
                { std::lock_guard<std::mutex> lk(r->mu); r->lent.push_back(r->cur); }
                r->cur = Buffer{};
                r->cur_len = 0;
                return 1;
            }
            r->next_name = fresh;
        }
    }
    // end of file; an unterminated header line still names the last record
    if (r->in_header) {
        r->in_header = false;
        const size_t sp = r->header.find(' ');
        std::string fresh = r->header.substr(0, sp);
        if (r->cur_len > 0) {
            // previous record first; the (empty) record the header names comes with the next call
            r->out_name = r->next_name;
            r->next_name = fresh;
            *name = r->out_name.c_str(); *bases = r->cur.p; *length = (int64_t)r->cur_len; *is_last = 0;
            { std::lock_guard<std::mutex> lk(r->mu); r->lent.push_back(r->cur); }
            r->cur = Buffer{};
            r->cur_len = 0;
            return 1;
        }
        r->next_name = fresh;
    }
    // ribbit.cpp:280: the last record is processed unconditionally, also when it is empty
    r->done = true;
    r->out_name = r->next_name;
    if (!r->cur.p) {
        const int rc = r->take(MIN_BUFFER, r->cur);
        if (rc) return fail(rc, r->error);
    }
    *name = r->out_name.c_str();
    *bases = r->cur.p;
    *length = (int64_t)r->cur_len;
    *is_last = 1;
    { std::lock_guard<std::mutex> lk(r->mu); r->lent.push_back(r->cur); }
    r->cur = Buffer{};
    r->cur_len = 0;
    return 1;
}

int ribbit_fasta_release(RibbitFastaReader *r, const char *bases) {
    if (!r || !bases) return fail(RIBBIT_E_ARG, "null argument");
    std::lock_guard<std::mutex> lk(r->mu);
    for (size_t i = 0; i < r->lent.size(); ++i)
        if (r->lent[i].p == bases) {
            r->spare.push_back(r->lent[i]);
            r->lent.erase(r->lent.begin() + (long)i);
            return RIBBIT_OK;
        }
    return fail(RIBBIT_E_ARG, "not a buffer this reader handed out");
}

int ribbit_fasta_close(RibbitFastaReader *r) {
    if (!r) return RIBBIT_OK;
    if (r->f) std::fclose(r->f);
    r->dealloc(r->cur);
    for (Buffer &b : r->spare) r->dealloc(b);
    for (Buffer &b : r->lent) r->dealloc(b);
    delete r;
    return RIBBIT_OK;
}

}  // extern "C"
