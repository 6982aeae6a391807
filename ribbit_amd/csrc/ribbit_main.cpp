// ribbit_main.cpp -- command-line front end with ribbit's surface (ribbit.cpp): same options, same
// defaults and quirks, FASTA in, BED out, the reference's progress lines on stderr.  Everything
// between reading a record and writing its BED rows goes through the C ABI of libribbit_hip.so.
//
//   ribbit-hip -i in.fa [-o out.bed] [-m 2] [-M 100] [-p 0.85] [-l N|file] [--min-units N|file] [--perfect-units N|file]
//              [--devices 0,1,...] [--jobs N]
//
// Records are independent (ribbit.cpp:269-280 handles them one after the other); here up to --jobs of them are in
// flight at once PER GPU, each on its own handle / HIP streams, so that the upload and GPU scans of one record overlap
// the host merges and refinement of the others (long-read inputs: thousands of 10-100 kb records); with --devices the
// records are dealt over several GPUs, one handle set per device (partitioning by record: no halo, no
// exchange).  Output order is the input order.  The file is read by ribbit_fasta_* (block reads, line bodies copied once into page-locked buffers that
// the GPU uploads from asynchronously and refinement reads in place) instead of getline + string +=.
//
// Reproduced quirks: -p is accepted and ignored (Q1); without -o the BED rows go to
// stderr (Q2); --help exits with status 1 (Q3); the record name ends at the first space and the last
// record is processed even when the file is empty (Q4).
#include <algorithm>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "ribbit_hip.h"

namespace {

struct Options {
    std::string fasta, out;
    int min_motif = 2, max_motif = 100;          // global_variables.cpp:21-22
    bool has_min_length = false, has_min_units = false, has_perfect_units = false;
    std::string min_length, min_units, perfect_units;
    int device = 0;
    std::vector<int> devices;                     // --devices / RIBBIT_DEVICES: GPUs the records are dealt over (empty: `device` alone)
    int jobs = 0;                                 // records in flight PER DEVICE; 0 = automatic
    std::string timing;                           // --timing FILE: a JSON record of the run (the reference has cerr progress lines only)
};

const char *kHelp =
    "Below are the running options for the tool.:\n"
    "  -h [ --help ]                 Ribbit tool identifies short tandem repeats with allowed levels of inpurity.\n"
    "  -i [ --input-file ] arg       File path for the input fasta file.\n"
    "  -o [ --output-file ] arg      File path for the input fasta file.\n"
    "  -m [ --min-motif-length ] arg The minimum length of the motif of the repeats to be identified. Default: 2\n"
    "  -M [ --max-motif-length ] arg The maximum length of the motif of the repeats to be identified, Default: 100\n"
    "  -p [ --purity ] arg           Threshold value for cotinuous number of ones found in a seed. Default: 0.85\n"
    "  -l [ --min-length ] arg       The minimum length of the repeat. Default: 12\n"
    "  --min-units arg               The minimum number of units of the repeat. Integer, or a tab separated file\n"
    "                                with the motif size and the unit cutoff. Default: 2\n"
    "  --perfect-units arg           The minimum number of complete units of the repeat. Integer, or a tab\n"
    "                                separated file with the motif size and the unit cutoff. Default: 2\n"
    "  --jobs arg                    (ribbit-hip) FASTA records processed side by side on each GPU. Default: automatic\n"
    "  --device arg                  (ribbit-hip) GPU ordinal. Default: 0\n"
    "  --devices arg                 (ribbit-hip) GPU ordinals, comma separated (or RIBBIT_DEVICES): the records of the\n"
    "                                FASTA are dealt over these GPUs, the longest of the look-ahead first; BED rows\n"
    "                                keep the input order\n"
    "  --timing arg                  (ribbit-hip) write a JSON record of the run to this file: records, bases, wall time and\n"
    "                                the wall time per stage summed over the records\n";

[[noreturn]] void die(const std::string &msg) {        // argument errors: main thread, before any worker exists
    std::cerr << "ribbit-hip: " << msg << "\n";
    std::exit(1);
}

// a failure of the GPU path inside the record pipeline: carried to main(), which lets the workers drain, closes the
// handles and returns 1 (exiting from a worker would run static destructors under live threads)
struct PathError { std::string what; };

bool parse_device_list(const std::string &value, std::vector<int> &out) {
    out.clear();
    size_t at = 0;
    while (at <= value.size()) {
        const size_t comma = std::min(value.find(',', at), value.size());
        const std::string item = value.substr(at, comma - at);
        if (item.empty() || !std::all_of(item.begin(), item.end(), [](unsigned char c) { return std::isdigit(c); })) return false;
        out.push_back(std::atoi(item.c_str()));
        at = comma + 1;
    }
    return !out.empty();
}

// returns 0 for --help (the caller exits 1, as the reference does), 1 on success
int parse_arguments(int argc, char **argv, Options &o) {
    static const std::map<std::string, std::string> longs = {
        {"help", "h"}, {"input-file", "i"}, {"output-file", "o"}, {"min-motif-length", "m"}, {"max-motif-length", "M"},
        {"purity", "p"}, {"min-length", "l"}, {"min-units", "U"}, {"perfect-units", "P"}, {"device", "D"}, {"jobs", "J"}, {"devices", "G"}, {"timing", "T"}};
    bool help = false;
    for (int a = 1; a < argc; ++a) {
        std::string arg = argv[a], key, value;
        bool has_value = false;
        if (arg.rfind("--", 0) == 0) {
            const size_t eq = arg.find('=');
            const std::string name = arg.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            auto it = longs.find(name);
            if (it == longs.end()) die("unrecognised option '" + arg + "'");
            key = it->second;
            if (eq != std::string::npos) { value = arg.substr(eq + 1); has_value = true; }
        } else if (arg.size() >= 2 && arg[0] == '-') {
            key = arg.substr(1, 1);
            if (std::string("hiomMpl").find(key) == std::string::npos) die("unrecognised option '" + arg + "'");
            if (arg.size() > 2) { value = arg.substr(2); has_value = true; }
        } else {
            die("too many positional options have been specified on the command line");
        }
        if (key == "h") { help = true; continue; }
        if (!has_value) {
            if (a + 1 >= argc) die("the required argument for option '" + arg + "' is missing");
            value = argv[++a];
        }
        if (key == "i") o.fasta = value;
        else if (key == "o") o.out = value;
        else if (key == "m") o.min_motif = std::atoi(value.c_str());
        else if (key == "M") o.max_motif = std::atoi(value.c_str());
        else if (key == "p") { /* declared, never read (ribbit.cpp:92) */ }
        else if (key == "l") { o.has_min_length = true; o.min_length = value; }
        else if (key == "U") { o.has_min_units = true; o.min_units = value; }
        else if (key == "P") { o.has_perfect_units = true; o.perfect_units = value; }
        else if (key == "D") o.device = std::atoi(value.c_str());
        else if (key == "J") o.jobs = std::atoi(value.c_str());
        else if (key == "T") o.timing = value;
        else if (key == "G") { if (!parse_device_list(value, o.devices)) die("--devices wants a comma separated list of GPU ordinals, got '" + value + "'"); }
    }
    if (help) { std::cerr << kHelp << "\n"; return 0; }                       // ribbit.cpp:114-117
    if (o.fasta.empty()) { std::cerr << "ERROR: Please specify an input fasta file!\n"; return 0; }   // :122-126
    return 1;
}

bool is_number(const std::string &s) { return !s.empty() && std::all_of(s.begin(), s.end(), [](unsigned char c) { return std::isdigit(c); }); }

// parseDualtypeArgs, ribbit.cpp:25-64: one integer for every motif size in range, or a two-column TSV
void dual_type(const std::string &value, std::map<int, int> &table, int m_lo, int m_hi) {
    if (is_number(value)) {
        for (int k = m_lo; k <= m_hi; ++k) table[k] = std::atoi(value.c_str());
        return;
    }
    std::ifstream in(value);
    std::string line;
    while (std::getline(in, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;
        table[std::atoi(line.substr(0, tab).c_str())] = std::atoi(line.substr(tab + 1).c_str());
    }
}

// ribbit.cpp:143-174 and the factor completion of :210-235
void build_refine_params(const Options &o, RibbitRefineParams &prm) {
    ribbit_refine_params_default(&prm, o.min_motif, o.max_motif);
    if (!o.has_min_length && !o.has_min_units && !o.has_perfect_units) return;
    std::map<int, int> min_length, min_units, perfect_units;
    if (o.has_min_length) dual_type(o.min_length, min_length, o.min_motif, o.max_motif);
    else if (o.has_min_units) {
        dual_type(o.min_units, min_units, o.min_motif, o.max_motif);
        for (auto &kv : min_units) min_length[kv.first] = kv.first * kv.second;
    } else {
        for (int k = o.min_motif; k <= o.max_motif; ++k) min_length[k] = std::max(12, 2 * k);
    }
    if (o.has_perfect_units) dual_type(o.perfect_units, perfect_units, o.min_motif, o.max_motif);
    else for (int m = 1; m <= o.max_motif; ++m) perfect_units[m] = m == 1 ? 8 : m == 2 ? 4 : m == 3 ? 3 : 2;
    for (int m = o.min_motif; m <= o.max_motif; ++m)
        for (int f = 1; f <= m / 2; ++f) {
            if (m % f) continue;
            if (!min_length.count(f)) min_length[f] = min_length[m];
            if (!perfect_units.count(f)) perfect_units[f] = perfect_units[m] * (m / f);
        }
    std::memset(prm.min_length, 0, sizeof prm.min_length);
    std::memset(prm.perfect_units, 0, sizeof prm.perfect_units);
    for (auto &kv : min_length) if (kv.first >= 0 && kv.first < RIBBIT_TABLE) prm.min_length[kv.first] = kv.second;
    for (auto &kv : perfect_units) if (kv.first >= 0 && kv.first < RIBBIT_TABLE) prm.perfect_units[kv.first] = kv.second;
}

void check(int rc) {
    if (rc != RIBBIT_OK) throw PathError{std::string("GPU path failed: ") + ribbit_hip_last_error()};
}

size_t count_failed(const RibbitSeed *s, size_t n) {
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) c += s[i].type == RIBBIT_RANK_N;
    return c;
}

// RIBBIT_PROFILE=1: wall time per stage, summed over the records, printed at exit
double g_stage_ms[6] = {0, 0, 0, 0, 0, 0};
std::mutex g_stage_mu;
struct StageClock {
    int slot;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit StageClock(int s) : slot(s) {}
    ~StageClock() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(g_stage_mu);
        g_stage_ms[slot] += ms;
    }
};

// processSequence (fasta_utils.cpp:59-250) through the C ABI, with the reference's progress lines
// Refinement of ONE record over several GPUs (ribbit_hip_adopt_dispatch): the dispatched seeds in as many slices as there are
// handles, equal numbers of seeds each (a seed's cost varies by orders of magnitude, but over millions of seeds the slices even
// out); every helper loads the record on its own GPU, makes the composed planes there and refines its slice on its GPU's share of
// the host threads; the texts in order are the record's BED.  An alignment with an empty query in any slice (it would see the
// previous seed's CIGAR, which may be another slice's) makes the record be refined again in one piece, on `h`.
struct Helper { RibbitHandle *h; int host_threads; };
bool refine_over_devices(RibbitHandle *h, const std::vector<Helper> &helpers, const RibbitRefineParams &prm, const std::string &name, const char *bases,
                         int64_t length, const RibbitSeed *d, size_t nd, std::ostream &out, std::ostream &log) {
    const size_t parts = helpers.size() + 1;
    const std::vector<RibbitSeed> all(d, d + nd);          // (`d` is `h`'s own list, which adopting a slice replaces)
    std::vector<std::string> text(parts), error(parts);
    std::vector<int> empty_query(parts, 0);
    auto slice = [&](size_t k, RibbitHandle *hk, bool load) {
        const size_t lo = nd * k / parts, hi = nd * (k + 1) / parts;
        if (load && ribbit_hip_load_record_pinned(hk, bases, length) != RIBBIT_OK) { error[k] = ribbit_hip_last_error(); return; }
        const char *t = nullptr;
        size_t len = 0;
        if (ribbit_hip_adopt_dispatch(hk, all.data() + lo, hi - lo) != RIBBIT_OK || ribbit_hip_refine_bed(hk, &prm, name.c_str(), &t, &len) != RIBBIT_OK) {
            error[k] = ribbit_hip_last_error();
            return;
        }
        text[k].assign(t, len);
        empty_query[k] = ribbit_hip_refine_met_empty_query(hk);
    };
    for (const Helper &hp : helpers) check(ribbit_hip_set_host_threads(hp.h, hp.host_threads));      // (anything that can throw: before the first thread exists)
    std::vector<std::thread> pool;
    pool.reserve(parts);
    try {
        for (size_t k = 1; k < parts; ++k) pool.emplace_back(slice, k, helpers[k - 1].h, true);
    } catch (...) {            // a thread that could not start: its slice and the ones after it run here, one after the other
        for (size_t k = pool.size() + 1; k < parts; ++k) slice(k, helpers[k - 1].h, true);
    }
    slice(0, h, false);
    for (std::thread &t : pool) t.join();
    for (size_t k = 0; k < parts; ++k)
        if (!error[k].empty()) throw PathError{"GPU path failed: " + error[k]};
    bool redo = false;
    for (size_t k = 1; k < parts; ++k) redo = redo || empty_query[k] != 0;
    if (redo) {
        const char *t = nullptr;
        size_t len = 0;
        check(ribbit_hip_adopt_dispatch(h, all.data(), nd));
        check(ribbit_hip_refine_bed(h, &prm, name.c_str(), &t, &len));
        out.write(t, (std::streamsize)len);
        log << "[devices] an alignment with an empty query at the head of a slice: the record was refined again in one piece\n";
        return false;
    }
    for (size_t k = 0; k < parts; ++k) out.write(text[k].data(), (std::streamsize)text[k].size());
    return true;
}

void process_sequence(RibbitHandle *h, const RibbitRefineParams &prm, const std::string &name, const char *bases, int64_t length,
                      std::ostream &out, std::ostream &log, const std::vector<Helper> *helpers = nullptr) {
    const time_t t0 = time(0);
    auto secs = [&]() { return difftime(time(0), t0); };
    { StageClock c(0); check(ribbit_hip_load_record_pinned(h, bases, length)); }
    log << "Generated shift XORs!\t Time elapsed:" << secs() << "secs\n";
    const RibbitSeed *p, *s, *a;
    size_t np, ns, na;
    { StageClock c(1); check(ribbit_hip_seeds_perfect(h, &p, &np)); }
    log << "Total number of perfect seeds: " << np << "\t Time elapsed: " << secs() << "secs\n";
    { StageClock c(2); check(ribbit_hip_seeds_substitutions(h, &p, &np, &s, &ns)); }
    log << "Total number of seeds considering substitutions: " << np + ns - count_failed(p, np) - count_failed(s, ns)
              << "\t Time elapsed: " << secs() << "secs\n";
    { StageClock c(3); check(ribbit_hip_seeds_anchored(h, &p, &np, &s, &ns, &a, &na)); }
    log << "Generated anchored shift XORs!\t Time elapsed: " << secs() << "secs\n";
    log << "Total number of seeds considering indels: "
              << np + ns + na - count_failed(p, np) - count_failed(s, ns) - count_failed(a, na) << "\t Time elapsed: " << secs() << "secs\n";
    const RibbitSeed *d;
    size_t nd;
    { StageClock c(4); check(ribbit_hip_dispatch_seeds(h, &d, &nd)); }
    // one record over several GPUs: only worth it from a few hundred thousand seeds on (RIBBIT_SHARD_MIN_SEEDS: a test hook)
    static const size_t shard_min = std::getenv("RIBBIT_SHARD_MIN_SEEDS") ? (size_t)std::atoll(std::getenv("RIBBIT_SHARD_MIN_SEEDS")) : 400000;
    if (helpers && !helpers->empty() && nd >= shard_min && nd >= 2 * (helpers->size() + 1)) {
        StageClock c(5);
        const bool sharded = refine_over_devices(h, *helpers, prm, name, bases, length, d, nd, out, log);
        if (std::getenv("RIBBIT_PROFILE"))
            log << "[devices] refinement of " << name << ": " << nd << " dispatched seeds " << (sharded ? "in " : "NOT in ") << helpers->size() + 1 << " slices over as many handles\n";
    } else {
        const char *text;
        size_t len;
        { StageClock c(5); check(ribbit_hip_refine_bed(h, &prm, name.c_str(), &text, &len)); }
        out.write(text, (std::streamsize)len);
    }
    log << "Total number of seeds that are processed for alignment: " << nd << "\t Time elapsed: " << secs() << "secs\n";
}

}  // namespace

int main(int argc, char **argv) {
    Options opt;
    if (!parse_arguments(argc, argv, opt)) return 1;                          // ribbit.cpp:193-195

    std::ofstream file;
    if (!opt.out.empty()) file.open(opt.out);
    std::ostream &out = opt.out.empty() ? std::cerr : file;                   // ribbit.cpp:199-205

    const auto t_run0 = std::chrono::steady_clock::now();
    RibbitRefineParams prm;
    build_refine_params(opt, prm);
    std::cerr << "Minimum motif:\t" << opt.min_motif << "\n";
    std::cerr << "Maximum motif:\t" << opt.max_motif << "\n";
    std::cerr << "Purity threshold: " << 0.85f << "\n";

    RibbitScanParams scan;
    ribbit_scan_params_default(&scan, opt.min_motif, opt.max_motif);
    // GPUs: --devices, else RIBBIT_DEVICES, else --device alone.  Records are independent (ribbit.cpp:269-280 handles them one
    // after the other), so several GPUs simply take different records; a device may be listed twice (two handle sets on it).
    std::vector<int> devices = opt.devices;
    if (devices.empty())
        if (const char *env = std::getenv("RIBBIT_DEVICES"))
            if (!parse_device_list(env, devices)) die(std::string("RIBBIT_DEVICES wants a comma separated list of GPU ordinals, got '") + env + "'");
    if (devices.empty()) devices.push_back(opt.device);
    const int ndev = (int)devices.size();
    RibbitHandle *h = nullptr;
    if (ribbit_hip_open(&scan, devices[0], &h) != RIBBIT_OK) die(std::string("GPU path failed: ") + ribbit_hip_last_error());

    // ---- record pipeline: the reader (this thread) parses records; `jobs` workers per GPU process them on their own
    // handles; results are written in input order.  A record weighs ceil(length / 4 Mbp) of its GPU's `jobs` tokens (at
    // most all of them), so many reads run side by side while a chromosome has one GPU and its share of the host threads
    // to itself.  A free worker takes the LONGEST record of the look-ahead window (the longest-first dealing of
    // independent units over devices, done as the records stream in), but never passes over the oldest record more than
    // 2 x workers times, so the output never waits on a starved record.
    unsigned cores = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u * (unsigned)ndev));
    if (const char *env = std::getenv("RIBBIT_THREADS")) cores = (unsigned)std::max(1, std::atoi(env));
    const unsigned dev_cores = std::max(1u, cores / (unsigned)ndev);      // host threads behind one GPU
    int jobs = (int)std::max(1u, std::min(8u, dev_cores / 2));
    if (const char *env = std::getenv("RIBBIT_JOBS")) jobs = std::max(1, std::atoi(env));
    if (opt.jobs > 0) jobs = opt.jobs;
    jobs = std::min(jobs, 64);
    const int workers = jobs * ndev;
    struct Record { size_t index; std::string name; const char *bases; int64_t length; };
    struct Result { std::string bed, log; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Record> queue;
    std::map<size_t, Result> done;
    size_t next_out = 0;
    bool reader_done = false;
    bool failed = false;            // some record's GPU path failed: everybody drains
    std::string failure;
    std::vector<int> tokens((size_t)ndev, jobs);
    int front_passed_over = 0;      // how often the oldest queued record has been passed over for a longer one
    std::vector<int64_t> dev_bases((size_t)ndev, 0);      // RIBBIT_PROFILE: bases each GPU has taken
    std::vector<size_t> dev_records((size_t)ndev, 0);
    RibbitFastaReader *reader = nullptr;
    if (ribbit_fasta_open(opt.fasta.c_str(), 1, &reader) != RIBBIT_OK) {
        // the reference's ifstream on a missing file simply yields no lines: one empty, unnamed record (Q4)
        std::cerr << "ribbit-hip: " << ribbit_fasta_last_error() << "\n";
    }

    auto flush_ready = [&]() {                      // call with mu held
        for (auto it = done.find(next_out); it != done.end(); it = done.find(next_out)) {
            std::cerr << it->second.log;
            out.write(it->second.bed.data(), (std::streamsize)it->second.bed.size());
            done.erase(it);
            ++next_out;
        }
    };
    // heavy: this GPU's first worker.  A record that takes all of a GPU's job tokens runs alone on it, and always on that worker's
    // handle: the buffers of a chromosome (gigabytes of page-locked and device memory, a second to allocate) exist once per
    // GPU, not once per worker (eight handles each meeting their first chromosome cost the whole-genome run 8 of its 83 s).
    auto worker = [&](RibbitHandle *wh, int dev, bool heavy) {
        for (;;) {
            Record rec;
            int weight;
            {
                std::unique_lock<std::mutex> lk(mu);
                size_t pick = 0;
                for (;;) {
                    if (!queue.empty()) {
                        pick = 0;
                        if (front_passed_over < 2 * workers)
                            for (size_t i = 1; i < queue.size() && i < (size_t)workers; ++i)
                                if (queue[i].length > queue[pick].length) pick = i;
                        weight = (int)std::min<size_t>((size_t)jobs, (size_t)queue[pick].length / 4000000 + 1);
                        if (tokens[(size_t)dev] >= weight && (weight < jobs || heavy || jobs == 1)) break;
                    } else if (reader_done) {
                        return;
                    }
                    cv.wait(lk);
                }
                front_passed_over = pick == 0 ? 0 : front_passed_over + 1;
                rec = std::move(queue[pick]);
                queue.erase(queue.begin() + (std::ptrdiff_t)pick);
                tokens[(size_t)dev] -= weight;
                dev_bases[(size_t)dev] += rec.length;
                ++dev_records[(size_t)dev];
            }
            cv.notify_all();
            std::ostringstream bed, log;
            bool ok = true;
            std::string why;
            {
                bool skip;
                { std::lock_guard<std::mutex> lk(mu); skip = failed; }
                if (!skip) {
                    try {
                        check(ribbit_hip_set_host_threads(wh, (int)std::max(1u, dev_cores * (unsigned)weight / (unsigned)jobs)));
                        log << "Processing sequence " << rec.name << "\n";
                        process_sequence(wh, prm, rec.name, rec.bases, rec.length, bed, log);
                    } catch (const PathError &e) { ok = false; why = e.what; }
                }
            }
            ribbit_fasta_release(reader, rec.bases);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!ok && !failed) { failed = true; failure = why; }
                done[rec.index] = Result{bed.str(), log.str()};
                tokens[(size_t)dev] += weight;
                if (!failed) flush_ready();
            }
            cv.notify_all();
        }
    };
    std::vector<RibbitHandle *> handles{h};
    std::vector<int> handle_dev{0};
    for (int j = 1; j < workers; ++j) {
        RibbitHandle *extra = nullptr;
        const int dev = j % ndev;              // worker j serves GPU j mod ndev: every GPU gets `jobs` of them
        if (ribbit_hip_open(&scan, devices[(size_t)dev], &extra) != RIBBIT_OK) { failed = true; failure = std::string("GPU path failed: ") + ribbit_hip_last_error(); break; }
        handles.push_back(extra);
        handle_dev.push_back(dev);
    }
    std::vector<std::thread> pool;
    for (size_t j = 0; j < handles.size(); ++j) pool.emplace_back(worker, handles[j], handle_dev[j], j < (size_t)ndev);      // handles 0 .. ndev-1: one per GPU

    // ribbit.cpp:269-279 -- records as the reference's getline loop delimits them; :280 -- the last record is processed
    // unconditionally and WITHOUT the "Processing sequence" line, also for an empty file (Q4): it bypasses the pipeline
    // once the pipeline has drained
    size_t n_records = 0;
    std::string last_name;
    const char *last_bases = nullptr;
    int64_t last_length = 0;
    for (;;) {
        const char *name = "", *bases = nullptr;
        int64_t length = 0;
        int is_last = 1;
        const int got = reader ? ribbit_fasta_next(reader, &name, &bases, &length, &is_last) : 0;
        if (got < 0) { std::lock_guard<std::mutex> lk(mu); failed = true; failure = ribbit_fasta_last_error(); break; }
        if (got == 0) break;
        if (is_last) { last_name = name; last_bases = bases; last_length = length; break; }
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return queue.size() < (size_t)(2 * workers) || failed; });       // bounded look-ahead
        if (failed) break;
        queue.push_back(Record{n_records++, name, bases, length});
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        reader_done = true;
        cv.notify_all();
    }
    for (std::thread &t : pool) t.join();
    int status = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) flush_ready();
    }
    if (!failed) {
        try {
            // The last record runs here, after the workers have gone: every other GPU of --devices is idle by now, so a long last
            // record -- a FASTA with ONE chromosome, above all -- has its refinement dealt over all of them, a slice of the
            // dispatched seeds per GPU with that GPU's share of the host threads (refine_over_devices; scans and merges stay on
            // the first GPU: 0.55 of a chromosome's 2 s).
            std::vector<Helper> helpers;
            for (int d = 1; d < ndev && (size_t)d < handles.size(); ++d) helpers.push_back(Helper{handles[(size_t)d], (int)dev_cores});
            check(ribbit_hip_set_host_threads(h, helpers.empty() ? 0 : (int)dev_cores));
            static const char kNoBases[1] = {0};
            process_sequence(h, prm, last_name, last_bases ? last_bases : kNoBases, last_length, out, std::cerr, &helpers);
        } catch (const PathError &e) { failed = true; failure = e.what; }
    }
    if (failed) { std::cerr << "ribbit-hip: " << failure << "\n"; status = 1; }
    for (size_t j = 1; j < handles.size(); ++j) ribbit_hip_close(handles[j]);
    ribbit_hip_close(h);
    if (reader) ribbit_fasta_close(reader);
    if (std::getenv("RIBBIT_PROFILE")) {
        char bus[64] = {0};
        for (int d = 0; d < ndev; ++d)
            if (ribbit_hip_device_pci_bus_id(devices[(size_t)d], bus, sizeof bus) == RIBBIT_OK)
                std::cerr << "[device] slot " << d << " is GPU " << devices[(size_t)d] << " at PCI " << bus << "\n";
    }
    if (std::getenv("RIBBIT_PROFILE") && ndev > 1)
        for (int d = 0; d < ndev; ++d)
            std::cerr << "[devices] slot " << d << " (GPU " << devices[(size_t)d] << "): " << dev_records[(size_t)d] << " records, " << dev_bases[(size_t)d] << " bases\n";
    if (!opt.timing.empty()) {
        // (stage times are wall clock per record, summed: with several records in flight their sum exceeds the run's wall time)
        std::ofstream tf(opt.timing);
        const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run0).count();
        int64_t total_bases = last_length;
        for (int d = 0; d < ndev; ++d) total_bases += dev_bases[(size_t)d];
        tf << "{\"records\": " << n_records + 1 << ", \"bases\": " << total_bases << ", \"wall_s\": " << wall_s << ", \"status\": " << status
           << ", \"min_motif\": " << opt.min_motif << ", \"max_motif\": " << opt.max_motif << ", \"devices\": " << ndev << ", \"jobs_per_device\": " << jobs
           << ", \"stage_ms_summed_over_records\": {\"load\": " << g_stage_ms[0] << ", \"perfect\": " << g_stage_ms[1] << ", \"substitutions\": " << g_stage_ms[2]
           << ", \"anchored\": " << g_stage_ms[3] << ", \"dispatch\": " << g_stage_ms[4] << ", \"refine_and_bed\": " << g_stage_ms[5] << "}}\n";
    }
    if (std::getenv("RIBBIT_PROFILE"))
        std::cerr << "[stages, ms over all records] load " << g_stage_ms[0] << "  perfect " << g_stage_ms[1] << "  substitutions "
                  << g_stage_ms[2] << "  anchored " << g_stage_ms[3] << "  dispatch " << g_stage_ms[4] << "  refine+BED " << g_stage_ms[5] << "\n";
    return status;
}
