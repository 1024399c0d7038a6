// mmseq -- drop-in for the reference's `mmseq` CLI (src/mmseq.cpp:179-1728) with the Gibbs loop
// (and the EM that seeds it) running on MI355X through libmmgibbs' C ABI (include/mmgibbs.h).
//
//   Usage: mmseq [OPTIONS...] hits_file output_base          (same flags as src/mmseq.cpp:156-177)
//
// What is the same as the reference: flags, validation and exit codes (:183-299), hits-file input
// (hitsio), transcript / hit-set index order (:395-441), every output file, its column names, row
// order and default 6-significant-digit formatting (:682-695, :823-831, :1033-1108, :1469-1669).
// What differs on purpose: random numbers come from keyed Philox streams (results do not depend on
// a thread count, cf. :834-838), the per-iteration "\r" progress line is throttled (:852), traces
// are written after the loop from the device-resident trace, VLAs are heap vectors (:1308-1348).
// There is no CPU sampler here: without a HIP device the program stops with an error.
#include <omp.h>
#include <atomic>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <charconv>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/mmgibbs.h"
#include "hitsio.hpp"
#include "huffenc.hpp"
#include "numerics.hpp"

#ifndef MMSEQ_VERSION
#define MMSEQ_VERSION "1.0.11-mi355x"
#endif

using namespace std;

static void printUsage(ostream &out)
{
    out << "Usage: mmseq [OPTIONS...] hits_file output_base" << endl
        << endl
        << "Mandatory arguments:" << endl
        << "  hits_file          hits file generated with `bam2hits`\n"
        << "  output_base        base name for output files" << endl
        << endl
        << "Optional arguments:\n"
        << "  -alpha FLOAT       value of alpha in Gamma prior for mu (default: 0.1)" << endl
        << "  -beta FLOAT        value of beta in Gamma prior for mu (default: 0.1)" << endl
        << "  -max_em_iter INT   maximum number of EM iterations (default: 1000)" << endl
        << "  -epsilon FLOAT     minimum loglik ratio between successive EM iterations (default: 0.1)" << endl
        << "  -gibbs_iter INT    number of Gibbs iterations (default: 16384)" << endl
        << "  -gibbs_ss INT      subsampling interval for Gibbs output (default: gibbs_iter/1024)" << endl
        << "  -seed INT          seed for the PRNG in thread 0 (default: 1234)" << endl
        << "  -percentiles STR   comma-separated list of real-scale marginal posterior percentiles to output (default: "
           "\"5,25,50,75,95\")"
        << endl
        << "  -debug             output additional diagnostic files" << endl
        << "  -help              print this help message" << endl
        << "  -version           print the version" << endl
        << "  -device INT        first HIP device to run on (default: 0)        [MI355X build]" << endl
        << "  -gpus INT          number of devices, device .. device+gpus-1 (default: 1).  With one chain the reads are sharded" << endl
        << "                     over them (same results as on one device); with -chains >= gpus every device runs chains/gpus chains" << endl
        << "  -chains INT        independent Gibbs chains (default: 1); log_mu, sd and mcse pool all chains, traces are chain 0's" << endl
        << "  -em_one_device     with -gpus > 1 and one chain: run the EM on the first device alone instead of over the read shards" << endl
        << endl;
}

// the fields of `text` between separators (empty fields dropped)
static vector<string> split_fields(const string &text, char sep)
{
    vector<string> out;
    string cur;
    for (char ch : text) {
        if (ch != sep) { cur += ch; continue; }
        if (!cur.empty()) out.push_back(cur);
        cur.clear();
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

static bool is_power_of_two(unsigned v) { return v != 0 && (v & (v - 1)) == 0; }

// Command line: a table of options -- name, the variable it sets, how its value is read -- walked once.  Same flags, defaults,
// messages and exit codes as the reference's loop at src/mmseq.cpp:206-276 (tests/test_cli.py holds them), plus -device / -gpus /
// -chains / -em_one_device of this build.
struct CliOption {
    const char *name;
    enum Kind { REAL, INT, FLAG, LIST, HELP, VERSION } kind;
    void *target;
};

// the background writer of .k / .M (if any) finishes its files before the process leaves on a device error: the reference has
// written both by then (src/mmseq.cpp:682-695), and exit() must not run under a thread that is still formatting
static std::thread *g_background_writer = nullptr;
// the thread that brings the HIP runtime up while the file is read: exit() under a thread that is still INSIDE the runtime's
// initialisation tears the runtime down under its feet (found by tools/hitsio_fuzz.py: 7 of 800 runs on damaged headers ended in the
// sanitizer's allocator instead of with exit code 1) -- every exit of the main thread waits for it
static std::thread *g_device_warmup = nullptr;
[[noreturn]] static void leave(int code)
{
    if (g_device_warmup && g_device_warmup->joinable()) g_device_warmup->join();
    exit(code);
}
// Worker threads (the trace writers and their fetchers) never call exit(): exit() runs the atexit handlers and the HIP / RCCL
// teardown under the main thread's feet, and two workers failing together would both join the background writer.  A worker RECORDS
// its error (first one wins) and returns; the main thread sees the flag, stops the workers (g_stop_workers: wakes them, joins them)
// and leaves with the message -- exit() on the main thread only.
static std::atomic<bool> g_worker_failed{false};
static std::mutex g_worker_mu;
static std::string g_worker_msg;
static std::function<void()> g_stop_workers; // set while worker threads of main() run: makes them return and joins them
static bool worker_failed(int rc, const char *what)
{
    if (rc == 0) return false;
    std::lock_guard<std::mutex> lk(g_worker_mu);
    if (!g_worker_failed.load()) g_worker_msg = std::string(mmg_last_error()) + " (" + what + ")";
    g_worker_failed.store(true);
    return true;
}
[[noreturn]] static void main_thread_exit(const std::string &msg)
{
    cerr << "Error: " << msg << endl;
    if (g_stop_workers) { auto f = g_stop_workers; g_stop_workers = nullptr; f(); }
    if (g_background_writer && g_background_writer->joinable()) g_background_writer->join();
    leave(1);
}
#define MMG_TRY(expr)                                                                     \
    do {                                                                                  \
        if ((expr) != 0) main_thread_exit(std::string(mmg_last_error()) + " (" + #expr + ")"); \
    } while (0)
// in a worker thread: record and leave the enclosing function / lambda
#define MMG_TRY_WORKER(expr) do { if (worker_failed((expr), #expr)) return; } while (0)

// gzip text sink: ONE standard gzip member whose deflate stream is produced chunk-wise in parallel.
// Every chunk is compressed independently as raw deflate and closed with a sync flush (byte-aligned,
// no dictionary carried over), so the concatenation is a valid deflate stream; the CRCs are merged with
// crc32_combine.  Any gzip reader (zlib, Boost gzip_decompressor of mmcollapse, R) reads it unchanged.
// Numbers are formatted with "%g" == default ostream formatting (6 significant digits).
// "%g" (6 significant digits, what the reference's default ostream formatting prints) without going through printf:
// to_chars(general, 6) is specified to produce exactly the %.6g digits; nan/inf keep printf's spelling.
static inline int fmt_g(char *tmp, double v)
{
    if (!(v - v == 0.0)) return snprintf(tmp, 40, "%g", v);
    auto r = to_chars(tmp, tmp + 39, v, chars_format::general, 6);
    return (int)(r.ptr - tmp);
}

// Trace files are megabytes of 6-digit numbers: deflate level 6 (the reference's Boost default) manages 12 MB/s per core on
// them, level 1 73 MB/s for files 11 % larger, Huffman coding alone (such text has next to no repeats for LZ77 to find) 110 MB/s for
// another 3 %.  Huffman-only unless MMSEQ_GZIP_LEVEL asks for a level (default strategy then); any gzip reader reads all of them.
// The Huffman-only blocks are written by host/huffenc.hpp (4.4 x zlib's rate for them: with zlib the trace writers needed more CPUs
// than a 16-CPU quota leaves next to the .M writer, and finished a second after the chain).
static int gzip_level()
{
    static const int level = [] { const char *e = getenv("MMSEQ_GZIP_LEVEL"); const int v = e ? atoi(e) : 1; return v >= 0 && v <= 9 ? v : 1; }();
    return level;
}
static int gzip_strategy()
{
    static const int strategy = getenv("MMSEQ_GZIP_LEVEL") ? Z_DEFAULT_STRATEGY : Z_HUFFMAN_ONLY;
    return strategy;
}

struct GzText {
    FILE *f = nullptr;
    uLong crc = 0;
    unsigned long long total = 0;
    string pending; // small writes are gathered here and compressed on flush
    explicit GzText(const string &path)
    {
        f = fopen(path.c_str(), "wb");
        if (!f) { cerr << "Error: cannot open " << path << " for writing.\n"; exit(1); }
        const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
        fwrite(hdr, 1, 10, f);
        crc = crc32(0L, Z_NULL, 0);
    }
    static string deflate_chunk(const string &in, bool last)
    {
        if (!last && gzip_strategy() == Z_HUFFMAN_ONLY) { // the default: host/huffenc.hpp, the same kind of block several times faster
            string out;
            if (huffenc::deflate_literals(in.data(), in.size(), out)) return out;
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, gzip_level(), Z_DEFLATED, -15, 8, gzip_strategy()) != Z_OK) {
            cerr << "Error initialising zlib.\n";
            exit(1);
        }
        string out;
        out.resize(deflateBound(&zs, (uLong)in.size()) + 16);
        zs.next_in = (Bytef *)in.data();
        zs.avail_in = (uInt)in.size();
        zs.next_out = (Bytef *)&out[0];
        zs.avail_out = (uInt)out.size();
        deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
        out.resize(out.size() - zs.avail_out);
        deflateEnd(&zs);
        return out;
    }
    // already deflated pieces (deflate_chunk of consecutive text), their crc32s and text sizes, appended in order
    void append_compressed(const vector<string> &comp, const vector<uLong> &crcs, const vector<size_t> &sizes)
    {
        for (size_t i = 0; i < comp.size(); ++i) {
            fwrite(comp[i].data(), 1, comp[i].size(), f);
            crc = crc32_combine(crc, crcs[i], (z_off_t)sizes[i]);
            total += sizes[i];
        }
    }
    void flush_pending()
    {
        if (pending.empty()) return;
        string c = deflate_chunk(pending, false);
        fwrite(c.data(), 1, c.size(), f);
        crc = crc32(crc, (const Bytef *)pending.data(), (uInt)pending.size());
        total += pending.size();
        pending.clear();
    }
    void str(const string &s) { pending += s; if (pending.size() > (1u << 24)) flush_pending(); }
    void num(double v)
    {
        char tmp[40];
        pending.append(tmp, fmt_g(tmp, v));
    }
    void close()
    {
        flush_pending();
        const string fin = deflate_chunk(string(), true); // final (empty) block
        fwrite(fin.data(), 1, fin.size(), f);
        unsigned char tr[8];
        const uint32_t c = (uint32_t)crc, n = (uint32_t)(total & 0xffffffffull);
        for (int i = 0; i < 4; ++i) { tr[i] = (unsigned char)(c >> (8 * i)); tr[4 + i] = (unsigned char)(n >> (8 * i)); }
        fwrite(tr, 1, 8, f);
        fclose(f);
        f = nullptr;
    }
};

// Rows of n numbers, each followed by a space, one line per sample (src/mmseq.cpp:912-916), from rows fetched on demand (the trace
// lives on the device, sample-major: a line of the file is a row there).  Three things overlap: the fetch of the next round of rows
// (device gather + copy), the formatting and compression of this round -- in parallel over pieces of <= 32 k columns of a line
// (~ 290 KB of text each, deflated independently and joined with sync flushes) -- and the write of the previous round's bytes.
static void write_trace_rows(GzText &gz, int n_lines, size_t n_cols, const function<void(int, int, double *)> &fetch,
                             const function<bool(size_t)> &keep, int threads = 0)
{
    if (threads < 1) threads = max(1, omp_get_max_threads());
    vector<char> mask(n_cols);
    size_t n_keep = 0;
    for (size_t c = 0; c < n_cols; ++c) n_keep += (mask[c] = keep(c) ? 1 : 0);
    if (n_keep == 0) { for (int i = 0; i < n_lines; ++i) gz.str("\n"); return; }
    const size_t cols_per_piece = 32768, pieces_per_line = (n_cols + cols_per_piece - 1) / cols_per_piece;
    const size_t want_pieces = (size_t)threads * 2;
    size_t lines_per_round = max<size_t>((size_t)(64u << 20) / (n_cols * 8), (want_pieces + pieces_per_line - 1) / pieces_per_line);
    lines_per_round = max<size_t>(1, min<size_t>({lines_per_round, (size_t)(512u << 20) / (n_cols * 8) + 1, (size_t)n_lines}));
    vector<double> buf[2];
    buf[0].resize(lines_per_round * n_cols);
    buf[1].resize(lines_per_round * n_cols);
    gz.flush_pending();
    fetch(0, (int)min<size_t>(lines_per_round, (size_t)n_lines), buf[0].data());
    thread writer;
    vector<string> comp_prev; // owned by the writer thread while it runs
    for (int l0 = 0, r = 0; l0 < n_lines && !g_worker_failed.load(); l0 += (int)lines_per_round, ++r) {
        const int cnt = (int)min<size_t>(lines_per_round, (size_t)(n_lines - l0));
        const int next0 = l0 + cnt, next_cnt = (int)min<size_t>(lines_per_round, (size_t)max(0, n_lines - next0));
        thread fetcher;
        if (next_cnt > 0) fetcher = thread([&, next0, next_cnt, r]() { fetch(next0, next_cnt, buf[(r + 1) & 1].data()); });
        const double *rows = buf[r & 1].data();
        const int64_t n_pieces = (int64_t)cnt * (int64_t)pieces_per_line;
        vector<string> comp((size_t)n_pieces);
        vector<uLong> crcs((size_t)n_pieces);
        vector<size_t> sizes((size_t)n_pieces);
#pragma omp parallel num_threads(threads)
        {
            string text;
            char tmp[48];
#pragma omp for schedule(dynamic, 1)
            for (int64_t pc = 0; pc < n_pieces; ++pc) {
                const size_t line = (size_t)pc / pieces_per_line, piece = (size_t)pc % pieces_per_line;
                const size_t c0 = piece * cols_per_piece, c1 = min(n_cols, c0 + cols_per_piece);
                const double *row = rows + line * n_cols;
                text.clear();
                for (size_t c = c0; c < c1; ++c) {
                    if (!mask[c]) continue;
                    int len = fmt_g(tmp, row[c]);
                    tmp[len++] = ' ';
                    text.append(tmp, (size_t)len);
                }
                if (piece + 1 == pieces_per_line) text += "\n";
                comp[(size_t)pc] = GzText::deflate_chunk(text, false);
                crcs[(size_t)pc] = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)text.data(), (uInt)text.size());
                sizes[(size_t)pc] = text.size();
            }
        }
        if (writer.joinable()) writer.join();
        comp_prev.swap(comp);
        writer = thread([&gz, &comp_prev, crcs, sizes]() { gz.append_compressed(comp_prev, crcs, sizes); });
        if (fetcher.joinable()) fetcher.join();
    }
    if (writer.joinable()) writer.join();
}

// Stage timings on stderr when MMSEQ_TIMING is set (not part of the reference's output)
struct StageTimer {
    bool on = getenv("MMSEQ_TIMING") != nullptr;
    double t0 = omp_get_wtime(), last = t0;
    void mark(const char *what)
    {
        if (!on) return;
        const double now = omp_get_wtime();
        fprintf(stderr, "[timing] %-28s %8.3f s\n", what, now - last);
        last = now;
    }
    void total() { if (on) fprintf(stderr, "[timing] %-28s %8.3f s\n", "total", omp_get_wtime() - t0); }
};

// CPUs this process may actually use: a container's CFS quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us) is invisible to OpenMP, which
// then starts one thread per core of the host -- 256 threads throttled to 16 CPUs' worth of time on the GPU boxes here.
static int cpu_quota()
{
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64];
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (int)max<long long>(1, (quota + period - 1) / period);
}

int main(int argc, char **argv)
{
    StageTimer stage;
    if (!getenv("OMP_NUM_THREADS")) { // an explicit thread count is the user's (src/mmseq.cpp:323 prints it); otherwise respect the quota
        const int q = cpu_quota();
        if (q > 0 && q < omp_get_max_threads()) omp_set_num_threads(q);
    }
    const int max_threads = omp_get_max_threads();

    // DEFAULT PARAMETER VALUES (src/mmseq.cpp:183-205)
    double alpha = 0.1, beta = 0.1;
    int max_em_iter = 1000;
    double epsilon = 0.1;
    int gibbs_iter = 16384;
    const int trace_length = 1024;
    int gibbs_ss = gibbs_iter / trace_length;
    vector<double> percentiles = {5.0, 25.0, 50.0, 75.0, 95.0};
    int seed = 1234;
    bool debug = false;
    int device = 0, gpus = 1, chains = 1;

    bool em_one_device = false;
    vector<string> percentile_fields;
    const CliOption options[] = {
        {"-alpha", CliOption::REAL, &alpha},        {"-beta", CliOption::REAL, &beta},
        {"-max_em_iter", CliOption::INT, &max_em_iter}, {"-epsilon", CliOption::REAL, &epsilon},
        {"-gibbs_iter", CliOption::INT, &gibbs_iter},   {"-gibbs_ss", CliOption::INT, &gibbs_ss},
        {"-seed", CliOption::INT, &seed},           {"-device", CliOption::INT, &device},
        {"-gpus", CliOption::INT, &gpus},           {"-chains", CliOption::INT, &chains},
        {"-percentiles", CliOption::LIST, &percentile_fields},
        {"-debug", CliOption::FLAG, &debug},        {"-em_one_device", CliOption::FLAG, &em_one_device},
        {"-h", CliOption::HELP, nullptr},           {"-help", CliOption::HELP, nullptr},       {"--help", CliOption::HELP, nullptr},
        {"-v", CliOption::VERSION, nullptr},        {"-version", CliOption::VERSION, nullptr}, {"--version", CliOption::VERSION, nullptr},
    };
    auto usage_error = [&](const string &msg) { cerr << msg << "\n"; printUsage(cerr); exit(1); };
    int pos = 1;                                   // next word of the command line
    for (;;) {
        const CliOption *opt = nullptr;
        if (pos < argc) for (const CliOption &o : options) if (strcmp(argv[pos], o.name) == 0) opt = &o;
        if (!opt) {                                // not an option: exactly the two positional arguments must be left
            if (argc - pos == 2) break;
            if (pos < argc && argv[pos][0] == '-') usage_error(string("Error: unrecognised option ") + argv[pos] + ".");
            usage_error("Error: mandatory arguments missing.");
        }
        if (opt->kind == CliOption::HELP) { cerr << "Calculate mmseq expression estimates.\n"; printUsage(cerr); exit(1); } // exit code 1, as src/mmseq.cpp:256-259
        if (opt->kind == CliOption::VERSION) { cerr << "mmseq-" << MMSEQ_VERSION << endl; exit(1); }
        if (opt->kind == CliOption::FLAG) { *(bool *)opt->target = true; pos += 1; continue; }
        if (pos + 1 >= argc) usage_error("Error: mandatory arguments missing.");
        const char *value = argv[pos + 1];
        if (opt->kind == CliOption::REAL) *(double *)opt->target = strtod(value, NULL);
        else if (opt->kind == CliOption::INT) *(int *)opt->target = atoi(value);
        else *(vector<string> *)opt->target = split_fields(value, ',');
        pos += 2;
    }
    if (!percentile_fields.empty()) {
        percentiles.resize(percentile_fields.size());
        for (size_t i = 0; i < percentile_fields.size(); i++) {
            const double v = strtod(percentile_fields[i].c_str(), NULL);
            if (!(v >= 0 && v <= 100)) { cerr << "Percentiles must be in (0,100)\n"; exit(1); }
            percentiles[i] = v;
        }
    }
    const vector<string> arguments = {argv[pos], argv[pos + 1]};   // hits_file, output_base
    if (gibbs_ss == 0 || gibbs_iter % gibbs_ss != 0) { // :278 (gibbs_ss == 0 is a division by zero there)
        cerr << "Error: gibbs_iter must be divisible by gibbs_ss.\n";
        printUsage(cerr);
        exit(1);
    }
    gibbs_ss = gibbs_iter / trace_length; // :284 -- the user's -gibbs_ss is overwritten, as in the reference
    if (gibbs_iter <= 0 || trace_length <= 0) {
        cerr << "Error: no. of iteratons or trace length <= 0. Possible integer overflow - is gibbs_iter too high?\n";
        printUsage(cerr);
        exit(1);
    }
    if (gibbs_ss < 1 || gibbs_iter % trace_length != 0) { // the reference overruns its trace / divides by zero here (App. A)
        cerr << "Error: gibbs_iter must be a positive multiple of " << trace_length << ".\n";
        printUsage(cerr);
        exit(1);
    }
    if (gpus < 1 || chains < 1 || (gpus > 1 && chains > 1 && chains % gpus != 0)) {
        cerr << "Error: -gpus and -chains must be positive, and chains a multiple of gpus when both exceed 1.\n";
        printUsage(cerr);
        exit(1);
    }
    if (!is_power_of_two((unsigned)trace_length)) {
        cerr << "Error: gibbs_iter/gibbs_ss must be a power of 2.\n";
        printUsage(cerr);
        exit(1);
    }
    const string hits_file = arguments[0];
    const string output_base(arguments[1]);

    HitsfileReader hitsfileReader(hits_file);

    cout << "Running mmseq with parameters:\n"
         << "  alpha:         " << alpha << endl
         << "  beta:          " << beta << endl
         << "  max_em_iter:   " << max_em_iter << endl
         << "  epsilon:       " << epsilon << endl
         << "  gibbs_iter:    " << gibbs_iter << endl
         << "  gibbs_ss:      " << gibbs_ss << endl
         << "  seed[0]:       " << seed << endl
         << "  debug:         " << debug << endl
         << "  threads:       " << max_threads << endl
         << "  device:        " << device << " (HIP, libmmgibbs ABI " << mmg_abi_version() << ")" << endl
         << "  gpus:          " << gpus << endl
         << "  chains:        " << chains << endl;

    // The HIP runtime, the device context and the library's code object are brought up while the hits file is read (they are first
    // needed at the device problem build, where they used to cost about two seconds of an otherwise idle GPU): one tiny kernel
    // launch on a thread of its own.  Its result is ignored -- a device that cannot be used is reported by mmg_problem_create.
    std::thread device_warmup([device]() {
        const uint32_t ctr[4] = {0, 0, 0, 0}, key[2] = {0, 0};
        uint32_t out[6];
        (void)mmg_selftest_philox(device, ctr, key, out);
    });
    g_device_warmup = &device_warmup;
    struct WarmupJoiner { std::thread &t; ~WarmupJoiner() { if (t.joinable()) t.join(); g_device_warmup = nullptr; } } device_warmup_join{device_warmup};

    // ---- header (src/mmseq.cpp:332-379)
    map<string, double> sidLen;
    map<string, int> sidSeqLen;
    vector<string> transcriptList;
    map<string, vector<string>> gene2transcripts;
    vector<vector<string>> identical_transcripts;
    hitsfileReader.readHeader(&transcriptList, &sidLen, &sidSeqLen, &gene2transcripts, &identical_transcripts);

    map<string, string> transcript2gene;
    {
        vector<string> transcriptListGI;
        for (auto &g : gene2transcripts)
            for (auto &t : g.second) {
                if (transcript2gene.count(t) > 0) {
                    cerr << "Error: transcripts must be nested within genes in GeneIsoforms metadata.\n";
                    leave(1);
                }
                transcriptListGI.push_back(t);
                transcript2gene[t] = g.first;
            }
        vector<string> a = transcriptList, b = transcriptListGI;
        sort(a.begin(), a.end());
        sort(b.begin(), b.end());
        if ((size_t)(unique(a.begin(), a.end()) - a.begin()) != transcriptList.size()) {
            cerr << "Error: duplicate transcripts in @TranscriptMetaData entries.\n";
            leave(1);
        }
        if ((size_t)(unique(b.begin(), b.end()) - b.begin()) != transcriptListGI.size()) {
            cerr << "Error: duplicate transcripts in @GeneIsoforms entries.\n";
            leave(1);
        }
        for (auto &t : transcriptList)
            if (transcript2gene.count(t) == 0) {
                cerr << "Error: " << t << " does not belong to a gene in the @GeneIsoforms header entries.\n";
                leave(1);
            }
    }
    const size_t nHeader = transcriptList.size();

    // ---- READ LOOP (src/mmseq.cpp:395-441): transcript index = first-seen order, row = first-seen hit set
    vector<int32_t> hdr2obs(nHeader, -1); // header index -> observed index
    vector<uint32_t> obs2hdr;             // indexSid
    vector<int> doublehits;
    vector<uint32_t> k;                   // multiplicity per hit set
    vector<uint64_t> row_ptr(1, 0);       // hit sets in first-seen order
    vector<uint32_t> col_idx;
    long long numbermappedreads = 0;
    {
        // hit set -> row: open-addressing table keyed by a 64-bit hash of the sorted set; an entry is (upper hash half, row id), a
        // candidate whose tag matches is confirmed against the stored row itself, so there is no per-set key allocation
        // (src/mmseq.cpp:395-441 keeps a map<vector<int>,int> and regrows M).  The stage is bound by cache misses (one table line
        // per read, the stored row for a repeat): the slot of a read a dozen ahead is prefetched, and the table starts at the size
        // the file suggests (a record is >= 16 compressed bytes) instead of being rebuilt at every doubling.
        constexpr uint64_t EMPTY = ~0ull;
        size_t table_size = 1u << 16;
        {
            struct stat st_;
            const uint64_t fsz = stat(hits_file.c_str(), &st_) == 0 ? (uint64_t)st_.st_size : 0;
            while (table_size < fsz / 16 && table_size < (1ull << 31)) table_size <<= 1;
        }
        vector<uint64_t> table(table_size, EMPTY);
        vector<uint64_t> row_hash;
        {
            // The arrays of the hit sets grow to GIGABYTES at 50 M reads (4 GB of column indices): grown by doubling they are copied
            // 8 GB worth and fault in twice their final pages -- on the thread every other stage waits for.  Address space is
            // reserved from the file's size instead (a binary record of c hits is >= 16 compressed bytes and inflates about 2.5 x;
            // untouched pages cost nothing); a reservation the system refuses is simply not made.
            struct stat st_;
            const uint64_t fsz = stat(hits_file.c_str(), &st_) == 0 ? (uint64_t)st_.st_size : 0;
            try {
                col_idx.reserve((size_t)min<uint64_t>(fsz * 3 / 4, 3ull << 30));
                const size_t rows = (size_t)min<uint64_t>(fsz / 24, 1ull << 28);
                row_ptr.reserve(rows + 1); row_hash.reserve(rows); k.reserve(rows);
                // ... and is advised to come in huge pages: 5 GB first touched on this thread are 1.2 M page faults otherwise
                auto huge = [](void *p, size_t bytes) {
                    const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
                    if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
                };
                huge(col_idx.data(), col_idx.capacity() * 4); huge(row_ptr.data(), row_ptr.capacity() * 8);
                huge(row_hash.data(), row_hash.capacity() * 8); huge(k.data(), k.capacity() * 4);
            } catch (const std::bad_alloc &) {}
        }
        auto grow = [&]() {
            vector<uint64_t> bigger(table.size() * 2, EMPTY);
            const size_t mask = bigger.size() - 1;
            for (uint32_t r = 0; r < (uint32_t)row_hash.size(); ++r) {
                size_t s = (size_t)row_hash[r] & mask;
                while (bigger[s] != EMPTY) s = (s + 1) & mask;
                bigger[s] = (row_hash[r] & 0xffffffff00000000ull) | r;
            }
            table.swap(bigger);
        };
        // Four stages over a ring of blocks of reads, each stage its own thread(s): (1) the reader -- inflate (a thread of its own inside
        // hitsio) + record decode, strictly sequential (src/hitsio.hpp:77-79); (2) first-seen transcript numbering (:399-408),
        // sequential as well; (3) NSORT threads, alternate blocks: every read's hit set sorted and freed of repeats, its hash -- the
        // longest stage at 50 M reads; (4) this thread: the hit-set table.
        struct Block { vector<uint32_t> len, idx, dups; vector<uint64_t> hash; bool last = false; };
        // (round 5: with the file inflated by several threads and the records parsed in place, the two sorters of round 4 became the
        // pace of the pipeline -- reader and numberer both waited 4.7 s of a 6.3 s read: four, over a ring twice as deep)
        constexpr int NB = 16, NSORT = 4;
        Block blocks[NB];
        int state[NB] = {0}; // 0: free for the reader, 1: decoded, 2: numbered, 3: prepared for the table
        mutex mtx;
        condition_variable cv;
        atomic<uint32_t> n_seen{0}; // transcripts numbered so far (progress line only)
        double waited[4] = {0.0, 0.0, 0.0, 0.0}; // seconds a stage spent waiting for a block (MMSEQ_TIMING: which stage bounds the pipeline)
        auto wait_for = [&](int b, int want, int stage_id) {
            const double t0 = omp_get_wtime();
            { unique_lock<mutex> lk(mtx); cv.wait(lk, [&] { return state[b] == want; }); }
            if (stage_id >= 0) waited[stage_id] += omp_get_wtime() - t0;
        };
        auto set_state = [&](int b, int v) { { lock_guard<mutex> lk(mtx); state[b] = v; } cv.notify_all(); };
        thread producer([&]() {
            bool more = true;
            for (int b = 0; more; b = (b + 1) % NB) {
                wait_for(b, 0, 0);
                Block &B = blocks[b];
                B.len.clear(); B.idx.clear();
                more = hitsfileReader.readReadMapRecordsBulk(B.len, B.idx, 65536); // the read names are not used (:395-441)
                B.last = !more;
                set_state(b, 1);
            }
        });
        thread numberer([&]() {
            bool last = false;
            for (int b = 0; !last; b = (b + 1) % NB) {
                wait_for(b, 1, 1);
                Block &B = blocks[b];
                last = B.last;
                for (uint32_t &x : B.idx) {
                    const uint32_t hidx = x;
                    if (hidx >= nHeader) {
                        cerr << "Error: a read maps to a transcript that has no @TranscriptMetaData entry (no length).\n";
                        hits_die(); // (a pipeline thread: no static destructors under the other threads)
                    }
                    if (hdr2obs[hidx] < 0) {
                        hdr2obs[hidx] = (int32_t)obs2hdr.size(); obs2hdr.push_back(hidx);
                        n_seen.store((uint32_t)obs2hdr.size(), memory_order_relaxed);
                    }
                    x = (uint32_t)hdr2obs[hidx];
                }
                set_state(b, 2);
            }
        });
        atomic<int> last_block{-1}; // index of the block that ends the file, once known
        vector<thread> sorters;
        for (int w = 0; w < NSORT; ++w)
            sorters.emplace_back([&, w]() {
                for (int b = w;; b = (b + NSORT) % NB) {
                    { // this sorter's next block, or the end of the file in the other sorter's hands
                        unique_lock<mutex> lk(mtx);
                        cv.wait(lk, [&] { return state[b] == 2 || last_block.load() >= 0; });
                        if (state[b] != 2) return;
                    }
                    Block &B = blocks[b];
                    B.hash.resize(B.len.size());
                    B.dups.clear();
                    size_t at = 0, out = 0;
                    for (size_t r = 0; r < B.len.size(); ++r) {
                        uint32_t *c = B.idx.data() + out; // the set is written over the block's own indices (never ahead of the read position)
                        const uint32_t nin = B.len[r];
                        for (uint32_t q = 0; q < nin; ++q) c[q] = B.idx[at++];
                        sort(c, c + nin);
                        uint32_t nu = 0;
                        for (uint32_t q = 0; q < nin; ++q) {
                            if (nu && c[nu - 1] == c[q]) B.dups.push_back(c[q]); // a transcript listed twice for one read (:421-424)
                            else c[nu++] = c[q];
                        }
                        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)nu;
                        for (uint32_t q = 0; q < nu; ++q) { h ^= c[q]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
                        B.hash[r] = h;
                        B.len[r] = nu;
                        out += nu;
                    }
                    const bool was_last = B.last;
                    if (was_last) last_block.store(b);
                    set_state(b, 3);
                    if (was_last) return;
                }
            });
        vector<uint32_t> all_dups;
        bool last = false;
        for (int b = 0; !last; b = (b + 1) % NB) {
            wait_for(b, 3, 3);
            const Block &B = blocks[b];
            last = B.last;
            all_dups.insert(all_dups.end(), B.dups.begin(), B.dups.end());
            size_t at = 0;
            for (size_t r = 0; r < B.len.size(); ++r) {
                numbermappedreads++;
                const uint32_t *comb = B.idx.data() + at;
                const uint32_t nc = B.len[r];
                at += nc;
                const uint64_t h = B.hash[r];
                const size_t mask = table.size() - 1;
                if (r + 12 < B.len.size()) __builtin_prefetch(&table[(size_t)B.hash[r + 12] & mask]);
                size_t s = (size_t)h & mask;
                uint32_t row = 0xffffffffu;
                for (;; s = (s + 1) & mask) {
                    const uint64_t e = table[s];
                    if (e == EMPTY) break;
                    if ((e ^ h) >> 32) continue; // another set's tag
                    const uint32_t rr = (uint32_t)e;
                    if (row_hash[rr] == h && row_ptr[rr + 1] - row_ptr[rr] == nc && equal(comb, comb + nc, col_idx.begin() + (ptrdiff_t)row_ptr[rr])) { row = rr; break; }
                }
                if (row == 0xffffffffu) {
                    row = (uint32_t)k.size();
                    if ((row & 0xffff) == 0)
                        cout << "Found " << n_seen.load(memory_order_relaxed) << " transcripts in " << row << " transcript combinations.\r" << flush;
                    table[s] = (h & 0xffffffff00000000ull) | row;
                    row_hash.push_back(h);
                    k.push_back(0);
                    col_idx.insert(col_idx.end(), comb, comb + nc);
                    row_ptr.push_back(col_idx.size());
                    if ((uint64_t)k.size() * 2 > table.size()) grow();
                }
                k[row]++;
            }
            set_state(b, 0);
        }
        numberer.join();
        cv.notify_all();
        for (auto &t : sorters) t.join();
        doublehits.assign(obs2hdr.size(), 0);
        for (uint32_t o : all_dups) doublehits[o]++;
        if (stage.on) fprintf(stderr, "[timing] ingest stages waited: decode %.1f s, numbering %.1f s, table %.1f s (the sort + hash stage is the rest)\n", waited[0], waited[1], waited[3]);
        producer.join();
        cout << "Found " << obs2hdr.size() << " transcripts in " << k.size() << " transcript combinations." << endl;
    }
    const uint32_t n = (uint32_t)obs2hdr.size();
    const uint64_t m = k.size();
    if (n == 0 || m == 0) { cerr << "Error: no reads with transcript hits found in the hits file.\n"; leave(1); }
    auto sid = [&](uint32_t t) -> const string & { return transcriptList[obs2hdr[t]]; };
    auto obs_of = [&](const string &name) -> int32_t { // sidIndex lookup by name
        static map<string, int32_t> cache;
        if (cache.empty()) for (uint32_t t = 0; t < n; ++t) cache[sid(t)] = (int32_t)t;
        auto it = cache.find(name);
        return it == cache.end() ? -1 : it->second;
    };

    stage.mark("read hits file + collapse");
    // ---- l[t] (src/mmseq.cpp:593-608)
    vector<double> l(n);
    for (uint32_t t = 0; t < n; t++) {
        if (sidLen.count(sid(t)) == 0) { cerr << "Error: transcript '" << sid(t) << "' has no length.\n"; leave(1); }
        l[t] = (double)sidLen[sid(t)] * (double)numbermappedreads / 1000000000.0;
        if (l[t] <= 0) { cerr << "Error: transcript '" << sid(t) << "' has a length of zero.\n"; leave(1); }
    }

    // ---- start values and unique hits (src/mmseq.cpp:610-638) come from the device once the problem is there
    //      (mmg_problem_start_values: the shares k_i / |row i| summed EXACTLY in fixed point -- the reference adds them in floating point
    //      in the order it read the file, so its start value depends on that order in the last bits; this one is a function of the hit
    //      sets).  The 100-bin histogram of shared counts is only ever written by -debug (.sharedcounts): a host pass, then.
    vector<vector<int>> counts_shared;
    vector<double> mu(n, 0.0);
    vector<int32_t> unique_hits(n, 0);
    if (debug) {
        counts_shared.assign(n, vector<int>(100, 0));
#pragma omp parallel num_threads(max(1, min(8, omp_get_max_threads() / 2))) // every thread reads the whole hit list: more only adds traffic
        {
            const uint64_t nth = (uint64_t)omp_get_num_threads(), tid = (uint64_t)omp_get_thread_num();
            const uint32_t lo = (uint32_t)((uint64_t)n * tid / nth), hi = (uint32_t)((uint64_t)n * (tid + 1) / nth);
            if (lo < hi)
                for (uint64_t i = 0; i < m; ++i) {
                    const uint64_t b = row_ptr[i], e = row_ptr[i + 1];
                    const int L = (int)(e - b);
                    for (uint64_t j = b; j < e; ++j) {
                        const uint32_t c = col_idx[j];
                        if (c - lo < hi - lo) counts_shared[c][min(L, 100) - 1] += (int)k[i];
                    }
                }
        }
    }

    stage.mark("l");
    // ---- unique hits to identical sets and genes: O(nnz) form of src/uh.cpp:3-26
    vector<int> identical_unique_hits(identical_transcripts.size(), 0), gene_unique_hits(gene2transcripts.size(), 0);
    {
        cerr << "Counting unique hits to sets of identical transcripts...";
        vector<vector<uint32_t>> t2sets(n);
        for (size_t v = 0; v < identical_transcripts.size(); ++v)
            for (auto &name : identical_transcripts[v]) {
                const int32_t t = obs_of(name);
                if (t >= 0) t2sets[t].push_back((uint32_t)v);
            }
        // rows in parallel, integer counts per thread summed at the end (the sums do not depend on the split)
        const int uh_threads = max(1, omp_get_max_threads());
        int64_t empty_rows_k = 0;
        {
            vector<vector<int>> part((size_t)uh_threads, vector<int>(identical_unique_hits.size(), 0));
#pragma omp parallel num_threads(uh_threads) reduction(+ : empty_rows_k)
            {
                vector<int> &mine = part[(size_t)omp_get_thread_num()];
                vector<uint32_t> cand, tmp;
#pragma omp for schedule(static)
                for (int64_t i = 0; i < (int64_t)m; ++i) {
                    const uint64_t b = row_ptr[i], e = row_ptr[i + 1];
                    if (b == e) { empty_rows_k += (int64_t)k[i]; continue; } // an empty row counts for every group
                    if (t2sets[col_idx[b]].empty()) continue;
                    cand = t2sets[col_idx[b]];
                    for (uint64_t j = b + 1; j < e && !cand.empty(); ++j) {
                        tmp.clear();
                        for (uint32_t s : cand)
                            if (find(t2sets[col_idx[j]].begin(), t2sets[col_idx[j]].end(), s) != t2sets[col_idx[j]].end()) tmp.push_back(s);
                        cand.swap(tmp);
                    }
                    sort(cand.begin(), cand.end());
                    cand.erase(unique(cand.begin(), cand.end()), cand.end());
                    for (uint32_t s : cand) mine[s] += (int)k[i];
                }
            }
            for (auto &pt : part) for (size_t v = 0; v < pt.size(); ++v) identical_unique_hits[v] += pt[v];
            for (auto &x : identical_unique_hits) x += (int)empty_rows_k;
        }
        cerr << "done." << endl;
        cerr << "Counting unique hits to genes...";
        map<string, int> gene2index;
        { int g = 0; for (auto &gt : gene2transcripts) gene2index[gt.first] = g++; }
        vector<int> t2g(n);
        for (uint32_t t = 0; t < n; ++t) t2g[t] = gene2index[transcript2gene[sid(t)]];
        {
            vector<vector<int>> part((size_t)uh_threads, vector<int>(gene_unique_hits.size(), 0));
#pragma omp parallel num_threads(uh_threads)
            {
                vector<int> &mine = part[(size_t)omp_get_thread_num()];
#pragma omp for schedule(static)
                for (int64_t i = 0; i < (int64_t)m; ++i) {
                    const uint64_t b = row_ptr[i], e = row_ptr[i + 1];
                    if (b == e) continue;
                    const int g = t2g[col_idx[b]];
                    bool uniq = true;
                    for (uint64_t j = b + 1; j < e; ++j) if (t2g[col_idx[j]] != g) { uniq = false; break; }
                    if (uniq) mine[g] += (int)k[i];
                }
            }
            for (auto &pt : part) for (size_t g = 0; g < pt.size(); ++g) gene_unique_hits[g] += pt[g];
            for (auto &x : gene_unique_hits) x += (int)empty_rows_k;
        }
        cerr << "done." << endl;
    }

    stage.mark("unique hits (sets, genes)");
    // ---- .k and .M (src/mmseq.cpp:682-695): 14 GB of text at 50 M reads, written by a thread of its own while the device builds the
    //      problem and runs EM and Gibbs (the arrays it reads are not touched again; joined before the run ends).  Its formatting
    //      threads leave three CPUs of the container's quota alone: with all of them busy the quota runs out and the main thread is
    //      throttled with them -- the upload of the matrix (4 GB of pageable memory through the runtime's staging copies) took 2.8 s
    //      instead of 0.3 s, and an EM sweep (a kernel and a read-back) 8 ms instead of 1.3.
    ofstream ofs;
    const int km_threads = max(1, omp_get_max_threads() - 3);
    std::thread km_writer([&, m, n, km_threads]() {   // integer tables: chunks of rows formatted in parallel (to_chars), written in order
        ofstream ofs;
        auto write_rows = [&](ofstream &o, const function<void(uint64_t, string &)> &fmt) {
            const uint64_t chunk = 1u << 16;
            const int64_t nchunks = (int64_t)((m + chunk - 1) / chunk);
            const int64_t batch = max<int64_t>(1, km_threads * 2);
            for (int64_t c0 = 0; c0 < nchunks; c0 += batch) {
                const int64_t nb = min(batch, nchunks - c0);
                vector<string> out(nb);
#pragma omp parallel for schedule(dynamic, 1) num_threads(km_threads)
                for (int64_t c = 0; c < nb; ++c) {
                    const uint64_t r0 = (uint64_t)(c0 + c) * chunk, r1 = min<uint64_t>(m, r0 + chunk);
                    for (uint64_t i = r0; i < r1; ++i) fmt(i, out[c]);
                }
                for (auto &x : out) o.write(x.data(), (streamsize)x.size());
            }
        };
        auto put = [](string &o, uint64_t v, char sep) {
            char tmp[24];
            auto r = to_chars(tmp, tmp + sizeof tmp, v);
            o.append(tmp, r.ptr - tmp);
            o.push_back(sep);
        };
        ofs.open((output_base + ".k").c_str());
        write_rows(ofs, [&](uint64_t i, string &o) { put(o, k[i], '\n'); });
        ofs.close(); ofs.clear();
        ofs.open((output_base + ".M").c_str());
        ofs << "#";
        for (uint32_t t = 0; t < n; t++) ofs << "\t" << sid(t);
        ofs << "\n";
        // one line "row<TAB>column" per hit (:690-694): 13.7 GB of text at 50 M reads.  The row's digits are formatted once per row,
        // the lines of a chunk go into one buffer sized beforehand (a row index and a column index have at most 10 digits each)
        write_rows(ofs, [&](uint64_t i, string &o) {
            const uint64_t b = row_ptr[i], e = row_ptr[i + 1];
            if (b == e) return;
            char head[24];
            const size_t hl = (size_t)(to_chars(head, head + 22, i).ptr - head);
            head[hl] = '\t';
            const size_t at = o.size();
            o.resize(at + (e - b) * (hl + 1 + 11));
            char *w = &o[at];
            for (uint64_t j = b; j < e; ++j) {
                memcpy(w, head, hl + 1);
                w = to_chars(w + hl + 1, w + hl + 12, col_idx[j]).ptr;
                *w++ = '\n';
            }
            o.resize((size_t)(w - o.data()));
        });
        ofs.close(); ofs.clear();
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } km_join{km_writer};
    g_background_writer = &km_writer;

    if (debug) { // src/mmseq.cpp:697-731
        ofs.open((output_base + ".sharedcounts").c_str());
        for (auto &name : transcriptList) {
            ofs << name << "\t";
            const int32_t t = obs_of(name);
            for (int i = 0; i < 100; i++) ofs << (t >= 0 ? counts_shared[t][i] : 0) << "\t";
            ofs << endl;
        }
        ofs.close(); ofs.clear();
        ofs.open((output_base + ".doublehits").c_str());
        for (uint32_t i = 0; i < n; i++) ofs << doublehits[i] << endl;
        ofs.close(); ofs.clear();
        // transposed matrix without consecutive duplicate rows, and the ids of the duplicates
        vector<vector<uint32_t>> Mt(n);
        for (uint64_t i = 0; i < m; ++i)
            for (uint64_t j = row_ptr[i]; j < row_ptr[i + 1]; ++j) Mt[col_idx[j]].push_back((uint32_t)i);
        ofs.open((output_base + ".Mt-nodups").c_str());
        ofstream ofs2((output_base + ".dupIDs").c_str());
        for (uint32_t t = 0; t < n; ++t) {
            if (t > 0 && Mt[t] == Mt[t - 1]) ofs2 << sid(t) << endl;
            else for (uint32_t r : Mt[t]) ofs << t << "\t" << r << endl;
        }
        ofs.close(); ofs.clear();
    }

    stage.mark("start the .k .M writer");
    // ---- device problem.  Rows go up in first-seen order and observed-transcript numbering, exactly as src/mmseq.cpp:399-418 builds
    //      them; the library stores the rows in its own canonical order and -- given tx_order -- numbers the transcripts gene by gene
    //      (header order, the isoforms of a gene adjacent: the sample kernel keeps a window of consecutive transcripts in LDS and
    //      wants a read's hits close together).  Both orders are irrelevant to the model; every array that comes back is in
    //      observed-transcript numbering.
    mmg_problem *prob = nullptr;
    {
        // key: (smallest header index of the transcript's gene, own header index)
        unordered_map<string, uint32_t> hdr_of_name;
        hdr_of_name.reserve(nHeader * 2);
        for (size_t i = 0; i < nHeader; ++i) hdr_of_name.emplace(transcriptList[i], (uint32_t)i);
        map<string, uint32_t> gene_first;
        for (auto &gt : gene2transcripts) {
            uint32_t f = 0xffffffffu;
            for (auto &name : gt.second) { auto it = hdr_of_name.find(name); if (it != hdr_of_name.end()) f = min(f, it->second); }
            gene_first[gt.first] = f;
        }
        vector<uint64_t> tx_order(n);
        for (uint32_t t = 0; t < n; ++t) {
            auto tg = transcript2gene.find(sid(t));
            const uint32_t gf = tg == transcript2gene.end() ? obs2hdr[t] : min(gene_first[tg->second], obs2hdr[t]);
            tx_order[t] = ((uint64_t)gf << 32) | obs2hdr[t];
        }
        mmg_problem_desc pd;
        memset(&pd, 0, sizeof pd);
        pd.m = m; pd.n = n; pd.row_ptr = row_ptr.data(); pd.col_idx = col_idx.data(); pd.k = k.data(); pd.l = l.data();
        pd.row_id_base = 0; pd.layout = MMG_LAYOUT_CANONICAL; pd.tx_order = tx_order.data();
        if (device_warmup.joinable()) device_warmup.join();
        MMG_TRY(mmg_problem_create(&pd, device, &prob));
        {
            mmg_problem_info inf0;
            MMG_TRY(mmg_problem_info_get(prob, &inf0));
            if (inf0.tx_renumbered & MMG_ORDER_SKIPPED)
                cerr << "Warning: not enough free device memory to try a gene order derived from the hit graph; the run uses the hits file's gene order "
                        "(slower on reads that hit paralogues, and the traces differ from a run that had the memory)" << endl;
        }
        if (stage.on) {
            mmg_problem_info inf;
            MMG_TRY(mmg_problem_info_get(prob, &inf));
            fprintf(stderr, "[timing] sample kernel %d (2 sliced-ELL stream, 0 CSR tiles), %llu of %llu tiles on the register path, %llu with far lists, %.1f MB on the device%s\n",
                    inf.sample_kernel, (unsigned long long)inf.fast_tiles, (unsigned long long)inf.n_tiles, (unsigned long long)inf.far_tiles, inf.device_bytes / 1e6,
                    (inf.tx_renumbered & 0xff) == 3 ? "; the genes reordered by the gene-level hit graph (reads that also hit paralogues)" : "");
        }
    }

    stage.mark("device problem build");
    MMG_TRY(mmg_problem_start_values(prob, mu.data(), unique_hits.data()));
    stage.mark("start values, unique hits (device)");
    // ---- several devices: the stored problem (canonical order, device numbering) is cut into contiguous read shards, one per device
    //      (one chain: EM and Gibbs both run sharded), or replicated (chains >= devices).  Cut on device 0 and copied device to
    //      device (mmg_problem_shard): nothing comes back to the host.
    vector<mmg_problem *> dprob;   // problems created here (besides prob)
    vector<mmg_problem *> part;    // problem of device i
    mmg_group *grp = nullptr;
    const bool shard = gpus > 1 && chains == 1;
    if (gpus > 1) {
        vector<int> devs(gpus);
        for (int i = 0; i < gpus; ++i) devs[i] = device + i;
        MMG_TRY(mmg_group_create(devs.data(), gpus, &grp));
        mmg_problem_info inf;
        MMG_TRY(mmg_problem_info_get(prob, &inf));
        vector<uint64_t> bounds(gpus + 1, 0);
        // cut by measured cost: every candidate shard is timed on device 0 with the start values as weights (0.1 s), so that the devices
        // finish their sweeps together whatever the mix of near rows, far rows and multiplicities (src/mmseq.cpp:864 splits the rows evenly)
        if (shard) MMG_TRY(mmg_problem_shard_bounds_timed(prob, mu.data(), gpus, bounds.data()));
        part.resize(gpus);
        for (int i = 0; i < gpus; ++i) {
            if (!shard && i == 0) { part[0] = prob; continue; }
            MMG_TRY(mmg_problem_shard(prob, shard ? bounds[i] : 0, shard ? bounds[i + 1] : inf.m, devs[i], &part[i]));
            dprob.push_back(part[i]);
        }
        if (stage.on && shard) {
            fprintf(stderr, "[timing] read shards (rows):");
            for (int i = 0; i < gpus; ++i) fprintf(stderr, " %llu", (unsigned long long)(bounds[i + 1] - bounds[i]));
            fprintf(stderr, "\n");
        }
        stage.mark(shard ? "read shards" : "replicas");
    }
    const bool shard_em = shard && !em_one_device;
    // the whole problem is needed on device 0 only while something runs on it: with sharded EM and Gibbs, not beyond this point
    if (shard && shard_em) { mmg_problem_destroy(prob); prob = nullptr; }
    // ---- EM on the device(s) (src/mmseq.cpp:741-811): mu stays there; this loop owns the stopping rule and the output
    GzText *gz_em = debug ? new GzText(output_base + ".trace_em.gz") : nullptr;
    if (gz_em) { for (uint32_t t = 0; t < n; t++) { gz_em->str(sid(t)); gz_em->str(" "); } gz_em->str("\n"); }
    {
        double loglik = 0.0;
        vector<mmg_em *> ems(shard_em ? gpus : 1, nullptr);
        if (shard_em) MMG_TRY(mmg_group_em_create(grp, part.data(), mu.data(), ems.data(), &loglik)); // exact integer sums: the bits of the unsharded EM
        else MMG_TRY(mmg_em_create(prob, mu.data(), &ems[0], &loglik));
        mmg_em *em = ems[0];
        stage.mark("EM set-up + first pass");
        double llr = epsilon + 1;
        int iter = 0;
        cout.precision(5);
        cout.setf(ios::fixed, ios::floatfield);
        while (iter < max_em_iter && llr > epsilon) {
            cout << "EM iteration " << iter << flush;
            if (gz_em) {
                if (iter) MMG_TRY(mmg_em_get_mu(em, mu.data()));
                for (uint32_t t = 0; t < n; t++) { gz_em->num(mu[t]); gz_em->str(" "); }
                gz_em->str("\n");
            }
            double ll = 0.0;
            MMG_TRY(mmg_em_step(em, &ll));
            llr = ll - loglik;
            loglik = ll;
            cout << ", log likelihood ratio: " << llr << "            \r";
            iter++;
        }
        MMG_TRY(mmg_em_get_mu(em, mu.data()));
        for (auto e : ems) mmg_em_destroy(e);
        if (shard && prob) { mmg_problem_destroy(prob); prob = nullptr; }   // (-em_one_device: the whole problem has served)
        cout << endl;
        cout.unsetf(ios::floatfield);
        cout.precision(6);
    }
    if (gz_em) { gz_em->close(); delete gz_em; }
    const vector<double> mu_em = mu;

    stage.mark("EM");
    // ---- Gibbs on the device(s) (src/mmseq.cpp:833-918); the trace stays there.
    //      one device: `chains` chains in one sampler.  several devices, one chain: the stored rows are cut into contiguous shards
    //      (mmg_shard_bounds), one per device, counts all-reduced over RCCL every iteration -- bit-identical to the one-device run.
    //      several devices, chains >= devices: the stored problem is replicated, every device runs chains / gpus chains.
    vector<mmg_sampler *> smps;
    mmg_sampler *smp = nullptr;
    mmg_summary *summ = nullptr;
    const size_t nI = identical_transcripts.size(), nG = gene2transcripts.size();
    const size_t nP = percentiles.size();
    vector<int> pind(nP);
    for (size_t i = 0; i < nP; i++) pind[i] = static_cast<int>(round(percentiles[i] / 100.0 * (trace_length - 1)));
    map<string, uint32_t> headerIndexOf;
    for (size_t i = 0; i < nHeader; ++i) headerIndexOf[transcriptList[i]] = (uint32_t)i;
    map<string, uint32_t> simuIndex; // isoform without hits -> its simulated ("virtual") trace
    {
        mmg_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.alpha = alpha; cfg.beta = beta; cfg.seed = (uint64_t)(int64_t)seed;
        cfg.n_chains = gpus > 1 ? max(1, chains / gpus) : chains; cfg.chain_base = 0; cfg.gibbs_iter = gibbs_iter; cfg.trace_len = trace_length;
        cfg.keep_trace = 1; cfg.timing = 0;
        if (gpus == 1) {
            smps.resize(1);
            MMG_TRY(mmg_sampler_create(prob, &cfg, mu_em.data(), &smps[0]));
        } else {
            smps.resize(gpus);
            for (int i = 0; i < gpus; ++i) {
                mmg_config ci = cfg;
                ci.chain_base = shard ? 0 : i * cfg.n_chains;
                MMG_TRY(mmg_sampler_create(part[i], &ci, mu_em.data(), &smps[i]));
            }
        }
        // ---- the posterior summary is set up BEFORE the loop and fed while it runs (src/mmseq.cpp:911-917 prints sample s inside the
        //      loop; :927-1108 derives the other traces after it): sample s is final after iteration s * gibbs_ss, so the four trace files
        //      are formatted, compressed and written by background threads while the device runs on -- rows come off the device on
        //      streams of their own (mmg_sampler_get_trace_rows_done, mmg_summary_get_rows), nothing waits for the chain.
        smp = smps[0]; // traces and per-feature summaries come from chain 0 (every shard holds the whole chain)
    }
    {
        vector<uint64_t> vid, iptr(1, 0), gptr(1, 0);
        vector<double> vscale;
        vector<uint32_t> imem, gmem;
        for (size_t v = 0; v < nI; ++v) {
            for (auto &name : identical_transcripts[v]) { const int32_t t = obs_of(name); if (t >= 0) imem.push_back((uint32_t)t); }
            iptr.push_back(imem.size());
        }
        size_t g = 0;
        for (auto &gt : gene2transcripts) {
            for (auto &name : gt.second) {
                const int32_t t = obs_of(name);
                if (t >= 0) { gmem.push_back((uint32_t)t); continue; }
                // no hits: simulate from the prior-only conditional (keyed by the header index, :971-978)
                simuIndex[name] = (uint32_t)vid.size();
                gmem.push_back(n + (uint32_t)vid.size());
                vid.push_back(headerIndexOf.count(name) ? headerIndexOf[name] : (uint64_t)nHeader + g);
                vscale.push_back(1.0 / (beta + sidLen[name] * (double)numbermappedreads / 1000000000.0));
            }
            gptr.push_back(gmem.size());
            g++;
        }
        mmg_summary_desc sd;
        memset(&sd, 0, sizeof sd);
        sd.chain = 0;
        sd.n_virtual = (uint32_t)vid.size(); sd.virtual_id = vid.data(); sd.virtual_scale = vscale.data();
        sd.n_identical = (uint32_t)nI; sd.identical_ptr = iptr.data(); sd.identical_member = imem.data();
        sd.n_genes = (uint32_t)nG; sd.gene_ptr = gptr.data(); sd.gene_member = gmem.data();
        sd.n_percentiles = (uint32_t)nP; sd.percentile_index = pind.data();
        MMG_TRY(mmg_summary_begin(smp, &sd, &summ));
    }
    const size_t nV = simuIndex.size();
    // (the writers outlive the loop: they are joined after the tables are written -- the tail of their work runs next to the summary
    // columns and the tables instead of in front of them)
    std::mutex ready_mu;
    std::condition_variable ready_cv;
    int samples_ready = 0;                           // samples whose rows may be fetched (trace and derived traces)
    auto wait_for = [&](int upto) { std::unique_lock<std::mutex> lk(ready_mu); ready_cv.wait(lk, [&] { return samples_ready >= upto || g_worker_failed.load(); }); };
    const int writer_threads = max(1, omp_get_max_threads());
    const int t_big = max(1, (writer_threads - 1) * 9 / 20), t_gene = max(1, writer_threads / 10);
    std::thread w_trace, w_ident, w_gene, w_prop;
    {
        w_trace = std::thread([&]() {
            GzText gz(output_base + ".trace_gibbs.gz");
            for (uint32_t t = 0; t < n; t++) { gz.str(sid(t)); gz.str(" "); }
            gz.str("\n");
            write_trace_rows(gz, trace_length, n, [&](int first, int count, double *out) { wait_for(first + count); if (g_worker_failed.load()) return; MMG_TRY_WORKER(mmg_sampler_get_trace_rows_done(smp, 0, first, count, out)); },
                             [](size_t) { return true; }, t_big);
            gz.close();
        });
        w_ident = std::thread([&]() {
            // a set whose first summed sample has no finite logarithm is left out of its trace file (:1040)
            wait_for(1);
            if (g_worker_failed.load()) return;
            vector<double> firstI(max<size_t>(nI, 1));
            MMG_TRY_WORKER(mmg_summary_get_rows(summ, MMG_SERIES_IDENTICAL, 0, 1, firstI.data()));
            vector<char> keepI(nI);
            for (size_t v = 0; v < nI; ++v) keepI[v] = isfinite(log(firstI[v])) != 0;
            GzText gi(output_base + ".identical.trace_gibbs.gz");
            for (size_t v = 0; v < nI; ++v)
                if (keepI[v]) {
                    for (size_t j = 0; j < identical_transcripts[v].size(); ++j) {
                        gi.str(identical_transcripts[v][j]);
                        if (identical_transcripts[v][j].compare(identical_transcripts[v].back()) != 0) gi.str("+");
                    }
                    gi.str(" ");
                }
            gi.str("\n");
            write_trace_rows(gi, trace_length, nI, [&](int first, int count, double *out) { wait_for(first + count); if (g_worker_failed.load()) return; MMG_TRY_WORKER(mmg_summary_get_rows(summ, MMG_SERIES_IDENTICAL, first, count, out)); },
                             [&](size_t v) { return keepI[v] != 0; }, 1);
            gi.close();
        });
        w_gene = std::thread([&]() {
            wait_for(1);                             // (:1068: the same rule for genes)
            if (g_worker_failed.load()) return;
            vector<double> firstG(max<size_t>(nG, 1));
            MMG_TRY_WORKER(mmg_summary_get_rows(summ, MMG_SERIES_GENE, 0, 1, firstG.data()));
            vector<char> keepG(nG);
            for (size_t g = 0; g < nG; ++g) keepG[g] = isfinite(log(firstG[g])) != 0;
            GzText gg(output_base + ".gene.trace_gibbs.gz");
            { size_t g = 0; for (auto &gt : gene2transcripts) { if (keepG[g]) { gg.str(gt.first); gg.str(" "); } g++; } }
            gg.str("\n");
            write_trace_rows(gg, trace_length, nG, [&](int first, int count, double *out) { wait_for(first + count); if (g_worker_failed.load()) return; MMG_TRY_WORKER(mmg_summary_get_rows(summ, MMG_SERIES_GENE, first, count, out)); },
                             [&](size_t g) { return keepG[g] != 0; }, t_gene);
            gg.close();
        });
        w_prop = std::thread([&]() {
            GzText gp(output_base + ".prop.trace_gibbs.gz");
            for (uint32_t t = 0; t < n; t++) { gp.str(sid(t)); gp.str(" "); }
            gp.str("\n");
            write_trace_rows(gp, trace_length, n, [&](int first, int count, double *out) { wait_for(first + count); if (g_worker_failed.load()) return; MMG_TRY_WORKER(mmg_summary_get_rows(summ, MMG_SERIES_TRANSCRIPT, first, count, out)); },
                             [](size_t) { return true; }, t_big);
            gp.close();
        });
        // how the main thread stops the four writers when it (or one of them) fails: flag, wake, join
        g_stop_workers = [&]() {
            { std::lock_guard<std::mutex> lk(ready_mu); g_worker_failed.store(true); }
            ready_cv.notify_all();
            for (std::thread *w : {&w_trace, &w_ident, &w_gene, &w_prop}) if (w->joinable()) w->join();
        };
        auto check_workers = [&]() {
            if (!g_worker_failed.load()) return;
            std::string msg;
            { std::lock_guard<std::mutex> lk(g_worker_mu); msg = g_worker_msg; }
            main_thread_exit(msg);
        };
        // chunks of 1/64 of the run: the writers start on a chunk's samples when it ends, so what is left of their work after the last
        // iteration is 1/64 of the files
        const int chunk = max(1, gibbs_iter / 64);
        double t_enqueue = 0.0, t_sync = 0.0, t_advance = 0.0;
        auto enqueue = [&](int it) {
            const double c0 = omp_get_wtime();
            if (gpus == 1) MMG_TRY(mmg_sampler_run(smps[0], it));
            else if (gpus > 1 && chains == 1) MMG_TRY(mmg_group_run_sharded(grp, smps.data(), it));
            else MMG_TRY(mmg_group_run_chains(grp, smps.data(), it));
            t_enqueue += omp_get_wtime() - c0;
        };
        // one chunk is always enqueued ahead of the one waited for: the device does not idle while this thread hands samples on (or is
        // held up: the writers use every CPU of the quota)
        enqueue(min(chunk, gibbs_iter));
        for (int done = 0; done < gibbs_iter; done += chunk) {
            cout << "Gibbs iteration " << done << "       \r" << flush;
            check_workers();
            const int it = min(chunk, gibbs_iter - done);
            if (done + it < gibbs_iter) enqueue(min(chunk, gibbs_iter - done - it));
            // sample s is kept by iteration s * gibbs_ss (:911): the samples of the iterations up to done + it are final once the
            // iteration that stored the last of them is
            const int final_samples = min(trace_length, (done + it - 1) / gibbs_ss + 1);
            const double c1 = omp_get_wtime();
            for (auto sp : smps) MMG_TRY(mmg_sampler_wait_iterations(sp, done + it < gibbs_iter ? (final_samples - 1) * gibbs_ss + 1 : gibbs_iter));
            const double c2 = omp_get_wtime();
            MMG_TRY(mmg_summary_advance(summ, final_samples));
            { std::lock_guard<std::mutex> lk(ready_mu); samples_ready = final_samples; }
            ready_cv.notify_all();
            t_sync += c2 - c1; t_advance += omp_get_wtime() - c2;
        }
        for (auto sp : smps) MMG_TRY(mmg_sampler_sync(sp));
        check_workers();
        cout << "Gibbs iteration " << gibbs_iter - 1 << "       \r" << endl;
        if (stage.on) fprintf(stderr, "[timing] Gibbs loop: enqueue %.3f s, wait for the device %.3f s, derived rows %.3f s\n", t_enqueue, t_sync, t_advance);
        stage.mark("Gibbs (trace files written alongside)");
    }
    // moments of log mu pooled over all chains and devices (one fp64 all-reduce): log_mu, sd and mcse of multi-chain runs
    vector<double> pooled_sl, pooled_sl2;
    int64_t pooled_ns = 0;
    if (chains > 1) {
        pooled_sl.resize(n); pooled_sl2.resize(n);
        if (grp) MMG_TRY(mmg_group_pool_moments(grp, smps.data(), pooled_sl.data(), pooled_sl2.data(), &pooled_ns));
        else {
            vector<double> a_(n), b_(n);
            for (int c = 0; c < chains; ++c) {
                int64_t ns = 0;
                MMG_TRY(mmg_sampler_get_moments(smp, c, a_.data(), b_.data(), &ns));
                for (uint32_t t = 0; t < n; ++t) { pooled_sl[t] += a_[t]; pooled_sl2[t] += b_[t]; }
                pooled_ns += ns;
            }
        }
    }

    cout << "Amalgamating transcripts and calculating summary statistics..." << flush;
    // ---- posterior summary on the device (src/mmseq.cpp:927-1363): the derived traces were computed while the chain ran; what is left
    //      are the per-series columns -- percentiles, log means, Sokal -- of which only the columns come back.
    MMG_TRY(mmg_summary_finish(summ));
    stage.mark("device summary");
    // (the trace writer still reads rows of the sampler, the derived-trace writers rows of the summary: both are released once the
    // writers are done, behind the tables)
    auto release_device = [&]() {
        w_trace.join(); w_ident.join(); w_gene.join(); w_prop.join();
        // (the writers fetch the last 1/64 of the rows behind the loop's last check: a failure there -- a HIP error in a row fetch --
        // would leave a valid but truncated trace file; it ends the run like any other)
        if (g_worker_failed.load()) {
            std::string msg;
            { std::lock_guard<std::mutex> lk(g_worker_mu); msg = g_worker_msg; }
            g_stop_workers = nullptr; // (joined above)
            main_thread_exit(msg);
        }
        mmg_summary_destroy(summ);
        for (auto sp : smps) mmg_sampler_destroy(sp);
        for (auto pp : dprob) mmg_problem_destroy(pp);
        mmg_problem_destroy(prob);
        if (grp) mmg_group_destroy(grp);
    };

    // ---- summary columns (src/mmseq.cpp:1110-1363)
    struct Series { vector<double> mean, sd, mcse, iact, pct; };
    auto fetch_series = [&](int kind, size_t count, Series &o) {
        o.mean.resize(max<size_t>(count, 1)); o.sd.resize(max<size_t>(count, 1)); o.mcse.resize(max<size_t>(count, 1));
        o.iact.resize(max<size_t>(count, 1)); o.pct.resize(max<size_t>(count * nP, 1));
        vector<double> var(max<size_t>(count, 1)), tau(max<size_t>(count, 1));
        vector<int32_t> rc(max<size_t>(count, 1));
        MMG_TRY(mmg_summary_get(summ, kind, o.mean.data(), var.data(), tau.data(), rc.data(), o.pct.data()));
        for (size_t t = 0; t < count; ++t) { // :1311-1324
            if (rc[t] != 0) { o.mcse[t] = trace_length; o.iact[t] = NAN; }
            else { o.mcse[t] = sqrt(tau[t] * var[t] / trace_length); o.iact[t] = tau[t]; }
            o.sd[t] = sqrt(var[t]);
        }
    };
    Series sT, sV, sI, sG;
    fetch_series(MMG_SERIES_TRANSCRIPT, n, sT);
    if (chains > 1 && pooled_ns > 1) { // all chains: mean and sd of log mu from the pooled moments, Monte Carlo error of the pooled mean
        for (uint32_t t = 0; t < n; ++t) {
            const double mean = pooled_sl[t] / (double)pooled_ns;
            const double var = (pooled_sl2[t] - (double)pooled_ns * mean * mean) / (double)(pooled_ns - 1);
            sT.mean[t] = mean;
            sT.sd[t] = sqrt(var > 0 ? var : 0.0);
            sT.mcse[t] = sT.mcse[t] / sqrt((double)chains);
        }
    }
    fetch_series(MMG_SERIES_VIRTUAL, nV, sV);
    fetch_series(MMG_SERIES_IDENTICAL, nI, sI);
    fetch_series(MMG_SERIES_GENE, nG, sG);
    const vector<double> &meanmu = sT.mean, &meanmu_identical = sI.mean, &meanmu_gene = sG.mean;
    const vector<double> &sd = sT.sd, &mumcse = sT.mcse, &iact = sT.iact;
    const vector<double> &sd_identical = sI.sd, &mumcse_identical = sI.mcse, &iact_identical = sI.iact;
    const vector<double> &sd_gene = sG.sd, &mumcse_gene = sG.mcse, &iact_gene = sG.iact;
    struct Props { vector<double> mean, probit_mean, probit_sd, pct; };
    auto fetch_props = [&](int kind, size_t count, Props &o) {
        o.mean.resize(max<size_t>(count, 1)); o.probit_mean.resize(max<size_t>(count, 1)); o.probit_sd.resize(max<size_t>(count, 1));
        o.pct.resize(max<size_t>(count * nP, 1));
        MMG_TRY(mmg_summary_get_proportions(summ, kind, o.mean.data(), o.probit_mean.data(), o.probit_sd.data(), o.pct.data()));
    };
    Props pT, pV;
    fetch_props(MMG_SERIES_TRANSCRIPT, n, pT);
    fetch_props(MMG_SERIES_VIRTUAL, nV, pV);
    const vector<double> &meanprop = pT.mean, &meanprobitprop = pT.probit_mean, &sdprobitprop = pT.probit_sd;
    auto pct_row = [&](const vector<double> &pct, size_t i) { return vector<double>(pct.begin() + (ptrdiff_t)(i * nP), pct.begin() + (ptrdiff_t)((i + 1) * nP)); };

    const double digalpha = mmnum::digamma(alpha);                 // gsl_sf_psi(alpha)        :1372
    const double sqrtpolygalpha = sqrt(mmnum::trigamma(alpha));    // sqrt(gsl_sf_psi_n(1,.))  :1373
    // (lookups that never insert: the table writers below run in several threads; a name the header did not describe reads as 0, which
    // is what the maps' operator[] would have inserted)
    auto len_of = [&](const string &name) { auto it = sidLen.find(name); return it == sidLen.end() ? 0.0 : it->second; };
    auto seqlen_of = [&](const string &name) { auto it = sidSeqLen.find(name); return it == sidSeqLen.end() ? 0 : it->second; };
    auto gene_size_of = [&](const string &name) {
        auto tg = transcript2gene.find(name);
        auto it = gene2transcripts.find(tg == transcript2gene.end() ? string() : tg->second);
        return it == gene2transcripts.end() ? (size_t)0 : it->second.size();
    };
    auto simu_of = [&](const string &name) { auto it = simuIndex.find(name); return it == simuIndex.end() ? 0u : it->second; };
    auto prior_logmu = [&](const string &name) { return digalpha - log(beta + len_of(name) * (double)numbermappedreads / 1000000000.0); };

    // ---- gene-level expression-weighted effective length (src/mmseq.cpp:1375-1395)
    vector<double> gene_lengths(nG, 0.0);
    {
        size_t g = 0;
        for (auto &gt : gene2transcripts) {
            if (isfinite(meanmu_gene[g]) != 0) {
                double sum = 0;
                for (auto &name : gt.second) {
                    const int32_t t = obs_of(name);
                    const double e = t >= 0 ? exp(meanmu[t]) : exp(prior_logmu(name));
                    gene_lengths[g] += len_of(name) * e;
                    sum += e;
                }
                gene_lengths[g] /= sum;
            }
            g++;
        }
    }

    auto join_pct = [&](ostream &o, const vector<double> &v, const char *term) {
        for (size_t i = 0; i < nP; i++) { o << v[i]; o << (i == nP - 1 ? term : ","); }
    };
    auto pct_header = [&](ostream &o, const char *label, const char *term) {
        o << label;
        for (size_t i = 0; i < nP; i++) { o << percentiles[i]; o << (i == nP - 1 ? term : ","); }
    };

    // ---- .gene.mmseq (src/mmseq.cpp:1615-1669)
    auto write_gene_table = [&]() {
    ofstream ofs((output_base + ".gene.mmseq").c_str());
    ofs << "# Mapped fragments: " << numbermappedreads << endl;
    ofs << "feature_id\tlog_mu\tsd\tmcse\tiact\teffective_length\ttrue_length\tunique_hits\tntranscripts\tobserved\t";
    pct_header(ofs, "percentiles", "\n");
    {
        size_t g = 0;
        for (auto &gt : gene2transcripts) {
            bool obs = false;
            for (auto &name : gt.second) if (obs_of(name) >= 0) { obs = true; break; }
            if (obs) {
                ofs << gt.first << "\t" << meanmu_gene[g] << "\t" << sd_gene[g] << "\t" << mumcse_gene[g] << "\t" << iact_gene[g] << "\t"
                    << gene_lengths[g] << "\t"
                    << "NA"
                    << "\t" << gene_unique_hits[g] << "\t" << gt.second.size() << "\t"
                    << "1"
                    << "\t";
            } else {
                ofs << gt.first << "\t" << meanmu_gene[g] << "\t" << sd_gene[g] << "\t" << sd_gene[g] / sqrt(trace_length) << "\t" << 1 << "\t"
                    << gene_lengths[g] << "\t"
                    << "NA"
                    << "\t"
                    << "0"
                    << "\t" << gt.second.size() << "\t"
                    << "0"
                    << "\t";
            }
            join_pct(ofs, pct_row(sG.pct, g), "\n");
            g++;
        }
    }
    ofs.close();
    };

    stage.mark("summary columns");
    // ---- .mmseq (src/mmseq.cpp:1469-1554)
    ofs.open((output_base + ".mmseq").c_str());
    ofs << "# Mapped fragments: " << numbermappedreads << endl;
    ofs << "feature_id\tlog_mu\tsd\tmcse\tiact\teffective_length\ttrue_length\tunique_hits\tmean_proportion\tmean_probit_proportion\tsd_"
           "probit_proportion\tlog_mu_em\tobserved\tntranscripts\t";
    pct_header(ofs, "percentiles", "\t");
    pct_header(ofs, "percentiles_proportion", "\n");
    // (the rows are formatted in parallel, a slice of the list per thread into a stream of its own with the default formatting of
    // the file stream, and written in order; the gene table, which shares nothing with this one, is written by a thread of its own)
    (void)obs_of(string());   // the name table exists before threads read it
    auto mmseq_row = [&](ostream &o, const string &name) {
        const int32_t t = obs_of(name);
        if (t >= 0) {
            o << name << "\t" << meanmu[t] << "\t" << sd[t] << "\t" << mumcse[t] << "\t" << iact[t] << "\t" << len_of(name) << "\t"
                << seqlen_of(name) << "\t" << unique_hits[t] << "\t" << meanprop[t] << "\t" << meanprobitprop[t] << "\t"
                << sdprobitprop[t] << "\t" << log(mu_em[t]) << "\t"
                << "1"
                << "\t" << gene_size_of(name) << "\t";
            join_pct(o, pct_row(sT.pct, t), "\t");
            join_pct(o, pct_row(pT.pct, t), "\n");
        } else {
            o << name << "\t" << prior_logmu(name) << "\t" << sqrtpolygalpha << "\t"
                << "0"
                << "\t"
                << "1"
                << "\t" << len_of(name) << "\t" << seqlen_of(name) << "\t" << 0 << "\t" << pV.mean[simu_of(name)] << "\t"
                << pV.probit_mean[simu_of(name)] << "\t" << pV.probit_sd[simu_of(name)] << "\t"
                << "NA"
                << "\t"
                << "0"
                << "\t" << gene_size_of(name) << "\t";
            join_pct(o, pct_row(sV.pct, simu_of(name)), "\t");
            join_pct(o, pct_row(pV.pct, simu_of(name)), "\n");
        }
        };
    std::thread gene_table([&]() { write_gene_table(); });
    {
        const int tt = max(1, min(omp_get_max_threads(), (int)(transcriptList.size() / 4096) + 1));
        vector<string> parts((size_t)tt);
#pragma omp parallel num_threads(tt)
        {
            const size_t me = (size_t)omp_get_thread_num(), nt = (size_t)omp_get_num_threads();
            const size_t lo = transcriptList.size() * me / nt, hi = transcriptList.size() * (me + 1) / nt;
            ostringstream o;
            for (size_t i = lo; i < hi; ++i) mmseq_row(o, transcriptList[i]);
            parts[me] = o.str();
        }
        for (auto &pt : parts) ofs.write(pt.data(), (streamsize)pt.size());
    }
    ofs.close(); ofs.clear();

    // ---- .identical.mmseq (src/mmseq.cpp:1556-1613)
    ofs.open((output_base + ".identical.mmseq").c_str());
    ofs << "# Mapped fragments: " << numbermappedreads << endl;
    ofs << "feature_id\tlog_mu\tsd\tmcse\tiact\teffective_length\ttrue_length\tunique_hits\tobserved\tntranscripts\t";
    pct_header(ofs, "percentiles", "\n");
    for (size_t v = 0; v < nI; ++v) {
        const vector<string> &set = identical_transcripts[v];
        const bool fin = isfinite(meanmu_identical[v]);
        for (auto &name : set) {
            ofs << name;
            if (name.compare(set.back()) != 0) ofs << "+";
            else if (fin)
                ofs << "\t" << meanmu_identical[v] << "\t" << sd_identical[v] << "\t" << mumcse_identical[v] << "\t" << iact_identical[v] << "\t"
                    << sidLen[set.front()] << "\t" << sidSeqLen[set.front()] << "\t" << identical_unique_hits[v] << "\t"
                    << "1"
                    << "\t" << set.size() << "\t";
            else
                ofs << "\t" << log((double)set.size()) + prior_logmu(name) << "\t" << sqrtpolygalpha << "\t"
                    << "0"
                    << "\t"
                    << "NA"
                    << "\t" << sidLen[set.front()] << "\t" << sidSeqLen[set.front()] << "\t" << 0 << "\t"
                    << "0"
                    << "\t" << set.size() << "\t";
        }
        if (fin) join_pct(ofs, pct_row(sI.pct, v), "\n");
        else for (size_t i = 0; i < nP; i++) ofs << "NA" << (i == nP - 1 ? "\n" : ",");
    }
    ofs.close(); ofs.clear();

    gene_table.join();

    cout << "done." << endl;
    cout << "Output files: " << endl
         << "  " << output_base << ".mmseq" << endl
         << "  " << output_base << ".identical.mmseq" << endl
         << "  " << output_base << ".gene.mmseq" << endl;
    cout << "  " << output_base << ".M" << endl << "  " << output_base << ".k" << endl << endl;
    cout << "  " << output_base << ".trace_gibbs.gz" << endl
         << "  " << output_base << ".identical.trace_gibbs.gz" << endl
         << "  " << output_base << ".gene.trace_gibbs.gz" << endl
         << "  " << output_base << ".prop.trace_gibbs.gz" << endl
         << endl;
    if (debug) {
        cout << endl
             << "  " << output_base << ".trace_em.gz" << endl
             << "  " << output_base << ".sharedcounts" << endl
             << "  " << output_base << ".Mt-nodups" << endl
             << "  " << output_base << ".doublehits" << endl
             << "  " << output_base << ".dupIDs" << endl;
    }
    stage.mark("write tables");
    release_device();
    stage.mark("trace files: the rest");
    if (km_writer.joinable()) km_writer.join();
    stage.mark("wait for the .k .M writer");
    stage.total();
    return 0;
}
