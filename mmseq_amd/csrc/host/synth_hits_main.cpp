// synth_hits -- writes the synthetic benchmark workload (mmg_problem_create_synthetic) as a hits FILE, so that the
// drop-in CLI can be run end to end at sizes no alignment in this image provides (a 50 M-read file).
//   synth_hits [-t] [-zipf CAP] [-genes G F] ROWS TRANSCRIPTS AVG_HITS OUT.hits [FAR_FRACTION [SEED [DEVICE]]]
// Reads come in generator order (a name-sorted BAM's order), transcripts in header order with genes of 1..7 consecutive isoforms;
// -genes G F: the generator's gene-block mode (mmg_synth_desc.gene_size / far_family): genes of G isoforms, a read's hits inside its
// gene, far hits to a gene of the read's paralogue family of F genes (0: anywhere) -- the header's genes are then those blocks;
// -zipf CAP: what a sequencing run looks like before mmseq collapses it (src/mmseq.cpp:409-440) -- every generated row is a hit SET that
// k reads share, k = min(CAP, floor((1 - u)^(-1 / 0.92))) (bench.py's `collapsed` workload: 47 % of the sets k = 1, 2.3 % above 64), and the
// file holds the reads of all sets in shuffled order (ROWS = 2 M, CAP = 10^6: 48 M reads);
// the rows are generated on the device, downloaded and written with HitsfileWriter (binary schema unless -t).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "hitsio.hpp"
#include "../../../include/mmgibbs.h"

static void die(const char *what) { std::fprintf(stderr, "synth_hits: %s: %s\n", what, mmg_last_error()); std::exit(1); }

int main(int argc, char **argv)
{
    bool text = false;
    int a = 1;
    if (a < argc && !std::strcmp(argv[a], "-t")) { text = true; ++a; }
    uint64_t zipf_cap = 0;
    if (a + 1 < argc && !std::strcmp(argv[a], "-zipf")) { zipf_cap = std::strtoull(argv[a + 1], nullptr, 10); a += 2; }
    uint32_t gene_size = 0, far_family = 0;
    if (a + 2 < argc && !std::strcmp(argv[a], "-genes")) { gene_size = (uint32_t)std::strtoul(argv[a + 1], nullptr, 10); far_family = (uint32_t)std::strtoul(argv[a + 2], nullptr, 10); a += 3; }
    if (argc - a < 4) {
        std::fprintf(stderr, "usage: synth_hits [-t] [-zipf CAP] [-genes G F] ROWS TRANSCRIPTS AVG_HITS OUT.hits [FAR_FRACTION [SEED [DEVICE]]]\n");
        return 1;
    }
    const uint64_t rows = std::strtoull(argv[a], nullptr, 10);
    const uint32_t n = (uint32_t)std::strtoul(argv[a + 1], nullptr, 10);
    const double avg = std::atof(argv[a + 2]);
    const char *path = argv[a + 3];
    const double far = argc - a > 4 ? std::atof(argv[a + 4]) : 0.0;
    const uint64_t seed = argc - a > 5 ? std::strtoull(argv[a + 5], nullptr, 10) : 1234;
    const int device = argc - a > 6 ? std::atoi(argv[a + 6]) : 0;

    mmg_synth_desc sd;
    std::memset(&sd, 0, sizeof sd);
    sd.seed = seed; sd.rows = rows; sd.row0 = 0; sd.n = n; sd.avg_hits = avg; sd.uniform = 0; sd.sorted = 0; sd.mapped_reads = rows;
    sd.far_fraction = far;
    sd.gene_size = gene_size; sd.far_family = far_family;
    mmg_problem *prob = nullptr;
    if (mmg_problem_create_synthetic(&sd, device, &prob)) die("mmg_problem_create_synthetic");
    mmg_problem_info inf;
    if (mmg_problem_info_get(prob, &inf)) die("mmg_problem_info_get");
    std::vector<uint64_t> rp(inf.m + 1);
    std::vector<uint32_t> ci(inf.nnz);
    std::vector<double> l(n);
    if (mmg_problem_download(prob, rp.data(), ci.data(), nullptr)) die("mmg_problem_download");
    if (mmg_problem_get_l(prob, l.data())) die("mmg_problem_get_l");
    mmg_problem_destroy(prob);

    // the reads of the file: one per row, or (-zipf) k per row in shuffled order
    std::vector<uint32_t> read_row;
    uint64_t n_reads = inf.m;
    if (zipf_cap) {
        uint64_t x = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        auto next = [&]() { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
        for (uint64_t r = 0; r < inf.m; ++r) {
            const double u = (double)(next() >> 11) * 0x1p-53;
            const double kf = std::floor(std::pow(1.0 - u, -1.0 / 0.92));
            const uint64_t k = kf >= (double)zipf_cap ? zipf_cap : (uint64_t)kf;
            read_row.insert(read_row.end(), (size_t)(k ? k : 1), (uint32_t)r);
        }
        n_reads = read_row.size();
        for (uint64_t i = n_reads - 1; i > 0; --i) { const uint64_t j = next() % (i + 1); std::swap(read_row[i], read_row[j]); }
    }

    FILE *out = std::fopen(path, "wb");
    if (!out) { std::fprintf(stderr, "synth_hits: cannot write %s\n", path); return 1; }
    {
        HitsfileWriter w(text ? "t" : "b", out);
        std::vector<std::string> name(n);
        char buf[32];
        for (uint32_t t = 0; t < n; ++t) {
            std::snprintf(buf, sizeof buf, "T%07u", t);
            name[t] = buf;
            const double efflen = l[t] * 1e9 / (double)rows; // l = efflen * N / 1e9 (src/mmseq.cpp:603; N = the sets with -zipf: a scale only)
            w.addTranscriptMetaData(name[t], efflen, (int)efflen + 180);
        }
        uint32_t g = 0;
        for (uint32_t t = 0; t < n; ++g) {
            std::snprintf(buf, sizeof buf, "G%06u", g);
            w.addGeneIsoformRecord(buf);
            const uint32_t size = gene_size ? gene_size : 1 + (uint32_t)(((uint64_t)g * 2654435761ull >> 7) % 7);
            for (uint32_t j = 0; j < size && t < n; ++j, ++t) w.addTranscriptToGeneIsoformRecord(name[t]);
        }
        w.writeHeader();
        for (uint64_t i = 0; i < n_reads; ++i) {
            const uint64_t r = zipf_cap ? read_row[i] : i;
            std::snprintf(buf, sizeof buf, "r%010llu", (unsigned long long)i);
            w.addReadMapRecord(buf);
            for (uint64_t j = rp[r]; j < rp[r + 1]; ++j) w.addTranscriptIndexToReadMapRecord(ci[j]);
            w.writeReadMapRecord();
        }
        w.close();
    }
    std::fclose(out);
    std::fprintf(stderr, "synth_hits: %llu reads in %llu hit sets, %llu hits per set list, %u transcripts -> %s\n", (unsigned long long)n_reads, (unsigned long long)inf.m, (unsigned long long)inf.nnz, n, path);
    return 0;
}
