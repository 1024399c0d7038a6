// hitsio.cpp -- see hitsio.hpp.  Written from the format description;
// behaviour notes cite the reference lines they reproduce.
#include "hitsio.hpp"
#include "pinflate.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <mutex>
#include <sstream>
#include <thread>
#include <vector>

namespace hitsio_detail {

// ---------------------------------------------------------------- buffered input (plain or zlib)
class ByteSource {
public:
    ByteSource(const std::string &fileName, bool &ok)
    {
        fp = std::fopen(fileName.c_str(), "rb");
        ok = fp != nullptr;
        if (!ok) return;
        int c = std::fgetc(fp);
        if (c != EOF) std::ungetc(c, fp);
        compressed = (c == 0x78); // "lazy but sufficient" zlib sniff, src/hitsio.cpp:258
        if (compressed) open_parallel(fileName);
        if (compressed && !par) {
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit(&zs) != Z_OK) { ok = false; return; }
            zinit = true;
        }
    }
    ~ByteSource()
    {
        if (worker.joinable()) {
            { std::lock_guard<std::mutex> lk(mtx); stop = true; }
            cv.notify_all();
            worker.join();
        }
        if (zinit) inflateEnd(&zs);
        if (par) {
            delete par;
            if (map_base) munmap((void *)map_base, map_len);
            if (std::getenv("MMSEQ_TIMING"))
                std::fprintf(stderr, "[timing] hits file: inflated by %d threads (host/pinflate.hpp); the reader waited for inflated data %.1f s\n", par_threads, t_reader_wait);
        }
        if (fp) std::fclose(fp);
        if (compressed && !map_base && std::getenv("MMSEQ_TIMING"))
            std::fprintf(stderr, "[timing] hits file: inflate thread busy %.1f s, waited for a free slab %.1f s; the reader waited for inflated data %.1f s\n",
                         t_inflate, t_inflate_wait, t_reader_wait);
    }
    // next byte or -1 at end of data
    int peek()
    {
        if (pos == len && !fill()) return -1;
        return (unsigned char)buf[pos];
    }
    int get()
    {
        int c = peek();
        if (c >= 0) ++pos;
        return c;
    }
    bool atEnd() { return peek() < 0; }
    // std::getline semantics: reads up to '\n' (consumed, not stored). Returns false if nothing at all could be read.
    bool getline(std::string &out)
    {
        out.clear();
        if (atEnd()) return false;
        for (;;) {
            if (pos == len && !fill()) return true; // last line without newline
            const char *p = (const char *)std::memchr(buf + pos, '\n', len - pos);
            if (p) {
                out.append(buf + pos, p - (buf + pos));
                pos = (size_t)(p - buf) + 1;
                return true;
            }
            out.append(buf + pos, len - pos);
            pos = len;
        }
    }
    bool read(void *dst, size_t n)
    {
        char *d = (char *)dst;
        while (n) {
            if (pos == len && !fill()) return false;
            size_t take = std::min(n, len - pos);
            std::memcpy(d, buf + pos, take);
            pos += take; d += take; n -= take;
        }
        return true;
    }
    // the bytes already inflated and not yet consumed, for parsers that work on the buffer itself; advance(n <= size) consumes
    void span(const char *&p, size_t &n) { if (pos == len) (void)fill(); p = buf + pos; n = len - pos; }
    void advance(size_t n) { pos += n; }
    bool readU32(uint32_t &v)
    {
        unsigned char b[4];
        if (!read(b, 4)) return false;
        v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); // little-endian raw
        return true;
    }

private:
    // A compressed file is ONE zlib stream (src/hitsio.cpp:127): it inflates in a thread of its own, a ring of slabs ahead of the
    // record decoding, which was a third of the reader's time at 50 M reads.  The ring is deep (256 MB): inflating bounds the ingest
    // of a large file, and with four slabs it stood still for 1.0 s of 14.5 s whenever a later stage (a growing hash table) paused.
    static constexpr int NSLAB = 256;
    static constexpr size_t SLAB = 1u << 20;
    struct Slab { std::vector<char> data; size_t len = 0; int state = 0; }; // 0 free, 1 filled, 2 being read
    // A large compressed file of its own (not a pipe) is inflated by several threads: host/pinflate.hpp.  MMSEQ_INFLATE_THREADS (default:
    // half the CPUs of the container's quota, at most 8; 1 = zlib on one thread as before), MMSEQ_INFLATE_CHUNK (compressed bytes per
    // chunk, default 4 MB) and MMSEQ_INFLATE_MIN (smallest file that takes this path, default 8 MB) are for tests and measurements.
    static long env_long(const char *name, long dflt)
    {
        const char *v = std::getenv(name);
        return v && *v ? std::atol(v) : dflt;
    }
    static int default_inflate_threads()
    {
        long cpus = (long)std::thread::hardware_concurrency();
        if (FILE *q = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2 quota: "max" or "<quota> <period>"
            char a[32];
            long per = 0;
            if (std::fscanf(q, "%31s %ld", a, &per) == 2 && std::strcmp(a, "max") != 0 && per > 0) cpus = std::min(cpus, (std::atol(a) + per - 1) / per);
            std::fclose(q);
        }
        return (int)std::max(1L, std::min(8L, cpus / 2));
    }
    void open_parallel(const std::string &fileName)
    {
        par_threads = (int)env_long("MMSEQ_INFLATE_THREADS", default_inflate_threads());
        if (par_threads < 2) return;
        struct stat st;
        if (fstat(fileno(fp), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < env_long("MMSEQ_INFLATE_MIN", 8L << 20)) return;
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
        if (m == MAP_FAILED) return;
        (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        map_base = (const uint8_t *)m;
        map_len = (size_t)st.st_size;
        par = new pinflate::Stream(map_base, map_len, par_threads, (size_t)env_long("MMSEQ_INFLATE_CHUNK", 4L << 20), /* start_now = */ false);
        if (par->longest_stretch_bytes() > (256u << 20)) { // stored / fixed blocks for hundreds of MB (incompressible data): one decode would hold gigabytes of symbols
            delete par;                                     // (no worker has started: nothing to wait for)
            par = nullptr;
            munmap((void *)map_base, map_len);
            map_base = nullptr;
        } else par->start();
    }
    bool fill()
    {
        pos = len = 0;
        if (par) {
            const uint8_t *p = nullptr;
            size_t n = 0;
            const auto w0 = std::chrono::steady_clock::now();
            const bool more = par->next(p, n);
            t_reader_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            if (!more) {
                if (!par->error().empty()) { std::cerr << par->error() << "\n"; hits_die(); }
                return false;
            }
            buf = (const char *)p;
            len = n;
            return true;
        }
        if (!compressed) {
            len = std::fread(raw, 1, sizeof raw, fp);
            buf = raw;
            return len > 0;
        }
        if (!worker.joinable() && !zdone) {
            worker = std::thread([this] { inflate_ahead(); });
        }
        std::unique_lock<std::mutex> lk(mtx);
        if (reading >= 0) { ring[reading].state = 0; reading = -1; cv.notify_all(); }
        const auto w0 = std::chrono::steady_clock::now();
        cv.wait(lk, [&] { return ring[next_read].state == 1 || zdone; });
        t_reader_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        if (ring[next_read].state != 1) return false;
        reading = next_read;
        next_read = (next_read + 1) % NSLAB;
        ring[reading].state = 2;
        buf = ring[reading].data.data();
        len = ring[reading].len;
        return len > 0;
    }
    void inflate_ahead()
    {
        for (int w = 0;; w = (w + 1) % NSLAB) {
            const auto w0 = std::chrono::steady_clock::now();
            {
                std::unique_lock<std::mutex> lk(mtx);
                cv.wait(lk, [&] { return ring[w].state == 0 || stop; });
                if (stop) return;
            }
            const auto w1 = std::chrono::steady_clock::now();
            t_inflate_wait += std::chrono::duration<double>(w1 - w0).count();
            Slab &sl = ring[w];
            if (sl.data.size() != SLAB) sl.data.resize(SLAB);   // (a small file never touches most of the ring)
            zs.next_out = (Bytef *)sl.data.data();
            zs.avail_out = (uInt)SLAB;
            bool end = false;
            while (zs.avail_out == SLAB) {
                if (zs.avail_in == 0) {
                    zs.avail_in = (uInt)std::fread(inbuf, 1, sizeof inbuf, fp);
                    zs.next_in = (Bytef *)inbuf;
                    if (zs.avail_in == 0) { end = true; break; } // truncated stream: stop like a short read
                }
                int rc = inflate(&zs, Z_NO_FLUSH);
                if (rc == Z_STREAM_END) { end = true; break; }
                if (rc != Z_OK && rc != Z_BUF_ERROR) {
                    std::cerr << "Error decompressing hits file (zlib error " << rc << ").\n";
                    hits_die();
                }
            }
            sl.len = SLAB - zs.avail_out;
            t_inflate += std::chrono::duration<double>(std::chrono::steady_clock::now() - w1).count();
            {
                std::lock_guard<std::mutex> lk(mtx);
                if (sl.len) sl.state = 1;
                if (end) zdone = true;
            }
            cv.notify_all();
            if (end) return;
        }
    }
    FILE *fp = nullptr;
    bool compressed = false, zinit = false, zdone = false, stop = false;
    z_stream zs;
    char raw[1 << 16];
    char inbuf[1 << 16];
    const char *buf = raw;
    size_t pos = 0, len = 0;
    Slab ring[NSLAB];
    int next_read = 0, reading = -1;
    std::mutex mtx;
    std::condition_variable cv;
    std::thread worker;
    double t_inflate = 0.0, t_inflate_wait = 0.0, t_reader_wait = 0.0; // MMSEQ_TIMING: which of the two threads paces the reader
    pinflate::Stream *par = nullptr;     // the parallel inflater over the mapped file, when it applies
    const uint8_t *map_base = nullptr;
    size_t map_len = 0;
    int par_threads = 1;
};

// ---------------------------------------------------------------- output (plain or zlib level 1)
class ByteSink {
public:
    ByteSink(FILE *f, bool compress) : fp(f), compressed(compress)
    {
        if (compressed) {
            std::memset(&zs, 0, sizeof zs);
            if (deflateInit(&zs, Z_BEST_SPEED) != Z_OK) { // zlib::best_speed, src/hitsio.cpp:127
                std::cerr << "Error initialising zlib.\n";
                hits_die();
            }
        }
    }
    ~ByteSink() { finish(); }
    void write(const void *p, size_t n)
    {
        if (!compressed) { std::fwrite(p, 1, n, fp); return; }
        zs.next_in = (Bytef *)p;
        zs.avail_in = (uInt)n;
        while (zs.avail_in) pump(Z_NO_FLUSH);
    }
    void writeStr(const std::string &s) { write(s.data(), s.size()); write("\n", 1); }
    void writeU32(uint64_t v)
    {
        if (v > std::numeric_limits<uint32_t>::max()) {
            std::cerr << "Numeric value overflow when writing the hits file" << std::endl; // src/hitsio.cpp:24-27
            hits_die();
        }
        unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)};
        write(b, 4);
    }
    void writeSmall(uint64_t v) // 1 byte if < 255 else 0xFF + u32, src/hitsio.cpp:36-45
    {
        if (v < 255) { unsigned char b = (unsigned char)v; write(&b, 1); }
        else { unsigned char b = 255; write(&b, 1); writeU32(v); }
    }
    void finish()
    {
        if (done) return;
        done = true;
        if (compressed) {
            zs.avail_in = 0;
            int rc;
            do { rc = pump(Z_FINISH); } while (rc != Z_STREAM_END);
            deflateEnd(&zs);
        }
        std::fflush(fp);
    }

private:
    int pump(int flush)
    {
        zs.next_out = (Bytef *)outbuf;
        zs.avail_out = sizeof outbuf;
        int rc = deflate(&zs, flush);
        std::fwrite(outbuf, 1, sizeof outbuf - zs.avail_out, fp);
        return rc;
    }
    FILE *fp;
    bool compressed, done = false;
    z_stream zs;
    char outbuf[1 << 16];
};

static std::string fmt_double(double v) // default ostream formatting (6 significant digits), src/hitsio.cpp:7-12
{
    std::ostringstream s;
    s << v;
    return s.str();
}
static double parse_double(const std::string &s) // src/hitsio.cpp:14-20
{
    std::istringstream i(s);
    double x;
    if (!(i >> x)) return 0;
    return x;
}

} // namespace hitsio_detail

using namespace hitsio_detail;

// ================================================================= writer
HitsfileWriter::HitsfileWriter(std::string fmt, FILE *out)
{
    hitsfileSchema = (!fmt.empty() && fmt[0] == 't') ? 0 : 1; // default binary, src/hitsio.cpp:118-126
    sink.reset(new ByteSink(out, hitsfileSchema == 1));
}
HitsfileWriter::~HitsfileWriter() { close(); }
void HitsfileWriter::close() { if (sink) sink->finish(); }

void HitsfileWriter::addTranscriptMetaData(std::string name, double efflen, int truelen)
{
    transcriptName.push_back(name);
    transcriptEffectiveLength.insert(std::make_pair(name, efflen));
    transcriptTrueLength.insert(std::make_pair(name, truelen));
    const uint32_t idx = (uint32_t)transcriptToIndex.size(); // evaluated BEFORE the insertion, src/hitsio.cpp:135-140
    transcriptToIndex[name] = idx;
}
void HitsfileWriter::addGeneIsoformRecord(std::string geneName)
{
    geneIsoforms.insert(std::make_pair(geneName, std::vector<std::string>()));
    currentGeneName = geneName;
}
void HitsfileWriter::addTranscriptToGeneIsoformRecord(std::string name) { geneIsoforms[currentGeneName].push_back(name); }
void HitsfileWriter::addIdenticalTranscriptsRecord() { identicalTranscripts.push_back(std::vector<std::string>()); }
void HitsfileWriter::addTranscriptToIdenticalTranscriptsRecord(std::string name) { identicalTranscripts.back().push_back(name); }

void HitsfileWriter::writeHeaderSchema0()
{
    std::ostringstream o;
    for (size_t i = 0; i < transcriptName.size(); i++)
        o << "@TranscriptMetaData\t" << transcriptName[i] << "\t" << transcriptEffectiveLength[transcriptName[i]] << "\t"
          << transcriptTrueLength[transcriptName[i]] << "\n";
    for (auto &g : geneIsoforms) { // std::map order of gene ids
        o << "@GeneIsoforms\t" << g.first;
        for (auto &t : g.second) o << "\t" << t;
        o << "\n";
    }
    for (auto &set : identicalTranscripts) {
        o << "@IdenticalTranscripts";
        for (auto &t : set) o << "\t" << t;
        o << "\n";
    }
    const std::string s = o.str();
    sink->write(s.data(), s.size());
}
void HitsfileWriter::writeHeaderSchema1()
{
    sink->writeStr(MMSEQ_HEADER);
    sink->writeU32((uint64_t)hitsfileSchema);
    sink->writeU32(transcriptName.size());
    for (size_t i = 0; i < transcriptName.size(); i++) {
        sink->writeStr(transcriptName[i]);
        sink->writeStr(fmt_double(transcriptEffectiveLength[transcriptName[i]]));
        sink->writeU32((uint64_t)(uint32_t)transcriptTrueLength[transcriptName[i]]);
    }
    sink->writeU32(geneIsoforms.size());
    for (auto &g : geneIsoforms) {
        sink->writeStr(g.first);
        sink->writeU32(g.second.size());
        for (auto &t : g.second) sink->writeStr(t);
    }
    sink->writeU32(identicalTranscripts.size());
    for (auto &set : identicalTranscripts) {
        sink->writeU32(set.size());
        for (auto &t : set) sink->writeStr(t);
    }
}
void HitsfileWriter::writeHeader() { if (hitsfileSchema == 0) writeHeaderSchema0(); else writeHeaderSchema1(); }

void HitsfileWriter::addReadMapRecord(std::string readName) { currentReadName = readName; currentReadTranscripts.clear(); currentReadIndices.clear(); }
void HitsfileWriter::addTranscriptToReadMapRecord(std::string name) { currentReadTranscripts.push_back(name); }
void HitsfileWriter::addTranscriptIndexToReadMapRecord(uint32_t index)
{
    if (index >= transcriptName.size()) { std::cerr << "Error: transcript index " << index << " not in the header.\n"; hits_die(); }
    currentReadIndices.push_back(index);
}

void HitsfileWriter::writeReadMapRecordSchema0()
{
    std::string s = ">" + currentReadName + "\n";
    for (auto &t : currentReadTranscripts) { s += t; s += "\n"; }
    for (uint32_t i : currentReadIndices) { s += transcriptName[i]; s += "\n"; }
    sink->write(s.data(), s.size());
}
void HitsfileWriter::writeReadMapRecordSchema1()
{
    // delta coding of the read name against the previous one, src/hitsio.cpp:77-100
    const std::string &cur = currentReadName;
    size_t nBeg = 0, nEnd = 0;
    const size_t lim = std::min(deltaBuffer.size(), cur.size());
    while (nBeg < lim && deltaBuffer[nBeg] == cur[nBeg]) nBeg++;
    while (nBeg + nEnd < lim && deltaBuffer[deltaBuffer.size() - 1 - nEnd] == cur[cur.size() - 1 - nEnd]) nEnd++;
    if (nBeg == 0 && nEnd == 0) {
        sink->writeStr(cur);
    } else {
        sink->writeStr("");
        sink->writeSmall(nBeg);
        sink->writeStr(cur.substr(nBeg, cur.size() - nBeg - nEnd));
        sink->writeSmall(nEnd);
    }
    deltaBuffer = cur;
    sink->writeU32(currentReadTranscripts.size() + currentReadIndices.size());
    for (auto &t : currentReadTranscripts) sink->writeU32(transcriptToIndex[t]);
    for (uint32_t i : currentReadIndices) sink->writeU32(i);
}
void HitsfileWriter::writeReadMapRecord() { if (hitsfileSchema == 0) writeReadMapRecordSchema0(); else writeReadMapRecordSchema1(); }

// ================================================================= reader
HitsfileReader::HitsfileReader(std::string fileName) : hitsfileSchema(-1), countReadMapRecord(0)
{
    bool ok = false;
    {
        ByteSource probe(fileName, ok);
        if (!ok) {
            std::cerr << "Error reading hits file \"" << fileName << "\".\n"; // src/hitsio.cpp:252-255
            hits_die();
        }
        // schema detection from the first line, src/hitsio.cpp:263-276
        std::string line;
        probe.getline(line);
        std::istringstream tokens(line);
        std::string token1;
        tokens >> token1;
        if (token1 == "@TranscriptMetaData") {
            hitsfileSchema = 0;
        } else {
            uint32_t v = 0xffffffffu;
            probe.readU32(v);
            hitsfileSchema = (v <= 4 && line == MMSEQ_HEADER) ? (int)v : -1;
        }
        if (hitsfileSchema < 0 || hitsfileSchema > 4) {
            std::cerr << "Input file \"" << fileName << "\" does not seem to be a hits file.\n";
            hits_die();
        }
    }
    src.reset(new ByteSource(fileName, ok)); // reopen from the start, src/hitsio.cpp:277-284
    if (!ok) {
        std::cerr << "Error reading hits file \"" << fileName << "\".\n";
        hits_die();
    }
}
HitsfileReader::~HitsfileReader() {}

void HitsfileReader::readHeaderSchema0(std::vector<std::string> *transcriptName, std::map<std::string, double> *efflen,
                                       std::map<std::string, int> *truelen, std::map<std::string, std::vector<std::string>> *geneIsoforms,
                                       std::vector<std::vector<std::string>> *identicalTranscripts)
{
    std::string line;
    while (!src->atEnd() && src->peek() != '>') {
        src->getline(line);
        std::istringstream tokens(line);
        std::string token1;
        tokens >> token1;
        if (token1 == "@TranscriptMetaData") {
            std::string name;
            double e = 0;
            int t = 0;
            tokens >> name >> e >> t;
            transcriptName->push_back(name);
            headerIndex.insert(std::make_pair(name, (uint32_t)headerTranscriptName.size()));
            headerTranscriptName.push_back(name);
            efflen->insert(std::make_pair(name, e));
            truelen->insert(std::make_pair(name, t));
        } else if (token1 == "@GeneIsoforms") {
            std::string gid, tid;
            std::vector<std::string> tids;
            tokens >> gid;
            while (tokens >> tid) tids.push_back(tid);
            geneIsoforms->insert(std::make_pair(gid, tids));
        } else if (token1 == "@IdenticalTranscripts") {
            std::string tid;
            std::vector<std::string> tids;
            while (tokens >> tid) tids.push_back(tid);
            identicalTranscripts->push_back(tids);
        } else {
            std::cerr << "Hits file looks malformed.\n"; // src/hitsio.cpp:324-327
            hits_die();
        }
    }
}

void HitsfileReader::readHeaderSchema1(std::vector<std::string> *transcriptName, std::map<std::string, double> *efflen,
                                       std::map<std::string, int> *truelen, std::map<std::string, std::vector<std::string>> *geneIsoforms,
                                       std::vector<std::vector<std::string>> *identicalTranscripts)
{
    std::string s, s2;
    uint32_t v = 0, n = 0;
    auto need = [&](bool ok) {
        if (!ok) { std::cerr << "Hits file looks malformed.\n"; hits_die(); }
    };
    need(src->getline(s)); // "MMSEQ_HITSFILE"
    need(src->readU32(v)); // schema
    hitsfileSchema = (int)v;
    need(src->readU32(n));
    for (uint32_t i = 0; i < n; i++) {
        need(src->getline(s));
        transcriptName->push_back(s);
        headerTranscriptName.push_back(s);
        need(src->getline(s2));
        efflen->insert(std::make_pair(s, parse_double(s2)));
        need(src->readU32(v));
        truelen->insert(std::make_pair(s, (int)v));
    }
    need(src->readU32(n));
    for (uint32_t i = 0; i < n; i++) {
        std::string gid;
        need(src->getline(gid));
        uint32_t cnt = 0;
        need(src->readU32(cnt));
        std::vector<std::string> tids;
        for (uint32_t j = 0; j < cnt; j++) { need(src->getline(s)); tids.push_back(s); }
        geneIsoforms->insert(std::make_pair(gid, tids));
    }
    need(src->readU32(n));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t cnt = 0;
        need(src->readU32(cnt));
        std::vector<std::string> tids;
        for (uint32_t j = 0; j < cnt; j++) { need(src->getline(s)); tids.push_back(s); }
        identicalTranscripts->push_back(tids);
    }
}

void HitsfileReader::readHeader(std::vector<std::string> *a, std::map<std::string, double> *b, std::map<std::string, int> *c,
                                std::map<std::string, std::vector<std::string>> *d, std::vector<std::vector<std::string>> *e)
{
    if (hitsfileSchema == 0) readHeaderSchema0(a, b, c, d, e);
    else if (hitsfileSchema == 1) readHeaderSchema1(a, b, c, d, e);
    else { std::cerr << "We should never get to this state!\n"; hits_die(); }
}

bool HitsfileReader::readReadMapRecordReadID(std::string &readID)
{
    if (hitsfileSchema == 0) {
        if (src->atEnd()) return false;
        src->getline(readID);
        if (readID.empty() || readID[0] != '>') { std::cerr << "Hits file looks malformed.\n"; hits_die(); }
        readID = readID.substr(1);
        if (src->atEnd()) { // src/hitsio.cpp:336-340
            std::cerr << "Warning: read record without any mapping transcripts found"
                      << " at the end of the hits file. The hits file may be corrupted.\n";
            return false;
        }
        return true;
    }
    // schema 1: delta-decoded name, src/hitsio.cpp:102-115, :413-420
    std::string s;
    if (!src->getline(s)) return false;
    if (s.empty()) {
        auto small = [&](uint32_t &v) {
            int b = src->get();
            if (b < 0) return false;
            if (b == 255) return src->readU32(v);
            v = (uint32_t)b;
            return true;
        };
        uint32_t nBeg = 0, nEnd = 0;
        if (!small(nBeg)) return false;
        if (!src->getline(s)) return false;
        if (!small(nEnd)) return false;
        if (nBeg > deltaBuffer.size() || nEnd > deltaBuffer.size()) { std::cerr << "Hits file looks malformed.\n"; hits_die(); }
        s = deltaBuffer.substr(0, nBeg) + s + deltaBuffer.substr(deltaBuffer.size() - nEnd, nEnd);
    }
    deltaBuffer = s;
    uint32_t cnt = 0;
    if (!src->readU32(cnt)) return false;
    readID = s;
    countReadMapRecord = cnt;
    return true;
}

bool HitsfileReader::readReadMapRecordTranscriptIndex(uint32_t &index)
{
    if (hitsfileSchema == 0) {
        std::string name;
        if (!readReadMapRecordTranscriptID(name)) return false;
        auto it = headerIndex.find(name);
        if (it == headerIndex.end()) { index = 0xffffffffu; return true; } // not in the header: caller reports it
        index = it->second;
        return true;
    }
    if (countReadMapRecord == 0 || src->atEnd()) return false;
    uint32_t v = 0;
    if (!src->readU32(v)) return false;
    countReadMapRecord--;
    index = v;
    return true;
}

bool HitsfileReader::skipReadMapRecordReadID()
{
    if (hitsfileSchema == 0) { std::string unused; return readReadMapRecordReadID(unused); }
    std::string &s = deltaBuffer; // scratch only: the names are not reconstructed
    if (!src->getline(s)) return false;
    if (s.empty()) {
        auto small = [&]() {
            int b = src->get();
            if (b < 0) return false;
            uint32_t v;
            return b != 255 || src->readU32(v);
        };
        if (!small() || !src->getline(s) || !small()) return false;
    }
    uint32_t cnt = 0;
    if (!src->readU32(cnt)) return false;
    countReadMapRecord = cnt;
    return true;
}

bool HitsfileReader::readReadMapRecordTranscriptIndices(std::vector<uint32_t> &out)
{
    if (hitsfileSchema == 0) {
        uint32_t v = 0;
        while (readReadMapRecordTranscriptIndex(v)) out.push_back(v);
        return true;
    }
    const size_t n = countReadMapRecord, at = out.size();
    if (n == 0) return true;
    // a record cannot name more transcripts than the header has (distinct indices): a corrupt count must not become a 16 GB allocation
    if (n > headerTranscriptName.size()) { std::cerr << "Hits file looks malformed.\n"; hits_die(); }
    out.resize(at + n);
    unsigned char *raw = (unsigned char *)(out.data() + at); // decoded in place: 4 little-endian bytes per index
    if (!src->read(raw, n * 4)) { out.resize(at); countReadMapRecord = 0; return false; }
    for (size_t i = 0; i < n; ++i) {
        const unsigned char *b = raw + 4 * i;
        out[at + i] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    }
    countReadMapRecord = 0;
    return true;
}

static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "readReadMapRecordsBulk copies the format's little-endian u32s as they are");
bool HitsfileReader::readReadMapRecordsBulk(std::vector<uint32_t> &lens, std::vector<uint32_t> &idx, size_t max_records)
{
    const size_t n_header = headerTranscriptName.size();
    for (size_t got = 0; got < max_records;) {
        if (hitsfileSchema == 1) {
            // fast path: records that lie inside the inflated buffer are parsed in place (src/hitsio.cpp:413-439: the name line -- empty
            // when delta-coded: count byte or 0xFF + u32, middle part, count byte --, u32 count, count x u32 indices, little-endian)
            const char *p;
            size_t avail;
            src->span(p, avail);
            const char *q = p, *const e = p + avail;
            size_t at = idx.size();
            while (got < max_records) {
                const char *r = q;
                const char *nl = (const char *)std::memchr(r, '\n', (size_t)(e - r));
                if (!nl) break;
                if (nl == r) { // delta-coded name
                    ++r;
                    auto small = [&]() { if (r >= e) return false; const unsigned char b = (unsigned char)*r++; if (b == 255) { if (e - r < 4) return false; r += 4; } return true; };
                    if (!small()) break;
                    nl = (const char *)std::memchr(r, '\n', (size_t)(e - r));
                    if (!nl) break;
                    r = nl + 1;
                    if (!small()) break;
                } else r = nl + 1;
                if (e - r < 4) break;
                uint32_t cnt;
                std::memcpy(&cnt, r, 4); // (little-endian host: the format's byte order)
                r += 4;
                if (cnt > n_header) { std::cerr << "Hits file looks malformed.\n"; hits_die(); }
                if ((size_t)(e - r) < (size_t)cnt * 4) break;
                if (idx.size() < at + cnt) idx.resize(std::max(idx.size() * 2, at + cnt + 1024));
                std::memcpy(idx.data() + at, r, (size_t)cnt * 4);
                at += cnt;
                q = r + (size_t)cnt * 4;
                lens.push_back(cnt);
                ++got;
            }
            idx.resize(at);
            src->advance((size_t)(q - p));
            if (got >= max_records) return true;
        }
        // a record that straddles the end of the buffer (or the text schema): one record through the byte-wise reader
        if (!skipReadMapRecordReadID()) return false;
        const size_t before = idx.size();
        readReadMapRecordTranscriptIndices(idx);
        lens.push_back((uint32_t)(idx.size() - before));
        ++got;
    }
    return true;
}

bool HitsfileReader::readReadMapRecordTranscriptID(std::string &transcriptID)
{
    if (hitsfileSchema == 0) {
        if (src->atEnd() || src->peek() == '>') return false;
        return src->getline(transcriptID);
    }
    uint32_t v = 0;
    if (!readReadMapRecordTranscriptIndex(v)) return false;
    if (v >= headerTranscriptName.size()) { std::cerr << "Hits file looks malformed.\n"; hits_die(); }
    transcriptID = headerTranscriptName[v];
    return true;
}
