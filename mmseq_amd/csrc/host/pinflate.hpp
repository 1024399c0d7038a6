// pinflate.hpp -- ONE zlib stream inflated by several threads.
//
// The binary hits file is a single zlib stream (src/hitsio.cpp:127: one zlib_compressor over the whole file), and inflating it
// bounded the ingest: 11.3 s of zlib on one thread for a 50 M-read file, with four other pipeline stages waiting on it.  A deflate
// stream has no index, but it can still be decoded from the middle (the idea of pugz / rapidgzip):
//   1. the compressed file is cut into chunks at fixed byte offsets; in every chunk a BLOCK FINDER looks for the first bit position
//      at which a dynamic-Huffman block header parses completely and validly (BFINAL = 0, BTYPE = 2, HLIT / HDIST in range, a
//      complete code-length code, code lengths that decode without an invalid repeat into a complete literal/length code with an
//      end-of-block symbol and a valid distance code): a candidate start;
//   2. every chunk is decoded from its candidate by an own inflate that writes 16-BIT symbols: a literal is its byte; a
//      back-reference that reaches before the chunk's first byte -- into the 32 KB window the chunk does not know -- becomes a
//      MARKER 0x8000 | position in that window, and copies of markers stay markers.  A chunk's decode ends exactly on the bit where
//      the next candidate starts (checked at every end of block; a candidate that is not a block boundary is passed over and the
//      decode runs on to the one after it: the chunk behind a false candidate is simply dropped);
//   3. in file order (cheap: 32 K look-ups per chunk) the window behind every accepted chunk is computed from the window before it
//      and the chunk's last 32 K symbols; with its window known a chunk's symbols are resolved to bytes -- by the worker threads
//      again -- and its Adler-32 is taken, combined in order and compared with the stream's trailer.
// Chunk 0 starts behind the two-byte zlib header with an empty window and is authoritative; every later chunk is accepted only if
// the chunk before it ended exactly on its start, so a false candidate can never contribute bytes.  Any stream zlib accepts decodes
// to the same bytes here (stored and fixed blocks are decoded, just never used as candidates); streams zlib rejects are rejected.
// Plain C++17 + zlib's adler32 / adler32_combine.  tests/test_pinflate.py checks it against Python's zlib on streams of every
// block type, with chunk sizes down to a few hundred bytes, and on truncated / damaged streams.
#pragma once
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace pinflate {

// ---------------------------------------------------------------------------------------------------------------- bit input
struct Bits {
    const uint8_t *base, *p, *end;
    uint64_t buf = 0;
    int cnt = 0;             // valid bits in buf
    bool over = false;       // read past the end of the data
    Bits(const uint8_t *b, const uint8_t *e, uint64_t bitpos) : base(b), p(b + (bitpos >> 3)), end(e)
    {
        if (p > end) { p = end; over = true; }
        refill();
        const int skip = (int)(bitpos & 7);
        buf >>= skip;
        cnt -= skip;
    }
    inline void refill()
    {
        if (end - p >= 8) {
            // eight bytes at once; the bits above cnt are the next byte's real low bits and are OR-ed in again, unchanged, next time
            uint64_t w;
            std::memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(int n)
    {
        buf >>= n;
        cnt -= n;
        if (cnt < 0) { over = true; cnt = 0; buf = 0; }
    }
    inline uint32_t take(int n)
    {
        if (cnt < n) refill();
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8 - (uint64_t)cnt; }
};

// ---------------------------------------------------------------------------------------------------------------- Huffman codes
// canonical code from code lengths (RFC 1951 3.2.2): a FAST table over the next FB bits for codes of at most FB bits, the
// count / symbol arrays of a bit-by-bit canonical walk for the longer ones
template <int FB, int MAXSYM>
struct Code {
    uint16_t fast[1 << FB];       // symbol << 4 | length, 0 = longer than FB bits (or no code)
    uint16_t count[16], symbol[MAXSYM];
    int max_len = 0;
    // 0 ok (complete), 1 incomplete, -1 over-subscribed
    int build(const uint8_t *len, int n)
    {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[len[i]]++;
        max_len = 0;
        for (int l = 15; l >= 1; --l) if (count[l]) { max_len = l; break; }
        int left = 1;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return -1;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        std::memset(fast, 0, sizeof fast);
        // canonical codes in increasing (length, symbol) order; a code's bits are sent most significant first, i.e. reversed in our
        // least-significant-bit-first buffer
        uint32_t code = 0;
        int idx = 0;
        for (int l = 1; l <= 15; ++l) {
            for (int k = 0; k < count[l]; ++k, ++idx, ++code) {
                if (l > FB) continue;
                uint32_t rev = 0;
                for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
                const uint16_t e = (uint16_t)(symbol[idx] << 4 | l);
                for (uint32_t fill = rev; fill < (1u << FB); fill += 1u << l) fast[fill] = e;
            }
            code <<= 1;
        }
        return left > 0 ? 1 : 0;
    }
    // next symbol, or -1 (no such code / out of data)
    inline int decode(Bits &in) const
    {
        if (in.cnt < 15) in.refill();
        const uint16_t e = fast[in.peek(FB)];
        if (e) { in.drop(e & 15); return e >> 4; }
        // longer than FB bits: walk the canonical code bit by bit (rare: frequent symbols have short codes)
        uint32_t code = 0, first = 0, index = 0;
        uint64_t b = in.buf;
        for (int l = 1; l <= max_len; ++l) {
            code |= (uint32_t)(b & 1);
            b >>= 1;
            const uint32_t c = count[l];
            if (code < first + c) { in.drop(l); return in.over ? -1 : symbol[index + (code - first)]; }
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

typedef Code<11, 288> LitCode;
typedef Code<9, 32> DistCode;

// The header of a dynamic block (behind its three type bits): code lengths into lit / dist.  false: not a valid header -- by the
// rules zlib's inflate applies (inflate.c: "too many length or distance symbols", "invalid code lengths set", "invalid bit length
// repeat", "invalid code -- missing end-of-block", "invalid literal/lengths set", "invalid distances set").
static inline bool read_dynamic_header(Bits &in, LitCode &lit, DistCode &dist)
{
    const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) cl[CL_ORDER[i]] = (uint8_t)in.take(3);
    Code<7, 19> pre;
    if (pre.build(cl, 19) != 0) return false;                       // the code-length code must be complete
    uint8_t len[286 + 30];
    int i = 0;
    while (i < hlit + hdist) {
        const int s = pre.decode(in);
        if (s < 0 || in.over) return false;
        if (s < 16) { len[i++] = (uint8_t)s; continue; }
        int rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = len[i - 1]; rep = 3 + (int)in.take(2); }
        else if (s == 17) rep = 3 + (int)in.take(3);
        else rep = 11 + (int)in.take(7);
        if (i + rep > hlit + hdist) return false;
        while (rep--) len[i++] = (uint8_t)val;
    }
    if (in.over || len[256] == 0) return false;                     // no end-of-block code
    const int rl = lit.build(len, hlit);
    if (rl < 0 || (rl > 0 && lit.max_len != 1)) return false;       // incomplete sets only in the one-code case, as zlib
    const int rd = dist.build(len + hlit, hdist);
    if (rd < 0 || (rd > 0 && dist.max_len > 1)) return false;
    return true;
}

// first bit position >= from (and < to) at which a non-final dynamic block header parses; ~0 if none
static inline uint64_t find_block(const uint8_t *base, const uint8_t *end, uint64_t from, uint64_t to)
{
    static thread_local LitCode lit;
    static thread_local DistCode dist;
    for (uint64_t pos = from; pos < to; ++pos) {
        // BFINAL = 0, BTYPE = 2 (bits: 0, then 0 1 least significant first): the three bits read as the number 4
        const uint64_t byte = pos >> 3;
        if (base + byte + 2 >= end) break;
        const uint32_t w = (uint32_t)base[byte] | (uint32_t)base[byte + 1] << 8;
        if (((w >> (pos & 7)) & 7u) != 4u) continue;
        Bits in(base, end, pos + 3);
        if (!read_dynamic_header(in, lit, dist)) continue;
        // a header alone is right one time in ~10^5 at random; the first symbols must decode as well
        bool ok = true;
        for (int k = 0; k < 64 && ok; ++k) {
            const int s = lit.decode(in);
            if (s < 0 || s > 285 || in.over) { ok = false; break; }
            if (s == 256) break;
            if (s > 256) {
                in.take(LEN_EXTRA[s - 257]);
                const int d = dist.decode(in);
                if (d < 0 || d > 29) { ok = false; break; }
                in.take(DIST_EXTRA[d]);
            }
        }
        if (ok && !in.over) return pos;
    }
    return ~0ull;
}

// ---------------------------------------------------------------------------------------------------------------- one chunk
struct Chunk {
    uint64_t start_bit = ~0ull;       // where its decode starts (candidate), ~0: none found
    uint64_t end_bit = 0;             // where it ended (a block boundary, or the end of the final block)
    bool final_seen = false;          // its decode reached the end of the stream's final block
    bool failed = false;              // invalid data met (a false candidate, or a damaged stream)
    bool too_big = false;             // ... or more symbols than one chunk may hold
    uint16_t *sym_buf = nullptr;      // decoded symbols (malloc): byte, or 0x8000 | window position
    size_t sym_len = 0;
    std::vector<uint8_t> window_in;   // the 32 KB before it (filled in file order)
    std::vector<uint8_t> bytes;       // resolved
    size_t n_bytes = 0;
    uint32_t adler = 1;
    int state = 0;                    // 0 to decode, 1 decoding, 2 decoded, 3 accepted + window known, 4 resolving, 5 ready, 6 consumed, 7 dropped
};

// decode from c.start_bit; known_empty_window: chunk 0 (a reference before the start is an error, as in zlib: "invalid distance too far back").
// targets: ascending candidate starts behind this chunk's; the decode ends on the first of them it meets exactly at an end of block.
static inline void decode_chunk(const uint8_t *base, const uint8_t *end, Chunk &c, bool known_empty_window, const std::vector<uint64_t> &starts, size_t first_target,
                                const std::atomic<bool> *stop = nullptr)
{
    static thread_local LitCode lit, fixed_lit;
    static thread_local DistCode dist, fixed_dist;
    static thread_local bool fixed_built = false;
    if (!fixed_built) {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        fixed_lit.build(l, 288);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        fixed_dist.build(d, 30);
        fixed_built = true;
    }
    Bits in(base, end, c.start_bit);
    // the symbols go to a malloc'ed buffer (a std::vector would zero-fill what the decode is about to write); sized for the usual
    // ratio, grown by doubling
    size_t cap = 1u << 20, o = 0;
    if (first_target < starts.size() && starts[first_target] != ~0ull) cap = std::max<size_t>(cap, (size_t)((starts[first_target] - c.start_bit) / 8) * 3);
    uint16_t *out = (uint16_t *)std::malloc(cap * 2);
    // (a chunk is held in memory whole, as 16-bit symbols: one that inflates beyond MAX_SYMBOLS = 2^30 of them, 2 GB -- deflate reaches
    // 1032 : 1 on runs of one byte -- is refused rather than allowed to take the machine's memory, and so is one the allocator refuses;
    // zlib on one thread streams such a file through fixed buffers)
    constexpr size_t MAX_SYMBOLS = (size_t)1 << 30;
    bool too_big = out == nullptr;
    auto grow = [&](size_t need) {
        while (o + need > cap) cap *= 2;
        if (cap > MAX_SYMBOLS) { too_big = true; return; }
        uint16_t *bigger = (uint16_t *)std::realloc(out, cap * 2);
        if (!bigger) { too_big = true; return; } // (out stays valid and is freed with the chunk)
        out = bigger;
    };
    auto finish = [&]() { c.sym_buf = out; c.sym_len = o; c.end_bit = in.bitpos(); };
    size_t tgt = first_target;
    auto fail = [&]() { c.failed = true; finish(); };
    if (too_big) { c.too_big = true; return fail(); }
    for (;;) {
        if (stop && stop->load(std::memory_order_relaxed)) return fail(); // the stream is being torn down: nobody will read this chunk
        // at a block boundary: have we arrived where the next chunk starts?
        const uint64_t pos = in.bitpos();
        while (tgt < starts.size() && (starts[tgt] == ~0ull || starts[tgt] < pos)) ++tgt;
        if (tgt < starts.size() && starts[tgt] == pos && pos != c.start_bit) return finish();
        const uint32_t bfinal = in.take(1), btype = in.take(2);
        if (in.over) return fail();
        if (btype == 0) { // stored
            in.drop(in.cnt & 7); // to the byte boundary
            const uint32_t len = in.take(16), nlen = in.take(16);
            if (in.over || (len ^ 0xffffu) != nlen) return fail();
            if (o + len + 320 > cap) { grow(len + 320); if (too_big) { c.too_big = true; return fail(); } }
            for (uint32_t k = 0; k < len; ++k) out[o++] = (uint16_t)in.take(8);
            if (in.over) return fail();
        } else if (btype == 3) return fail();
        else {
            const LitCode *L = &fixed_lit;
            const DistCode *D = &fixed_dist;
            if (btype == 2) {
                if (!read_dynamic_header(in, lit, dist)) return fail();
                L = &lit; D = &dist;
            }
            for (;;) {
                if (o + 320 > cap) { grow(320); if (too_big) { c.too_big = true; return fail(); } }
                // one refill serves a whole literal / length + distance group: 15 + 5 + 15 + 13 bits at most
                if (in.cnt < 48) {
                    in.refill();
                    if (in.over) return fail(); // (a truncated stream: the zero bits behind its end would decode as symbols for ever)
                }
                int s;
                {
                    const uint16_t e = L->fast[in.peek(11)];
                    if (e) { in.drop(e & 15); s = e >> 4; }
                    else s = L->decode(in);
                }
                if (s < 256) {
                    if (s < 0) return fail();
                    out[o++] = (uint16_t)s;
                    continue;
                }
                if (s == 256) break;
                if (s > 285) return fail();
                const int xl = LEN_EXTRA[s - 257];
                const uint32_t len = LEN_BASE[s - 257] + in.peek(xl);
                in.drop(xl);
                int ds;
                {
                    const uint16_t e = D->fast[in.peek(9)];
                    if (e) { in.drop(e & 15); ds = e >> 4; }
                    else ds = D->decode(in);
                }
                if (ds < 0 || ds > 29) return fail();
                const int xd = DIST_EXTRA[ds];
                const uint32_t d = DIST_BASE[ds] + in.peek(xd);
                in.drop(xd);
                if (in.over) return fail();
                if (d > o) {
                    if (known_empty_window || d - o > 32768) return fail(); // before the start of the stream / beyond any window
                    // (part of) the source lies in the window this chunk does not know: markers
                    for (uint32_t k = 0; k < len; ++k, ++o) out[o] = d > o ? (uint16_t)(0x8000u | (uint32_t)(32768 - (d - o))) : out[o - d];
                } else if (d >= len && len > 16) {
                    std::memcpy(&out[o], &out[o - d], len * 2); // no overlap
                    o += len;
                } else {
                    const uint16_t *src = out + (o - d);
                    uint16_t *dst = out + o;
                    for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
                    o += len;
                }
            }
            if (in.over) return fail();
        }
        if (bfinal) { c.final_seen = true; in.drop(in.cnt & 7); return finish(); }
    }
}

// ---------------------------------------------------------------------------------------------------------------- the stream
// Inflates the zlib stream [data, data + size) with `threads` workers; next() hands the decompressed bytes out in order, a chunk at
// a time (zero copy: the buffer stays valid until the next call).  error() is non-empty after a failure (then next() returns false).
class Stream {
public:
    // start_now = false: only the candidate block starts are searched (cheap); the decode begins with start() -- a caller that may still
    // turn the stream down (longest_stretch_bytes) does so before any worker holds gigabytes of symbols
    Stream(const uint8_t *data, size_t size, int threads, size_t chunk_bytes, bool start_now = true) : base(data), end(data + size), T(threads < 1 ? 1 : threads)
    {
        if (size < 6 || (data[0] & 0x0f) != 8 || ((data[0] << 8 | data[1]) % 31) != 0 || (data[1] & 0x20)) { err = "not a zlib stream"; done_all = true; return; }
        if (chunk_bytes < 64) chunk_bytes = 64;
        const size_t n = (size - 2 + chunk_bytes - 1) / chunk_bytes;
        chunks.resize(n ? n : 1);
        starts.assign(chunks.size(), ~0ull);
        starts[0] = 16;
        chunk_sz = chunk_bytes;
        // candidates for every chunk first (cheap), then the decode / resolve loop
        std::atomic<size_t> next{1};
        auto finder = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < chunks.size();)
                starts[i] = find_block(base, end, (2 + (uint64_t)i * chunk_sz) * 8, (2 + (uint64_t)(i + 1) * chunk_sz) * 8);
        };
        {
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(finder);
            finder();
            for (auto &x : th) x.join();
        }
        for (size_t i = 0; i < chunks.size(); ++i) chunks[i].start_bit = starts[i];
        for (size_t i = 1; i < chunks.size(); ++i) if (starts[i] == ~0ull) chunks[i].state = 7; // no candidate: the chunk before runs through
        chunks[0].window_in.assign(32768, 0);
        if (start_now) start();
    }
    void start()
    {
        if (!workers.empty() || done_all) return;
        for (int t = 0; t < T; ++t) workers.emplace_back([this] { work(); });
    }
    ~Stream()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; stop_flag.store(true); }
        cv.notify_all();
        for (auto &w : workers) w.join();
        for (Chunk &c : chunks) std::free(c.sym_buf);
    }
    // the next run of decompressed bytes; false at the end of the stream or after an error
    bool next(const uint8_t *&p, size_t &n)
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (!err.empty()) return false;
            if (held != (size_t)-1) { // release the buffer handed out last time
                std::vector<uint8_t>().swap(chunks[held].bytes);
                chunks[held].state = 6;
                held = (size_t)-1;
                cv.notify_all();
            }
            while (read_at < chunks.size() && (chunks[read_at].state == 7 || chunks[read_at].state == 6)) ++read_at;
            if (read_at >= chunks.size() || (stream_ended && read_at > last_chunk)) return false;
            Chunk &c = chunks[read_at];
            if (c.state == 5) {
                held = read_at++;
                p = c.bytes.data();
                n = c.bytes.size();
                if (n == 0) continue;
                return true;
            }
            cv.wait(lk);
        }
    }
    const std::string &error() const { return err; }
    // the most compressed bytes ONE decode has to cover: the longest stretch between two candidate starts (or to the end).  A stream of
    // stored / fixed blocks has no candidates at all: one thread would decode all of it into 16-bit symbols -- callers with a plain
    // zlib path should take that instead when this is large
    uint64_t longest_stretch_bytes() const
    {
        uint64_t last = 16, worst = 0;
        for (size_t i = 1; i < starts.size(); ++i)
            if (starts[i] != ~0ull) { worst = std::max(worst, starts[i] - last); last = starts[i]; }
        worst = std::max(worst, (uint64_t)(end - base) * 8 - last);
        return worst / 8;
    }
    // chunks whose decode was used / chunks dropped (no candidate, or a candidate that was not a block boundary)
    void stats(size_t &accepted, size_t &dropped)
    {
        std::lock_guard<std::mutex> lk(mu);
        accepted = dropped = 0;
        for (const Chunk &c : chunks) { accepted += c.state >= 3 && c.state <= 6; dropped += c.state == 7; }
    }

private:
    void fail(const std::string &what)
    {
        if (err.empty()) err = what;
        cv.notify_all();
    }
    // file order: accept decoded chunks whose start is where the chunk before ended, hand them their window (mu held)
    void chain()
    {
        while (chain_at < chunks.size() && !stream_ended) {
            Chunk &c = chunks[chain_at];
            if (c.state == 7) { ++chain_at; continue; }
            if (c.start_bit < expect_bit) { // a candidate the chunk before ran over: not a block boundary (its decode, if any, is dropped)
                if (c.state == 1) return;   // still decoding: let it finish, then drop
                std::free(c.sym_buf); c.sym_buf = nullptr; c.sym_len = 0;
                c.state = 7;
                ++chain_at;
                continue;
            }
            if (c.state != 2) return;       // not decoded yet
            if (c.start_bit != expect_bit) return fail("inflate: lost the block chain (internal error)");
            if (c.failed && c.too_big) return fail("Error decompressing hits file (a stretch of the deflate stream inflates to more than 2 GB: set MMSEQ_INFLATE_THREADS=1).");
            if (c.failed) return fail("Error decompressing hits file (invalid deflate data).");
            c.window_in = window;           // (chunk 0: zeros, never referenced)
            // the window behind it: its last 32 K symbols over the window before it
            const size_t n = c.sym_len;
            std::vector<uint8_t> w(32768);
            for (size_t k = 0; k < 32768; ++k) {
                if (n >= 32768 - k) { const uint16_t v = c.sym_buf[n - (32768 - k)]; w[k] = v & 0x8000 ? window[v & 0x7fff] : (uint8_t)v; }
                else w[k] = window[k + n];
            }
            window.swap(w);
            expect_bit = c.end_bit;
            c.state = 3;
            if (c.final_seen) { stream_ended = true; last_chunk = chain_at; }
            ++chain_at;
        }
        if (chain_at >= chunks.size() && !stream_ended && err.empty()) fail("Error decompressing hits file (unexpected end of the deflate stream).");
    }
    void work()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (stop || !err.empty()) return;
            // 1. resolve the first accepted chunk that waits for it
            size_t pick = (size_t)-1;
            for (size_t i = resolve_from; i < chain_at; ++i) {
                if (chunks[i].state == 3) { pick = i; break; }
            }
            while (resolve_from < chain_at && chunks[resolve_from].state >= 4) ++resolve_from;
            if (pick != (size_t)-1) {
                Chunk &c = chunks[pick];
                c.state = 4;
                lk.unlock();
                const size_t n = c.sym_len;
                c.bytes.resize(n);
                // one table look-up per symbol: a byte is itself, a marker the window's byte
                std::vector<uint8_t> lut(65536, 0);
                for (int v = 0; v < 256; ++v) lut[v] = (uint8_t)v;
                std::memcpy(&lut[0x8000], c.window_in.data(), 32768);
                const uint16_t *s = c.sym_buf;
                uint8_t *b = c.bytes.data();
                for (size_t k = 0; k < n; ++k) b[k] = lut[s[k]];
                uint32_t a = 1;
                for (size_t k = 0; k < n; k += 1u << 30) a = (uint32_t)adler32(a, b + k, (uInt)std::min<size_t>(n - k, 1u << 30));
                c.adler = a;
                c.n_bytes = n;
                std::free(c.sym_buf); c.sym_buf = nullptr;
                std::vector<uint8_t>().swap(c.window_in);
                lk.lock();
                c.state = 5;
                finish_adler();
                cv.notify_all();
                continue;
            }
            // 2. decode the next chunk, not too far ahead of the reader
            size_t todo = (size_t)-1;
            if (!stream_ended) {
                while (decode_at < chunks.size() && chunks[decode_at].state != 0) ++decode_at;
                if (decode_at < chunks.size() && decode_at < read_at + (size_t)(3 * T + 2)) todo = decode_at++;
            }
            if (todo != (size_t)-1) {
                Chunk &c = chunks[todo];
                c.state = 1;
                lk.unlock();
                decode_chunk(base, end, c, todo == 0, starts, todo + 1, &stop_flag);
                lk.lock();
                c.state = 2;
                chain();
                cv.notify_all();
                continue;
            }
            cv.wait(lk);
        }
    }
    // Adler-32 of the whole stream, combined in file order as the chunks become ready; compared with the trailer at the end (mu held)
    void finish_adler()
    {
        while (adler_at < chunks.size()) {
            Chunk &c = chunks[adler_at];
            if (c.state == 7) { ++adler_at; continue; }
            if (c.state < 5) return;
            total_adler = (uint32_t)adler32_combine(total_adler, c.adler, (z_off_t)c.n_bytes);
            const bool last = stream_ended && adler_at == last_chunk;
            ++adler_at;
            if (last) {
                const uint64_t byte = expect_bit >> 3;
                if (base + byte + 4 > end) return fail("Error decompressing hits file (truncated zlib trailer).");
                const uint32_t want = (uint32_t)base[byte] << 24 | (uint32_t)base[byte + 1] << 16 | (uint32_t)base[byte + 2] << 8 | base[byte + 3];
                if (want != total_adler) return fail("Error decompressing hits file (zlib error -3: incorrect data check).");
                adler_at = chunks.size();
            }
        }
    }

    const uint8_t *base, *end;
    int T;
    size_t chunk_sz = 0;
    std::vector<Chunk> chunks;
    std::vector<uint64_t> starts;
    std::vector<uint8_t> window = std::vector<uint8_t>(32768, 0);
    uint64_t expect_bit = 16;
    size_t chain_at = 0, decode_at = 0, resolve_from = 0, read_at = 0, adler_at = 0, last_chunk = 0, held = (size_t)-1;
    bool stream_ended = false, stop = false, done_all = false;
    std::atomic<bool> stop_flag{false}; // `stop` for the decoders, which run without the lock (polled once per deflate block)
    uint32_t total_adler = 1;
    std::string err;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> workers;
};

} // namespace pinflate
