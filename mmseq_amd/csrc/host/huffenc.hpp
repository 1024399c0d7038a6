// huffenc.hpp -- Huffman-only raw deflate of a text piece, for the trace files (src/mmseq.cpp:911-917 writes them through a gzip
// filter): lines of 6-digit numbers have next to no repeats for LZ77 to find, so a piece is ONE dynamic-Huffman block of literals
// (RFC 1951 section 3.2.7) followed by an empty stored block that byte-aligns it -- what zlib's Z_HUFFMAN_ONLY + Z_SYNC_FLUSH emits, at
// several times its speed: a byte histogram, code lengths from a two-queue Huffman construction, canonical codes, and a table-driven
// packing loop through a 64-bit bit buffer.  Pieces concatenate into a valid deflate stream; any inflate reads it.
// A piece whose code would be deeper than deflate's 15 bits (a very skewed histogram) is left to the caller's zlib path.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace huffenc {

struct BitSink {
    char *p;          // write position in a buffer the caller sized for the worst case
    uint64_t buf = 0;
    int n = 0;        // bits held in buf (< 32 between calls)
    explicit BitSink(char *dst) : p(dst) {}
    inline void put(uint32_t bits, int len) // len <= 32, bits packed from the least significant bit (RFC 1951 3.1.1)
    {
        buf |= (uint64_t)bits << n;
        n += len;
        if (n >= 32) {
            const uint32_t w = (uint32_t)buf;
            std::memcpy(p, &w, 4); // little endian host (x86-64)
            p += 4;
            buf >>= 32;
            n -= 32;
        }
    }
    inline void align()
    {
        while (n > 0) { *p++ = (char)(buf & 0xff); buf >>= 8; n -= 8; }
        buf = 0;
        n = 0;
    }
};

static inline uint32_t bit_reverse(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// Code lengths of an optimal prefix code for the symbols with freq > 0 (at least two of them); returns the maximum length.
static inline int huffman_lengths(const uint64_t *freq, int nsym, uint8_t *len)
{
    struct Node { uint64_t w; int left, right; };
    std::vector<std::pair<uint64_t, int>> leaves;
    for (int s = 0; s < nsym; ++s) { len[s] = 0; if (freq[s]) leaves.push_back({freq[s], s}); }
    std::sort(leaves.begin(), leaves.end());
    const int L = (int)leaves.size();
    std::vector<Node> nodes;
    nodes.reserve(2 * (size_t)L);
    for (auto &lf : leaves) nodes.push_back({lf.first, -1, lf.second});
    // two queues: the sorted leaves and the internal nodes in the order they are made (their weights are non-decreasing)
    int qa = 0, qb = L;
    auto pop = [&]() {
        if (qa < L && (qb >= (int)nodes.size() || nodes[(size_t)qa].w <= nodes[(size_t)qb].w)) return qa++;
        return qb++;
    };
    for (int made = 0; made < L - 1; ++made) {
        const int x = pop(), y = pop();
        nodes.push_back({nodes[(size_t)x].w + nodes[(size_t)y].w, x, y});
    }
    // depths from the root (the last node) down
    std::vector<int> depth(nodes.size(), 0);
    int maxlen = 0;
    for (int i = (int)nodes.size() - 1; i >= 0; --i) {
        const Node &nd = nodes[(size_t)i];
        if (nd.left < 0) { len[nd.right] = (uint8_t)depth[(size_t)i]; maxlen = std::max(maxlen, depth[(size_t)i]); }
        else { depth[(size_t)nd.left] = depth[(size_t)i] + 1; depth[(size_t)nd.right] = depth[(size_t)i] + 1; }
    }
    return maxlen;
}

// Appends the piece to `out` as raw deflate (not final, byte-aligned).  false: nothing was appended, use another encoder.
static inline bool deflate_literals(const char *text, size_t n, std::string &out)
{
    if (n == 0) return false;
    // histogram, four tables against store-to-load stalls on runs of one byte
    uint64_t freq[257];
    {
        uint32_t h[4][256];
        std::memset(h, 0, sizeof h);
        const unsigned char *p = (const unsigned char *)text;
        size_t i = 0;
        for (; i + 4 <= n; i += 4) { h[0][p[i]]++; h[1][p[i + 1]]++; h[2][p[i + 2]]++; h[3][p[i + 3]]++; }
        for (; i < n; ++i) h[0][p[i]]++;
        for (int s = 0; s < 256; ++s) freq[s] = (uint64_t)h[0][s] + h[1][s] + h[2][s] + h[3][s];
        if (n >= (1ull << 32)) return false; // (the 32-bit counters)
    }
    freq[256] = 1; // end of block
    uint8_t len[257];
    if (huffman_lengths(freq, 257, len) > 15) return false;
    int used = 0;
    for (int s = 0; s < 257; ++s) used += len[s] != 0;
    if (used < 2) return false; // (never: the end-of-block symbol and at least one literal -- both get length 1)
    // canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first stream
    uint32_t code[257], next[17], count[17];
    std::memset(count, 0, sizeof count);
    for (int s = 0; s < 257; ++s) count[len[s]]++;
    count[0] = 0;
    uint32_t c = 0;
    for (int b = 1; b <= 15; ++b) { c = (c + count[b - 1]) << 1; next[b] = c; }
    for (int s = 0; s < 257; ++s) code[s] = len[s] ? bit_reverse(next[len[s]]++, len[s]) : 0;

    const size_t start = out.size();
    out.resize(start + 2 * n + 1024);   // a literal takes at most 15 bits; the header ~ 150 bytes
    BitSink bs(&out[start]);
    bs.put(0, 1);      // BFINAL = 0
    bs.put(2, 2);      // BTYPE = 10: dynamic Huffman codes
    bs.put(0, 5);      // HLIT: 257 literal/length codes
    bs.put(1, 5);      // HDIST: 2 distance codes (one bit each, never used: what zlib itself sends for a block without matches)
    bs.put(15, 4);     // HCLEN: all 19 code length codes
    // the code length alphabet: lengths 0..15 as 4-bit codes (a complete code: code of value v is v), no repeat symbols 16-18
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int i = 0; i < 19; ++i) bs.put(order[i] >= 16 ? 0u : 4u, 3);
    for (int s = 0; s < 257; ++s) bs.put(bit_reverse(len[s], 4), 4);
    bs.put(bit_reverse(1, 4), 4); // the two distance codes: length 1 each
    bs.put(bit_reverse(1, 4), 4);
    // the literals, two at a time through a table of (code, length)
    {
        uint32_t tc[256];
        uint8_t tl[256];
        for (int s = 0; s < 256; ++s) { tc[s] = code[s]; tl[s] = len[s]; }
        const unsigned char *p = (const unsigned char *)text;
        size_t i = 0;
        for (; i + 2 <= n; i += 2) {
            const unsigned a = p[i], b = p[i + 1];
            bs.put(tc[a] | (tc[b] << tl[a]), tl[a] + tl[b]); // <= 30 bits
        }
        for (; i < n; ++i) bs.put(tc[p[i]], tl[p[i]]);
    }
    bs.put(code[256], len[256]);
    // empty stored block: 3 header bits, padding to the byte, LEN = 0, NLEN = 0xffff (the sync-flush marker)
    bs.put(0, 1);
    bs.put(0, 2);
    bs.align();
    std::memcpy(bs.p, "\x00\x00\xff\xff", 4);
    bs.p += 4;
    out.resize((size_t)(bs.p - out.data()));
    return true;
}

} // namespace huffenc
