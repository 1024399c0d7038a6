// huffenc_test IN OUT.gz [PIECE]: gzip member of IN made of huffenc pieces (tests/test_huffenc.py reads it back with Python's gzip);
// prints the encoder's speed next to zlib's Z_HUFFMAN_ONLY on the same pieces.
#include "huffenc.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <zlib.h>

static std::string zlib_piece(const char *p, size_t n)
{
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_HUFFMAN_ONLY);
    std::string out(deflateBound(&zs, (uLong)n) + 16, '\0');
    zs.next_in = (Bytef *)p; zs.avail_in = (uInt)n;
    zs.next_out = (Bytef *)&out[0]; zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_SYNC_FLUSH);
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    return out;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::cerr << "usage: huffenc_test IN OUT.gz [PIECE]\n"; return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    const size_t piece = argc > 3 ? (size_t)atoll(argv[3]) : 290000;
    std::string body, zbody;
    size_t fallbacks = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t o = 0; o < text.size(); o += piece) {
        const size_t n = std::min(piece, text.size() - o);
        if (!huffenc::deflate_literals(text.data() + o, n, body)) { body += zlib_piece(text.data() + o, n); ++fallbacks; }
    }
    const auto t1 = std::chrono::steady_clock::now();
    for (size_t o = 0; o < text.size(); o += piece) zbody += zlib_piece(text.data() + o, std::min(piece, text.size() - o));
    const auto t2 = std::chrono::steady_clock::now();
    const uLong crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)text.data(), (uInt)text.size());
    const auto t3 = std::chrono::steady_clock::now();
    FILE *f = fopen(argv[2], "wb");
    const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    fwrite(hdr, 1, 10, f);
    fwrite(body.data(), 1, body.size(), f);
    const unsigned char fin[2] = {0x03, 0x00}; // final empty fixed-Huffman block
    fwrite(fin, 1, 2, f);
    unsigned char tr[8];
    const uint32_t c = (uint32_t)crc, n = (uint32_t)text.size();
    for (int i = 0; i < 4; ++i) { tr[i] = (unsigned char)(c >> (8 * i)); tr[4 + i] = (unsigned char)(n >> (8 * i)); }
    fwrite(tr, 1, 8, f);
    fclose(f);
    auto mbps = [&](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return (double)text.size() / 1e6 / std::max(1e-9, std::chrono::duration<double>(b - a).count());
    };
    printf("%zu bytes: huffenc %zu bytes at %.0f MB/s (%zu pieces left to zlib), zlib Z_HUFFMAN_ONLY %zu bytes at %.0f MB/s, crc32 %.0f MB/s\n", text.size(),
           body.size(), mbps(t0, t1), fallbacks, zbody.size(), mbps(t1, t2), mbps(t2, t3));
    return 0;
}
