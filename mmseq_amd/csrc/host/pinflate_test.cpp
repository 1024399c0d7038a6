// pinflate_test -- inflates a zlib stream with host/pinflate.hpp (tests/test_pinflate.py compares the result with Python's zlib):
//   pinflate_test IN.z OUT THREADS CHUNK_BYTES      exit 0 and the bytes in OUT, or exit 2 and the error on stderr
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pinflate.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: pinflate_test IN.z OUT THREADS CHUNK_BYTES\n"); return 1; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> in;
    uint8_t buf[1 << 16];
    for (size_t n; (n = std::fread(buf, 1, sizeof buf, f)) > 0;) in.insert(in.end(), buf, buf + n);
    std::fclose(f);
    FILE *o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    {
        pinflate::Stream s(in.data(), in.size(), std::atoi(argv[3]), (size_t)std::atoll(argv[4]));
        const uint8_t *p;
        size_t n;
        while (s.next(p, n)) std::fwrite(p, 1, n, o);
        std::fclose(o);
        if (!s.error().empty()) { std::fprintf(stderr, "%s\n", s.error().c_str()); return 2; }
        size_t acc, drop;
        s.stats(acc, drop);
        if (std::getenv("PINFLATE_STATS")) std::fprintf(stderr, "chunks used %zu, dropped %zu\n", acc, drop);
    }
    return 0;
}
