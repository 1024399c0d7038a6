"""Differential fuzz of host/pinflate.hpp against Python's zlib (CPU): random payloads x compression parameters (level, strategy,
memLevel, window bits, flush points) x chunk sizes x thread counts, plus random damage (a stream zlib rejects must be rejected, one it
accepts must give zlib's bytes).   usage: pinflate_fuzz.py [n_cases] [first_seed]"""
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mmseq_amd", "csrc", "pinflate_test")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_pinflate import hits_like  # noqa: E402


def payload(rng):
    kind = int(rng.integers(0, 6))
    n = int(rng.integers(0, 400000))
    if kind == 0:
        return hits_like(rng, n // 60 + 1)
    if kind == 1:
        return bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    if kind == 2:
        return bytes(rng.integers(97, 101, size=n, dtype=np.uint8))
    if kind == 3:
        return (b"ACGT" * 64 + bytes(rng.integers(65, 70, size=32, dtype=np.uint8))) * (n // 300 + 1)
    if kind == 4:
        return bytes(n)
    parts = []
    for _ in range(int(rng.integers(1, 6))):
        parts.append(payload(rng)[:int(rng.integers(1, 90000))])
    return b"".join(parts)


def compress(rng, data):
    level = int(rng.choice([0, 1, 1, 1, 3, 6, 9]))
    strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
    mem = int(rng.choice([1, 2, 8, 8, 9]))
    wbits = int(rng.choice([9, 12, 15, 15, 15]))
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strat)
    out, at = [], 0
    for _ in range(int(rng.integers(0, 4))):
        cut = int(rng.integers(at, len(data) + 1))
        out.append(co.compress(data[at:cut]))
        out.append(co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))))
        at = cut
    out.append(co.compress(data[at:]))
    out.append(co.flush())
    return b"".join(out), (level, strat, mem, wbits)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tmp = tempfile.mkdtemp()
    src, dst = os.path.join(tmp, "in.z"), os.path.join(tmp, "out.bin")
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        data = payload(rng)
        comp, params = compress(rng, data)
        damaged = bool(rng.integers(0, 4) == 0) and len(comp) > 20
        if damaged:
            b = bytearray(comp)
            how = int(rng.integers(0, 3))
            if how == 0:
                for k in rng.integers(2, len(b), size=int(rng.integers(1, 4))):
                    b[int(k)] ^= 1 << int(rng.integers(0, 8))
            elif how == 1:
                b = b[:int(rng.integers(2, len(b)))]
            else:
                b[-int(rng.integers(1, 5))] ^= 0x20
            comp = bytes(b)
        try:
            want = zlib.decompress(comp)
            ok = True
        except zlib.error:
            want, ok = None, False
        threads, chunk = int(rng.integers(1, 9)), int(rng.choice([200, 700, 3000, 20000, 150000, 1 << 30]))
        open(src, "wb").write(comp)
        r = subprocess.run([BIN, src, dst, str(threads), str(chunk)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        got = open(dst, "rb").read()
        if ok:
            assert r.returncode == 0 and got == want, (seed, params, threads, chunk, damaged, r.returncode, r.stderr[-200:], len(got), len(want))
        else:
            assert r.returncode == 2, (seed, params, threads, chunk, "zlib rejects this stream, pinflate returned %d" % r.returncode)
        if seed % 50 == 0:
            print("seed %d ok (%d bytes -> %d, params %s, %d threads, chunk %d, damaged %s, zlib accepts %s)" % (seed, len(data), len(comp), params, threads, chunk, damaged, ok), flush=True)
    print("all %d cases agree with zlib" % n)


if __name__ == "__main__":
    main()
