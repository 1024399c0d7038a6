"""host/pinflate.hpp -- one zlib stream inflated by several threads (block finder + 16-bit marker decode + window chain) -- against
Python's zlib: streams of every block type (dynamic, fixed, stored, empty sync blocks), every strategy and level, data shaped like a
binary hits file, chunk sizes from a few hundred bytes (many chunks, candidates inside blocks, chunks without a candidate) to larger
than the stream, 1 to 8 threads; damaged, truncated and non-zlib input must be rejected like zlib rejects it."""
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(os.environ.get("MMSEQ_HOST_BIN_DIR", os.path.join(ROOT, "mmseq_amd", "csrc")), "pinflate_test")


def run(tmp_path, comp, threads, chunk):
    src, dst = str(tmp_path / "in.z"), str(tmp_path / "out.bin")
    open(src, "wb").write(comp)
    r = subprocess.run([BIN, src, dst, str(threads), str(chunk)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return r.returncode, (open(dst, "rb").read() if os.path.exists(dst) else b""), r.stderr.decode()


def hits_like(rng, n_records):
    """records of a binary hits file: a short delta-coded name, a u32 count, u32 transcript indices (src/hitsio.cpp:189-213)"""
    out = bytearray()
    t = 0
    for i in range(n_records):
        out += bytes([1, 48 + i % 10])
        k = int(rng.integers(1, 30))
        out += int(k).to_bytes(4, "little")
        t = (t + int(rng.integers(0, 50))) % 200000
        for j in range(k):
            out += int((t + j * int(rng.integers(1, 4))) % 200000).to_bytes(4, "little")
    return bytes(out)


def payloads():
    rng = np.random.default_rng(7)
    text = (b"the quick brown fox jumps over the lazy dog " * 3000) + bytes(rng.integers(97, 123, size=50000, dtype=np.uint8))
    return {
        "hits": hits_like(rng, 40000),
        "text": text,
        "random": bytes(rng.integers(0, 256, size=300000, dtype=np.uint8)),
        "zeros": bytes(400000),
        "mixed": hits_like(rng, 8000) + bytes(rng.integers(0, 256, size=70000, dtype=np.uint8)) + text[:90000] + bytes(50000),
        "tiny": b"abc",
        "empty": b"",
    }


PAYLOADS = payloads()


def streams():
    out = []
    for name, data in PAYLOADS.items():
        for level in (1, 6, 9, 0):
            out.append(("%s-l%d" % (name, level), data, zlib.compress(data, level)))
    for name in ("hits", "text", "mixed"):
        data = PAYLOADS[name]
        for strat, sname in ((zlib.Z_FIXED, "fixed"), (zlib.Z_HUFFMAN_ONLY, "huff"), (zlib.Z_RLE, "rle"), (zlib.Z_FILTERED, "filtered")):
            co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, strat)
            out.append(("%s-%s" % (name, sname), data, co.compress(data) + co.flush()))
        co = zlib.compressobj(1)                       # sync and full flushes in the middle: empty stored blocks, byte-aligned restarts
        parts = [co.compress(data[:len(data) // 3]), co.flush(zlib.Z_SYNC_FLUSH), co.compress(data[len(data) // 3:2 * len(data) // 3]),
                 co.flush(zlib.Z_FULL_FLUSH), co.compress(data[2 * len(data) // 3:]), co.flush()]
        out.append(("%s-flushes" % name, data, b"".join(parts)))
        co = zlib.compressobj(1, zlib.DEFLATED, 15, 1)  # memLevel 1: blocks of 128 symbols, thousands of block boundaries
        out.append(("%s-mem1" % name, data, co.compress(data) + co.flush()))
    return out


STREAMS = streams()


@pytest.mark.parametrize("threads,chunk", [(1, 1 << 30), (3, 300), (8, 1000), (4, 4096), (2, 65536), (8, 20000)])
def test_parallel_inflate_equals_zlib(tmp_path, threads, chunk):
    assert os.path.exists(BIN), "pinflate_test not built (make -C mmseq_amd/csrc)"
    for name, data, comp in STREAMS:
        assert zlib.decompress(comp) == data
        rc, got, err = run(tmp_path, comp, threads, chunk)
        assert rc == 0, (name, threads, chunk, err)
        assert got == data, (name, threads, chunk, len(got), len(data))


def test_parallel_inflate_rejects_what_zlib_rejects(tmp_path):
    rng = np.random.default_rng(3)
    data = PAYLOADS["hits"]
    comp = bytearray(zlib.compress(data, 1))
    for what in ("trailer", "truncated", "body", "body2", "header", "garbage"):
        bad = bytearray(comp)
        if what == "trailer":
            bad[-1] ^= 0x55                                    # Adler-32 of the stream
        elif what == "truncated":
            bad = bad[:len(bad) // 2]
        elif what == "body":
            bad[len(bad) // 2] ^= 0xff
        elif what == "body2":
            for k in rng.integers(100, len(bad) - 100, size=5):
                bad[int(k)] ^= 0x10
        elif what == "header":
            bad[0] = 0x79
        else:
            bad = bytearray(rng.integers(0, 256, size=5000, dtype=np.uint8).tobytes())
        try:
            zlib.decompress(bytes(bad))
            zlib_ok = True
        except zlib.error:
            zlib_ok = False
        for threads, chunk in ((1, 1 << 30), (4, 700), (8, 5000)):
            rc, got, err = run(tmp_path, bytes(bad), threads, chunk)
            if zlib_ok:
                assert rc == 0 and got == zlib.decompress(bytes(bad)), what
            else:
                assert rc == 2 and err.strip(), (what, threads, chunk, rc)


def test_parallel_inflate_on_streams_cut_anywhere(tmp_path):
    """A truncated stream must end in an error wherever it is cut (the bits behind the end read as zeros, which decode as symbols: the
    first version decoded them for ever -- found by tools/pinflate_fuzz.py)."""
    rng = np.random.default_rng(9)
    for name in ("hits", "text", "zeros"):
        comp = zlib.compress(PAYLOADS[name], 1)
        for cut in [2, 3, 7, len(comp) - 1, len(comp) - 4, len(comp) - 5] + [int(x) for x in rng.integers(8, len(comp) - 6, size=12)]:
            for threads, chunk in ((2, 150000), (5, 900)):
                rc, got, err = run(tmp_path, comp[:cut], threads, chunk)
                assert rc == 2 and err.strip(), (name, cut, threads, chunk, rc)
