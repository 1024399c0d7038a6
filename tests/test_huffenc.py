"""The trace files' Huffman-only deflate writer (mmseq_amd/csrc/host/huffenc.hpp; src/mmseq.cpp:911-917 writes the traces through a gzip
filter): what it writes must read back with a stock inflate.  huffenc_test encodes a file piece by piece into one gzip member; Python's
gzip is the judge.  CPU only."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(os.environ.get("MMSEQ_HOST_BIN_DIR") or os.path.join(ROOT, "mmseq_amd", "csrc"), "huffenc_test")


def _round_trip(tmp_path, data, piece):
    src, dst = tmp_path / "in.bin", tmp_path / "out.gz"
    src.write_bytes(data)
    r = subprocess.run([BIN, str(src), str(dst), str(piece)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert gzip.open(dst, "rb").read() == data
    return r.stdout


@pytest.mark.skipif(not os.path.exists(BIN), reason="host tools not built")
@pytest.mark.parametrize("piece", [290000, 4096, 7, 1])
def test_trace_text_reads_back(tmp_path, piece):
    rng = np.random.default_rng(piece)
    x = np.exp(rng.normal(-8, 3, size=40000 if piece > 100 else 300))
    x[::97] = 0.0
    x[5::131] = np.nan
    text = "\n".join(" ".join("%g" % v for v in row) + " " for row in x.reshape(-1, 100)) + "\n"
    out = _round_trip(tmp_path, text.encode(), piece)
    assert "(0 pieces left to zlib)" in out


@pytest.mark.skipif(not os.path.exists(BIN), reason="host tools not built")
def test_odd_inputs_read_back(tmp_path):
    rng = np.random.default_rng(3)
    _round_trip(tmp_path, b"", 1000)                                        # nothing: only the member's frame
    _round_trip(tmp_path, b"a" * 100001, 1000)                              # one symbol: a one-bit code next to the end-of-block code
    _round_trip(tmp_path, bytes(range(256)) * 41, 5000)                      # every byte value
    _round_trip(tmp_path, rng.integers(0, 256, 70001, dtype=np.uint8).tobytes(), 65536)
    # frequencies 1, 2, 4, ... 2^18: an optimal code 19 bits deep, more than deflate's 15 -- the piece goes to zlib, and still reads back
    skew = b"".join(bytes([65 + i]) * (1 << i) for i in range(19))
    out = _round_trip(tmp_path, skew, len(skew))
    assert "(1 pieces left to zlib)" in out
