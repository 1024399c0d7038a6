"""hits-file codec: the C++ hitsio (mmseq_amd/csrc/host/hitsio.cpp, exercised through the `hitstools`
binary = the reference's only in-tree use of the writer API, src/hitstools.cpp:42-75) against the
Python restatement of the format (oracle/host_oracle.py), both schemas, both directions."""
import os
import subprocess
import zlib

import numpy as np
import pytest

from oracle import host_oracle as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN_DIR = os.environ.get("MMSEQ_HOST_BIN_DIR") or os.path.join(ROOT, "mmseq_amd", "csrc")   # (make -C mmseq_amd/csrc asan: a sanitizer build)
TOOLS = os.path.join(BIN_DIR, "hitstools")


def _dataset(seed=0, n_t=40, n_reads=300, long_ids=False):
    rng = np.random.default_rng(seed)
    names = ["T%07d" % i for i in range(n_t)]
    efflen = {n: float(rng.integers(50, 5000)) + (0.5 if i % 3 == 0 else 0.0) for i, n in enumerate(names)}
    efflen[names[1]] = 1234567.0      # exercises 6-significant-digit formatting (1.23457e+06)
    truelen = {n: int(efflen[n]) + 180 for n in names}
    genes, i, g = {}, 0, 0
    while i < n_t:
        sz = int(1 + rng.poisson(2))
        genes["G%06d" % g] = names[i:i + sz]
        i += sz; g += 1
    identical = [[names[3], names[4]], [names[10], names[11], names[12]]]
    reads = []
    for r in range(n_reads):
        d = int(min(n_t, 1 + rng.poisson(2)))
        ts = sorted(set(rng.choice(n_t, d, replace=False).tolist()))
        rid = ("r%09d" % r) if not long_ids else ("HWI-ST%d:%d:" % (r % 7, r)) + "x" * (300 if r % 50 in (0, 1) else 3) + "/1"
        reads.append((rid, [names[t] for t in ts]))
    reads[5] = ("completely-different", reads[5][1])    # no common prefix/suffix -> plain name in binary
    return H.HitsData(names, efflen, truelen, genes, identical, reads)


def _run(cmd, path):
    return subprocess.run([TOOLS, cmd, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout


@pytest.mark.parametrize("long_ids", [False, True])
def test_text_and_binary_round_trips(tmp_path, long_ids):
    h = _dataset(1, long_ids=long_ids)
    txt, binz = H.write_hits_text(h), H.write_hits_binary(h)
    pt, pb = tmp_path / "a.hits", tmp_path / "b.hits"
    pt.write_bytes(txt); pb.write_bytes(binz)
    # C++ reader on both schemas -> text: must equal the oracle's text byte for byte
    assert _run("t", str(pt)) == txt
    assert _run("t", str(pb)) == txt
    assert _run("inspect", str(pb)) == txt
    # C++ writer, binary: decompressed payload must equal the oracle's encoding byte for byte
    for src in (pt, pb):
        out = _run("b", str(src))
        assert out[:1] == b"\x78"
        assert zlib.decompress(out) == H.encode_hits_binary_payload(h)
    # and the oracle reader agrees with itself across schemas
    a, b = H.read_hits(txt), H.read_hits(binz)
    assert a.reads == b.reads == h.reads and a.names == b.names and a.identical == b.identical
    assert list(a.genes.items()) == list(b.genes.items())


def test_header_only_and_name_delta_small_uint(tmp_path):
    h = _dataset(2, long_ids=True)
    p = tmp_path / "x.hits"
    p.write_bytes(H.write_hits_binary(h))
    hdr = _run("header", str(p)).decode()
    assert hdr.count("@TranscriptMetaData") == 40 and ">" not in hdr
    assert "\t1.23457e+06\t" in hdr
    # names sharing >= 255 characters use the 0xFF + u32 escape (src/hitsio.cpp:36-55)
    payload = H.encode_hits_binary_payload(h)
    assert b"\n\xff" in payload


def test_reader_error_paths(tmp_path):
    p = tmp_path / "bad.hits"
    p.write_bytes(b"@Nonsense\tfoo\n>r1\nT1\n")
    r = subprocess.run([TOOLS, "t", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"does not seem to be a hits file" in r.stderr
    p.write_bytes(b"@TranscriptMetaData\tT1\t100\t280\n@Bogus\tx\n>r1\nT1\n")
    r = subprocess.run([TOOLS, "t", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Hits file looks malformed" in r.stderr
    r = subprocess.run([TOOLS, "t", str(tmp_path / "missing.hits")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Error reading hits file" in r.stderr
    # trailing read id without transcripts: warning, record dropped (src/hitsio.cpp:336-340)
    p.write_bytes(b"@TranscriptMetaData\tT1\t100\t280\n@GeneIsoforms\tG1\tT1\n>r1\nT1\n>r2\n")
    r = subprocess.run([TOOLS, "t", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"Warning: read record without any mapping transcripts" in r.stderr
    assert r.stdout.endswith(b">r1\nT1\n")
    r = subprocess.run([TOOLS], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Usage" in r.stderr


def test_ingest_oracle_first_seen_order():
    h = H.HitsData(["A", "B", "C", "D"], {}, {}, {"G": ["A", "B", "C", "D"]}, [],
                   [("r1", ["C", "A"]), ("r2", ["A", "C"]), ("r3", ["B"]), ("r4", ["C", "C", "A"]), ("r5", ["B", "D"])])
    g = H.ingest(h)
    assert g["index_sid"] == ["C", "A", "B", "D"]             # first-seen transcript order (src/mmseq.cpp:403)
    assert g["rows"] == [(0, 1), (2,), (2, 3)]                 # sorted combos in first-seen order (:412-418)
    assert g["k"].tolist() == [3, 1, 1] and g["mapped"] == 5
    assert g["doublehits"] == [1, 0, 0, 0]                     # within-read duplicate counted (:404-409)


def test_t2g_hits_gene_level_records(tmp_path):
    """src/t2g_hits.cpp:33-121: transcripts -> genes per read, de-duplicated, sorted; records only."""
    h = _dataset(3)
    t2g = {t: g for g, ts in h.genes.items() for t in ts}
    exp = "".join(">%s\n%s" % (rid, "".join(g + "\n" for g in sorted({t2g[t] for t in ts}))) for rid, ts in h.reads).encode()
    exe = os.path.join(BIN_DIR, "t2g_hits")
    for data in (H.write_hits_text(h), H.write_hits_binary(h)):
        p = tmp_path / "in.hits"
        p.write_bytes(data)
        r = subprocess.run([exe, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == exp
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Usage: t2g_hits" in r.stderr


def test_binary_hits_file_read_through_the_parallel_inflate(tmp_path):
    """A binary hits file is ONE zlib stream (src/hitsio.cpp:127).  Large files are inflated by several threads (host/pinflate.hpp:
    block finder, marker decode, window chain; tests/test_pinflate.py holds it to zlib); here the READER is forced onto that path for a
    small file with chunks of a few hundred compressed bytes, and every tool must print what it prints on the one-thread zlib path --
    records, header, gene-level records -- and reject a damaged file with zlib's message class."""
    h = _dataset(4, n_t=60, n_reads=4000, long_ids=True)
    p = tmp_path / "b.hits"
    p.write_bytes(H.write_hits_binary(h))
    forced = dict(os.environ, MMSEQ_INFLATE_THREADS="4", MMSEQ_INFLATE_CHUNK="700", MMSEQ_INFLATE_MIN="0", MMSEQ_TIMING="1")
    plain = dict(os.environ, MMSEQ_INFLATE_THREADS="1")
    for cmd in ("t", "header", "inspect"):
        a = subprocess.run([TOOLS, cmd, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=plain, check=True)
        b = subprocess.run([TOOLS, cmd, str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=forced, check=True)
        assert a.stdout == b.stdout and len(a.stdout) > 1000
        assert b"inflated by 4 threads" in b.stderr and b"inflated by" not in a.stderr
    assert subprocess.run([TOOLS, "t", str(p)], stdout=subprocess.PIPE, env=forced, check=True).stdout == H.write_hits_text(h)
    exe = os.path.join(BIN_DIR, "t2g_hits")
    assert subprocess.run([exe, str(p)], stdout=subprocess.PIPE, env=forced, check=True).stdout == subprocess.run([exe, str(p)], stdout=subprocess.PIPE, env=plain, check=True).stdout
    bad = bytearray(p.read_bytes())
    bad[len(bad) // 2] ^= 0xff
    q = tmp_path / "bad.hits"
    q.write_bytes(bytes(bad))
    for env in (plain, forced):
        r = subprocess.run([TOOLS, "t", str(q)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 1 and b"Error decompressing hits file" in r.stderr


@pytest.mark.parametrize("env", [{"MMSEQ_INFLATE_THREADS": "1"}, {"MMSEQ_INFLATE_THREADS": "3", "MMSEQ_INFLATE_CHUNK": "500", "MMSEQ_INFLATE_MIN": "0"}])
def test_bulk_record_reader_gives_the_hit_sets_of_the_file(tmp_path, env):
    """HitsfileReader::readReadMapRecordsBulk -- what the mmseq CLI reads a file with: whole records parsed in place in the inflated
    buffer, the byte-wise reader for records that straddle a buffer end (with 500-byte inflate chunks: most of them) -- against the
    Python restatement of the format, both schemas, plain and delta-coded names incl. the 0xFF + u32 escape."""
    for seed, long_ids in ((5, False), (6, True)):
        h = _dataset(seed, n_t=50, n_reads=3000, long_ids=long_ids)
        index = {n: i for i, n in enumerate(h.names)}
        want = "".join(" ".join(str(index[t]) for t in ts) + "\n" for _, ts in h.reads).encode()
        for data in (H.write_hits_binary(h), H.write_hits_text(h)):
            p = tmp_path / "in.hits"
            p.write_bytes(data)
            r = subprocess.run([TOOLS, "hitsets", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert r.returncode == 0 and r.stdout == want
