"""numpy / pure-Python restatements of the HOST-side pieces around the Gibbs loop -- TEST
INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference lines it follows
(paths relative to /root/reference).

  hits-file codec      src/hitsio.cpp:22-115 (encodings), :162-248 (writer), :250-447 (reader)
  ingest / collapse    src/mmseq.cpp:395-441
  summaries            src/mmseq.cpp:927-1008, :1110-1395
  table writers        src/mmseq.cpp:1469-1669
"""
import math
import struct
import zlib
from collections import OrderedDict

import numpy as np

MMSEQ_HEADER = b"MMSEQ_HITSFILE"


def fmt6(x):
    """C++ default ostream formatting of a double / int (6 significant digits, %g)."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    x = float(x)
    if math.isnan(x):
        return "nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return "%g" % x


# ------------------------------------------------------------------------------------ hits files
class HitsData:
    def __init__(self, names, efflen, truelen, genes, identical, reads):
        self.names = list(names)              # header order
        self.efflen = dict(efflen)
        self.truelen = dict(truelen)
        self.genes = OrderedDict(sorted(genes.items()))   # std::map order (byte-wise)
        self.identical = [list(s) for s in identical]
        self.reads = [(rid, list(t)) for rid, t in reads]


def write_hits_text(h):
    """Schema 0, src/hitsio.cpp:162-187."""
    out = []
    for n in h.names:
        out.append("@TranscriptMetaData\t%s\t%s\t%d\n" % (n, fmt6(h.efflen[n]), h.truelen[n]))
    for g, ts in h.genes.items():
        out.append("@GeneIsoforms\t%s%s\n" % (g, "".join("\t" + t for t in ts)))
    for s in h.identical:
        out.append("@IdenticalTranscripts%s\n" % "".join("\t" + t for t in s))
    for rid, ts in h.reads:
        out.append(">%s\n" % rid)
        out.extend(t + "\n" for t in ts)
    return "".join(out).encode()


def _small(v):
    return bytes([v]) if v < 255 else b"\xff" + struct.pack("<I", v)


def encode_hits_binary_payload(h):
    """Decompressed schema-1 byte stream, src/hitsio.cpp:189-213, :232-240, delta coding :77-100."""
    b = bytearray()
    b += MMSEQ_HEADER + b"\n" + struct.pack("<I", 1)
    b += struct.pack("<I", len(h.names))
    index = {}
    for n in h.names:
        index.setdefault(n, len(index))
        b += n.encode() + b"\n" + fmt6(h.efflen[n]).encode() + b"\n" + struct.pack("<I", h.truelen[n] & 0xffffffff)
    b += struct.pack("<I", len(h.genes))
    for g, ts in h.genes.items():
        b += g.encode() + b"\n" + struct.pack("<I", len(ts))
        for t in ts:
            b += t.encode() + b"\n"
    b += struct.pack("<I", len(h.identical))
    for s in h.identical:
        b += struct.pack("<I", len(s))
        for t in s:
            b += t.encode() + b"\n"
    prev = ""
    for rid, ts in h.reads:
        nb = 0
        lim = min(len(prev), len(rid))
        while nb < lim and prev[nb] == rid[nb]:
            nb += 1
        ne = 0
        while nb + ne < lim and prev[len(prev) - 1 - ne] == rid[len(rid) - 1 - ne]:
            ne += 1
        if nb == 0 and ne == 0:
            b += rid.encode() + b"\n"
        else:
            b += b"\n" + _small(nb) + rid[nb:len(rid) - ne].encode() + b"\n" + _small(ne)
        prev = rid
        b += struct.pack("<I", len(ts))
        for t in ts:
            b += struct.pack("<I", index[t])
    return bytes(b)


def write_hits_binary(h):
    return zlib.compress(encode_hits_binary_payload(h), 1)   # zlib (RFC 1950) stream, best_speed


def read_hits(data):
    """Parses either schema from bytes; returns HitsData (src/hitsio.cpp:250-447)."""
    if data[:1] == b"\x78":
        data = zlib.decompress(data)
    if data.startswith(MMSEQ_HEADER + b"\n"):
        return _read_binary(data)
    return _read_text(data)


def _read_text(data):
    names, efflen, truelen, genes, identical, reads = [], {}, {}, {}, [], []
    lines = data.decode().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    i = 0
    while i < len(lines) and not lines[i].startswith(">"):
        tok = lines[i].split()
        if tok and tok[0] == "@TranscriptMetaData":
            names.append(tok[1]); efflen.setdefault(tok[1], float(tok[2])); truelen.setdefault(tok[1], int(tok[3]))
        elif tok and tok[0] == "@GeneIsoforms":
            genes.setdefault(tok[1], tok[2:])
        elif tok and tok[0] == "@IdenticalTranscripts":
            identical.append(tok[1:])
        else:
            raise ValueError("Hits file looks malformed.")
        i += 1
    while i < len(lines):
        rid = lines[i][1:]
        i += 1
        if i >= len(lines):
            break   # trailing read id without transcripts: warning + stop (src/hitsio.cpp:336-340)
        ts = []
        while i < len(lines) and not lines[i].startswith(">"):
            ts.append(lines[i]); i += 1
        reads.append((rid, ts))
    return HitsData(names, efflen, truelen, genes, identical, reads)


def _read_binary(d):
    pos = [0]

    def line():
        j = d.index(b"\n", pos[0])
        s = d[pos[0]:j].decode()
        pos[0] = j + 1
        return s

    def u32():
        v = struct.unpack_from("<I", d, pos[0])[0]
        pos[0] += 4
        return v

    def small():
        v = d[pos[0]]
        pos[0] += 1
        return u32() if v == 255 else v

    line(); u32()
    names, efflen, truelen, genes, identical, reads = [], {}, {}, {}, [], []
    for _ in range(u32()):
        n = line(); e = line(); t = u32()
        names.append(n); efflen.setdefault(n, float(e)); truelen.setdefault(n, t)
    for _ in range(u32()):
        g = line()
        genes.setdefault(g, [line() for _ in range(u32())])
    for _ in range(u32()):
        identical.append([line() for _ in range(u32())])
    prev = ""
    while pos[0] < len(d):
        s = line()
        if s == "":
            nb = small(); mid = line(); ne = small()
            s = prev[:nb] + mid + prev[len(prev) - ne:]
        prev = s
        cnt = u32()
        reads.append((s, [names[u32()] for _ in range(cnt)]))
    return HitsData(names, efflen, truelen, genes, identical, reads)


# ------------------------------------------------------------------------------------ ingest
def ingest(h):
    """src/mmseq.cpp:395-441: transcript index = first-seen order (:403); within-read duplicates dropped
    and counted (:404-409); comb sorted (:412); row index = first-seen order of the set (:417); k[row]++ (:440).
    Returns dict(sid_index, index_sid, rows (list of tuples), k, doublehits, mapped)."""
    sid_index, index_sid, comb_index, rows, k, doublehits = {}, [], {}, [], [], []
    mapped = 0
    for rid, ts in h.reads:
        mapped += 1
        comb = []
        for t in ts:
            if t not in sid_index:
                sid_index[t] = len(index_sid); index_sid.append(t); doublehits.append(0)
            i = sid_index[t]
            if i not in comb:
                comb.append(i)
            else:
                doublehits[i] += 1
        comb = tuple(sorted(comb))
        if comb not in comb_index:
            comb_index[comb] = len(rows); rows.append(comb); k.append(0)
        k[comb_index[comb]] += 1
    return dict(sid_index=sid_index, index_sid=index_sid, rows=rows, k=np.array(k, np.int64),
                doublehits=doublehits, mapped=mapped)


def device_row_order(rows):
    """The CLI's device layout: rows stably sorted by (leading transcript, length); empty rows first."""
    first = np.array([r[0] if len(r) else -1 for r in rows], np.int64)
    lens = np.array([len(r) for r in rows], np.int64)
    return np.lexsort((lens, first))


# ------------------------------------------------------------------------------------ summaries
def c_round(x):
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def percentile_indices(percentiles, trace_len=1024):
    """src/mmseq.cpp:1111-1114."""
    return [c_round(p / 100.0 * (trace_len - 1)) for p in percentiles]


def probit(p):
    from scipy.special import ndtri
    return ndtri(p)


def digamma(x):
    from scipy.special import psi
    return float(psi(x))


def trigamma(x):
    from scipy.special import polygamma
    return float(polygamma(1, x))
