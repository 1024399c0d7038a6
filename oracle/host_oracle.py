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


# ------------------------------------------------------------------------------------ full pipeline
def expected_run(h, alpha=0.1, beta=0.1, seed=1234, gibbs_iter=1024, max_em_iter=1000, epsilon=0.1,
                 percentiles=(5.0, 25.0, 50.0, 75.0, 95.0), trace_len=1024, chain_fn=None):
    """What `mmseq hits out` must produce for hits data `h` (numbers, not text): restates
    src/mmseq.cpp:395-1669 with the keyed-stream chain of oracle/mmseq_oracle.c in the device row order."""
    from . import binding as B
    g = ingest(h)
    n, rows, k = len(g["index_sid"]), g["rows"], g["k"]
    N = g["mapped"]
    sid = g["index_sid"]
    l = np.array([h.efflen[s] * float(N) / 1e9 for s in sid])
    # what the CLI hands to mmg_problem_create: rows in first-seen order, tx_order = (first header index of the gene, own header
    # index).  The library numbers the transcripts by that key and stores the rows in its canonical order (B.canonical_layout on
    # the renumbered rows); a row's hits are walked in ascending DEVICE id, random streams and sums are keyed by the caller's ids.
    hdr_pos = {name: i for i, name in enumerate(h.names)}
    gene_first, gene_of_t = {}, {}
    for gid, ts in h.genes.items():
        gene_first[gid] = min([hdr_pos[t] for t in ts if t in hdr_pos] or [0xffffffff])
        for t in ts:
            gene_of_t[t] = gid
    tkey = [((min(gene_first[gene_of_t[s]], hdr_pos[s]) if s in gene_of_t else hdr_pos[s]) << 32) | hdr_pos[s] for s in sid]
    obs_of_dev = np.argsort(np.array(tkey, np.uint64), kind="stable")
    dev_of_obs = np.empty(n, np.int64)
    dev_of_obs[obs_of_dev] = np.arange(n)
    dev_rows = [sorted(int(dev_of_obs[c]) for c in r) for r in rows]
    rp0 = np.cumsum([0] + [len(r) for r in dev_rows]).astype(np.uint64)
    ci0 = np.array([c for r in dev_rows for c in r], np.uint32)
    rp, ci_dev, k_st, _ = B.canonical_layout(rp0, ci0, k.astype(np.uint32))
    p = B.Problem(rp, obs_of_dev[ci_dev.astype(np.int64)].astype(np.uint32), l, k=k_st)
    # start values (src/mmseq.cpp:617-638): the shares k_i / |row i| of the STORED rows summed exactly, as the CLI takes them from the
    # device (mmg_problem_start_values) -- the reference's floating-point sum in file order differs in the last bits and depends on that
    # order; unique hits (:633) are integers either way
    mu0 = B.start_values_exact(p)
    uh = np.zeros(n, np.int64)
    for r, kk in zip(rows, k):
        if len(r) == 1:
            uh[r[0]] += kk
    mu_em, em_iters, ll = B.em(p, mu0, max_iter=max_em_iter, epsilon=epsilon)
    # chain_fn: another engine for src/mmseq.cpp:833-918 (tests/test_compare_outputs.py: a plain numpy Gibbs sampler, with the right and
    # with deliberately wrong weights) -- called as chain_fn(p, mu_em, alpha, beta, seed, gibbs_iter, trace_len) -> {"trace": [n, trace_len]}
    chain = (chain_fn(p, mu_em, alpha, beta, seed, gibbs_iter, trace_len) if chain_fn
             else B.gibbs_keyed(p, mu_em, alpha=alpha, beta=beta, seed=seed, n_iter=gibbs_iter, trace_len=trace_len))
    trace = chain["trace"]                                   # [n, trace_len], real scale, observed (first-seen) order
    hdr_index = {name: i for i, name in enumerate(h.names)}
    genes = list(h.genes.items())
    # identical / gene sums, simulated traces for isoforms without hits (:927-1008)
    t_ident = np.zeros((len(h.identical), trace_len))
    for v, s in enumerate(h.identical):
        for name in s:
            if name in g["sid_index"]:
                t_ident[v] += trace[g["sid_index"][name]]
    t_gene = np.zeros((len(genes), trace_len))
    simu = {}
    for gi, (gid, ts) in enumerate(genes):
        for name in ts:
            if name in g["sid_index"]:
                t_gene[gi] += trace[g["sid_index"][name]]
            else:
                simu[name] = B.simu_gamma_trace(seed, hdr_index[name], alpha, 1.0 / (beta + h.efflen[name] * float(N) / 1e9), trace_len)
                t_gene[gi] += simu[name]
    gene_of = {name: gi for gi, (gid, ts) in enumerate(genes) for name in ts}
    prop = np.full((n, trace_len), np.nan)
    prop_simu = {}
    for gi, (gid, ts) in enumerate(genes):
        for name in ts:
            if name in g["sid_index"]:
                prop[g["sid_index"][name]] = trace[g["sid_index"][name]] / t_gene[gi]
            else:
                prop_simu[name] = simu[name] / t_gene[gi]
    pind = percentile_indices(percentiles, trace_len)

    def pct(tr):
        s = np.sort(tr)
        return [s[i] for i in pind]

    def sok(logtr):
        rc, var, tau, m = B.sokal(logtr)
        if rc != 0:
            return np.sqrt(var), float(trace_len), float("nan")
        return np.sqrt(var), np.sqrt(tau * var / trace_len), tau

    def prop_summ(pt, multi):
        with np.errstate(invalid="ignore"):
            if multi:
                z = probit(np.minimum(np.maximum(pt, 1e-9), 1 - 1e-9))
            else:
                z = np.full(trace_len, np.inf)
            s1, s2 = z.sum(), (z * z).sum()
            return pt.mean(), s1 / trace_len, np.sqrt((s2 - s1 * s1 / trace_len) / (trace_len - 1.0))

    dig, sqtri = digamma(alpha), math.sqrt(trigamma(alpha))
    prior = lambda name: dig - math.log(beta + h.efflen[name] * float(N) / 1e9)
    with np.errstate(divide="ignore", invalid="ignore"):
        ltrace, lident, lgene = np.log(trace), np.log(t_ident), np.log(t_gene)
    tx = []
    for name in h.names:
        ntx = len(h.genes[[gid for gid, ts in genes if name in ts][0]])
        if name in g["sid_index"]:
            t = g["sid_index"][name]
            sd, mcse, iact = sok(ltrace[t])
            mp, mpp, sdpp = prop_summ(prop[t], ntx > 1)
            tx.append(dict(feature_id=name, log_mu=ltrace[t].mean(), sd=sd, mcse=mcse, iact=iact, effective_length=h.efflen[name],
                           true_length=h.truelen[name], unique_hits=int(uh[t]), mean_proportion=mp, mean_probit_proportion=mpp,
                           sd_probit_proportion=sdpp, log_mu_em=math.log(mu_em[t]) if mu_em[t] > 0 else -math.inf, observed=1,
                           ntranscripts=ntx, percentiles=pct(trace[t]), percentiles_proportion=pct(prop[t])))
        else:
            mp, mpp, sdpp = prop_summ(prop_simu[name], ntx > 1)
            tx.append(dict(feature_id=name, log_mu=prior(name), sd=sqtri, mcse=0.0, iact=1.0, effective_length=h.efflen[name],
                           true_length=h.truelen[name], unique_hits=0, mean_proportion=mp, mean_probit_proportion=mpp,
                           sd_probit_proportion=sdpp, log_mu_em="NA", observed=0, ntranscripts=ntx,
                           percentiles=pct(simu[name]), percentiles_proportion=pct(prop_simu[name])))
    # unique hits to groups: src/uh.cpp literal semantics via the C oracle
    pf = B.Problem(np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64), np.array([c for r in rows for c in r], np.uint32), l,
                   k=k.astype(np.uint32))
    mem_i = np.zeros((n, max(len(h.identical), 1)), np.uint8)
    for v, s in enumerate(h.identical):
        for name in s:
            if name in g["sid_index"]:
                mem_i[g["sid_index"][name], v] = 1
    mem_g = np.zeros((n, len(genes)), np.uint8)
    for gi, (gid, ts) in enumerate(genes):
        for name in ts:
            if name in g["sid_index"]:
                mem_g[g["sid_index"][name], gi] = 1
    uh_i = B.uh(pf, mem_i) if h.identical else []
    uh_g = B.uh(pf, mem_g)
    ident = []
    for v, s in enumerate(h.identical):
        fid = "+".join(s)
        m_ = lident[v].mean()
        if np.isfinite(m_):
            sd, mcse, iact = sok(lident[v])
            ident.append(dict(feature_id=fid, log_mu=m_, sd=sd, mcse=mcse, iact=iact, effective_length=h.efflen[s[0]],
                              true_length=h.truelen[s[0]], unique_hits=int(uh_i[v]), observed=1, ntranscripts=len(s),
                              percentiles=pct(t_ident[v])))
        else:
            ident.append(dict(feature_id=fid, log_mu=math.log(len(s)) + prior(s[-1]), sd=sqtri, mcse=0.0, iact="NA",
                              effective_length=h.efflen[s[0]], true_length=h.truelen[s[0]], unique_hits=0, observed=0,
                              ntranscripts=len(s), percentiles=["NA"] * len(pind)))
    gene = []
    for gi, (gid, ts) in enumerate(genes):
        sd, mcse, iact = sok(lgene[gi])
        obs = any(name in g["sid_index"] for name in ts)
        w = np.array([math.exp(ltrace[g["sid_index"][nm]].mean()) if nm in g["sid_index"] else math.exp(prior(nm)) for nm in ts])
        glen = float((np.array([h.efflen[nm] for nm in ts]) * w).sum() / w.sum())
        gene.append(dict(feature_id=gid, log_mu=lgene[gi].mean(), sd=sd, mcse=mcse if obs else sd / math.sqrt(trace_len),
                         iact=iact if obs else 1.0, effective_length=glen, true_length="NA", unique_hits=int(uh_g[gi]) if obs else 0,
                         ntranscripts=len(ts), observed=1 if obs else 0, percentiles=pct(t_gene[gi])))
    return dict(ingest=g, rows=rows, k=k, mapped=N, l=l, mu0=mu0, mu_em=mu_em, em_iters=em_iters, trace=trace, t_ident=t_ident,
                t_gene=t_gene, prop=prop, transcripts=tx, identical=ident, genes=gene, gene_ids=[gid for gid, _ in genes])
