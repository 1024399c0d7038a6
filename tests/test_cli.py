"""The `mmseq` drop-in CLI (mmseq_amd/csrc/host/mmseq_main.cpp).  CPU: flag surface, exit codes and
error text of src/mmseq.cpp:156-299 and the loud failure without a device.  GPU (-m gpu): a full run on a
small hits file compared, file by file, with the Python restatement of the pipeline (oracle.host_oracle)."""
import gzip
import math
import os
import subprocess

import numpy as np
import pytest

from oracle import host_oracle as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN_DIR = os.environ.get("MMSEQ_HOST_BIN_DIR") or os.path.join(ROOT, "mmseq_amd", "csrc")   # (make -C mmseq_amd/csrc asan: a sanitizer build)
MMSEQ = os.path.join(BIN_DIR, "mmseq")


def run(args, **kw):
    return subprocess.run([MMSEQ] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def dataset(seed=3, n_t=60, n_reads=4000):
    rng = np.random.default_rng(seed)
    names = ["T%07d" % i for i in range(n_t)]
    efflen = {n: float(rng.integers(200, 4000)) for n in names}
    truelen = {n: int(efflen[n]) + 180 for n in names}
    genes, i, g = {}, 0, 0
    while i < n_t:
        sz = int(1 + rng.poisson(1.5))
        if i < 40 < i + sz:
            sz = 40 - i
        if i == 40:
            sz = 4                                   # transcripts 40..43 form one gene that never gets a hit
        genes["G%06d" % g] = names[i:i + sz]
        i += sz; g += 1
    theta = np.exp(rng.normal(0, 1.5, n_t))
    theta[rng.random(n_t) < 0.25] = 0.0            # unobserved transcripts (closed-form rows, simulated traces)
    theta[40:44] = 0.0                               # a whole run of unobserved ones -> (likely) an unobserved gene
    identical = [[names[5], names[6]], [names[41], names[42]]]   # one observed set, one never-hit set
    theta[5] = theta[6] = 3.0
    w = theta * np.array([efflen[n] for n in names])
    w = w / w.sum()
    reads = []
    for r in range(n_reads):
        t0 = int(rng.choice(n_t, p=w))
        d = int(min(5, rng.poisson(1.2)))
        cand = [t for t in range(max(0, t0 - 3), min(n_t, t0 + 4)) if t != t0 and theta[t] > 0]
        extra = list(rng.choice(cand, min(d, len(cand)), replace=False)) if cand and d else []
        ts = sorted(set([t0] + [int(x) for x in extra]))
        if r % 97 == 0 and len(ts) > 1:
            ts = ts + [ts[0]]                        # a within-read duplicate (doublehit, src/mmseq.cpp:404-409)
        reads.append(("r%09d" % r, [names[t] for t in ts]))
    return H.HitsData(names, efflen, truelen, genes, identical, reads)


# ------------------------------------------------------------------------------------------ CPU
def test_help_and_version_exit_1_on_stderr():
    for flag in ("-help", "-h", "--help"):
        r = run([flag])
        assert r.returncode == 1 and b"Usage: mmseq [OPTIONS...] hits_file output_base" in r.stderr and r.stdout == b""
    for flag in ("-version", "-v", "--version"):
        r = run([flag])
        assert r.returncode == 1 and r.stderr.startswith(b"mmseq-")


def test_flag_validation_messages(tmp_path):
    r = run(["-bogus", "a", "b"])
    assert r.returncode == 1 and b"Error: unrecognised option -bogus." in r.stderr
    r = run(["onlyone"])
    assert r.returncode == 1 and b"Error: mandatory arguments missing." in r.stderr
    r = run(["-gibbs_iter", "1000", "-gibbs_ss", "7", "a", "b"])
    assert r.returncode == 1 and b"Error: gibbs_iter must be divisible by gibbs_ss." in r.stderr
    r = run(["-percentiles", "5,120", "a", "b"])
    assert r.returncode == 1 and b"Percentiles must be in (0,100)" in r.stderr
    r = run(["-gibbs_iter", "1000", "-gibbs_ss", "10", "a", "b"])      # 1000 is not a multiple of 1024
    assert r.returncode == 1 and b"gibbs_iter must be a positive multiple of 1024" in r.stderr
    r = run([str(tmp_path / "nope.hits"), str(tmp_path / "out")])
    assert r.returncode == 1 and b"Error reading hits file" in r.stderr


def test_header_consistency_errors(tmp_path):
    p = tmp_path / "h.hits"
    p.write_bytes(b"@TranscriptMetaData\tA\t100\t280\n@TranscriptMetaData\tB\t100\t280\n@GeneIsoforms\tG1\tA\n>r\nA\n")
    r = run([str(p), str(tmp_path / "o")])
    assert r.returncode == 1 and b"Error: B does not belong to a gene in the @GeneIsoforms header entries." in r.stderr
    p.write_bytes(b"@TranscriptMetaData\tA\t100\t280\n@GeneIsoforms\tG1\tA\n@GeneIsoforms\tG2\tA\n>r\nA\n")
    r = run([str(p), str(tmp_path / "o")])
    assert r.returncode == 1 and b"transcripts must be nested within genes" in r.stderr
    p.write_bytes(b"@TranscriptMetaData\tA\t0\t280\n@GeneIsoforms\tG1\tA\n>r\nA\n")
    r = run([str(p), str(tmp_path / "o")])
    assert r.returncode == 1 and b"Error: transcript 'A' has a length of zero." in r.stderr
    # a read that maps to a transcript without a header entry: reported from inside the ingest pipeline's threads, exit code 1
    p.write_bytes(b"@TranscriptMetaData\tA\t100\t280\n@GeneIsoforms\tG1\tA\n>r1\nA\n>r2\nA\nZ\n")
    r = run([str(p), str(tmp_path / "o")], timeout=60)
    assert r.returncode == 1 and b"has no @TranscriptMetaData entry" in r.stderr


def test_without_device_fails_loudly_after_writing_k_and_M(tmp_path):
    from mmseq_amd import gibbs
    if gibbs.device_count() > 0:
        pytest.skip("a HIP device is present")
    h = dataset(n_reads=300)
    p = tmp_path / "x.hits"
    p.write_bytes(H.write_hits_text(h))
    r = run([str(p), str(tmp_path / "out")])
    assert r.returncode == 1 and b"no HIP device available" in r.stderr
    g = H.ingest(h)
    assert (tmp_path / "out.k").read_text().split() == [str(v) for v in g["k"]]     # src/mmseq.cpp:682-684
    lines = (tmp_path / "out.M").read_text().split("\n")
    assert lines[0] == "#" + "".join("\t" + s for s in g["index_sid"])               # :686-689
    exp = ["%d\t%d" % (i, c) for i, r_ in enumerate(g["rows"]) for c in r_]
    assert lines[1:-1] == exp
    assert not (tmp_path / "out.mmseq").exists()


# ------------------------------------------------------------------------------------------ GPU
def _table(path):
    lines = open(path).read().rstrip("\n").split("\n")
    assert lines[0].startswith("# Mapped fragments: ")
    hdr = lines[1].split("\t")
    return int(lines[0].split(": ")[1]), hdr, [dict(zip(hdr, ln.split("\t"))) for ln in lines[2:]]


def _same_number(txt, val, rel=2e-5):
    if isinstance(val, str):
        return txt == val
    if isinstance(val, (int, np.integer)):
        return txt == str(int(val))
    val = float(val)
    if math.isnan(val):
        return "nan" in txt
    if math.isinf(val):
        return txt in ("inf", "-inf") and (txt[0] == "-") == (val < 0)
    got = float(txt)
    return abs(got - val) <= rel * max(abs(val), 1e-300) + 1e-12


def _trace_file(path):
    with gzip.open(path, "rt") as f:
        lines = f.read().split("\n")
    return lines[0].split(" ")[:-1], [ln.split(" ")[:-1] for ln in lines[1:-1]]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["text", "binary"])
def test_full_run_matches_python_pipeline(tmp_path, fmt, gpu):
    h = dataset()
    p = tmp_path / "in.hits"
    p.write_bytes(H.write_hits_text(h) if fmt == "text" else H.write_hits_binary(h))
    out = str(tmp_path / "out")
    r = run(["-gibbs_iter", "2048", "-seed", "77", "-debug", str(p), out], timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    e = H.expected_run(h, seed=77, gibbs_iter=2048)
    g = e["ingest"]
    # .k / .M : exact
    assert open(out + ".k").read().split() == [str(v) for v in e["k"]]
    mlines = open(out + ".M").read().split("\n")
    assert mlines[0] == "#" + "".join("\t" + s for s in g["index_sid"])
    assert mlines[1:-1] == ["%d\t%d" % (i, c) for i, row in enumerate(e["rows"]) for c in row]
    assert open(out + ".doublehits").read().split() == [str(v) for v in g["doublehits"]]
    # the Gibbs trace: every printed value equals the oracle chain's value printed the same way
    ids, rows = _trace_file(out + ".trace_gibbs.gz")
    assert ids == g["index_sid"] and len(rows) == 1024
    exp_rows = [[H.fmt6(v) for v in e["trace"][:, s]] for s in range(1024)]
    assert rows == exp_rows
    ids, rows = _trace_file(out + ".gene.trace_gibbs.gz")
    keepg = [i for i in range(len(e["gene_ids"])) if np.isfinite(np.log(e["t_gene"][i, 0]))]
    assert ids == [e["gene_ids"][i] for i in keepg]
    assert rows[0] == [H.fmt6(e["t_gene"][i, 0]) for i in keepg] and rows[-1] == [H.fmt6(e["t_gene"][i, -1]) for i in keepg]
    ids, rows = _trace_file(out + ".prop.trace_gibbs.gz")
    assert ids == g["index_sid"] and rows[7] == [H.fmt6(v) for v in e["prop"][:, 7]]
    ids, rows = _trace_file(out + ".identical.trace_gibbs.gz")
    assert ids == ["+".join(h.identical[0])] and rows[3] == [H.fmt6(e["t_ident"][0, 3])]
    # the three tables
    mapped, hdr, tab = _table(out + ".mmseq")
    assert mapped == e["mapped"]
    assert hdr == ["feature_id", "log_mu", "sd", "mcse", "iact", "effective_length", "true_length", "unique_hits", "mean_proportion",
                   "mean_probit_proportion", "sd_probit_proportion", "log_mu_em", "observed", "ntranscripts",
                   "percentiles5,25,50,75,95", "percentiles_proportion5,25,50,75,95"]
    assert [t["feature_id"] for t in tab] == h.names                      # header-transcript order (src/mmseq.cpp:1491)
    n_unobs = 0
    for got, exp in zip(tab, e["transcripts"]):
        for col in hdr[1:14]:
            assert _same_number(got[col], exp[col]), (got["feature_id"], col, got[col], exp[col])
        for a, b in zip(got[hdr[14]].split(","), exp["percentiles"]):
            assert _same_number(a, b)
        for a, b in zip(got[hdr[15]].split(","), exp["percentiles_proportion"]):
            assert _same_number(a, b)
        n_unobs += exp["observed"] == 0
    assert n_unobs >= 5
    unobs = [t for t in tab if t["observed"] == "0"][0]
    assert unobs["sd"] == "10.0714" and unobs["mcse"] == "0" and unobs["iact"] == "1" and unobs["log_mu_em"] == "NA"
    mapped, hdr, tab = _table(out + ".identical.mmseq")
    assert hdr[:10] == ["feature_id", "log_mu", "sd", "mcse", "iact", "effective_length", "true_length", "unique_hits", "observed",
                        "ntranscripts"]
    for got, exp in zip(tab, e["identical"]):
        assert got["feature_id"] == exp["feature_id"]
        for col in hdr[1:10]:
            assert _same_number(got[col], exp[col]), (got["feature_id"], col, got[col], exp[col])
        for a, b in zip(got[hdr[10]].split(","), exp["percentiles"]):
            assert _same_number(a, b)
    assert tab[1]["observed"] == "0" and tab[1]["iact"] == "NA" and tab[1][hdr[10]] == "NA,NA,NA,NA,NA"
    mapped, hdr, tab = _table(out + ".gene.mmseq")
    assert hdr[:10] == ["feature_id", "log_mu", "sd", "mcse", "iact", "effective_length", "true_length", "unique_hits", "ntranscripts",
                        "observed"]
    assert [t["feature_id"] for t in tab] == e["gene_ids"]                 # std::map order (src/mmseq.cpp:1630)
    for got, exp in zip(tab, e["genes"]):
        for col in hdr[1:10]:
            assert _same_number(got[col], exp[col]), (got["feature_id"], col, got[col], exp[col])
    assert any(t["observed"] == "0" for t in tab)
    # EM trace (debug): first line = start values, one line per EM iteration
    ids, rows = _trace_file(out + ".trace_em.gz")
    assert len(rows) == e["em_iters"] and rows[0] == [H.fmt6(v) for v in e["mu0"]]


@pytest.mark.gpu
def test_wide_span_reads_do_not_cost_the_stream_kernel(tmp_path, gpu):
    """700 transcripts, reads in random order (first-seen numbering scatters the isoforms), 8 % of the reads hit two
    transcripts hundreds of ids apart, part of the header lists isoforms out of order: the CLI numbers transcripts gene by
    gene in header order and sorts wide rows last, so the
    problem still qualifies for the stream kernel -- and every number equals the Python pipeline's."""
    h = dataset(seed=5, n_t=700, n_reads=6000)
    rng = np.random.default_rng(6)
    glist = [ts for ts in h.genes.values() if len(ts) >= 2 and 100 <= h.names.index(ts[0]) < 600]
    for ga, gb in zip(glist[0::2], glist[1::2]):             # a header that interleaves the isoforms of neighbouring genes
        ia, ib = h.names.index(ga[0]), h.names.index(gb[0])
        if ib == ia + len(ga):
            block = [x for pair in zip(ga, gb) for x in pair] + ga[len(gb):] + gb[len(ga):]
            h.names[ia:ia + len(ga) + len(gb)] = block
    live = sorted({t for _, ts in h.reads for t in ts})
    for r in range(0, len(h.reads), 12):
        a, b = rng.choice(len(live), 2, replace=False)
        if abs(int(a) - int(b)) > 300:
            h.reads[r] = (h.reads[r][0], sorted([live[int(a)], live[int(b)]]))
    p = tmp_path / "in.hits"
    p.write_bytes(H.write_hits_binary(h))
    out = str(tmp_path / "out")
    r = run(["-gibbs_iter", "1024", "-seed", "5", str(p), out], timeout=300, env=dict(os.environ, MMSEQ_TIMING="1"))
    assert r.returncode == 0, r.stderr.decode()
    assert b"[timing] sample kernel 2 " in r.stderr, r.stderr.decode()
    e = H.expected_run(h, seed=5, gibbs_iter=1024)
    ids, rows = _trace_file(out + ".trace_gibbs.gz")
    assert ids == e["ingest"]["index_sid"]
    assert rows == [[H.fmt6(v) for v in e["trace"][:, s]] for s in range(1024)]
    mapped, hdr, tab = _table(out + ".mmseq")
    for got, exp in zip(tab, e["transcripts"]):
        for col in ("log_mu", "sd", "mcse", "iact", "unique_hits", "log_mu_em", "observed", "mean_proportion"):
            assert _same_number(got[col], exp[col]), (got["feature_id"], col, got[col], exp[col])


@pytest.mark.gpu
def test_runs_are_reproducible(tmp_path, gpu):
    h = dataset(seed=8, n_reads=1500)
    p = tmp_path / "in.hits"
    p.write_bytes(H.write_hits_binary(h))
    outs = []
    for i in range(2):
        out = str(tmp_path / ("o%d" % i))
        assert run([str(p), out, ], timeout=300, env=dict(os.environ, OMP_NUM_THREADS=str(1 + 3 * i))).returncode == 0
        outs.append(open(out + ".mmseq").read() + gzip.open(out + ".trace_gibbs.gz", "rt").read())
    assert outs[0] == outs[1]          # independent of host thread count and of run (cf. src/mmseq.cpp:834-838)


def config1_dataset(orc):
    """BASELINE.json configs[0]: 10k reads, 1k transcripts, hits/read = 1+Poisson(3), from the App. D generator."""
    p, aux = orc.synth_problem(R=10000, T=1000, avg_hits=4, seed=1234, sort=False)
    rng = np.random.default_rng(1234)
    names = ["T%07d" % i for i in range(1000)]
    genes, i, g = {}, 0, 0
    while i < 1000:
        sz = int(1 + rng.poisson(3))
        genes["G%06d" % g] = names[i:i + sz]
        i += sz; g += 1
    efflen = {n: float(aux["efflen"][i]) for i, n in enumerate(names)}
    truelen = {n: int(aux["efflen"][i]) + 180 for i, n in enumerate(names)}
    rp = p.row_ptr.astype(np.int64)
    reads = [("r%09d" % r, [names[c] for c in p.col_idx[rp[r]:rp[r + 1]]]) for r in range(10000)]
    return H.HitsData(names, efflen, truelen, genes, [[names[0], names[1]]], reads)


@pytest.mark.gpu
def test_config1_plumbing_text_and_binary_agree(tmp_path, gpu, orc):
    """configs[0] through the full CLI, 1024 iterations, text and binary hits files: identical outputs, and the
    transcript table equals the Python pipeline's."""
    h = config1_dataset(orc)
    (tmp_path / "t.hits").write_bytes(H.write_hits_text(h))
    (tmp_path / "b.hits").write_bytes(H.write_hits_binary(h))
    outs = {}
    for f in ("t", "b"):
        out = str(tmp_path / ("out_" + f))
        r = run(["-gibbs_iter", "1024", str(tmp_path / (f + ".hits")), out], timeout=300, env=dict(os.environ, MMSEQ_TIMING="1"))
        assert r.returncode == 0, r.stderr.decode()
        # the CLI's device layout (header-order transcripts, rows by leading transcript) must qualify for the stream kernel
        assert b"[timing] sample kernel 2 " in r.stderr, r.stderr.decode()
        outs[f] = {ext: open(out + ext, "rb").read() for ext in (".mmseq", ".gene.mmseq", ".identical.mmseq", ".k", ".M")}
        outs[f]["trace"] = gzip.open(out + ".trace_gibbs.gz", "rb").read()
    assert outs["t"] == outs["b"]
    e = H.expected_run(h, gibbs_iter=1024)
    mapped, hdr, tab = _table(str(tmp_path / "out_t") + ".mmseq")
    assert mapped == 10000 and len(tab) == 1000
    for got, exp in zip(tab, e["transcripts"]):
        for col in ("log_mu", "sd", "mcse", "iact", "unique_hits", "log_mu_em", "observed", "mean_proportion"):
            assert _same_number(got[col], exp[col]), (got["feature_id"], col, got[col], exp[col])


@pytest.mark.gpu
def test_chains_flag_pools_moments_and_keeps_chain0_traces(tmp_path, gpu):
    """-chains 4 on one device: chain 0 is the single-chain run (same traces, percentiles, iact), log_mu / sd pool the four
    chains (close to chain 0's within Monte Carlo error, mcse halved), -gpus 1 is accepted, bad combinations are refused."""
    h = dataset(seed=11, n_reads=3000)
    p = tmp_path / "in.hits"
    p.write_bytes(H.write_hits_binary(h))
    o1, o4 = str(tmp_path / "c1"), str(tmp_path / "c4")
    assert run(["-gibbs_iter", "1024", str(p), o1], timeout=300).returncode == 0
    r = run(["-gibbs_iter", "1024", "-chains", "4", "-gpus", "1", str(p), o4], timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert b"chains:        4" in r.stdout
    for ext in (".trace_gibbs.gz", ".gene.trace_gibbs.gz", ".prop.trace_gibbs.gz"):
        assert gzip.open(o1 + ext, "rb").read() == gzip.open(o4 + ext, "rb").read()
    _, ghdr, g1 = _table(o1 + ".gene.mmseq")
    _, _, g4 = _table(o4 + ".gene.mmseq")
    for a, b in zip(g1, g4):                          # gene rows are chain 0's, except the expression-weighted length (pooled log_mu)
        assert all(a[c] == b[c] for c in ghdr if c != "effective_length")
        assert abs(float(a["effective_length"]) - float(b["effective_length"])) <= 0.05 * float(a["effective_length"])
    _, hdr, t1 = _table(o1 + ".mmseq")
    _, _, t4 = _table(o4 + ".mmseq")
    n_obs = 0
    for a, b in zip(t1, t4):
        assert a["feature_id"] == b["feature_id"] and a["iact"] == b["iact"] and a[hdr[14]] == b[hdr[14]]
        if a["observed"] == "1":
            n_obs += 1
            sd = float(a["sd"])
            assert abs(float(a["log_mu"]) - float(b["log_mu"])) <= 6 * max(float(a["mcse"]), 1e-3 * sd) + 1e-6
            assert 0.7 * sd <= float(b["sd"]) <= 1.4 * sd
            assert abs(float(b["mcse"]) - float(a["mcse"]) / 2) <= 1e-4 * float(a["mcse"]) + 1e-12
        else:
            assert a == b
    assert n_obs > 20
    r = run(["-gpus", "2", "-chains", "3", str(p), str(tmp_path / "x")])
    assert r.returncode == 1 and b"chains a multiple of gpus" in r.stderr


@pytest.mark.gpu
def test_synth_hits_writes_the_generator_rows_and_the_cli_runs_on_them(tmp_path, gpu, orc):
    """synth_hits (the benchmark workload as a hits FILE, for end-to-end runs at sizes no alignment here provides): both schemas hold
    the oracle generator's rows in generator order (the writer's transcript-index fast path), 3 % of them with a far hit, and the CLI's
    outputs on that file equal the Python pipeline's."""
    tool = os.path.join(BIN_DIR, "synth_hits")
    R, T = 3000, 500
    p, aux = orc.synth_problem(R=R, T=T, avg_hits=5, seed=1234, sort=False, far_fraction=0.03)
    rp = p.row_ptr.astype(np.int64)
    parsed = []
    for flag, name in ((["-t"], "t.hits"), ([], "b.hits")):
        path = str(tmp_path / name)
        r = subprocess.run([tool] + flag + [str(R), str(T), "5", path, "0.03"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()
        h = H.read_hits(open(path, "rb").read())
        assert h.names == ["T%07d" % i for i in range(T)] and len(h.reads) == R
        assert [rid for rid, _ in h.reads[:3]] == ["r0000000000", "r0000000001", "r0000000002"]
        for i in (0, 1, 17, R - 1):
            assert h.reads[i][1] == ["T%07d" % c for c in p.col_idx[rp[i]:rp[i + 1]]]
        assert sum(len(t) for _, t in h.reads) == p.col_idx.size
        assert all(abs(h.efflen[n] - float(aux["efflen"][j])) <= 1e-5 * float(aux["efflen"][j]) for j, n in enumerate(h.names))
        assert sorted(t for ts in h.genes.values() for t in ts) == h.names and max(len(ts) for ts in h.genes.values()) <= 7
        parsed.append(h)
    assert parsed[0].reads == parsed[1].reads and parsed[0].genes == parsed[1].genes
    out = str(tmp_path / "out")
    r = run(["-gibbs_iter", "1024", str(tmp_path / "b.hits"), out], timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    e = H.expected_run(parsed[1], gibbs_iter=1024)
    ids, rows = _trace_file(out + ".trace_gibbs.gz")
    assert ids == e["ingest"]["index_sid"]
    assert rows == [[H.fmt6(v) for v in e["trace"][:, s]] for s in range(1024)]


@pytest.mark.gpu
def test_two_devices_give_the_output_of_one(tmp_path, gpu):
    """`mmseq -gpus 2` (one chain: the stored rows cut into two read shards by measured cost, EM and Gibbs sharded over RCCL; and with
    -em_one_device the EM on the first device alone) writes the files `mmseq -gpus 1` writes, byte for byte -- integer all-reduces and
    keyed random streams: src/mmseq.cpp:864, :893-899 across devices.  Needs two HIP devices (the boxes of this pool have one: skipped
    there; the same arithmetic runs on one device in tests/test_gpu_parity.py and tests/test_gpu_config4.py)."""
    if gpu.device_count() < 2:
        pytest.skip("needs two HIP devices")
    h = dataset()
    p = tmp_path / "in.hits"
    p.write_bytes(H.write_hits_binary(h))
    outs = {}
    for tag, flags in (("one", []), ("two", ["-gpus", "2"]), ("two_em1", ["-gpus", "2", "-em_one_device"])):
        o = str(tmp_path / tag)
        r = run(["-gibbs_iter", "1024", "-seed", "5"] + flags + [str(p), o], timeout=600)
        assert r.returncode == 0, r.stderr.decode()
        outs[tag] = o
    for ext in (".mmseq", ".identical.mmseq", ".gene.mmseq", ".k", ".M"):
        a = open(outs["one"] + ext, "rb").read()
        assert a == open(outs["two"] + ext, "rb").read(), ext
        assert a == open(outs["two_em1"] + ext, "rb").read(), ext
    for ext in (".trace_gibbs.gz", ".gene.trace_gibbs.gz", ".identical.trace_gibbs.gz", ".prop.trace_gibbs.gz"):
        a = gzip.open(outs["one"] + ext, "rb").read()
        assert a == gzip.open(outs["two"] + ext, "rb").read(), ext
        assert a == gzip.open(outs["two_em1"] + ext, "rb").read(), ext


@pytest.mark.gpu
def test_ingest_through_the_parallel_inflate_writes_the_same_files(tmp_path, gpu):
    """Round 5's ingest: the hits file's zlib stream inflated by several threads (host/pinflate.hpp) and the records parsed in place
    (HitsfileReader::readReadMapRecordsBulk).  Forced here on a small file -- inflate chunks of 600 bytes, so that most records
    straddle a buffer end -- every output must be byte for byte what the one-thread zlib path writes (.k and .M show the row order,
    src/mmseq.cpp:412-418; the tables and traces everything else)."""
    h = dataset(seed=11, n_t=80, n_reads=20000)
    p = tmp_path / "in.hits"
    p.write_bytes(H.write_hits_binary(h))
    outs = {}
    for tag, env in (("plain", {"MMSEQ_INFLATE_THREADS": "1"}),
                     ("parallel", {"MMSEQ_INFLATE_THREADS": "4", "MMSEQ_INFLATE_CHUNK": "600", "MMSEQ_INFLATE_MIN": "0", "MMSEQ_TIMING": "1"})):
        out = str(tmp_path / tag)
        r = run(["-gibbs_iter", "1024", "-debug", str(p), out], env=dict(os.environ, **env), timeout=300)
        assert r.returncode == 0, r.stderr.decode()
        if tag == "parallel":
            assert b"inflated by 4 threads" in r.stderr
        outs[tag] = {f[len(tag):]: open(os.path.join(str(tmp_path), f), "rb").read() for f in sorted(os.listdir(str(tmp_path))) if f.startswith(tag + ".")}
    assert set(outs["plain"]) == set(outs["parallel"]) and ".k" in outs["plain"] and ".M" in outs["plain"] and ".mmseq" in outs["plain"]
    for name in outs["plain"]:
        assert outs["plain"][name] == outs["parallel"][name], name
    g = H.ingest(h)                                      # and both are what the Python restatement of the ingest expects (:682-684)
    assert outs["parallel"][".k"].decode().split() == [str(v) for v in g["k"]]
