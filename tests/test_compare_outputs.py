"""tools/compare_outputs.py: the file-level check a maintainer with a reference `mmseq` binary runs (SURVEY App. E.1 exact, App. E.3
statistical with DESIGN section 6's dated deviations).  No reference binary exists in this image (src/Makefile:16 needs GSL and Boost),
so the tool is tested on outputs this repo can make: CPU -- the Python restatement of the pipeline (oracle.host_oracle.expected_run)
written to files, two seeds of the keyed oracle chain against each other and against a plain numpy Gibbs sampler (another engine, other
random numbers: must pass), and two negatives (a numpy sampler that weights the allocation by mu * l, and a shifted unique_hits: must
fail); GPU -- two seeds of the drop-in CLI."""
import gzip
import os
import sys

import numpy as np
import pytest

from oracle import host_oracle as H

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import compare_outputs as CO  # noqa: E402
from test_cli import dataset, run  # noqa: E402


def _cells(vals):
    return ",".join("NA" if isinstance(v, str) else H.fmt6(v) for v in vals)


def _f(v):
    return v if isinstance(v, str) else H.fmt6(v)


def write_outputs(base, e, h):
    """The files of src/mmseq.cpp:1675-1685 from expected_run's numbers (App. B.3-B.6 of the survey)."""
    g = e["ingest"]
    with open(base + ".k", "w") as f:
        f.write("".join("%d\n" % v for v in e["k"]))
    with open(base + ".M", "w") as f:
        f.write("#" + "".join("\t" + s for s in g["index_sid"]) + "\n")
        f.write("".join("%d\t%d\n" % (i, c) for i, row in enumerate(e["rows"]) for c in row))
    pl = "5,25,50,75,95"
    with open(base + ".mmseq", "w") as f:
        f.write("# Mapped fragments: %d\n" % e["mapped"])
        cols = ["feature_id", "log_mu", "sd", "mcse", "iact", "effective_length", "true_length", "unique_hits", "mean_proportion",
                "mean_probit_proportion", "sd_probit_proportion", "log_mu_em", "observed", "ntranscripts"]
        f.write("\t".join(cols + ["percentiles" + pl, "percentiles_proportion" + pl]) + "\n")
        for t in e["transcripts"]:
            f.write("\t".join([_f(t[c]) for c in cols] + [_cells(t["percentiles"]), _cells(t["percentiles_proportion"])]) + "\n")
    with open(base + ".identical.mmseq", "w") as f:
        f.write("# Mapped fragments: %d\n" % e["mapped"])
        cols = ["feature_id", "log_mu", "sd", "mcse", "iact", "effective_length", "true_length", "unique_hits", "observed", "ntranscripts"]
        f.write("\t".join(cols + ["percentiles" + pl]) + "\n")
        for t in e["identical"]:
            f.write("\t".join([_f(t[c]) for c in cols] + [_cells(t["percentiles"])]) + "\n")
    with open(base + ".gene.mmseq", "w") as f:
        f.write("# Mapped fragments: %d\n" % e["mapped"])
        cols = ["feature_id", "log_mu", "sd", "mcse", "iact", "effective_length", "true_length", "unique_hits", "ntranscripts", "observed"]
        f.write("\t".join(cols + ["percentiles" + pl]) + "\n")
        for t in e["genes"]:
            f.write("\t".join([_f(t[c]) for c in cols] + [_cells(t["percentiles"])]) + "\n")

    def trace_file(path, ids, tr):
        with gzip.open(path, "wt") as f:
            f.write("".join(i + " " for i in ids) + "\n")
            for s in range(tr.shape[1]):
                f.write("".join(H.fmt6(v) + " " for v in tr[:, s]) + "\n")
    trace_file(base + ".trace_gibbs.gz", g["index_sid"], e["trace"])
    keep = [i for i in range(len(e["gene_ids"])) if np.isfinite(np.log(e["t_gene"][i, 0]))]
    trace_file(base + ".gene.trace_gibbs.gz", [e["gene_ids"][i] for i in keep], e["t_gene"][keep])


def numpy_gibbs(wrong_weights=False):
    """src/mmseq.cpp:851-918 in plain numpy: per row Multinomial(k_i; weights mu[cols] -- or, wrong on purpose, mu[cols] * l[cols]),
    column sums, Gamma(alpha + cnt, 1 / (beta + l)).  numpy's generator: nothing of the keyed spec in it."""
    def fn(p, mu0, alpha, beta, seed, n_iter, trace_len):
        rng = np.random.default_rng(seed + 1000003)
        rp = p.row_ptr.astype(np.int64)
        L = np.diff(rp)
        k = np.ones(p.m, np.int64) if p.k is None else p.k.astype(np.int64)
        rid = np.repeat(np.arange(p.m), L)
        pos = np.arange(rp[-1]) - rp[:-1][rid]
        cols = np.zeros((p.m, int(L.max())), np.int64)
        mask = np.zeros(cols.shape, bool)
        cols[rid, pos] = p.col_idx
        mask[rid, pos] = True
        mu = np.maximum(mu0.copy(), 1e-300)
        ss = n_iter // trace_len
        trace = np.empty((p.n, trace_len))
        for it in range(n_iter):
            w = np.where(mask, (mu * p.l if wrong_weights else mu)[cols], 0.0)
            w = w / w.sum(axis=1, keepdims=True)
            x = rng.multinomial(k, w)
            cnt = np.bincount(cols[mask], weights=x[mask], minlength=p.n)
            mu = rng.gamma(alpha + cnt, 1.0 / (beta + p.l))
            if it % ss == 0:
                trace[:, it // ss] = mu
        return {"trace": trace}
    return fn


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    d = tmp_path_factory.mktemp("cmp")
    h = dataset(seed=11, n_t=500, n_reads=60000)
    out = {}
    for name, kw in (("keyed1", dict(seed=1)), ("keyed2", dict(seed=2)), ("numpy", dict(seed=3, chain_fn=numpy_gibbs())),
                     ("wrong", dict(seed=4, chain_fn=numpy_gibbs(wrong_weights=True)))):
        e = H.expected_run(h, **kw)
        write_outputs(str(d / name), e, h)
        out[name] = str(d / name)
    return out


def test_two_seeds_of_the_keyed_chain_pass(runs, capsys):
    assert CO.main([runs["keyed1"], runs["keyed2"]]) == 0
    txt = capsys.readouterr().out
    assert "FAIL" not in txt and txt.count("PASS") >= 30 and "E3 genes" in txt and "S  transcripts: table of A" in txt


def test_another_engine_with_the_right_weights_passes(runs, capsys):
    """numpy's multinomial and gamma against the keyed Philox chain: the same model, nothing else in common"""
    assert CO.main([runs["keyed1"], runs["numpy"]]) == 0, capsys.readouterr().out


def test_allocation_by_mu_times_length_fails_the_statistical_clauses(runs, capsys):
    """SURVEY App. E.2-4's wrong-weight detector at file level: reads shared between isoforms of different length split by mu, not mu * l"""
    assert CO.main([runs["keyed1"], runs["wrong"]]) == 1
    txt = capsys.readouterr().out
    failed = [ln for ln in txt.split("\n") if ln.startswith("FAIL")]
    assert failed and all(ln.split()[1] == "E3" for ln in failed), txt     # everything exact still agrees: only E.3 sees it


def test_a_shifted_unique_hits_fails_the_exact_clauses(runs, tmp_path, capsys):
    import shutil
    for ext in (".mmseq", ".identical.mmseq", ".gene.mmseq", ".k", ".M", ".trace_gibbs.gz", ".gene.trace_gibbs.gz"):
        shutil.copy(runs["keyed2"] + ext, str(tmp_path / "x") + ext)
    lines = open(str(tmp_path / "x") + ".mmseq").read().split("\n")
    cells = lines[5].split("\t")
    cells[7] = str(int(cells[7]) + 1)                                      # unique_hits of one transcript
    lines[5] = "\t".join(cells)
    open(str(tmp_path / "x") + ".mmseq", "w").write("\n".join(lines))
    assert CO.main([runs["keyed1"], str(tmp_path / "x")]) == 1
    txt = capsys.readouterr().out
    assert [ln.split()[1:4] for ln in txt.split("\n") if ln.startswith("FAIL")] == [["E1", "transcripts:", "unique_hits"]]


@pytest.mark.gpu
def test_two_seeds_of_the_cli_pass(tmp_path, gpu, capsys):
    h = dataset(seed=11, n_t=500, n_reads=60000)
    p = tmp_path / "in.hits"
    p.write_bytes(H.write_hits_binary(h))
    for seed in (1, 2):
        r = run(["-seed", str(seed), str(p), str(tmp_path / ("s%d" % seed))], timeout=300)
        assert r.returncode == 0, r.stderr.decode()
    assert CO.main([str(tmp_path / "s1"), str(tmp_path / "s2")]) == 0, capsys.readouterr().out
