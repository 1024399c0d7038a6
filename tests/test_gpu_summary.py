"""The posterior summary computed on the device (mmg_summary_*, src/mmseq.cpp:927-1363) against numpy and the oracle's Sokal
(which is pinned to the reference's own sokal.cc): trace sums, proportions, simulated traces and percentiles bit for bit (they
are sums / quotients in the reference's order and order statistics of the trace itself), log means and Sokal to the rounding of
log(), probit summaries to the accuracy of AS 241."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(gpu, orc, trace_len, tx_order=False, seed=5):
    p, _ = orc.synth_problem(R=30000, T=900, avg_hits=5, seed=seed, sort=False)
    n = p.n
    rng = np.random.default_rng(seed)
    txo = None
    if tx_order:
        txo = (rng.permutation(n).astype(np.uint64) // np.uint64(3)) << np.uint64(32)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, tx_order=txo)
    mu0, _ = prob.start_values()
    s = gpu.Sampler(prob, mu0, seed=31, n_chains=2, gibbs_iter=2 * trace_len, trace_len=trace_len)
    s.run(2 * trace_len)
    return prob, s, n, rng


@pytest.mark.parametrize("tx_order", [False, True])
def test_device_summary_matches_numpy_and_the_pinned_sokal(gpu, orc, tx_order):
    S = 1024
    prob, s, n, rng = _setup(gpu, orc, S, tx_order)
    chain = 1
    trace = s.trace(chain)                                      # [n, S], caller's numbering
    # 40 isoforms without hits, genes of 1..5 members mixing observed and virtual transcripts, a few identical sets
    nv = 40
    vid = (10_000 + np.arange(nv) * 7).astype(np.uint64)
    vscale = rng.uniform(0.01, 2.0, nv)
    members = rng.permutation(n + nv)
    genes, i = [], 0
    while i < members.size:
        sz = int(rng.integers(1, 6))
        genes.append([int(m) for m in members[i:i + sz]])
        i += sz
    identical = [[3, 4], [10, 11, 12], [700]]
    pidx = [int(np.floor(abs(q / 100.0 * (S - 1)) + 0.5)) for q in (5, 25, 50, 75, 95)]   # src/mmseq.cpp:1113
    q = gpu.Summary(s, chain=chain, virtual_id=vid, virtual_scale=vscale, identical=identical, genes=genes, percentile_index=pidx)

    # simulated traces: the keyed draws of the oracle (src/mmseq.cpp:971-978)
    V = np.stack([orc.simu_gamma_trace(31, int(vid[v]), 0.1, vscale[v], S) for v in range(nv)])
    full = np.concatenate([trace, V])                            # member index -> trace
    t_gene = np.zeros((len(genes), S))
    for g, ms in enumerate(genes):
        for m in ms:
            t_gene[g] += full[m]                                 # same order of additions as :947-1008
    t_ident = np.zeros((len(identical), S))
    for g, ms in enumerate(identical):
        for m in ms:
            t_ident[g] += full[m]
    gene_of = np.empty(n + nv, np.int64)
    for g, ms in enumerate(genes):
        gene_of[ms] = g
    prop = full / t_gene[gene_of]
    assert np.array_equal(q.rows(gpu.SERIES_GENE).T, t_gene)
    assert np.array_equal(q.rows(gpu.SERIES_IDENTICAL).T, t_ident)
    assert np.array_equal(q.rows(gpu.SERIES_TRANSCRIPT).T, prop[:n])
    assert np.array_equal(q.rows(gpu.SERIES_GENE, 17, 5), t_gene[:, 17:22].T)

    def check_series(kind, tr):
        r = q.series(kind)
        assert np.array_equal(r["percentiles"], np.sort(tr, axis=1)[:, pidx])
        with np.errstate(divide="ignore"):
            lt = np.log(tr)
        np.testing.assert_allclose(r["log_mean"], lt.mean(axis=1), rtol=1e-12, atol=1e-12)
        assert (r["rc"] == 0).all()
        for i in range(0, tr.shape[0], max(1, tr.shape[0] // 150)):
            rc, var, tau, m = orc.sokal(lt[i])                   # pinned to the reference's compiled sokal.cc (test_oracle_sokal.py)
            assert rc == 0
            np.testing.assert_allclose([r["var"][i], r["tau"][i]], [var, tau], rtol=1e-9)

    check_series(gpu.SERIES_TRANSCRIPT, trace)
    check_series(gpu.SERIES_VIRTUAL, V)
    check_series(gpu.SERIES_IDENTICAL, t_ident)
    check_series(gpu.SERIES_GENE, t_gene)

    from scipy.special import ndtri
    multi = np.array([len(genes[g]) > 1 for g in gene_of])
    for kind, pr, mu in ((gpu.SERIES_TRANSCRIPT, prop[:n], multi[:n]), (gpu.SERIES_VIRTUAL, prop[n:], multi[n:])):
        r = q.proportions(kind)
        assert np.array_equal(r["percentiles"], np.sort(pr, axis=1)[:, pidx])
        np.testing.assert_allclose(r["mean"], pr.mean(axis=1), rtol=1e-12)
        z = ndtri(np.clip(pr, 1e-9, 1 - 1e-9))
        s1, s2 = z.sum(axis=1), (z * z).sum(axis=1)
        np.testing.assert_allclose(r["probit_mean"][mu], (s1 / S)[mu], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["probit_sd"][mu], np.sqrt((s2 - s1 * s1 / S) / (S - 1.0))[mu], rtol=1e-7)
        assert np.isinf(r["probit_mean"][~mu]).all() and np.isnan(r["probit_sd"][~mu]).all()   # :1243-1262 with a single-transcript gene
    q.close()


def test_summary_edge_shapes(gpu, orc):
    """No virtual transcripts, no identical sets, a transcript outside every gene, a trace length that is no power of two
    (Sokal refuses: return code 201 as src/sokal.cc:119-126), percentile index at both ends."""
    S = 48
    prob, s, n, rng = _setup(gpu, orc, S, seed=8)
    genes = [[t for t in range(1, n)]]                           # transcript 0 belongs to no gene
    q = gpu.Summary(s, chain=0, genes=genes, percentile_index=[0, S - 1])
    trace = s.trace(0)
    r = q.series(gpu.SERIES_TRANSCRIPT)
    assert (r["rc"] == 201).all() and (r["var"] == 0).all()
    assert np.array_equal(r["percentiles"], np.stack([trace.min(axis=1), trace.max(axis=1)], axis=1))
    np.testing.assert_allclose(r["log_mean"], np.log(trace).mean(axis=1), rtol=1e-12)
    pr = q.rows(gpu.SERIES_TRANSCRIPT)
    assert np.isnan(pr[:, 0]).all() and np.isfinite(pr[:, 1:]).all()
    tg = np.zeros(S)
    for t in range(1, n):
        tg += trace[t]
    assert np.array_equal(q.rows(gpu.SERIES_GENE)[:, 0], tg)
    assert q.series(gpu.SERIES_VIRTUAL)["log_mean"].size == 0 and q.rows(gpu.SERIES_IDENTICAL).shape == (S, 0)
    q.close()


@pytest.mark.parametrize("S", [2048, 4096, 8192, 16384, 12000])
def test_summary_of_long_traces(gpu, orc, S):
    """Traces longer than the reference's default 1024 samples: up to 8192 the series is sorted and transformed in LDS, beyond that in a
    global workspace (the reference's sokal takes up to 2^21 samples, src/sokal.cc:36); 12000 is no power of two: percentiles and
    means exist, Sokal returns 201."""
    p, _ = orc.synth_problem(R=3000, T=70, avg_hits=4, seed=12, sort=False)
    n = p.n
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    mu0, _ = prob.start_values()
    s = gpu.Sampler(prob, mu0, seed=9, gibbs_iter=S, trace_len=S)
    s.run(S)
    trace = s.trace(0)
    genes = [[0, 1, 2], [3], list(range(4, n))]
    pidx = [0, int(np.floor(0.5 * (S - 1) + 0.5)), S - 1]
    q = gpu.Summary(s, genes=genes, identical=[[5, 6]], virtual_id=np.array([901], np.uint64), virtual_scale=np.array([0.5]), percentile_index=pidx)
    V = orc.simu_gamma_trace(9, 901, 0.1, 0.5, S)[None, :]
    t_gene = np.zeros((3, S))
    for g, ms in enumerate(genes):
        for m in ms:
            t_gene[g] += trace[m]
    assert np.array_equal(q.rows(gpu.SERIES_GENE).T, t_gene)
    pow2 = S & (S - 1) == 0
    for kind, tr in ((gpu.SERIES_TRANSCRIPT, trace), (gpu.SERIES_GENE, t_gene), (gpu.SERIES_VIRTUAL, V)):
        r = q.series(kind)
        assert np.array_equal(r["percentiles"], np.sort(tr, axis=1)[:, pidx])
        with np.errstate(divide="ignore"):
            lt = np.log(tr)
        np.testing.assert_allclose(r["log_mean"], lt.mean(axis=1), rtol=1e-12, atol=1e-12)
        assert (r["rc"] == (0 if pow2 else 201)).all()
        for i in range(0, tr.shape[0], 7):
            rc, var, tau, m = orc.sokal(lt[i])
            assert rc == (0 if pow2 else 201)
            if pow2:
                np.testing.assert_allclose([r["var"][i], r["tau"][i]], [var, tau], rtol=1e-9)
    gene_of = np.empty(n, np.int64)
    for g, ms in enumerate(genes):
        gene_of[ms] = g
    pr = trace / t_gene[gene_of]
    r = q.proportions(gpu.SERIES_TRANSCRIPT)
    assert np.array_equal(r["percentiles"], np.sort(pr, axis=1)[:, pidx])
    np.testing.assert_allclose(r["mean"], pr.mean(axis=1), rtol=1e-12)
    q.close()


def test_summary_in_steps_while_the_chain_runs_equals_the_summary_after_it(gpu, orc):
    """mmg_summary_begin / _advance / _finish, mmg_sampler_wait_iterations and mmg_sampler_get_trace_rows_done: the trace writers of src/mmseq.cpp:911-917 print sample s
    inside the loop; here the caller feeds finished samples to the summary and fetches their rows WHILE later iterations are enqueued
    (nothing waits for them: own streams), and ends up with the bits of the summary computed after the chain."""
    S = 64
    p, _ = orc.synth_problem(R=20000, T=700, avg_hits=5, seed=11, sort=False)
    n = p.n
    rng = np.random.default_rng(2)
    txo = (rng.permutation(n).astype(np.uint64) // np.uint64(4)) << np.uint64(32)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, tx_order=txo)
    mu0, _ = prob.start_values()
    nv = 9
    vid = (500 + np.arange(nv) * 3).astype(np.uint64)
    vscale = rng.uniform(0.1, 1.0, nv)
    members = rng.permutation(n + nv)
    genes = [[int(m) for m in members[i:i + 3]] for i in range(0, members.size, 3)]
    kw = dict(chain=0, virtual_id=vid, virtual_scale=vscale, identical=[[1, 2], [5]], genes=genes, percentile_index=[3, 31, 60])
    a = gpu.Sampler(prob, mu0, seed=3, gibbs_iter=2 * S, trace_len=S)   # a sample every second iteration
    qa = gpu.Summary(a, staged=True, **kw)
    with pytest.raises(gpu._lib.MMGError):
        qa.rows(gpu.SERIES_GENE, 0, 1)                  # nothing advanced yet
    got_rows, got_gene = [], []
    a.run(32)
    with pytest.raises(gpu._lib.MMGError):
        a.wait_iterations(33)                            # more than was enqueued
    for c in range(4):                                   # 4 chunks of 32 iterations = 16 samples
        if c < 3:
            a.run(32)                                    # the next chunk is enqueued BEFORE this one is waited for
        # mmg_sampler_wait_iterations: sample 16 c + 15 is stored by iteration 32 c + 30, the 32 c + 31st; the last chunk asks for every
        # iteration (behind the last stored sample: the whole stream)
        a.wait_iterations(32 * c + 31 if c < 3 else 128)
        qa.advance(16 * (c + 1))
        got_rows.append(a.trace_rows_done(0, 16 * c, 16))
        got_gene.append(qa.rows(gpu.SERIES_GENE, 16 * c, 16))
        if c < 3:
            with pytest.raises(gpu._lib.MMGError):
                qa.series(gpu.SERIES_GENE)               # columns exist after finish()
            with pytest.raises(gpu._lib.MMGError):
                qa.finish()                              # ... which wants every sample
    qa.finish()
    # the reference: the same chain in one go, summarised after it
    b = gpu.Sampler(prob, mu0, seed=3, gibbs_iter=2 * S, trace_len=S)
    b.run(2 * S)
    qb = gpu.Summary(b, **kw)
    for kind in (gpu.SERIES_TRANSCRIPT, gpu.SERIES_VIRTUAL, gpu.SERIES_IDENTICAL, gpu.SERIES_GENE):
        sa, sb = qa.series(kind), qb.series(kind)
        for k in sa:
            assert np.array_equal(sa[k], sb[k], equal_nan=True), (kind, k)
    for kind in (gpu.SERIES_TRANSCRIPT, gpu.SERIES_VIRTUAL):
        pa, pb = qa.proportions(kind), qb.proportions(kind)
        for k in pa:
            assert np.array_equal(pa[k], pb[k], equal_nan=True), (kind, k)
    for kind in (gpu.SERIES_TRANSCRIPT, gpu.SERIES_IDENTICAL, gpu.SERIES_GENE):
        assert np.array_equal(qa.rows(kind), qb.rows(kind), equal_nan=True)
    assert np.array_equal(np.concatenate(got_gene), qb.rows(gpu.SERIES_GENE))
    assert np.array_equal(np.concatenate(got_rows), b.trace_rows(0))
    qa.close(); qb.close(); a.close(); b.close(); prob.close()


def test_row_fetches_larger_than_a_staging_buffer(gpu):
    """mmg_sampler_get_trace_rows_done and mmg_summary_get_rows copy through two 16 MB pinned buffers (mmg_host.h: PinnedStage): rows of
    200 k transcripts x 24 samples are 38 MB, three chunks with a ragged last one -- the same numbers as the one-piece copies."""
    n, S = 200_000, 24
    prob = gpu.Problem.synthetic(400_000, n, 6, seed=4)
    mu0, _ = prob.start_values()
    s = gpu.Sampler(prob, mu0, seed=2, gibbs_iter=S, trace_len=S)
    s.run(S)
    s.sync()
    whole = s.trace_rows(0)                                       # one hipMemcpy of the gathered rows
    assert np.array_equal(s.trace_rows_done(0, 0, S), whole)
    assert np.array_equal(s.trace_rows_done(0, 5, 17), whole[5:22])
    assert np.array_equal(s.trace(0).T, whole)
    genes = [[t, t + 1] for t in range(0, n, 2)]
    q = gpu.Summary(s, genes=genes, percentile_index=[0])
    prop = q.rows(gpu.SERIES_TRANSCRIPT)                          # [S, n]: 38 MB through the summary's buffers
    gsum = q.rows(gpu.SERIES_GENE)
    assert np.array_equal(gsum, whole[:, 0::2] + whole[:, 1::2])
    assert np.array_equal(prop, whole / np.repeat(gsum, 2, axis=1))
    q.close(); s.close(); prob.close()
