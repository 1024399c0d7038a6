"""SURVEY App. E.3 on the device: the product chain (keyed Philox streams, categorical draws, canonical row order) against the
REFERENCE-STRUCTURED engine of the oracle (one MT19937 per thread seeded seed + tid, count slabs, multinomial by conditional
binomials, Marsaglia-Tsang Gamma: the structure of src/mmseq.cpp:834-918) on the same problem and start value.  The two share
no random numbers, no row order and no sampling algorithm, only the model -- so agreement of the posterior summaries within
Monte Carlo error is evidence that does not depend on the builder's own keyed spec.  Summaries as the reference computes them:
mean of the logged trace (src/mmseq.cpp:1195-1227), Sokal's variance and autocorrelation time (:1311-1324)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed_dev,seed_ref", [(11, 1001), (12, 1002), (13, 1003)])
def test_device_chain_agrees_with_reference_structured_engine(gpu, orc, seed_dev, seed_ref):
    R, T, S = 500_000, 5_000, 1024
    q, _ = orc.synth_problem(R=R, T=T, avg_hits=6, seed=77, sort=False)         # generator order, as a reader would deliver it
    prob = gpu.Problem.from_csr(q.row_ptr, q.col_idx, q.l)
    mu0, _ = prob.start_values()
    mu_em, it, _ = prob.em(mu0)                                                   # both chains start at the EM optimum (src/mmseq.cpp:820)
    s = gpu.Sampler(prob, mu_em, seed=seed_dev, gibbs_iter=S, trace_len=S)
    s.run(S)
    summ = gpu.Summary(s, chain=0)
    dev = summ.series(gpu.SERIES_TRANSCRIPT)
    assert (dev["rc"] == 0).all()
    threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    ref = orc.gibbs_ref(q, mu_em, seed=seed_ref, n_iter=S, trace_len=S, threads=threads)["trace"]
    with np.errstate(divide="ignore"):
        lref = np.log(ref)
    obs = np.unique(q.col_idx)
    z, sdr, sdtol = [], [], []
    for t in obs:
        rc, var_b, tau_b, _ = orc.sokal(lref[t].copy())
        var_a, tau_a = dev["var"][t], dev["tau"][t]
        if rc or not (tau_a < 20 and tau_b < 20 and var_a > 0 and var_b > 0):
            continue
        mc = np.sqrt(tau_a * var_a / S + tau_b * var_b / S)                       # mcse of either mean, :1320-1323
        z.append((dev["log_mean"][t] - lref[t].mean()) / mc)
        sdr.append(np.sqrt(var_a / var_b))
        sdtol.append(5 * np.sqrt(max(tau_a, tau_b) / (2.0 * S)))
    z, sdr, sdtol = np.array(z), np.array(sdr), np.array(sdtol)
    assert len(z) > 0.9 * len(obs)
    # SURVEY App. E.3 AS WRITTEN (the null distribution of these statistics -- one engine against itself, 24 seed pairs each for the
    # reference-structured and the keyed engine on this very problem, profiles/r05_e3_null_{ref,keyed}.txt -- sits well inside every
    # bound: |mean| <= 0.017, variance 1.09-1.19, median sd ratio within 0.0023 of 1, all |z| <= 7.2)
    assert (np.abs(z) <= 5).mean() >= 0.99 and np.abs(z).max() < 8               # |delta log_mu| <= 5 sqrt(mcse_a^2 + mcse_b^2) for >= 99 %, none beyond 8 x
    # sd ratio within 1 +- 5 sqrt(tau / (2 * 1024)) per transcript: DEVIATION from App. E.3, which asks it of every transcript.  The
    # bound assumes Gaussian samples; log mu of a transcript with few reads is log-Gamma with a heavy left tail, and one engine
    # against ITSELF holds it for 95.8 % of the transcripts (tools/e3_null.py, profiles/r05_e3_null_sd_clause.txt; 98.3 % within twice
    # the bound).  Held here at the null's level: a wrong posterior sd would show in the median (next line, as written) long before.
    assert (np.abs(sdr - 1) <= sdtol).mean() >= 0.94
    assert abs(np.median(sdr) - 1) < 0.01                                         # median ratio within 1 +- 0.01
    assert abs(z.mean()) < 0.05 and 0.8 < z.var() < 1.3                           # pooled z: |mean| < 0.05, variance in [0.8, 1.3]
    summ.close(); s.close(); prob.close()


def _compare(orc, dev, lref, ids, S_a, S_b, min_frac=0.9):
    """z scores and sd ratios of device summaries (log_mean, var, tau per series) against logged reference traces lref[id]."""
    z, sdr = [], []
    for j, t in enumerate(ids):
        x = lref[j]
        if not np.isfinite(x).all():
            continue
        rc, var_b, tau_b, _ = orc.sokal(x[:1 << int(np.log2(x.size))].copy())
        var_a, tau_a = dev["var"][t], dev["tau"][t]
        if rc or not (tau_a < 20 and tau_b < 20 and var_a > 0 and var_b > 0):
            continue
        mc = np.sqrt(tau_a * var_a / S_a + tau_b * var_b / S_b)
        z.append((dev["log_mean"][t] - x.mean()) / mc)
        sdr.append(np.sqrt(var_a / var_b))
    z, sdr = np.array(z), np.array(sdr)
    assert len(z) > min_frac * len(ids), (len(z), len(ids))
    return z, sdr


def test_real_file_shape_transcripts_and_genes_agree_with_reference_structured_engine(gpu, orc):
    """App. E.3 on the shape every real hits file has: identical reads collapsed into rows with multiplicities -- k = 2..64 (stored as
    k rows by the canonical layout), k up to 2000 (conditional-binomial chain, the multiplicity kernel) -- and 20 % of the rows with a
    hit anywhere in the transcriptome (far tiles).  Transcript level AND gene level ("Same for gene-level rows": sums of the isoforms'
    mu per sample, src/mmseq.cpp:1069-1090), against the reference-structured engine, which collapses nothing, knows no tiles and
    draws every row with conditional binomials from its own MT19937 streams."""
    R, T, S = 300_000, 4_000, 1024
    q, _ = orc.synth_problem(R=R, T=T, avg_hits=6, seed=78, sort=False, far_fraction=0.2)
    rng = np.random.default_rng(5)
    k = rng.choice([1, 1, 1, 1, 1, 1, 2, 3, 9, 40, 64, 65, 300, 2000], size=q.m, p=[.14] * 6 + [.05, .04, .03, .02, .005, .005, .005, .005]).astype(np.uint32)
    qk = orc.Problem(q.row_ptr, q.col_idx, q.l * (k.sum() / R), k=k)             # l = efflen * mapped reads / 1e9 (src/mmseq.cpp:603)
    prob = gpu.Problem.from_csr(qk.row_ptr, qk.col_idx, qk.l, k=k)
    inf = prob.info
    assert inf.far_tiles > 0 and inf.m > q.m and inf.total_k == int(k.astype(np.int64).sum())
    mu0, _ = prob.start_values()
    mu_em, _, _ = prob.em(mu0)
    genes = [list(range(g, min(g + 4, T))) for g in range(0, T, 4)]               # consecutive isoforms, as App. D generates them
    s = gpu.Sampler(prob, mu_em, seed=21, gibbs_iter=S, trace_len=S)
    s.run(S)
    summ = gpu.Summary(s, chain=0, genes=genes)
    dev_t, dev_g = summ.series(gpu.SERIES_TRANSCRIPT), summ.series(gpu.SERIES_GENE)
    threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    ref = orc.gibbs_ref(qk, mu_em, seed=2001, n_iter=S, trace_len=S, threads=threads)["trace"]
    obs = np.unique(q.col_idx)
    with np.errstate(divide="ignore"):
        z, sdr = _compare(orc, dev_t, np.log(ref[obs]), obs, S, S, min_frac=0.8)   # (fewer reads per transcript than E.3's shape: more iact >= 20)
        gobs = [g for g, mem in enumerate(genes) if np.isin(mem, obs).any()]
        zg, sdrg = _compare(orc, dev_g, np.log(np.stack([ref[genes[g]].sum(axis=0) for g in gobs])), gobs, S, S, min_frac=0.8)
    # transcripts: SURVEY App. E.3 as written (null of this very shape, the reference-structured engine against itself over 20 seed pairs,
    # profiles/r05_e3_null_ref_realshape.txt: |mean| <= 0.038, variance 1.13-1.24, median sd ratio within 0.0022 of 1, all |z| <= 5.6)
    assert (np.abs(z) <= 5).mean() >= 0.99 and np.abs(z).max() < 8
    assert abs(np.median(sdr) - 1) < 0.01
    assert abs(z.mean()) < 0.05 and 0.8 < z.var() < 1.3
    # genes ("same for gene-level rows"): DEVIATION from App. E.3 in two figures.  There are 894 gene series here, not 4 000: in the null
    # (profiles/r05_e3_null_ref_realshape_genes.txt, 20 seed pairs) the pooled mean reaches 0.057 and the variance 1.298 -- one engine
    # against itself fails "|mean| < 0.05" once in twenty and sits on "variance < 1.3".  Held at |mean| < 0.08, variance in [0.8, 1.35];
    # within-5, the 8 x limit and the median sd ratio as written (null: sd ratio within 0.006).
    assert (np.abs(zg) <= 5).mean() >= 0.99 and np.abs(zg).max() < 8
    assert abs(np.median(sdrg) - 1) < 0.01
    assert abs(zg.mean()) < 0.08 and 0.8 < zg.var() < 1.35
    summ.close(); s.close(); prob.close()


@pytest.mark.parametrize("seed_dev,seed_ref", [(31, 3001), (32, 3002)])
def test_four_pooled_device_chains_against_one_long_reference_structured_chain(gpu, orc, seed_dev, seed_ref):
    """BASELINE configs[2]/[3]: the pooled moments of four device chains (fused pairs; mmg_sampler_get_moments, the payload of the one
    all-reduce of chains mode) against ONE reference-structured chain four times as long: pooled mean of log mu and pooled sd agree
    within Monte Carlo error -- chains differ only in their Philox key and are exchangeable with one long run."""
    R, T, S, C = 300_000, 3_000, 1024, 4
    q, _ = orc.synth_problem(R=R, T=T, avg_hits=6, seed=79, sort=False)
    prob = gpu.Problem.from_csr(q.row_ptr, q.col_idx, q.l)
    mu0, _ = prob.start_values()
    mu_em, _, _ = prob.em(mu0)
    s = gpu.Sampler(prob, mu_em, seed=seed_dev, n_chains=C, gibbs_iter=S, trace_len=S)
    s.run(S)
    B = 128                                                # every chain starts at the EM point: four transients against one in the
    obs = np.unique(q.col_idx)                             # long chain would show as a bias of 0.1 mcse -- compared without them
    n = C * (S - B)
    sl = np.zeros(T); sl2 = np.zeros(T)
    for c in range(C):
        a, b, ns = s.moments(c)
        with np.errstate(divide="ignore"):
            lt = np.log(s.trace(c))
        assert ns == S
        np.testing.assert_allclose(a[obs], lt[obs].sum(axis=1), rtol=1e-11)          # the moments ARE the sums over the kept samples
        np.testing.assert_allclose(b[obs], (lt[obs] ** 2).sum(axis=1), rtol=1e-11)
        sl[obs] += lt[obs, B:].sum(axis=1); sl2[obs] += (lt[obs, B:] ** 2).sum(axis=1)
    mean_a = sl / n
    var_a = (sl2 - n * mean_a ** 2) / (n - 1)
    threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    ref = orc.gibbs_ref(q, mu_em, seed=seed_ref, n_iter=C * S, trace_len=C * S, threads=threads)["trace"]
    z, sdr = [], []
    with np.errstate(divide="ignore"):
        for t in obs:
            x = np.log(ref[t])
            rc, var_b, tau_b, _ = orc.sokal(x.copy())
            x = x[B:]
            if rc or not (tau_b < 20 and var_b > 0 and var_a[t] > 0):
                continue
            z.append((mean_a[t] - x.mean()) / np.sqrt(tau_b * var_b * (1.0 / n + 1.0 / x.size)))
            sdr.append(np.sqrt(var_a[t] / var_b))
    z, sdr = np.array(z), np.array(sdr)
    assert len(z) > 0.9 * len(obs)
    print("pooled z: mean %+.3f var %.3f within5 %.4f median sd ratio %.4f" % (z.mean(), z.var(), (np.abs(z) <= 5).mean(), np.median(sdr)))
    # The z of neighbouring transcripts share reads: their mean scatters by ~0.1 between seed pairs, not by 1 / sqrt(len(z)).  At 4096
    # samples Sokal's adaptive window (src/sokal.cc:73-83) cuts the slow tail of the autocorrelation off and the mcse comes out ~20 %
    # low: two runs of ONE engine against each other give z variances of 1.4-1.5 at 16384 samples (keyed vs keyed, reference-structured
    # vs reference-structured, on the CPU), so the bound here is wider than at 1024 samples.
    assert (np.abs(z) <= 5).mean() >= 0.99 and abs(z.mean()) < 0.25 and 0.75 < z.var() < 2.5
    assert abs(np.median(sdr) - 1) < 0.01
    s.close(); prob.close()


def _tiny(gpu, rows, k, l, **kw):
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    return gpu.Problem.from_csr(rp, ci, np.asarray(l, np.float64), k=np.asarray(k, np.uint32), **kw)


def test_analytic_posteriors_on_the_device(gpu):
    """App. E.2 on the device, against closed forms (no oracle involved): (1) an isolated transcript's samples are iid
    Gamma(alpha + c, 1/(beta + l)); (2) a symmetric pair: mu_A + mu_B ~ Gamma(2 alpha + k, 1/(beta + l)), proportion symmetric about
    1/2; (4) the wrong-weight detector: with l_A = 10 l_B the shared reads split by mu (src/mmseq.cpp:876), not by mu * l."""
    from scipy.special import digamma, polygamma
    S = 4096
    # (1) rows {0} x 7, {0} x 3, {1} x 5, {1,2} x 4: transcript 0 is isolated with c = 10
    prob = _tiny(gpu, [[0], [0], [1], [1, 2]], [7, 3, 5, 4], [0.8, 1.3, 0.6])
    s = gpu.Sampler(prob, np.ones(3), seed=5, gibbs_iter=S, trace_len=S)
    s.run(S)
    lg = np.log(s.trace(0)[0])
    assert abs(lg.mean() - (digamma(10.1) - np.log(0.9))) < 5 * np.sqrt(polygamma(1, 10.1) / S)
    assert abs(lg.std(ddof=1) - np.sqrt(polygamma(1, 10.1))) < 0.03          # 0.3226
    s.close()
    s = gpu.Sampler(prob, np.ones(3), seed=5, gibbs_iter=2048, trace_len=2048)
    s.run(2048)
    summ = gpu.Summary(s, chain=0)
    assert 0.7 < summ.series(gpu.SERIES_TRANSCRIPT)["tau"][0] < 1.4           # iid => iact ~ 1
    summ.close(); s.close(); prob.close()
    # (2) one row {A, B} with k = 60, alpha = 3
    S2 = 16384
    prob = _tiny(gpu, [[0, 1]], [60], [1.0, 1.0])
    assert prob.info.m == 1                   # (60 stored rows per uploaded row is beyond the expansion limit of 8: the row keeps k = 60 and draws 60 categoricals)
    s = gpu.Sampler(prob, np.ones(2), seed=6, alpha=3.0, gibbs_iter=S2, trace_len=S2)
    s.run(S2)
    tr = s.trace(0)
    tot = tr[0] + tr[1]
    assert abs(np.log(tot).mean() - (digamma(66.0) - np.log(1.1))) < 0.01
    prop = tr[0] / tot
    assert abs(prop.mean() - 0.5) < 0.03 and abs(np.percentile(prop, 25) + np.percentile(prop, 75) - 1.0) < 0.06
    s.close(); prob.close()
    # (4) rows {A} x 200, {B} x 200, {A,B} x 400 (the binomial chain: k > 64), l_A = 10 l_B
    a, b, c, lA, lB = 200, 200, 400, 10.0, 1.0
    prob = _tiny(gpu, [[0], [1], [0, 1]], [a, b, c], [lA, lB])
    s = gpu.Sampler(prob, np.ones(2), seed=7, gibbs_iter=S, trace_len=S)
    s.run(S)
    got = s.trace(0)[:, 512:].mean(axis=1)

    def fixed(weight_with_l):
        mA, mB = 1.0, 1.0
        for _ in range(2000):
            wA, wB = (mA * lA, mB * lB) if weight_with_l else (mA, mB)
            xA, xB = a + c * wA / (wA + wB), b + c * wB / (wA + wB)
            mA, mB = (0.1 + xA) / (0.1 + lA), (0.1 + xB) / (0.1 + lB)
        return mA, mB
    good, bad = fixed(False), fixed(True)
    assert abs(got[0] - good[0]) / good[0] < 0.03 and abs(got[1] - good[1]) / good[1] < 0.03
    assert abs(got[0] - bad[0]) / bad[0] > 0.15
    s.close(); prob.close()
