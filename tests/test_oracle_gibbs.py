"""Pins the oracle's Gibbs engines with analytic known answers and checks that
the keyed-stream engine (the HIP kernels' bit-exact spec) and the reference-structured engine
(MT19937 per thread, conditional binomials; the timed CPU baseline) agree statistically (App. E.3).
The reference itself ships no tests for this path and cannot be built here (GSL/Boost absent):
parity with the reference binary is UNPINNED; these tests pin the restatement to the mathematics."""
import json
import os

import numpy as np
import pytest
from scipy.special import digamma, polygamma

GOLD = os.path.join(os.path.dirname(__file__), "golden", "keyed_chain_tiny.json")


def _problem(orc, rows, k, l):
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows]) if rp[-1] else np.zeros(0, np.uint32)
    return orc.Problem(rp, ci, np.asarray(l, np.float64), k=None if k is None else np.asarray(k, np.uint32))


ENGINES = ["keyed", "ref"]


def _run(orc, engine, p, mu0, n_iter=1024, trace_len=1024, seed=1234, **kw):
    if engine == "keyed":
        return orc.gibbs_keyed(p, mu0, seed=seed, n_iter=n_iter, trace_len=trace_len, **kw)
    return orc.gibbs_ref(p, mu0, seed=seed, n_iter=n_iter, trace_len=trace_len, threads=2, **kw)


@pytest.mark.parametrize("engine", ENGINES)
def test_isolated_transcript_posterior(orc, engine):
    """E.2-1: all rows containing t are {t}: samples are iid Gamma(alpha + c, 1/(beta + l))."""
    p = _problem(orc, [[0], [0], [1], [1, 2]], [7, 3, 5, 4], [0.8, 1.3, 0.6])
    r = _run(orc, engine, p, np.ones(3), n_iter=4096, trace_len=4096)
    lg = np.log(r["trace"][0])
    c = 10
    assert abs(lg.mean() - (digamma(0.1 + c) - np.log(0.1 + 0.8))) < 5 * np.sqrt(polygamma(1, 0.1 + c) / 4096)
    assert abs(lg.std(ddof=1) - np.sqrt(polygamma(1, 0.1 + c))) < 0.03      # sd = 0.3226 for c = 10
    rc, var, tau, m = orc.sokal(lg.copy())
    assert rc == 0 and 0.7 < tau < 1.4                                       # iid => iact ~ 1


@pytest.mark.parametrize("engine", ENGINES)
def test_symmetric_pair(orc, engine):
    """E.2-2: A,B equal l, only rows {A,B}: mu_A + mu_B ~ Gamma(2 alpha + k), proportion symmetric."""
    # alpha = 3 keeps the two-mode Beta(alpha + x, alpha + k - x) mixing fast enough for a short chain
    p = _problem(orc, [[0, 1]], [60], [1.0, 1.0])
    r = _run(orc, engine, p, np.array([1.0, 1.0]), n_iter=16384, trace_len=16384, alpha=3.0)
    tot = r["trace"][0] + r["trace"][1]
    assert abs(np.log(tot).mean() - (digamma(6.0 + 60) - np.log(1.1))) < 0.01
    prop = r["trace"][0] / tot
    assert abs(prop.mean() - 0.5) < 0.03
    assert abs(np.percentile(prop, 25) + np.percentile(prop, 75) - 1.0) < 0.06


@pytest.mark.parametrize("engine", ENGINES)
def test_weights_are_mu_not_mu_times_length(orc, engine):
    """E.2-4 wrong-weight detector: the categorical weight is mu_t (src/mmseq.cpp:876), length enters
    only through the Gamma rate.  With l_A = 10 l_B the stationary split of the shared reads follows
    the fixed point of  mu_t = (alpha + x_t)/(beta + l_t), x_A = a + c mu_A/(mu_A+mu_B)."""
    a, b, c = 200, 200, 400
    lA, lB = 10.0, 1.0
    p = _problem(orc, [[0], [1], [0, 1]], [a, b, c], [lA, lB])
    r = _run(orc, engine, p, np.array([1.0, 1.0]), n_iter=4096, trace_len=4096)
    # deterministic fixed point of the mean recursion with weights mu (correct) vs mu*l (wrong)
    def fixed(weight_with_l):
        mA, mB = 1.0, 1.0
        for _ in range(2000):
            wA, wB = (mA * lA, mB * lB) if weight_with_l else (mA, mB)
            xA = a + c * wA / (wA + wB)
            xB = b + c * wB / (wA + wB)
            mA, mB = (0.1 + xA) / (0.1 + lA), (0.1 + xB) / (0.1 + lB)
        return mA, mB
    good, bad = fixed(False), fixed(True)
    got = r["trace"][:, 512:].mean(axis=1)
    assert abs(got[0] - good[0]) / good[0] < 0.03 and abs(got[1] - good[1]) / good[1] < 0.03
    assert abs(got[0] - bad[0]) / bad[0] > 0.15      # the wrong model is clearly excluded


@pytest.mark.parametrize("engine", ENGINES)
def test_count_conservation_every_iteration(orc, engine):
    p, _ = orc.synth_problem(R=3000, T=200, avg_hits=5, seed=9)
    rng = np.random.default_rng(1)
    p = orc.Problem(p.row_ptr, p.col_idx, p.l * 100, k=rng.integers(1, 40, p.m).astype(np.uint32))
    mu0, _ = orc.start_values(p)
    for it in (1, 2, 5):
        r = _run(orc, engine, p, mu0, n_iter=it, trace_len=it)
        assert int(r["cnt"].sum()) == p.total_k()     # sum_t c_t = sum_i k_i (App. A invariant)


def test_keyed_engine_independent_of_thread_count(orc):
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from oracle import binding as B
        p, _ = B.synth_problem(R=5000, T=300, avg_hits=6, seed=2)
        mu0, _ = B.start_values(p)
        r = B.gibbs_keyed(p, mu0, seed=7, n_iter=16, trace_len=16)
        print(r['trace'].tobytes().hex()[:64], int(r['cnt'].sum()), float(r['trace'].sum()).hex())
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for nt in ("1", "3", "8"):
        env = dict(os.environ, OMP_NUM_THREADS=nt)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env).decode())
    assert outs[0] == outs[1] == outs[2]


def test_keyed_vs_reference_structured_statistical_agreement(orc):
    """App. E.3: per-transcript |delta log_mu| <= 5 sqrt(mcse_a^2 + mcse_b^2) for >= 99%, none beyond 8x;
    median sd ratio within 1 +- 0.03; pooled z-scores centred with unit-ish variance."""
    p, _ = orc.synth_problem(R=20000, T=400, avg_hits=5, seed=4)
    mu0, _ = orc.start_values(p)
    mu_em, _, _ = orc.em(p, mu0)
    a = orc.gibbs_keyed(p, mu_em, seed=1, n_iter=1024, trace_len=1024)["trace"]
    b = orc.gibbs_ref(p, mu_em, seed=2, n_iter=1024, trace_len=1024, threads=2)["trace"]
    obs = np.unique(p.col_idx)
    z, sdr = [], []
    for t in obs:
        la, lb = np.log(a[t]), np.log(b[t])
        ra, rb = orc.sokal(la.copy()), orc.sokal(lb.copy())
        if ra[0] or rb[0] or not (ra[2] < 20 and rb[2] < 20):
            continue
        mc = np.sqrt(ra[2] * ra[1] / 1024 + rb[2] * rb[1] / 1024)
        z.append((la.mean() - lb.mean()) / mc)
        sdr.append(np.sqrt(ra[1] / rb[1]))
    z, sdr = np.array(z), np.array(sdr)
    assert len(z) > 100
    assert (np.abs(z) <= 5).mean() >= 0.99 and np.abs(z).max() < 8
    assert abs(np.median(sdr) - 1) < 0.03
    assert abs(z.mean()) < 0.15 and 0.6 < z.var() < 1.6


def test_degenerate_and_edge_rows(orc):
    # empty rows contribute nothing; single-hit rows need no randomness; zero-weight rows fall back to uniform
    p = _problem(orc, [[], [2], [0, 1], [], [0, 1, 2]], [5, 4, 6, 1, 3], [1.0, 1.0, 1.0])
    cnt = orc.sample_counts(p, np.array([0.0, 0.0, 0.0]), 1, 0, 0)
    assert int(cnt.sum()) == 4 + 6 + 3 and cnt[2] >= 4
    cnt = orc.sample_counts(p, np.array([1.0, 0.0, 5.0]), 1, 0, 0)
    assert cnt[1] == 0                                 # a zero-weight transcript is never chosen
    cnt = orc.sample_counts(p, np.array([1e-320, 3e-320, 1e-320]), 1, 0, 0)   # subnormal weights still work
    assert int(cnt.sum()) == 13


@pytest.mark.parametrize("name,seed", [("keyed_chain_tiny.json", 1234), ("keyed_chain_k_draws.json", 4321)])
def test_golden_tiny_chain(orc, name, seed):
    """keyed_chain_tiny.json: k up to 1000, single hits, small multiplicities, binomial chains.  keyed_chain_k_draws.json (spec version 8):
    rows on either side of the boundary k <= min(64, 16 (hits - 1)) between k categorical draws and the conditional-binomial chain."""
    g = json.load(open(os.path.join(os.path.dirname(GOLD), name)))
    f = lambda hs: np.array([float.fromhex(h) for h in hs])
    p = orc.Problem(np.array(g["row_ptr"], np.uint64), np.array(g["col_idx"], np.uint32), f(g["l"]),
                    k=np.array(g["k"], np.uint32))
    mu0, uh = orc.start_values(p)
    assert np.array_equal(mu0, f(g["mu0"])) and uh.tolist() == g["unique_hits"]
    r = orc.gibbs_keyed(p, mu0, seed=seed, n_iter=32, trace_len=16)
    assert np.array_equal(r["trace"].ravel(), f(g["trace"]))
    assert r["cnt"].tolist() == g["cnt_last"] and np.array_equal(r["mu"], f(g["mu_last"]))


@pytest.mark.parametrize("k", [96, 97])
def test_multiplicity_rows_on_both_sides_of_the_draws_boundary_are_multinomial(orc, k):
    """A row of 7 hits draws its categoricals one by one up to k = 16 * 6 = 96 and runs the conditional-binomial chain above
    (src/mmseq.cpp:880 is a multinomial either way): 4000 such rows, one sweep at fixed weights, against k * p."""
    R, L = 4000, 7
    rp = (np.arange(R + 1) * L).astype(np.uint64)
    ci = np.tile(np.arange(L, dtype=np.uint32), R)
    p = orc.Problem(rp, ci, np.ones(L), k=np.full(R, k, np.uint32))
    w = np.array([0.4, 0.25, 0.15, 0.1, 0.05, 0.03, 0.02])
    r = orc.gibbs_keyed(p, w.copy(), seed=77 + k, n_iter=1, trace_len=1)
    cnt = r["cnt"].astype(np.float64)
    n = float(R * k)
    assert cnt.sum() == n
    z = (cnt - n * w) / np.sqrt(n * w * (1 - w))
    assert np.abs(z).max() < 4.5, z


def test_em_and_start_values(orc):
    """src/mmseq.cpp:617-638 and :741-811 on a case solvable by hand: disjoint singletons converge in one
    step to mu_t = k_t / l_t (the MLE), and the likelihood never decreases."""
    p = _problem(orc, [[0], [1], [2]], [10, 4, 1], [2.0, 0.5, 4.0])
    mu0, uh = orc.start_values(p)
    assert np.allclose(mu0, [5.0, 8.0, 0.25]) and uh.tolist() == [10, 4, 1]
    mu, it, ll = orc.em(p, mu0)
    assert np.allclose(mu, [5.0, 8.0, 0.25]) and it == 1
    p2, _ = orc.synth_problem(R=4000, T=150, avg_hits=4, seed=3)
    m0, _ = orc.start_values(p2)
    prev = -np.inf
    mu = m0
    for _ in range(5):
        mu, it, ll = orc.em(p2, mu, max_iter=1, epsilon=-1.0)
        assert ll >= prev - 1e-9
        prev = ll


def test_uh_literal_semantics(orc):
    """src/uh.cpp:3-26: rows entirely inside a group count k_i; an EMPTY row counts for every group."""
    p = _problem(orc, [[0], [0, 1], [2], [], [1, 2]], [3, 4, 5, 7, 1], [1, 1, 1])
    member = np.array([[1, 0], [1, 0], [0, 1]], np.uint8)   # group0={0,1}, group1={2}
    assert orc.uh(p, member).tolist() == [3 + 4 + 7, 5 + 7]


def test_em_fixed_point_spec_tracks_reference_order(orc):
    """orc.em (exact fixed-point sums, the spec the device implements) against orc.em_seq (src/mmseq.cpp:761-811 in the
    reference's own summation order): same iteration counts, same zero pattern, mu within fp64 rounding of the sums --
    over 300 sweeps, with transcripts decaying towards zero, and from start values spanning 10^+-150."""
    p, _ = orc.synth_problem(R=20000, T=900, avg_hits=6, seed=5)
    mu0, _ = orc.start_values(p)
    a, b = orc.em(p, mu0), orc.em_seq(p, mu0)
    assert a[1] == b[1] and abs(a[2] - b[2]) < 1e-6
    np.testing.assert_allclose(a[0], b[0], rtol=1e-12)
    a = orc.em_x(p, mu0, max_iter=300, epsilon=-1e308)
    b = orc.em_seq(p, mu0, max_iter=300, epsilon=-1e308)
    assert a[3] == 0                                     # ordinary sweeps never need the measured-exponent repeat
    live = b[0] > 0
    assert np.array_equal(a[0] == 0, b[0] == 0) and b[0][live].min() < 1e-60
    np.testing.assert_allclose(a[0][live], b[0][live], rtol=1e-11)
    rng = np.random.default_rng(0)
    k = rng.integers(1, 1000, size=p.m).astype(np.uint32)
    pk = orc.Problem(p.row_ptr, p.col_idx, p.l * 10, k=k)
    mu0, _ = orc.start_values(pk)
    for scale in (1e-150, 1e-30, 1e150):
        m0 = mu0.copy()
        m0[::7] *= scale
        m0[3] = 0.0
        a = orc.em_x(pk, m0, max_iter=40, epsilon=-1e308)
        b = orc.em_seq(pk, m0, max_iter=40, epsilon=-1e308)
        live = b[0] > 0
        assert a[3] >= 1 and np.array_equal(a[0] == 0, b[0] == 0)
        np.testing.assert_allclose(a[0][live], b[0][live], rtol=1e-11)
        assert abs(a[2] - b[2]) < 1e-4


def test_em_term_limbs_are_exact_shifts(orc):
    """One sweep of orc.em_x against a restatement in Python integers and fractions of its header (oracle/mmseq_oracle.c, 'EM with
    order-independent accumulation'): measured exponents, HI += floor(x 2^E), LO += floor(frac(x 2^E) 2^sl), S from the two limbs."""
    from fractions import Fraction
    import math
    p, _ = orc.synth_problem(R=3000, T=150, avg_hits=5, seed=11)
    mu0, _ = orc.start_values(p)
    mu0[::5] *= 1e-40                                    # terms far below one unit of their transcript's HI limb
    n = p.n
    rp, ci = p.row_ptr, p.col_idx
    N = np.bincount(ci, minlength=n)
    sl = [63 - int(v).bit_length() for v in N]
    x = np.zeros(p.m)
    for i in range(p.m):
        d = 0.0
        for j in range(int(rp[i]), int(rp[i + 1])): d += mu0[ci[j]]
        x[i] = 1.0 / d if rp[i + 1] > rp[i] else 0.0
    XE = [None] * n
    for i in range(p.m):
        for j in range(int(rp[i]), int(rp[i + 1])):
            e = math.frexp(x[i])[1] - 1
            XE[ci[j]] = e if XE[ci[j]] is None else max(XE[ci[j]], e)
    HI, LO = [0] * n, [0] * n
    for i in range(p.m):
        for j in range(int(rp[i]), int(rp[i + 1])):
            t = int(ci[j])
            Y = Fraction(float(x[i])) * Fraction(2) ** (sl[t] - 2 - XE[t])
            HI[t] += Y.numerator // Y.denominator
            LO[t] += math.floor((Y - Y.numerator // Y.denominator) * 2 ** sl[t])
    want = mu0.copy()
    for t in range(n):
        if XE[t] is None:
            want[t] = mu0[t] * 0.0 / p.l[t]
            continue
        assert HI[t] < 2 ** 63 and LO[t] < 2 ** 63
        S = math.ldexp(float(HI[t]) + math.ldexp(float(LO[t]), -sl[t]), -(sl[t] - 2 - XE[t]))
        want[t] = mu0[t] * S / p.l[t]
    got = orc.em_x(p, mu0, max_iter=1, epsilon=-1e308)[0]
    assert np.array_equal(got, want)


def _em_golden():
    import json
    gd = os.path.join(os.path.dirname(__file__), "golden")
    g = json.load(open(os.path.join(gd, "keyed_chain_tiny.json")))
    e = json.load(open(os.path.join(gd, "em_fixed_tiny.json")))
    fh = lambda xs: np.array([float.fromhex(x) for x in xs])
    return g, e, fh


def test_em_golden_vectors(orc):
    """tests/golden/em_fixed_tiny.json (tools/gen_golden.py): the exact-sum EM reproduces its committed bits."""
    g, e, fh = _em_golden()
    p = orc.Problem(np.asarray(g["row_ptr"], np.uint64), np.asarray(g["col_idx"], np.uint32), fh(g["l"]), k=np.asarray(g["k"], np.uint32))
    assert any(r["repeated_passes"] > 0 for r in e["runs"])
    for r in e["runs"]:
        mu, it, ll, redo = orc.em_x(p, fh(r["mu_start"]), max_iter=r["sweeps"], epsilon=-1e308)
        assert [float(x).hex() for x in mu] == r["mu"] and float(ll).hex() == r["loglik"] and redo == r["repeated_passes"]
