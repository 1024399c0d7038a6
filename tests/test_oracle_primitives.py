"""Pins the oracle's primitives: Random123 / MT19937 known answers, fdlibm-style log/exp accuracy,
and the samplers' distributions against scipy (the reference reaches these through GSL, which is
not in its tree: gsl_ran_gamma src/mmseq.cpp:907, gsl_ran_multinomial :880, gsl_rng_mt19937 :836)."""
import numpy as np
import pytest
from scipy import stats


def test_philox_random123_known_answers(orc):
    # Random123 kat_vectors for philox4x32-10
    assert [hex(v) for v in orc.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in orc.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox2x32_random123_known_answers(orc):
    assert [hex(v) for v in orc.philox2x32([0, 0], 0)] == ["0xff1dae59", "0x6cd10df2"]
    assert [hex(v) for v in orc.philox2x32([0xffffffff, 0xffffffff], 0xffffffff)] == ["0x2c3f628b", "0xab4fd7ad"]
    assert [hex(v) for v in orc.philox2x32([0x243f6a88, 0x85a308d3], 0x13198a2e)] == ["0xdd7ce038", "0xf62a4c12"]


def test_mt19937_known_answers(orc):
    o = orc.mt19937(5489, 10000)
    assert o[0] == 3499211612 and o[9999] == 4123659995   # C++11 [rand.predef] mt19937 10000th value
    assert orc.mt19937(0, 1)[0] == orc.mt19937(4357, 1)[0]  # GSL maps seed 0 to 4357


def test_log_exp_within_one_ulp_of_libm(orc):
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(-700, 700, 400000))
    got, ref = orc.log_v(x), np.log(x)
    assert (np.abs(got - ref) <= np.spacing(np.abs(ref))).all()
    y = rng.uniform(-745, 709, 400000)
    got, ref = orc.exp_v(y), np.exp(y)
    assert (np.abs(got - ref) <= np.spacing(ref)).all()
    sp = orc.log_v(np.array([0.0, 1.0, np.inf, 5e-324]))
    assert sp[0] == -np.inf and sp[1] == 0.0 and sp[2] == np.inf and abs(sp[3] - np.log(5e-324)) < 1e-12
    assert np.isnan(orc.log_v(np.array([-1.0]))[0])
    assert orc.exp_v(np.array([-800.0]))[0] == 0.0 and orc.exp_v(np.array([800.0]))[0] == np.inf


def _ks(sample, dist):
    return stats.kstest(sample, dist.cdf).pvalue


@pytest.mark.parametrize("shape", [0.1, 0.5, 1.0, 1.1, 3.7, 250.1])
def test_gamma_samplers_match_scipy(orc, shape):
    n = 200000
    a = np.empty(n)
    orc.lib().orc_keyed_gamma_v(11, shape, 2.5, n, a)
    b = np.empty(n)
    orc.lib().orc_mt_gamma_v(11, shape, 2.5, n, b)
    d = stats.gamma(shape, scale=2.5)
    for s in (a, b):
        assert _ks(s, d) > 1e-4
        assert abs(s.mean() - d.mean()) < 6 * d.std() / np.sqrt(n)
        # log-moments: what the posterior summary actually uses (E log mu = psi(a) + log scale)
        from scipy.special import digamma, polygamma
        lg = np.log(s)
        assert abs(lg.mean() - (digamma(shape) + np.log(2.5))) < 6 * np.sqrt(polygamma(1, shape) / n)


def test_keyed_normal_is_standard_normal(orc):
    n = 400000
    z = np.empty(n)
    orc.lib().orc_keyed_normal_v(3, n, z)
    assert _ks(z, stats.norm()) > 1e-4
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.01


@pytest.mark.parametrize("nn,p", [(1, 0.3), (7, 0.5), (30, 0.2), (30, 0.9), (1000, 0.02), (1000, 0.4), (100000, 0.77),
                                  (12, 1e-3)])
def test_binomial_samplers_match_scipy(orc, nn, p):
    n = 200000
    for fn, seed in ((orc.lib().orc_keyed_binomial_v, 5), (orc.lib().orc_mt_binomial_v, 5)):
        x = np.empty(n, np.uint32)
        fn(seed, nn, p, n, x)
        assert x.max() <= nn
        d = stats.binom(nn, p)
        assert abs(x.mean() - d.mean()) < 6 * d.std() / np.sqrt(n) + 1e-12
        assert abs(x.var() - d.var()) < 0.05 * d.var() + 1e-3
        # chi-square on the bulk of the support
        lo, hi = int(max(0, d.ppf(1e-4))), int(min(nn, d.ppf(1 - 1e-4)))
        obs = np.bincount(np.clip(x, lo, hi), minlength=hi + 1)[lo:hi + 1].astype(float)
        pm = d.pmf(np.arange(lo, hi + 1))
        pm[0] += d.cdf(lo - 1)
        pm[-1] += d.sf(hi)
        exp = pm * n
        keep = exp > 5
        if keep.sum() > 1:
            obs2 = np.append(obs[keep], obs[~keep].sum())
            exp2 = np.append(exp[keep], exp[~keep].sum())
            if exp2[-1] == 0:
                obs2, exp2 = obs2[:-1], exp2[:-1]
            chi = ((obs2 - exp2) ** 2 / exp2).sum()
            assert stats.chi2(len(exp2) - 1).sf(chi) > 1e-5


def test_binomial_edge_cases(orc):
    x = np.empty(10, np.uint32)
    orc.lib().orc_keyed_binomial_v(1, 17, 0.0, 10, x)
    assert (x == 0).all()
    orc.lib().orc_keyed_binomial_v(1, 17, 1.0, 10, x)
    assert (x == 17).all()
    orc.lib().orc_keyed_binomial_v(1, 0, 0.5, 10, x)
    assert (x == 0).all()


def test_row_stream_pairs_share_a_philox_block(orc):
    """Row stream spec (mmseq_amd/csrc/mmg_math.h:Stream2): rows 2q and 2q+1 take the two output words of ONE Philox2x32-10 block
    keyed (seed, chain, tag, q >> 32) with counter (q, iteration), as 32-bit uniforms (x + 1/2) 2^-32.  Observed through the keyed
    Binomial(1, 1/2) hook, which is [u > 1/2] of the row's first uniform (sequential-search inversion)."""
    import numpy as np
    n, seed = 4000, 99
    got = np.empty(n, np.uint32)
    orc.lib().orc_keyed_binomial_v(seed, 1, 0.5, n, got)                    # rows 0..n-1 of stream (seed, chain 0, tag ROW = 1, iteration 0)
    exp = np.empty(n, np.uint32)
    for rid in range(n):
        q = rid >> 1
        key = ((seed & 0xffffffff) ^ (((seed >> 32) * 0x9E3779B1) & 0xffffffff) ^ ((1 << 28) & 0xffffffff) ^ (((q >> 32) * 0xC2B2AE35) & 0xffffffff))
        w = orc.philox2x32([q & 0xffffffff, 0], key & 0xffffffff)
        u = (float(w[rid & 1]) + 0.5) * 2.0 ** -32
        exp[rid] = 1 if u > 0.5 else 0
    assert np.array_equal(got, exp)
    assert 0.45 < got.mean() < 0.55


def test_canonical_layout_properties(orc):
    """oracle.binding.canonical_layout (the restatement of mmseq_amd/csrc/layout.hip): keys ascend, the order is idempotent and does
    not depend on the order the rows were given in; empty rows come first, far rows last."""
    import numpy as np
    p, _ = orc.synth_problem(R=8000, T=900, avg_hits=6, seed=3, sort=False, far_fraction=0.05)
    rng = np.random.default_rng(0)
    k = rng.choice([1, 1, 2, 9, 70], size=p.m).astype(np.uint32)
    rp, ci, kk, perm = orc.canonical_layout(p.row_ptr, p.col_idx, k)
    # rows that draw 2 <= k <= min(64, 16 (hits - 1)) categoricals are stored as k rows with k = 1 (ABI 4, spec version 8); the rows of the
    # conditional-binomial chain and the rows of one hit keep their multiplicity
    m_st = rp.size - 1
    L = np.diff(p.row_ptr.astype(np.int64))
    reps = np.where((k >= 2) & (k <= np.minimum(64, 16 * (L - 1))), k, 1)
    assert ((reps == 1) & (k == 9)).any() and ((reps == 9) & (k == 9)).any()         # (k = 9: stored 9 times, except on rows of one hit)
    assert m_st == int(reps.sum()) and set(np.unique(kk)) == {1, 2, 9, 70}
    assert int(kk.astype(np.int64).sum()) == int(k.astype(np.int64).sum())
    assert np.array_equal(np.bincount(perm, minlength=p.m), reps)   # perm[stored row] = caller row
    assert np.array_equal(np.diff(rp.astype(np.int64)), np.diff(p.row_ptr.astype(np.int64))[perm])
    key, h = orc.row_keys(rp, orc.sort_hits(rp, ci), kk)   # key and tie word are functions of the SET of hits
    assert (key[1:] >= key[:-1]).all()
    same = key[1:] == key[:-1]
    assert (h[1:][same] >= h[:-1][same]).all()
    rp2, ci2, kk2, perm2 = orc.canonical_layout(rp, ci, kk)
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(kk2, kk) and np.array_equal(perm2, np.arange(m_st))
    sh = rng.permutation(p.m)
    rps, cis, ks = orc.permute_rows(p.row_ptr, p.col_idx, k, sh)
    rp3, ci3, kk3, _ = orc.canonical_layout(rps, cis, ks)
    assert np.array_equal(rp3, rp) and np.array_equal(ci3, ci) and np.array_equal(kk3, kk)
    _, _, k_none, _ = orc.canonical_layout(p.row_ptr, p.col_idx, np.where(L >= 2, np.minimum(k, 9), 1).astype(np.uint32))
    assert k_none is None                                  # every multiplicity expanded: an array of ones is no array
    far = key >> np.uint64(63)
    assert 0 < far.sum() < m_st and (np.diff(far.astype(np.int64)) >= 0).all()
    # a far row is stored with the hits inside its home window first, both parts ascending; every other row ascending
    wbase = (((key >> np.uint64(18)) & np.uint64((1 << 45) - 1)) << np.uint64(6)).astype(np.int64)
    n_far_hits = 0
    for r in range(m_st):
        row = ci[int(rp[r]):int(rp[r + 1])].astype(np.int64)
        if not far[r]:
            assert (np.diff(row) > 0).all()
            continue
        inside = (row >= wbase[r]) & (row < wbase[r] + 255)
        nin = int(inside.sum())
        assert inside[:nin].all() and (np.diff(row[:nin]) > 0).all() and (np.diff(row[nin:]) > 0).all()
        n_far_hits += row.size - nin
    assert n_far_hits > 0
