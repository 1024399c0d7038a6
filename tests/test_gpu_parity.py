"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer outputs and -- because both sides implement the same keyed-stream spec
with once-rounded fp64 arithmetic -- fp64 traces are required to be BIT-IDENTICAL."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_unsorted_rows_still_bit_exact(gpu, orc):
    """Row order only affects speed (LDS window hit rate), never the result."""
    p, aux = orc.synth_problem(R=40000, T=6000, avg_hits=6, seed=3, sort=False)
    mu0, _ = orc.start_values(p)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    s = gpu.Sampler(prob, mu0, seed=8, gibbs_iter=16, trace_len=16)
    s.run(16)
    ref = orc.gibbs_keyed(p, mu0, seed=8, n_iter=16, trace_len=16)
    assert np.array_equal(s.counts(0), ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])


def _mk(orc, R, T, avg, seed=1234, **kw):
    p, aux = orc.synth_problem(R=R, T=T, avg_hits=avg, seed=seed, **kw)
    mu0, uh = orc.start_values(p)
    return p, mu0, uh


def test_device_math_matches_oracle(gpu, orc):
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 300000)), rng.uniform(-745, 709, 300000),
                        rng.uniform(0, 4, 100000), [5e-324, 1e-310, 1.0, 0.5, 2.0, 1e300]])
    r = gpu.selftest_math(x, 0)
    assert np.array_equal(r["log"], orc.log_v(x), equal_nan=True)
    assert np.array_equal(r["exp"], orc.exp_v(x), equal_nan=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        assert np.array_equal(r["sqrt"], np.sqrt(x), equal_nan=True)      # correctly rounded on both
        assert np.array_equal(r["rcp"], 1.0 / x, equal_nan=True)


def test_device_philox_kat(gpu):
    assert [hex(v) for v in gpu.selftest_philox([0, 0, 0, 0], [0, 0], 0)] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8", "0xff1dae59", "0x6cd10df2"]
    assert [hex(v) for v in gpu.selftest_philox([0xffffffff] * 4, [0xffffffff] * 2, 0)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd", "0x2c3f628b", "0xab4fd7ad"]


@pytest.mark.parametrize("shape", [0.1, 0.9, 1.0, 1.1, 7.3, 1234.1])
def test_device_gamma_bit_exact(gpu, orc, shape):
    n = 50000
    got = gpu.selftest_gamma(99, shape, 0.37, n, 0)
    ref = np.empty(n)
    orc.lib().orc_keyed_gamma_v(99, shape, 0.37, n, ref)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("nn,p", [(1, 0.3), (9, 0.5), (40, 0.2), (1000, 0.31), (100000, 0.93), (77, 1e-4)])
def test_device_binomial_bit_exact(gpu, orc, nn, p):
    n = 20000
    got = gpu.selftest_binomial(5, nn, p, n, 0)
    ref = np.empty(n, np.uint32)
    orc.lib().orc_keyed_binomial_v(5, nn, p, n, ref)
    assert np.array_equal(got, ref)


def test_sample_counts_bit_exact_single_sweep(gpu, orc):
    p, mu0, _ = _mk(orc, 50000, 3000, 6)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    s = gpu.Sampler(prob, mu0, seed=42, gibbs_iter=4, trace_len=4)
    s.sample()
    cnt = s.counts(0)
    ref = orc.sample_counts(p, mu0, 42, 0, 0)
    assert np.array_equal(cnt, ref)
    assert int(cnt.sum()) == p.total_k()
    s.update()
    mu1 = s.mu(0)
    assert np.array_equal(mu1, orc.gamma_update(ref, p.l, 0.1, 0.1, 42, 0, 0))


# the three sample kernels: sliced-ELL 8-bit stream (default without multiplicities), 16-bit tile stream, 32-bit CSR tiles
KERNEL_ENVS = {2: {}, 1: {"MMG_K1_SELL": "0"}, 0: {"MMG_K1_SELL": "0", "MMG_K1_S16": "0"}}


@pytest.mark.parametrize("kernel", [2, 1, 0])
@pytest.mark.parametrize("R,T,avg", [(10000, 1000, 4), (30000, 500, 12), (2000, 4000, 2), (70000, 3000, 30)])
def test_full_chain_bit_exact(gpu, orc, monkeypatch, R, T, avg, kernel):
    for k, v in KERNEL_ENVS[kernel].items():
        monkeypatch.setenv(k, v)
    p, mu0, _ = _mk(orc, R, T, avg)                      # rows sorted by (leading transcript, length): the fast layout
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    got = prob.info.sample_kernel                        # a stream kernel is only chosen when >= 90 % of its tiles qualify
    assert got <= kernel and (got == kernel or (R, T, avg) not in [(10000, 1000, 4), (70000, 3000, 30)])
    s = gpu.Sampler(prob, mu0, seed=1234, gibbs_iter=128, trace_len=64)
    s.run(128)
    ref = orc.gibbs_keyed(p, mu0, seed=1234, n_iter=128, trace_len=64)
    assert np.array_equal(s.counts(0), ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])
    assert np.array_equal(s.mu(0), ref["mu"])
    sl, sl2, ns = s.moments(0)
    assert ns == 64
    assert np.array_equal(sl, ref["sum_log"]) and np.array_equal(sl2, ref["sum_log2"])
    rows = s.trace_rows(0)
    assert np.array_equal(rows.T, ref["trace"])


def test_rows_with_multiplicity_bit_exact(gpu, orc):
    """k > 1 rows: k <= 8 -> repeated categorical draws; k > 8 -> conditional-binomial chain."""
    p, mu0, _ = _mk(orc, 5000, 400, 5)
    rng = np.random.default_rng(7)
    k = rng.choice([1, 2, 3, 8, 9, 50, 1000, 20000], size=p.m).astype(np.uint32)
    pk = orc.Problem(p.row_ptr, p.col_idx, p.l * 50, k=k)
    mu0, uh = orc.start_values(pk)
    prob = gpu.Problem.from_csr(pk.row_ptr, pk.col_idx, pk.l, k=pk.k)
    s = gpu.Sampler(prob, mu0, seed=9, gibbs_iter=32, trace_len=32)
    s.run(32)
    ref = orc.gibbs_keyed(pk, mu0, seed=9, n_iter=32, trace_len=32)
    cnt = s.counts(0)
    assert int(cnt.sum()) == pk.total_k()
    assert np.array_equal(cnt, ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])


def test_edge_rows(gpu, orc):
    """Empty rows, single-hit rows, a row longer than a tile (> 4096 hits), ragged tail."""
    T = 6000
    rows = [[], [5], [1, 2], list(range(0, 5000)), [7], [], [3, 4, 5], list(range(100, 4300)), [T - 1, ]]
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    l = np.linspace(0.5, 2.0, T)
    k = np.array([3, 4, 1, 9, 1, 1, 30, 2, 1], np.uint32)
    p = orc.Problem(rp, ci, l, k=k)
    mu0 = np.full(T, 0.25)
    mu0[::7] = 1e-300
    prob = gpu.Problem.from_csr(rp, ci, l, k=k)
    assert prob.info.max_row_len == 5000
    s = gpu.Sampler(prob, mu0, seed=5, gibbs_iter=16, trace_len=16)
    s.run(16)
    ref = orc.gibbs_keyed(p, mu0, seed=5, n_iter=16, trace_len=16)
    assert np.array_equal(s.counts(0), ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])
    assert int(s.counts(0).sum()) == int(k[[1, 2, 3, 4, 6, 7, 8]].sum())


@pytest.mark.parametrize("with_k", [False, True])
def test_sliced_ell_kernel_edge_rows(gpu, orc, monkeypatch, with_k):
    """The sliced-ELL kernel (forced even though few tiles qualify) on the row shapes its format singles out: rows of
    exactly 32 / 33 hits (the register cache holds 8 groups), 255 and 256+ hits (the length byte; longer rows make their
    tile a slow tile), single-hit and empty rows, a window slide inside a workgroup's range, a ragged last tile,
    degenerate weights (all mu of a row zero / infinite) -- bit-exact against the oracle, with and without multiplicities."""
    monkeypatch.setenv("MMG_K1_SELL", "2")
    monkeypatch.setenv("MMG_K1_SELL_WAVES_PER_CU", "1")       # few workgroups: long tile ranges, several window slides each
    rng = np.random.default_rng(9)
    T = 40000
    rows = []
    for lead in range(0, T - 300, 37):                         # leading transcripts ascend: the layout the window wants
        for L in (1, 2, 3, 4, 5, 31, 32, 33, 34, 40, 0, 7):
            rows.append(sorted(rng.choice(np.arange(lead, lead + 120), size=L, replace=False).tolist()) if L else [])
    rows[500] = list(range(18500, 18500 + 254))               # fits the 255-wide window
    rows[501] = list(range(18500, 18500 + 255))               # 255 hits: still a fast row
    rows[900] = list(range(33300, 33300 + 300))               # too long for the length byte: slow tile
    rows[1200] = [5, 39000]                                    # far outside any window: slow tile
    rows = rows[:-13]                                          # ragged last tile
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows if r])
    l = np.linspace(0.5, 2.0, T)
    k = rng.choice([1, 1, 2, 8, 9, 500], size=len(rows)).astype(np.uint32) if with_k else None
    p = orc.Problem(rp, ci, l, k=k)
    mu0 = rng.gamma(0.3, 1.0, size=T)
    mu0[::5] = 1e-300
    mu0[2000:2200] = 0.0                                       # rows whose every weight is zero: uniform pick
    mu0[3000:3100] = np.inf                                    # total not finite: uniform pick as well
    prob = gpu.Problem.from_csr(rp, ci, l, k=k)
    assert prob.info.sample_kernel == 2
    s = gpu.Sampler(prob, mu0, seed=17, gibbs_iter=12, trace_len=12)
    s.sample()
    ref1 = orc.sample_counts(p, mu0, seed=17, chain=0, it=0)
    assert np.array_equal(s.counts(0), ref1)
    s.update()
    s.run(11)
    ref = orc.gibbs_keyed(p, mu0, seed=17, n_iter=12, trace_len=12)
    assert np.array_equal(s.counts(0), ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])


def test_chains_and_shards_reproduce_single_chain(gpu, orc):
    """(a) chain c of a multi-chain sampler == a single-chain sampler with chain_base=c;
    (b) read-sharding: two shards' counts summed (the all-reduce) == the unsharded chain."""
    p, mu0, _ = _mk(orc, 20000, 800, 5)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    multi = gpu.Sampler(prob, mu0, seed=11, n_chains=3, gibbs_iter=32, trace_len=32)
    multi.run(32)
    for c in range(3):
        ref = orc.gibbs_keyed(p, mu0, seed=11, chain=c, n_iter=32, trace_len=32)
        assert np.array_equal(multi.trace(c), ref["trace"])
    # (b) shard rows [0,h) and [h,m) with row_id_base; emulate the int32 all-reduce on the host
    h = p.m // 3
    nz = int(p.row_ptr[h])
    pa = gpu.Problem.from_csr(p.row_ptr[:h + 1], p.col_idx[:nz], p.l)
    pb = gpu.Problem.from_csr(p.row_ptr[h:] - p.row_ptr[h], p.col_idx[nz:], p.l, row_id_base=h)
    ref = orc.gibbs_keyed(p, mu0, seed=11, chain=0, n_iter=1, trace_len=1)
    sa = gpu.Sampler(pa, mu0, seed=11, gibbs_iter=1, trace_len=1)
    sb = gpu.Sampler(pb, mu0, seed=11, gibbs_iter=1, trace_len=1)
    sa.sample(); sb.sample()
    assert np.array_equal(sa.counts(0) + sb.counts(0), ref["cnt"])


def test_device_generator_matches_oracle_generator(gpu, orc):
    for (R, T, avg, uni, row0, srt) in [(30000, 2000, 8, False, 0, True), (5000, 300, 20, False, 12345, True),
                                        (4000, 5000, 3, True, 7, True), (1000, 50, 20, False, 0, False),
                                        (20000, 1000, 8, False, 99, False)]:
        prob = gpu.Problem.synthetic(R, T, avg, seed=1234, row0=row0, uniform=uni, mapped_reads=R, sort=srt)
        rp, ci = prob.download()
        p, aux = orc.synth_problem(R=R, T=T, avg_hits=avg, seed=1234, uniform=uni, row0=row0, sort=srt)
        assert np.array_equal(rp, p.row_ptr)
        assert np.array_equal(ci, p.col_idx)
        assert np.array_equal(prob.l(), p.l)
        lens = np.diff(rp.astype(np.int64))
        assert lens.min() >= 1 and lens.max() <= 100
        # rows ascend strictly (distinct transcripts, sorted)
        d = np.diff(ci.astype(np.int64))
        inner = np.ones(ci.size - 1, bool)
        inner[(rp[1:-1] - 1).astype(np.int64)] = False
        assert (d[inner] > 0).all()


def test_start_values_and_em_match_oracle(gpu, orc):
    p, mu0, uh = _mk(orc, 40000, 1500, 4)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    g_mu0, g_uh = prob.start_values()
    assert np.array_equal(g_uh, uh)                       # integer: bit-exact
    np.testing.assert_allclose(g_mu0, mu0, rtol=1e-12, atol=0)
    em_o, it_o, ll_o = orc.em(p, mu0)
    em_g, it_g, ll_g = prob.em(mu0)
    assert it_g == it_o
    assert ll_g == ll_o                                      # exact fixed-point sums: the log-likelihood too
    assert np.array_equal(em_g, em_o)                        # the EM trajectory is order-independent and bit-exact
    em_g2, _, _ = prob.em(mu0)
    assert np.array_equal(em_g, em_g2)
    # single-sweep calls restart the scale carry-over, on the device as in the oracle
    mu, mo = mu0, mu0
    for _ in range(3):
        mu, it1, _ = prob.em(mu, max_iter=1, epsilon=-1e308)
        mo = orc.em(p, mo, max_iter=1, epsilon=-1e308)[0]
        assert it1 == 1
    assert np.array_equal(mu, mo)
    np.testing.assert_allclose(mu, orc.em(p, mu0, max_iter=3, epsilon=-1e308)[0], rtol=1e-13)


@pytest.mark.parametrize("sort,stream_env,grid", [(True, None, None), (True, None, "7"), (True, "1", None), (True, "1", "7"),
                                                  (True, "0", None), (False, None, None)])
def test_em_stepper_paths_match_oracle(gpu, orc, monkeypatch, sort, stream_env, grid):
    """Tile-stream kernel (sorted rows), row-per-thread kernel (forced / unsorted rows): same bits as the oracle,
    with multiplicities, dead transcripts and an empty row; the oracle in turn tracks the reference's summation order."""
    if stream_env is not None:
        monkeypatch.setenv("MMG_EM_STREAM", stream_env)
    if grid is not None:
        monkeypatch.setenv("MMG_EM_GRID", grid)      # few workgroups: long tile ranges, the window slides many times
    p, aux = orc.synth_problem(R=60000, T=2500, avg_hits=7, seed=11, sort=sort)
    rng = np.random.default_rng(5)
    k = rng.choice([1, 1, 1, 2, 7, 1000], size=p.m).astype(np.uint32)
    rp = np.concatenate([p.row_ptr[:1000], p.row_ptr[999:]])          # an empty row in the middle
    k = np.concatenate([k[:999], [3], k[999:]]).astype(np.uint32)
    pk = orc.Problem(rp, p.col_idx, p.l * 20, k=k)
    mu0, _ = orc.start_values(pk)
    mu0[5::97] = 0.0                                                    # dead from the start
    prob = gpu.Problem.from_csr(pk.row_ptr, pk.col_idx, pk.l, k=pk.k)
    em = prob.em_stepper(mu0)
    st = em.stats()
    want = 0 if (not sort or stream_env == "0") else (1 if stream_env == "1" else 2)
    assert em.stats_raw()["stream_kernel"] == want
    lls = [em.loglik]
    for _ in range(12):
        lls.append(em.step())
    mu_g = em.mu()
    assert em.stats()["sweeps"] == 12
    for it in (0, 1, 12):
        mu_o, _, ll_o = orc.em(pk, mu0, max_iter=it, epsilon=-1e308)
        assert lls[it] == ll_o
    assert np.array_equal(mu_g, mu_o)
    assert (mu_g[5::97] == 0).all()
    em.close()
    # without dead transcripts and the empty row (the reference's arithmetic turns those into NaN), the same device
    # path follows the reference's own summation order to rounding
    mu1, _ = orc.start_values(pk)
    keep = np.ones(pk.m, bool)
    keep[999] = False
    p2 = orc.Problem(p.row_ptr, p.col_idx, pk.l, k=pk.k[keep])
    prob2 = gpu.Problem.from_csr(p2.row_ptr, p2.col_idx, p2.l, k=p2.k)
    mu_g2, it_g2, ll_g2 = prob2.em(mu1, max_iter=12, epsilon=-1e308)
    mu_s, _, ll_s = orc.em_seq(p2, mu1, max_iter=12, epsilon=-1e308)
    np.testing.assert_allclose(mu_g2, mu_s, rtol=1e-11)
    np.testing.assert_allclose(ll_g2, ll_s, rtol=1e-12)
    prob.close()
    prob2.close()


@pytest.mark.parametrize("scale", [1e-200, 1e-30, 1e30, 1e150])
def test_em_wild_start_values_take_the_repeat_path(gpu, orc, scale):
    """Start values spanning hundreds of orders of magnitude force passes to be repeated on measured exponents:
    the device takes the same decisions as the oracle (same bits) and both stay on the reference trajectory."""
    p, aux = orc.synth_problem(R=20000, T=900, avg_hits=6, seed=5)
    mu0, _ = orc.start_values(p)
    mu0[::7] *= scale
    mu0[3] = 0.0
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    em = prob.em_stepper(mu0)
    for _ in range(20):
        em.step()
    mu_o, _, ll_o, redo = orc.em_x(p, mu0, max_iter=20, epsilon=-1e308)
    assert redo >= 1 and em.stats()["repeated_passes"] == redo
    assert np.array_equal(em.mu(), mu_o) and em.loglik == ll_o
    mu_s, _, _ = orc.em_seq(p, mu0, max_iter=20, epsilon=-1e308)
    live = mu_s > 0
    np.testing.assert_allclose(mu_o[live], mu_s[live], rtol=1e-11)
    assert np.array_equal(mu_o == 0, mu_s == 0)


def test_errors_are_loud(gpu):
    with pytest.raises(Exception):
        gpu.Problem.from_csr(np.array([0, 2], np.uint64), np.array([0, 9], np.uint32), np.ones(3))
    with pytest.raises(Exception):
        gpu.Problem.from_csr(np.array([0, 1], np.uint64), np.array([0], np.uint32), np.array([1.0, 0.0]))
    prob = gpu.Problem.from_csr(np.array([0, 1], np.uint64), np.array([0], np.uint32), np.ones(2))
    with pytest.raises(Exception):
        gpu.Sampler(prob, np.ones(2), gibbs_iter=100, trace_len=64)
    s = gpu.Sampler(prob, np.ones(2), gibbs_iter=4, trace_len=4, keep_trace=False)
    s.run(4)
    with pytest.raises(Exception):
        s.trace(0)
    with pytest.raises(Exception):
        s.update()


def test_golden_tiny_chain_on_device(gpu):
    """The committed golden fixture (tests/golden/keyed_chain_tiny.json): k up to 1000, all three row paths."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keyed_chain_tiny.json")))
    f = lambda hs: np.array([float.fromhex(h) for h in hs])
    prob = gpu.Problem.from_csr(np.array(g["row_ptr"], np.uint64), np.array(g["col_idx"], np.uint32), f(g["l"]),
                                k=np.array(g["k"], np.uint32))
    mu0, uh = prob.start_values()
    assert uh.tolist() == g["unique_hits"]
    np.testing.assert_allclose(mu0, f(g["mu0"]), rtol=1e-14)
    s = gpu.Sampler(prob, f(g["mu0"]), seed=1234, gibbs_iter=32, trace_len=16)
    s.run(32)
    assert np.array_equal(s.trace(0).ravel(), f(g["trace"]))
    assert s.counts(0).tolist() == g["cnt_last"]
    assert np.array_equal(s.mu(0), f(g["mu_last"]))


def test_golden_em_on_device(gpu):
    """tests/golden/em_fixed_tiny.json on the device: mu, log-likelihood and the repeat count, bit for bit."""
    gd = os.path.join(os.path.dirname(__file__), "golden")
    g = json.load(open(os.path.join(gd, "keyed_chain_tiny.json")))
    e = json.load(open(os.path.join(gd, "em_fixed_tiny.json")))
    fh = lambda xs: np.array([float.fromhex(x) for x in xs])
    prob = gpu.Problem.from_csr(np.asarray(g["row_ptr"], np.uint64), np.asarray(g["col_idx"], np.uint32), fh(g["l"]),
                                k=np.asarray(g["k"], np.uint32))
    for r in e["runs"]:
        em = prob.em_stepper(fh(r["mu_start"]))
        for _ in range(r["sweeps"]):
            em.step()
        assert [float(x).hex() for x in em.mu()] == r["mu"] and float(em.loglik).hex() == r["loglik"]
        assert em.stats()["repeated_passes"] == r["repeated_passes"]
        em.close()


def test_torch_view_of_device_buffers_and_single_rank_collectives(gpu, orc):
    """dist glue on one GPU: zero-copy torch views of the library's buffers, kernels on torch's stream,
    a 1-rank RCCL process group around the shard-mode step."""
    import time
    t0 = time.time()
    def stamp(what):  # phase timings: stdout (-s / on failure) and, when present, the scratch directory of the GPU run
        line = "[torch_view] %-36s %.2fs" % (what, time.time() - t0)
        print(line, flush=True)
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "torch_view_timing.log"), "a") as f:
                f.write(line + "\n")
    import torch
    import torch.distributed as dist
    from mmseq_amd import dist as mdist
    stamp("imports")
    p, aux = orc.synth_problem(R=20000, T=900, avg_hits=6, seed=5)
    mu0, _ = orc.start_values(p)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    s = gpu.Sampler(prob, mu0, seed=3, gibbs_iter=8, trace_len=8)
    mdist.use_current_stream(s)
    counts = mdist.counts_tensor(s)
    assert counts.dtype == torch.int32 and counts.numel() == 900 and counts.is_cuda
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        stamp("before init_process_group")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        stamp("init_process_group")
    try:
        for _ in range(8):
            s.sample()
            dist.all_reduce(counts)            # world size 1: identity, but exercises RCCL on the same stream
            s.update()
        stamp("8 x sample/all_reduce/update issued")
        torch.cuda.synchronize()
        stamp("synchronize")
        mom = mdist.moments_tensor(s)
        mdist.pool_moments(mom)
        stamp("pool_moments")
    finally:
        if own:
            dist.destroy_process_group()
            stamp("destroy_process_group")
    ref = orc.gibbs_keyed(p, mu0, seed=3, n_iter=8, trace_len=8)
    assert np.array_equal(s.trace(0), ref["trace"])
    assert np.array_equal(mom.cpu().numpy()[:900], ref["sum_log"])


def test_64bit_row_offsets_path(gpu, orc, monkeypatch):
    """nnz >= 2^32 switches the device row_ptr to u64; MMG_FORCE_IDX64 exercises that instantiation on a small problem."""
    monkeypatch.setenv("MMG_FORCE_IDX64", "1")
    p, mu0, uh = _mk(orc, 30000, 900, 7)
    rng = np.random.default_rng(2)
    k = rng.choice([1, 1, 1, 2, 9, 300], size=p.m).astype(np.uint32)
    pk = orc.Problem(p.row_ptr, p.col_idx, p.l * 10, k=k)
    mu0, _ = orc.start_values(pk)
    prob = gpu.Problem.from_csr(pk.row_ptr, pk.col_idx, pk.l, k=pk.k)
    assert prob.info.index_bits == 64
    g_mu0, g_uh = prob.start_values()
    np.testing.assert_allclose(g_mu0, mu0, rtol=1e-12)
    em_g, it_g, _ = prob.em(mu0)
    em_o, it_o, _ = orc.em(pk, mu0)
    assert it_g == it_o and np.array_equal(em_g, em_o)
    s = gpu.Sampler(prob, mu0, seed=4, gibbs_iter=32, trace_len=32)
    s.run(32)
    ref = orc.gibbs_keyed(pk, mu0, seed=4, n_iter=32, trace_len=32)
    assert np.array_equal(s.trace(0), ref["trace"]) and np.array_equal(s.counts(0), ref["cnt"])
    rp, ci = prob.download()
    assert np.array_equal(rp, pk.row_ptr) and np.array_equal(ci, pk.col_idx)
    syn = gpu.Problem.synthetic(5000, 300, 6, seed=1234)
    assert syn.info.index_bits == 64
    q, _ = orc.synth_problem(R=5000, T=300, avg_hits=6, seed=1234)
    rp, ci = syn.download()
    assert np.array_equal(rp, q.row_ptr) and np.array_equal(ci, q.col_idx)


@pytest.mark.parametrize("n_chains", [2, 4, 8, 11])
def test_fused_chains_equal_independent_single_chains(gpu, orc, monkeypatch, n_chains):
    """Chains advanced together by the fused walk (groups of 8/4/2/1) are bit-identical to single-chain runs."""
    monkeypatch.setenv("MMG_K1_SELL", "0")               # the fused walk lives in k_sample16
    p, mu0, _ = _mk(orc, 40000, 1500, 9)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    s = gpu.Sampler(prob, mu0, seed=21, n_chains=n_chains, chain_base=3, gibbs_iter=16, trace_len=16)
    s.run(16)
    for c in sorted({0, 1, n_chains // 2, n_chains - 1}):
        ref = orc.gibbs_keyed(p, mu0, seed=21, chain=3 + c, n_iter=16, trace_len=16)
        assert np.array_equal(s.trace(c), ref["trace"]), c
        assert np.array_equal(s.counts(c), ref["cnt"]), c
