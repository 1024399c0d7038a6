"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer outputs and -- because both sides implement the same keyed-stream spec
with once-rounded fp64 arithmetic -- fp64 traces are required to be BIT-IDENTICAL."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(gpu, orc, p, **kw):
    """Uploads oracle problem p; returns (device problem, the oracle's view of it IN STORED ORDER): the library keeps rows in
    its own canonical order, and a checker replays the chain row by row from mmg_problem_download."""
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, k=p.k, **kw)
    rp, ci, k = prob.download(with_k=True)
    return prob, orc.Problem(rp, ci, p.l, k=(k if p.k is not None else None))


def test_kept_unsorted_rows_still_bit_exact(gpu, orc):
    """MMG_LAYOUT_KEEP_ROWS with rows in generator order: row order only affects speed (LDS window hit rate), never the result."""
    p, aux = orc.synth_problem(R=40000, T=6000, avg_hits=6, seed=3, sort=False)
    mu0, _ = orc.start_values(p)
    for kern in (-1, 0):
        with gpu.options(sample_kernel=kern):
            prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, keep_rows=True)
            rp, ci = prob.download()
            assert np.array_equal(rp, p.row_ptr) and np.array_equal(ci, p.col_idx) and prob.info.layout == 1
            s = gpu.Sampler(prob, mu0, seed=8, gibbs_iter=16, trace_len=16)
            s.run(16)
            ref = orc.gibbs_keyed(p, mu0, seed=8, n_iter=16, trace_len=16)
            assert np.array_equal(s.counts(0), ref["cnt"])
            assert np.array_equal(s.trace(0), ref["trace"])


def test_canonical_layout_is_the_oracles_restatement_and_ignores_the_upload_order(gpu, orc):
    """mmg_problem_create stores rows sorted by (near/far, band, multiplicity class, length, content hash): the download equals
    oracle.binding.canonical_layout, and any permutation of the same rows -- the reference's first-seen order
    (src/mmseq.cpp:409-418) is just one -- yields the same stored problem, hence the same chain."""
    p, aux = orc.synth_problem(R=30000, T=3000, avg_hits=7, seed=5, sort=False, far_fraction=0.1)
    rng = np.random.default_rng(1)
    k = rng.choice([1, 1, 1, 2, 5, 9, 300], size=p.m).astype(np.uint32)
    rp_c, ci_c, k_c, perm = orc.canonical_layout(p.row_ptr, p.col_idx, k)
    mu0, _ = orc.start_values(orc.Problem(p.row_ptr, p.col_idx, p.l, k=k))
    traces = []
    for trial in range(3):
        if trial == 0:
            rp, ci, kk = p.row_ptr, p.col_idx, k
        else:
            sh = rng.permutation(p.m)
            rp, ci, kk = orc.permute_rows(p.row_ptr, p.col_idx, k, sh)
        prob = gpu.Problem.from_csr(rp, ci, p.l, k=kk)
        d_rp, d_ci, d_k = prob.download(with_k=True)
        assert np.array_equal(d_rp, rp_c) and np.array_equal(d_ci, ci_c) and np.array_equal(d_k, k_c)
        assert prob.info.sample_kernel == 2 and prob.info.layout == 0
        s = gpu.Sampler(prob, mu0, seed=8, gibbs_iter=8, trace_len=8)
        s.run(8)
        traces.append((s.trace(0), s.counts(0)))
        s.close(); prob.close()
    for t, c in traces[1:]:
        assert np.array_equal(t, traces[0][0]) and np.array_equal(c, traces[0][1])
    ref = orc.gibbs_keyed(orc.Problem(rp_c, ci_c, p.l, k=k_c), mu0, seed=8, n_iter=8, trace_len=8)
    assert np.array_equal(traces[0][0], ref["trace"]) and np.array_equal(traces[0][1], ref["cnt"])


def test_reads_that_also_hit_a_paralogue_family_get_a_gene_order_that_brings_the_family_together(gpu, orc):
    """Spec version 7.  An aligner's output: a read's hits are isoforms of ONE gene, and a fifth of the reads also hit an isoform of a
    paralogue -- a gene of a small fixed family whose other members lie anywhere in the caller's gene order (the CLI's is the name
    order of a std::map, src/mmseq.cpp:337-357).  The generator's gene-block mode makes such rows (device = oracle restatement, bit
    for bit); with tx_order = gene << 32 | transcript the library looks at which GENES share rows, reorders the genes so that a family
    is contiguous (the isoforms of a gene stay together, in the caller's order), and the far rows disappear.  Everything crossing
    the ABI stays in the caller's numbering; the chain is the oracle's replay of the downloaded rows; the order is a function of the
    SET of rows."""
    R, T, G, F = 60000, 6000, 24, 3
    p, _ = orc.synth_problem(R=R, T=T, avg_hits=10, seed=19, sort=False, far_fraction=0.2, gene_size=G, far_family=F)
    dev = gpu.Problem.synthetic(R, T, 10, seed=19, sort=False, far_fraction=0.2, gene_size=G, far_family=F)
    d_rp, d_ci = dev.download()
    assert np.array_equal(d_rp, p.row_ptr) and np.array_equal(d_ci, p.col_idx)       # the generator on the device = its restatement
    dev.close()
    rp = p.row_ptr.astype(np.int64)
    gene = (p.col_idx // G).astype(np.int64)
    two = np.minimum.reduceat(gene, rp[:-1]) != np.maximum.reduceat(gene, rp[:-1])
    assert 0.15 < two.mean() < 0.25                                                  # a fifth of the reads hit a second gene
    tx_order = ((np.arange(T, dtype=np.uint64) // np.uint64(G)) << np.uint64(32)) | np.arange(T, dtype=np.uint64)
    with gpu.options(derive_order=0):                                                # spec version 6: the caller's gene order is final
        v6 = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, tx_order=tx_order)
        i6 = v6.info
        assert i6.tx_renumbered == 1 and i6.far_tiles > 0.15 * i6.n_tiles
        v6.close()
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, tx_order=tx_order)
    inf = prob.info
    assert inf.tx_renumbered == 3 and inf.sample_kernel == 2
    assert inf.far_tiles <= 0.05 * inf.n_tiles, (inf.far_tiles, inf.n_tiles)        # a family of 3 genes x 24 isoforms fits an LDS window
    perm = prob.tx_perm()                                                            # device id of the caller's transcript t
    assert np.array_equal(np.sort(perm), np.arange(T))
    for g in range(0, T // G, 17):                                                   # a gene's isoforms: contiguous, in the caller's order
        assert np.array_equal(perm[g * G:(g + 1) * G], perm[g * G] + np.arange(G))
    q_rp, q_ci = prob.download()                                                     # stored rows, the caller's numbering
    ci_q = perm[p.col_idx]
    for r in range(p.m):
        ci_q[rp[r]:rp[r + 1]].sort()
    cq_rp, cq_ci, _, _ = orc.canonical_layout(p.row_ptr, ci_q)
    assert np.array_equal(q_rp, cq_rp) and np.array_equal(perm[q_ci], cq_ci)         # = the canonical layout of the renumbered rows
    mu0, _ = prob.start_values()
    assert np.array_equal(mu0, orc.start_values_exact(orc.Problem(q_rp, q_ci, p.l)))
    s = gpu.Sampler(prob, mu0, seed=3, n_chains=2, gibbs_iter=8, trace_len=8)
    s.run(8)
    for c in range(2):
        ref = orc.gibbs_keyed(orc.Problem(q_rp, q_ci, p.l), mu0, seed=3, chain=c, n_iter=8, trace_len=8)
        assert np.array_equal(s.trace(c), ref["trace"]) and np.array_equal(s.counts(c), ref["cnt"])
    mu_e, it_e, ll_e = prob.em(mu0, max_iter=12, epsilon=-1e308)
    mu_eo, _, ll_eo = orc.em(orc.Problem(q_rp, q_ci, p.l), mu0, max_iter=12, epsilon=-1e308)
    assert np.array_equal(mu_e, mu_eo) and ll_e == ll_eo
    shuffled = np.random.default_rng(8).permutation(p.m)
    s_rp, s_ci, _ = orc.permute_rows(p.row_ptr, p.col_idx, None, shuffled)
    again = gpu.Problem.from_csr(s_rp, s_ci, p.l, tx_order=tx_order)
    a_rp, a_ci = again.download()
    assert np.array_equal(again.tx_perm(), perm) and np.array_equal(a_rp, q_rp) and np.array_equal(a_ci, q_ci)
    # far hits that go ANYWHERE (far_family = 0) link every gene to every other: no order helps, the caller's stays
    pu, _ = orc.synth_problem(R=R, T=T, avg_hits=10, seed=19, sort=False, far_fraction=0.2, gene_size=G, far_family=0)
    uni = gpu.Problem.from_csr(pu.row_ptr, pu.col_idx, pu.l, tx_order=tx_order)
    assert uni.info.tx_renumbered == 1
    uni.close(); again.close(); s.close(); prob.close()


def test_power_law_families_and_hub_transcripts_get_a_gene_order_that_keeps_neighbours_together(gpu, orc):
    """A hit graph with the tail of a real transcriptome (tools/families.py; src/bam2hits.cpp:271-300 keeps up to 100 alignments per
    read): paralogue families of power-law size -- up to 5 000 transcripts, far more than an LDS window -- scattered over the caller's gene
    order, a read's second gene a neighbour in its family's chain, and 1 % of the reads on hub transcripts that share rows with hundreds of
    genes.  Spec version 7's gene order from the group-level hit graph lays a large family out as a run of windows (breadth-first levels
    from a pseudo-peripheral member; the hubs are left out of the traversal): nearly all paralogue rows become near rows, the decision
    is reported (tx_renumbered 3), and the chain on the stored rows is the oracle's bit for bit."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import families as fam
    R, T, G = 60000, 12000, 16
    gen = gpu.Problem.synthetic(R, T, 6.0, seed=77, sort=False, gene_size=G)
    rp, ci = gen.download()
    l = gen.l()
    gen.close()
    rp2, ci2, tx_order, info = fam.power_law_families(rp, ci, T, G, seed=5, max_family_transcripts=5000, n_hubs=3)   # (3 hubs at this size: 200 reads each)
    assert info["largest_family_transcripts"] >= 1000 and 0.1 < info["paralogue_reads"] < 0.25 and 0.005 < info["hub_reads"] < 0.02
    def far_rows(pr):
        # share of the stored rows that do not fit one window in the DEVICE numbering (mmg_types.h: near).  (The tile counts of
        # mmg_problem_info say little at this size: every band with a far row has a far tile, however few rows it holds.)
        d_rp, d_ci = pr.download()
        dev = pr.tx_perm()[d_ci].astype(np.int64)
        st = d_rp[:-1].astype(np.int64)
        lo, hi = np.minimum.reduceat(dev, st), np.maximum.reduceat(dev, st)
        return float((hi - (lo & ~63) >= 240).mean())
    with gpu.options(derive_order=0):
        plain = gpu.Problem.from_csr(rp2, ci2, l, tx_order=tx_order)        # the caller's gene order: a paralogue read spans the transcriptome
        i0, far0 = plain.info, far_rows(plain)
        plain.close()
    prob = gpu.Problem.from_csr(rp2, ci2, l, tx_order=tx_order)
    inf, far1 = prob.info, far_rows(prob)
    assert (i0.tx_renumbered & 0xff) == 1 and far0 > 0.12
    assert (inf.tx_renumbered & 0xff) == 3 and inf.sample_kernel == 2 and far1 < 0.04, (far0, far1, inf.far_tiles, i0.far_tiles, inf.n_tiles)
    q_rp, q_ci = prob.download()
    p = orc.Problem(q_rp, q_ci, l)
    mu0, _ = prob.start_values()
    assert np.array_equal(mu0, orc.start_values_exact(p))
    s = gpu.Sampler(prob, mu0, seed=3, n_chains=2, gibbs_iter=6, trace_len=6)
    s.run(6)
    for c in range(2):
        ref = orc.gibbs_keyed(p, mu0, seed=3, chain=c, n_iter=6, trace_len=6)
        assert np.array_equal(s.counts(c), ref["cnt"]) and np.array_equal(s.trace(c), ref["trace"])
    s.close(); prob.close()


def test_an_order_that_cannot_be_tried_is_flagged_and_an_error_when_forced(gpu, orc):
    """the order from the hit graph (spec versions 6 / 7) is built beside the problem it may replace and needs about three
    times its device memory free.  Without that memory the caller's order stands -- a different stored order, hence a different chain,
    than the same call with memory to spare: mmg_problem_info.tx_renumbered then carries MMG_ORDER_SKIPPED (0x100), and
    MMG_OPT_DERIVE_ORDER = 1 ("always") turns the silent fallback into an error."""
    import torch
    from mmseq_amd._lib import MMGError
    p, _ = orc.synth_problem(R=30000, T=3000, avg_hits=6, seed=12, sort=False)
    rng = np.random.default_rng(2)
    scat = rng.permutation(p.n).astype(np.uint32)            # first-seen numbering: no locality in the caller's ids, no tx_order
    ci_ext = scat[p.col_idx]
    l_ext = np.empty(p.n); l_ext[scat] = p.l
    roomy = gpu.Problem.from_csr(p.row_ptr, ci_ext, l_ext)
    assert roomy.info.tx_renumbered == 2                       # with memory to spare: derived, no flag
    roomy.close()
    torch.cuda.synchronize()
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(free - (56 << 20), 0), dtype=torch.uint8, device="cuda")   # leave 56 MB: enough for the problem, not for the attempt (64 MB + ...)
    try:
        tight = gpu.Problem.from_csr(p.row_ptr, ci_ext, l_ext)
        assert tight.info.tx_renumbered == 0x100 and tight.info.sample_kernel == 0   # the caller's order, flagged
        tight.close()
        with gpu.options(derive_order=1):
            with pytest.raises(MMGError) as e:
                gpu.Problem.from_csr(p.row_ptr, ci_ext, l_ext)
        assert "free device memory" in str(e.value)
    finally:
        del hog
        torch.cuda.empty_cache()


def test_first_seen_numbering_with_tx_order_takes_the_fast_kernel(gpu, orc):
    """The reference numbers transcripts in first-seen order (src/mmseq.cpp:399-408), which scatters the isoforms of a gene over
    the index range.  Uploaded like that the rows span the whole range and only the CSR kernel applies; with tx_order (gene
    ordinal << 32 | ordinal within the gene) the library renumbers internally, runs the sliced-ELL kernel, and every array
    comes back in the caller's numbering -- the same numbers the oracle produces from the downloaded rows, and the same trace
    as an upload that did the sorting itself."""
    p, aux = orc.synth_problem(R=40000, T=4000, avg_hits=6, seed=9, sort=False)
    rng = np.random.default_rng(4)
    scat = rng.permutation(p.n).astype(np.uint32)            # caller id of generator transcript g: "first-seen" scatter
    ci_ext = scat[p.col_idx]
    rp = p.row_ptr.astype(np.int64)
    for r in range(p.m):                                     # rows ascend in the CALLER's numbering, as at src/mmseq.cpp:412
        ci_ext[rp[r]:rp[r + 1]].sort()
    l_ext = np.empty(p.n); l_ext[scat] = p.l
    tx_order = np.empty(p.n, np.uint64)
    tx_order[scat] = (np.arange(p.n, dtype=np.uint64) // np.uint64(7)) << np.uint64(32) | (np.arange(p.n, dtype=np.uint64) % np.uint64(7))
    with gpu.options(derive_order=0):                        # what every caller without tx_order got up to spec version 5
        plain = gpu.Problem.from_csr(p.row_ptr, ci_ext, l_ext)
        assert plain.info.sample_kernel == 0 and plain.info.tx_renumbered == 0
        plain.close()
    # spec version 6: the library derives an order from the hit graph -- which transcripts share rows -- and uses it like a caller's
    # tx_order: the stream kernel, nearly every tile on the register path, every array still in the caller's numbering, the chain the
    # oracle's replay of the downloaded rows; and the order is a function of the SET of rows (any upload order, the same problem)
    plain = gpu.Problem.from_csr(p.row_ptr, ci_ext, l_ext)
    pin = plain.info
    assert pin.sample_kernel == 2 and pin.tx_renumbered == 2 and pin.fast_tiles >= 0.9 * pin.n_tiles, (pin.fast_tiles, pin.far_tiles, pin.n_tiles)
    perm = plain.tx_perm()
    assert np.array_equal(np.sort(perm), np.arange(p.n))
    q_rp, q_ci = plain.download()
    ci_q = perm[ci_ext]
    for r in range(p.m):
        ci_q[rp[r]:rp[r + 1]].sort()
    cq_rp, cq_ci, _, _ = orc.canonical_layout(p.row_ptr, ci_q)
    assert np.array_equal(q_rp, cq_rp) and np.array_equal(perm[q_ci], cq_ci)       # the canonical layout of the renumbered rows
    mu_q, _ = plain.start_values()
    sq = gpu.Sampler(plain, mu_q, seed=3, gibbs_iter=8, trace_len=8)
    sq.run(8)
    refq = orc.gibbs_keyed(orc.Problem(q_rp, q_ci, l_ext), mu_q, seed=3, n_iter=8, trace_len=8)
    assert np.array_equal(sq.trace(0), refq["trace"]) and np.array_equal(sq.counts(0), refq["cnt"])
    shuffled = rng.permutation(p.m)
    s_rp, s_ci, _ = orc.permute_rows(p.row_ptr, ci_ext, None, shuffled)
    again = gpu.Problem.from_csr(s_rp, s_ci, l_ext)
    a_rp, a_ci = again.download()
    assert np.array_equal(again.tx_perm(), perm) and np.array_equal(a_rp, q_rp) and np.array_equal(a_ci, q_ci)
    sq.close(); again.close(); plain.close()
    prob = gpu.Problem.from_csr(p.row_ptr, ci_ext, l_ext, tx_order=tx_order)
    inf = prob.info
    assert inf.sample_kernel == 2 and inf.tx_renumbered == 1 and inf.fast_tiles == inf.n_tiles
    int_of_ext = prob.tx_perm()
    assert np.array_equal(np.argsort(tx_order, kind="stable"), np.argsort(int_of_ext, kind="stable"))
    d_rp, d_ci = prob.download()
    pd = orc.Problem(d_rp, d_ci, l_ext)
    # stored rows walk their hits in ascending DEVICE id and are the canonical order of the renumbered rows
    ci_int = int_of_ext[ci_ext]
    for r in range(p.m):
        ci_int[rp[r]:rp[r + 1]].sort()
    c_rp, c_ci, _, _ = orc.canonical_layout(p.row_ptr, ci_int)
    assert np.array_equal(d_rp, c_rp) and np.array_equal(int_of_ext[d_ci], c_ci)
    assert np.array_equal(prob.l(), l_ext)
    mu0_g, uh_g = prob.start_values()
    assert np.array_equal(mu0_g, orc.start_values_exact(pd)) and np.array_equal(uh_g, orc.start_values(pd)[1])
    em_g, it_g, ll_g = prob.em(mu0_g, max_iter=6, epsilon=-1e308)
    em_o, it_o, ll_o = orc.em(pd, mu0_g, max_iter=6, epsilon=-1e308)
    assert np.array_equal(em_g, em_o) and ll_g == ll_o
    s = gpu.Sampler(prob, em_g, seed=77, n_chains=2, gibbs_iter=24, trace_len=12)
    s.run(24)
    for c in range(2):
        ref = orc.gibbs_keyed(pd, em_g, seed=77, chain=c, n_iter=24, trace_len=12)
        assert np.array_equal(s.trace(c), ref["trace"]) and np.array_equal(s.counts(c), ref["cnt"])
        assert np.array_equal(s.mu(c), ref["mu"]) and np.array_equal(s.trace_rows(c).T, ref["trace"])
        sl, sl2, ns = s.moments(c)
        assert np.array_equal(sl, ref["sum_log"]) and np.array_equal(sl2, ref["sum_log2"])
    # an upload that already is in stored order (and says so) runs the same chain
    pre = gpu.Problem.from_csr(d_rp, d_ci, l_ext, tx_order=tx_order, keep_rows=True)
    assert pre.info.sample_kernel == 2
    s2 = gpu.Sampler(pre, em_g, seed=77, gibbs_iter=24, trace_len=12)
    s2.run(24)
    assert np.array_equal(s2.trace(0), s.trace(0))


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
    prob, p = _dev(gpu, orc, p)
    s = gpu.Sampler(prob, mu0, seed=42, gibbs_iter=4, trace_len=4)
    s.sample()
    cnt = s.counts(0)
    ref = orc.sample_counts(p, mu0, 42, 0, 0)
    assert np.array_equal(cnt, ref)
    assert int(cnt.sum()) == p.total_k()
    s.update()
    mu1 = s.mu(0)
    assert np.array_equal(mu1, orc.gamma_update(ref, p.l, 0.1, 0.1, 42, 0, 0))


# the two sample kernels: sliced-ELL 8-bit stream (the default), 32-bit CSR tiles (problems whose rows mostly span more than a window)
@pytest.mark.parametrize("kernel", [2, 0])
@pytest.mark.parametrize("R,T,avg", [(10000, 1000, 4), (30000, 500, 12), (2000, 4000, 2), (70000, 3000, 30)])
def test_full_chain_bit_exact(gpu, orc, R, T, avg, kernel):
    p, mu0, _ = _mk(orc, R, T, avg)                      # the generator's rows in canonical order
    with gpu.options(sample_kernel=kernel):
        prob, pd = _dev(gpu, orc, p)
    assert np.array_equal(pd.row_ptr, p.row_ptr) and np.array_equal(pd.col_idx, p.col_idx)   # canonical order is idempotent
    assert prob.info.sample_kernel == kernel
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
    """k > 1 rows: k <= min(64 (MMG_K_SMALL), 16 (hits - 1)) -> repeated categorical draws; above -> conditional-binomial chain (spec version
    8: sampled from the list of those rows by k_sample_bigk)."""
    p, mu0, _ = _mk(orc, 5000, 400, 5)
    rng = np.random.default_rng(7)
    k = rng.choice([1, 2, 3, 8, 9, 50, 1000, 20000], size=p.m).astype(np.uint32)
    pk = orc.Problem(p.row_ptr, p.col_idx, p.l * 50, k=k)
    mu0, uh = orc.start_values(pk)
    prob, pk = _dev(gpu, orc, pk)
    s = gpu.Sampler(prob, mu0, seed=9, gibbs_iter=32, trace_len=32)
    s.run(32)
    ref = orc.gibbs_keyed(pk, mu0, seed=9, n_iter=32, trace_len=32)
    cnt = s.counts(0)
    assert int(cnt.sum()) == pk.total_k()
    assert np.array_equal(cnt, ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])


@pytest.mark.parametrize("per_wave,side", [(0, 1), (1, 1), (7, 0), (64, 1), (200, 0)])
def test_rows_on_the_binomial_chain_from_their_list(gpu, orc, per_wave, side):
    """k_sample_bigk (bigk_kernels.h): the rows on the conditional-binomial chain (k > min(64, 16 (hits - 1)), src/mmseq.cpp:880 on
    collapsed hit sets) are sampled from a list, a lane per row, every lane a state machine that runs ahead of its neighbours -- and the
    inversion settles x = 0 from a bound on r0 where it can.  Bit-exact against the oracle's sequential loop for every piece size
    (1 row per wave ... the whole list in one wave), on the sampler's stream and beside it, for two chains of one sampler: rows of 2 ...
    300 hits (CSR-walked tiles), far rows, k from 17 to 10^6, weights from 1e-300 to 1e6 (n p from 1e-290 up: every branch of the binomial),
    rows whose weights are all zero or infinite (uniform split), and the rows that are NOT on the list beside them."""
    rng = np.random.default_rng(11)
    T = 30000
    rows, ks = [], []
    for lead in range(0, T - 400, 53):
        for L in (2, 2, 3, 4, 5, 9, 17, 33, 1, 6):
            rows.append(sorted(rng.choice(np.arange(lead, lead + 200), size=L, replace=False).tolist()))
            ks.append(int(rng.choice([1, 3, 16, 17, 33, 49, 64, 65, 66, 100, 304, 305, 1000, 20000, 1000000])))
    rows[40] = list(range(9000, 9300)); ks[40] = 5000                     # 300 hits: a CSR-walked tile
    rows[41] = [3, 29000]; ks[41] = 777                                    # far row
    rows[42] = [10, 11, 12, 25000, 25001]; ks[42] = 100000                 # far row, long tail
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    l = np.linspace(0.5, 2.0, T)
    k = np.asarray(ks, np.uint32)
    p = orc.Problem(rp, ci, l, k=k)
    mu0 = np.exp(rng.normal(0.0, 4.0, size=T))                             # six orders of magnitude inside a row
    mu0[::5] = 1e-300
    mu0[1::11] = 1e-12
    mu0[2000:2200] = 0.0
    mu0[3000:3100] = np.inf
    opts = dict(sample_kernel=2, bigk_side_stream=side)
    if per_wave:
        opts["bigk_per_wave"] = per_wave
    with gpu.options(**opts):
        prob, ps = _dev(gpu, orc, p)
        Ls = np.diff(ps.row_ptr.astype(np.int64))
        on_list = (Ls >= 2) & (ps.k > np.minimum(64, 16 * (Ls - 1)))
        assert 1000 < int(on_list.sum()) < ps.m and prob.info.sample_kernel == 2
        s = gpu.Sampler(prob, mu0, seed=23, n_chains=2, gibbs_iter=6, trace_len=6)
        s.sample()
        for c in range(2):
            assert np.array_equal(s.counts(c), orc.sample_counts(ps, mu0, seed=23, chain=c, it=0))
        s.update()
        s.run(5)
        for c in range(2):
            ref = orc.gibbs_keyed(ps, mu0, seed=23, chain=c, n_iter=6, trace_len=6)
            assert np.array_equal(s.counts(c), ref["cnt"]) and np.array_equal(s.trace(c), ref["trace"])
            assert int(s.counts(c).astype(np.int64).sum()) == ps.total_k()
        s.close(); prob.close()


def test_btrs_pretest_never_contradicts_the_exact_test(gpu):
    """mmg_math.h: btrs_pretest -- k_sample_bigk decides BTRS's exact acceptance test from an fp32 estimate where the estimate is farther
    from zero than the bound on its own error, and runs the fp64 test (the oracle's, binomial()'s) for the rest.  The draw is the
    sequential code's only if a DECIDED case never differs from the fp64 test: 2.5 10^8 attempts per range of n (21 ... 2^32, p from 10 / n
    to 1/2, the uniforms of the sampler's row stream), none may; and the estimate has to be worth its instructions (> 90 % decided,
    its largest error well inside the bound)."""
    for lo, hi in ((21, 200), (200, 20000), (2e4, 1e7), (1e7, 4.29e9), (21, 4.29e9)):
        reached, decided, wrong, accepted, share = gpu.selftest_btrs_pretest(seed=5 + int(lo), n_cases=250_000_000, n_lo=lo, n_hi=hi)
        assert reached > 30_000_000 and 0.3 < accepted / reached < 0.8
        assert wrong == 0
        assert decided / reached > 0.9 and share < 250_000                  # (error <= a quarter of the bound; measured: 0.07)


def test_binv_pretest_never_contradicts_the_fp64_search(gpu):
    """mmg_math.h: binv_pretest -- k_sample_bigk runs the inversion's search (n p < 10) on fp32 terms and sums and takes its x where no
    boundary of the cumulative sum comes closer to the uniform than the bound on the fp32 sum's error; the fp64 search of the
    sequential code runs for the rest.  10^8 searches per range of n (n p from 1e-6 to 10): a decided case never differs -- not even
    with the bound at a sixteenth of the sampler's --, and at least 99.9 % are decided.  (n stops at 10^7 here: a search whose fp64 terms
    sum short of the uniform walks all n terms before the sequential code draws again, 4 in 10^9 at n ~ 10^9.)"""
    for lo, hi in ((1, 20), (20, 2000), (2000, 1e7)):
        for slack in (1.0, 1.0 / 16):
            cases, decided, wrong, fell, total = gpu.selftest_binv_pretest(seed=9 + int(lo), n_cases=100_000_000, n_lo=lo, n_hi=hi, slack=slack)
            assert cases > 80_000_000 and 0.1 < total / cases < 1.0
            assert wrong == 0
            assert decided / cases > 0.999


def test_chain_rows_whose_exact_test_falls_back_to_fp64(gpu, orc):
    """Rows of 2 and 3 hits with k = 3 10^8 ... 5 10^8 and weights of one size: every step is BTRS at n ~ 10^8, where one exact test in twenty is
    too close for the fp32 estimate and takes the fp64 path inside k_sample_bigk (bigk_kernels.h: SLOW) -- bit-exact against the oracle."""
    rng = np.random.default_rng(3)
    n_rows, T = 600, 2400
    rows = [[4 * i, 4 * i + 1] if i % 2 else [4 * i, 4 * i + 1, 4 * i + 2] for i in range(n_rows)]
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    k = rng.integers(300_000_000, 500_000_000, size=n_rows).astype(np.uint32)
    p = orc.Problem(rp, ci, np.linspace(0.5, 2.0, T), k=k)
    mu0 = rng.uniform(0.5, 2.0, size=T)
    with gpu.options(sample_kernel=2):
        prob, ps = _dev(gpu, orc, p)
        s = gpu.Sampler(prob, mu0, seed=77, n_chains=2, gibbs_iter=8, trace_len=8)
        s.run(8)
        for c in range(2):
            ref = orc.gibbs_keyed(ps, mu0, seed=77, chain=c, n_iter=8, trace_len=8)
            assert np.array_equal(s.counts(c), ref["cnt"]) and np.array_equal(s.trace(c), ref["trace"])
        s.close(); prob.close()


@pytest.mark.parametrize("keep_rows", [False, True])
def test_edge_rows(gpu, orc, keep_rows):
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
    prob, p = _dev(gpu, orc, p, keep_rows=keep_rows)
    assert prob.info.max_row_len == 5000
    if keep_rows:
        assert np.array_equal(p.row_ptr, rp) and np.array_equal(p.col_idx, ci)
    s = gpu.Sampler(prob, mu0, seed=5, gibbs_iter=16, trace_len=16)
    s.run(16)
    ref = orc.gibbs_keyed(p, mu0, seed=5, n_iter=16, trace_len=16)
    assert np.array_equal(s.counts(0), ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])
    assert int(s.counts(0).sum()) == int(k[[1, 2, 3, 4, 6, 7, 8]].sum())


@pytest.mark.parametrize("keep_rows", [True, False])
@pytest.mark.parametrize("with_k", [False, True])
def test_sliced_ell_kernel_edge_rows(gpu, orc, with_k, keep_rows):
    """The sliced-ELL kernel (forced even though few tiles qualify) on the row shapes its format singles out: rows of
    exactly 32 / 33 hits (the register cache holds 8 groups), 255 and 256+ hits (the length byte; longer rows make their
    tile a slow tile), single-hit and empty rows, a window slide inside a workgroup's range, a ragged last tile,
    degenerate weights (all mu of a row zero / infinite) -- bit-exact against the oracle, with and without multiplicities."""
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
    with gpu.options(sample_kernel=2, sell_waves_per_cu=1):  # few workgroups: long tile ranges, several window slides each
        prob, p = _dev(gpu, orc, p, keep_rows=keep_rows)
    # canonical order: nearly every tile runs from the register stream, the 300-hit row and the far row sit in slow tiles; kept
    # as given, the bands change from row to row and (almost) every 64-row tile is walked from the CSR by the same kernel
    assert prob.info.sample_kernel == 2 and prob.info.fast_tiles < prob.info.n_tiles
    assert keep_rows or prob.info.fast_tiles > 0.9 * prob.info.n_tiles
    s = gpu.Sampler(prob, mu0, seed=17, gibbs_iter=12, trace_len=12)
    s.sample()
    ref1 = orc.sample_counts(p, mu0, seed=17, chain=0, it=0)
    assert np.array_equal(s.counts(0), ref1)
    s.update()
    s.run(11)
    ref = orc.gibbs_keyed(p, mu0, seed=17, n_iter=12, trace_len=12)
    assert np.array_equal(s.counts(0), ref["cnt"])
    assert np.array_equal(s.trace(0), ref["trace"])


@pytest.mark.parametrize("with_k", [False, True])
def test_far_tiles_bit_exact(gpu, orc, with_k):
    """Rows with hits outside their window (reads that also hit a paralogue elsewhere in the transcriptome) sort by their HOME band
    and are walked from far tiles: window bytes, escapes, a far list per lane.  Far hits below and above the window, rows that are
    mostly far, multiplicities up to the binomial chain, EM and fused chains (which still walk these rows from the CSR)."""
    rng = np.random.default_rng(21)
    p, _ = orc.synth_problem(R=60000, T=20000, avg_hits=9, seed=77, sort=False, far_fraction=0.3)
    rp, ci = p.row_ptr.astype(np.int64), p.col_idx.copy()
    rows = [ci[rp[i]:rp[i + 1]].tolist() for i in range(0, 3000)]
    # hand-made shapes: a row scattered over the whole range, one far hit in front / behind, exactly 254 / 255 apart, a far-only pair
    rows += [sorted(rng.choice(20000, size=40, replace=False).tolist()) for _ in range(100)]
    rows += [[3, 9000, 9001, 9002, 9050], [9000, 9001, 9050, 19999], [6400, 6400 + 253], [6400, 6400 + 254], [6400, 6400 + 255],
             [0, 19999], [64, 5000, 5001], []]
    ci = np.concatenate([p.col_idx[rp[3000]:], np.concatenate([np.asarray(r, np.uint32) for r in rows if r])]).astype(np.uint32)
    lens = np.concatenate([np.diff(rp)[3000:], [len(r) for r in rows]])
    rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    k = rng.choice([1, 1, 1, 2, 8, 9, 300], size=lens.size).astype(np.uint32) if with_k else None
    q = orc.Problem(rp2, ci, p.l, k=k)
    mu0, _ = orc.start_values(q)
    mu0[::7] = 1e-200
    prob, qs = _dev(gpu, orc, q)
    inf = prob.info
    assert inf.sample_kernel == 2 and inf.far_tiles > 0.2 * inf.n_tiles and inf.fast_tiles + inf.far_tiles >= inf.n_tiles - 2
    s = gpu.Sampler(prob, mu0, seed=5, gibbs_iter=8, trace_len=8)
    s.sample()
    assert np.array_equal(s.counts(0), orc.sample_counts(qs, mu0, seed=5, chain=0, it=0))
    s.update()
    s.run(7)
    ref = orc.gibbs_keyed(qs, mu0, seed=5, n_iter=8, trace_len=8)
    assert np.array_equal(s.counts(0), ref["cnt"]) and np.array_equal(s.trace(0), ref["trace"])
    g_mu, g_it, g_ll = prob.em(mu0, max_iter=5, epsilon=-1e308)
    o_mu, o_it, o_ll = orc.em(qs, mu0, max_iter=5, epsilon=-1e308)
    assert np.array_equal(g_mu, o_mu) and g_ll == o_ll
    if not with_k:
        m2 = gpu.Sampler(prob, mu0, seed=5, n_chains=2, gibbs_iter=4, trace_len=4)
        m2.run(2)                                     # left alone, chains of a problem with far tiles run one per launch
        with gpu.options(fuse_chains=2):
            m2.run(2)                                 # forced pairs: the fused kernel walks far tiles from the CSR
        for c in range(2):
            assert np.array_equal(m2.trace(c), orc.gibbs_keyed(qs, mu0, seed=5, chain=c, n_iter=4, trace_len=4)["trace"])


def test_unsorted_and_very_long_rows(gpu, orc):
    """Hits of a row delivered in arbitrary order (the library sorts them: insertion sort, heap sort above 64 hits), rows far longer
    than a tile can hold (CSR-walked), kept rows with a long far row (the rank-median's cap): stored order = the oracle's restatement,
    chain bit-exact."""
    rng = np.random.default_rng(33)
    T = 30000
    rows = [rng.choice(np.arange(lead, lead + 100), size=int(rng.integers(1, 30)), replace=False).tolist() for lead in rng.integers(0, T - 200, size=4000)]
    rows.append(rng.permutation(T)[:6000].tolist())                        # 6000 hits all over the range, shuffled
    rows.append(rng.permutation(np.arange(500, 800)).tolist())            # 300 hits in a window and a half, shuffled
    rows.append([17])
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    l = np.linspace(0.3, 3.0, T)
    p = orc.Problem(rp, ci, l)
    mu0 = rng.gamma(0.5, 1.0, size=T)
    for keep in (False, True):
        prob, ps = _dev(gpu, orc, p, keep_rows=keep)
        if not keep:
            c_rp, c_ci, _, _ = orc.canonical_layout(rp, ci)
            assert np.array_equal(ps.row_ptr, c_rp) and np.array_equal(ps.col_idx, c_ci)
        else:
            assert np.array_equal(ps.row_ptr, rp) and np.array_equal(ps.col_idx, ci)
        s = gpu.Sampler(prob, mu0, seed=3, gibbs_iter=6, trace_len=6)
        s.run(6)
        ref = orc.gibbs_keyed(ps, mu0, seed=3, n_iter=6, trace_len=6)
        assert np.array_equal(s.counts(0), ref["cnt"]) and np.array_equal(s.trace(0), ref["trace"])
        s.close(); prob.close()


def test_shards_of_a_stored_problem_with_far_rows_and_tx_order(gpu, orc):
    """What `mmseq -gpus N` does: the stored rows of device 0's problem (download), cut at even boundaries, uploaded as kept rows with
    the same transcript numbering -- the shards' counts add up to the unsharded chain's, far rows (stored window-hits-first, NOT
    ascending) and multiplicities included."""
    rng = np.random.default_rng(8)
    p, _ = orc.synth_problem(R=50000, T=12000, avg_hits=7, seed=31, sort=False, far_fraction=0.25)
    scat = rng.permutation(p.n).astype(np.uint32)
    ci_ext = scat[p.col_idx]
    l_ext = np.empty(p.n); l_ext[scat] = p.l
    tx_order = np.empty(p.n, np.uint64); tx_order[scat] = np.arange(p.n, dtype=np.uint64)
    k = rng.choice([1, 1, 1, 2, 5, 70], size=p.m).astype(np.uint32)
    prob = gpu.Problem.from_csr(p.row_ptr, ci_ext, l_ext, k=k, tx_order=tx_order)
    assert prob.info.far_tiles > 0
    rp, ci, kk = prob.download(with_k=True)
    mu0 = rng.gamma(0.7, 1.0, size=p.n)
    s = gpu.Sampler(prob, mu0, seed=19, gibbs_iter=1, trace_len=1)
    s.sample()
    whole = s.counts(0)
    assert np.array_equal(whole, orc.sample_counts(orc.Problem(rp, ci, l_ext, k=kk), mu0, seed=19, chain=0, it=0))
    m_st = rp.size - 1                                         # stored rows: a caller row that draws k categoricals is k of them
    assert m_st > p.m and {1, 70} <= set(np.unique(kk)) <= {1, 2, 5, 70}   # (rows of one hit keep their 2 or 5)
    cut = (m_st // 2) & ~1
    nz = int(rp[cut])
    parts = []
    for lo, hi in ((0, cut), (cut, m_st)):
        a, b = int(rp[lo]), int(rp[hi])
        sh = gpu.Problem.from_csr(rp[lo:hi + 1] - rp[lo], ci[a:b], l_ext, k=kk[lo:hi], row_id_base=lo, keep_rows=True, tx_order=tx_order)
        d_rp, d_ci = sh.download()
        assert np.array_equal(d_ci, ci[a:b])                       # hit order kept as stored
        ss = gpu.Sampler(sh, mu0, seed=19, gibbs_iter=1, trace_len=1)
        ss.sample()
        parts.append(ss.counts(0))
    assert np.array_equal(parts[0] + parts[1], whole) and nz > 0


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_sharded_chain_on_one_device_equals_the_unsharded_chain(gpu, orc, parts):
    """mmg_group_run_sharded's arithmetic on one device (mmg_selftest_gibbs_shards: the int32 count exchange done by a kernel instead of
    RCCL): the stored rows of a problem with far rows, multiplicities of every class and a transcript order, cut where the LIBRARY cuts
    (mmg_problem_shard_bounds, by modelled cost) with mmg_problem_shard -- trace, counts and mu of every shard's sampler equal the
    unsharded chain's and the oracle's bit for bit (src/mmseq.cpp:864, :893-899: the reference's split of the rows and its reduction)."""
    rng = np.random.default_rng(70 + parts)
    p, _ = orc.synth_problem(R=60000, T=9000, avg_hits=7, seed=13, sort=False, far_fraction=0.25)
    k = rng.choice([1, 1, 1, 1, 2, 5, 70, 300], size=p.m).astype(np.uint32)
    scat = rng.permutation(p.n).astype(np.uint32)              # the caller numbers the transcripts at random; tx_order restores locality
    l_ext = np.empty(p.n); l_ext[scat] = p.l
    tx_order = np.empty(p.n, np.uint64); tx_order[scat] = np.arange(p.n, dtype=np.uint64)
    prob = gpu.Problem.from_csr(p.row_ptr, scat[p.col_idx], l_ext, k=k, tx_order=tx_order)
    assert prob.info.far_tiles > 0 and prob.info.sample_kernel == 2
    rp, ci, kk = prob.download(with_k=True)
    ps = orc.Problem(rp, ci, l_ext, k=kk)
    mu0 = rng.gamma(0.7, 1.0, size=p.n)
    iters = 6
    ref = orc.gibbs_keyed(ps, mu0, seed=21, n_iter=iters, trace_len=iters)
    whole = gpu.Sampler(prob, mu0, seed=21, gibbs_iter=iters, trace_len=iters)
    whole.run(iters)
    assert np.array_equal(whole.trace(0), ref["trace"]) and np.array_equal(whole.counts(0), ref["cnt"])
    b = prob.shard_bounds(parts) if parts != 3 else prob.shard_bounds_timed(mu0, parts)   # modelled cost / measured cost: same bits wherever the cut
    assert b[0] == 0 and b[-1] == prob.info.m and np.all(np.diff(b.astype(np.int64)) > 0) and np.all(b[:-1] % 2 == 0)
    shards = [prob.shard(int(b[i]), int(b[i + 1])) for i in range(parts)]
    assert sum(sh.info.total_k for sh in shards) == prob.info.total_k
    assert all(sh.info.sample_kernel == 2 for sh in shards)      # a shard of a canonical problem stays on the stream kernel
    smps = [gpu.Sampler(sh, mu0, seed=21, gibbs_iter=iters, trace_len=iters, keep_trace=(i in (0, parts - 1))) for i, sh in enumerate(shards)]
    gpu.gibbs_shards_selftest(smps, 2)
    gpu.gibbs_shards_selftest(smps, iters - 2)                    # the call may be repeated: the chain continues
    for i in (0, parts - 1):
        assert np.array_equal(smps[i].trace(0), ref["trace"])
    for sm in smps:
        assert sm.iteration == iters
        assert np.array_equal(sm.counts(0), ref["cnt"]) and np.array_equal(sm.mu(0), ref["mu"])
    with pytest.raises(gpu._lib.MMGError):                        # samplers that disagree on the iteration are refused
        smps[0].run(1)
        gpu.gibbs_shards_selftest(smps, 1)
    for sm in smps:
        sm.close()
    for sh in shards:
        sh.close()
    whole.close(); prob.close()


def test_derived_order_on_graphs_without_a_band(gpu, orc):
    """The order derived from the hit graph (spec version 6) is kept only when the model prices the rebuilt problem a fifth lower, and is
    not even attempted on a graph without locality: hits drawn uniformly over the transcripts stay on the CSR-tile kernel in the
    caller's numbering; disconnected gene families in random numbering (the shape of a real transcriptome: small components, plus
    transcripts no read hits and one hub that shares rows with everything) are laid out component by component -- and every chain
    is the oracle's replay of the downloaded rows."""
    rng = np.random.default_rng(2)
    u, _ = orc.synth_problem(R=120000, T=30000, avg_hits=8, seed=3, uniform=True, sort=False)
    pu = gpu.Problem.from_csr(u.row_ptr, u.col_idx, u.l)
    assert pu.info.sample_kernel == 0 and pu.info.tx_renumbered == 0
    pu.close()
    # 3000 gene families of 2..9 transcripts, numbered at random; reads hit 1..6 transcripts of one family; a hub in 1 % of the reads
    T = 16000
    ids = rng.permutation(T - 500)                      # the last 500 ids: transcripts without any read
    fams, i = [], 0
    while i < ids.size:
        sz = int(rng.integers(2, 10))
        fams.append(ids[i:i + sz]); i += sz
    hub = int(ids[0])
    rows = []
    for _ in range(150000):
        f = fams[int(rng.integers(len(fams)))]
        r = rng.choice(f, size=int(rng.integers(1, min(6, f.size) + 1)), replace=False).tolist()
        if rng.random() < 0.01 and hub not in r:
            r.append(hub)
        rows.append(sorted(r))
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    l = rng.uniform(0.2, 3.0, T)
    with gpu.options(derive_order=0):
        p0 = gpu.Problem.from_csr(rp, ci, l)
        k0 = p0.info.sample_kernel
        p0.close()
    p1 = gpu.Problem.from_csr(rp, ci, l)
    inf = p1.info
    assert k0 == 0 and inf.sample_kernel == 2 and inf.tx_renumbered == 2 and inf.fast_tiles + inf.far_tiles >= 0.95 * inf.n_tiles
    perm = p1.tx_perm()
    for f in fams[:200]:                                # a family's transcripts end up within an LDS window of each other
        if hub in f:
            continue
        assert int(perm[f].max()) - int(perm[f].min()) < 240, (f, perm[f])
    d_rp, d_ci = p1.download()
    mu0, _ = p1.start_values()
    s = gpu.Sampler(p1, mu0, seed=9, gibbs_iter=6, trace_len=6)
    s.run(6)
    ref = orc.gibbs_keyed(orc.Problem(d_rp, d_ci, l), mu0, seed=9, n_iter=6, trace_len=6)
    assert np.array_equal(s.trace(0), ref["trace"]) and np.array_equal(s.counts(0), ref["cnt"])
    s.close(); p1.close()


def test_a_heavily_collapsed_file_is_not_uncollapsed(gpu, orc):
    """Rows that draw k >= 2 categoricals (k <= min(64, 16 (hits - 1))) are stored k times (step 0 of the canonical layout) -- unless that would store more than 8 rows per uploaded
    row: a file whose hit sets are shared by dozens of reads each keeps its multiplicities (memory and work stay with the hit sets, not
    the reads) and the multiplicity kernel draws the same k categoricals per row.  Stored order = the oracle's restatement, chain = the
    oracle's, on both sides of the limit."""
    rng = np.random.default_rng(21)
    p, _ = orc.synth_problem(R=30000, T=4000, avg_hits=6, seed=17, sort=False, far_fraction=0.05)
    mu0 = rng.gamma(0.6, 1.0, size=p.n)
    for lo, hi, expanded in ((30, 64, False), (1, 12, True)):
        k = rng.integers(lo, hi + 1, size=p.m).astype(np.uint32)
        prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, k=k)
        rp, ci, kk = prob.download(with_k=True)
        c_rp, c_ci, c_k, _ = orc.canonical_layout(p.row_ptr, p.col_idx, k)
        assert np.array_equal(rp, c_rp) and np.array_equal(ci, c_ci)
        assert prob.info.m == c_rp.size - 1 and (prob.info.m > p.m) == expanded and (prob.info.m == p.m) == (not expanded)
        assert np.array_equal(kk, c_k if c_k is not None else np.ones(rp.size - 1, np.uint32))
        assert prob.info.total_k == int(k.sum())
        s = gpu.Sampler(prob, mu0, seed=4, gibbs_iter=5, trace_len=5)
        s.run(5)
        ref = orc.gibbs_keyed(orc.Problem(rp, ci, p.l, k=(None if c_k is None else kk)), mu0, seed=4, n_iter=5, trace_len=5)
        assert np.array_equal(s.trace(0), ref["trace"]) and np.array_equal(s.counts(0), ref["cnt"])
        assert int(s.counts(0).astype(np.int64).sum()) == int(k.sum())
        s.close(); prob.close()


@pytest.mark.parametrize("kernel", [2, 0])
def test_replicated_count_vectors_give_the_same_bits(gpu, orc, kernel):
    """K1's workgroups flush their LDS counts into one of CNT_REPLICAS global vectors (workgroup index mod 8: neighbouring ranges meet at
    different addresses) when many ranges share a band, and K2 sums them -- integer sums: the chain, the counts between sample and
    update (folded into the public vector by mmg_sampler_sample) and the fused chains are the same bits with 1 and with 8 vectors."""
    rng = np.random.default_rng(12)
    p, _ = orc.synth_problem(R=50000, T=7000, avg_hits=6, seed=77, sort=False, far_fraction=0.1)
    k = rng.choice([1, 1, 1, 3, 90], size=p.m).astype(np.uint32)
    mu0 = rng.gamma(0.6, 1.0, size=p.n)
    out = {}
    for reps in (1, 8):
        with gpu.options(cnt_replicas=reps, sample_kernel=kernel):
            prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, k=k)
            assert prob.info.sample_kernel == kernel
            s = gpu.Sampler(prob, mu0, seed=5, n_chains=3, gibbs_iter=4, trace_len=4)
            s.run(2)
            s.sample()
            mid = [s.counts(c) for c in range(3)]
            s.update()
            s.run(1)
            out[reps] = (mid, [s.trace(c) for c in range(3)], [s.counts(c) for c in range(3)])
            if reps == 8:
                rp, ci, kk = prob.download(with_k=True)
                ref = orc.gibbs_keyed(orc.Problem(rp, ci, p.l, k=kk), mu0, seed=5, chain=1, n_iter=4, trace_len=4)
                assert np.array_equal(out[8][1][1], ref["trace"]) and np.array_equal(out[8][2][1], ref["cnt"])
            s.close(); prob.close()
    for a, b in zip(out[1], out[8]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert int(out[8][0][0].astype(np.int64).sum()) == int(k.astype(np.int64).sum())


def test_shard_bounds_balance_modelled_cost_not_hits(gpu, orc):
    """The canonical order stores every far row behind every near row, and a far tile costs about three register-path tiles: the
    library's cut (mmg_problem_shard_bounds) gives the shards that hold the far rows FEWER hits, where equal hit counts
    (mmg_shard_bounds; the reference's static split of the rows, src/mmseq.cpp:864) would make them the slowest devices."""
    p, _ = orc.synth_problem(R=200000, T=12000, avg_hits=8, seed=5, sort=False, far_fraction=0.2)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    inf = prob.info
    assert inf.sample_kernel == 2 and inf.far_tiles > 0
    rp, _ = prob.download()
    parts = 8
    b = prob.shard_bounds(parts)
    hits = np.diff(rp[b.astype(np.int64)].astype(np.int64))
    by_hits = gpu.shard_bounds(rp, parts)
    assert np.all(np.diff(by_hits.astype(np.int64)) > 0)
    # the last shard is all far rows: markedly fewer hits than the first (all near rows)
    assert hits[-1] < 0.6 * hits[0], hits
    # near-only shards among themselves: equal cost = (nearly) equal hits
    near_only = hits[:2]
    assert near_only.max() <= 1.08 * near_only.min(), hits
    # the measured cut sees the same thing
    mu0, _ = prob.start_values()
    bt = prob.shard_bounds_timed(mu0, parts)
    ht = np.diff(rp[bt.astype(np.int64)].astype(np.int64))
    assert bt[0] == 0 and bt[-1] == inf.m and np.all(np.diff(bt.astype(np.int64)) > 0) and np.all(bt[:-1] % 2 == 0)
    assert ht[-1] < 0.75 * ht[0], ht
    # a problem without far rows: the cost cut and the hit cut agree to within a tile's rounding
    q, _ = orc.synth_problem(R=200000, T=12000, avg_hits=8, seed=5, sort=False)
    pq = gpu.Problem.from_csr(q.row_ptr, q.col_idx, q.l)
    rq, _ = pq.download()
    hq = np.diff(rq[pq.shard_bounds(parts).astype(np.int64)].astype(np.int64))
    assert hq.max() <= 1.10 * hq.min(), hq                       # (bands end in partly filled tiles, and a tile costs the same full or not)
    pq.close(); prob.close()


@pytest.mark.parametrize("parts,wild", [(2, False), (3, True), (8, False)])
def test_em_over_read_shards_equals_the_unsharded_em(gpu, orc, parts, wild):
    """mmg_group_em_create's arithmetic on one device (mmg_selftest_em_shards: the exchange done by kernels instead of RCCL): the
    stored rows of a problem with multiplicities and far rows cut into `parts` shards (mmg_shard_bounds), every phase of a sweep on
    every shard, xe / accumulators / column counts exchanged -- mu, log-likelihood and the repeat decisions equal the unsharded
    EM's bit for bit, also from start values that force passes to be repeated on measured exponents."""
    rng = np.random.default_rng(3)
    p, _ = orc.synth_problem(R=40000, T=3000, avg_hits=6, seed=41, sort=False, far_fraction=0.1)
    k = rng.choice([1, 1, 1, 2, 7, 300], size=p.m).astype(np.uint32)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l * 30, k=k)
    rp, ci, kk = prob.download(with_k=True)
    mu0, _ = prob.start_values()
    if wild:
        mu0[::7] *= 1e150
        mu0[3] = 0.0
    sweeps = 6
    em = prob.em_stepper(mu0)
    for _ in range(sweeps):
        em.step()
    want_mu, want_ll, want_rep = em.mu(), em.loglik, em.stats()["repeated_passes"]
    em.close()
    if wild:
        assert want_rep >= 1
    b = gpu.shard_bounds(rp, parts)
    shards = []
    for i in range(parts):
        lo, hi = int(b[i]), int(b[i + 1])
        a, e = int(rp[lo]), int(rp[hi])
        if parts == 8:   # cut on the device (mmg_problem_shard), as `mmseq -gpus N` does
            sh = prob.shard(lo, hi)
            s_rp, s_ci, s_k = sh.download(with_k=True)
            assert np.array_equal(s_rp, rp[lo:hi + 1] - rp[lo]) and np.array_equal(s_ci, ci[a:e]) and np.array_equal(s_k, kk[lo:hi])
            assert sh.info.row_id_base == lo and sh.info.layout == 1 and sh.info.total_k == int(kk[lo:hi].astype(np.int64).sum())
            shards.append(sh)
        else:
            shards.append(gpu.Problem.from_csr(rp[lo:hi + 1] - rp[lo], ci[a:e], prob.l(), k=kk[lo:hi], row_id_base=lo, keep_rows=True))
    pb = prob.shard_bounds(parts)                # the library's own cut balances modelled cost, not hits (below): any cut gives the same bits
    assert pb[0] == 0 and pb[-1] == prob.info.m and np.all(np.diff(pb.astype(np.int64)) >= 0) and np.all(pb[:-1] % 2 == 0)
    mu, ll, rep = gpu.em_shards_selftest(shards, mu0, sweeps)
    # the Gibbs counts of the shards add up to the unsharded sweep's
    s = gpu.Sampler(prob, mu0 + 1e-3, seed=3, gibbs_iter=1, trace_len=1)
    s.sample()
    tot = np.zeros(prob.info.n, np.int64)
    for sh in shards:
        ss = gpu.Sampler(sh, mu0 + 1e-3, seed=3, gibbs_iter=1, trace_len=1)
        ss.sample()
        tot += ss.counts(0)
        ss.close()
    assert np.array_equal(tot, s.counts(0).astype(np.int64))
    s.close()
    assert np.array_equal(mu, want_mu, equal_nan=True) and ll == want_ll and rep == want_rep
    for sh in shards:
        sh.close()
    prob.close()


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
    pa = gpu.Problem.from_csr(p.row_ptr[:h + 1], p.col_idx[:nz], p.l, keep_rows=True)
    pb = gpu.Problem.from_csr(p.row_ptr[h:] - p.row_ptr[h], p.col_idx[nz:], p.l, row_id_base=h, keep_rows=True)
    ref = orc.gibbs_keyed(p, mu0, seed=11, chain=0, n_iter=1, trace_len=1)
    sa = gpu.Sampler(pa, mu0, seed=11, gibbs_iter=1, trace_len=1)
    sb = gpu.Sampler(pb, mu0, seed=11, gibbs_iter=1, trace_len=1)
    sa.sample(); sb.sample()
    assert np.array_equal(sa.counts(0) + sb.counts(0), ref["cnt"])


def test_device_generator_matches_oracle_generator(gpu, orc):
    for (R, T, avg, uni, row0, srt, far) in [(30000, 2000, 8, False, 0, True, 0.0), (5000, 300, 20, False, 12345, True, 0.0),
                                             (4000, 5000, 3, True, 7, True, 0.0), (1000, 50, 20, False, 0, False, 0.0),
                                             (20000, 1000, 8, False, 99, False, 0.0), (30000, 3000, 8, False, 5, True, 0.2),
                                             (9000, 2500, 5, False, 0, False, 0.02)]:
        prob = gpu.Problem.synthetic(R, T, avg, seed=1234, row0=row0, uniform=uni, mapped_reads=R, sort=srt, far_fraction=far)
        rp, ci = prob.download()
        p, aux = orc.synth_problem(R=R, T=T, avg_hits=avg, seed=1234, uniform=uni, row0=row0, sort=srt, far_fraction=far)
        assert np.array_equal(rp, p.row_ptr)
        assert np.array_equal(ci, p.col_idx)
        assert np.array_equal(prob.l(), p.l)
        lens = np.diff(rp.astype(np.int64))
        assert lens.min() >= 1 and lens.max() <= 100
        # rows ascend strictly (distinct transcripts, sorted); stored far rows keep their window hits in front instead
        if not srt:
            d = np.diff(ci.astype(np.int64))
            inner = np.ones(ci.size - 1, bool)
            inner[(rp[1:-1] - 1).astype(np.int64)] = False
            assert (d[inner] > 0).all()


def test_start_values_and_em_match_oracle(gpu, orc):
    p, mu0, uh = _mk(orc, 40000, 1500, 4)
    prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    g_mu0, g_uh = prob.start_values()
    assert np.array_equal(g_uh, uh)                       # integer: bit-exact
    assert np.array_equal(g_mu0, orc.start_values_exact(p))  # exact fixed-point sum of the shares: order-independent, bit-exact
    assert np.array_equal(g_mu0, prob.start_values()[0])
    np.testing.assert_allclose(g_mu0, mu0, rtol=1e-13, atol=0)  # the reference's sequential fp64 sum (src/mmseq.cpp:617-638)
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


@pytest.mark.parametrize("sort,em_kernel,grid", [(True, -1, -1), (True, -1, 7), (True, 0, -1), (False, -1, -1)])
def test_em_stepper_paths_match_oracle(gpu, orc, sort, em_kernel, grid):
    """Sliced-ELL stream kernel (canonical rows; also with few workgroups: long tile ranges, the window slides many times),
    row-per-thread kernel (forced / rows kept in generator order): same bits as the oracle, with multiplicities, dead
    transcripts and an empty row; the oracle in turn tracks the reference's summation order."""
    p, aux = orc.synth_problem(R=60000, T=2500, avg_hits=7, seed=11, sort=sort)
    rng = np.random.default_rng(5)
    k = rng.choice([1, 1, 1, 2, 7, 1000], size=p.m).astype(np.uint32)
    rp = np.concatenate([p.row_ptr[:1000], p.row_ptr[999:]])          # an empty row in the middle
    k = np.concatenate([k[:999], [3], k[999:]]).astype(np.uint32)
    pk = orc.Problem(rp, p.col_idx, p.l * 20, k=k)
    mu0, _ = orc.start_values(pk)
    mu0[5::97] = 0.0                                                    # dead from the start
    with gpu.options(em_kernel=em_kernel, em_grid=grid):
        prob, pk = _dev(gpu, orc, pk, keep_rows=not sort)
        em = prob.em_stepper(mu0)
    want = 0 if (not sort or em_kernel == 0) else 2
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
    keep = np.ones(k.size, bool)
    keep[999] = False
    p2 = orc.Problem(p.row_ptr, p.col_idx, pk.l, k=k[keep])
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


@pytest.mark.parametrize("name,seed", [("keyed_chain_tiny.json", 1234), ("keyed_chain_k_draws.json", 4321)])
def test_golden_tiny_chain_on_device(gpu, name, seed):
    """The committed golden fixtures: keyed_chain_tiny.json (k up to 1000, all three row paths) and keyed_chain_k_draws.json (spec
    version 5: rows on either side of the boundary between k categorical draws and the conditional-binomial chain)."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", name)))
    f = lambda hs: np.array([float.fromhex(h) for h in hs])
    prob = gpu.Problem.from_csr(np.array(g["row_ptr"], np.uint64), np.array(g["col_idx"], np.uint32), f(g["l"]),
                                k=np.array(g["k"], np.uint32), keep_rows=True)   # the fixture pins the chain of THESE rows in THIS order
    mu0, uh = prob.start_values()
    assert uh.tolist() == g["unique_hits"]
    np.testing.assert_allclose(mu0, f(g["mu0"]), rtol=1e-14)
    s = gpu.Sampler(prob, f(g["mu0"]), seed=seed, gibbs_iter=32, trace_len=16)
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
                                k=np.asarray(g["k"], np.uint32), keep_rows=True)  # (the canonical layout stores a row that draws 2 <= k <= 64 categoricals as k
                                                                                 # rows: k terms 1/d instead of one term k/d, other low bits)
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
        import tempfile
        stamp("before init_process_group")
        # (rendezvous through a file: a fixed port can be taken by another job on the host)
        dist.init_process_group("nccl", init_method="file://" + os.path.join(tempfile.mkdtemp(prefix="mmseq_nccl_"), "store"), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
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


def test_64bit_row_offsets_path(gpu, orc):
    """nnz >= 2^32 switches the device row_ptr to u64; MMG_OPT_FORCE_IDX64 exercises that instantiation on a small problem."""
    with gpu.options(force_idx64=1):
        _64bit_body(gpu, orc)


def _64bit_body(gpu, orc):
    p, mu0, uh = _mk(orc, 30000, 900, 7)
    rng = np.random.default_rng(2)
    k = rng.choice([1, 1, 1, 2, 9, 300], size=p.m).astype(np.uint32)
    pk = orc.Problem(p.row_ptr, p.col_idx, p.l * 10, k=k)
    mu0, _ = orc.start_values(pk)
    prob, pk = _dev(gpu, orc, pk)
    assert prob.info.index_bits == 64
    g_mu0, g_uh = prob.start_values()
    assert np.array_equal(g_mu0, orc.start_values_exact(pk))
    em_g, it_g, _ = prob.em(mu0)
    em_o, it_o, _ = orc.em(pk, mu0)
    assert it_g == it_o and np.array_equal(em_g, em_o)
    s = gpu.Sampler(prob, mu0, seed=4, gibbs_iter=32, trace_len=32)
    s.run(32)
    ref = orc.gibbs_keyed(pk, mu0, seed=4, n_iter=32, trace_len=32)
    assert np.array_equal(s.trace(0), ref["trace"]) and np.array_equal(s.counts(0), ref["cnt"])
    syn = gpu.Problem.synthetic(5000, 300, 6, seed=1234)
    assert syn.info.index_bits == 64
    q, _ = orc.synth_problem(R=5000, T=300, avg_hits=6, seed=1234)
    rp, ci = syn.download()
    assert np.array_equal(rp, q.row_ptr) and np.array_equal(ci, q.col_idx)


@pytest.mark.parametrize("fuse", [4, 2, 1])
@pytest.mark.parametrize("n_chains", [2, 4, 8, 11])
def test_chains_of_one_sampler_equal_independent_single_chains(gpu, orc, n_chains, fuse):
    """Chains advanced together by the fused walk (k_sample_sell_multi: groups of 4 / 2 / 1 chains per launch) are bit-identical
    to single-chain runs keyed with the same global chain index -- also across window slides, a far row and a 40-hit row."""
    p, mu0, _ = _mk(orc, 40000, 1500, 9, far_fraction=0.01)
    with gpu.options(fuse_chains=fuse, sell_waves_per_cu=2):
        prob, p = _dev(gpu, orc, p)
        s = gpu.Sampler(prob, mu0, seed=21, n_chains=n_chains, chain_base=3, gibbs_iter=16, trace_len=16)
        s.run(16)
    assert prob.info.sample_kernel == 2
    for c in sorted({0, 1, n_chains // 2, n_chains - 1}):
        ref = orc.gibbs_keyed(p, mu0, seed=21, chain=3 + c, n_iter=16, trace_len=16)
        assert np.array_equal(s.trace(c), ref["trace"]), c
        assert np.array_equal(s.counts(c), ref["cnt"]), c
