"""Differential fuzz (GPU box): random problems through every device path against the oracle, bit for bit.
usage: fuzz_parity.py [n_cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one_case(seed, gpu, orc, verbose=True):
    rng = np.random.default_rng(seed)
    T = int(rng.choice([40, 300, 2000, 20000]))
    R = int(rng.integers(200, 60000))
    avg = float(rng.choice([1.5, 3, 8, 20, 45]))
    sort = bool(rng.integers(0, 4) != 0)
    far = float(rng.choice([0.0, 0.0, 0.03, 0.5]))      # rows with a hit anywhere in the transcriptome: far tiles
    genes = int(rng.choice([0, 0, 0, 6, 24]))             # gene-block mode of the generator: a read's hits inside one gene ...
    uni = bool(rng.integers(0, 5) == 0) and not genes
    p, _ = orc.synth_problem(R=R, T=T, avg_hits=avg, seed=int(rng.integers(1, 1 << 30)), sort=sort,
                             uniform=uni, far_fraction=far, gene_size=genes,
                             far_family=int(rng.choice([0, 2, 3])) if genes else 0)   # ... far hits to a paralogue family of 2 / 3 genes, or anywhere
    if rng.integers(0, 4) == 0:                          # hits of a row in arbitrary order: the canonical layout sorts them
        rp64 = p.row_ptr.astype(np.int64)
        rid = np.repeat(np.arange(p.m), np.diff(rp64))
        p.col_idx[:] = p.col_idx[np.lexsort((rng.random(p.col_idx.size), rid))]
    dups = bool(rng.integers(0, 5) == 0)                 # a transcript twice in a row (the ABI takes rows as lists: its weight counts twice)
    if dups and p.col_idx.size > 4:
        rp64 = p.row_ptr.astype(np.int64)
        rows = rng.choice(np.nonzero(np.diff(rp64) >= 2)[0], size=max(1, p.m // 7))
        for r in rows:
            b, e = rp64[r], rp64[r + 1]
            p.col_idx[rng.integers(b, e)] = p.col_idx[rng.integers(b, e)]
    k = None
    if rng.integers(0, 2):
        k = rng.choice([1, 1, 1, 2, 3, 8, 9, 17, 40, 70, 150, 700, 5000, 300000], size=p.m).astype(np.uint32)   # 17 / 40: draws or binomial chain by row length (spec version 8)
    rp, ci = p.row_ptr.copy(), p.col_idx
    if rng.integers(0, 3) == 0 and p.m > 10:            # a few empty rows
        cut = np.sort(rng.choice(np.arange(1, p.m), size=3, replace=False))
        rp = np.insert(rp, cut, rp[cut])
        if k is not None:
            k = np.insert(k, cut, 1).astype(np.uint32)
    pk = orc.Problem(rp, ci, p.l * float(rng.choice([0.01, 1.0, 50.0])), k=k)
    opts = {}
    if rng.integers(0, 4) == 0: opts["force_idx64"] = 1
    kern = int(rng.choice([2, 2, -1, 0]))
    if kern >= 0: opts["sample_kernel"] = kern
    if rng.integers(0, 3) == 0: opts["sell_waves_per_cu"] = 1
    if rng.integers(0, 3) == 0: opts["em_grid"] = int(rng.integers(1, 9))
    if rng.integers(0, 3) == 0: opts["em_kernel"] = 0
    if rng.integers(0, 4) == 0: opts["fuse_chains"] = int(rng.choice([1, 4]))
    if rng.integers(0, 2) == 0: opts["cnt_replicas"] = int(rng.choice([1, 8]))     # one count vector per chain / eight (K2 sums them)
    if rng.integers(0, 3) == 0: opts["derive_order"] = int(rng.choice([0, 1]))     # never / always try an order from the hit graph
    if rng.integers(0, 3) == 0: opts["bigk_per_wave"] = int(rng.choice([1, 5, 64, 300]))   # list entries per wave of k_sample_bigk (the rows on the binomial chain)
    if rng.integers(0, 3) == 0: opts["bigk_side_stream"] = 0                          # ... on the sampler's stream instead of beside it
    keep_rows = bool(rng.integers(0, 3) == 0)
    tx_order = None
    if rng.integers(0, 3) == 0:                          # device renumbering: random gene sizes over a random scatter
        tx_order = (rng.permutation(T).astype(np.uint64) // np.uint64(int(rng.integers(1, 9)))) << np.uint64(32)
    elif genes:                                          # the CLI's keys for the generator's genes: gene << 32 | transcript (spec 7: the genes may be reordered)
        tx_order = ((np.arange(T, dtype=np.uint64) // np.uint64(genes)) << np.uint64(32)) | np.arange(T, dtype=np.uint64)
    with gpu.options(**opts):
        mu0, uh = orc.start_values(pk)
        if rng.integers(0, 3) == 0: mu0[rng.integers(0, T, size=max(1, T // 20))] = 0.0
        if rng.integers(0, 4) == 0: mu0 *= 10.0 ** rng.uniform(-120, 120, size=T)
        rid0 = int(rng.choice([0, 0, 12345, (1 << 32) - 100, (1 << 40) + 7]))    # shard offset: crosses 2^32 in the middle of a problem
        prob = gpu.Problem.from_csr(pk.row_ptr, pk.col_idx, pk.l, k=pk.k, row_id_base=rid0, keep_rows=keep_rows, tx_order=tx_order)
        d_rp, d_ci, d_k = prob.download(with_k=True)     # stored order: what the oracle replays
        pk = orc.Problem(d_rp, d_ci, pk.l, k=(d_k if k is not None else None))
        g_mu0, g_uh = prob.start_values()
        assert np.array_equal(g_uh, uh), "unique hits"
        assert np.array_equal(g_mu0, orc.start_values_exact(pk)), "start values"
        n_it = int(rng.integers(1, 6))
        chains = int(rng.choice([1, 1, 2, 3, 5]))        # pairs over the fast tiles, grid.y launches over far / multiplicity tiles, an odd chain
        alpha, beta = float(rng.choice([0.1, 1.0])), float(rng.choice([0.1, 2.0]))
        s = gpu.Sampler(prob, mu0, alpha=alpha, beta=beta, seed=seed, n_chains=chains, chain_base=2, gibbs_iter=n_it, trace_len=n_it)
        s.run(n_it)
        for c in range(chains):
            ref = orc.gibbs_keyed(pk, mu0, alpha=alpha, beta=beta, seed=seed, n_iter=n_it, trace_len=n_it, chain=2 + c, row_id_base=rid0)
            assert np.array_equal(s.counts(c), ref["cnt"]), "counts chain %d" % c
            assert np.array_equal(s.trace(c), ref["trace"]), "trace chain %d" % c
        s.close()
        live = np.isfinite(mu0) & (mu0 > 0)
        if live.any():
            sweeps = int(rng.integers(1, 5))
            mu_g, it_g, ll_g = prob.em(mu0, max_iter=sweeps, epsilon=-1e308)
            mu_o, it_o, ll_o = orc.em(pk, mu0, max_iter=sweeps, epsilon=-1e308)
            assert np.array_equal(mu_g, mu_o, equal_nan=True), "EM mu"
            assert ll_g == ll_o or (np.isnan(ll_g) and np.isnan(ll_o)), "EM loglik %r %r" % (ll_g, ll_o)
        info = prob.info
        prob.close()
        if verbose:
            print("seed %d ok: R=%d T=%d avg=%g far=%g sort=%s dups=%s k=%s keep=%s tx=%s kernel=%d tiles %d fast %d far %d opts=%s" % (
                seed, pk.m, T, avg, far, sort, dups, k is not None, keep_rows, tx_order is not None, info.sample_kernel, info.n_tiles, info.fast_tiles,
                info.far_tiles, opts), flush=True)


if __name__ == "__main__":
    from mmseq_amd import gibbs as gpu
    from oracle import binding as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for seed in range(s0, s0 + n):
        one_case(seed, gpu, orc)
    print("all %d cases bit-identical" % n)
