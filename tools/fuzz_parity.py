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
    p, _ = orc.synth_problem(R=R, T=T, avg_hits=avg, seed=int(rng.integers(1, 1 << 30)), sort=sort,
                             uniform=bool(rng.integers(0, 5) == 0))
    k = None
    if rng.integers(0, 2):
        k = rng.choice([1, 1, 1, 2, 3, 8, 9, 40, 5000], size=p.m).astype(np.uint32)
    rp, ci = p.row_ptr.copy(), p.col_idx
    if rng.integers(0, 3) == 0 and p.m > 10:            # a few empty rows
        cut = np.sort(rng.choice(np.arange(1, p.m), size=3, replace=False))
        rp = np.insert(rp, cut, rp[cut])
        if k is not None:
            k = np.insert(k, cut, 1).astype(np.uint32)
    pk = orc.Problem(rp, ci, p.l * float(rng.choice([0.01, 1.0, 50.0])), k=k)
    env = {}
    if rng.integers(0, 4) == 0: env["MMG_FORCE_IDX64"] = "1"
    kern = int(rng.choice([2, 2, 1, 0]))
    if kern <= 1: env["MMG_K1_SELL"] = "0"
    if kern == 0: env["MMG_K1_S16"] = "0"
    if rng.integers(0, 3) == 0: env["MMG_K1_SELL_WAVES_PER_CU"] = "1"
    if rng.integers(0, 3) == 0: env["MMG_EM_GRID"] = str(int(rng.integers(1, 9)))
    if rng.integers(0, 3) == 0: env["MMG_EM_WAVES"] = "1"
    em_stream = rng.choice(["", "1", "0"])
    if em_stream: env["MMG_EM_STREAM"] = str(em_stream)
    keys = ["MMG_FORCE_IDX64", "MMG_K1_SELL", "MMG_K1_S16", "MMG_K1_SELL_WAVES_PER_CU", "MMG_EM_GRID", "MMG_EM_STREAM", "MMG_EM_WAVES"]
    for key in keys: os.environ.pop(key, None)
    os.environ.update(env)
    try:
        mu0, uh = orc.start_values(pk)
        if rng.integers(0, 3) == 0: mu0[rng.integers(0, T, size=max(1, T // 20))] = 0.0
        if rng.integers(0, 4) == 0: mu0 *= 10.0 ** rng.uniform(-120, 120, size=T)
        rid0 = int(rng.choice([0, 0, 12345, (1 << 32) - 100, (1 << 40) + 7]))    # shard offset: crosses 2^32 in the middle of a problem
        prob = gpu.Problem.from_csr(pk.row_ptr, pk.col_idx, pk.l, k=pk.k, row_id_base=rid0)
        g_mu0, g_uh = prob.start_values()
        assert np.array_equal(g_uh, uh), "unique hits"
        n_it = int(rng.integers(1, 6))
        chains = int(rng.choice([1, 1, 3]))
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
            print("seed %d ok: R=%d T=%d avg=%g sort=%s k=%s kernel=%d env=%s" % (seed, pk.m, T, avg, sort, k is not None, info.sample_kernel, env), flush=True)
    finally:
        for key in keys: os.environ.pop(key, None)


if __name__ == "__main__":
    from mmseq_amd import gibbs as gpu
    from oracle import binding as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for seed in range(s0, s0 + n):
        one_case(seed, gpu, orc)
    print("all %d cases bit-identical" % n)
