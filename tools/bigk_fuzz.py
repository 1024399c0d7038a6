"""Differential fuzz of the rows on the conditional-binomial chain (k_sample_bigk, bigk_kernels.h: the state machine, the bound-settled
inversion, the fp32 exact test and fp32 search with their fp64 paths) against the oracle's sequential loop, bit for bit: random rows of 2 ...
60 hits, k log-uniform from 17 to 2 10^8, weights spread over 1 ... 12 orders of magnitude, zero / tiny / infinite weights, two chains.
usage: bigk_fuzz.py [n_cases] [first_seed]      (GPU box; the oracle is the checker)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one_case(seed, gpu, orc):
    rng = np.random.default_rng(seed)
    T = int(rng.choice([600, 5000, 40000]))
    n_rows = int(rng.integers(300, 3000))
    span = int(rng.choice([60, 200, 5000]))                       # how far a row's hits lie apart: window rows / far rows
    lmax = int(rng.choice([3, 8, 60]))
    kmax = float(rng.choice([300, 2e4, 1e6, 2e8]))
    rows, ks = [], []
    budget = (1 << 31) - 1
    for _ in range(n_rows):
        L = int(rng.integers(2, lmax + 1))
        lead = int(rng.integers(0, max(1, T - span)))
        rows.append(rng.choice(np.arange(lead, min(T, lead + span)), size=min(L, min(T, lead + span) - lead), replace=False).tolist())
        k = int(np.exp(rng.uniform(np.log(17.0), np.log(kmax))))
        k = min(k, budget // 4)                                     # (a transcript's count is an int32: the whole problem stays below 2^31 reads)
        budget -= k
        if budget < 1000:
            break
        ks.append(k)
    rows = rows[:len(ks)]
    if rng.integers(0, 3) == 0:
        rows = [sorted(r) for r in rows]
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    p = orc.Problem(rp, ci, np.exp(rng.normal(0.0, 0.5, size=T)), k=np.asarray(ks, np.uint32))
    mu0 = np.exp(rng.normal(0.0, float(rng.choice([0.3, 2.0, 6.0, 14.0])), size=T))
    if rng.integers(0, 2):
        mu0[rng.integers(0, T, size=T // 7)] = float(rng.choice([0.0, 1e-300, 1e-30]))
    if rng.integers(0, 4) == 0:
        mu0[rng.integers(0, T, size=5)] = np.inf
    opts = dict(sample_kernel=2, bigk_side_stream=int(rng.integers(0, 2)))
    if rng.integers(0, 2):
        opts["bigk_per_wave"] = int(rng.choice([1, 9, 64, 500]))
    chains, iters = int(rng.choice([1, 2])), int(rng.choice([2, 5]))
    with gpu.options(**opts):
        prob = gpu.Problem.from_csr(p.row_ptr, p.col_idx, p.l, k=p.k)
        d_rp, d_ci, d_k = prob.download(with_k=True)
        ps = orc.Problem(d_rp, d_ci, p.l, k=d_k)
        s = gpu.Sampler(prob, mu0, seed=seed, n_chains=chains, gibbs_iter=iters, trace_len=iters)
        s.run(iters)
        ok = True
        for c in range(chains):
            ref = orc.gibbs_keyed(ps, mu0, seed=seed, chain=c, n_iter=iters, trace_len=iters)
            ok = ok and np.array_equal(s.counts(c), ref["cnt"]) and np.array_equal(s.trace(c), ref["trace"])
        s.close(); prob.close()
    return ok, "rows %d T %d hits<=%d k<=%g span %d chains %d iters %d opts %s" % (len(ks), T, lmax, kmax, span, chains, iters, opts)


if __name__ == "__main__":
    from mmseq_amd import gibbs as gpu
    from oracle import binding as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    for seed in range(first, first + n):
        ok, what = one_case(seed, gpu, orc)
        print("seed %d %s: %s" % (seed, "ok" if ok else "DIFFERENT", what), flush=True)
        bad += 0 if ok else 1
    print("all %d cases bit-identical" % n if bad == 0 else "%d of %d cases DIFFER" % (bad, n))
    sys.exit(1 if bad else 0)
