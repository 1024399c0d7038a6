"""Quick perf probe (GPU box): time K1/K2 on the benchmark shapes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler

def probe(R, T, avg, iters=20, chains=1, uniform=False, sort=True):
    t0 = time.time()
    prob = Problem.synthetic(R, T, avg, seed=1234, uniform=uniform, sort=sort)
    inf = prob.info
    t1 = time.time()
    mu0, uh = prob.start_values()
    t2 = time.time()
    s = Sampler(prob, mu0, n_chains=chains, gibbs_iter=1024, trace_len=1024, keep_trace=True, timing=True)
    s.run(4); s.sync(); s.reset_timing()
    t3 = time.time()
    s.run(iters); s.sync()
    t4 = time.time()
    tm = s.timing()
    B = 4 * (inf.m + 1) + 4 * inf.nnz + 28 * chains * T
    k1 = tm["sample_ms"] / tm["sample_launches"]
    k2 = tm["update_ms"] / tm["update_launches"]
    wall = (t4 - t3) / iters * 1e3
    print(f"R={R} T={T} avg={avg} C={chains} uni={uniform} sort={sort} nnz={inf.nnz} tiles={inf.n_tiles} gen={t1-t0:.1f}s start={t2-t1:.2f}s "
          f"K1={k1:.3f}ms K2={k2:.3f}ms wall/iter={wall:.3f}ms  B={B/1e6:.1f}MB  K1 GB/s={(4*(inf.m+1)+4*inf.nnz)/k1/1e6:.0f} "
          f"iter/s={1e3/wall:.1f} chain-it/s={chains*1e3/wall:.1f} frac8TB={B/(wall*1e-3)/8e12:.3f}", flush=True)
    cnt = s.counts(0)
    assert int(cnt.sum()) == inf.total_k, (cnt.sum(), inf.total_k)
    s.close(); prob.close()

if __name__ == "__main__":
    probe(5_000_000, 50_000, 8)
    probe(5_000_000, 50_000, 8, sort=False)
    probe(50_000_000, 200_000, 20, iters=10)
    probe(50_000_000, 200_000, 20, iters=5, sort=False)
