"""Perf probe (GPU box): time K1/K2 on the benchmark shapes for each K1 variant, with a parity check."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler

def parity(variant):
    from oracle import binding as B
    p, _ = B.synth_problem(R=30000, T=3000, avg_hits=8, seed=7)
    mu0, _ = B.start_values(p)
    prob = Problem.from_csr(p.row_ptr, p.col_idx, p.l)
    s = Sampler(prob, mu0, seed=3, gibbs_iter=8, trace_len=8)
    s.run(8)
    ref = B.gibbs_keyed(p, mu0, seed=3, n_iter=8, trace_len=8)
    ok = np.array_equal(s.trace(0), ref["trace"]) and np.array_equal(s.counts(0), ref["cnt"])
    s.close(); prob.close()
    return ok

def probe(R, T, avg, iters=20, chains=1, uniform=0, sort=True, tag="", check=True):
    t0 = time.time()
    prob = Problem.synthetic(R, T, avg, seed=1234, uniform=int(os.environ.get('MMG_PROBE_UNIFORM', uniform)), sort=sort)
    inf = prob.info
    t1 = time.time()
    mu0, uh = prob.start_values()
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
    print(f"{tag} kernel={inf.sample_kernel} stream={inf.stream_bytes/1e9:.3f}GB R={R} T={T} avg={avg} C={chains} sort={sort} tiles={inf.n_tiles} gen={t1-t0:.1f}s "
          f"K1={k1:.3f}ms K2={k2:.3f}ms wall/iter={wall:.3f}ms K1 GB/s={(4*(inf.m+1)+4*inf.nnz)/k1/1e6:.0f} "
          f"iter/s={1e3/wall:.1f} chain-it/s={chains*1e3/wall:.1f} frac8TB={B/(wall*1e-3)/8e12:.3f}", flush=True)
    if check:
        cnt = s.counts(0)
        assert int(cnt.sum()) == inf.total_k, (cnt.sum(), inf.total_k)
    s.close(); prob.close()

if __name__ == "__main__":
    specs = (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4").split(",")
    chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for spec in specs:
        v, _, bpc = spec.partition(":")
        v = int(v)
        os.environ["MMG_K1_VARIANT"] = str(v)
        if bpc:
            os.environ["MMG_K1_BLOCKS_PER_CU"] = bpc
        else:
            os.environ.pop("MMG_K1_BLOCKS_PER_CU", None)
        ok = parity(v) if v != 5 else None
        tag = f"[v{spec} s16={os.environ.get('MMG_K1_S16', '1')} parity={ok}]"
        probe(5_000_000, 50_000, 8, tag=tag, check=(v != 5), chains=chains)
        probe(50_000_000, 200_000, 20, iters=10, tag=tag, check=(v != 5), chains=chains)
