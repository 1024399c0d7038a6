"""Create / use / destroy every kind of handle many times and watch the free device memory (torch.cuda.mem_get_info): a leak in a create or
an error path shows as a drift.  usage: soak.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmseq_amd import gibbs as G


class _Csr:
    """rows of 1 + Poisson(6) distinct hits within +-64 of a random centre, every tenth row with one hit anywhere"""
    def __init__(self, R, T, rng):
        rows = []
        for i in range(R):
            c = int(rng.integers(0, T))
            L = 1 + int(rng.poisson(6))
            h = np.unique(np.clip(c + rng.integers(-64, 65, size=L), 0, T - 1))
            if i % 10 == 0:
                h = np.unique(np.append(h, rng.integers(0, T)))
            rows.append(h.astype(np.uint32))
        self.m, self.n = R, T
        self.row_ptr = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.uint64)
        self.col_idx = np.concatenate(rows)
        self.l = rng.uniform(0.5, 3.0, T)


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(1)
p = _Csr(60000, 9000, rng)
k = rng.choice([1, 1, 1, 2, 6, 80, 400], size=p.m).astype(np.uint32)
scat = rng.permutation(p.n).astype(np.uint32)
l_ext = np.empty(p.n); l_ext[scat] = p.l
free0 = None
for r in range(rounds):
    prob = G.Problem.from_csr(p.row_ptr, scat[p.col_idx], l_ext, k=k)            # derived order: two builds, one discarded
    mu0, _ = prob.start_values()
    b = prob.shard_bounds_timed(mu0, 4)
    shards = [prob.shard(int(b[i]), int(b[i + 1])) for i in range(4)]
    smps = [G.Sampler(sh, mu0, seed=5, gibbs_iter=8, trace_len=8, timing=1) for sh in shards]
    G.gibbs_shards_selftest(smps, 8)
    _ = [s.timing() for s in smps]
    mu, ll, rep = G.em_shards_selftest(shards, mu0, 3)
    whole = G.Sampler(prob, mu0, seed=5, n_chains=3, gibbs_iter=8, trace_len=8)
    q = G.Summary(whole, staged=True, genes=[[0, 1, 2], [3]], identical=[[4, 5]], percentile_index=[1, 6])
    whole.run(4); whole.sync(); q.advance(4); _ = q.rows(G.SERIES_GENE, 0, 4); _ = whole.trace_rows_done(0, 0, 4)
    whole.run(4); whole.sync(); q.advance(8); q.finish(); _ = q.series(G.SERIES_GENE)
    try:
        q.advance(3)                                                              # an error path
    except Exception:
        pass
    em = prob.em_stepper(mu0); em.step(); em.close()
    grp = G.Group([0]); grp.run_sharded([whole], 0); grp.close()
    q.close(); whole.close()
    for s in smps: s.close()
    for sh in shards: sh.close()
    prob.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if r == 4:
        free0 = free
    if r % 10 == 9 or r == rounds - 1:
        print("round %3d: free %.1f MB (drift since round 5: %+.2f MB)" % (r + 1, free / 1e6, ((free - free0) / 1e6) if free0 else 0.0), flush=True)
assert free0 is not None and abs(free - free0) < 64e6, "device memory drifted by %.1f MB" % ((free - free0) / 1e6)
print("no drift")
