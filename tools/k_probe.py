"""K1 with multiplicities (what every real hits file produces: identical reads collapse to rows with k > 1, src/mmseq.cpp:409-418):
the config-3 workload with a fraction of the rows given k in 2..3 (and a few large), against the k = 1 kernel.
usage: k_probe.py [rows transcripts avg [fraction_with_k]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler

R, T, A = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (50_000_000, 200_000, 20.0)
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
base = Problem.synthetic(R, T, A, seed=1234)
rp, ci = base.download()
l = base.l()
rng = np.random.default_rng(1)
for name, k in (("k = 1 (no k array)", None),
                ("%g of the rows k in 2..3" % frac, np.where(rng.random(R) < frac, rng.integers(2, 4, size=R), 1).astype(np.uint32)),
                ("%g of the rows k in 2..3, 1e-4 k = 100" % frac, None)):
    if name.endswith("100"):
        k = np.where(rng.random(R) < frac, rng.integers(2, 4, size=R), 1).astype(np.uint32)
        k[rng.random(R) < 1e-4] = 100
    prob = Problem.from_csr(rp, ci, l, k=k)
    inf = prob.info
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(30); s.sync(); s.reset_timing()
    s.run(30); s.sync()
    tm = s.timing()
    print("%-45s K1 %.3f ms  K2 %.3f ms  tiles %d fast %d, grid %d" % (name, tm["sample_ms"] / tm["sample_launches"], tm["update_ms"] / tm["update_launches"],
                                                                   inf.n_tiles, inf.fast_tiles, inf.sample_grid), flush=True)
    del s, prob
