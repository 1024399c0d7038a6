"""K1 on rows that all carry a large multiplicity (the conditional-binomial path, src/mmseq.cpp:880): bigk_probe.py [rows transcripts avg k]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler
R, T, A, K = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2_000_000, 200_000, 20.0, 1000)
p0 = Problem.synthetic(R, T, A, seed=1234)
rp, ci = p0.download(); l = p0.l(); p0.close()
rng = np.random.default_rng(1)
for name, k in (("k = %d on every row" % K, np.full(R, K, np.uint32)), ("k = 65..%d uniform" % K, rng.integers(65, K + 1, size=R).astype(np.uint32)),
                ("k = 1 (no array)", None)):
    prob = Problem.from_csr(rp, ci, l, k=k)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, n_chains=1, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(60); s.sync(); s.reset_timing(); s.run(40); s.sync()
    tm = s.timing(); inf = prob.info
    print("%-28s K1 %.4f ms  K2 %.4f ms   rows %d hits %d tiles %d" % (name, tm["sample_ms"] / tm["sample_launches"], tm["update_ms"] / tm["update_launches"], inf.m, inf.nnz, inf.n_tiles), flush=True)
    s.close(); prob.close()
