import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from mmseq_amd import Problem, Sampler
R, T, A = 2_000_000, 200_000, 20.0
p0 = Problem.synthetic(R, T, A, seed=1234)
rp, ci = p0.download(); l = p0.l(); p0.close()
for K, keep in ((64, True), (32, True), (65, True), (65, False), (200, False)):
    prob = Problem.from_csr(rp, ci, l, k=np.full(R, K, np.uint32), keep_rows=keep)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, n_chains=1, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(30); s.sync(); s.reset_timing(); s.run(20); s.sync()
    tm = s.timing(); inf = prob.info
    print("k=%d keep_rows=%s  K1 %.4f ms  stored rows %d tiles %d" % (K, keep, tm["sample_ms"] / tm["sample_launches"], inf.m, inf.n_tiles), flush=True)
    s.close(); prob.close()
