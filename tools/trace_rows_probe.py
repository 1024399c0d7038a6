import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from mmseq_amd import Problem, Sampler
prob = Problem.synthetic(2_000_000, 200_000, 20.0, seed=1)
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=True)
s.run(1024); s.sync()
for _ in range(2):
    t0 = time.time(); r = s.trace_rows(0); t1 = time.time()
    print("trace_rows %.3f s for %.2f GB -> %.2f GB/s" % (t1 - t0, r.nbytes / 1e9, r.nbytes / 1e9 / (t1 - t0)))
