import sys, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from mmseq_amd import Problem, Sampler
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R, T, A = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (50_000_000, 200_000, 20.0)
F = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
prob = Problem.synthetic(R, T, A, seed=1234, far_fraction=F)
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, n_chains=C, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(300); s.sync(); s.reset_timing()
s.run(100); s.sync()
tm = s.timing()
print("chains", C, "K1 %.4f ms" % (tm["sample_ms"] / tm["sample_launches"]), "K2 %.4f" % (tm["update_ms"] / tm["update_launches"]))
