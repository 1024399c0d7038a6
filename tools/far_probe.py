import sys, os
sys.path.insert(0, '/root/repo')
import torch
from mmseq_amd import Problem, Sampler
far = float(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prob = Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234, far_fraction=far)
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, n_chains=C, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(100); s.sync(); s.reset_timing(); s.run(50); s.sync()
tm = s.timing(); inf = prob.info
print("far %.2f chains %d: sample() %.4f ms, tiles %d fast %d far %d" % (far, C, tm["sample_ms"] / tm["sample_launches"], inf.n_tiles, inf.fast_tiles, inf.far_tiles))
