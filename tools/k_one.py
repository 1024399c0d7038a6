"""Config 3 with the multiplicities of a collapsed hits file, one process (for rocprofv3 --kernel-trace: the two launches of a sweep
show up as k_sample_sell<.., false> and k_sample_sell<.., true>)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler
R, T, A = 50_000_000, 200_000, 20.0
prob = Problem.synthetic(R, T, A, seed=1234)
rp, ci = prob.download(); l = prob.l(); prob.close()
rng = np.random.default_rng(1234)
u = rng.random(R)
k = np.ones(R, np.uint32)
for thr, val in ((0.064, 2), (0.011, 3), (0.0035, 4), (0.002, 6)):
    k[u < thr] = val
big = u < 0.0012
k[big] = rng.integers(9, 37, size=int(big.sum())).astype(np.uint32)
if len(sys.argv) > 1 and sys.argv[1] == "nobig":
    k[big] = 2
prob = Problem.from_csr(rp, ci, l, k=k)
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(40); s.sync(); s.reset_timing()
s.run(40); s.sync()
tm = s.timing()
print("K1 (both launches) %.4f ms" % (tm["sample_ms"] / tm["sample_launches"]))
