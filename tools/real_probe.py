"""The 50 M x 200 k workload as a real hits file has it (multiplicities of a collapsed file + 2 % rows with a far hit), C chains:
ms per sweep; under `rocprofv3 --kernel-trace --stats` the per-launch times of the three kernels.  usage: real_probe.py [chains] [far]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmseq_amd import Problem, Sampler
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
far = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
rows = 50_000_000
prob = Problem.synthetic(rows, 200_000, 20.0, seed=1234, far_fraction=far, mapped_reads=rows)
rp, ci = prob.download(); l = prob.l(); prob.close()
rng = np.random.default_rng(1234)
u = rng.random(rows)
k = np.ones(rows, np.uint32)
for thr, val in ((0.064, 2), (0.011, 3), (0.0035, 4), (0.002, 6)):
    k[u < thr] = val
big = u < 0.0012
k[big] = rng.integers(9, 37, size=int(big.sum())).astype(np.uint32)
prob = Problem.from_csr(rp, ci, l, k=k)
del rp, ci, k, u
inf = prob.info
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, n_chains=C, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(20); s.sync(); s.reset_timing()
t0 = time.perf_counter(); s.run(20); s.sync(); el = time.perf_counter() - t0
tm = s.timing()
print("chains %d far %.2f: %.4f ms per sweep (%.0f chain-it/s), K1 launches %.4f ms, tiles %d fast %d far %d" % (
    C, far, el / 20 * 1e3, C * 20 / el, tm["sample_ms"] / tm["sample_launches"], inf.n_tiles, inf.fast_tiles, inf.far_tiles))
