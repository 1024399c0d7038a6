"""A/B of libmmgibbs builds on the config-3 workload WITH multiplicities (k distribution of a collapsed 50 M-read file): K1 = both launches.
usage: k_ab.py lib_a.so lib_b.so ..."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from mmseq_amd import _lib
_lib.LIB_PATH = sys.argv[1]
_lib._share_hip_runtime_with_torch()
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
prob = Problem.from_csr(rp, ci, l, k=k)
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(40); s.sync(); s.reset_timing()
s.run(40); s.sync()
tm = s.timing()
print(json.dumps({"k1_ms": tm["sample_ms"] / tm["sample_launches"]}))
''' % ROOT
for l in sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l)], capture_output=True, text=True)
    line = [x for x in out.stdout.splitlines() if x.startswith("{")]
    print(os.path.basename(l), json.loads(line[-1]) if line else ("FAILED " + out.stderr[-400:]), flush=True)
