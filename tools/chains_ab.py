"""A/B of the fused-chain kernel between library builds on one box: chains_ab.py lib1.so lib2.so ... (8 chains, fuse 2)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from mmseq_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from mmseq_amd import gibbs as G
prob = G.Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234)
mu0, _ = prob.start_values()
for fuse in (1, 2):
    with G.options(fuse_chains=fuse):
        s = G.Sampler(prob, mu0, n_chains=8, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
        s.run(40); s.sync(); s.reset_timing()
        t0 = time.perf_counter(); s.run(40); s.sync(); el = time.perf_counter() - t0
        print("%%s fuse %%d: %%.0f chain-it/s" %% (sys.argv[1].split("/")[-1], fuse, 8 * 40 / el), flush=True)
        s.close()
''' % ROOT
for r in range(2):
    for l in sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l)], capture_output=True, text=True)
        print(out.stdout.strip() or out.stderr[-400:], flush=True)
