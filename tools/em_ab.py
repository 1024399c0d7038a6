import os, subprocess, sys
ROOT = "/root/repo"
CHILD = r'''
import sys, time
sys.path.insert(0, "/root/repo")
from mmseq_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from mmseq_amd import gibbs as G
prob = G.Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234)
mu0, _ = prob.start_values()
em = prob.em_stepper(mu0)
for _ in range(3): em.step()
t0 = time.time()
for _ in range(20): em.step()
print("%s EM %.3f ms per sweep" % (sys.argv[1].split("/")[-1], (time.time() - t0) / 20 * 1e3), flush=True)
'''
for l in sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l)], capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-300:], flush=True)
