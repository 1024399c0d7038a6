"""Time of mmg_problem_create from HOST arrays (what the CLI pays): build_probe.py [rows transcripts avg]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem
R, T, A = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (50_000_000, 200_000, 20.0)
p0 = Problem.synthetic(R, T, A, seed=1234)
rp, ci = p0.download()
l = p0.l()
p0.close()
k = np.ones(R, np.uint32); k[::16] = 2
for rep in range(2):
    t0 = time.perf_counter()
    p = Problem.from_csr(rp, ci, l, k=k)
    t1 = time.perf_counter()
    print("from_csr with k: %.3f s  (rows %d, hits %d, stored rows %d)" % (t1 - t0, R, ci.size, p.info.m), flush=True)
    p.close()
t0 = time.perf_counter()
p = Problem.from_csr(rp, ci, l)
print("from_csr without k: %.3f s" % (time.perf_counter() - t0), flush=True)
