"""EM probe (GPU box): time EM sweeps (mmg_em_step) on the benchmark shapes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem

def probe(R, T, avg, sweeps=40):
    prob = Problem.synthetic(R, T, avg, seed=1234, sort=True, uniform=bool(int(os.environ.get('MMG_PROBE_UNIFORM', '0'))))
    inf = prob.info
    mu0, _ = prob.start_values()
    t0 = time.time()
    em = prob.em_stepper(mu0)
    t1 = time.time()
    for _ in range(3):
        em.step()
    t2 = time.time()
    for _ in range(sweeps):
        em.step()
    t3 = time.time()
    per = (t3 - t2) / sweeps
    B = 4 * (inf.m + 1) + 4 * inf.nnz + 28 * T
    print(f"EM R={R} T={T} avg={avg}: create (column counts + measured first pass) {t1-t0:.3f}s; per sweep {per*1e3:.3f} ms "
          f"({B/per/1e9:.0f} GB/s of CSR-equivalent bytes); loglik {em.loglik:.6f}; {em.stats()}", flush=True)
    em.close(); prob.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "2"
    if "2" in which: probe(5_000_000, 50_000, 8)
    if "3" in which: probe(50_000_000, 200_000, 20)
