import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem
from oracle import binding as B
R, T, avg = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
prob = Problem.synthetic(R, T, avg, seed=1234, sort=True)
rp, ci = prob.download()
l = prob.l() if hasattr(prob, "l") else None
mu0, _ = prob.start_values()
print("info", prob.info.n_tiles, flush=True)
p = B.Problem(rp, ci, l)
try:
    em = prob.em_stepper(mu0)
    print("create ok", em.loglik, em.stats())
    for i in range(3):
        em.step()
    mu_g = em.mu()
    mu_o, _, ll_o, redo = B.em_x(p, mu0, max_iter=3, epsilon=-1e308)
    print("equal", np.array_equal(mu_g, mu_o), em.loglik == ll_o, redo, em.stats())
except Exception as e:
    print("FAILED", e)
    os.environ["MMG_EM_STREAM"] = "0"
    em = prob.em_stepper(mu0)
    print("fallback create ok", em.loglik, em.stats())
