"""EM sweep timing on the device (mmg_em_*): em_probe.py [rows transcripts avg_hits [far_fraction]]; sliced-ELL kernel and the
row-per-thread kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmseq_amd import gibbs as G
R, T, A = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (50_000_000, 200_000, 20.0)
F = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
prob = G.Problem.synthetic(R, T, A, seed=1234, far_fraction=F)
mu0, _ = prob.start_values()
for kern in (-1, 0):
    with G.options(em_kernel=kern):
        em = prob.em_stepper(mu0)
    for _ in range(3): em.step()
    t0 = time.time()
    for _ in range(20): em.step()
    print("EM kernel %d: %.3f ms per sweep, loglik %.6g, repeats %d" % (em.stats_raw()["stream_kernel"], (time.time() - t0) / 20 * 1e3, em.loglik,
                                                                       em.stats()["repeated_passes"]), flush=True)
    em.close()
