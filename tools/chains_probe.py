"""BASELINE configs[2] (8 chains in one GPU): chain-iterations/s with chains advanced 1 / 2 / 4 per K1 launch, same box, same problem."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmseq_amd import gibbs as G
R, T, A = 50_000_000, 200_000, 20.0
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prob = G.Problem.synthetic(R, T, A, seed=1234)
mu0, _ = prob.start_values()
for fuse in (1, 2, 4, 1, 2, 4):
    with G.options(fuse_chains=fuse):
        s = G.Sampler(prob, mu0, n_chains=C, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
        s.run(40); s.sync(); s.reset_timing()
        t0 = time.perf_counter()
        s.run(40); s.sync()
        el = time.perf_counter() - t0
        tm = s.timing()
        print("fuse %d: %.0f chain-iterations/s, K1 for all %d chains %.3f ms (%.4f per chain), K2 %.3f ms" % (
            fuse, C * 40 / el, C, tm["sample_ms"] / tm["sample_launches"], tm["sample_ms"] / tm["sample_launches"] / C, tm["update_ms"] / tm["update_launches"]), flush=True)
        s.close()
