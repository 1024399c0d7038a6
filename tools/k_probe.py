"""K1 timing on a problem WITH multiplicities (collapsed hit sets): config-2 shape, k drawn from a heavy-tailed law."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler
R, T, avg = 5_000_000, 50_000, 8
base = Problem.synthetic(R, T, avg, seed=1234, sort=True)
rp, ci = base.download(); l = base.l(); base.close()
rng = np.random.default_rng(1)
k = np.minimum(rng.zipf(1.7, size=R), 100000).astype(np.uint32)   # ~60 % ones, a tail of large multiplicities
print("k: mean %.2f, ==1 %.2f, >8 %.3f, max %d" % (k.mean(), (k == 1).mean(), (k > 8).mean(), k.max()))
for env in ({}, {"MMG_K1_SELL": "0"}):
    os.environ.pop("MMG_K1_SELL", None); os.environ.update(env)
    prob = Problem.from_csr(rp, ci, l, k=k)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, timing=True)
    s.run(4); s.sync(); s.reset_timing(); s.run(20); s.sync()
    tm = s.timing()
    print("kernel", prob.info.sample_kernel, "K1 %.3f ms" % (tm["sample_ms"] / tm["sample_launches"]), "digest", int(s.counts(0).astype(np.int64).sum()), hash(s.counts(0).tobytes()) & 0xffffffff)
    s.close(); prob.close()
    if not env:
        import time
        prob = Problem.from_csr(rp, ci, l, k=k)
        mu0, _ = prob.start_values()
        em = prob.em_stepper(mu0)
        for _ in range(3): em.step()
        t0 = time.time()
        for _ in range(40): em.step()
        print("EM with multiplicities: %.3f ms per sweep" % ((time.time() - t0) / 40 * 1e3), em.stats_raw())
        em.close(); prob.close()
