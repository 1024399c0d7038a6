"""K1 / EM timing when a fraction of the rows has hits far outside any window (real data: reads hitting paralogues)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler
R, T, avg = 5_000_000, 50_000, 8
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
base = Problem.synthetic(R, T, avg, seed=1234, sort=True)
rp, ci = base.download(); l = base.l(); base.close()
rp = rp.astype(np.int64)
rng = np.random.default_rng(1)
far = rng.random(R) < frac
last = rp[1:] - 1                                   # replace the last hit of a far row by a random distant transcript
idx = np.where(far & (np.diff(rp) >= 2))[0]
ci = ci.copy()
off = rng.integers(400, 20000, size=idx.size)
lead = ci[rp[idx]].astype(np.int64)
up = lead + off < T
ci[last[idx[up]]] = (lead[up] + off[up]).astype(np.uint32)        # larger than every other hit of the row
low = idx[~up]                                                     # distant transcript BELOW the row: it becomes the first hit
if low.size:
    lens_low = (rp[low + 1] - rp[low]).astype(np.int64)
    dst = np.repeat(rp[low], lens_low - 1) + 1 + (np.arange(int((lens_low - 1).sum())) - np.repeat(np.cumsum(lens_low - 1) - (lens_low - 1), lens_low - 1))
    vals = ci[dst - 1].copy()
    ci[dst] = vals
    ci[rp[low]] = (lead[~up] - off[~up]).astype(np.uint32)
# rows stay ascending; sort wide rows behind the rest
span = ci[last].astype(np.int64) - ci[rp[:-1]].astype(np.int64)
order = np.lexsort((np.diff(rp), ci[rp[:-1]], (span >= 160).astype(np.int64)))
lens = np.diff(rp)[order]
nrp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
gather = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in order[:0]]) if False else None
src = np.repeat(rp[:-1][order], lens) + (np.arange(int(lens.sum())) - np.repeat(nrp[:-1].astype(np.int64), lens))
nci = ci[src]
print("far rows: %.3f" % (span >= 160).mean(), flush=True)
for env in ({}, {"MMG_K1_SELL": "0"}, {"MMG_K1_SELL": "0", "MMG_K1_S16": "0"}):
    for k in ("MMG_K1_SELL", "MMG_K1_S16"): os.environ.pop(k, None)
    os.environ.update(env)
    prob = Problem.from_csr(nrp, nci, l)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, timing=True)
    s.run(4); s.sync(); s.reset_timing(); s.run(20); s.sync()
    tm = s.timing()
    em = prob.em_stepper(mu0)
    for _ in range(3): em.step()
    t0 = time.time()
    for _ in range(20): em.step()
    print("kernel", prob.info.sample_kernel, "K1 %.3f ms" % (tm["sample_ms"] / tm["sample_launches"]), "EM %.3f ms" % ((time.time() - t0) / 20 * 1e3), em.stats_raw()["stream_kernel"], flush=True)
    em.close(); s.close(); prob.close()
