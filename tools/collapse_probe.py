"""Why is K1 slow on a collapsed 50 M-read problem?  Generator rows -> identical rows merged (k = multiplicity), as the CLI's ingest
does (src/mmseq.cpp:409-418) -> K1 timing, then with the large multiplicities clipped.  usage: collapse_probe.py [rows transcripts avg]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler

R, T, A = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (50_000_000, 200_000, 20.0)
base = Problem.synthetic(R, T, A, seed=1234, sort=False)
rp, ci = base.download()
l = base.l()
del base
t0 = time.time()
rp64 = rp.astype(np.int64)
L = np.diff(rp64)
h = L.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
with np.errstate(over="ignore"):
    for j in range(int(L.max())):
        sel = np.nonzero(L > j)[0]
        hj = (h[sel] ^ ci[rp64[sel] + j].astype(np.uint64)) * np.uint64(0xFF51AFD7ED558CCD)
        h[sel] = hj ^ (hj >> np.uint64(32))
u, first, cnt = np.unique(h, return_index=True, return_counts=True)
order = np.sort(first)                      # first-seen order of the distinct rows
k_of = dict()
kk = cnt[np.argsort(first)].astype(np.uint32)
lens = L[order]
nrp = np.zeros(order.size + 1, np.uint64); nrp[1:] = np.cumsum(lens)
idx = np.repeat(rp64[order] - nrp[:-1].astype(np.int64), lens) + np.arange(int(lens.sum()), dtype=np.int64)
nci = np.ascontiguousarray(ci[idx])
print("collapsed %d reads to %d rows in %.0f s; k: max %d, rows with k > 8: %d, k > 100: %d" % (R, order.size, time.time() - t0, kk.max(), (kk > 8).sum(), (kk > 100).sum()), flush=True)
big = kk > 8
variants = (("k as collapsed", kk), ("k clipped to 8", np.minimum(kk, 8)), ("k clipped to 2", np.minimum(kk, 2)),
            ("k = 1, array present", np.minimum(kk, 1)), ("k = 2 for every row", np.full(kk.size, 2, np.uint32)),
            ("k = 9 where collapsed k > 8, else 1", np.where(big, 9, 1).astype(np.uint32)),
            ("k as collapsed where > 8, else 1", np.where(big, kk, 1).astype(np.uint32)), ("no k array", None))
if len(sys.argv) > 4:
    variants = [v for v in variants if sys.argv[4] in v[0]]
for name, k in variants:
    prob = Problem.from_csr(nrp, nci, l, k=k)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(20); s.sync(); s.reset_timing()
    s.run(20); s.sync()
    tm = s.timing()
    print("%-32s K1 %.3f ms  (tiles %d, grid %d)" % (name, tm["sample_ms"] / tm["sample_launches"], prob.info.n_tiles, prob.info.sample_grid), flush=True)
    del s, prob
