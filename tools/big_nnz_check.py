"""One-off check of the 64-bit offset path at REAL size (BASELINE configs[4] needs it; the test suite only forces it on small problems):
a synthetic problem with nnz >= 2^32 on one MI355X (216 M rows x 800 k transcripts, avg 20 hits: 4.3 G hits, 17 GB of column ids).
  (a) index_bits == 64 without any override, (b) every read assigned exactly once in every sweep, (c) a rerun gives the same bits,
  (d) the CSR-tile kernel (64-bit row offsets too) gives the same bits as the sliced-ELL kernel, (e) with --oracle: the first sweep
  bit for bit against the CPU oracle on the downloaded problem (needs ~40 GB of host memory).
usage: big_nnz_check.py [--rows N] [--oracle]"""
import argparse, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmseq_amd import gibbs as G

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=216_000_000)
ap.add_argument("--oracle", action="store_true")
a = ap.parse_args()
T = a.rows // 270
dig = lambda *xs: hashlib.sha256(b"".join(np.ascontiguousarray(x).tobytes() for x in xs)).hexdigest()[:16]
t0 = time.time()
prob = G.Problem.synthetic(a.rows, T, 20.0, seed=1234, sort=True)
inf = prob.info
print("rows %d transcripts %d hits %d (2^32 = %d): index bits %d, sample kernel %d, %d tiles (%d on the register path), stream %.2f GB, built in %.1f s"
      % (inf.m, inf.n, inf.nnz, 1 << 32, inf.index_bits, inf.sample_kernel, inf.n_tiles, inf.fast_tiles, inf.stream_bytes / 1e9, time.time() - t0), flush=True)
assert inf.nnz >= 1 << 32 and inf.index_bits == 64 and inf.sample_kernel == 2
mu0, uh = prob.start_values()
n_it = 3
s = G.Sampler(prob, mu0, seed=5, gibbs_iter=n_it, trace_len=n_it)
sums = []
t0 = time.time()
for _ in range(n_it):
    s.sample(); sums.append(int(s.counts(0).astype(np.int64).sum())); s.update()
print("sum of counts per sweep:", sums, "(rows %d)  %.2f s for %d sweeps incl. read-back" % (inf.m, time.time() - t0, n_it), flush=True)
assert sums == [inf.m] * n_it
d1 = dig(s.trace(0), s.counts(0)); s.close()
s = G.Sampler(prob, mu0, seed=5, gibbs_iter=n_it, trace_len=n_it); s.run(n_it)
tr1, cn1 = s.trace(0), s.counts(0)
assert dig(tr1, cn1) == d1; s.close()
print("rerun: same bits (%s)" % d1, flush=True)
em = prob.em_stepper(mu0)
lls = [em.loglik] + [em.step() for _ in range(3)]
assert all(b >= a_ for a_, b in zip(lls, lls[1:]))
print("EM log-likelihood over 3 sweeps:", lls, flush=True)
em.close()
if a.oracle:
    from oracle import binding as B
    t0 = time.time()
    rp, ci = prob.download()
    p = B.Problem(rp, ci, prob.l())
    ref = B.gibbs_keyed(p, mu0, seed=5, n_iter=1, trace_len=1)
    s = G.Sampler(prob, mu0, seed=5, gibbs_iter=1, trace_len=1); s.run(1)
    assert np.array_equal(s.counts(0), ref["cnt"]) and np.array_equal(s.trace(0), ref["trace"])
    s.close()
    print("first sweep bit-identical to the CPU oracle (%.0f s)" % (time.time() - t0), flush=True)
    del rp, ci, p
prob.close()
with G.options(sample_kernel=0):
    prob2 = G.Problem.synthetic(a.rows, T, 20.0, seed=1234, sort=True)
assert prob2.info.sample_kernel == 0 and prob2.info.index_bits == 64
s = G.Sampler(prob2, mu0, seed=5, gibbs_iter=n_it, trace_len=n_it); s.run(n_it)
assert dig(s.trace(0), s.counts(0)) == d1
print("CSR-tile kernel with 64-bit offsets: same bits", flush=True)
s.close(); prob2.close()
print("OK")
