"""Why do read shards of equal hits take different K1 times?  Cuts the config-3 problem into P equal-cost shards and prints, per shard,
the K1 time next to the facts that could explain it: tiles, distinct windows (lead bands), the share of the shard's reads the most
popular transcript / the top 16 receive in one sweep (same-address LDS atomics inside a tile, same-address global atomics at the flush)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from mmseq_amd import gibbs as G

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=50_000_000)
ap.add_argument("--parts", type=int, default=32)
ap.add_argument("--iters", type=int, default=64)
a = ap.parse_args()
prob = G.Problem.synthetic(a.rows, 200_000, 20.0, seed=1234, mapped_reads=a.rows)
mu0, _ = prob.start_values()
s = G.Sampler(prob, mu0, seed=3, gibbs_iter=1 << 20, trace_len=1, keep_trace=False)
s.run(400); s.sync(); s.close()
b = prob.shard_bounds(a.parts)
print("shard  rows(M)  tiles   bands  top1   top16  K1 ms   ms/Mtile")
for i in range(a.parts):
    sh = prob.shard(int(b[i]), int(b[i + 1]))
    rp, ci = sh.download()
    lead = np.minimum.reduceat(ci, rp[:-1].astype(np.int64)) >> 6
    bands = np.unique(lead).size
    sm = G.Sampler(sh, mu0, seed=7, gibbs_iter=1 << 20, trace_len=1, keep_trace=False, timing=1)
    sm.run(8); sm.reset_timing()
    sm.sample(); c = np.sort(sm.counts(0).astype(np.int64))[::-1]; sm.update()
    sm.reset_timing()
    sm.run(a.iters)
    t = sm.timing(); ms = t["sample_ms"] / t["sample_launches"]
    inf = sh.info
    print("%5d  %7.2f  %6d  %5d  %.3f  %.3f  %.4f  %.3f" % (i, inf.m / 1e6, inf.n_tiles, bands, c[0] / c.sum(), c[:16].sum() / c.sum(), ms, ms / inf.n_tiles * 1e6), flush=True)
    sm.close(); sh.close()
