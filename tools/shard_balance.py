"""Per-shard K1 time of the read-sharded chain at 8 shards of BASELINE config 3 (50 M reads x 200 k transcripts), all shards on ONE
MI355X (mmg_selftest_gibbs_shards; HIP events around every shard's sample kernels), for the library's cost-balanced cut
(mmg_problem_shard_bounds) and for the equal-hit cut (mmg_shard_bounds: what round 3 shipped, and what the reference's
schedule(static) over rows amounts to, src/mmseq.cpp:864).  The sharded chain advances at the pace of its slowest shard: max / mean
of the shards' K1 times is what read-shard scaling loses before any collective is involved.

usage: shard_balance.py [--rows N] [--parts P] [--iters I] [--json out.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime in the process: torch's, loaded first)
from mmseq_amd import gibbs as G


def file_like_k(rows, seed):
    """the multiplicities of a collapsed 50 M-read file of the generator (tools/collapse_probe.py; bench.py side_measurement)"""
    rng = np.random.default_rng(seed)
    u = rng.random(rows)
    k = np.ones(rows, np.uint32)
    for thr, val in ((0.064, 2), (0.011, 3), (0.0035, 4), (0.002, 6)):
        k[u < thr] = val
    big = u < 0.0012
    k[big] = rng.integers(9, 37, size=int(big.sum())).astype(np.uint32)
    return k


def measure(prob, bounds, mu0, iters, warm):
    parts = len(bounds) - 1
    shards = [prob.shard(int(bounds[i]), int(bounds[i + 1])) for i in range(parts)]
    smps = [G.Sampler(sh, mu0, seed=7, gibbs_iter=1 << 20, trace_len=1, keep_trace=False, timing=1) for sh in shards]
    G.gibbs_shards_selftest(smps, warm)
    for s in smps:
        s.reset_timing()
    G.gibbs_shards_selftest(smps, iters)
    ms = []
    for s in smps:
        t = s.timing()
        ms.append(t["sample_ms"] / max(t["sample_launches"], 1))
    hits = [sh.info.nnz for sh in shards]
    far = [sh.info.far_tiles / max(sh.info.n_tiles, 1) for sh in shards]
    kern = [sh.info.sample_kernel for sh in shards]
    for s in smps:
        s.close()
    for sh in shards:
        sh.close()
    return ms, hits, far, kern


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50_000_000)
    ap.add_argument("--transcripts", type=int, default=200_000)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--iters", type=int, default=48)
    ap.add_argument("--json", default=None)
    ap.add_argument("--workloads", default="near,far20,file", help="comma-separated: near, far20, file")
    a = ap.parse_args()
    out = []
    for wid, name, kw, mult in (("near", "near rows only (the headline workload)", dict(), False),
                                ("far20", "20 % of the rows with a hit anywhere in the transcriptome", dict(far_fraction=0.2), False),
                                ("file", "multiplicities of a collapsed file + 2 % far rows", dict(far_fraction=0.02), True)):
        if wid not in a.workloads.split(","):
            continue
        t0 = time.time()
        prob = G.Problem.synthetic(a.rows, a.transcripts, 20.0, seed=1234, mapped_reads=a.rows, **kw)
        if mult:
            rp, ci = prob.download()
            l = prob.l()
            prob.close()
            prob = G.Problem.from_csr(rp, ci, l, k=file_like_k(a.rows, 1234))
        rp = prob.download()[0]
        mu0, _ = prob.start_values()
        inf = prob.info
        print("== %s: %d stored rows, %d hits, %d tiles (%d far), built in %.1f s" % (name, inf.m, inf.nnz, inf.n_tiles, inf.far_tiles, time.time() - t0), flush=True)
        # clocks up first
        s = G.Sampler(prob, mu0, seed=3, gibbs_iter=1 << 20, trace_len=1, keep_trace=False)
        s.run(400); s.sync(); s.close()
        rec = {"workload": name, "stored_rows": inf.m, "hits": inf.nnz, "parts": a.parts, "cuts": {}}
        t0 = time.time()
        bt = prob.shard_bounds_timed(mu0, a.parts)
        print("  (the timed cut took %.3f s)" % (time.time() - t0))
        for cut, b in (("measured cost (mmg_problem_shard_bounds_timed)", bt), ("modelled cost (mmg_problem_shard_bounds)", prob.shard_bounds(a.parts)),
                       ("equal hits (mmg_shard_bounds)", G.shard_bounds(rp, a.parts))):
            ms, hits, far, kern = measure(prob, b, mu0, a.iters, 8)
            r = max(ms) / (sum(ms) / len(ms))
            print("  %-46s K1 ms per shard: %s" % (cut, " ".join("%.4f" % x for x in ms)))
            print("  %-46s hits per shard (M): %s   far-tile share: %s   kernels %s" % ("", " ".join("%.1f" % (h / 1e6) for h in hits), " ".join("%.2f" % f for f in far), kern))
            print("  %-46s max / mean = %.3f   (slowest shard %.4f ms; sum %.4f ms)" % ("", r, max(ms), sum(ms)), flush=True)
            rec["cuts"][cut] = {"k1_ms_per_shard": ms, "hits_per_shard": hits, "far_tile_share": far, "max_over_mean": r}
        out.append(rec)
        prob.close()
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
