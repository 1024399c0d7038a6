"""The fp32 tests of k_sample_bigk (mmg_math.h: btrs_pretest, binv_pretest) in place, at full size: the chains of the bench's `heavy` and
`collapsed` workloads and of 2 M rows with k = 1000 must be the same bits with the tests and without them (a -DBK_NO_FP32_TESTS build runs
the fp64 code only).  Prints one line per workload with digests of the counts and of the trace after ITERS iterations; run it with both
libraries and compare the lines:
    python tools/build_variant.py nofp32 'Makefile::-Wno-unused-value::-Wno-unused-value -DBK_NO_FP32_TESTS'
    python tools/pretest_fullsize_check.py > a.txt;  MMSEQ_AMD_LIB=build_ab/lib_nofp32.so python tools/pretest_fullsize_check.py > b.txt;  cmp a.txt b.txt"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler
import bench
ITERS = 24
side = {s[0]: s[2] for s in bench.SIDE}
for what in ("heavy", "collapsed", "bigk"):
    if what == "bigk":
        R, T, A = 2_000_000, 200_000, 20.0
        p0 = Problem.synthetic(R, T, A, seed=1234)
        rp, ci = p0.download(); l = p0.l(); p0.close()
        k = np.full(R, 1000, np.uint32)
    else:
        cfg = side[what]
        p0 = Problem.synthetic(cfg["rows"], cfg["transcripts"], cfg["avg_hits"], seed=1234, mapped_reads=cfg["rows"])
        rp, ci = p0.download(); l = p0.l(); p0.close()
        rng = np.random.default_rng(1234); u = rng.random(cfg["rows"])
        if what == "collapsed":
            k = np.minimum(1e6, np.floor((1.0 - u) ** (-1.0 / 0.92))).astype(np.uint32)
        else:
            k = np.ones(cfg["rows"], np.uint32)
            for lo_u, hi_u, lo_k, hi_k in ((0.5, 0.8, 2, 8), (0.8, 0.95, 9, 64), (0.95, 0.99, 65, 300), (0.99, 1.0, 300, 20000)):
                sel = (u >= lo_u) & (u < hi_u)
                k[sel] = np.exp(rng.uniform(np.log(lo_k), np.log(hi_k + 1), size=int(sel.sum()))).astype(np.uint32).clip(lo_k, hi_k)
    prob = Problem.from_csr(rp, ci, l, k=k)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, seed=99, n_chains=2, gibbs_iter=ITERS, trace_len=ITERS)
    s.run(ITERS); s.sync()
    out = []
    for c in range(2):
        cnt, tr = s.counts(c), s.trace(c)
        out.append("chain %d counts %s (sum %d) trace %s" % (c, hashlib.sha256(np.ascontiguousarray(cnt).tobytes()).hexdigest()[:16], int(cnt.astype(np.int64).sum()),
                                                             hashlib.sha256(np.ascontiguousarray(tr).tobytes()).hexdigest()[:16]))
    inf = prob.info
    print("%-10s %d stored rows, %d reads: %s" % (what, inf.m, inf.total_k, "; ".join(out)), flush=True)
    s.close(); prob.close()
