"""Phase statistics of k_sample_bigk (bigk_kernels.h) from a diagnostics build of the library:
    python tools/build_variant.py bkstats 'Makefile::-Wno-unused-value::-Wno-unused-value -DMMG_BIGK_STATS'
    MMSEQ_AMD_LIB=build_ab/lib_bkstats.so python tools/bigk_stats.py [rows transcripts avg_hits k | heavy | collapsed]
prints, per phase, the runs, the mean lanes served per run and the runs per row-step."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler, _lib
import bench
lib = _lib.load()
what = sys.argv[1] if len(sys.argv) > 1 else "bigk"
if what in ("heavy", "collapsed"):
    side = {s[0]: s[2] for s in bench.SIDE}[what]
    p0 = Problem.synthetic(side["rows"], side["transcripts"], side["avg_hits"], seed=1234, mapped_reads=side["rows"])
    rp, ci = p0.download(); l = p0.l(); p0.close()
    rng = np.random.default_rng(1234); u = rng.random(side["rows"])
    if what == "collapsed":
        k = np.minimum(1e6, np.floor((1.0 - u) ** (-1.0 / 0.92))).astype(np.uint32)
    else:
        k = np.ones(side["rows"], np.uint32)
        for lo_u, hi_u, lo_k, hi_k in ((0.5, 0.8, 2, 8), (0.8, 0.95, 9, 64), (0.95, 0.99, 65, 300), (0.99, 1.0, 300, 20000)):
            sel = (u >= lo_u) & (u < hi_u)
            k[sel] = np.exp(rng.uniform(np.log(lo_k), np.log(hi_k + 1), size=int(sel.sum()))).astype(np.uint32).clip(lo_k, hi_k)
else:
    R, T, A, K = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2_000_000, 200_000, 20.0, 1000)
    p0 = Problem.synthetic(R, T, A, seed=1234)
    rp, ci = p0.download(); l = p0.l(); p0.close()
    k = np.full(R, K, np.uint32)
prob = Problem.from_csr(rp, ci, l, k=k)
rp2, ci2, k2 = prob.download(with_k=True)
L = np.diff(rp2).astype(np.int64)
big = (L >= 2) & (k2.astype(np.int64) > np.minimum(64, 16 * (L - 1)))      # mmg_types.h: bigk_row (spec version 8)
steps = int((L[big] - 1).sum())
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, n_chains=1, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(20); s.sync()
out = (ctypes.c_ulonglong * 16)()
lib.mmg_selftest_bigk_stats(out)
N = 20
s.run(N); s.sync()
lib.mmg_selftest_bigk_stats(out)
tm = s.timing()
print("%s: %d rows on the list, %d binomial steps per sweep, K1 %.4f ms" % (what, int(big.sum()), steps, tm["sample_ms"] / tm["sample_launches"]))
for q, name in enumerate(("FETCH", "STEP", "BTRS", "SLOW", "IFULL", "SLOW64", "IFULL64")):
    runs, lanes = out[q] / N, out[8 + q] / N
    print("%-6s runs %10.0f  lanes/run %5.1f  lane-runs per step %.3f" % (name, runs, lanes / max(runs, 1), lanes / max(steps, 1)))
