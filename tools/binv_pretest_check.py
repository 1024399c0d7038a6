"""mmg_math.h: binv_pretest (the fp32 search k_sample_bigk tries before the fp64 inversion) against the fp64 search on 3 10^9 cases over the
whole range of (n, n p), at the bound the sampler uses (slack 1) and at bounds scaled down until decided cases start to differ: how many
the fp32 search decides, how many it decides WRONGLY (none may at slack 1).   binv_pretest_check.py [more rounds at slack 1: 0]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmseq_amd import gibbs
N = 250_000_000
for slack in (1.0, 0.25, 1.0 / 16, 1.0 / 64, 1.0 / 256):
    tot = [0, 0, 0, 0, 0]
    for lo, hi in ((1, 20), (20, 2000), (2000, 1e6), (1e6, 1e8)):   # (above: a search that falls off the end walks n terms -- minutes)
        t = time.time()
        c = gibbs.selftest_binv_pretest(seed=int(1000 * slack) + int(lo), n_cases=N if slack == 1.0 else N // 2, n_lo=lo, n_hi=hi, slack=slack)
        if slack == 1.0:
            print("slack 1: n in [%g, %g]: %d cases, %.4f %% decided, %d WRONG, %d ran off the end, mean x %.3f  (%.1f s)"
                  % (lo, hi, c[0], 100.0 * c[1] / max(c[0], 1), c[2], c[3], c[4] / max(c[0], 1), time.time() - t), flush=True)
        tot = [a + b for a, b in zip(tot, c)]
    print("slack %.4f total: %d cases, %.4f %% decided, %d wrong" % (slack, tot[0], 100.0 * tot[1] / tot[0], tot[2]), flush=True)
more = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if more:
    tot = [0, 0, 0, 0, 0]
    for rep in range(1, more + 1):
        for lo, hi in ((1, 20), (20, 2000), (2000, 1e6), (1e6, 1e8)):
            c = gibbs.selftest_binv_pretest(seed=7919 * rep + int(lo), n_cases=N, n_lo=lo, n_hi=hi, slack=1.0)
            tot = [a + b for a, b in zip(tot, c)]
    print("slack 1, %d more rounds: %d cases, %.4f %% decided, %d wrong, %d ran off the end" % (more, tot[0], 100.0 * tot[1] / tot[0], tot[2], tot[3]), flush=True)
