"""mmg_math.h: btrs_pretest (the fp32 estimate k_sample_bigk tries before the four-logarithm acceptance test of BTRS) against the fp64 test
on 2.4 10^9 attempts over the whole range of (n, p): how many reach the test, how many the estimate decides, how many it decides WRONGLY (none
may), and how close its actual error comes to the bound it allows itself.   btrs_pretest_check.py [seeds per range: 4]"""
import sys, time
sys.path.insert(0, ".")
from mmseq_amd import gibbs as G
tot = [0, 0, 0, 0, 0]
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for lo, hi in ((21, 100), (100, 2000), (2000, 1e5), (1e5, 1e7), (1e7, 4.29e9), (21, 4.29e9)):
    t = time.time()
    c = [0, 0, 0, 0, 0]
    for seed in range(1, seeds + 1):
        r = G.selftest_btrs_pretest(seed * 7919, 100_000_000, lo, hi)
        c = [a + b for a, b in zip(c[:4], r[:4])] + [max(c[4], r[4])]
    print("n in [%g, %g]: %d reach the exact test, %.3f %% decided, %d WRONG, accept rate %.3f, largest error %.3f of its bound  (%.1f s)" % (lo, hi, c[0], 100.0 * c[1] / max(c[0], 1), c[2], c[3] / max(c[0], 1), c[4] / 1e6, time.time() - t), flush=True)
    tot = [a + b for a, b in zip(tot[:4], c[:4])] + [max(tot[4], c[4])]
print("total: %d cases at the exact test, %.3f %% decided, %d wrong, largest error %.3f of its bound" % (tot[0], 100.0 * tot[1] / tot[0], tot[2], tot[4] / 1e6))
