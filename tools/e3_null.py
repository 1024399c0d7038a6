#!/usr/bin/env python3
"""The null distribution of SURVEY App. E.3's statistics: ONE engine against ITSELF (two seeds), many seed pairs, on the CPU.

App. E.3 asks of two engines: per observed transcript with iact < 20, |delta log_mu| <= 5 sqrt(mcse_a^2 + mcse_b^2) for >= 99 %,
none beyond 8 x, median sd ratio within 1 +- 0.01, pooled z with |mean| < 0.05 and variance in [0.8, 1.3].  Whether a bound can be held
depends on what two runs of the SAME sampler give: transcripts share reads, so their z scores are correlated and the pooled mean does
not shrink like 1 / sqrt(#transcripts); Sokal's adaptive window (src/sokal.cc:73-83) truncates the autocorrelation sum, so mcse is
biased low on slowly mixing transcripts.  This tool measures that, for the test shapes of tests/test_gpu_statistical.py:

    python tools/e3_null.py --pairs 20 --engine ref   [--rows 500000 --transcripts 5000 --avg 6 --iters 1024 --threads 8]
    python tools/e3_null.py --pairs 20 --engine keyed

Output: one line per seed pair (z mean, z var, share within 5, max |z|, median sd ratio, transcripts used) and the quantiles over the
pairs -- the table DESIGN.md section 6 cites next to every bound of the tests that deviates from App. E.3 as written."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=20)
ap.add_argument("--engine", choices=["ref", "keyed"], default="ref")
ap.add_argument("--rows", type=int, default=500_000)
ap.add_argument("--transcripts", type=int, default=5_000)
ap.add_argument("--avg", type=float, default=6.0)
ap.add_argument("--iters", type=int, default=1024)
ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) // 2))
ap.add_argument("--problem-seed", type=int, default=77)
ap.add_argument("--burn", type=int, default=0, help="samples dropped at the head of both chains (both start at the EM point)")
ap.add_argument("--far", type=float, default=0.0, help="fraction of the rows with a hit anywhere in the transcriptome")
ap.add_argument("--kmix", action="store_true", help="the multiplicities of test_real_file_shape_... (k = 1 .. 2000)")
ap.add_argument("--genes", type=int, default=0, help="also the gene level: sums of G consecutive isoforms per sample")
ap.add_argument("--pooled", type=int, default=0, help="side a = C chains of --iters pooled (burn dropped from each), side b = ONE chain of C x iters: "
                "the statistic of test_four_pooled_device_chains_... (mcse from the long chain alone)")
a = ap.parse_args()

B.lib()
q, _ = B.synth_problem(R=a.rows, T=a.transcripts, avg_hits=a.avg, seed=a.problem_seed, sort=False, far_fraction=a.far)
if a.kmix:
    rng = np.random.default_rng(5)
    k = rng.choice([1, 1, 1, 1, 1, 1, 2, 3, 9, 40, 64, 65, 300, 2000], size=q.m, p=[.14] * 6 + [.05, .04, .03, .02, .005, .005, .005, .005]).astype(np.uint32)
    q = B.Problem(q.row_ptr, q.col_idx, q.l * (k.sum() / a.rows), k=k)
mu0 = B.start_values(q)
mu_em = B.em(q, mu0)[0]
obs = np.unique(q.col_idx)
S = a.iters


def chain(seed, n_iter):
    if a.engine == "ref":
        return B.gibbs_ref(q, mu_em, seed=seed, n_iter=n_iter, trace_len=n_iter, threads=a.threads)["trace"]
    return B.gibbs_keyed(q, mu_em, seed=seed, n_iter=n_iter, trace_len=n_iter)["trace"]


def run(seed):
    tr = chain(seed, S)
    if a.genes:
        T = a.transcripts
        tr = np.stack([tr[g:min(g + a.genes, T)].sum(axis=0) for g in range(0, T, a.genes) if np.isin(np.arange(g, min(g + a.genes, T)), obs).any()])
    else:
        tr = tr[obs]
    with np.errstate(divide="ignore"):
        return np.log(tr[:, a.burn:])


def pooled_stats(seed):
    """C short chains (seeds seed .. seed + C - 1) against one chain of C x S (seed + C), as the pooled-chains test compares them."""
    C, Bn = a.pooled, a.burn
    n = C * (S - Bn)
    sl = np.zeros(len(obs)); sl2 = np.zeros(len(obs))
    with np.errstate(divide="ignore"):
        for c in range(C):
            lt = np.log(chain(seed + c, S)[obs][:, Bn:])
            sl += lt.sum(axis=1); sl2 += (lt ** 2).sum(axis=1)
        lb = np.log(chain(seed + C, C * S)[obs])
    mean_a = sl / n
    var_a = (sl2 - n * mean_a ** 2) / (n - 1)
    z, sdr = [], []
    for i in range(len(obs)):
        x = lb[i]
        if not np.isfinite(x).all() or not np.isfinite(mean_a[i]):
            continue
        rc, vb, tb, _ = B.sokal(x.copy())
        x = x[Bn:]
        if rc or not (tb < 20 and vb > 0 and var_a[i] > 0):
            continue
        z.append((mean_a[i] - x.mean()) / np.sqrt(tb * vb * (1.0 / n + 1.0 / x.size)))
        sdr.append(np.sqrt(var_a[i] / vb))
    z, sdr = np.array(z), np.array(sdr)
    return z.mean(), z.var(), (np.abs(z) <= 5).mean(), np.abs(z).max(), np.median(sdr), len(z)


def stats(la, lb):
    n = la.shape[1]
    n2 = 1 << int(np.log2(n))
    z, sdr, sdtol = [], [], []
    for i in range(la.shape[0]):
        xa, xb = la[i], lb[i]
        if not (np.isfinite(xa).all() and np.isfinite(xb).all()):
            continue
        ra, va, ta, _ = B.sokal(xa[:n2].copy())
        rb, vb, tb, _ = B.sokal(xb[:n2].copy())
        if ra or rb or not (ta < 20 and tb < 20 and va > 0 and vb > 0):
            continue
        z.append((xa.mean() - xb.mean()) / np.sqrt(ta * va / n + tb * vb / n))
        sdr.append(np.sqrt(va / vb))
        sdtol.append(5 * np.sqrt(max(ta, tb) / (2.0 * n)))
    z, sdr, sdtol = np.array(z), np.array(sdr), np.array(sdtol)
    global last_sd_within
    last_sd_within = ((np.abs(sdr - 1) <= sdtol).mean(), (np.abs(sdr - 1) <= 2 * sdtol).mean())
    return z.mean(), z.var(), (np.abs(z) <= 5).mean(), np.abs(z).max(), np.median(sdr), len(z)


rows = []
last_sd_within = (float('nan'), float('nan'))
print("# engine %s, %d rows x %d transcripts, avg %.0f hits, %d iterations, burn %d, %d observed transcripts, far %.2f, kmix %d, genes %d, pooled %d" % (
    a.engine, a.rows, a.transcripts, a.avg, S, a.burn, len(obs), a.far, a.kmix, a.genes, a.pooled), flush=True)
print("# pair  z_mean   z_var  within5  max|z|  median_sd_ratio  used", flush=True)
for p in range(a.pairs):
    r = pooled_stats(10_000 + 16 * p) if a.pooled else stats(run(10_000 + 2 * p), run(10_001 + 2 * p))
    rows.append(r)
    print("%5d  %+.4f  %.4f  %.4f  %6.2f  %.5f  %d" % ((p,) + r) + ("" if a.pooled else "   sd ratio within 1 +- 5 sqrt(tau / 2n): %.4f, within twice that: %.4f" % last_sd_within), flush=True)
R = np.array(rows)
for name, col in (("z_mean", 0), ("z_var", 1), ("within5", 2), ("max|z|", 3), ("median_sd_ratio", 4)):
    x = R[:, col]
    print("# %-16s min %+.4f  q10 %+.4f  median %+.4f  q90 %+.4f  max %+.4f   |.|max %.4f" % (
        name, x.min(), np.quantile(x, 0.1), np.median(x), np.quantile(x, 0.9), x.max(), np.abs(x).max()), flush=True)
