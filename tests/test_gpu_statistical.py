"""SURVEY App. E.3 on the device: the product chain (keyed Philox streams, categorical draws, canonical row order) against the
REFERENCE-STRUCTURED engine of the oracle (one MT19937 per thread seeded seed + tid, count slabs, multinomial by conditional
binomials, Marsaglia-Tsang Gamma: the structure of src/mmseq.cpp:834-918) on the same problem and start value.  The two share
no random numbers, no row order and no sampling algorithm, only the model -- so agreement of the posterior summaries within
Monte Carlo error is evidence that does not depend on the builder's own keyed spec.  Summaries as the reference computes them:
mean of the logged trace (src/mmseq.cpp:1195-1227), Sokal's variance and autocorrelation time (:1311-1324)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed_dev,seed_ref", [(11, 1001), (12, 1002), (13, 1003)])
def test_device_chain_agrees_with_reference_structured_engine(gpu, orc, seed_dev, seed_ref):
    R, T, S = 500_000, 5_000, 1024
    q, _ = orc.synth_problem(R=R, T=T, avg_hits=6, seed=77, sort=False)         # generator order, as a reader would deliver it
    prob = gpu.Problem.from_csr(q.row_ptr, q.col_idx, q.l)
    mu0, _ = prob.start_values()
    mu_em, it, _ = prob.em(mu0)                                                   # both chains start at the EM optimum (src/mmseq.cpp:820)
    s = gpu.Sampler(prob, mu_em, seed=seed_dev, gibbs_iter=S, trace_len=S)
    s.run(S)
    summ = gpu.Summary(s, chain=0)
    dev = summ.series(gpu.SERIES_TRANSCRIPT)
    assert (dev["rc"] == 0).all()
    threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    ref = orc.gibbs_ref(q, mu_em, seed=seed_ref, n_iter=S, trace_len=S, threads=threads)["trace"]
    with np.errstate(divide="ignore"):
        lref = np.log(ref)
    obs = np.unique(q.col_idx)
    z, sdr = [], []
    for t in obs:
        rc, var_b, tau_b, _ = orc.sokal(lref[t].copy())
        var_a, tau_a = dev["var"][t], dev["tau"][t]
        if rc or not (tau_a < 20 and tau_b < 20 and var_a > 0 and var_b > 0):
            continue
        mc = np.sqrt(tau_a * var_a / S + tau_b * var_b / S)                       # mcse of either mean, :1320-1323
        z.append((dev["log_mean"][t] - lref[t].mean()) / mc)
        sdr.append(np.sqrt(var_a / var_b))
    z, sdr = np.array(z), np.array(sdr)
    assert len(z) > 0.9 * len(obs)
    assert (np.abs(z) <= 5).mean() >= 0.99                                        # App. E.3: |delta log_mu| <= 5 sqrt(mcse_a^2 + mcse_b^2)
    assert abs(np.median(sdr) - 1) < 0.01                                         # posterior sd: same to 1 %
    assert abs(z.mean()) < 0.1 and 0.8 < z.var() < 1.3                            # differences are Monte Carlo noise of the stated size
    summ.close(); s.close(); prob.close()
