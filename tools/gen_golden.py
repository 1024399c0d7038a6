"""Generates tests/golden/*.json.  Run in the BUILD container (needs /root/reference for the
compiled reference sokal.cc in oracle/_ref; `make -C oracle` builds it).

  sokal_reference.json   inputs -> (rc, var, tau, m) produced by the REFERENCE's own sokal()
                         (src/sokal.cc, compiled unmodified)
  keyed_chain_tiny.json  a tiny problem and the keyed-stream Gibbs trace/counts it must produce
  keyed_chain_k_draws.json  rows on either side of the draws / binomial-chain boundary of spec version 5
  em_fixed_tiny.json     the same problem through the exact-sum EM: mu and log-likelihood after 0/1/2/8 sweeps
                         (produced by the CPU oracle; guards oracle and kernels against co-drift)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as B  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def hexf(a):
    return [float(x).hex() for x in np.asarray(a, np.float64).ravel()]


def lcg_ar1(n, rho, seed):
    # deterministic AR(1) driven by a 64-bit LCG + Box-Muller (no numpy RNG dependence)
    state = seed & 0xFFFFFFFFFFFFFFFF
    out = np.empty(n)
    x = 0.0
    for i in range(n):
        us = []
        for _ in range(2):
            state = (state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            us.append(((state >> 11) + 0.5) / 2.0 ** 53)
        z = np.sqrt(-2.0 * np.log(us[0])) * np.cos(2 * np.pi * us[1])
        x = rho * x + z
        out[i] = x
    return out


def gen_sokal():
    cases = []
    specs = [("ar1_rho0.5_n1024", lcg_ar1(1024, 0.5, 1)), ("ar1_rho0.9_n1024", lcg_ar1(1024, 0.9, 2)),
             ("ar1_rho0.0_n1024", lcg_ar1(1024, 0.0, 3)), ("ar1_rho0.99_n1024", lcg_ar1(1024, 0.99, 4)),
             ("ar1_rho0.5_n64", lcg_ar1(64, 0.5, 5)), ("ar1_rho0.7_n4096", lcg_ar1(4096, 0.7, 6)),
             ("constant_n1024", np.full(1024, 3.25)), ("ramp_n16", np.arange(16.0)),
             ("n4", np.array([1.0, -2.0, 0.5, 4.0])), ("bad_n1000", lcg_ar1(1000, 0.5, 7)),
             ("bad_n2", np.array([1.0, 2.0])), ("log_gamma_like", np.log(np.abs(lcg_ar1(1024, 0.3, 8)) + 1e-3))]
    for name, x in specs:
        r = B.sokal_ref(x)
        if r is None:
            raise SystemExit("oracle/_ref/libsokal_ref.so missing: run `make -C oracle` where /root/reference exists")
        rc, var, tau, m = r
        cases.append(dict(name=name, x=hexf(x), rc=rc, var=float(var).hex(), tau=float(tau).hex(), m=m))
    json.dump(dict(source="reference src/sokal.cc compiled unmodified (oracle/Makefile target ref)", cases=cases),
              open(os.path.join(OUT, "sokal_reference.json"), "w"), indent=0)


def gen_tiny_chain():
    rows = [[0, 1], [1, 2, 3], [0], [2, 3], [4, 5, 6, 7], [1, 7], [5], [3, 4, 5], [0, 7], [6, 7], [2], [1, 2, 3, 4, 5, 6]]
    k = [3, 1, 12, 5, 40, 2, 7, 9, 1, 300, 4, 1000]
    l = [0.5, 1.5, 0.25, 2.0, 1.0, 0.75, 3.0, 0.1]
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    p = B.Problem(rp, ci, np.asarray(l), k=np.asarray(k, np.uint32))
    mu0, uh = B.start_values(p)
    r = B.gibbs_keyed(p, mu0, alpha=0.1, beta=0.1, seed=1234, chain=0, n_iter=32, trace_len=16)
    json.dump(dict(source="oracle/mmseq_oracle.c orc_gibbs_keyed (keyed Philox streams), seed 1234, chain 0, "
                          "alpha=beta=0.1, 32 iterations, 16 kept",
                   row_ptr=[int(v) for v in rp], col_idx=[int(v) for v in ci], k=k, l=hexf(l), mu0=hexf(mu0),
                   unique_hits=[int(v) for v in uh], trace=hexf(r["trace"]), cnt_last=[int(v) for v in r["cnt"]],
                   mu_last=hexf(r["mu"])),
              open(os.path.join(OUT, "keyed_chain_tiny.json"), "w"), indent=0)


def gen_k_draws_chain():
    """Spec version 8: a row draws its k categoricals one by one while k <= min(K_SMALL, K_DRAWS_PER_HIT * (hits - 1)) = min(64, 16 (hits - 1));
    above, the conditional-binomial chain.  One row on either side of the boundary for five row lengths (2, 3, 4 hits: the per-hit limit;
    5 and 8 hits: K_SMALL), and a row of one hit.  A fixture of its own: keyed_chain_tiny.json has no row near the boundary and stays
    byte for byte what it was (spec versions 5 and 8 agree on its rows)."""
    rows = [[1, 7], [3, 5], [5, 6, 7], [0, 2, 4], [1, 2, 3, 4], [0, 5, 6, 7], [0, 2, 4, 5, 6], [1, 2, 3, 4, 5],
            [0, 1, 2, 3, 4, 5, 6, 7], [0, 1, 2, 3, 4, 5, 6, 7], [6]]
    # draws: 16 on 2 hits, 32 on 3, 48 on 4, 64 on 5 and on 8; binomial chain: 17 on 2, 33 on 3, 49 on 4, 65 on 5 and on 8
    k = [16, 17, 32, 33, 48, 49, 64, 65, 64, 65, 500]
    l = [0.5, 1.5, 0.25, 2.0, 1.0, 0.75, 3.0, 0.1]
    rp = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    ci = np.concatenate([np.asarray(r, np.uint32) for r in rows])
    p = B.Problem(rp, ci, np.asarray(l), k=np.asarray(k, np.uint32))
    mu0, uh = B.start_values(p)
    r = B.gibbs_keyed(p, mu0, alpha=0.1, beta=0.1, seed=4321, chain=0, n_iter=32, trace_len=16)
    json.dump(dict(source="oracle/mmseq_oracle.c orc_gibbs_keyed (keyed Philox streams, spec version 8), seed 4321, chain 0, "
                          "alpha=beta=0.1, 32 iterations, 16 kept",
                   row_ptr=[int(v) for v in rp], col_idx=[int(v) for v in ci], k=k, l=hexf(l), mu0=hexf(mu0),
                   unique_hits=[int(v) for v in uh], trace=hexf(r["trace"]), cnt_last=[int(v) for v in r["cnt"]],
                   mu_last=hexf(r["mu"])),
              open(os.path.join(OUT, "keyed_chain_k_draws.json"), "w"), indent=0)


def gen_tiny_em():
    """The tiny problem of keyed_chain_tiny.json through the exact-sum EM (orc_em): log-likelihoods and mu after 1, 2, 8
    sweeps, plus a start with one dead and one wildly scaled transcript (takes the measured-exponent repeat path)."""
    g = json.load(open(os.path.join(OUT, "keyed_chain_tiny.json")))
    p = B.Problem(np.asarray(g["row_ptr"], np.uint64), np.asarray(g["col_idx"], np.uint32),
                  np.array([float.fromhex(x) for x in g["l"]]), k=np.asarray(g["k"], np.uint32))
    mu0 = np.array([float.fromhex(x) for x in g["mu0"]])
    out = {"source": "oracle/mmseq_oracle.c orc_em (exact fixed-point sums) on the problem of keyed_chain_tiny.json", "runs": []}
    wild = mu0.copy()
    wild[2] = 0.0
    wild[5] *= 1e-120
    for name, start in (("start_values", mu0), ("dead_and_wild", wild)):
        for sweeps in (0, 1, 2, 8):
            mu, it, ll, redo = B.em_x(p, start, max_iter=sweeps, epsilon=-1e308)
            out["runs"].append(dict(start=name, mu_start=hexf(start), sweeps=sweeps, mu=hexf(mu), loglik=float(ll).hex(), repeated_passes=int(redo)))
    json.dump(out, open(os.path.join(OUT, "em_fixed_tiny.json"), "w"), indent=0)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_sokal()
    gen_tiny_chain()
    gen_k_draws_chain()
    gen_tiny_em()
    print("wrote", os.listdir(OUT))
