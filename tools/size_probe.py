"""K1 time per million rows as a function of the problem size (same rows-per-transcript density): a stream that fits the 256 MB
Infinity Cache against one that has to come from HBM.  usage: size_probe.py [chains]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # one HIP runtime per process: torch's first
from mmseq_amd import Problem, Sampler
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for R in (3_125_000, 6_250_000, 12_500_000, 25_000_000, 50_000_000):
    T = R // 250
    prob = Problem.synthetic(R, T, 20.0, seed=1234)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, n_chains=C, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(200); s.sync(); s.reset_timing()
    s.run(100); s.sync()
    tm = s.timing()
    k1 = tm["sample_ms"] / tm["sample_launches"]
    inf = prob.info
    print("R %9d T %6d chains %d stream %7.1f MB grid %6d  K1 %.4f ms  = %.3f us per M rows per chain  K2 %.4f" % (
        R, T, C, inf.stream_bytes / 1e6, inf.sample_grid, k1, k1 * 1e3 / (R / 1e6) / C, tm["update_ms"] / tm["update_launches"]), flush=True)
    s.close(); prob.close()
