"""K2 (k_update: the Gamma redraw of src/mmseq.cpp:905-908) alone, per launch, at BASELINE configs[1] (T = 50 k) and configs[2] (T = 200 k):
    k2_probe.py        with MMSEQ_AMD_LIB=build_ab/lib_<variant>.so for A/B timing of a variant (profiles/r06_k2_ab.txt)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler
for R, T, A in ((5_000_000, 50_000, 8.0), (50_000_000, 200_000, 20.0)):
    prob = Problem.synthetic(R, T, A, seed=1234)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=True, timing=1)
    s.run(200); s.sync(); s.reset_timing()
    s.run(400); s.sync()
    tm = s.timing()
    print("T = %6d: K2 %.2f us per launch (K1 %.1f us)" % (T, 1e3 * tm["update_ms"] / tm["update_launches"], 1e3 * tm["sample_ms"] / tm["sample_launches"]), flush=True)
    s.close(); prob.close()
