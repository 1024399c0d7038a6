"""Which measurement condition changes K1's HIP-event time?  (stream, event cadence, trace)"""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmseq_amd import Problem, Sampler
prob = Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234)
mu0, _ = prob.start_values()
for legacy, timing, trace in [(0, 1, 0), (1, 1, 0), (0, 4, 0), (0, 1, 1), (1, 4, 1), (0, 1, 0)]:
    s = Sampler(prob, mu0, n_chains=1, gibbs_iter=1024, trace_len=1024, keep_trace=bool(trace), timing=timing)
    if legacy:
        s.set_stream(1)
    s.run(300); s.sync(); s.reset_timing()
    s.run(200); s.sync()
    tm = s.timing()
    print("legacy_stream=%d timing_every=%d keep_trace=%d: K1 %.4f ms over %d launches, K2 %.4f" % (
        legacy, timing, trace, tm["sample_ms"] / tm["sample_launches"], tm["sample_launches"], tm["update_ms"] / tm["update_launches"]), flush=True)
    s.close()
