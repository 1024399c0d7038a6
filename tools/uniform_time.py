"""A few sweeps of the no-locality fallback (hits uniform over all transcripts -> k_sample, CSR tiles) for a profiler: uniform_time.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmseq_amd import Problem, Sampler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
prob = Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234, uniform=True)
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(2); s.sync(); s.reset_timing(); s.run(N); s.sync()
tm = s.timing()
print("uniform hits: sample kernel %d, K1 %.3f ms" % (prob.info.sample_kernel, tm["sample_ms"] / tm["sample_launches"]))
