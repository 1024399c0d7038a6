"""K1 on hits uniform over all transcripts (SURVEY App. D worst case): the CSR-tile kernel against the sliced-ELL kernel with far tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmseq_amd import Problem, Sampler, gibbs

R, T, A = 50_000_000, 200_000, 20.0
for kern in (0, 2):
    with gibbs.options(sample_kernel=kern):
        prob = Problem.synthetic(R, T, A, seed=1234, uniform=True)
    inf = prob.info
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(10); s.sync(); s.reset_timing()
    s.run(10); s.sync()
    tm = s.timing()
    print("kernel option %d -> sample_kernel %d: K1 %.3f ms, tiles %d fast %d far %d, stream %.2f GB" % (
        kern, inf.sample_kernel, tm["sample_ms"] / tm["sample_launches"], inf.n_tiles, inf.fast_tiles, inf.far_tiles, inf.stream_bytes / 1e9), flush=True)
    del s, prob
