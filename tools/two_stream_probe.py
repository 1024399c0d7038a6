"""8 chains as ONE sampler on one stream vs TWO samplers of 4 chains on two streams (K2 of one overlaps K1 of the other):
two_stream_probe.py [iterations]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmseq_amd import Problem, Sampler
from mmseq_amd import dist as mdist
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prob = Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234)
mu0, _ = prob.start_values()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0
one = Sampler(prob, mu0, seed=1, n_chains=8, gibbs_iter=4096, trace_len=1, keep_trace=False)
mdist.use_current_stream(one)
one.run(64)
t1 = timed(lambda: one.run(N))
print("one sampler, 8 chains:           %.4f ms per sweep of 8 chains, %.0f chain-it/s" % (t1 / N * 1e3, 8 * N / t1), flush=True)
one.close()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sa):
    a = Sampler(prob, mu0, seed=1, n_chains=4, chain_base=0, gibbs_iter=4096, trace_len=1, keep_trace=False); mdist.use_current_stream(a)
with torch.cuda.stream(sb):
    b = Sampler(prob, mu0, seed=1, n_chains=4, chain_base=4, gibbs_iter=4096, trace_len=1, keep_trace=False); mdist.use_current_stream(b)
def both(n):
    for _ in range(n // 8):      # interleave the enqueues so that neither stream runs ahead by much
        a.run(8); b.run(8)
both(64)
t2 = timed(lambda: both(N))
print("two samplers x 4 chains, 2 streams: %.4f ms per sweep of 8 chains, %.0f chain-it/s" % (t2 / N * 1e3, 8 * N / t2), flush=True)
