"""What overlapping K2 and the launch ramps with other K1 work could buy: S independent single-chain samplers on S streams against one
(aggregate chain-iterations/s; every sampler streams the whole problem).   overlap_probe.py [iterations]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmseq_amd import Problem, Sampler
from mmseq_amd import dist as mdist
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234)
mu0, _ = prob.start_values()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    smp = []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            s = Sampler(prob, mu0, seed=1, n_chains=1, chain_base=i, gibbs_iter=8192, trace_len=1, keep_trace=False)
            mdist.use_current_stream(s)
            smp.append(s)
    def go(n):
        for _ in range(n // 16):
            for s in smp: s.run(16)
    go(64)
    t = timed(lambda: go(N))
    print("%d sampler(s) on %d stream(s): %.4f ms per chain-iteration, %.0f chain-it/s" % (S, S, t / N / S * 1e3, S * N / t), flush=True)
    for s in smp: s.close()
