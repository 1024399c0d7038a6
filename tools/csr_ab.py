import os, subprocess, sys
CHILD = r'''
import sys
sys.path.insert(0, "/root/repo")
from mmseq_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from mmseq_amd import Problem, Sampler
for kw in (dict(uniform=True), dict(sort=False)):
    prob = Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234, **kw)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(6); s.sync(); s.reset_timing()
    s.run(6); s.sync()
    tm = s.timing()
    print("%s %s kernel %d grid %d: K1 %.3f ms" % (sys.argv[1].split("/")[-1], kw, prob.info.sample_kernel, prob.info.sample_grid, tm["sample_ms"] / tm["sample_launches"]), flush=True)
    del s, prob
'''
for l in sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l)], capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-300:], flush=True)
