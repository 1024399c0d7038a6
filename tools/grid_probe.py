"""K1 time against the number of persistent workgroups (MMG_OPT_SELL_WAVES_PER_CU): how much of a launch is ramp and tail.
usage: grid_probe.py [rows transcripts avg]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmseq_amd import Problem, Sampler, gibbs
R, T, A = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (5_000_000, 50_000, 8.0)
for w in (28, 20, 14, 8, 4):
    with gibbs.options(sell_waves_per_cu=w):
        prob = Problem.synthetic(R, T, A, seed=1234)
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(300); s.sync(); s.reset_timing()
    s.run(300); s.sync()
    tm = s.timing()
    print("waves per CU %2d (grid %5d): K1 %.4f ms  K2 %.4f ms" % (w, prob.info.sample_grid, tm["sample_ms"] / tm["sample_launches"], tm["update_ms"] / tm["update_launches"]), flush=True)
    del s, prob
