"""The multiplicity path of K1 (src/mmseq.cpp:880 on collapsed hit sets, :409-440): the bench's `heavy` and `collapsed` side measurements
and rows that ALL carry k = 1000, one line each: ms per step, K1 (both launches), K2.   kmult_probe.py [heavy,collapsed,bigk]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mmseq_amd import gibbs as _g
for name in ("bigk_per_wave", "bigk_side_stream"):   # e.g. BIGK_SIDE_STREAM=0 BIGK_PER_WAVE=128
    if os.environ.get(name.upper()):
        _g.selftest_option(_g.OPT[name], int(os.environ[name.upper()]))
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["heavy", "collapsed", "bigk"]
side = {s[0]: s[2] for s in bench.SIDE}
for w in which:
    if w in side:
        r = bench.side_measurement(w, **side[w])
        print("%-10s step %.4f ms  K1 %.4f ms  K2 %.4f ms   rows %d total_k %d hits %d tiles %d" % (w, r["ms_per_step"], r["k1_ms_all_chains"], r["k2_ms"], r["reads"], r["total_k"], r["hits"], r["n_tiles"]), flush=True)
    elif w == "bigk":
        from mmseq_amd import Problem, Sampler
        R, T, A, K = 2_000_000, 200_000, 20.0, 1000
        p0 = Problem.synthetic(R, T, A, seed=1234)
        rp, ci = p0.download(); l = p0.l(); p0.close()
        prob = Problem.from_csr(rp, ci, l, k=np.full(R, K, np.uint32))
        mu0, _ = prob.start_values()
        s = Sampler(prob, mu0, n_chains=1, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
        s.run(60); s.sync(); s.reset_timing(); s.run(40); s.sync()
        tm = s.timing(); inf = prob.info
        assert int(s.counts(0).astype(np.int64).sum()) == inf.total_k
        print("%-10s K1 %.4f ms  K2 %.4f ms   rows %d hits %d tiles %d (k = %d on every row)" % (w, tm["sample_ms"] / tm["sample_launches"], tm["update_ms"] / tm["update_launches"], inf.m, inf.nnz, inf.n_tiles, K), flush=True)
        s.close(); prob.close()
