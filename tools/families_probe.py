"""What the parts of the power-law family workload (tools/families.py) cost K1: config 3 in gene-block mode (32 isoforms per gene),
uploaded with the CLI's keys, (a) as generated, (b) + hub reads only, (c) + paralogue reads only, (d) both.   families_probe.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmseq_amd import Problem, Sampler
import families as fam
R = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
T, G = 200_000, 32
gen = Problem.synthetic(R, T, 20.0, seed=1234, sort=False, gene_size=G)
rp0, ci0 = gen.download(); l = gen.l(); gen.close()
for name, kw in (("near only", dict(paralogue=0.0, hub=0.0)), ("+ 1 % hub reads", dict(paralogue=0.0)), ("+ 17 % paralogue reads", dict(hub=0.0)), ("both", dict()),
                 ("both, chain distance 1 only", dict(max_distance=1))):
    rp, ci, tx, info = fam.power_law_families(rp0, ci0, T, G, seed=1234, **kw)
    prob = Problem.from_csr(rp, ci, l, tx_order=tx)
    del rp, ci
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, n_chains=1, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(150); s.sync(); s.reset_timing(); s.run(60); s.sync()
    tm, inf = s.timing(), prob.info
    print("%-30s K1 %.4f ms  tiles %d far %d (%.2f %%) stream %.3f GB slots/hit %.3f tx_renumbered %d  %s" % (
        name, tm["sample_ms"] / tm["sample_launches"], inf.n_tiles, inf.far_tiles, 100.0 * inf.far_tiles / inf.n_tiles, inf.stream_bytes / 1e9,
        inf.padded_slots / max(inf.nnz, 1), inf.tx_renumbered, {k: round(v, 4) if isinstance(v, float) else v for k, v in info.items()}), flush=True)
    s.close(); prob.close()
