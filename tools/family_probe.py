"""K1 per iteration and the EM sweep on the generator's gene-block mode over paralogue families of growing size (50 M reads x 200 k
transcripts x 20 hits, a fraction of the reads with one hit in another gene of the gene's family): what the gene-level reorder of
spec version 7 buys when a family fits a window of 255 transcripts and what is left when it does not.
usage: family_probe.py [far_fraction]        (one line per (gene_size, family) pair)"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from mmseq_amd import Problem, Sampler
far = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
for gene, fam in ((32, 0), (32, 3), (32, 8), (32, 32), (8, 16), (8, 64), (8, 256), (4, 1024)):
    prob = Problem.synthetic(50_000_000, 200_000, 20.0, seed=1234, sort=False, far_fraction=far if fam else 0.0, gene_size=gene, far_family=fam)
    # as the CLI uploads a hits file (bench.py side_measurement, genes=): rows in generator order, tx_order = gene << 32 | transcript
    rp, ci = prob.download()
    l = prob.l()
    prob.close()
    t_ids = np.arange(200_000, dtype=np.uint64)
    prob = Problem.from_csr(rp, ci, l, tx_order=((t_ids // np.uint64(gene)) << np.uint64(32)) | t_ids)
    del rp, ci
    mu0, _ = prob.start_values()
    s = Sampler(prob, mu0, n_chains=1, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
    s.run(100); s.sync(); s.reset_timing(); s.run(50); s.sync()
    tm, inf = s.timing(), prob.info
    print("genes of %3d, families of %4d genes (%5d transcripts), %2.0f %% of the reads with a hit in another gene of the family: K1 %.4f ms, %d of %d tiles with far lists, tx_renumbered %d"
          % (gene, fam, gene * fam, 100 * (far if fam else 0), tm["sample_ms"] / tm["sample_launches"], inf.far_tiles, inf.n_tiles, inf.tx_renumbered), flush=True)
    del s, prob
