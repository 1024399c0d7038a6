"""End-to-end scale check of the mmseq CLI on a synthetic hits file (GPU box).  Prints stage timings."""
import os, sys, time, subprocess, gzip
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as B, host_oracle as H

R, T, AVG = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 20_000, 8
t0 = time.time()
p, aux = B.synth_problem(R=R, T=T, avg_hits=AVG, seed=1234, sort=False)
names = ["T%07d" % i for i in range(T)]
rng = np.random.default_rng(1)
genes, i, g = {}, 0, 0
while i < T:
    sz = int(1 + rng.poisson(3)); genes["G%06d" % g] = names[i:i + sz]; i += sz; g += 1
efflen = {n: float(aux["efflen"][j]) for j, n in enumerate(names)}
truelen = {n: int(aux["efflen"][j]) + 180 for j, n in enumerate(names)}
rp = p.row_ptr.astype(np.int64); ci = p.col_idx
reads = [("r%09d" % r, [names[c] for c in ci[rp[r]:rp[r + 1]]]) for r in range(R)]
h = H.HitsData(names, efflen, truelen, genes, [], reads)
path = "/tmp/scale.hits"
open(path, "wb").write(H.write_hits_binary(h))
print("generated %d reads, %.1f MB binary hits file in %.1fs" % (R, os.path.getsize(path) / 1e6, time.time() - t0), flush=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmseq_amd", "csrc", "mmseq")
t0 = time.time()
r = subprocess.run([exe, "-gibbs_iter", "1024", path, "/tmp/scale_out"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                   env=dict(os.environ, MMSEQ_TIMING="1"))
print("mmseq rc=%d wall=%.1fs" % (r.returncode, time.time() - t0))
out = r.stdout.decode().replace("\r", "\n")
print("EM iterations:", out.count("EM iteration"), " Gibbs lines:", out.count("Gibbs iteration"))
print(out[-1500:])
print(r.stderr.decode()[-1500:])
tab = open("/tmp/scale_out.mmseq").read().split("\n")
print(tab[0], "| rows", len(tab) - 3)
uh = sum(int(l.split("\t")[7]) for l in tab[2:-1])
print("sum unique_hits", uh, "trace bytes", os.path.getsize("/tmp/scale_out.trace_gibbs.gz"))
