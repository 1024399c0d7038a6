"""A/B timing of K1 between builds of libmmgibbs.so on ONE box (same box, same process regime: one HIP runtime per process).
usage: k1_ab.py lib_a.so lib_b.so ... [--rows R --transcripts T --avg A --far F --rounds N]
Each library is measured in its own subprocess, round-robin, N rounds; prints the mean K1 launch time per library and round."""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from mmseq_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import ctypes, os
if os.environ.get("K1_AB_RUNTIME", "torch") == "torch":
    _lib._share_hip_runtime_with_torch()     # the HIP runtime bench.py and the tests run on (PyTorch's bundled libamdhip64)
else:
    _lib._share_hip_runtime_with_torch = lambda: None   # K1_AB_RUNTIME=system: /opt/rocm's only, what the C++ CLI binds.  NEVER both
                                             # in one process: the library then sees 12 resident waves per CU instead of 28
_probe = ctypes.CDLL(sys.argv[1])
for _name in list(_lib.SYMBOLS):          # older builds export fewer entry points: bind what is there
    if not hasattr(_probe, _name):
        del _lib.SYMBOLS[_name]
from mmseq_amd import Problem, Sampler
R, T, A, C, F = int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
prob = Problem.synthetic(R, T, A, seed=1234, far_fraction=F)
mu0, _ = prob.start_values()
s = Sampler(prob, mu0, n_chains=C, gibbs_iter=1024, trace_len=1024, keep_trace=False, timing=1)
s.run(300); s.sync(); s.reset_timing()
s.run(200); s.sync()
tm = s.timing()
print("grid", prob.info.sample_grid, "CUs", prob.info.cu_count, file=sys.stderr)
print(json.dumps({"k1_ms": tm["sample_ms"] / tm["sample_launches"] / C, "grid": prob.info.sample_grid, "cus": prob.info.cu_count, "k2_ms": tm["update_ms"] / tm["update_launches"],
                  "stream": prob.info.stream_bytes}))
''' % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rows", type=int, default=50_000_000)
ap.add_argument("--transcripts", type=int, default=200_000)
ap.add_argument("--avg", type=float, default=20.0)
ap.add_argument("--chains", type=int, default=1)
ap.add_argument("--far", type=float, default=0.0, help="fraction of the rows with one hit anywhere in the transcriptome")
ap.add_argument("--rounds", type=int, default=2)
a = ap.parse_args()
res = {l: [] for l in a.libs}
for r in range(a.rounds):
    for l in a.libs:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l), str(a.rows), str(a.transcripts), str(a.avg), str(a.chains), str(a.far)],
                             capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", out.stderr[-500:], flush=True)
            continue
        d = json.loads(line[-1])
        res[l].append(d["k1_ms"])
        print("round %d %-40s K1 %.4f ms  K2 %.4f ms  stream %.3f GB  grid %s on %s CUs" % (r, os.path.basename(l), d["k1_ms"], d["k2_ms"], d["stream"] / 1e9,
                                                                                       d.get("grid"), d.get("cus")), flush=True)
for l in a.libs:
    if res[l]:
        print("%-40s mean K1 %.4f ms over %d rounds" % (os.path.basename(l), sum(res[l]) / len(res[l]), len(res[l])))
