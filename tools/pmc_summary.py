"""Summarise the rocprofv3 passes of tools/profile_round.sh (rocpd .db files under gpurun_out/<tag>_*/): per-kernel mean of every
counter, the HBM traffic of K1 (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH bytes = 2 x FETCH_SIZE x 1024 on gfx950 per
MI355X_MICROARCH.md, calibrated in profiles/r01_pmc_summary.md), its instruction mix, and profiles/pmc_counters.json -- what
bench.py reads back, stamped with a hash of the kernel sources so that it cannot outlive the kernel it was measured on.
usage: pmc_summary.py <tag> [kernel_substr]      (prints markdown)"""
import glob, hashlib, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_sample_sell"
KERNEL_SOURCES = ["mmg_math.h", "mmg_types.h", "gibbs_kernels.h", "sell_kernels.h", "k1.hip"]


def dbs(pass_name):
    return sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, pass_name), "**", "*.db"), recursive=True))


def table_names(db):
    return [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]


def counters(pass_name):
    out = {}
    for path in dbs(pass_name):
        db = sqlite3.connect(path)
        names = table_names(db)
        src = "counters_collection" if "counters_collection" in names else None
        if not src:
            print("no counters_collection in", path, names[:20], file=sys.stderr)
            continue
        for kname, cname, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from %s group by kernel_name, counter_name" % src):
            out.setdefault(kname, {})[cname] = (n, avg)
    return out


allc = {}
for p in ("fetch", "write", "sqa", "sqb", "sqc", "sqd", "sqe"):
    for k, v in counters(p).items():
        allc.setdefault(k, {}).update(v)
k1 = [k for k in allc if sub in k]
print("# %s PMC passes (tools/profile_round.sh; one counter set per run, --kernel-trace + --pmc only)\n" % tag)
cols = sorted({c for k in allc for c in allc[k]})
for k in sorted(allc, key=lambda k: -allc[k].get("SQ_INSTS_VALU", (0, 0))[1]):
    name = k if len(k) < 100 else k[:97] + "..."
    print("## `%s`\n" % name)
    print("| counter | launches | mean per launch |\n|---|---|---|")
    for c in cols:
        if c in allc[k]:
            print("| %s | %d | %.6g |" % (c, allc[k][c][0], allc[k][c][1]))
    print()
if not k1:
    print("kernel %s not found in the passes" % sub, file=sys.stderr)
    sys.exit(0)
c = allc[k1[0]]
g = lambda name: c.get(name, (0, None))[1]
bench = None
try:
    bench = json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", "%s_bench.json" % tag)) if l.startswith("{")][-1])
except Exception as e:
    print("no bench line:", e, file=sys.stderr)
import re
h = hashlib.sha256()           # same rule as bench.py:kernel_hash(): comments and white space do not count
for f in KERNEL_SOURCES:
    src = open(os.path.join(ROOT, "mmseq_amd", "csrc", f)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    h.update(re.sub(r"\s+", "", src).encode())
out = {"kernel": sub, "kernel_sources_sha16": h.hexdigest()[:16],
       "workload": {"rows": 50_000_000, "transcripts": 200_000, "avg_hits": 20.0, "chains": 1},
       "hbm_read_bytes_per_launch": 2 * 1024 * g("FETCH_SIZE") if g("FETCH_SIZE") is not None else None,
       "hbm_write_bytes_per_launch": 1024 * g("WRITE_SIZE") if g("WRITE_SIZE") is not None else None,
       "counters_per_launch": {k: v[1] for k, v in c.items()},
       "source": "profiles/%s_pmc_summary.md (rocprofv3 --kernel-trace --pmc, separate passes; FETCH doubled per MI355X_MICROARCH.md)" % tag}
valu = g("SQ_INSTS_VALU")
if valu is not None:
    # a wave64 VALU instruction occupies its SIMD for one pass of 4 clocks; v_mad_u64_u32 (the Philox2x32 multiply) runs at
    # quarter rate, 3 more passes each: 10 per PAIR of tiles in k_sample_sell (disassembly), i.e. 5 per register-path tile --
    # bench.py multiplies by the tile count of the problem it runs.  (SQ_INSTS_VALU_INT64 also counts full-rate 64-bit adds.)
    out["valu_insts_per_launch"] = valu
    out["quarter_rate_valu_per_fast_tile"] = 5
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_counters.json"), "w"), indent=1)
print("## K1 summary\n")
if out["hbm_read_bytes_per_launch"] is not None:
    print("HBM read %.4f GB + written %.4f GB per launch" % (out["hbm_read_bytes_per_launch"] / 1e9, (out["hbm_write_bytes_per_launch"] or 0) / 1e9))
if valu is not None:
    tot = g("SQ_INSTS") or 0
    print("\ninstructions per launch: VALU %.4g, SALU %.4g, SMEM %.4g, LDS %.4g, VMEM %.4g, branch %.4g, all %.4g" % (
        valu, g("SQ_INSTS_SALU") or 0, g("SQ_INSTS_SMEM") or 0, g("SQ_INSTS_LDS") or 0, g("SQ_INSTS_VMEM") or 0, g("SQ_INSTS_BRANCH") or 0, tot))
