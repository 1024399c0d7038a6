"""Summarise the rocprofv3 passes of tools/profile_round.sh (rocpd .db files under gpurun_out/<tag>_*/): per-kernel mean of every
counter as markdown, and one entry of profiles/pmc_counters.json -- what bench.py reads back, stamped with a hash of the kernel
sources so that it cannot outlive the kernels it was measured on.  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH bytes = 2 x FETCH_SIZE x
1024 on gfx950 per MI355X_MICROARCH.md (calibrated in profiles/r01_pmc_summary.md).
usage: pmc_summary.py <tag> <kernel_substr> <entry>      (prints markdown)"""
import glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, sub, entry = sys.argv[1], sys.argv[2], sys.argv[3]


def counters(pass_name):
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, pass_name), "**", "*.db"), recursive=True)):
        db = sqlite3.connect(path)
        names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in names:
            print("no counters_collection in", path, file=sys.stderr)
            continue
        for kname, cname, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            out.setdefault(kname, {})[cname] = (n, avg)
    return out


def durations(pass_name):
    """mean kernel duration (ns) inside one PMC pass: the clock of THAT pass = its GRBM_GUI_ACTIVE / 8 / its duration"""
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, pass_name), "**", "*.db"), recursive=True)):
        db = sqlite3.connect(path)
        try:
            for name, avg in db.execute("select name, avg(duration) from kernels group by name"):
                out[name] = avg
        except sqlite3.Error as e:
            print("no kernel durations in", path, e, file=sys.stderr)
    return out


dur_sqd = durations("sqd")
allc = {}
for p in ("fetch", "write", "sqa", "sqb", "sqc", "sqd", "sqe"):
    for k, v in counters(p).items():
        allc.setdefault(k, {}).update(v)
print("# %s PMC passes (tools/profile_round.sh; one counter set per run, --kernel-trace + --pmc only)\n" % tag)
cols = sorted({c for k in allc for c in allc[k]})
for k in sorted(allc, key=lambda k: -allc[k].get("SQ_INSTS_VALU", (0, 0))[1]):
    if "mmg::" not in k:
        continue                       # library kernels only
    name = k if len(k) < 100 else k[:97] + "..."
    print("## `%s`\n" % name)
    print("| counter | launches | mean per launch |\n|---|---|---|")
    for c in cols:
        if c in allc[k]:
            print("| %s | %d | %.6g |" % (c, allc[k][c][0], allc[k][c][1]))
    print()
hit = [k for k in allc if sub in k]
if not hit:
    print("kernel %s not found in the passes" % sub, file=sys.stderr)
    sys.exit(0)
c = allc[hit[0]]
g = lambda name: c.get(name, (0, None))[1]
from bench import kernel_hash  # the same stamp bench.py checks
path = os.path.join(ROOT, "profiles", "pmc_counters.json")
try:
    doc = json.load(open(path))
    if "entries" not in doc or doc.get("kernel_sources_sha16") != kernel_hash():
        doc = {"entries": {}}
except Exception:
    doc = {"entries": {}}
doc["kernel_sources_sha16"] = kernel_hash()
doc["entries"][entry] = {
    "kernel": hit[0][:120],
    "hbm_read_bytes_per_launch": 2 * 1024 * g("FETCH_SIZE") if g("FETCH_SIZE") is not None else None,
    "hbm_write_bytes_per_launch": 1024 * g("WRITE_SIZE") if g("WRITE_SIZE") is not None else None,
    "counters_per_launch": {k: v[1] for k, v in c.items()},
    "duration_ns_in_the_grbm_pass": next((v for k, v in dur_sqd.items() if sub in k), None),
    "source": "profiles/%s_pmc_summary.md (rocprofv3 --kernel-trace --pmc, separate passes; FETCH doubled per MI355X_MICROARCH.md)" % tag}
json.dump(doc, open(path, "w"), indent=1)
e = doc["entries"][entry]
print("## summary of `%s` (entry %s)\n" % (sub, entry))
if e["hbm_read_bytes_per_launch"] is not None:
    print("HBM read %.4f GB + written %.4f GB per launch" % (e["hbm_read_bytes_per_launch"] / 1e9, (e["hbm_write_bytes_per_launch"] or 0) / 1e9))
if g("SQ_INSTS_VALU") is not None:
    print("\ninstructions per launch: VALU %.4g, SALU %.4g, SMEM %.4g, LDS %.4g, VMEM %.4g, branch %.4g, all %.4g" % (
        g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU") or 0, g("SQ_INSTS_SMEM") or 0, g("SQ_INSTS_LDS") or 0, g("SQ_INSTS_VMEM") or 0, g("SQ_INSTS_BRANCH") or 0, g("SQ_INSTS") or 0))
if g("GRBM_GUI_ACTIVE") is not None and g("SQ_LDS_IDX_ACTIVE") is not None:
    cyc = g("GRBM_GUI_ACTIVE") / 8.0          # the counter is summed over the 8 XCDs
    print("\nkernel cycles %.4g; LDS pipe busy %.0f %% (bank conflicts %.0f %% of its cycles); waves: %.0f %% of their cycles at s_waitcnt, %.0f %% waiting to issue" % (
        cyc, 100 * g("SQ_LDS_IDX_ACTIVE") / 256 / cyc, 100 * (g("SQ_LDS_BANK_CONFLICT") or 0) / max(g("SQ_LDS_IDX_ACTIVE"), 1),
        100 * (g("SQ_WAIT_ANY") or 0) / max(g("SQ_WAVE_CYCLES") or 1, 1), 100 * (g("SQ_WAIT_INST_ANY") or 0) / max(g("SQ_WAVE_CYCLES") or 1, 1)))
