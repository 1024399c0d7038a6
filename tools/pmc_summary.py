"""Per-kernel HBM traffic from two rocprofv3 PMC passes (rocpd .db): FETCH_SIZE and WRITE_SIZE, collected separately.
Counters are in KiB; FETCH bytes = 2 x FETCH_SIZE x 1024 on gfx950 (MI355X_MICROARCH.md, HBM section; calibrated in
profiles/r01_pmc_summary.md), WRITE bytes = WRITE_SIZE x 1024.  usage: pmc_summary.py fetch.db write.db [json_out kernel_substr]"""
import json, sqlite3, sys

def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                      "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}

f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | FETCH_SIZE (KiB, mean) | HBM read = 2 x 1024 x FETCH (GB) | WRITE_SIZE (KiB, mean) | HBM written (GB) |")
print("|---|---|---|---|---|---|")
for k in f:
    n, fv = f[k]
    wv = w.get(k, (0, 0.0))[1]
    name = k if len(k) < 90 else k[:87] + "..."
    print("| %s | %d | %.1f | %.4f | %.1f | %.4f |" % (name, n, fv, 2 * 1024 * fv / 1e9, wv, 1024 * wv / 1e9))
if len(sys.argv) > 4:
    sub = sys.argv[4]
    k = [x for x in f if sub in x][0]
    out = {"kernel": sub, "hbm_read_bytes_per_launch": 2 * 1024 * f[k][1], "hbm_write_bytes_per_launch": 1024 * w[k][1]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
