#!/bin/bash
# PMC passes over the no-locality fallback kernel (k_sample on hits uniform over all transcripts) -> gpurun_out/r03_uniform_pmc.md
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d "$OUT/uni_$i" -- python3 "$REPO/tools/uniform_time.py" 4 > "$OUT/uni_$i.log" 2> "$OUT/uni_$i.err"
done
cd "$REPO"
python3 - "$OUT" <<'PY' > "$OUT/r03_uniform_pmc.md"
import glob, sqlite3, sys
out = sys.argv[1]
print("# `k_sample` (CSR tiles) on 50 M rows x 20 hits uniform over 200 k transcripts: PMC counters per launch (rocprofv3 --kernel-trace --pmc, one pass per line group)\n")
print("| counter | launches | mean per launch |\n|---|---|---|")
for d in sorted(glob.glob(out + "/uni_*/")):
    for path in glob.glob(d + "**/*.db", recursive=True):
        db = sqlite3.connect(path)
        for cname, n, avg in db.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_sample<%' group by counter_name"):
            print("| %s | %d | %.6g |" % (cname, n, avg))
PY
cat "$OUT/uni_1.log" | tail -1 >> "$OUT/r03_uniform_pmc.md"
rm -rf "$OUT"/uni_*/
cat "$OUT/r03_uniform_pmc.md"
