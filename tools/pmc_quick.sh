#!/bin/bash
# One PMC pass over tools/k1_time.py (no bench protocol): [KPAT=k_update] [PROBLEM="5000000 50000 8"] tools/pmc_quick.sh <tag> <chains> <counters...>
set -u
TAG=$1; CH=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/${TAG}_q" -- python3 "$REPO/tools/k1_time.py" "$CH" ${PROBLEM:-} > "$OUT/${TAG}_q.log" 2> "$OUT/${TAG}_q.err"
cd "$REPO"
KPAT=${KPAT:-k_sample_sell} python3 - "$OUT/${TAG}_q" <<'PY'
import glob, sqlite3, sys
for path in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(path)
    for kname, cname, n, avg in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%" + __import__('os').environ.get('KPAT', 'k_sample_sell') + "%' group by kernel_name, counter_name"):
        print("%-34s %-24s launches %4d  mean %.6g" % (kname[:34].replace("void mmg::", ""), cname, n, avg))
PY
rm -rf "$OUT/${TAG}_q"
