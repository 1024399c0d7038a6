#!/bin/bash
# One quick counter pass of a kernel under two builds of the library (A/B of a kernel change on one box):
#   tools/pmc_quick.sh <kernel substring> "<counters>" lib_a.so lib_b.so -- <bench.py arguments>
# prints the mean of every counter per launch of the kernel for each library.  --kernel-trace + --pmc only.
set -u
SUB=$1; CNT=$2; shift 2
LIBS=()
while [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for L in "${LIBS[@]}"; do
  D=/tmp/pmcq_$(basename $L .so)
  rm -rf $D
  export MMSEQ_AMD_LIB=$REPO/$L
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $D -- python3 $REPO/bench.py --no-extra --no-cpu-baseline --no-live-pmc --settle-iters 0 "$@" > /dev/null 2> $D.err
  python3 - "$D" "$SUB" "$L" <<'PY'
import csv, glob, sys, collections
d, sub, lib = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(lib, {k: "%.4g (n=%d)" % (v[0] / max(v[1], 1), v[1]) for k, v in sorted(acc.items())})
PY
done
