#!/bin/bash
# Profile passes of a bench workload on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03 [extra bench.py arguments, e.g. --chains 8]
#        -> gpurun_out/<tag>_{kt,fetch,write,sqa,sqb,sqc,sqd,sqe}/ + gpurun_out/<tag>_bench.json
# Counters are collected in their own runs (--kernel-trace + --pmc only; never together with other trace domains), FETCH_SIZE and
# WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes.  tools/pmc_summary.py turns the result into profiles/.
set -u
TAG=${1:-r03}
shift || true
ARGS="$*"
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 8 --warmup 2 --settle-iters 0 --no-extra --no-cpu-baseline $ARGS"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt" -- python3 "$REPO/bench.py" --no-extra --no-cpu-baseline $ARGS > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_kt.err"
python3 "$REPO/bench.py" --no-extra --no-cpu-baseline $ARGS > "$OUT/${TAG}_bench_plain.json" 2>> "$OUT/${TAG}_kt.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/${TAG}_fetch" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/${TAG}_write" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_write.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH -d "$OUT/${TAG}_sqa" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqa.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/${TAG}_sqb" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqb.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d "$OUT/${TAG}_sqc" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqc.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS SQ_WAVES -d "$OUT/${TAG}_sqd" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqd.err"
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH -d "$OUT/${TAG}_sqe" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqe.err"
cd "$REPO"
python3 tools/pmc_summary.py "$TAG" > "$OUT/${TAG}_pmc_summary.md" 2> "$OUT/${TAG}_pmc_summary.err"
python3 tools/rocpd_summary.py "$(find "$OUT/${TAG}_kt" -name "*.db" | head -1)" > "$OUT/${TAG}_kernel_stats.md" 2>> "$OUT/${TAG}_pmc_summary.err"
cp profiles/pmc_counters.json "$OUT/${TAG}_pmc_counters.json"
# the raw rocpd databases are large (gpurun copies at most 64 MiB back): the summaries above are what is kept
rm -rf "$OUT/${TAG}_fetch" "$OUT/${TAG}_write" "$OUT/${TAG}_sqa" "$OUT/${TAG}_sqb" "$OUT/${TAG}_sqc" "$OUT/${TAG}_sqd" "$OUT/${TAG}_sqe" "$OUT/${TAG}_kt"
tail -5 "$OUT/${TAG}_pmc_summary.err"
head -c 400 "$OUT/${TAG}_bench.json"
