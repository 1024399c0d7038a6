#!/bin/bash
# Profile passes of the default bench workload on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02      -> gpurun_out/r02_{kt,fetch,write,sqa,sqb,sqc,sqd}/ + gpurun_out/r02_bench.json
# Counters are collected in their own runs (--kernel-trace + --pmc only; never together with other trace domains), FETCH_SIZE and
# WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes.  tools/pmc_summary.py turns the result into profiles/.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 8 --warmup 2 --settle-iters 0 --no-extra --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt" -- python3 "$REPO/bench.py" --no-extra --no-cpu-baseline > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_kt.err"
python3 "$REPO/bench.py" --no-extra --no-cpu-baseline > "$OUT/${TAG}_bench_plain.json" 2>> "$OUT/${TAG}_kt.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/${TAG}_fetch" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/${TAG}_write" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_write.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH -d "$OUT/${TAG}_sqa" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqa.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/${TAG}_sqb" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqb.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d "$OUT/${TAG}_sqc" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqc.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS SQ_WAVES -d "$OUT/${TAG}_sqd" -- python3 "$REPO/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_sqd.err"
cd "$REPO"
python3 tools/pmc_summary.py "$TAG" > "$OUT/${TAG}_pmc_summary.md" 2> "$OUT/${TAG}_pmc_summary.err"
tail -5 "$OUT/${TAG}_pmc_summary.err"
head -c 600 "$OUT/${TAG}_bench.json"
