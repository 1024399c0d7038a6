#!/bin/bash
# Profile passes of one workload on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> <kernel substring> <entry name> <script> [script arguments]
#   e.g.  tools/profile_round.sh r03_k1  k_sample_sell\<        k1_1chain   bench.py --no-extra --no-cpu-baseline
#         tools/profile_round.sh r03_k1m k_sample_sell_multi     k1m_8chains bench.py --no-extra --no-cpu-baseline --chains 8 --steps 32 --warmup 8
#         tools/profile_round.sh r03_em  k_em_sell               em_sweep    tools/em_probe.py
#   -> gpurun_out/<tag>_{kernel_stats.md,pmc_summary.md,pmc_counters.json,bench.json}
# Counters are collected in their own runs (--kernel-trace + --pmc only; never together with other trace domains), FETCH_SIZE and
# WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes.  tools/pmc_summary.py turns the result into profiles/.
set -u
TAG=$1; SUB=$2; ENTRY=$3; SCRIPT=$4; shift 4
ARGS="$*"
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
case "$SCRIPT" in *bench.py) ARGS="$ARGS --no-live-pmc";; esac   # (bench.py's own counter passes are child runs under rocprofv3: not inside one)
SHORT="$ARGS"
case "$SCRIPT" in *bench.py) SHORT="$ARGS --steps 8 --warmup 2 --settle-iters 0";; esac
rocprofv3 --kernel-trace --stats -d "$OUT/${TAG}_kt" -- python3 "$REPO/$SCRIPT" $ARGS > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_kt.err"
python3 "$REPO/$SCRIPT" $ARGS > "$OUT/${TAG}_bench_plain.json" 2>> "$OUT/${TAG}_kt.err"
pass() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/${TAG}_${name}" -- python3 "$REPO/$SCRIPT" $SHORT > /dev/null 2> "$OUT/${TAG}_${name}.err"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sqa SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass sqb SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sqc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass sqd SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS SQ_WAVES
pass sqe SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH
cd "$REPO"
python3 tools/pmc_summary.py "$TAG" "$SUB" "$ENTRY" > "$OUT/${TAG}_pmc_summary.md" 2> "$OUT/${TAG}_pmc_summary.err"
python3 tools/rocpd_summary.py "$(find "$OUT/${TAG}_kt" -name "*.db" | head -1)" > "$OUT/${TAG}_kernel_stats.md" 2>> "$OUT/${TAG}_pmc_summary.err"
cp profiles/pmc_counters.json "$OUT/pmc_counters.json"
# the raw rocpd databases are large (gpurun copies at most 64 MiB back): the summaries above are what is kept
rm -rf "$OUT/${TAG}_fetch" "$OUT/${TAG}_write" "$OUT/${TAG}_sqa" "$OUT/${TAG}_sqb" "$OUT/${TAG}_sqc" "$OUT/${TAG}_sqd" "$OUT/${TAG}_sqe" "$OUT/${TAG}_kt"
tail -3 "$OUT/${TAG}_pmc_summary.err"
tail -12 "$OUT/${TAG}_pmc_summary.md"
