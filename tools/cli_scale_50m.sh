#!/bin/bash
# End-to-end run of the drop-in CLI on a 50 M-read hits FILE (GPU box): synth_hits writes the benchmark workload (config 3 shape) as a
# binary hits file, mmseq reads it, collapses, runs EM + 1024 Gibbs iterations and writes every output.  Stage timings to stdout.
#   [ITER=16384] [GENES="32 3"] tools/cli_scale_50m.sh [ROWS [TRANSCRIPTS [AVG [FAR]]]]   (ITER: -gibbs_iter, default 1024; the reference default is 16384;
#   ZIPF=CAP: ROWS hit sets, each shared by k reads, k Zipf up to CAP, the reads shuffled (what mmseq collapses: ROWS=2000000 AVG=10 ZIPF=1000000 is bench.py's `collapsed`);
#   GENES="G F": the generator's gene-block mode -- genes of G isoforms, far hits to a gene of the read's paralogue family of F genes)
set -u
R=${1:-50000000}; T=${2:-200000}; A=${3:-20}; F=${4:-0.0}
D=$(mktemp -d /tmp/mmseq_scale.XXXXXX)
BIN=$(pwd)/mmseq_amd/csrc
t0=$(date +%s%N)
$BIN/synth_hits ${ZIPF:+-zipf $ZIPF} ${GENES:+-genes $GENES} $R $T $A $D/in.hits $F || exit 1
t1=$(date +%s%N)
echo "synth_hits: $(( (t1 - t0) / 1000000 )) ms, file $(stat -c %s $D/in.hits) bytes"; df -h $D | tail -1
if [ -n "${PROFILE:-}" ]; then   # PROFILE=<dir under gpurun_out>: the same run under rocprofv3 --kernel-trace --stats
  export MMSEQ_TIMING=1
  REPO=$(pwd)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/$PROFILE -- $BIN/mmseq -gibbs_iter ${ITER:-1024} $D/in.hits $D/out > $D/stdout.log 2> $D/stderr.log)
  rc=$?
else
  MMSEQ_TIMING=1 $BIN/mmseq -gibbs_iter ${ITER:-1024} $D/in.hits $D/out > $D/stdout.log 2> $D/stderr.log
  rc=$?
fi
t2=$(date +%s%N)
echo "mmseq rc=$rc wall $(( (t2 - t1) / 1000000 )) ms"
grep -c "EM iteration" $D/stdout.log | sed 's/^/EM iterations: /'
grep "\[timing\]" $D/stderr.log
tail -3 $D/stderr.log
ls -l $D | awk '{print $5, $9}'
head -3 $D/out.mmseq | cut -c1-200
rm -rf $D
