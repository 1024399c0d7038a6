#!/bin/bash
# The CLI's read phase on ONE 50 M-read file for several MMSEQ_INFLATE_THREADS (GPU box): synth_hits once, mmseq -gibbs_iter 1024 per setting.
#   tools/cli_inflate_threads.sh "1 4 6 8 12"
set -u
D=$(mktemp -d /tmp/mmseq_scale.XXXXXX)
BIN=$(pwd)/mmseq_amd/csrc
$BIN/synth_hits 50000000 200000 20 $D/in.hits 0.0 || exit 1
for T in ${1:-1 4 6 8 12}; do
  MMSEQ_TIMING=1 MMSEQ_INFLATE_THREADS=$T $BIN/mmseq -gibbs_iter 1024 $D/in.hits $D/out > $D/stdout.log 2> $D/stderr.log
  echo "inflate threads $T: $(grep 'read hits file' $D/stderr.log) | $(grep 'ingest stages' $D/stderr.log | sed 's/.*waited: //') | $(grep 'total' $D/stderr.log | head -1)"
  rm -f $D/out.*
done
rm -rf $D
