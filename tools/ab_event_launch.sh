#!/bin/bash
# A/B of whole bench steps between builds of libmmgibbs.so on one box (the driver's command, three rounds, alternating):
#   tools/ab_event_launch.sh build_ab/lib_a.so mmseq_amd/csrc/libmmgibbs.so
# (round 5: recorded timing events against events attached to the launches -- value, ms per step, K1 and K2 per launch)
for r in 1 2 3; do
for lib in "$@"; do
  MMSEQ_AMD_LIB=$lib timeout 200 python bench.py --no-extra --no-cpu-baseline --no-live-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['k_update_avg_launch_ms'])"
done; done
