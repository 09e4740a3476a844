#!/bin/bash
# Round 5 probe (gpurun): what the partly filled last position of every row costs.  All BASELINE widths are 15/16 of a power of
# two, so a row's units fill 15/16 of its positions (3840 B = 3.75 KiB; + the half-block shift -> 4 positions).  Here the same
# kernels run on pictures whose rows fill their positions (4080 / 8176 wide): same code, same row count, 99.6 % of the lanes busy.
# usage: bash tools/dev/r05_lane_fill_probe.sh name...   (library variants in tools/bin, see build_variant.sh)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for n in "$@"; do
  for cw in ${CW:-3:8:0 3:8:4080 4:8:0 4:8:8176 2:8:0 2:8:4080}; do
    c=${cw%%:*}; r=${cw#*:}; b=${r%%:*}; r=${r#*:}; w=${r%%:*}; st=${r#*:}; [ "$st" = "$r" ] && st=0
    VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$GRAFT_REPO_ROOT/tools/bin/$n.so python3 tools/bench_config.py --config $c --batch $b --steps 200 --width $w --stride $st 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('round $round  %-12s cfg %d x%d width %9s  %8.3f us/frame  %.4f  %s' % ('$n', d['config'], d['frames_per_launch'], '$w/$st', d['us_per_frame'], d['frac_of_8TBps'], d['workload']))"
  done
done
done
