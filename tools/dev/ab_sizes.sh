#!/bin/bash
# Developer helper for gpurun: time prebuilt library variants (tools/bin/*.so) over picture SIZES of one configuration, interleaved on ONE box.
# usage: [CFG=7] [SIZES="1280x720:32 1920x1080:16 ..."] [ROUNDS=3] bash tools/dev/ab_sizes.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${ROUNDS:-3}); do
for szb in ${SIZES:-1280x720:32 1920x1080:16 2560x1440:16 3840x2160:8 5120x2880:8 7680x4320:8}; do
  sz=${szb%:*}; b=${szb#*:}; w=${sz%x*}; h=${sz#*x}
  for n in "$@"; do
    VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$GRAFT_REPO_ROOT/tools/bin/$n.so python3 tools/bench_config.py --config ${CFG:-7} --batch $b --width $w --height $h --steps ${STEPS:-100} ${EXTRA:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('round $round  %-20s cfg %d %-10s x%-3d %8.3f us/frame  %.4f' % ('$n', d['config'], '$sz', d['frames_per_launch'], d['us_per_frame'], d['frac_of_8TBps']))"
  done
done
done
