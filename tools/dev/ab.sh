#!/bin/bash
# Developer helper for gpurun: time prebuilt library variants (tools/bin/*.so, see build_variant.sh) interleaved on ONE box.
# usage: [VCFG="4:8 0:8"] [ROUNDS=3] bash tools/dev/ab.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${ROUNDS:-3}); do
for n in "$@"; do
  for cb in ${VCFG:-4:8}; do c=${cb%:*}; b=${cb#*:}; VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$GRAFT_REPO_ROOT/tools/bin/$n.so python3 tools/bench_config.py --config $c --batch $b --steps ${STEPS:-100} ${EXTRA:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('round $round  %-28s cfg %d x%d  %8.3f us/frame  %.4f' % ('$n', d['config'], d['frames_per_launch'], d['us_per_frame'], d['frac_of_8TBps']))"; done
done
done
