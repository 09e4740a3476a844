#!/bin/bash
# Round 5 probe (gpurun): the same pictures with the row pitch padded to a power of two (tools/walk_probe.hip: streams at a power-of-two pitch run 3-5 % faster).
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for cws in 4:8:7680:0 4:8:7680:8192 2:8:3840:0 2:8:3840:4096 3:8:3840:0 3:8:3840:4096 0:32:1920:0 0:32:1920:2048 5:8:3840:0 5:8:3840:4096; do
  c=${cws%%:*}; r=${cws#*:}; b=${r%%:*}; r=${r#*:}; w=${r%%:*}; st=${r#*:}
  python3 tools/bench_config.py --config $c --batch $b --steps 200 --stride $st 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('round $round cfg %2d x%-2d stride %5s  %8.3f us/frame  %.4f (of the picture bytes)  %s' % (d['config'], d['frames_per_launch'], '$st', d['us_per_frame'], d['frac_of_8TBps'], d['workload']))"
done
done
