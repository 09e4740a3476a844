#!/bin/bash
# Developer helper for gpurun: all BASELINE configs at 1 and 8 frames per launch (no profiler)
cd $GRAFT_REPO_ROOT
for c in 0 1 2 3 4; do for b in 1 8; do python3 tools/bench_config.py --config $c --batch $b --steps 100 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['workload'], 'batch', d['frames_per_launch'], 'us/frame', d['us_per_frame'], 'GB/s', d['GBps'], 'frac', d['frac_of_8TBps'])"; done; done
