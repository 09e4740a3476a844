#!/bin/bash
# Developer helper for gpurun (round 4): the config matrix of the shipped library -- in place, out of place, fused 8-bit output --
# at 8 frames per launch and at a ~400 MB batch, one JSON line each (profiles/r04_config_matrix.jsonl).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_config_matrix.jsonl
: > $OUT
for c in 0 1 2 3 4 5 6; do
  case $c in 0|1) bs="1 8 32";; 4) bs="1 8";; *) bs="1 8 16";; esac
  for b in $bs; do python3 tools/bench_config.py --config $c --batch $b --steps 100 2>/dev/null >> $OUT; done
done
for m in copy copy8; do for c in 0 1 2 3 4 7; do
  if [ $m = copy8 ] && [ $c = 3 ]; then continue; fi
  case $c in 0|1) bs="8 32";; 4) bs="8";; *) bs="8";; esac
  for b in $bs; do python3 tools/bench_config.py --config $c --batch $b --steps 100 --mode $m 2>/dev/null >> $OUT; done
done; done
python3 - <<'PY'
import json
for l in open('gpurun_out/r04_config_matrix.jsonl'):
    d = json.loads(l)
    print('%d %-42s %-7s x%-3d %9.2f us/launch %8.3f us/frame  %.4f  %s' % (d['config'], d['workload'], d['mode'], d['frames_per_launch'], d['launch_us'], d['us_per_frame'], d['frac_of_8TBps'], d['kernel']))
PY
