#!/bin/bash
# Developer helper for gpurun (round 4): frames-per-launch sweep of the BASELINE configs + the everyday 8-bit formats,
# the out-of-place forms, host cost per call at the multi-rank shapes.  Plain stream, device-resident frames.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r04_sweep.jsonl
: > $OUT
for c in 0 1 2 3 5 6; do for b in 8 16 32 64; do
  python3 tools/bench_config.py --config $c --batch $b --steps 100 2>/dev/null >> $OUT
done; done
for c in 4; do for b in 8 16; do python3 tools/bench_config.py --config $c --batch $b --steps 100 2>/dev/null >> $OUT; done; done
for m in copy copy8; do for c in 0 2 4; do
  python3 tools/bench_config.py --config $c --batch 8 --steps 100 --mode $m 2>/dev/null >> $OUT
done; done
python3 - <<'PY'
import json
for l in open('gpurun_out/r04_sweep.jsonl'):
    d = json.loads(l)
    print('%d %-42s %-7s x%-3d %9.2f us/launch %8.3f us/frame  %.4f' % (d['config'], d['workload'], d['mode'], d['frames_per_launch'], d['launch_us'], d['us_per_frame'], d['frac_of_8TBps']))
PY
python3 tools/host_overhead.py > gpurun_out/r04_host_overhead.log 2>&1; cat gpurun_out/r04_host_overhead.log
