#!/bin/bash
# Round 5: GPU tests + frame-list vs contiguous batches (bench_config --list) on one box.  Usage (gpurun): bash tools/gpu_r5_list.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_check.sh || exit 1
for round in 1 2; do
for cb in 0:32 0:8 1:32 3:8 4:8 2:16; do c=${cb%:*}; b=${cb#*:}
  for m in "" "--list"; do
  python3 tools/bench_config.py --config $c --batch $b --steps 200 $m 2>/dev/null | tee -a gpurun_out/r05_frame_list.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('round $round cfg', d['config'], d['workload'], 'x', d['frames_per_launch'], 'list' if d['frame_list'] else 'pitch', 'us/launch', d['launch_us'], 'host us/call', d['host_us_per_call'], 'frac', d['frac_of_8TBps'])"
  done
done
done
