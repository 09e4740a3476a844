#!/bin/bash
# Round 5: GPU tests + the 8-bit configs of the tracked matrix (3: 2160p 4:4:4 AFGS1, 5: 2160p 4:2:0 AFGS1, 6: 2160p 4:2:0 fgs_sei) and the
# headline (4) at 8 frames per launch.  Usage (gpurun): bash tools/gpu_r5_w16.sh [TAG]
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_w16}
mkdir -p gpurun_out
bash tools/gpu_check.sh || exit 1
for c in 3 5 6 4 0 2; do for b in 8 32; do
  [ $c = 4 ] && [ $b = 32 ] && continue
  python3 tools/bench_config.py --config $c --batch $b --steps 200 2>/dev/null | tee -a gpurun_out/${TAG}_matrix.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['workload'], 'x', d['frames_per_launch'], 'us/launch', d['launch_us'], 'frac', d['frac_of_8TBps'], d['kernel'])"
done; done
