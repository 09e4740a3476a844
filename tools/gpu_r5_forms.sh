#!/bin/bash
# Round 5 (gpurun): GPU tests + the matrix lines of the forms that changed: 4:4:4 with general-form chroma (four workgroups per CU since the chroma
# image holds one LUT pair), rows walked in parts with one-pattern forms.  Usage: bash tools/gpu_r5_forms.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_check.sh || exit 1
for round in 1 2; do
for cb in 11:8 12:8 8:2 13:2 14:2 15:2 3:8 4:8; do c=${cb%:*}; b=${cb#*:}
  python3 tools/bench_config.py --config $c --batch $b --steps 200 2>/dev/null | tee -a gpurun_out/r05_forms.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('round $round cfg', d['config'], d['workload'], 'x', d['frames_per_launch'], 'us/launch', d['launch_us'], 'frac', d['frac_of_8TBps'], d['kernel'])"
done
done
