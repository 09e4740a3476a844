#!/bin/bash
# Round 5: launch duration vs frames per launch for the 8-bit configurations (a + b x frames: what is fixed cost, what is the marginal rate)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r05_sweep_8bit.jsonl
for c in 3 5 6; do for b in 1 2 4 8 16 32; do
  timeout -k 10 120 python3 tools/bench_config.py --config $c --batch $b >> gpurun_out/r05_sweep_8bit.jsonl 2>> gpurun_out/r05_sweep_8bit.err || exit 1
done; done
python3 - <<'PY'
import json, collections
rows = collections.defaultdict(list)
for l in open('gpurun_out/r05_sweep_8bit.jsonl'):
    r = json.loads(l); rows[r['config']].append(r)
for c, rs in rows.items():
    for r in rs: print(c, 'x%-2d' % r['frames_per_launch'], r['launch_us'], r['frac_of_8TBps'], r['kernel'])
    x8, x32 = [r for r in rs if r['frames_per_launch'] == 8][0], [r for r in rs if r['frames_per_launch'] == 32][0]
    b = (x32['launch_us'] - x8['launch_us']) / 24; a = x8['launch_us'] - 8 * b
    print('   marginal %.3f us/frame = %.4f of 8 TB/s; fixed %.2f us per launch' % (b, x8['algorithmic_bytes_per_frame'] / b / 8e6, a))
PY
