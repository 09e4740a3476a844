#!/bin/bash
# Developer helper for gpurun (round 6: with the stripe stream reached by jumps): the N-rank bench path rehearsed on ONE GPU (the box allows at most 6 processes on the card -- N ranks + the
# launcher agent -- so N = 5 is the most that can be rehearsed here), weak and strong, + the host cost per call at the 1..8 rank shapes.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06_multirank_rehearsal_one_gpu.jsonl
: > $OUT
for n in ${RANKS:-2 4 5}; do for sc in weak strong; do
  timeout -k 10 240 python bench.py --gpus $n --rehearse-on-one-gpu --scaling $sc --steps 20 --warmup 5 --no-ceiling --no-region 2>> gpurun_out/r06_multirank_full.err >> $OUT || echo "{\"error\": \"gpus $n $sc rc $?\"}" >> $OUT
done; done
python3 - <<'PY'
import json
for l in open('gpurun_out/r06_multirank_rehearsal_one_gpu.jsonl'):
    d = json.loads(l)
    if 'error' in d: print(d); continue
    print('gpus %d %-6s ranks seen %d parity %s frames/step %3d  %8.1f Mpixels/s  ms/step %.3f  launch us per rank %s' % (d['n_gpus'], d['scaling'], d['config']['n_ranks_seen'], d['parity_checked'], d['config']['frames_per_step'], d['value'], d['ms_per_step'], d['config']['launch_us_per_rank']))
PY
