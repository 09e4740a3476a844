#!/bin/bash
# Round 5 (gpurun): GPU tests, host cost per call at the 1..8 rank shapes (python + C caller), the default bench line, a 2-rank
# rehearsal on one GPU (weak + strong).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/bin
bash tools/gpu_check.sh || exit 1
python3 tools/host_overhead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_host_overhead.log
hipcc -O2 -w -o tools/bin/host_call_bench tools/host_call_bench.cpp -Iinclude -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd && \
  for g in "1920 1080" "3840 2160" "7680 4320"; do tools/bin/host_call_bench $g 2000; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_host_call_bench.log
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; tail -c 600 gpurun_out/r05_bench_default.json; tail -3 gpurun_out/r05_bench_default.err
for sc in weak strong; do
  timeout -k 10 240 python bench.py --gpus 2 --rehearse-on-one-gpu --scaling $sc --steps 20 --warmup 5 --no-ceiling --no-region 2>> gpurun_out/r05_multirank.err >> gpurun_out/r05_multirank_rehearsal.jsonl
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r05_multirank_rehearsal.jsonl'):
    d = json.loads(l)
    if 'error' in d: print(d); continue
    print('gpus %d %-6s ranks seen %d parity %s frames/step %3d  %8.1f Mpixels/s  ms/step %.3f  launch us per rank %s  affinity %s' % (d['n_gpus'], d['scaling'], d['config']['n_ranks_seen'], d['parity_checked'], d['config']['frames_per_step'], d['value'], d['ms_per_step'], d['config']['launch_us_per_rank'], d['config'].get('cpu_affinity_rank0')))
PY
