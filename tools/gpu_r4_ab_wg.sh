#!/bin/bash
# Developer helper for gpurun (round 4): workgroup bytes vs frames per launch (same box, interleaved), the floor of a tiny launch,
# and the bench line with its new `configs` leg.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
VCFG="0:8 0:32 1:8 1:32 3:8 3:32 5:8 5:32 6:8 6:32" ROUNDS=2 STEPS=100 bash tools/dev/ab.sh base wg48k wg16k
echo "== tiny launches (one frame per call, plain stream): the floor a launch pays"
for hgt in 16 64 256 1080; do VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$GRAFT_REPO_ROOT/tools/bin/base.so python3 tools/bench_config.py --config 0 --batch 1 --steps 400 --height $hgt 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-50s x%d %8.3f us/launch' % (d['workload'], d['frames_per_launch'], d['launch_us']))"; done
} > gpurun_out/r04_ab1_wg_bytes_vs_batch.log 2>&1
cat gpurun_out/r04_ab1_wg_bytes_vs_batch.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err; echo "bench rc $?"; cat gpurun_out/r04_bench_a.json; tail -5 gpurun_out/r04_bench_a.err
