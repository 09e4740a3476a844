#!/bin/bash
# Developer helper for gpurun: per-config kernel statistics (rocprofv3 --kernel-trace --stats, program directly after --),
# the config matrix, the host round-trip experiment.  Everything lands in gpurun_out/r03_*; copy what is to be judged to profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/bin
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $R/cfgprof_*
: > $R/r03_config_lines.jsonl
for c in 0 1 2 3 4 5 6; do
  for b in 1 8; do
    python3 tools/bench_config.py --config $c --batch $b >> $R/r03_config_lines.jsonl 2>> $R/r03_config.err
  done
  python3 tools/bench_config.py --config $c --batch 1 --steps 400 --overlap >> $R/r03_config_lines.jsonl 2>> $R/r03_config.err     # one frame per call inside an overlap region
  python3 tools/bench_config.py --config $c --batch 8 --overlap >> $R/r03_config_lines.jsonl 2>> $R/r03_config.err                 # 8 frames per call inside an overlap region
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/cfgprof_$c -- python3 $GRAFT_REPO_ROOT/tools/bench_config.py --config $c --batch 8 > $R/cfgprof_$c.log 2>&1 )
done
cat $R/r03_config_lines.jsonl
python3 - <<'PY'
import csv, glob, json
rows = []
for c in range(7):
    for f in glob.glob(f'gpurun_out/cfgprof_{c}/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'grain_' in r['Name'] and 'kernel' in r['Name']:
                rows.append({'config': c, **{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'StdDev')}})
json.dump(rows, open('gpurun_out/r03_config_kernel_stats.json', 'w'), indent=1)
print(json.dumps(rows, indent=1))
PY
python3 tools/host_pipeline_bench.py > $R/r03_host_pipeline.jsonl 2>/dev/null; python3 tools/host_pipeline_bench.py --devices 0,0 >> $R/r03_host_pipeline.jsonl 2>/dev/null; cat $R/r03_host_pipeline.jsonl

# the same one-frame-per-call pattern from a C caller (no ctypes in the way): host us per call, device us per frame
( for g in "1920 1080 4000" "3840 2160 2000" "7680 4320 600"; do timeout -k 10 120 tools/bin/host_call_bench $g; done ) > $R/r03_host_call_bench.jsonl 2>&1; cat $R/r03_host_call_bench.jsonl
