#!/bin/bash
# Developer helper for gpurun: per-config kernel statistics (rocprofv3 --kernel-trace --stats, program directly after --) at the
# frames per launch the bench line quotes, plus the one-frame-per-call lines (plain and inside an overlap region) and the
# host-memory paths.  Everything lands in gpurun_out/${ROUND}_*; copy what is to be judged to profiles/.
cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-r04}
mkdir -p gpurun_out tools/bin
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $R/cfgprof_*
: > $R/${ROUND}_config_lines.jsonl
for cb in 0:8 0:32 1:8 1:32 2:8 3:8 4:8 5:8 6:8 10:8; do
  c=${cb%:*}; b=${cb#*:}
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/cfgprof_${c}_x$b -- python3 $GRAFT_REPO_ROOT/tools/bench_config.py --config $c --batch $b > $R/cfgprof_${c}_x$b.log 2>&1 )
  grep '^{' $R/cfgprof_${c}_x$b.log >> $R/${ROUND}_config_lines.jsonl
done
for c in 0 2 4; do
  python3 tools/bench_config.py --config $c --batch 1 --steps 400 >> $R/${ROUND}_config_lines.jsonl 2>> $R/${ROUND}_config.err
  python3 tools/bench_config.py --config $c --batch 1 --steps 400 --overlap >> $R/${ROUND}_config_lines.jsonl 2>> $R/${ROUND}_config.err     # one frame per call inside an overlap region
done
python3 - <<PY
import csv, glob, json, re, shutil
rows = []
for d in sorted(glob.glob('gpurun_out/cfgprof_*_x*/')):
    m = re.search(r'cfgprof_(\d+)_x(\d+)', d)
    for f in glob.glob(d + '**/*kernel_stats.csv', recursive=True):
        shutil.copyfile(f, 'gpurun_out/${ROUND}_config%s_x%s_kernel_stats.csv' % (m.group(1), m.group(2)))
        for r in csv.DictReader(open(f)):
            if 'grain_' in r['Name'] and 'kernel' in r['Name']:
                rows.append({'config': int(m.group(1)), 'frames_per_launch': int(m.group(2)), **{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'StdDev')}})
json.dump(rows, open('gpurun_out/${ROUND}_config_kernel_stats.json', 'w'), indent=1)
lines = [json.loads(l) for l in open('gpurun_out/${ROUND}_config_lines.jsonl')]
for r in rows:
    ev = [l for l in lines if l['config'] == r['config'] and l['frames_per_launch'] == r['frames_per_launch'] and not l['overlap_region']]
    print('config %2d x%-2d rocprof avg %9.2f us (%s calls)  HIP events of the same run %9.2f us  %s' % (r['config'], r['frames_per_launch'], float(r['AverageNs']) / 1e3, r['Calls'], ev[0]['launch_us'] if ev else -1, r['Name'][:70]))
PY
python3 tools/host_pipeline_bench.py > $R/${ROUND}_host_pipeline.jsonl 2>/dev/null; cat $R/${ROUND}_host_pipeline.jsonl
