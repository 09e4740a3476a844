#!/bin/bash
# Developer helper for gpurun: per-config kernel statistics (rocprofv3 --kernel-trace --stats, program directly after --) at the
# frames per launch the bench line quotes, plus the one-frame-per-call lines (plain and inside an overlap region) and the
# host-memory paths.  Everything lands in gpurun_out/${ROUND}_*; copy what is to be judged to profiles/.
cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-r05}
mkdir -p gpurun_out tools/bin
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $R/cfgprof_*
: > $R/${ROUND}_config_lines.jsonl
# config:frames per launch[:flag of bench_config.py] -- every entry of bench.py's `configs` (incl. the frame list and the natural-like content)
# and the tracked matrix beside it
for cb in ${CBS:-0:8 0:32 0:32:--list 1:8 1:32 2:8 2:16 3:8 3:16 4:8 4:8:--content=ramp 5:8 6:8 10:8 11:8 12:8 13:2 14:2 15:2}; do
  c=${cb%%:*}; r=${cb#*:}; b=${r%%:*}; fl=${r#*:}; [ "$fl" = "$r" ] && fl=""
  tag=${c}_x$b$(echo "$fl" | tr -c 'a-z0-9' '_' | sed 's/_*$//;s/__*/_/g')
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/cfgprof_$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_config.py --config $c --batch $b $fl > $R/cfgprof_$tag.log 2>&1 )
  grep '^{' $R/cfgprof_$tag.log >> $R/${ROUND}_config_lines.jsonl
done
for c in 0 2 4; do
  python3 tools/bench_config.py --config $c --batch 1 --steps 400 >> $R/${ROUND}_config_lines.jsonl 2>> $R/${ROUND}_config.err
  python3 tools/bench_config.py --config $c --batch 1 --steps 400 --overlap >> $R/${ROUND}_config_lines.jsonl 2>> $R/${ROUND}_config.err     # one frame per call inside an overlap region
done
python3 - <<PY
import csv, glob, json, re, shutil
rows = []
for d in sorted(glob.glob('gpurun_out/cfgprof_*_x*/')):
    m = re.search(r'cfgprof_(\d+)_x(\d+)(_[a-z0-9_]+)?/', d)
    var = (m.group(3) or '')
    for f in glob.glob(d + '**/*kernel_stats.csv', recursive=True):
        shutil.copyfile(f, 'gpurun_out/${ROUND}_config%s_x%s%s_kernel_stats.csv' % (m.group(1), m.group(2), var))
        for r in csv.DictReader(open(f)):
            if 'grain_' in r['Name'] and 'kernel' in r['Name']:
                rows.append({'config': int(m.group(1)), 'frames_per_launch': int(m.group(2)), 'variant': var.strip('_'), **{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'StdDev')}})
json.dump(rows, open('gpurun_out/${ROUND}_config_kernel_stats.json', 'w'), indent=1)
lines = [json.loads(l) for l in open('gpurun_out/${ROUND}_config_lines.jsonl')]
for r in rows:
    ev = [l for l in lines if l['config'] == r['config'] and l['frames_per_launch'] == r['frames_per_launch'] and not l['overlap_region']
          and bool(l.get('frame_list')) == ('list' in r['variant']) and (l['content'] == 'ramp') == ('ramp' in r['variant'])]
    r['hip_events_launch_us_same_run'] = ev[0]['launch_us'] if ev else None
    r['frac_of_8TBps_from_rocprof'] = round(ev[0]['algorithmic_bytes_per_frame'] * r['frames_per_launch'] / float(r['AverageNs']) / 8000, 4) if ev else None
    print('config %2d x%-2d %-12s rocprof avg %9.2f us (%s calls)  HIP events of the same run %9.2f us  frac %s  %s' % (r['config'], r['frames_per_launch'], r['variant'], float(r['AverageNs']) / 1e3, r['Calls'], ev[0]['launch_us'] if ev else -1, r['frac_of_8TBps_from_rocprof'], r['Name'][:60]))
json.dump(rows, open('gpurun_out/${ROUND}_config_kernel_stats.json', 'w'), indent=1)
PY
python3 tools/host_pipeline_bench.py > $R/${ROUND}_host_pipeline.jsonl 2>/dev/null; cat $R/${ROUND}_host_pipeline.jsonl
