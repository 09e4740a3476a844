#!/bin/bash
# Developer helper for gpurun: the default bench line + rocprofv3 kernel stats + PMC traffic of the SAME command.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq gpurun_out/prof_lds     # (stale passes of earlier calls must not be summarised)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ceiling --no-parity --no-region --no-configs > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ceiling --no-parity --no-region --no-configs --preroll-ms 20 --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_write -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ceiling --no-parity --no-region --no-configs --preroll-ms 20 --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sq -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ceiling --no-parity --no-region --no-configs --preroll-ms 20 --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lds -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ceiling --no-parity --no-region --no-configs --preroll-ms 20 --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_lds.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json, datetime, sys
sys.path.insert(0, '.')
import bench
out = {}
for f in glob.glob('gpurun_out/prof_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'grain_' in r['Name'] and 'kernel' in r['Name']:
            out['kernel_stats'] = {k: r[k] for k in ('Name','Calls','TotalDurationNs','AverageNs','MinNs','MaxNs','StdDev')}
for d in ('prof_fetch','prof_write','prof_sq','prof_lds'):
    for f in glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'grain_' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            out.setdefault('pmc_mean_per_launch', {})[k] = sum(v)/len(v)
pm = out.get('pmc_mean_per_launch', {})
if 'FETCH_SIZE' in pm and 'WRITE_SIZE' in pm:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a
    # wide coalesced streaming read (16 B/lane) -> x2; WRITE_SIZE is exact for 16-B-per-lane streaming stores
    out['hbm_traffic'] = {'read_bytes': pm['FETCH_SIZE']*1024*2, 'write_bytes': pm['WRITE_SIZE']*1024,
                          'bytes_per_launch': pm['FETCH_SIZE']*1024*2 + pm['WRITE_SIZE']*1024, 'batch': 8,
                          'kernel_sha16': bench.kernel_sha(), 'sources': list(bench.PROFILED_SOURCES),
                          'date': datetime.date.today().isoformat(),
                          'correction': 'FETCH_SIZE x2 (gfx950, 16 B/lane streaming reads), WRITE_SIZE x1; KiB units'}
json.dump(out, open('gpurun_out/profile_summary.json','w'), indent=1)
if 'hbm_traffic' in out:
    json.dump(out['hbm_traffic'], open('profiles/hbm_traffic.json','w'), indent=1)     # (the box's copy: the bench lines below then carry the traffic of THIS source)
print(json.dumps(out, indent=1))
PY
# the bench lines last: they read profiles/hbm_traffic.json written above (same kernel source, same box)
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_args.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_driver_args.json
python bench.py > gpurun_out/bench_default.json 2>> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
python bench.py --batch 1 --steps 1000 --warmup 300 --pool 24 --no-cpu --no-configs > gpurun_out/bench_b1.json 2>> gpurun_out/bench_default.err; cat gpurun_out/bench_b1.json
