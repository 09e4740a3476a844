#!/bin/bash
# Round 5: PMC passes (separate from the kernel trace, MI355X_MICROARCH.md "rocprofv3 PMC slots") for the 8-bit kernels at the
# shipped source: BASELINE config 3 (2160p 8-bit 4:4:4 AFGS1), 8-bit 4:2:0 AFGS1 (config 5) and 8-bit 4:2:0 fgs_sei (config 6,
# general form), 8 frames per launch -- the same counter set as profiles/r04_profile_summary.json took for the headline kernel.
# Usage (gpurun): bash tools/gpu_r5_pmc8.sh [TAG] [configs...]; summary -> gpurun_out/${TAG}_pmc.json
cd $GRAFT_REPO_ROOT
TAG=${1:-r05_pmc_8bit}; shift
CFGS=${@:-3 5 6}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $R/pmc8_*
cd /tmp
for c in $CFGS; do
  B="python3 $GRAFT_REPO_ROOT/tools/bench_config.py --config $c --batch ${BATCH:-8} ${EXTRA:-}"
  echo "config $c: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/pmc8_${c}_kt -- $B --steps 200 > $R/pmc8_${c}_kt.log 2>&1 || exit 1
  S="--steps 8 --preroll-ms 20"
  echo "config $c: pmc sq"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/pmc8_${c}_sq -- $B $S > $R/pmc8_${c}_sq.log 2>&1 || exit 1
  echo "config $c: pmc lds"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/pmc8_${c}_lds -- $B $S > $R/pmc8_${c}_lds.log 2>&1 || exit 1
  # (names that may not exist on this ROCm: a pass of their own, allowed to fail)
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/pmc8_${c}_x -- $B $S > $R/pmc8_${c}_x.log 2>&1 || echo "   (extra pass failed: see pmc8_${c}_x.log)"
  echo "config $c: pmc fetch / write"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc8_${c}_fetch -- $B $S > $R/pmc8_${c}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc8_${c}_write -- $B $S > $R/pmc8_${c}_write.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT
python3 - "$TAG" $CFGS <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
import bench
tag, cfgs = sys.argv[1], sys.argv[2:]
out = {"kernel_sha16": bench.kernel_sha(), "frames_per_launch": int(__import__("os").environ.get("BATCH", "8")), "configs": {}}
for c in cfgs:
    o = {}
    for l in open(f'gpurun_out/pmc8_{c}_kt.log'):
        if l.startswith('{'):
            o['bench_line'] = json.loads(l)
    for f in glob.glob(f'gpurun_out/pmc8_{c}_kt/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'grain_' in r['Name'] and 'kernel' in r['Name']:
                o['kernel_stats'] = {k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'StdDev')}
    for d in ('sq', 'lds', 'x', 'fetch', 'write'):
        for f in glob.glob(f'gpurun_out/pmc8_{c}_{d}/**/*counter_collection.csv', recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'grain_' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
            for k, v in acc.items():
                o.setdefault('pmc_mean_per_launch', {})[k] = sum(v) / len(v)
    pm = o.get('pmc_mean_per_launch', {})
    if 'kernel_stats' in o and pm:
        us = float(o['kernel_stats']['AverageNs']) / 1e3
        d = {}
        # SQ_* cycle counters count quad-cycles summed over waves (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*); SQ_BUSY_CYCLES per SE-ish
        # units; GRBM_GUI_ACTIVE summed over the 8 XCDs.  Derived figures (same arithmetic as VERDICT r04 used for the headline):
        if 'GRBM_GUI_ACTIVE' in pm:
            cyc = pm['GRBM_GUI_ACTIVE'] / 8
            d['kernel_cycles_per_xcd'] = cyc
            if 'SQ_INSTS_VALU' in pm:
                d['valu_busy_frac_at_4_cycles_per_inst'] = pm['SQ_INSTS_VALU'] * 4 / (1024 * cyc)
            if 'SQ_LDS_IDX_ACTIVE' in pm:
                d['lds_active_frac_of_cu_cycles'] = pm['SQ_LDS_IDX_ACTIVE'] / (256 * cyc)
        if 'SQ_LDS_IDX_ACTIVE' in pm and 'SQ_LDS_BANK_CONFLICT' in pm:
            d['lds_conflict_share'] = pm['SQ_LDS_BANK_CONFLICT'] / pm['SQ_LDS_IDX_ACTIVE']
        if 'SQ_WAVE_CYCLES' in pm:
            for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU'):
                if k in pm:
                    d[k.lower() + '_share_of_wave_cycles'] = pm[k] / pm['SQ_WAVE_CYCLES']
        if 'SQ_WAIT_INST_LDS' in pm and 'SQ_WAVE_CYCLES' in pm:
            d['sq_wait_inst_lds_share_of_wave_cycles'] = pm['SQ_WAIT_INST_LDS'] / pm['SQ_WAVE_CYCLES']
        if 'FETCH_SIZE' in pm and 'WRITE_SIZE' in pm and 'bench_line' in o:
            alg = o['bench_line']["algorithmic_bytes_per_frame"] * int(__import__("os").environ.get("BATCH", "8"))
            tr = pm['FETCH_SIZE'] * 1024 * 2 + pm['WRITE_SIZE'] * 1024      # gfx950: FETCH_SIZE x2 for 16 B/lane streaming reads, KiB units
            d['hbm_traffic_bytes_per_launch'] = tr
            d['traffic_over_algorithmic'] = tr / alg
        d['frac_of_8TBps_from_rocprof'] = o['bench_line']["algorithmic_bytes_per_frame"] * int(__import__("os").environ.get("BATCH", "8")) / (us * 1e-6) / 8e12 if 'bench_line' in o else None
        o['derived'] = d
    out['configs'][c] = o
json.dump(out, open(f'gpurun_out/{tag}.json', 'w'), indent=1)
for c, o in out['configs'].items():
    print(c, o.get('kernel_stats', {}).get('Name', '?')[:80], o.get('kernel_stats', {}).get('AverageNs'))
    print('   ', json.dumps(o.get('derived', {})))
PY
