#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
P() { echo "pass $1"; timeout -k 5 120 rocprofv3 --pmc $2 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mem_$1 -- ${@:3} > $GRAFT_REPO_ROOT/gpurun_out/mem_$1.log 2>&1; echo "rc=$?"; }
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-parity --no-region --steps 4 --warmup 1"
P k1 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" $B
P k2 "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" $B
P k3 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" $B
P k4 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" $B
P k5 "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_REQ_sum TCC_CYCLE_sum" $B
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/mem_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "grain" in r["Kernel_Name"] or "diag_rmw_np_nt" in r["Kernel_Name"]:
                acc[("G " if "grain" in r["Kernel_Name"] else "NT ") + r["Counter_Name"]].append(float(r['Counter_Value']))
        print(d.split('/')[1], {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
