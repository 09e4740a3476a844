#!/bin/bash
# (belongs to tools/dev/r05_result_table_kernels.patch: needs a library built from the patched sources and its --tables option of bench_config.py)
# Developer helper for gpurun: the result-table kernels of prebuilt variants (tools/bin/*.so), interleaved on ONE box:
# arithmetic kernels (--tables 0) vs table kernels with sequential phases / classes of workgroups.
# usage: [VCFG="3:8 5:8"] [ROUNDS=2] bash tools/dev/ab_tbl.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
run() { # lib tables classes cfg batch label
  VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$GRAFT_REPO_ROOT/tools/bin/$1.so VFGS_HIP_TBL_CLASSES=$3 python3 tools/bench_config.py --config $4 --batch $5 --tables $2 --steps ${STEPS:-100} ${EXTRA:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('round $round  %-34s cfg %d x%-2d %9.2f us/launch %8.3f us/frame  %.4f  %s' % ('$6', d['config'], d['frames_per_launch'], d['launch_us'], d['us_per_frame'], d['frac_of_8TBps'], d['kernel'][:40]))"
}
for round in $(seq 1 ${ROUNDS:-2}); do
  for cb in ${VCFG:-3:8 5:8}; do c=${cb%:*}; b=${cb#*:}
    run $1 0 0 $c $b "$1 arithmetic"
    for n in "$@"; do
      run $n 2 0 $c $b "$n tables, phases"
      run $n 2 1 $c $b "$n tables, classes"
    done
  done
done
