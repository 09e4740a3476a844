#!/bin/bash
# Developer helper for gpurun: build library variants with extra -D flags and run the config matrix with each.
# usage: [VCFG="4:8 3:8"] bash tools/gpu_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...      ("" = as shipped; VCFG = config:frames-per-launch list)
cd $GRAFT_REPO_ROOT
C=versatilefilmgrain_amd/csrc
i=0
for flags in "$@"; do
  out=/tmp/libvfgs_var$i.so
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -DVFGS_DEV_BUILD $flags -DVFGS_FW_TABLES_PATH="\"$GRAFT_REPO_ROOT/$C/fw_tables.bin\"" -o $out $C/vfgs_kernel.hip $C/vfgs_fw_kernel.hip $C/vfgs_host.cpp $C/vfgs_fw_host.cpp $C/vfgs_cfg_host.cpp || exit 1
  i=$((i+1))
done
NV=$i
for round in 1 2; do
for i in $(seq 0 $((NV-1))); do
  echo "== variant $i round $round"
  for cb in ${VCFG:-0:8 2:8 3:8 4:1 4:8}; do c=${cb%:*}; b=${cb#*:}; VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=/tmp/libvfgs_var$i.so python3 tools/bench_config.py --config $c --batch $b --steps 100 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  cfg', d['config'], 'batch', d['frames_per_launch'], 'us/frame', d['us_per_frame'], 'frac', d['frac_of_8TBps'])"; done
done
done
