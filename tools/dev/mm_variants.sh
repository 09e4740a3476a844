#!/bin/bash
# usage: bash tools/dev/mm_variants.sh "<trace> <w> <h>" "<flags 1>" "<flags 2>" ...   (GPU box; developer helper)
cd $GRAFT_REPO_ROOT
export VFGS_ALLOW_DEV_BUILD=1      # (the variants are developer builds: versatilefilmgrain_amd.hw refuses them otherwise)
C=versatilefilmgrain_amd/csrc
args=$1; shift
i=0
for flags in "$@"; do
  out=/tmp/libvfgs_mm$i.so
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -DVFGS_DEV_BUILD $flags -DVFGS_FW_TABLES_PATH="\"$GRAFT_REPO_ROOT/$C/fw_tables.bin\"" -o $out $C/vfgs_kernel.hip $C/vfgs_fw_kernel.hip $C/vfgs_host.cpp $C/vfgs_fw_host.cpp $C/vfgs_cfg_host.cpp || exit 1
  echo "== variant $i: $flags"
  for rep in 1 2; do VFGS_LIB=$out timeout -k 10 120 python3 tools/dev/mismatch.py $args 2>&1 | grep -v amdgpu.ids | grep "mismatches\|rows \["; done
  i=$((i+1))
done
