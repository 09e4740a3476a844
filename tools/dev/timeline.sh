#!/bin/bash
# GPU box: build the library with the instrumented kernel source and print the average wave timeline.  usage: bash tools/dev/timeline.sh "<extra flags>" ...
cd $GRAFT_REPO_ROOT
export VFGS_ALLOW_DEV_BUILD=1      # (the variants are developer builds: versatilefilmgrain_amd.hw refuses them otherwise)
C=versatilefilmgrain_amd/csrc
python3 tools/dev/make_timeline_src.py > /dev/null; cp tools/dev/vfgs_kernel_timeline.hip.txt $C/_tl.hip
for flags in "$@"; do
  echo "== flags: $flags"
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -DVFGS_DEV_BUILD $flags -DVFGS_FW_TABLES_PATH="\"$GRAFT_REPO_ROOT/$C/fw_tables.bin\"" -o /tmp/libvfgs_tl.so $C/_tl.hip $C/vfgs_fw_kernel.hip $C/vfgs_host.cpp $C/vfgs_fw_host.cpp $C/vfgs_cfg_host.cpp || exit 1
  VFGS_LIB=/tmp/libvfgs_tl.so python3 tools/dev/timeline.py 2>&1 | grep -v amdgpu.ids
done
rm -f $C/_tl.hip
