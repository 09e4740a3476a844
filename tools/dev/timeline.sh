#!/bin/bash
# Wave timeline of the grain kernel (luma waves; s_memrealtime marks in an instrumented copy of the kernel source).
# Build container:  bash tools/dev/timeline.sh build      -> tools/bin/tl.so
# GPU box:          bash tools/dev/timeline.sh [WxHxFRAMES[:trace] ...]
cd "$(dirname "$0")/../.." || exit 1
C=versatilefilmgrain_amd/csrc
if [ "$1" = build ]; then
  python3 tools/dev/make_timeline_src.py > /dev/null && cp tools/dev/vfgs_kernel_timeline.hip.txt $C/_tl.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -DVFGS_DEV_BUILD -DVFGS_FW_TABLES_PATH="\"$PWD/$C/fw_tables.bin\"" \
    -o tools/bin/tl.so $C/_tl.hip $C/vfgs_fw_kernel.hip $C/vfgs_host.cpp $C/vfgs_fw_host.cpp $C/vfgs_cfg_host.cpp
  rc=$?; rm -f $C/_tl.hip; exit $rc
fi
export VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$PWD/tools/bin/tl.so
for shape in "${@:-7680x4320x8}"; do
  TL_SHAPE=${shape%%:*} TL_TRACE=$( [ "$shape" != "${shape#*:}" ] && echo ${shape#*:} || echo fgs_sei_10_420 ) python3 tools/dev/timeline.py 2>&1 | grep -v amdgpu.ids
done
