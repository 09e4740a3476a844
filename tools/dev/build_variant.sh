#!/bin/bash
# Developer helper (runs in the build container): build a library variant with extra -D flags into tools/bin/<name>.so
# usage: [SRC=<dir with a copy of csrc/, e.g. from make_ablation_src.py>] tools/dev/build_variant.sh <name> [flags...]
cd "$(dirname "$0")/../.." || exit 1
C=${SRC:-versatilefilmgrain_amd/csrc}
name=$1; shift
mkdir -p tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -w -DVFGS_DEV_BUILD "$@" -DVFGS_FW_TABLES_PATH="\"$PWD/versatilefilmgrain_amd/csrc/fw_tables.bin\"" \
  -o tools/bin/$name.so $C/vfgs_kernel.hip $C/vfgs_fw_kernel.hip $C/vfgs_host.cpp $C/vfgs_fw_host.cpp $C/vfgs_cfg_host.cpp && echo built tools/bin/$name.so
