#!/bin/bash
# Developer helper (runs in the build container): build a library variant with extra -D flags into tools/bin/<name>.so -- the product's
# translation units (versatilefilmgrain_amd/build.py: the grain kernels once per sample depth, one code object each), compiled side by side.
# usage: [SRC=<dir with a copy of csrc/, e.g. from make_ablation_src.py>] tools/dev/build_variant.sh <name> [flags...]
cd "$(dirname "$0")/../.." || exit 1
C=${SRC:-versatilefilmgrain_amd/csrc}
name=$1; shift
mkdir -p tools/bin
T=$(mktemp -d /tmp/vfgs_variant_XXXXXX)
CC=(/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -w -DVFGS_DEV_BUILD "-DVFGS_FW_TABLES_PATH=\"$PWD/versatilefilmgrain_amd/csrc/fw_tables.bin\"" "$@")
pids=()
"${CC[@]}" -DVFGS_KERNEL_DEPTH=10 -c $C/vfgs_kernel.hip -o $T/k10.o & pids+=($!)
"${CC[@]}" -DVFGS_KERNEL_DEPTH=8 -c $C/vfgs_kernel.hip -o $T/k8.o & pids+=($!)
for f in vfgs_fw_kernel.hip vfgs_host.cpp vfgs_fw_host.cpp vfgs_cfg_host.cpp; do "${CC[@]}" -c $C/$f -o $T/${f%.*}.o & pids+=($!); done
ok=1; for p in "${pids[@]}"; do wait $p || ok=0; done
[ $ok = 1 ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/$name.so $T/*.o && echo built tools/bin/$name.so
rm -rf $T
