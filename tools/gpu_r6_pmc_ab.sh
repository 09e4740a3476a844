#!/bin/bash
# Round 6: same-box A/B of packed-16-bit variants (tools/bin/*.so, tools/dev/build_variant.sh) and PMC passes for two of them.
# Usage (gpurun): [PMC_LIBS="r6_base r6_pk"] [PMC_CFGS="3 5"] bash tools/gpu_r6_pmc_ab.sh TAG lib1 lib2 ...
cd $GRAFT_REPO_ROOT
TAG=${1:-r06_ab}; shift
mkdir -p gpurun_out
VCFG="${VCFG:-3:8 5:8 3:32 5:32}" ROUNDS=${ROUNDS:-2} bash tools/dev/ab.sh "$@" 2>&1 | tee gpurun_out/${TAG}_ab.log
export VFGS_ALLOW_DEV_BUILD=1
for n in ${PMC_LIBS:-}; do
  export VFGS_LIB=$GRAFT_REPO_ROOT/tools/bin/$n.so
  bash tools/gpu_r5_pmc8.sh ${TAG}_pmc_$n ${PMC_CFGS:-3 5} 2>&1 | tee gpurun_out/${TAG}_pmc_$n.log || exit 1
done
