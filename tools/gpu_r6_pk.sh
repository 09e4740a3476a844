#!/bin/bash
# Round 6: the packed 16-bit form of the 8-bit one-pattern kernels.  d16 probe, smoke + GPU tests, then the same-box A/B of
# prebuilt libraries (tools/bin/r6_base.so = round 5's byte bank + dword LUT, built with -DVFGS_NO_PK16; r6_pk.so = the tree).
# Usage (gpurun): bash tools/gpu_r6_pk.sh [TAG] [lib names...]
cd $GRAFT_REPO_ROOT
TAG=${1:-r06_pk}; shift
LIBS=${@:-r6_base r6_pk}
mkdir -p gpurun_out
[ -x tools/bin/d16_probe ] && tools/bin/d16_probe > gpurun_out/${TAG}_d16_probe.log 2>&1; cat gpurun_out/${TAG}_d16_probe.log
if [ -z "$SKIP_TESTS" ]; then bash tools/gpu_check.sh || exit 1; fi
VCFG="${VCFG:-3:8 5:8 3:32 5:32 6:8 14:8}" ROUNDS=${ROUNDS:-2} bash tools/dev/ab.sh $LIBS 2>&1 | tee gpurun_out/${TAG}_ab.log
