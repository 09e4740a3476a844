#!/bin/bash
# Round 6 (gpurun): host cost per call at the rank shapes of a stripe split, with the stripe stream reached by jumps (default) and with the
# contiguous window (VFGS_HIP_STRIPE_JUMP=0), C caller and ctypes; the pinned-allocation probe for the look-ahead ring.
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
mkdir -p gpurun_out tools/bin
hipcc -O2 -w -o tools/bin/host_call_bench tools/host_call_bench.cpp -Iinclude -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd || exit 1
for rep in 1 2; do
  echo "# jump (run $rep)"; tools/bin/host_call_bench parts 2>&1 | grep -v amdgpu.ids
  echo "# contiguous window, VFGS_HIP_STRIPE_JUMP=0 (run $rep)"; VFGS_HIP_STRIPE_JUMP=0 tools/bin/host_call_bench parts 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/${TAG}_host_call_bench_rank_shapes.log
( echo "# jump"; python3 tools/host_overhead.py; echo "# contiguous window, VFGS_HIP_STRIPE_JUMP=0"; VFGS_HIP_STRIPE_JUMP=0 python3 tools/host_overhead.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_host_overhead_weak_and_strong_shapes.log
if [ -x tools/bin/pinned_alloc_probe ]; then for m in 0 1 2 3 4 5 0 1 4; do tools/bin/pinned_alloc_probe $m; done 2>&1 | grep "mode" | tee gpurun_out/${TAG}_pinned_alloc_probe.log; fi
