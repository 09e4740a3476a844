#!/bin/bash
# Developer helper for gpurun: smoke + GPU tests (without -x: a failure shows how far it reaches).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -1 gpurun_out/smoke.log
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log | cut -c1-200
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | cut -c1-220 | head -40
