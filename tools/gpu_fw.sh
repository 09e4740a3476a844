#!/bin/bash
# Developer helper for gpurun: firmware-layer tests, configuration-switch bench, kernel durations.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fw.py -x -q -m gpu > gpurun_out/pytest_fw.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_fw.log; tail -3 gpurun_out/pytest_fw.log | cut -c1-200
timeout -k 10 300 python tools/bench_fw.py > gpurun_out/fw_bench.json 2> gpurun_out/fw_bench.err && cat gpurun_out/fw_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fwprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fwprof -- python3 $GRAFT_REPO_ROOT/tools/bench_fw.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/fw_prof.err
f=$(find /tmp/fwprof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/fw_kernel_stats.csv && cut -c1-160 "$f"
