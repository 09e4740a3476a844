#!/bin/bash
# Round 6: where do the 80-100 ms go that VFGS_HIP_FRAME_HEIGHT adds to a SHORT run of the unchanged CLI (profiles/r05_cli_short_runs.log)?
# The LD_PRELOAD shim (tools/dev/line_time_shim.c) times every vfgs_add_grain_line of the unchanged reference program.
cd $GRAFT_REPO_ROOT
W=${W:-1920}; H=${H:-1080}; N=${N:-10}
python3 - <<PY
import sys
sys.path.insert(0,'tests')
import vfgs_testlib as T
frames,_=T.lcg_frames($W,$H,10,2,2,3)
with open('/dev/shm/in.yuv','wb') as f:
    for i in range($N): f.write(frames[i%3].picture_bytes())
PY
CLI=oracle/_ref/vfgs_hip_cli
for r in 1 2 3; do
echo "== plain (run $r)"; ( time LD_PRELOAD=$PWD/tools/bin/line_time_shim.so $CLI -w $W -h $H -b 10 -n $N -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null ) 2>&1 | grep -v "^$\|user\|sys"
md5sum /dev/shm/out.yuv
echo "== promised (run $r)"; ( time VFGS_HIP_FRAME_HEIGHT=$H LD_PRELOAD=$PWD/tools/bin/line_time_shim.so $CLI -w $W -h $H -b 10 -n $N -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null ) 2>&1 | grep -v "^$\|user\|sys"
md5sum /dev/shm/out.yuv
done
rm -f /dev/shm/in.yuv /dev/shm/out.yuv
