cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys
sys.path.insert(0,'tests')
import vfgs_testlib as T
frames,_=T.lcg_frames(7680,4320,10,2,2,2)
with open('/dev/shm/in.yuv','wb') as f:
    for i in range(4): f.write(frames[i%2].picture_bytes())
PY
CLI=oracle/_ref/vfgs_hip_cli
echo "== plain"; time LD_PRELOAD=$PWD/tools/bin/line_time_shim.so $CLI -w 7680 -h 4320 -b 10 -n 3 -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null
echo "== promised"; time VFGS_HIP_FRAME_HEIGHT=4320 LD_PRELOAD=$PWD/tools/bin/line_time_shim.so $CLI -w 7680 -h 4320 -b 10 -n 3 -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null
