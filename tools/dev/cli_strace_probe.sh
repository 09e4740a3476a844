#!/bin/bash
# GPU box: where does the unchanged CLI's first (line by line) frame spend its time?  syscall profile of one 1080p frame.
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys
sys.path.insert(0,'tests')
import vfgs_testlib as T
frames,_=T.lcg_frames(1920,1080,10,2,2,2)
with open('/dev/shm/in.yuv','wb') as f:
    for i in range(4): f.write(frames[i%2].picture_bytes())
PY
CLI=oracle/_ref/vfgs_hip_cli
which strace ltrace perf 2>&1 | head -3
for n in 1 2 3; do echo "frames $n:"; time $CLI -w 1920 -h 1080 -b 10 -n $n -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null; done
echo "no lookahead, frames 1:"; export VFGS_HIP_LINE_LOOKAHEAD=0; time $CLI -w 1920 -h 1080 -b 10 -n 1 -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null
echo "no lookahead, frames 3:"; time $CLI -w 1920 -h 1080 -b 10 -n 3 -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null
unset VFGS_HIP_LINE_LOOKAHEAD
if which strace > /dev/null 2>&1; then strace -f -c -o /tmp/st.txt $CLI -w 1920 -h 1080 -b 10 -n 1 -r 12345 /dev/shm/in.yuv /dev/shm/out.yuv > /dev/null 2>&1; head -15 /tmp/st.txt; fi
