cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, subprocess, time, os
sys.path.insert(0,'tests')
import vfgs_testlib as T
w,h=7680,4320
frames,_=T.lcg_frames(w,h,10,2,2,2)
inp='/dev/shm/in.yuv'
with open(inp,'wb') as f:
    for i in range(6): f.write(frames[i%2].picture_bytes())
cli=str(T.REF_DIR/'vfgs_hip_cli')
def run(n, env=None):
    t0=time.perf_counter()
    subprocess.run([cli,'-w',str(w),'-h',str(h),'-b','10','-n',str(n),'-r','12345',inp,'/dev/shm/out.yuv'],check=True,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL,env=dict(os.environ,**(env or {})))
    return time.perf_counter()-t0
run(1)
for n in (1,2,3,4,6):
    print('frames',n,'plain %.2f s'%run(n),' promised %.2f s'%run(n,{'VFGS_HIP_FRAME_HEIGHT':str(h)}),' no lookahead %.2f s'%(run(n,{'VFGS_HIP_LINE_LOOKAHEAD':'0'}) if n<=2 else -1))
PY
