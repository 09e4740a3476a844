#!/bin/bash
# Round 6 (gpurun): short runs of the unchanged reference CLI (oracle/_ref/vfgs_hip_cli = vfgs_main.c + vfgs_fw.c + yuv.c over this library)
# with and without VFGS_HIP_FRAME_HEIGHT, the two INTERLEAVED (round 5 read +80-100 ms for the promise out of runs in a fixed order).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 - <<'PY' | tee gpurun_out/${1:-r06}_cli_short_runs.log
import sys, subprocess, time, os
sys.path.insert(0,'tests')
import vfgs_testlib as T
for (w,h,n) in ((1920,1080,10),(3840,2160,6),(7680,4320,3)):
    frames,_=T.lcg_frames(w,h,10,2,2,2)
    inp='/dev/shm/in.yuv'
    with open(inp,'wb') as f:
        for i in range(n): f.write(frames[i%2].picture_bytes())
    def run(exe, env=None, tag=''):
        t0=time.perf_counter()
        subprocess.run([str(T.REF_DIR/exe),'-w',str(w),'-h',str(h),'-b','10','-n',str(n),'-r','12345',inp,'/dev/shm/out_%s%s.yuv'%(exe,tag)],check=True,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL,env=dict(os.environ,**(env or {})))
        return time.perf_counter()-t0
    run('vfgs_hip_cli'); run('vfgs_ref')
    P={'VFGS_HIP_FRAME_HEIGHT':str(h)}
    for rep in range(3):
        a=run('vfgs_hip_cli'); b=run('vfgs_hip_cli',P,'_p'); c=run('vfgs_hip_cli'); d=run('vfgs_hip_cli',P,'_p'); r=run('vfgs_ref'); e=run('vfgs_hip_cli',{'VFGS_HIP_LINE_LOOKAHEAD':'0'},'_n')
        print('%dx%d x%d: reference %.3f s   library %.3f %.3f s   library + VFGS_HIP_FRAME_HEIGHT %.3f %.3f s   library, no look-ahead %.3f s' % (w,h,n,r,a,c,b,d,e), flush=True)
    ref=open('/dev/shm/out_vfgs_ref.yuv','rb').read()
    same = all(ref == open('/dev/shm/out_vfgs_hip_cli%s.yuv'%t,'rb').read() for t in ('','_p','_n'))
    print('   outputs byte-identical (plain, promised, no look-ahead):', same, flush=True)
    for t in ('vfgs_ref','vfgs_hip_cli','vfgs_hip_cli_p','vfgs_hip_cli_n'): os.unlink('/dev/shm/out_%s.yuv'%t)
os.unlink('/dev/shm/in.yuv')
PY
