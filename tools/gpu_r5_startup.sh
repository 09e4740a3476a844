#!/bin/bash
# Round 5 (gpurun): where a short run of the unchanged CLI spends its time: process start split into runtime init / library init / first
# launch (tools/dev/startup_probe.cpp, several fresh processes), then the CLI itself on 10 frames of 1080p (reference binary vs the
# library behind the same main, with and without the frame-height promise).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/bin
hipcc -O2 -w -o tools/bin/startup_probe tools/dev/startup_probe.cpp -Iinclude -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd || exit 1
for i in 1 2 3 4; do tools/bin/startup_probe 2>/dev/null; done | tee gpurun_out/r05_startup_probe.jsonl
for i in 1 2; do HIP_ENABLE_DEFERRED_LOADING=0 tools/bin/startup_probe 2>/dev/null | sed 's/^{/{"deferred_loading": 0, /'; done | tee -a gpurun_out/r05_startup_probe.jsonl
python3 - <<'PY' | tee gpurun_out/r05_cli_short_runs.log
import sys, subprocess, time, os
sys.path.insert(0,'tests')
import vfgs_testlib as T
for (w,h,n) in ((1920,1080,10),(3840,2160,6)):
    frames,_=T.lcg_frames(w,h,10,2,2,2)
    inp='/dev/shm/in.yuv'
    with open(inp,'wb') as f:
        for i in range(n): f.write(frames[i%2].picture_bytes())
    def run(exe, env=None):
        t0=time.perf_counter()
        subprocess.run([str(T.REF_DIR/exe),'-w',str(w),'-h',str(h),'-b','10','-n',str(n),'-r','12345',inp,'/dev/shm/out_%s.yuv'%exe],check=True,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL,env=dict(os.environ,**(env or {})))
        return time.perf_counter()-t0
    run('vfgs_hip_cli'); run('vfgs_ref')
    for rep in range(3):
        print('%dx%d x%d: reference %.3f s   library %.3f s   library + VFGS_HIP_FRAME_HEIGHT %.3f s   library, no look-ahead %.3f s' % (w,h,n,run('vfgs_ref'),run('vfgs_hip_cli'),run('vfgs_hip_cli',{'VFGS_HIP_FRAME_HEIGHT':str(h)}),run('vfgs_hip_cli',{'VFGS_HIP_LINE_LOOKAHEAD':'0'})))
    same = open('/dev/shm/out_vfgs_ref.yuv','rb').read() == open('/dev/shm/out_vfgs_hip_cli.yuv','rb').read()
    print('   outputs byte-identical:', same)
PY
