#!/bin/bash
# Round 5 (gpurun): whole-process wall times (the runtime initialises before main() under some settings, so in-process timers miss it)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/bin
hipcc -O2 -w -o tools/bin/startup_probe tools/dev/startup_probe.cpp -Iinclude -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd || exit 1
hipcc -O2 -w -o tools/bin/startup_bare tools/dev/startup_bare.cpp || exit 1
python3 - <<'PY' | tee gpurun_out/r05_startup_wall.log
import subprocess, time, os
def wall(cmd, env=None, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **(env or {}))); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e3, ts[0] * 1e3
for name, cmd, env in (("bare HIP process (malloc + memset + sync)", ["tools/bin/startup_bare"], None),
                       ("library: init + 4 frame launches 1080p", ["tools/bin/startup_probe"], None),
                       ("library, HIP_ENABLE_DEFERRED_LOADING=0", ["tools/bin/startup_probe"], {"HIP_ENABLE_DEFERRED_LOADING": "0"}),
                       ("bare, HIP_ENABLE_DEFERRED_LOADING=0", ["tools/bin/startup_bare"], {"HIP_ENABLE_DEFERRED_LOADING": "0"}),
                       ("/bin/true", ["/bin/true"], None)):
    m, b = wall(cmd, env)
    print("%-50s median %7.1f ms   best %7.1f ms" % (name, m, b))
PY
