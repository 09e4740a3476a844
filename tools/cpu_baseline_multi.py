#!/usr/bin/env python3
"""Secondary CPU baseline of SURVEY 8d: P independent processes, each running the REAL reference
hot path (oracle/_ref/libvfgs_ref.so, line loop of vfgs_main.c:664-682) on its own 4320p frames --
the only way to use more than one core with the reference, whose state is process-global and whose
PRNG is sequential.  Prints aggregate Mpixels/s for P = 1, 4, 16 (the GPU box gives one GPU 16 cores)."""
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))

W, H = 7680, 4320


def worker(rank, frames, q, go):
    import vfgs_testlib as T
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[rank % len(os.sched_getaffinity(0))]})
    except Exception:
        pass
    rec = T.load_trace("fgs_sei_10_420")
    hw = T.ReferenceHW()
    T.replay(hw, rec)
    lib = T.oracle_lib()
    f, _ = T.lcg_frames(W, H, 10, 2, 2, 1)
    fr = f[0]
    line = C.cast(hw.lib.vfgs_add_grain_line, C.c_void_p)
    go.wait()
    t0 = time.perf_counter()
    for _ in range(frames):
        lib.vfgs_oracle_drive_lines(line, C.c_void_p(fr.Y.ctypes.data), C.c_void_p(fr.U.ctypes.data), C.c_void_p(fr.V.ctypes.data),
                                    fr.width, fr.height, fr.stride, fr.cstride, 2, 2)
    q.put(time.perf_counter() - t0)


def main():
    import vfgs_testlib as T
    assert T.have_reference(), "needs the prebuilt oracle/_ref/libvfgs_ref.so"
    T.build_oracle()
    out = {}
    frames = 20
    for p in (1, 4, 16):
        if p > len(os.sched_getaffinity(0)):
            continue
        q, go = mp.Queue(), mp.Event()
        procs = [mp.Process(target=worker, args=(r, frames, q, go)) for r in range(p)]
        for x in procs:
            x.start()
        time.sleep(8)     # every worker has generated its frame by now
        go.set()
        times = [q.get() for _ in procs]
        for x in procs:
            x.join()
        out[p] = {"processes": p, "frames_each": frames, "slowest_s": round(max(times), 2),
                  "aggregate_Mpixels_per_s": round(p * frames * W * H / max(times) / 1e6, 1)}
        print(json.dumps(out[p]), flush=True)


if __name__ == "__main__":
    main()
