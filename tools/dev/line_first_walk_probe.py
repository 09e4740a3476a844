#!/usr/bin/env python3
"""Developer probe (GPU box): what does ONE vfgs_add_grain_line round trip cost in a fresh process without torch -- the
situation of the unchanged reference CLI -- frame after frame, with the look-ahead switched off?"""
import ctypes as C
import os
import sys
import time
from pathlib import Path

os.environ["VFGS_HIP_NO_TORCH_RUNTIME"] = "1"
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402

w, hh = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
look = int(sys.argv[2]) if len(sys.argv) > 2 else 0
h = hw.VfgsHip(device=0)
rec = T.load_trace("fgs_sei_10_420")
T.replay(h, rec)
h.line_lookahead(look)
olib = T.oracle_lib()
fn = C.cast(h.lib.vfgs_add_grain_line, C.c_void_p)
f = T.lcg_frames(w, hh, 10, 2, 2, 1)[0][0]
for i in range(4):
    t0 = time.perf_counter()
    olib.vfgs_oracle_drive_lines(fn, C.c_void_p(f.Y.ctypes.data), C.c_void_p(f.U.ctypes.data), C.c_void_p(f.V.ctypes.data), w, hh, f.stride, f.cstride, 2, 2)
    dt = time.perf_counter() - t0
    print(f"{w}x{hh} lookahead {look}: walk {i}: {dt * 1e3:8.1f} ms = {dt / hh * 1e6:6.1f} us per line", flush=True)
