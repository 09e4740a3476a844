#!/usr/bin/env python3
"""Developer tool: the UNCHANGED caller's view -- the reference's per-line frame loop (vfgs_main.c:664-682) over frames that live
in ordinary host memory, driven through vfgs_add_grain_line of (a) libvfgs_hip.so and (b) the real reference hardware layer
(oracle/_ref/libvfgs_ref.so), the same C loop for both (oracle: vfgs_oracle_drive_lines), no file I/O.  Every frame of the
library run is compared with the reference run (bit-exact or the tool fails).  One JSON line per size.

The first walk through a buffer is computed line by line by design (the library reads ahead only inside rows the caller has
proven to own, include/vfgs_hip.h); `--declare` makes the promise up front with vfgs_hip_declare_frame().
"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime per process: torch's first)
import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1920x1080,3840x2160,7680x4320")
    ap.add_argument("--trace", default="fgs_sei_10_420")
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--declare", action="store_true")
    args = ap.parse_args()
    rec = T.load_trace(args.trace)
    depth, sx, sy = T.trace_geometry(rec)
    olib = T.oracle_lib()
    import os
    if os.environ.get("VFGS_LIB"):
        hw.load(os.environ["VFGS_LIB"])      # a variant build (tools/dev/build_variant.sh)
    h = hw.VfgsHip(device=0)
    hip_line = C.cast(h.lib.vfgs_add_grain_line, C.c_void_p)
    for size in args.sizes.split(","):
        w, hh = (int(x) for x in size.split("x"))
        h.lib.vfgs_hip_reset_state()
        T.replay(h, rec)
        ref = T.ReferenceHW() if T.have_reference() else None
        if ref:
            T.replay(ref, rec)
            ref_line = C.cast(ref.lib.vfgs_add_grain_line, C.c_void_p)
        src, _ = T.lcg_frames(w, hh, depth, sx, sy, 2)          # two different pictures, alternating, in ONE buffer
        buf = src[0].copy()                                     # (the CLI reads every frame into the same buffer, yuv.c:162-186)
        want = src[0].copy()
        sz = 2 if depth > 8 else 1
        if args.declare:
            h.declare_frame(buf.Y.ctypes.data, buf.U.ctypes.data, buf.V.ctypes.data, w, hh, buf.stride, buf.cstride)

        def drive(fn, fr):
            olib.vfgs_oracle_drive_lines(fn, C.c_void_p(fr.Y.ctypes.data), C.c_void_p(fr.U.ctypes.data), C.c_void_p(fr.V.ctypes.data),
                                         fr.width, fr.height, fr.stride, fr.cstride, sz, sy)
        t_hip, t_ref, ok = [], [], True
        for i in range(args.frames):
            for p, q, r in zip(buf.planes(), want.planes(), src[i % 2].planes()):
                np.copyto(p, r)
                np.copyto(q, r)
            t0 = time.perf_counter()
            drive(hip_line, buf)
            t_hip.append(time.perf_counter() - t0)
            if ref:
                t0 = time.perf_counter()
                drive(ref_line, want)
                t_ref.append(time.perf_counter() - t0)
                ok = ok and buf.equal_all(want)
        steady = sorted(t_hip[2:])[len(t_hip[2:]) // 2] if len(t_hip) > 2 else t_hip[-1]
        out = {"size": size, "trace": args.trace, "declared": args.declare, "frames": args.frames,
               "hip_first_frame_ms": round(t_hip[0] * 1e3, 2), "hip_second_frame_ms": round(t_hip[1] * 1e3, 2) if len(t_hip) > 1 else None,
               "hip_steady_ms_per_frame": round(steady * 1e3, 3), "hip_steady_frames_per_s": round(1 / steady, 1),
               "hip_steady_host_GBps_each_way": round(sz * (w * hh + 2 * (w // sx) * (hh // sy)) / steady / 1e9, 2)}
        if ref:
            rs = sorted(t_ref)[len(t_ref) // 2]
            out.update({"reference_ms_per_frame": round(rs * 1e3, 2), "reference_frames_per_s": round(1 / rs, 2), "speedup_steady": round(rs / steady, 1),
                        "bit_exact_vs_reference": bool(ok)})
        print(json.dumps(out), flush=True)
        if ref and not ok:
            sys.exit(1)


if __name__ == "__main__":
    main()
