"""Developer tool (GPU box): frames in host memory -- the pipelined entry point against one synchronous whole-frame call
per frame.  python3 tools/host_pipeline_bench.py [--frames 12] [--width 7680 --height 4320]"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import vfgs_testlib as T
from versatilefilmgrain_amd import hw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--trace", default="fgs_sei_10_420")
    ap.add_argument("--devices", default="0", help="comma separated device list for vfgs_hip_init_devices (a device may repeat)")
    args = ap.parse_args()
    h = hw.VfgsHip(device=0)
    devs = [int(x) for x in args.devices.split(",")]
    if len(devs) > 1:
        h.init_devices(devs)
    rec = T.load_trace(args.trace)
    T.replay(h, rec)
    depth, sx, sy = T.trace_geometry(rec)
    w, hh, n = args.width, args.height, args.frames
    dt = np.uint16 if depth > 8 else np.uint8
    sz = np.dtype(dt).itemsize
    shapes = [(hh, w), (hh // sy, w // sx), (hh // sy, w // sx)]
    frame_bytes = sum(a * b for a, b in shapes) * sz
    rng = np.random.default_rng(1)
    src = [rng.integers(0, 1 << depth, s, dtype=np.int32).astype(dt) for s in shapes]

    def make(pinned):
        frames, keep = [], []
        for _ in range(n):
            planes = []
            for a in src:
                if pinned:
                    p = h.host_alloc(a.nbytes)
                    keep.append(p)
                    v = np.frombuffer((C.c_char * a.nbytes).from_address(p), dtype=dt).reshape(a.shape)
                    v[...] = a
                else:
                    v = a.copy()
                planes.append(v)
            frames.append(planes)
        return frames, keep

    for pinned in (True, False):
        frames, keep = make(pinned)
        ptr = lambda i: [f[i].ctypes.data for f in frames]
        res = {}
        for mode in ("pipelined", "one synchronous call per frame"):
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                if mode == "pipelined":
                    h.add_grain_frames_host(ptr(0), ptr(1), ptr(2), w, hh, w, w // sx)
                else:
                    for f in frames:
                        h.add_grain_stripe(f[0].ctypes.data, f[1].ctypes.data, f[2].ctypes.data, 0, w, hh, w, w // sx)
                best = min(best, time.perf_counter() - t0)
            res[mode] = best
        for mode, t in res.items():
            print(json.dumps({"devices": devs, "memory": "pinned" if pinned else "pageable", "mode": mode, "frames": n, "geometry": f"{w}x{hh} {depth}-bit {sx}{sy}",
                              "ms_per_frame": round(t / n * 1e3, 3), "GBps_each_way": round(frame_bytes / (t / n) / 1e9, 1),
                              "Mpixels_per_s": round(w * hh / (t / n) / 1e6, 1)}), flush=True)
        del frames
        for p in keep:
            h.host_free(p)


if __name__ == "__main__":
    main()
