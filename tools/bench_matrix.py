#!/usr/bin/env python3
"""Developer tool: the five BASELINE.json configurations (+ natural-like content, + the host-pointer
stripe path incl. PCIe) on one MI355X.  Numbers go into DESIGN.md; bench.py stays the contract line."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402

CONFIGS = [  # name, w, h, depth, (subx, suby), trace
    ("1080p 10b 420 fgs_sei", 1920, 1080, 10, (2, 2), "fgs_sei_10_420"),
    ("1080p 10b 420 ff_test1", 1920, 1080, 10, (2, 2), "fgs_sei_ff_test1_10_420"),
    ("2160p 10b 420 ar_test1", 3840, 2160, 10, (2, 2), "fgs_sei_ar_test1_10_420"),
    ("2160p 8b 444 afgs1_test1", 3840, 2160, 8, (1, 1), "fgs_afgs1_test1_8_444"),
    ("4320p 10b 420 fgs_sei", 7680, 4320, 10, (2, 2), "fgs_sei_10_420"),
]


def planes(w, h, depth, sx, sy, n, natural=False):
    dt = torch.int16 if depth > 8 else torch.uint8
    mx = 1 << depth
    g = torch.Generator(device="cuda").manual_seed(3)

    def mk(hh, ww):
        if natural:   # smooth ramp + +-4 LSB noise (SURVEY 8d "natural-like")
            ramp = (torch.arange(ww, device="cuda").float()[None, :] / ww * 0.6 + torch.arange(hh, device="cuda").float()[:, None] / hh * 0.3 + 0.05) * mx
            t = ramp[None].expand(n, hh, ww) + torch.randint(-4, 5, (n, hh, ww), device="cuda", generator=g)
            return t.clamp(0, mx - 1).to(dt).contiguous()
        return torch.randint(0, mx, (n, hh, ww), dtype=torch.int32, device="cuda", generator=g).to(dt)
    return mk(h, w), mk(h // sy, w // sx), mk(h // sy, w // sx)


def time_device(h, w, hh, depth, sx, sy, batch, natural=False, rounds=5):
    pool = 3
    sets = [planes(w, hh, depth, sx, sy, batch, natural) for _ in range(pool)]
    st = torch.cuda.current_stream().cuda_stream
    sz = 2 if depth > 8 else 1
    best = []
    for r in range(rounds + 1):
        steps = max(3, 48 // batch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            Y, U, V = sets[i % pool]
            h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, w, w // sx, batch,
                                   Y[0].numel() * sz, U[0].numel() * sz, st)
        e1.record()
        torch.cuda.synchronize()
        if r:
            best.append(e0.elapsed_time(e1) / steps / batch * 1e3)
    best.sort()
    return best[len(best) // 2]


def main():
    h = hw.VfgsHip(device=0)
    rows = []
    for name, w, hh, depth, (sx, sy), trace in CONFIGS:
        h.lib.vfgs_hip_reset_state()
        T.replay(h, T.load_trace(trace))
        samples = w * hh * (1 + 2 / (sx * sy))
        sz = 2 if depth > 8 else 1
        nbytes = 2 * sz * samples
        for batch in (1, 8):
            us = time_device(h, w, hh, depth, sx, sy, batch)
            rows.append({"config": name, "content": "uniform random", "frames_per_launch": batch, "us_per_frame": round(us, 2),
                         "GBps": round(nbytes / us / 1e3, 1), "Mpixels_per_s": round(w * hh / us, 1), "Msamples_per_s": round(samples / us, 1)})
            print(rows[-1], flush=True)
        if "4320p" in name:
            us = time_device(h, w, hh, depth, sx, sy, 8, natural=True)
            rows.append({"config": name, "content": "ramp + -4..4 LSB noise", "frames_per_launch": 8, "us_per_frame": round(us, 2),
                         "GBps": round(nbytes / us / 1e3, 1), "Mpixels_per_s": round(w * hh / us, 1), "Msamples_per_s": round(samples / us, 1)})
            print(rows[-1], flush=True)
    # host-pointer stripe path (H2D + kernel + D2H + sync per call), 4320p and 1080p, pageable and pinned host memory
    for name, w, hh in (("4320p 10b 420 fgs_sei", 7680, 4320), ("1080p 10b 420 fgs_sei", 1920, 1080)):
        h.lib.vfgs_hip_reset_state()
        T.replay(h, T.load_trace("fgs_sei_10_420"))
        for pinned in (False, True):
            mk = (lambda *s: torch.randint(0, 1024, s, dtype=torch.int16).pin_memory()) if pinned else (lambda *s: torch.randint(0, 1024, s, dtype=torch.int16))
            Y, U, V = mk(hh, w), mk(hh // 2, w // 2), mk(hh // 2, w // 2)
            ts = []
            for r in range(6):
                t0 = time.perf_counter()
                h.add_grain_stripe(Y.data_ptr(), U.data_ptr(), V.data_ptr(), 0, w, hh, w, w // 2)
                ts.append(time.perf_counter() - t0)
            ms = sorted(ts[1:])[len(ts[1:]) // 2] * 1e3
            rows.append({"config": name, "path": "vfgs_add_grain_stripe (host pointers, PCIe both ways)", "host_memory": "pinned" if pinned else "pageable",
                         "ms_per_frame": round(ms, 3), "Mpixels_per_s": round(w * hh / ms / 1e3, 1), "host_GBps_each_way": round(w * hh * 3 / ms / 1e6, 2)})
            print(rows[-1], flush=True)
    # line API (the drop-in call), 1080p
    h.lib.vfgs_hip_reset_state()
    T.replay(h, T.load_trace("fgs_sei_10_420"))
    import ctypes as C
    fs, _ = T.lcg_frames(1920, 1080, 10, 2, 2, 4)
    line = C.cast(h.lib.vfgs_add_grain_line, C.c_void_p)
    drive = T.oracle_lib().vfgs_oracle_drive_lines      # the frame loop of vfgs_main.c:664-682 in C (no per-line Python cost)
    for i, f in enumerate(fs):
        t0 = time.perf_counter()
        drive(line, C.c_void_p(f.Y.ctypes.data), C.c_void_p(f.U.ctypes.data), C.c_void_p(f.V.ctypes.data), f.width, f.height, f.stride, f.cstride, 2, 2)
        dt = time.perf_counter() - t0
        rows.append({"config": "1080p 10b 420 fgs_sei", "path": f"vfgs_add_grain_line x 1080 (drop-in loop), frame {i}", "ms_per_frame": round(dt * 1e3, 2), "us_per_line": round(dt / f.height * 1e6, 1)})
        print(rows[-1], flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "bench_matrix.json").write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
