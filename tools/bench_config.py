#!/usr/bin/env python3
"""Developer tool: ONE BASELINE.json configuration on one MI355X, device-resident frames, in place.
`python3 tools/bench_config.py --config 0..4 [--batch 8] [--steps 200]` prints one JSON line; run directly behind
`rocprofv3 --kernel-trace --stats -- python3 tools/bench_config.py ...` for the per-config kernel statistics
kept under profiles/ (tools/gpu_profiles.sh)."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402
import vfgs_testlib as T  # noqa: E402
from versatilefilmgrain_amd import hw  # noqa: E402

CONFIGS = [  # BASELINE.json configs[i]: name, w, h, depth, (subx, suby), trace, kernel
    ("1920x1080 10-bit 4:2:0 fgs_sei", 1920, 1080, 10, (2, 2), "fgs_sei_10_420", "grain_rw_kernel<10,2,2,...> (row walk; the one-pattern flags depend on the cfg)"),
    ("1920x1080 10-bit 4:2:0 fgs_sei_ff_test1", 1920, 1080, 10, (2, 2), "fgs_sei_ff_test1_10_420", "grain_rw_kernel<10,2,2,...> (row walk; the one-pattern flags depend on the cfg)"),
    ("3840x2160 10-bit 4:2:0 fgs_sei_ar_test1", 3840, 2160, 10, (2, 2), "fgs_sei_ar_test1_10_420", "grain_rw_kernel<10,2,2,...> (row walk; the one-pattern flags depend on the cfg)"),
    ("3840x2160 8-bit 4:4:4 fgs_afgs1_test1", 3840, 2160, 8, (1, 1), "fgs_afgs1_test1_8_444", "grain_rw_kernel<8,1,1,...> (row walk; the one-pattern flags depend on the cfg)"),
    ("7680x4320 10-bit 4:2:0 fgs_sei", 7680, 4320, 10, (2, 2), "fgs_sei_10_420", "grain_rw_kernel<10,2,2,...> (row walk; the one-pattern flags depend on the cfg)"),
    # not BASELINE configs: the everyday formats of the 8-bit paths
    ("3840x2160 8-bit 4:2:0 fgs_afgs1_test1", 3840, 2160, 8, (2, 2), "fgs_afgs1_test1_8_420", "grain_rw_kernel<8,2,2,...> (row walk; the one-pattern flags depend on the cfg)"),
    ("3840x2160 8-bit 4:2:0 fgs_sei", 3840, 2160, 8, (2, 2), "fgs_sei_8_420", "grain_rw_kernel<8,2,2,...> (row walk; the one-pattern flags depend on the cfg)"),
    ("3840x2160 10-bit 4:2:0 fgs_afgs1_test1", 3840, 2160, 10, (2, 2), "fgs_afgs1_test1_10_420", "grain_rw_kernel<10,2,2,...> (row walk; the one-pattern flags depend on the cfg)"),
    # experiments: every plane with luma geometry / a one-pattern luma image at the headline size
    ("7680x4320 10-bit 4:4:4 fgs_sei", 7680, 4320, 10, (1, 1), "fgs_sei_10_444", "grain_rw_kernel<10,1,1,...>"),
    ("7680x4320 10-bit 4:2:0 fgs_afgs1_test1", 7680, 4320, 10, (2, 2), "fgs_afgs1_test1_10_420", "grain_rw_kernel<10,2,2,true,true>"),
    # the mainstream SEI case: several luma patterns (general form) at 2160p
    ("3840x2160 10-bit 4:2:0 fgs_sei", 3840, 2160, 10, (2, 2), "fgs_sei_10_420", "grain_rw_kernel<10,2,2,...>"),
    # 11, 12: 4:4:4 with several chroma patterns (general-form chroma: the image that used to hold a CU to three workgroups)
    ("3840x2160 10-bit 4:4:4 fgs_sei_ff_test6", 3840, 2160, 10, (1, 1), "fgs_sei_ff_test6_10_444", "grain_rw_kernel<10,1,1,...>"),
    ("3840x2160 8-bit 4:4:4 fgs_sei_ff_test6", 3840, 2160, 8, (1, 1), "fgs_sei_ff_test6_8_444", "grain_rw_kernel<8,1,1,...>"),
    # 13 .. 15: rows of 1024 blocks (walked in two parts): one-pattern forms, one-pattern chroma under general-form luma
    ("16384x2160 10-bit 4:2:0 fgs_afgs1_test1", 16384, 2160, 10, (2, 2), "fgs_afgs1_test1_10_420", "grain_rw_kernel<10,2,2,false,true,true,true,false>"),
    ("16384x2160 8-bit 4:4:4 fgs_afgs1_test1", 16384, 2160, 8, (1, 1), "fgs_afgs1_test1_8_444", "grain_rw_kernel<8,1,1,false,true,true,true,false>"),
    ("16384x2160 10-bit 4:2:0 fgs_sei", 16384, 2160, 10, (2, 2), "fgs_sei_10_420", "grain_rw_kernel<10,2,2,false,false,true,true,false>"),
    # 16: one luma pattern over several chroma patterns at 4:2:0 (one-pattern luma, general-form chroma: 15 KB of LDS)
    ("3840x2160 10-bit 4:2:0 fgs_sei_ff_test6", 3840, 2160, 10, (2, 2), "fgs_sei_ff_test6_10_420", "grain_rw_kernel<10,2,2,false,true,false,false,false>"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--preroll-ms", type=float, default=150.0)
    ap.add_argument("--content", choices=["uniform", "ramp"], default="uniform",
                    help="uniform: random over the full code range (worst case for LUT / pattern divergence, SURVEY 8d); "
                         "ramp: smooth diagonal ramp + -4..+4 noise (natural-like, reported separately)")
    ap.add_argument("--mode", choices=["inplace", "copy", "copy8"], default="inplace",
                    help="inplace (default); copy: out of place, same depth; copy8: 10-bit in, 8-bit out (the CLI's --outdepth 8 fused into the store)")
    ap.add_argument("--overlap", action="store_true", help="time the calls inside one vfgs_hip_overlap_begin/_end region (independent frames, one per call)")
    ap.add_argument("--region-every", type=int, default=0, help="with --overlap: close and reopen the region every N calls (a join every N calls)")
    ap.add_argument("--streams", type=int, default=1, help="experiment: issue consecutive launches round-robin on this many streams (they may overlap)")
    ap.add_argument("--pool", type=int, default=0, help="buffer sets cycled through (default: enough for > 1.5 GB)")
    ap.add_argument("--single-alloc", action="store_true", help="each set is ONE allocation [Y frames | U frames | V frames] (bench.py's layout)")
    ap.add_argument("--list", action="store_true", help="every frame of a launch is an allocation of its own (shuffled order, odd gaps), handed over "
                                                        "as a list of plane pointers: vfgs_hip_add_grain_frame_list_dev (inplace mode)")
    ap.add_argument("--width", type=int, default=0, help="override the picture width (experiments)")
    ap.add_argument("--height", type=int, default=0, help="override the picture height (experiments)")
    ap.add_argument("--stride", type=int, default=0, help="luma row pitch in samples (default: the width; chroma: stride / subx), inplace mode only (experiments)")
    args = ap.parse_args()
    name, w, hh, depth, (sx, sy), trace, kernel = CONFIGS[args.config]
    if args.width or args.height:
        w, hh = args.width or w, args.height or hh
        name += " [size %dx%d]" % (w, hh)
    if args.stride:
        name += " [stride %d]" % args.stride
    import os
    if os.environ.get("VFGS_LIB"):
        hw.load(os.environ["VFGS_LIB"])      # a variant build (tools/gpu_variants.sh)
    h = hw.VfgsHip(device=0)
    T.replay(h, T.load_trace(trace))
    dt = torch.int16 if depth > 8 else torch.uint8
    sz = 2 if depth > 8 else 1
    g = torch.Generator(device="cuda").manual_seed(3)
    ws = args.stride or w          # allocated row length in samples
    assert ws >= w and (ws == w or args.mode == "inplace")
    frame_bytes = sz * (ws * hh + 2 * (ws // sx) * (hh // sy))
    pool = max(3, int(1.5e9 // (frame_bytes * args.batch)) + 1)       # cycle through > 1.5 GB: nothing is served by the Infinity Cache
    pool = min(pool, 64)
    if args.pool:
        pool = args.pool

    def mk(rows, cols):
        if args.content == "ramp":
            r = torch.arange(rows, device="cuda", dtype=torch.int32).view(1, rows, 1)
            c = torch.arange(cols, device="cuda", dtype=torch.int32).view(1, 1, cols)
            f = torch.arange(args.batch, device="cuda", dtype=torch.int32).view(args.batch, 1, 1)
            lo, hi = (16 << (depth - 8)), (235 << (depth - 8))
            base = lo + ((r * 3 + c * 2 + f * 37) >> 3) % (hi - lo)
            noise = torch.randint(-4, 5, (args.batch, rows, cols), dtype=torch.int32, device="cuda", generator=g)
            return (base + noise).clamp(0, (1 << depth) - 1).to(dt)
        return torch.randint(0, 1 << depth, (args.batch, rows, cols), dtype=torch.int32, device="cuda", generator=g).to(dt)
    if args.list:
        assert args.mode == "inplace" and not args.stride
        import random
        rnd = random.Random(5)
        sets, keep = [], []
        for _ in range(pool):
            planes = [None] * (3 * args.batch)
            order = list(range(3 * args.batch))
            rnd.shuffle(order)
            for k in order:
                rows, cols = (hh, w) if k % 3 == 0 else (hh // sy, w // sx)
                planes[k] = torch.randint(0, 1 << depth, (rows, cols), dtype=torch.int32, device="cuda", generator=g).to(dt)
                keep.append(torch.empty(rnd.randrange(1, 64) * 4096, dtype=torch.uint8, device="cuda"))      # a gap of odd size behind it
            sets.append(h.frame_list([(planes[3 * f].data_ptr(), planes[3 * f + 1].data_ptr(), planes[3 * f + 2].data_ptr()) for f in range(args.batch)]))
            keep.append(planes)
    elif args.single_alloc:
        sets = []
        for _ in range(pool):
            ny, nc = args.batch * hh * w, args.batch * (hh // sy) * (w // sx)
            b = torch.randint(0, 1 << depth, (ny + 2 * nc,), dtype=torch.int32, device="cuda", generator=g).to(dt)
            sets.append((b[:ny].view(args.batch, hh, w), b[ny:ny + nc].view(args.batch, hh // sy, w // sx), b[ny + nc:].view(args.batch, hh // sy, w // sx)))
    else:
        sets = [(mk(hh, ws), mk(hh // sy, ws // sx), mk(hh // sy, ws // sx)) for _ in range(pool)]
    st = torch.cuda.current_stream().cuda_stream
    extra_streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else []
    stream_of = (lambda i: extra_streams[i % len(extra_streams)].cuda_stream) if extra_streams else (lambda i: st)

    dsts = None
    if args.mode != "inplace":
        ddt = torch.uint8 if args.mode == "copy8" else dt
        dsts = [(torch.zeros((args.batch, hh, w), dtype=ddt, device="cuda"), torch.zeros((args.batch, hh // sy, w // sx), dtype=ddt, device="cuda"),
                 torch.zeros((args.batch, hh // sy, w // sx), dtype=ddt, device="cuda")) for _ in range(min(pool, 4))]

    def step(i):
        if args.list:
            h.add_grain_frame_list_dev(sets[i % pool], w, hh, w, w // sx, stream_of(i))
            return
        Y, U, V = sets[i % pool]
        if args.mode == "inplace":
            h.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), w, hh, ws, ws // sx, args.batch, Y[0].numel() * sz, U[0].numel() * sz, stream_of(i))
            return
        dY, dU, dV = dsts[i % len(dsts)]
        if args.mode == "copy":
            h.add_grain_copy_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), dY.data_ptr(), dU.data_ptr(), dV.data_ptr(), w, hh, 0, hh, w, w // sx,
                                 args.batch, Y[0].numel() * sz, U[0].numel() * sz, st)
        else:
            h.add_grain_copy8_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), dY.data_ptr(), dU.data_ptr(), dV.data_ptr(), w, hh, 0, hh, w, w // sx,
                                  w, w // sx, args.batch, Y[0].numel() * sz, U[0].numel() * sz, dY[0].numel(), dU[0].numel(), st)
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < args.preroll_ms:
        if args.overlap:
            h.overlap_begin(st)       # (the first region creates the two internal streams and their hardware queues: milliseconds, once)
        for _ in range(8):
            step(n)
            n += 1
        if args.overlap:
            h.overlap_end(st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    th0 = time.perf_counter()
    if args.overlap:
        h.overlap_begin(st)
    for i in range(args.steps):
        step(i)
        if args.overlap and args.region_every and (i + 1) % args.region_every == 0 and i + 1 < args.steps:
            h.overlap_end(st)
            h.overlap_begin(st)
    if args.overlap:
        h.overlap_end(st)
    host_us = (time.perf_counter() - th0) / args.steps * 1e6     # what the calling thread spends per call (it never waits here)
    if extra_streams:
        for xs in extra_streams:
            torch.cuda.current_stream().wait_stream(xs)
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) / args.steps * 1e3
    us = launch_us / args.batch
    samples = w * hh * (1 + 2 / (sx * sy))
    nbytes = (sz + (1 if args.mode == "copy8" else sz)) * samples
    info = h.last_launch_info()
    kernel = info["kernel"] if info else kernel
    print(json.dumps({"config": args.config, "workload": name, "content": args.content, "mode": args.mode, "frame_list": bool(args.list), "streams": args.streams, "overlap_region": bool(args.overlap), "kernel": kernel, "frames_per_launch": args.batch, "steps": args.steps,
                      "launch_us": round(launch_us, 2), "host_us_per_call": round(host_us, 2), "us_per_frame": round(us, 3), "algorithmic_bytes_per_frame": int(nbytes),
                      "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 4),
                      "Mpixels_per_s": round(w * hh / us, 1), "Msamples_per_s": round(samples / us, 1)}), flush=True)


if __name__ == "__main__":
    main()
